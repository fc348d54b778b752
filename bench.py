#!/usr/bin/env python3
"""bench.py -- Mbp of DNA decoded per second (BASELINE.json metric), ab-initio human model, on N MI355X.

One "step" = one pass of the hot path (prep -> trellis -> back-trace, all on the GPU) over one batch of synthetic
input that is already resident in HBM: BASELINE.json configs[2], 100 contigs x 1 Mbp of upper-case uniform-random
DNA per GPU (weak scaling: every rank decodes its own 100 contigs; contigs are independent, there is no data-path
collective).  Launch:  python bench.py [--gpus N --steps K --warmup W]   (N>1 via torch.distributed.run).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def synth_contigs(n_contigs, length, seed0):
    out = []
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n_contigs):
        rng = np.random.default_rng(seed0 + i)
        out.append(lut[rng.integers(0, 4, size=length, dtype=np.uint8)].tobytes())
    return out


def cpu_baseline(cfg, sample_bp):
    """The reference's own CPU path (oracle/_ref/augustus_ref, 1 thread) on a bounded sample of the same workload."""
    from helpers import REF_AUGUSTUS, write_fasta, twin_decode
    import tempfile
    seq = synth_contigs(1, sample_bp, 999)[0].decode()
    if os.path.exists(REF_AUGUSTUS):
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "s.fa")
            write_fasta(fa, [("sample", seq)])
            env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg)
            t0 = time.time()
            r = subprocess.run([REF_AUGUSTUS, "--species=human", fa], capture_output=True, env=env)
            dt = time.time() - t0
            if r.returncode == 0:
                return {"value": sample_bp / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "reference",
                        "sample": "1 contig x %d bp uniform-random DNA, --species=human, reference binary wall-clock incl. parameter load" % sample_bp}
    import augustus_amd as ax
    m = ax.Model(cfg, "human")
    t0 = time.time()
    twin_decode(m.tables_ptr, seq, m.n_states)
    dt = time.time() - t0
    return {"value": sample_bp / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
            "sample": "1 contig x %d bp uniform-random DNA, oracle/ghmm_twin.cc" % sample_bp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=100)
    ap.add_argument("--contig-len", type=int, default=1000000)
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches of --contigs contigs resident per GPU; consecutive steps alternate between them on separate "
                         "HIP streams, so the prep kernels of one step overlap the (100-workgroup) trellis kernel of the other")
    ap.add_argument("--cpu-sample-bp", type=int, default=300000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "AUGX_BENCH_DEVICE" in os.environ:  # (testing the multi-rank path on a box with fewer GPUs than ranks)
        local = int(os.environ["AUGX_BENCH_DEVICE"])
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AUGX_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")  # ("nccl" is RCCL)
        dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the decode path has no CPU fallback)")
    torch.cuda.set_device(local)

    import augustus_amd as ax
    from helpers import config_path
    cfg = config_path()
    model = ax.Model(cfg, "human")
    S = model.n_states
    bases = a.contigs * a.contig_len
    # one step = one pass of the decode path over one batch of --contigs contigs.  --inflight batches (different contigs) are
    # resident; step s runs on batch s % inflight, each batch on its own decoder (HIP stream) driven by its own host
    # thread, so that consecutive steps overlap the way consecutive batches of a genome do in the CLI driver.
    n_fl = max(1, min(a.inflight, a.steps))
    decs, batches = [], []
    for i in range(n_fl):
        d = ax.Decoder(model, local)
        try:  # H2D upload: inputs are resident in HBM before the timed region
            b = ax.Batch(d, synth_contigs(a.contigs, a.contig_len, 12345 + 1000 * rank + 17 * i))
            b.decode(sync=True)          # (first decode of a batch object sizes its candidate lists and buffer: untimed;
            b.decode(sync=True)          #  the second one gives back what the first estimate took too much)
        except ax.AugxError as e:
            if i == 0 or e.code not in (ax.AUGX_E_NOMEM, ax.AUGX_E_HIP):
                raise
            break                        # not enough free HBM for another resident batch: run with the ones we have
        decs.append(d); batches.append(b)
    n_fl = len(batches)

    def sync():
        for d in decs:
            ax._check(ax.lib().augx_batch_sync(d._h))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    import threading
    trellis_ms, prep_ms, back_ms = [], [], []
    lock = threading.Lock()
    todo = [0]

    def worker(batch, n_total, record):
        while True:
            with lock:
                if todo[0] >= n_total:
                    return
                todo[0] += 1
            batch.decode(sync=False)
            k = batch.kernel_ms()        # HIP events on the decoder's stream (also waits for the step)
            if record:
                with lock:
                    trellis_ms.append(k["trellis_ms"]); prep_ms.append(k["prep_ms"]); back_ms.append(k["backtrace_ms"])

    def run_steps(n_total, record):
        todo[0] = 0
        ts = [threading.Thread(target=worker, args=(b, n_total, record)) for b in batches]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    run_steps(a.warmup, False)
    sync()
    t0 = time.perf_counter()
    run_steps(a.steps, True)             # EXACTLY --steps steps
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # sanity: the decode produced feasible paths
    for b in batches:
        assert all(r.status == 0 for r in b.paths()), "decode failed"
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * bases * a.steps / dt / 1e6
        # roofline of the dominant kernel (trellis): algorithmic bytes per bp = 0.25 + 20*S (SURVEY.md 8d / DESIGN.md)
        alg_bytes = (0.25 + 20.0 * S) * bases
        tr_s = float(np.mean(trellis_ms)) / 1e3
        achieved = alg_bytes / tr_s / 1e9
        # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/ (FETCH_SIZE and
        # WRITE_SIZE in separate rocprofv3 runs, gfx950 correction applied; see profiles/r01_hbm_traffic.json)
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tj):
            with open(tj) as fh:
                traffic = json.load(fh)["kTrellis"]["traffic_bytes_per_bp"] * bases
        out = {
            "metric": "Mbp DNA decoded/sec (whole node), ab-initio human model",
            "value": value, "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic uniform-random DNA, %d contigs x %d bp per GPU, --species=human ab initio (47 states, sample=0)"
                                   % (a.contigs, a.contig_len), "pieces_in_flight_per_gpu": a.contigs * n_fl, "batches_in_flight_per_gpu": n_fl,
                       "sharding": "contigs sharded over ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/r01_hbm_traffic.json)", "kernel": "kTrellis", "kernel_ms": float(np.mean(trellis_ms)),
                         "prep_ms": float(np.mean(prep_ms)), "backtrace_ms": float(np.mean(back_ms)),
                         "positions_per_s_per_piece": a.contig_len / tr_s},
        }
        if not a.no_cpu_baseline and world == 1:  # the reference's CPU path, timed beside it (rank 0, N = 1 only)
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_sample_bp)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
