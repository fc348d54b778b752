#!/usr/bin/env python3
"""bench.py -- Mbp of DNA decoded per second (BASELINE.json metric), ab-initio human model, on N MI355X.

One "step" = one pass of the hot path (prep -> candidates -> trellis -> back-trace, all on the GPU) over one batch of
synthetic input that is already resident in HBM: BASELINE.json configs[2], 100 contigs x 1 Mbp of upper-case uniform-random
DNA.  `value` (the headline) is WEAK scaling: every rank decodes its own 100 contigs per step.  The same run also times the
STRONG-scaling form of config 3 as it is written (100 contigs in total, split over the ranks longest-first) and reports it
under "strong"; and, at N = 1, the end-to-end rates of the drop-in ("e2e": host buffers -> augx_decode_batch -> paths ->
genes -> GFF text; "cli": the augustus executable, FASTA file in, GFF out) and the reference's CPU path ("cpu_baseline").
Contigs are independent: there is no data-path collective, ranks only meet in the barriers around the timed regions.

Launch:  python bench.py [--gpus N --steps K --warmup W]
  N > 1: one process per GPU.  Under torch.distributed.run (RANK/WORLD_SIZE in the environment) this process is one rank;
  started plainly with --gpus N > 1 it re-launches itself under torch.distributed.run with N ranks on 127.0.0.1.
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED0 = 12345


def synth_contigs(n_contigs, length, seed0):
    out = []
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n_contigs):
        rng = np.random.default_rng(seed0 + i)
        out.append(lut[rng.integers(0, 4, size=length, dtype=np.uint8)].tobytes())
    return out


def synth_isochore_contigs(n_contigs, length, seed0):
    """contigs whose GC content changes in runs of 50-300 kb between 34 % and 62 % -- the isochores of a vertebrate genome in
    caricature: pieces then run through several of the model's GC-content classes (class steps inside a piece, several planes of
    the class-dependent arrays, the reference's snippet cache replayed), which uniform-random DNA never does"""
    out = []
    for i in range(n_contigs):
        rng = np.random.default_rng(seed0 + i)
        parts, total = [], 0
        while total < length:
            run = int(rng.integers(50000, 300000))
            gc = float(rng.choice([0.34, 0.40, 0.46, 0.52, 0.58, 0.62]))
            p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]  # A C G T
            parts.append(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=run, p=p))
            total += run
        out.append(np.concatenate(parts)[:length].tobytes())
    return out


def rank_contigs(mode, rank, world, n_contigs, length, batch=0):
    """The contigs rank `rank` of `world` decodes.
    weak:   its own n_contigs contigs (seeds disjoint from every other rank's and batch's)
    strong: its share of the SAME n_contigs contigs (seeds SEED0..SEED0+n-1), split longest-first over the ranks by the
            partition the product's multi-GPU path uses (augx_partition_lpt)"""
    if mode == "weak":
        return synth_contigs(n_contigs, length, SEED0 + 1000 * rank + 17 * batch)
    import augustus_amd as ax
    bins = ax.partition_lpt([length] * n_contigs, world)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i, b in enumerate(bins):
        if b == rank:
            rng = np.random.default_rng(SEED0 + i)
            out.append(lut[rng.integers(0, 4, size=length, dtype=np.uint8)].tobytes())
    return out


def max_over_ranks(dt, dist):
    if dist is None:
        return dt
    import torch
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


LINE_LIMIT = 5400   # the driver keeps an 8 KB tail of stdout: the JSON line must fit it with room to spare


def parse_timing(stderr_text):
    """The executable's AUGX_TIMING laps as {lap: seconds}; the per-batch lines (one per cut-finder round: hundreds on a long
    contig) are COUNTED, never kept -- round 3 printed 175 of them into the JSON line and the driver could not parse it."""
    laps, n_batches = {}, 0
    for line in stderr_text.splitlines():
        if not line.startswith("augx timing:"):
            continue
        body = line[len("augx timing:"):]
        if body.strip().startswith("cut finder: scout"):  # "cut finder: scout 0.123 s (115 tiles), 8 batches, 423 windows decoded, 174 used"
            import re
            mt = re.search(r"scout ([0-9.]+) s \((\d+) tiles\), (\d+) batches, (\d+) windows decoded, (\d+) used", body)
            if mt and int(mt.group(4)) > 0:  # (a run without a cut finder round says nothing here)
                laps["cut_scout_s"], laps["cut_batches"], laps["cut_windows"], laps["cut_windows_used"] = float(mt.group(1)), int(mt.group(3)), int(mt.group(4)), int(mt.group(5))
            continue
        if body.strip().startswith("near ties on the chosen paths"):  # "... : 3 cells in 2 decodes"
            import re
            mt = re.search(r": (\d+) cells in (\d+) decodes", body)
            if mt and int(mt.group(1)) > 0:
                laps["near_ties"] = int(mt.group(1))
            continue
        if body.startswith("   "):
            n_batches += 1
            continue
        w = body.rsplit(None, 2)
        try:
            laps[w[0].strip()] = float(w[1])
        except (IndexError, ValueError):
            pass
    if n_batches:
        laps["batches"] = n_batches
    return laps


def bounded_line(out, limit=LINE_LIMIT):
    """json.dumps(out) made to fit `limit` bytes: floats to 6 significant digits, then -- only while it is still too long -- the
    descriptive strings of the secondary legs go (longest first), then whole secondary legs (last added first).  The contract keys
    (metric, value, ..., config, roofline, cpu_baseline) are never touched.  The untrimmed object goes to gpurun_out/bench_full.json."""
    def rnd(x):
        if isinstance(x, float):
            return float("%.6g" % x)
        if isinstance(x, dict):
            return {k: rnd(v) for k, v in x.items()}
        if isinstance(x, list):
            return [rnd(v) for v in x]
        return x
    out = rnd(out)
    keep = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline"}
    def strings(o, path=()):
        if isinstance(o, dict):
            for k, v in o.items():
                if not path and k in keep:  # (the contract keys themselves -- "metric" is a long string -- stay whatever their length)
                    if isinstance(v, dict) and k == "cpu_baseline":
                        yield from strings(v, path + (k,))
                    continue
                if isinstance(v, str) and len(v) > 40 and (not path or path[0] not in keep or path[0] == "cpu_baseline" and len(path) > 1):
                    yield len(v), path + (k,)
                else:
                    yield from strings(v, path + (k,))
    while len(json.dumps(out)) >= limit:
        cand = sorted(strings(out), reverse=True)
        if cand:
            _, pth = cand[0]
            o = out
            for k in pth[:-1]:
                o = o[k]
            del o[pth[-1]]
            continue
        # then the lap tables of the executable's runs (largest first), then whole secondary legs -- the ones that carry a parity
        # flag against the reference's goldens (product) last
        laps = []
        def find_laps(o, path=()):
            if isinstance(o, dict):
                for k, v in o.items():
                    if k == "laps_s" and isinstance(v, dict):
                        laps.append((len(json.dumps(v)), path + (k,)))
                    else:
                        find_laps(v, path + (k,))
        find_laps(out)
        if laps:
            _, pth = max(laps)
            o = out
            for k in pth[:-1]:
                o = o[k]
            del o[pth[-1]]
            continue
        extra = [k for k in out if k not in keep]
        if not extra:
            break
        order = ["cli_sampled", "cli", "e2e", "two_batches_in_flight", "fasta_to_gff", "gc_steps", "strong", "timed_region", "utr", "product"]
        extra.sort(key=lambda k: order.index(k) if k in order else -1)
        del out[extra[0]]
    line = json.dumps(out)
    assert len(line) < limit, "bench line of %d bytes" % len(line)
    return line


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sample_bp, host_sample_bp):
    """The reference's own CPU path (oracle/_ref/augustus_ref, built from /root/reference by oracle/Makefile) on a bounded
    sample of the same workload: one process pinned to one core (the headline x-factor), and one pinned process per host core
    on disjoint contigs (the whole-host figure; the reference has no threading).  The parameter-load time (the same binary on
    a 2 kb record) is measured and subtracted."""
    from helpers import REF_AUGUSTUS, write_fasta, twin_decode
    cores = sorted(os.sched_getaffinity(0))
    if not os.path.exists(REF_AUGUSTUS):
        import augustus_amd as ax
        m = ax.Model(cfg, "human")
        seq = synth_contigs(1, min(sample_bp, 300000), 999)[0].decode()
        t0 = time.time()
        twin_decode(m.tables_ptr, seq, m.n_states)
        dt = time.time() - t0
        return {"value": len(seq) / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port", "cpu": cpu_model_string(),
                "sample": "1 contig x %d bp uniform-random DNA, oracle/ghmm_twin.cc (oracle/_ref not present)" % len(seq)}
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg)
    with tempfile.TemporaryDirectory() as d:
        def run(files, pin):
            t0 = time.time()
            ps = [subprocess.Popen(["taskset", "-c", str(c), REF_AUGUSTUS, "--species=human", f], stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, env=env) for f, c in zip(files, pin)]
            ok = all(p.wait() == 0 for p in ps)
            return time.time() - t0, ok
        tiny = os.path.join(d, "tiny.fa")
        write_fasta(tiny, [("tiny", synth_contigs(1, 2000, 998)[0].decode())])
        t_load, _ = run([tiny], cores[:1])
        one = os.path.join(d, "one.fa")
        write_fasta(one, [("sample", synth_contigs(1, sample_bp, 999)[0].decode())])
        t1, ok1 = run([one], cores[:1])
        out = {"value": sample_bp / 1e6 / max(t1 - t_load, 1e-9), "unit": "Mbp/s", "cores": 1, "kind": "reference", "cpu": cpu_model_string(),
               "param_load_s": t_load, "ok": ok1,
               "sample": "1 contig x %d bp uniform-random DNA, --species=human, reference binary pinned with taskset to one core, wall-clock minus "
                         "parameter load" % sample_bp}
        # the same with posterior sampling (--sample=100) on a shorter sample: the forward pass and 99 sampled paths
        smp_bp = min(sample_bp, 200000)
        smp = os.path.join(d, "smp.fa")
        write_fasta(smp, [("sample", synth_contigs(1, smp_bp, 999)[0].decode())])
        t0 = time.time()
        p = subprocess.Popen(["taskset", "-c", str(cores[0]), REF_AUGUSTUS, "--species=human", "--sample=100", smp], stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL, env=env)
        oks = p.wait() == 0
        ts = time.time() - t0
        out["sampled"] = {"value": smp_bp / 1e6 / max(ts - t_load, 1e-9), "unit": "Mbp/s", "cores": 1, "ok": oks,
                          "sample": "1 contig x %d bp, --species=human --sample=100, pinned to one core, wall-clock minus parameter load" % smp_bp}
        cores = cores[:16]  # (a bounded sample: boxes whose affinity mask shows hundreds of cores may grant far fewer)
        files = []
        for i, c in enumerate(cores):
            f = os.path.join(d, "h%d.fa" % i)
            write_fasta(f, [("h%d" % i, synth_contigs(1, host_sample_bp, 2000 + i)[0].decode())])
            files.append(f)
        tn, okn = run(files, cores)
        out["whole_host"] = {"value": len(cores) * host_sample_bp / 1e6 / max(tn - t_load, 1e-9), "unit": "Mbp/s", "cores": len(cores), "ok": okn,
                             "host_cores_visible": len(os.sched_getaffinity(0)),
                             "sample": "%d processes pinned (taskset) to %d distinct cores, each 1 contig x %d bp" % (len(cores), len(cores), host_sample_bp)}
        return out


def e2e_legs(cfg, model, local, contigs):
    """End-to-end rates of the drop-in on this rank's contigs (FASTA in -> GFF out, model load excluded)."""
    import augustus_amd as ax
    L = ax.lib()
    n = len(contigs)
    bases = sum(len(c) for c in contigs)
    out = {}
    # ---- e2e: host buffers -> augx_decode_batch (H2D, all kernels, path D2H + unpack) -> genes -> GFF text
    dec = ax.Decoder(model, local)
    names = ["rand%03d" % i for i in range(n)]

    class PR(ctypes.Structure):
        _fields_ = [("record", ctypes.c_int32), ("status", ctypes.c_int32), ("begin", ctypes.c_int64), ("end", ctypes.c_int64),
                    ("states", ctypes.c_void_p), ("n_states", ctypes.c_int32)]
    P = (ax._Piece * n)()
    for i, s in enumerate(contigs):
        P[i].seq, P[i].len, P[i].init_kind, P[i].term_kind = s, len(s), 0, 0
    paths = (ax._Path * n)()
    L.augx_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.augx_format_records.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]
    buf = ctypes.create_string_buffer(256 << 20)
    c_names = (ctypes.c_char_p * n)(*[x.encode() for x in names])
    c_seqs = (ctypes.c_char_p * n)(*contigs)
    c_lens = (ctypes.c_int64 * n)(*[len(c) for c in contigs])
    best = None
    for rep in range(2):  # (the first pass pays the one-time allocations of a fresh decoder: report the second)
        t0 = time.perf_counter()
        ax._check(L.augx_decode_batch(dec._h, P, n, paths))
        t1 = time.perf_counter()
        R = (PR * n)()
        for i in range(n):
            R[i].record, R[i].status, R[i].begin, R[i].end = i, paths[i].status, 0, len(contigs[i]) - 1
            R[i].states, R[i].n_states = ctypes.cast(paths[i].states, ctypes.c_void_p), paths[i].n_states
        ax._check(L.augx_format_records(model._h, n, c_names, c_seqs, c_lens, n, R, buf, 256 << 20))
        t2 = time.perf_counter()
        for i in range(n):
            L.augx_path_free(ctypes.byref(paths[i]))
        best = (t1 - t0, t2 - t1)
    out["e2e"] = {"value": bases / 1e6 / (best[0] + best[1]), "unit": "Mbp/s", "decode_s": best[0], "genes_gff_s": best[1],
                  "gff_bytes": len(buf.value),
                  "region": "host buffers -> augx_decode_batch (H2D + all kernels + path D2H) -> gene structures -> GFF text; model load excluded"}
    # ---- the same call on contigs with GC-content steps inside every piece (what real genomes look like to the model)
    iso = synth_isochore_contigs(n, len(contigs[0]), SEED0 + 4242)
    P2 = (ax._Piece * n)()
    for i, s in enumerate(iso):
        P2[i].seq, P2[i].len, P2[i].init_kind, P2[i].term_kind = s, len(s), 0, 0
    tb = None
    for rep in range(2):
        t0 = time.perf_counter()
        ax._check(L.augx_decode_batch(dec._h, P2, n, paths))
        tb = time.perf_counter() - t0
        for i in range(n):
            L.augx_path_free(ctypes.byref(paths[i]))
    out["gc_steps"] = {"value": sum(len(c) for c in iso) / 1e6 / tb, "unit": "Mbp/s", "decode_s": tb,
                       "workload": "%d contigs x %d bp whose GC content changes in runs of 50-300 kb between 34 %% and 62 %%: several GC-content "
                                   "classes per piece" % (n, len(iso[0])),
                       "region": "host buffers -> augx_decode_batch (H2D, classification incl. its host round trip, all kernels, the replay of the "
                                 "reference's snippet cache and the second trellis run it asks for, path D2H); second of two calls"}
    dec.close()
    # ---- cli: the augustus executable on the same contigs as a FASTA file (includes process start, parameter load, FASTA parse)
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "bench.fa")
            with open(fa, "wb") as f:
                for nm, s in zip(names, contigs):
                    f.write(b">" + nm.encode() + b"\n")
                    a = np.frombuffer(s, dtype=np.uint8)
                    k = len(a) // 60 * 60
                    rows = np.concatenate([a[:k].reshape(-1, 60), np.full((k // 60, 1), 10, dtype=np.uint8)], axis=1)
                    f.write(rows.tobytes())
                    if k < len(a):
                        f.write(a[k:].tobytes() + b"\n")
            env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg, AUGX_DEVICES=str(local) + ",", AUGX_TIMING="1")
            best = None
            time.sleep(4.0)  # (dec.close() above gave back what its pool held -- since round 6 up to a whole batch: the driver wipes it at ~40 GB/s)
            for rep2 in range(3):  # (the first run of the executable on a fresh box pages the library and the HIP runtime in; a run that starts
                                   #  while the driver still wipes freed device memory waits seconds in its first allocations: the fastest of three)
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--species=human", "--outfile=" + os.path.join(d, "out.gff"), fa], capture_output=True, env=env)
                dt = time.perf_counter() - t0
                laps = parse_timing(r.stderr.decode(errors="replace"))
                if best is None or dt < best[0]:
                    best = (dt, laps, r.returncode)
            dt, laps, rcode = best
            out["cli"] = {"value": bases / 1e6 / dt, "unit": "Mbp/s", "wall_s": dt, "returncode": rcode, "laps_s": laps,
                          "region": "augustus --species=human bench.fa (process start, parameter load, FASTA parse, decode on 1 GPU, GFF file written); the fastest of three runs (a run that starts while the driver still clears the device memory of the process before it waits seconds in its first allocations)"}
            # first byte of FASTA read -> last byte of GFF written, the one-time model create (parameter files, HIP
            # context, table upload) excluded: the laps of the executable's own clock (AUGX_TIMING)
            core = [laps.get(k2) for k2 in ("FASTA read", "cut finder", "decode of the pieces", "genes + GFF")]
            if all(x is not None for x in core):
                out["fasta_to_gff"] = {"value": bases / 1e6 / sum(core), "unit": "Mbp/s", "seconds": sum(core),
                                       "region": "FASTA file read + cut finder + decode of all pieces on 1 GPU + gene structures + GFF file written; "
                                                 "excluded: process start, parameter load, HIP context / decoder create, teardown (laps_s of cli)"}
            # ---- the same executable with posterior sampling (the default of 162 of the reference's species): forward algorithm on the
            #      device, 99 sampled paths per contig on the host, posterior probabilities in the GFF
            ns = n  # (round 3 timed 32 of the contigs; with 100 the decoder takes two batches, 64 + 36 Mbp: the forward matrix halves what fits)
            fa2 = os.path.join(d, "bench_s.fa")
            with open(fa2, "wb") as f:
                for nm, s in list(zip(names, contigs))[:ns]:
                    f.write(b">" + nm.encode() + b"\n" + s + b"\n")
            best = None
            for rep2 in range(2):  # (the faster of two runs, as for `cli`: the box is one GPU of a shared 8-GPU node, and a run of 6 s was seen to take 10)
                # (a process that starts while the driver still clears the 100+ GB of HBM the process before it gave back waits in its
                #  first allocations -- "decode + paths" of the first batch 3.2 s instead of 0.19 s: a pause outside the timed region)
                time.sleep(4.0)
                t0 = time.perf_counter()
                r2 = subprocess.run([exe, "--species=human", "--sample=100", "--outfile=" + os.path.join(d, "out_s.gff"), fa2], capture_output=True, env=env)
                dt2 = time.perf_counter() - t0
                if best is None or (r2.returncode == 0 and dt2 < best[0]) or best[1].returncode != 0:
                    best = (dt2, r2)
            dt, r = best
            b2 = sum(len(c) for c in contigs[:ns])
            laps = parse_timing(r.stderr.decode(errors="replace"))
            try:  # (the per-batch lines of the run, for whoever reads the number: gpurun_out/ is scratch)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "cli_sampled.err"), "wb") as f:
                    f.write(r.stderr)
            except OSError:
                pass
            out["cli_sampled"] = {"value": b2 / 1e6 / dt, "unit": "Mbp/s", "wall_s": dt, "returncode": r.returncode, "contigs": ns, "laps_s": laps,
                                  "region": "augustus --species=human --sample=100 (Viterbi + forward on 1 GPU, 99 sampled paths per contig on the host, "
                                            "posterior probabilities in the GFF); the faster of two runs"}
    return out


def product_leg(cfg, a, n_dev):
    """The PRODUCT's multi-GPU path (one process, one decoder and host thread per device: augx_main / augx_decode_sharded), which
    is what a user of the drop-in runs -- beside the rank-per-GPU legs above.  Two inputs: the 100-contig FASTA of config 3, and ONE
    contig of 23 Mbp under the fly model (200 kb pieces: a serial chain of ~115 cut-finder rounds, then the pieces over all
    devices; SURVEY.md F10, BASELINE config 2's shape)."""
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    out = {"devices": n_dev}
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        def write(fa, names, seqs):
            with open(fa, "wb") as f:
                for nm, sq in zip(names, seqs):
                    f.write(b">" + nm.encode() + b"\n")
                    arr = np.frombuffer(sq, dtype=np.uint8)
                    k = len(arr) // 60 * 60
                    f.write(np.concatenate([arr[:k].reshape(-1, 60), np.full((k // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
                    if k < len(arr):
                        f.write(arr[k:].tobytes() + b"\n")

        def run(args, fa, bases, reps=2, golden=None):
            """golden: key of tests/golden/golden_long.json -- the run is made with --progress=true and its cut points + GFF are
            compared (sha256) with what the REFERENCE binary printed for the same input (tests/golden/make_golden_long.py)"""
            env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg, AUGX_DEVICES=str(n_dev), AUGX_TIMING="1")
            best = None
            want = None
            if golden:
                try:
                    want = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_long.json"))).get(golden)
                except OSError:
                    want = None
            for _ in range(reps):
                t0 = time.perf_counter()
                r = subprocess.run([exe] + args + (["--progress=true"] if want else []) + ["--outfile=" + os.path.join(d, "o.gff"), fa], capture_output=True, env=env)
                dt = time.perf_counter() - t0
                err = r.stderr.decode(errors="replace")
                if golden:  # (developer aid: the executable's own phase and per-batch timing lines of the full-size runs)
                    try:
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                        with open(os.path.join(ROOT, "gpurun_out", "timing_%s.txt" % golden), "w") as fh:
                            fh.write("\n".join(l for l in err.splitlines() if l.startswith("augx timing")) + "\n")
                    except OSError:
                        pass
                laps = parse_timing(err)
                parity = None
                if want and r.returncode == 0:
                    import hashlib
                    from helpers import gff_body
                    txt = "\n".join([l for l in err.splitlines() if l.startswith("examining piece")] + gff_body(open(os.path.join(d, "o.gff")).read())) + "\n"
                    parity = hashlib.sha256(txt.encode()).hexdigest() == want["sha256"]
                if best is None or dt < best["wall_s"] or parity is False:
                    best = {"value": bases / 1e6 / dt, "unit": "Mbp/s", "wall_s": dt, "returncode": r.returncode, "laps_s": laps}
                    if want:
                        best["parity"] = parity
                        best["parity_against"] = "reference binary's cut points + GFF on the same %d bp contig (tests/golden/golden_long.json: %s)" % (bases, golden)
                if parity is False:
                    break
            return best
        contigs = synth_contigs(a.contigs, a.contig_len, SEED0)
        fa = os.path.join(d, "c3.fa")
        write(fa, ["rand%03d" % i for i in range(len(contigs))], contigs)
        out["contigs"] = run(["--species=human"], fa, sum(len(c) for c in contigs))
        out["contigs"]["workload"] = "%d contigs x %d bp, --species=human (config 3), %d device(s) in one process" % (a.contigs, a.contig_len, n_dev)
        big = synth_contigs(1, a.long_contig_len, SEED0 + 77)
        fb = os.path.join(d, "long.fa")
        write(fb, ["long"], big)
        rounds = (a.long_contig_len + 199999) // 200000
        from helpers import REF_AUGUSTUS
        core = sorted(os.sched_getaffinity(0))[0]

        def ref_rate(flags, nbp):
            """the reference binary on the first nbp bases of the same contig (> one 200 kb piece: a cut-finder round included),
            pinned to one core, parameter load (the same call on 2 kb) subtracted"""
            env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg)
            def t(fa):
                t0 = time.time()
                ok = subprocess.call(["taskset", "-c", str(core), REF_AUGUSTUS] + flags + [fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env) == 0
                return time.time() - t0, ok
            tiny, smp = os.path.join(d, "tiny.fa"), os.path.join(d, "smp.fa")
            write(tiny, ["tiny"], [big[0][:2000]])
            write(smp, ["smp"], [big[0][:nbp]])
            t_load, _ = t(tiny)
            t1, ok = t(smp)
            return {"value": nbp / 1e6 / max(t1 - t_load, 1e-9), "unit": "Mbp/s", "cores": 1, "kind": "reference", "ok": ok, "sample_bp": nbp}
        for key, flags, nbp in (("long_contig", ["--species=fly", "--UTR=off", "--sample=0", "--softmasking=0"], 400000),
                                ("long_contig_utr", ["--species=fly", "--UTR=on", "--sample=0", "--softmasking=0"], 250000)):
            if key == "long_contig_utr" and a.no_long_utr:
                continue
            r2 = run(flags, fb, a.long_contig_len, reps=2 if key == "long_contig" else 1,
                     golden=({"long_contig": "long", "long_contig_utr": "long_utr"}[key] if a.long_contig_len == 23000000 else None))
            r2["workload"] = "1 contig x %d bp uniform-random, %s (200 kb pieces, ~%d cut-finder rounds), %d device(s)" % (a.long_contig_len, " ".join(flags), rounds, n_dev)
            if not a.no_cpu_baseline and os.path.exists(REF_AUGUSTUS):
                r2["cpu_baseline"] = ref_rate(flags, nbp)
                r2["x_cpu"] = r2["value"] / r2["cpu_baseline"]["value"]
            out[key] = r2
        if not a.no_genome_like:
            out["genome_like"] = genome_like_leg(cfg, a, n_dev, d, exe, run)
        if not a.no_genome_1g:
            out["genome_1g"] = genome_1g_leg(cfg, a, n_dev, d, exe, run)
    return out


def genome_1g_leg(cfg, a, n_dev, d, exe, run):
    """BASELINE config 5 AT SIZE: the 1.0 Gbp stand-in (tests/golden/make_golden_long.py: genome_1g_records -- the genome_like recipe
    scaled to six chromosome-scale records, the first of 250 Mbp like GRCh38's chr1) through the executable at --species=human default
    flags: rate, number of device batches, HBM high-water and host peak RSS; and the 250 Mbp record ALONE with its cut points + GFF
    compared (sha256) with the reference binary's golden (tests/golden/golden_long.json: genome_1g_chr1)."""
    import resource
    import threading
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from helpers import read_fasta
    from make_golden_long import genome_1g_records
    if not os.path.exists(os.path.join(d, "genome.fa")):
        import tarfile
        with tarfile.open(os.path.join(ROOT, "tests", "golden", "big_inputs.tar.gz")) as t:
            t.extractall(d)
    g = read_fasta(os.path.join(d, "genome.fa"))[0][1]

    def write(fa, recs):
        with open(fa, "wb") as f:
            for nm, sq in recs:
                f.write(b">" + nm.encode() + b"\n")
                arr = np.frombuffer(sq.encode(), dtype=np.uint8)
                k = len(arr) // 60 * 60
                f.write(np.concatenate([arr[:k].reshape(-1, 60), np.full((k // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
                if k < len(arr):
                    f.write(arr[k:].tobytes() + b"\n")
    recs = genome_1g_records(g)
    bases, n_bases = sum(len(s) for _, s in recs), sum(s.count("N") for _, s in recs)
    fa, f1 = os.path.join(d, "genome_1g.fa"), os.path.join(d, "genome_1g_chr1.fa")
    write(fa, recs)
    write(f1, recs[:1])
    n1 = len(recs[0][1])
    nrec = len(recs)
    del recs
    free0 = [torch.cuda.mem_get_info(i)[0] for i in range(n_dev)]
    low = list(free0)
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            for i in range(n_dev):
                low[i] = min(low[i], torch.cuda.mem_get_info(i)[0])
            stop.wait(0.05)
    th = threading.Thread(target=watch)
    th.start()
    try:
        time.sleep(8.0)  # (as before the genome_like leg: the driver still wipes what the process before gave back)
        r = run(["--species=human"], fa, bases, reps=1)
        time.sleep(8.0)
        r1 = run(["--species=human"], f1, n1, reps=1, golden="genome_1g_chr1")
    finally:
        stop.set()
        th.join()
    r["workload"] = "%d records, %.2f Gbp (%.1f Mbp of N; the longest record %d Mbp), --species=human default flags, %d device(s)" % (nrec, bases / 1e9, n_bases / 1e6, n1 // 1000000, n_dev)
    r["hbm_high_water_gb"] = max(f - l for f, l in zip(free0, low)) / 1e9
    r["host_peak_rss_gb"] = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
    r["chr1_alone"] = r1
    return r


def genome_like_leg(cfg, a, n_dev, d, exe, run):
    """BASELINE config 5 in shape: the ~100 Mbp GRCh38-like stand-in (tests/golden/make_golden_long.py: genome_like_big_records --
    chromosome-scale records of real soft-masked DNA tiled in both orientations, isochores, megabase N runs, scaffolds) through the
    executable at --species=human default flags, FASTA file -> GFF file; cut points + GFF compared with the reference binary's
    golden; host peak RSS of the process and the HBM high-water mark (device-wide, sampled every 50 ms) beside the rate."""
    import resource
    import tarfile
    import threading
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from helpers import read_fasta, write_fasta
    from make_golden_long import genome_like_big_records
    with tarfile.open(os.path.join(ROOT, "tests", "golden", "big_inputs.tar.gz")) as t:
        t.extractall(d)
    recs = genome_like_big_records(read_fasta(os.path.join(d, "genome.fa"))[0][1])
    fa = os.path.join(d, "genome_like_big.fa")
    write_fasta(fa, recs)
    bases = sum(len(s) for _, s in recs)
    free0 = [torch.cuda.mem_get_info(i)[0] for i in range(n_dev)]
    low = list(free0)
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            for i in range(n_dev):
                low[i] = min(low[i], torch.cuda.mem_get_info(i)[0])
            stop.wait(0.05)
    th = threading.Thread(target=watch)
    th.start()
    rss0 = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    # (a process that starts while the driver still clears the 100+ GB of HBM its predecessors gave back waits seconds in its first
    #  allocations -- 4 s measured in ensureArrays here, 2 ms on a quiet device: a pause, as before the sampled leg)
    time.sleep(8.0)
    try:
        r = run(["--species=human"], fa, bases, reps=1, golden="genome_like_big")
    finally:
        stop.set()
        th.join()
    r["workload"] = "%d records, %.1f Mbp (%.1f Mbp of N), --species=human default flags, %d device(s)" % (len(recs), bases / 1e6, sum(s.count("N") for _, s in recs) / 1e6, n_dev)
    r["hbm_high_water_gb"] = max(f - l for f, l in zip(free0, low)) / 1e9
    r["host_peak_rss_gb"] = max(resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss, rss0) / 1e6  # (largest child so far: this run is the largest input)
    return r


def utr_leg(cfg, local, a):
    """BASELINE config 4's model on the bench's data: --species=human --UTR=on, the 71-state trellis with untranslated regions
    (dense kernels, device/dense.h: one workgroup per piece, ln V dense in HBM), synthetic uniform-random contigs resident in HBM,
    EXACTLY --steps timed decodes; roofline with the algorithmic 0.25 + 20 * 71 = 1420 B/bp; the reference binary with --UTR=on
    timed beside it on one pinned core (a bounded sample)."""
    import augustus_amd as ax
    import torch
    from helpers import REF_AUGUSTUS, write_fasta
    model = ax.Model(cfg, "human", UTR="on")
    S = model.n_states
    d = ax.Decoder(model, local)
    seqs = synth_contigs(a.utr_contigs, a.utr_contig_len, SEED0 + 555)
    bases = sum(len(x) for x in seqs)
    b = ax.Batch(d, seqs)
    b.decode(sync=True)
    b.decode(sync=True)
    for _ in range(a.warmup):
        b.decode(sync=True)
    torch.cuda.synchronize()
    ms = {"trellis": [], "prep": [], "back": []}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b.decode(sync=False)
        k = b.kernel_ms()
        ms["trellis"].append(k["trellis_ms"]); ms["prep"].append(k["prep_ms"]); ms["back"].append(k["backtrace_ms"])
    ax._check(ax.lib().augx_batch_sync(d._h))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(r.status == 0 for r in b.paths()), "UTR decode failed"
    b.close()
    # the same with FEW pieces (a genome of 24 contigs): the dense kernels run one workgroup per piece, so 24 pieces use 24 of the 256 compute
    # units -- the rate a batch of few pieces sees, next to the one that fills the chip
    few_n = min(24, a.utr_contigs)
    bf = ax.Batch(d, seqs[:few_n])
    bf.decode(sync=True); bf.decode(sync=True)
    kf = []
    tf0 = time.perf_counter()
    for _ in range(min(a.steps, 4)):
        bf.decode(sync=False)
        kf.append(bf.kernel_ms()["trellis_ms"])
    ax._check(ax.lib().augx_batch_sync(d._h))
    tfd = time.perf_counter() - tf0
    few = {"value": few_n * a.utr_contig_len * len(kf) / tfd / 1e6, "unit": "Mbp/s", "pieces": few_n, "kernel_ms": float(np.mean(kf))}
    bf.close(); d.close()
    tr_s = float(np.mean(ms["trellis"])) / 1e3
    achieved = (0.25 + 20.0 * S) * bases / tr_s / 1e9
    # HBM bytes of the kernel from the PMC passes of profiles/run_pmc.sh, taken with this very tree (as for the headline kernel)
    traffic, tsrc = None, "no profiles/*_utr_hbm_traffic.json was taken with this source tree: run profiles/run_pmc.sh"
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from source_sha import source_sha
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("_utr_hbm_traffic.json"):
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                tj = json.load(fh)
            kd = [v for k, v in tj.get("kernels", {}).items() if re.match(r"kDense<\d+, 0[,>]", k)]  # (the Viterbi pass: kDense<BLK, 0, TIES>)
            if tj.get("source_sha") == source_sha() and kd:
                traffic, tsrc = kd[0]["traffic_bytes_per_bp"] * bases, "profiles/" + name
                break
    out = {"value": bases * a.steps / dt / 1e6, "unit": "Mbp/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps,
           "config": {"workload": "synthetic uniform-random DNA, %d contigs x %d bp, --species=human --UTR=on ab initio (%d states, sample=0)"
                                  % (a.utr_contigs, a.utr_contig_len, S)},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                        "traffic_unit": "bytes per launch (PMC, %s)" % tsrc,
                        "kernel": "kDense<4,0>", "kernel_ms": tr_s * 1e3, "prep_ms": float(np.mean(ms["prep"])), "backtrace_ms": float(np.mean(ms["back"])),
                        "positions_per_s_per_piece": a.utr_contig_len / tr_s},
           "few_pieces": few}
    if not a.no_cpu_baseline and os.path.exists(REF_AUGUSTUS):
        env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg)
        core = sorted(os.sched_getaffinity(0))[0]
        with tempfile.TemporaryDirectory() as dd:
            def run(fa):
                t1 = time.time()
                ok = subprocess.call(["taskset", "-c", str(core), REF_AUGUSTUS, "--species=human", "--UTR=on", fa], stdout=subprocess.DEVNULL,
                                     stderr=subprocess.DEVNULL, env=env) == 0
                return time.time() - t1, ok
            tiny = os.path.join(dd, "tiny.fa")
            write_fasta(tiny, [("tiny", synth_contigs(1, 2000, 998)[0].decode())])
            t_load, _ = run(tiny)
            one = os.path.join(dd, "one.fa")
            nbp = min(a.cpu_sample_bp, 300000)
            write_fasta(one, [("sample", synth_contigs(1, nbp, 999)[0].decode())])
            t1, ok = run(one)
            out["cpu_baseline"] = {"value": nbp / 1e6 / max(t1 - t_load, 1e-9), "unit": "Mbp/s", "cores": 1, "kind": "reference", "ok": ok,
                                   "sample": "1 contig x %d bp uniform-random DNA, --species=human --UTR=on, reference binary pinned to one core, "
                                             "wall-clock minus parameter load" % nbp}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=100)
    ap.add_argument("--contig-len", type=int, default=1000000)
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches of --contigs contigs resident per GPU (default 1: the trellis cuts the pieces into segments and fills "
                         "the chip with one batch).  With 2, consecutive steps alternate between two batches on separate HIP streams, "
                         "the prep kernels of one overlapping the trellis of the other; that figure is reported as well at N = 1")
    ap.add_argument("--no-two-batches", action="store_true")
    ap.add_argument("--share", type=int, default=0, help="decoders per device the segment planner assumes (default: the resident batches)")
    ap.add_argument("--cpu-sample-bp", type=int, default=1000000)
    ap.add_argument("--cpu-host-sample-bp", type=int, default=300000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-strong", action="store_true")
    ap.add_argument("--no-product", action="store_true",
                    help="do not time the product's own multi-GPU path (the executable over --gpus devices in ONE process) on the 100-contig "
                         "FASTA and on one 23 Mbp contig at the fly model's 200 kb pieces")
    ap.add_argument("--long-contig-len", type=int, default=23000000)
    ap.add_argument("--no-long-utr", action="store_true", help="skip the 23 Mbp contig with --UTR=on (BASELINE config 4's shape)")
    ap.add_argument("--no-genome-like", action="store_true", help="skip the ~100 Mbp GRCh38-shaped genome through the executable (BASELINE config 5's shape)")
    ap.add_argument("--no-genome-1g", action="store_true", help="skip the 1.0 Gbp genome through the executable (BASELINE config 5 at size; ~40 s with building the input)")
    ap.add_argument("--no-utr", action="store_true", help="skip the --UTR=on leg (the 71-state model, BASELINE config 4's trellis)")
    ap.add_argument("--utr-contigs", type=int, default=256)
    ap.add_argument("--utr-contig-len", type=int, default=160000)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: become N ranks, one per GPU (the driver normally launches torch.distributed.run itself)
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "AUGX_BENCH_DEVICE" in os.environ:  # (testing the multi-rank path on a box with fewer GPUs than ranks)
        local = int(os.environ["AUGX_BENCH_DEVICE"])
    dist = None
    if world > 1 or "RANK" in os.environ:  # (under torch.distributed.run also as the only rank: one code path for N = 1 ... 8)
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
    import threading

    def sync_all(decs):
        for d in decs:
            ax._check(ax.lib().augx_batch_sync(d._h))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed_phase(mode, n_fl_want):
        """Resident batches for `mode`, --warmup untimed steps, EXACTLY --steps timed steps bracketed by barrier + synchronize,
        max over ranks.  One step = one decode of one batch; step s runs on batch s % inflight, each batch on its own decoder
        (HIP stream) driven by its own host thread, so consecutive steps overlap like consecutive batches of a genome."""
        decs, batches = [], []
        n_dec = max(1, min(n_fl_want, a.steps))
        for i in range(n_dec):
            d = ax.Decoder(model, local)
            d.set_share(a.share if a.share > 0 else 1)   # (measured: planning every batch's segments for the whole chip is as good as or better than for a share of it)
            try:  # H2D upload: inputs are resident in HBM before the timed region
                seqs = rank_contigs(mode, rank, world, a.contigs, a.contig_len, batch=i)
                if not seqs:
                    break
                b = ax.Batch(d, seqs)
                b.decode(sync=True)      # (first decode of a batch object sizes its candidate lists and buffer: untimed;
                b.decode(sync=True)      #  the second one gives back what the first estimate took too much)
            except ax.AugxError as e:
                if i == 0 or e.code not in (ax.AUGX_E_NOMEM, ax.AUGX_E_HIP):
                    raise
                break                    # not enough free HBM for another resident batch: run with the ones we have
            decs.append(d); batches.append(b)
        ms = {"trellis": [], "prep": [], "back": []}
        lock = threading.Lock()
        todo = [0]

        def worker(batch, n_total, record):
            while True:
                with lock:
                    if todo[0] >= n_total:
                        return
                    todo[0] += 1
                batch.decode(sync=False)
                k = batch.kernel_ms()    # HIP events on the decoder's stream (also waits for the step)
                if record:
                    with lock:
                        ms["trellis"].append(k["trellis_ms"]); ms["prep"].append(k["prep_ms"]); ms["back"].append(k["backtrace_ms"])

        def run_steps(n_total, record):
            todo[0] = 0
            ts = [threading.Thread(target=worker, args=(b, n_total, record)) for b in batches]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

        run_steps(a.warmup, False)
        sync_all(decs)
        t0 = time.perf_counter()
        run_steps(a.steps, True)         # EXACTLY --steps steps
        sync_all(decs)
        dt = max_over_ranks(time.perf_counter() - t0, dist)
        for b in batches:                # sanity: the decode produced feasible paths
            assert all(r.status == 0 for r in b.paths()), "decode failed"
        bases_rank = [sum(b.lens) for b in batches]
        info = {"dt": dt, "n_fl": len(batches), "bases_per_step_rank": bases_rank[0] if bases_rank else 0,
                "trellis_ms": float(np.mean(ms["trellis"])) if ms["trellis"] else 0.0,
                "prep_ms": float(np.mean(ms["prep"])) if ms["prep"] else 0.0, "back_ms": float(np.mean(ms["back"])) if ms["back"] else 0.0}
        for b in batches:
            b.close()
        for d in decs:
            d.close()
        return info

    bases = a.contigs * a.contig_len
    weak = timed_phase("weak", a.inflight)
    two = None
    if world == 1 and a.inflight == 1 and not a.no_two_batches and a.steps >= 2:
        two = timed_phase("weak", 2)   # (reported beside the headline: round 1's configuration, 2 x the HBM footprint)
    strong = None
    if world > 1 and not a.no_strong:
        strong = timed_phase("strong", a.inflight)
    if rank == 0:
        ms_per_step = weak["dt"] / a.steps * 1e3
        value = world * bases * a.steps / weak["dt"] / 1e6
        # roofline of the dominant kernel (trellis): algorithmic bytes per bp = 0.25 + 20*S (SURVEY.md 8d / DESIGN.md)
        alg_bytes = (0.25 + 20.0 * S) * bases
        tr_s = weak["trellis_ms"] / 1e3
        achieved = alg_bytes / tr_s / 1e9
        # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/ (FETCH_SIZE and
        # WRITE_SIZE in separate rocprofv3 runs, gfx950 correction applied)
        # -- quoted only from a profile taken with THIS tree (profiles/source_sha.py; the PMC passes are separate rocprofv3 runs,
        # profiles/run_pmc.sh): a stale file is reported as such, never read silently
        traffic, tsrc = None, None
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from source_sha import source_sha
        sha = source_sha()
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if not name.endswith("_hbm_traffic.json"):
                continue
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                tj = json.load(fh)
            if tj.get("source_sha") == sha and "kTrellis" in tj:
                traffic = tj["kTrellis"]["traffic_bytes_per_bp"] * bases
                tsrc = "profiles/" + name
                break
        if traffic is None:
            tsrc = "no profiles/*_hbm_traffic.json was taken with this source tree (source_sha %s): run profiles/run_pmc.sh" % sha
        out = {
            "metric": "Mbp DNA decoded/sec (whole node), ab-initio human model",
            "value": value, "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "timed_region": "device pipeline (prep, candidates, trellis, back-trace) on HBM-resident input; see e2e / cli for the drop-in end to end",
            "config": {"workload": "synthetic uniform-random DNA, %d contigs x %d bp per GPU, --species=human ab initio (47 states, sample=0)"
                                   % (a.contigs, a.contig_len), "pieces_in_flight_per_gpu": a.contigs * weak["n_fl"], "batches_in_flight_per_gpu": weak["n_fl"],
                       "sharding": "contigs sharded over ranks (one process per GPU), no data-path collective",
                       "note": "the timed steps re-decode batches that stay resident in HBM and lie in one GC class each (uniform-random DNA): no upload, no class steps, no GFF; e2e / fasta_to_gff / product are the legs with those"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, %s)" % tsrc, "kernel": "kTrellis", "kernel_ms": weak["trellis_ms"],
                         "prep_ms": weak["prep_ms"], "backtrace_ms": weak["back_ms"],
                         "positions_per_s_per_piece": a.contig_len / tr_s},
        }
        if two is not None and two["n_fl"] == 2:
            out["two_batches_in_flight"] = {"value": bases * a.steps / two["dt"] / 1e6, "unit": "Mbp/s", "ms_per_step": two["dt"] / a.steps * 1e3,
                                            "kernel_ms": two["trellis_ms"], "prep_ms": two["prep_ms"],
                                            "note": "steps alternate between two resident batches on two HIP streams: the prep kernels of one run beside the trellis passes of the other"}
        if strong is not None:
            out["strong"] = {"value": bases * a.steps / strong["dt"] / 1e6, "unit": "Mbp/s", "ms_per_step": strong["dt"] / a.steps * 1e3,
                             "workload": "BASELINE config 3 as written: the same %d contigs x %d bp in total, split longest-first over %d ranks"
                                         % (a.contigs, a.contig_len, world), "bases_per_step_rank0": strong["bases_per_step_rank"],
                             "trellis_ms": strong["trellis_ms"], "prep_ms": strong["prep_ms"]}
        if world == 1:
            if not a.no_e2e:
                out.update(e2e_legs(cfg, model, local, rank_contigs("weak", 0, 1, a.contigs, a.contig_len)))
            if not a.no_cpu_baseline:  # the reference's CPU path, timed beside it (rank 0, N = 1 only)
                out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_sample_bp, a.cpu_host_sample_bp)
            if not a.no_utr:
                out["utr"] = utr_leg(cfg, local, a)
        if not a.no_product:  # (rank 0 drives every device from one process, as the executable does; the other ranks wait in the barrier)
            out["product"] = product_leg(cfg, a, world)
        try:  # the whole object, untrimmed, beside the line (gpurun_out/ is merged back from the GPU box)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError:
            pass
        print(bounded_line(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
