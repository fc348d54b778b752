"""The multi-GPU product path on CPU (no device needed): the partition of pieces over devices, the ordered gather with global
gene numbering (augx_format_records), and the rank sharding + max-over-ranks timing of bench.py in a world-size-2 gloo run."""
import ctypes
import json
import os
import random
import subprocess
import sys
import textwrap

import numpy as np

import augustus_amd as ax
from helpers import *


def test_partition_lpt_properties():
    rng = random.Random(7)
    for n_bins in (1, 2, 3, 8):
        lens = [rng.randint(1, 2_000_000) for _ in range(57)] + [5_000_000, 1, 1]
        bins = ax.partition_lpt(lens, n_bins)
        assert len(bins) == len(lens) and set(bins) <= set(range(n_bins))
        load = [sum(l for l, b in zip(lens, bins) if b == k) for k in range(n_bins)]
        # the LPT guarantee: no bin exceeds the mean load by more than the longest item
        assert max(load) <= sum(lens) / n_bins + max(lens)
        assert bins == ax.partition_lpt(lens, n_bins)  # deterministic
    # 100 equal contigs over 8 devices (BASELINE config 3): 12 or 13 each
    bins = ax.partition_lpt([1_000_000] * 100, 8)
    assert sorted(bins.count(k) for k in range(8)) == [12] * 4 + [13] * 4


class _PR(ctypes.Structure):
    _fields_ = [("record", ctypes.c_int32), ("status", ctypes.c_int32), ("begin", ctypes.c_int64), ("end", ctypes.c_int64),
                ("states", ctypes.c_void_p), ("n_states", ctypes.c_int32)]


def format_records(model, recs, pieces):
    """pieces: [(record index, begin, end, path [(b,e,state,type)], status)] in ANY order"""
    L = ax.lib()
    n = len(recs)
    names = (ctypes.c_char_p * n)(*[r[0].encode() for r in recs])
    keep = [r[1].encode() for r in recs]
    seqs = (ctypes.c_char_p * n)(*keep)
    lens = (ctypes.c_int64 * n)(*[len(r[1]) for r in recs])
    P = (_PR * len(pieces))()
    hold = []
    for i, (rec, b, e, path, status) in enumerate(pieces):
        sts = (St * max(1, len(path)))()
        for k, (pb, pe, s, t) in enumerate(path):
            sts[k].begin, sts[k].end, sts[k].state, sts[k].type = pb, pe, s, t
        hold.append(sts)
        P[i].record, P[i].status, P[i].begin, P[i].end = rec, status, b, e
        P[i].states, P[i].n_states = ctypes.cast(sts, ctypes.c_void_p), len(path)
    buf = ctypes.create_string_buffer(64 << 20)
    L.augx_format_records.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]
    rc = L.augx_format_records(model._h, n, names, seqs, lens, len(pieces), P, buf, 64 << 20)
    assert rc == 0, L.augx_last_error()
    return buf.value.decode().splitlines()


from helpers import _St as St


def test_gather_and_global_numbering_any_order():
    """pieces of all golden records, handed over in shuffled order (as they come back from several devices), give the
    reference binary's GFF byte for byte: blocks in input order, gene ids g1.. numbered across records"""
    for cfg in ("human", "fly"):
        species, opts = GOLDEN_CFGS[cfg]
        m = ax.Model(config_path(), species, **opts)
        recs = golden_inputs()
        gold = golden_paths(cfg)["records"]
        pieces = [(i, 0, len(seq) - 1, [(b, e, 0, t) for b, e, t in g["path"]], 0) for i, ((name, seq), g) in enumerate(zip(recs, gold))]
        random.Random(3).shuffle(pieces)
        ours = format_records(m, recs, pieces)
        want = golden_gff(cfg)
        # (the block header "#" line before each record but the first belongs to the block: compare whole bodies)
        assert ours[1:] == want if ours[0] == "#" else ours == want


def test_two_rank_gloo_bench_sharding(tmp_path):
    """bench.py's N>1 logic on CPU, world size 2, gloo: disjoint deterministic contig shards per rank for weak scaling, the
    LPT split of the SAME 100 contigs for strong scaling, and the max-over-ranks reduction of the step time"""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        weak = bench.rank_contigs("weak", rank, world, 6, 1000)
        strong = bench.rank_contigs("strong", rank, world, 7, 1000)
        dig = lambda seqs: [int.from_bytes(__import__("hashlib").md5(s).digest()[:6], "little") for s in seqs]
        objs = [None] * world
        dist.all_gather_object(objs, {"weak": dig(weak), "strong": dig(strong)})
        dt = bench.max_over_ranks(1.0 + rank, dist)
        if rank == 0:
            allc = dig(bench.synth_contigs(7, 1000, bench.SEED0))
            print(json.dumps({"max": dt, "weak_disjoint": len(set(objs[0]["weak"]) & set(objs[1]["weak"])) == 0,
                              "weak_n": [len(o["weak"]) for o in objs], "strong_n": [len(o["strong"]) for o in objs],
                              "strong_cover": sorted(objs[0]["strong"] + objs[1]["strong"]) == sorted(allc)}))
        dist.barrier(); dist.destroy_process_group()
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["max"] == 2.0 and out["weak_disjoint"] and out["weak_n"] == [6, 6]
    assert sorted(out["strong_n"]) == [3, 4] and out["strong_cover"]
