#!/usr/bin/env python3
"""Goldens from the REAL reference (oracle/_ref, built by oracle/Makefile from /root/reference) for the long-contig legs that
bench.py times (BASELINE configs 2 and 4 in shape: ONE contig of 23 Mbp, fly model, 200 kb pieces = a serial chain of ~115
cut-finder rounds).  Run in the build container:  python tests/golden/make_golden_long.py [long long_utr]

Input     bench.synth_contigs(1, 23000000, SEED0 + 77)[0] -- exactly the contig `bench.py: product_leg` writes as long.fa
          genome_like_big_records(examples/autoAug/genome.fa) -- the config-5 stand-in, ~100 Mbp in 8 records (~15 min on one core)
Outputs   golden_long.gff.gz       --species=fly --UTR=off --sample=0 --softmasking=0   (~4 min on one core)
          golden_long_utr.gff.gz   --species=fly --UTR=on  --sample=0 --softmasking=0   (~10 min on one core)
          each = the `examining piece` lines of --progress=true (the cut points, reference src/namgene.cc:575-603) followed by
          the prediction part of stdout; golden_long.json holds sha256 of that text so bench.py can check parity without gunzip.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import REF_AUGUSTUS, gff_body, write_fasta  # noqa

LONG_CFGS = {
    "long": ["--species=fly", "--UTR=off", "--sample=0", "--softmasking=0"],
    "long_utr": ["--species=fly", "--UTR=on", "--sample=0", "--softmasking=0"],
    # BASELINE config 5 stand-in (genome_like_big_records): --species=human at default flags (soft-masking bonus on, sample 0)
    "genome_like_big": ["--species=human"],
    "genome_1g_chr1": ["--species=human"],   # (the first record of genome_1g_records alone: 250 Mbp)
}
LONG_LEN = 23000000


def long_contig():
    import bench
    return bench.synth_contigs(1, LONG_LEN, bench.SEED0 + 77)[0].decode()


def genome_like_big_records(g, scale=1.0, sizes=(48e6, 30e6, 22e6), seed0=501, names="ABCDEFGHIJ"):
    """BASELINE config 5 in shape (GRCh38 primary contigs: chromosome-scale records with megabase N runs, isochores, soft-masked
    repeats, plus short unplaced scaffolds) built from what the container has: tiles of examples/autoAug/genome.fa (`g`, 1.0 Mbp of
    real soft-masked DNA) in both orientations, synthetic stretches whose GC content steps between 34 % and 62 % every 50-300 kb
    (several GC classes of the human model inside one 2 Mbp piece), N runs of 0.1-5 Mbp (the all-N piece shortcut,
    src/namgene.cc:205-226, and cut points next to them), lower-cased runs.  Three records of 48 / 30 / 22 Mbp and five scaffolds
    of 10-200 kb, ~100 Mbp in all; deterministic (numpy default_rng).  `scale` shrinks every length (CPU-side smoke of the recipe)."""
    import numpy as np
    ga = np.frombuffer(g.encode(), dtype=np.uint8)
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
        comp[a] = b
    rc = comp[ga[::-1]]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def isochores(rng, n):
        parts, total = [], 0
        while total < n:
            run = int(rng.integers(50000, 300000))
            gc = float(rng.choice([0.34, 0.40, 0.46, 0.52, 0.58, 0.62]))
            parts.append(rng.choice(acgt, size=run, p=[(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]))
            total += run
        a = np.concatenate(parts)[:n].copy()
        # soft-masked "repeats": lower-cased runs of 0.2-8 kb over ~15 % of the stretch
        k = 0
        while k < n:
            k += int(rng.integers(2000, 40000))
            ln = int(rng.integers(200, 8000))
            a[k:k + ln] |= 0x20
            k += ln
        return a

    def record(seed, target):
        rng = np.random.default_rng(seed)
        parts, total = [], 0
        while total < target:
            kind = rng.choice(["real", "iso", "n"], p=[0.5, 0.42, 0.08])
            if kind == "real":
                src = ga if rng.random() < 0.5 else rc
                lo = int(rng.integers(0, len(src) - 200000))
                hi = int(rng.integers(lo + 100000, len(src) + 1))
                a = src[lo:hi]
            elif kind == "iso":
                a = isochores(rng, int(rng.integers(300000, 2500000) * max(scale, 0.2)))
            else:
                a = np.full(int(10 ** rng.uniform(5, 6.7) * scale), ord("N"), dtype=np.uint8)
            parts.append(a)
            total += len(a)
        return np.concatenate(parts)[:target].tobytes().decode()

    recs = [("chr" + names[i], record(seed0 + i, int(sz * scale))) for i, sz in enumerate(sizes)]
    rng = np.random.default_rng(seed0 + 3)
    for i, n in enumerate((200000, 120000, 60000, 25000, 10000)):
        lo = int(rng.integers(0, len(ga) - n))
        recs.append(("scaffold%d" % i, (ga if i % 2 == 0 else rc)[lo:lo + n].tobytes().decode()))
    return recs


# BASELINE config 5 AT SIZE (round 6): the same recipe scaled to a genome -- 1.0 Gbp in six chromosome-scale records, the first as
# long as GRCh38's chr1 (250 Mbp = a chain of ~125 cuts at the human model's 2 Mbp pieces), plus the five scaffolds.  Too large to
# commit: `genome_1g_records` rebuilds it from the committed 1 Mbp of real DNA.  The golden is the reference binary's output for the
# 250 Mbp record ALONE (`genome_1g_chr1`, ~35 min on one core): cut points + GFF.
GENOME_1G_SIZES = (250e6, 200e6, 160e6, 140e6, 130e6, 120e6)


def genome_1g_records(g, only_first=False):
    return genome_like_big_records(g, sizes=GENOME_1G_SIZES[:1] if only_first else GENOME_1G_SIZES, seed0=601, names="123456")[:1 if only_first else None]


def golden_text(stdout_text, stderr_text):
    """what the tests and bench.py compare: cut points (--progress lines) + prediction part of the GFF"""
    return "\n".join([l for l in stderr_text.splitlines() if l.startswith("examining piece")] + gff_body(stdout_text)) + "\n"


def main():
    import tempfile
    d = tempfile.mkdtemp()
    fa = os.path.join(d, "long.fa")
    write_fasta(fa, [("long", long_contig())])
    fg = os.path.join(d, "genome_like_big.fa")
    from helpers import read_fasta
    write_fasta(fg, genome_like_big_records(read_fasta("/root/reference/examples/autoAug/genome.fa")[0][1]))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    only = [a for a in sys.argv[1:] if a in LONG_CFGS]
    f1 = os.path.join(d, "genome_1g_chr1.fa")
    if "genome_1g_chr1" in only:  # (only when asked for by name: 250 Mbp, half an hour)
        write_fasta(f1, genome_1g_records(read_fasta("/root/reference/examples/autoAug/genome.fa")[0][1], only_first=True))
    elif not only:
        only = [c for c in LONG_CFGS if c != "genome_1g_chr1"]
    meta_path = os.path.join(HERE, "golden_long.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    procs = {}
    for cfg, flags in LONG_CFGS.items():
        if only and cfg not in only:
            continue
        procs[cfg] = subprocess.Popen([REF_AUGUSTUS] + flags + ["--progress=true", fg if cfg == "genome_like_big" else f1 if cfg == "genome_1g_chr1" else fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    for cfg, p in procs.items():
        out, err = p.communicate()
        assert p.returncode == 0, err[-2000:]
        txt = golden_text(out, err)
        with gzip.GzipFile(os.path.join(HERE, "golden_%s.gff.gz" % cfg), "wb", mtime=0) as f:
            f.write(txt.encode())
        meta[cfg] = {"flags": LONG_CFGS[cfg], "sha256": hashlib.sha256(txt.encode()).hexdigest(),
                     "lines": txt.count("\n"), "cuts": sum(1 for l in txt.splitlines() if l.startswith("examining piece"))}
        print(cfg, meta[cfg])
    if os.path.exists(meta_path):  # (several runs of this script side by side: keep what the others wrote meanwhile)
        meta = dict(json.load(open(meta_path)), **{k: meta[k] for k in procs})
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
