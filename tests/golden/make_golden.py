#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the REAL reference (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the build container:  python tests/golden/make_golden.py

Outputs
  inputs.fa                 the fixed test inputs (examples/example.fa records + seeded synthetic/edge records)
  golden_paths_<cfg>.json   per record: ln Viterbi score (%.17g) and the raw state path, from oracle/_ref/ref_harness
  golden_<cfg>.gff          the reference binary's GFF for the same inputs (prediction part only)
for cfg in {human, human_nosm, fly, arabidopsis, saccharomyces}.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import *  # noqa

CFGS = {
    "human": ("human", []),
    "human_nosm": ("human", ["--softmasking=0"]),
    "fly": ("fly", ["--UTR=off", "--sample=0", "--softmasking=0"]),
    # old parameter-file format (no [EMISSION] sections) and a donor window of 8 bases: block size 4 on the device
    "arabidopsis": ("arabidopsis", ["--UTR=off", "--sample=0", "--softmasking=0"]),
    # 3 GC classes (the multigc_* records have up to three of them in one piece) and ass_end = 0 (an exon may follow base 0
    # without an acceptor site)
    "saccharomyces": ("saccharomyces", ["--UTR=off", "--sample=0", "--softmasking=0"]),
    # --genemodel=intronless: three states (intergenic, single-exon gene on either strand); the intergenic model keeps its own
    # content model (no intron model to tie it to); human: several GC classes inside a piece
    "human_intronless": ("human", ["--genemodel=intronless", "--softmasking=0"]),
    "fly_intronless": ("fly", ["--genemodel=intronless", "--UTR=off", "--sample=0"]),
    # --UTR=on: the 71-state model with untranslated regions (dense kernels); human: two GC classes, tss / tts / exon lines;
    # fly: UTR on is the species' default, soft-masking bonus on; the UTR lines in their own format and as GFF3
    "human_utr": ("human", ["--UTR=on"]),
    "human_utr_nosm": ("human", ["--UTR=on", "--softmasking=0"]),
    "fly_utr": ("fly", ["--sample=0"]),
    "fly_utr_print": ("fly", ["--sample=0", "--softmasking=0", "--print_utr=on", "--gff3=on", "--introns=on"]),
}


def build_inputs():
    recs = read_fasta("/root/reference/examples/example.fa")
    ex = recs[0][1]
    s = list(random_dna(40000, 5))
    for st, ln in [(1000, 50), (5000, 1), (5003, 2), (9000, 700), (20000, 3000), (39990, 10)]:
        for i in range(st, st + ln):
            s[i] = "N"
    recs += [
        ("rand60k", random_dna(60000, 12345)),
        ("rand20k_b", random_dna(20000, 777)),
        ("withN", "".join(s)),
        ("allN", "N" * 3000),
        ("short7", random_dna(7, 3)),
        ("short100", random_dna(100, 4)),
        ("short600", random_dna(600, 6)),
        ("iupac", random_dna(3000, 8) + "RYKMSW" + random_dna(3000, 9)),
        ("trunc_left", ex[1500:6000]),
        ("trunc_right", ex[500:4400]),
        ("trunc_both", ex[2000:5300]),
        ("revcomp", ex[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))),
    ]
    # soft-masked records (lower-case runs = nonexonpart hints with default flags): real gene with masked intron and
    # exon parts, random DNA with short and long masked runs, an all-lower-case record
    sm = ex[:1200] + ex[1200:2500].lower() + ex[2500:6000] + ex[6000:6400].lower() + ex[6400:]
    r = list(random_dna(30000, 99))
    for a, b in [(100, 160), (5000, 7000), (12000, 12010), (20000, 26000), (29900, 30000)]:
        for i in range(a, b):
            r[i] = r[i].lower()
    recs += [("softmask_gene", sm), ("softmask_rand", "".join(r)), ("softmask_all", random_dna(8000, 5).lower())]
    # records whose GC content changes along the sequence: more than one GC class inside one piece for the human model
    # (2 classes, windows of 3000 bases); the class-dependent tables switch at the steps (src/namgene.cc:245-248)
    import random

    def gc_dna(n, gc, seed):
        rng = random.Random(seed)
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))

    rc = ex[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
    recs += [
        ("multigc_gene", gc_dna(8000, 0.33, 1) + ex + gc_dna(9000, 0.62, 2)),
        ("multigc_two", gc_dna(6000, 0.62, 3) + rc + gc_dna(7000, 0.30, 4) + ex),
        ("multigc_rand", gc_dna(9000, 0.35, 5) + gc_dna(9000, 0.6, 6) + gc_dna(9000, 0.35, 7)),
        ("multigc_levels", gc_dna(6000, 0.25, 8) + gc_dna(6000, 0.38, 9) + ex + gc_dna(6000, 0.48, 10) + gc_dna(6000, 0.58, 11) + gc_dna(6000, 0.7, 12)),
    ]
    return recs


def main():
    recs = build_inputs()
    fa = os.path.join(HERE, "inputs.fa")
    write_fasta(fa, recs)
    only = sys.argv[1:]
    for cfg, (species, extra) in CFGS.items():
        if only and cfg not in only:
            continue
        res, err = ref_harness(fa, species, extra, cfg="/root/reference/config/")
        assert len(res) == len(recs), (cfg, err)
        out = {"species": species, "extra": extra,
               "records": [{"name": r["name"], "n": r["n"], "lnv": repr(r["lnv"]), "path": r["path"]} for r in res]}
        json.dump(out, open(os.path.join(HERE, "golden_paths_%s.json" % cfg), "w"))
        env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species] + extra + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        lines = txt.stdout.splitlines()
        i0 = [k for k, l in enumerate(lines) if l.startswith("# ----- prediction")][0]
        body = [l for l in lines[i0:] if not l.startswith("# command line")][:-1]
        open(os.path.join(HERE, "golden_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        print(cfg, len(res), "records", len(body), "gff lines")


def single():
    """golden_single_<cfg>.gff: the reference binary with --singlestrand=true (helpers.SINGLE_CFGS) on inputs.fa"""
    fa = os.path.join(HERE, "inputs.fa")
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    for cfg, (species, opts) in SINGLE_CFGS.items():
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        body = gff_body(txt.stdout)
        open(os.path.join(HERE, "golden_single_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        print(cfg, len(body), "gff lines")


def genemodels():
    """golden_genemodel_<cfg>.gff / golden_genemodel_paths_<cfg>.json: the reference with two intergenic states
    (helpers.GENEMODEL_CFGS) on the records of inputs.fa in which such a model has a feasible path"""
    import tempfile
    recs = genemodel_records()
    fa = os.path.join(tempfile.mkdtemp(), "gm.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    for cfg, (species, opts) in GENEMODEL_CFGS.items():
        extra = ["--%s=%s" % kv for kv in opts.items()]
        res, err = ref_harness(fa, species, [e for e in extra if not e.startswith("--sample")] + ["--sample=0"], cfg="/root/reference/config/")
        assert len(res) == len(recs) and all(r["lnv"] is not None for r in res), (cfg, err)
        json.dump({"records": [{"name": r["name"], "n": r["n"], "lnv": repr(r["lnv"]), "path": r["path"]} for r in res]},
                  open(os.path.join(HERE, "golden_genemodel_paths_%s.json" % cfg), "w"))
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species] + extra + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        body = gff_body(txt.stdout)
        open(os.path.join(HERE, "golden_genemodel_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        print(cfg, len(res), "records", len(body), "gff lines")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "genemodels":
        genemodels()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "single":
        single()
        sys.exit(0)
    main()
