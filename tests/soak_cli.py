#!/usr/bin/env python3
"""Randomised end-to-end soak on a GPU box: random inputs and option combinations through the `augustus` executable and through the
REAL reference binary (oracle/_ref/augustus_ref, which travels with the repository), GFF compared byte for byte.
    python tests/soak_cli.py SEED0 N          (prints one line per case; exit code = number of failures)"""
import os
import random
import subprocess
import sys
import tarfile
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import *  # noqa

EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")


def make_case(seed, g):
    """the records, species and options of soak case `seed` (g: the 1 Mbp of real DNA of tests/golden/big_inputs.tar.gz)"""
    rng = random.Random(seed)

    def gc_dna(k, gc):
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(k))
    recs = []
    for k in range(rng.randint(1, 4)):
        parts = []
        for _ in range(rng.randint(1, 5)):
            L = rng.choice([800, 3000, 7000, 15000, 40000])
            r = rng.random()
            if r < (0.9 if os.environ.get("SOAK_REAL") else 0.45): # (SOAK_REAL=1: nine parts in ten are slices of real DNA)
                st = rng.randrange(0, len(g) - L)
                s = g[st:st + L]
                if rng.random() < 0.5:
                    s = s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
                parts.append(s)
            elif r < 0.8:
                parts.append(gc_dna(L, rng.choice([0.3, 0.4, 0.45, 0.5, 0.6, 0.7])))
            else:
                parts.append(gc_dna(L // 2, 0.45) + "N" * rng.choice([1, 50, 900]) + gc_dna(L // 2, 0.55).lower())
            if os.environ.get("SOAK_NRUNS") and rng.random() < 0.5:  # (SOAK_NRUNS=1: long runs of N -- chain-only tiles and jumps of the trellis kernel)
                parts.append("N" * rng.choice([3000, 12000, 50000, 150000]))
        recs.append(("r%d" % k, "".join(parts)))
    if os.environ.get("SOAK_EQLEN"):  # (SOAK_EQLEN=1: the records of a case have ONE length -- with UTR states the reference then answers the
        L = min(len(s) for _, s in recs)   #  TSS window at base 0 of a record from what the record before left in its cache)
        recs = [(nm, s[:L]) for nm, s in recs]
        if len(recs) < 3:
            recs += [("q%d" % k, (recs[0][1][::-1] if k else recs[-1][1][L // 3:] + recs[-1][1][:L // 3])) for k in range(2)]
    species = rng.choice(["human", "fly", "arabidopsis", "saccharomyces", "human", "fly"])
    if os.environ.get("SOAK_SPECIES"):  # (SOAK_SPECIES=1: the other species of the fixtures -- up to five GC classes, models on the dense kernels, gc donor sites, ciliate code)
        species = rng.choice(["nasonia", "rice", "Vitrella_brassicaformis", "maize", "chlamy2011", "tetrahymena", "caenorhabditis", "fusarium_graminearum", "phanerochaete_chrysosporium", "human"])
    opts = {"UTR": "off", "sample": rng.choice(["0", "0", "30", "100"])}
    if rng.random() < 0.4:
        opts["softmasking"] = "0"
    dense = os.environ.get("AUGX_SOAK_DENSE") and rng.random() < 0.7
    if dense:  # the models of the dense kernels: UTR states / two intergenic states (AUGX_SOAK_DENSE=2: human, several GC classes, as well)
        multi = os.environ.get("AUGX_SOAK_DENSE") == "2"
        if rng.random() < 0.6:
            species = rng.choice(["fly", "human", "human"]) if multi else "fly"
            if os.environ.get("AUGX_SOAK_DENSE") == "3":  # (3: + the species whose UTR content tables have their own Markov order and gc donor sites)
                species = rng.choice(["chlamy2011", "caenorhabditis", "fly"])
            opts["UTR"] = "on"
            if rng.random() < 0.3:
                opts["print_utr"] = "on"
            if rng.random() < 0.3:
                opts["genemodel"] = "complete"
        else:
            species = rng.choice(["fly", "arabidopsis", "saccharomyces"] + (["human", "human"] if multi else []))
            opts["genemodel"] = rng.choice(["atleastone", "exactlyone"])
    elif rng.random() < 0.3:
        opts["singlestrand"] = "true"
    elif rng.random() < 0.2:
        opts["genemodel"] = rng.choice(["intronless", "complete"])
    if rng.random() < 0.3:
        opts["strand"] = rng.choice(["forward", "backward"])
    if rng.random() < 0.4:
        opts["maxDNAPieceSize"] = rng.choice(["20000", "50000"])
    if os.environ.get("SOAK_FORCED"):  # (SOAK_FORCED=1: pieces below the cut finder's exam window of 50 kb -- every window and most pieces have ONE length)
        opts["maxDNAPieceSize"] = rng.choice(["20000", "20000", "30000"])
    if rng.random() < 0.2:
        opts["gff3"] = "on"
    if rng.random() < 0.2:
        opts["introns"] = "on"
    if rng.random() < 0.15:
        opts["noInFrameStop"] = "true"
    if os.environ.get("SOAK_TT") and rng.random() < 0.6:  # (SOAK_TT=1: genetic codes other than the standard one; tetrahymena: its own table 6, intron content of order 3)
        if opts["UTR"] == "off" and "genemodel" not in opts and rng.random() < 0.3:
            species = "tetrahymena"
        else:
            opts["translation_table"] = rng.choice(["4", "6", "10", "11", "12", "15", "1"])
    if opts["sample"] != "0" and rng.random() < 0.2: # (the order of alternatives with EQUAL mean state probability follows heap
        opts["alternatives-from-sampling"] = "true"  #  addresses in the reference, DESIGN.md section 6: restated as the reverse order
        if rng.random() < 0.5:                       #  of creation, which a later record of a run may not keep -- a FAIL that only
                                                     #  swaps two t-numbers of a gene is that)
            opts["maxtracks"] = rng.choice(["1", "2", "3"])
    if os.environ.get("SOAK_ALT"):  # (SOAK_ALT=1: every case with sampled alternatives; SOAK_ALT=one: and ONE record, where the order of
        if opts["sample"] == "0":    #  alternatives of EQUAL mean state probability is the reference's on any machine)
            opts["sample"] = rng.choice(["30", "100"])
        opts["alternatives-from-sampling"] = "true"
        if os.environ["SOAK_ALT"] == "one":
            recs = recs[:1]
    return recs, species, opts


def real_dna():
    d = tempfile.mkdtemp()
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(d)
    return d, read_fasta(os.path.join(d, "genome.fa"))[0][1]


def main():
    seed0, n = int(sys.argv[1]), int(sys.argv[2])
    d, g = real_dna()
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    fails = 0
    for seed in range(seed0, seed0 + n):
        recs, species, opts = make_case(seed, g)
        fa = os.path.join(d, "c%d.fa" % seed)
        write_fasta(fa, recs)
        args = ["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa]
        ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
        ours = subprocess.run([EXE] + args, capture_output=True, text=True, env=env)
        ok = ref.returncode == ours.returncode and (ref.returncode != 0 or gff_body(ref.stdout) == gff_body(ours.stdout))
        print("seed", seed, species, opts, [len(s) for _, s in recs], "OK" if ok else "FAIL rc %d/%d" % (ref.returncode, ours.returncode), flush=True)
        if not ok:
            fails += 1
            if ref.returncode == 0 and ours.returncode == 0:
                import difflib
                print("\n".join(list(difflib.unified_diff(gff_body(ref.stdout), gff_body(ours.stdout), lineterm="", n=0))[:12]))
            else:
                print(ours.stderr[-300:], ref.stderr[-300:])
        else:
            os.remove(fa)
    return fails


if __name__ == "__main__":
    sys.exit(main())
