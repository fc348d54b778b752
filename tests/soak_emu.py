#!/usr/bin/env python3
"""Randomised soak on the CPU (build container only: needs /root/reference and oracle/_ref): random records x species x options
through the lane-loop emulator of the kernels (decode, forward, sampled paths) and the host gene stage, against the reference
binary's GFF.  Record level: no piece cuts, no single-strand runs (tests/soak_cli.py does those on the GPU box).
    python tests/soak_emu.py FIRST_SEED N_CASES
A FAIL whose only difference is the t-numbers of two alternatives of one gene is the order of equals of DESIGN.md section 6 and is
reported as such."""
import os
import random
import re
import subprocess
import sys
import tarfile
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import *  # noqa
import augustus_amd as ax

CFG = "/root/reference/config/"


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    d = tempfile.mkdtemp()
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(d)
    g = read_fasta(os.path.join(d, "genome.fa"))[0][1]
    gene = dict(golden_inputs())["HS04636"]
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=CFG)
    fails = 0
    for seed in range(first, first + count):
        rng = random.Random(seed)

        def gc_dna(k, gc):
            return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(k))
        recs = []
        for r in range(rng.randint(1, 3)):
            parts = []
            for _ in range(rng.randint(1, 4)):
                L = rng.choice([600, 2500, 6000, 12000])
                x = rng.random()
                if x < 0.45:
                    st = rng.randrange(0, len(g) - L)
                    s = g[st:st + L]
                    if rng.random() < 0.4:
                        s = s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
                    parts.append(s)
                elif x < 0.6:
                    st = rng.randrange(0, max(1, len(gene) - L))
                    parts.append(gene[st:st + L])
                elif x < 0.9:
                    parts.append(gc_dna(L, rng.choice([0.3, 0.38, 0.45, 0.5, 0.58, 0.68])))
                else:
                    parts.append(gc_dna(L // 2, 0.45) + "N" * rng.choice([1, 40, 700]) + gc_dna(L // 2, 0.55).lower())
            recs.append(("r%d" % r, "".join(parts)))
        species = rng.choice(["human", "fly", "arabidopsis", "saccharomyces", "nasonia", "rice", "human", "fly"])
        opts = {"UTR": "off", "sample": rng.choice(["0", "20", "50", "100"])}
        if rng.random() < 0.5:
            opts["softmasking"] = "0"
        dense = os.environ.get("AUGX_SOAK_DENSE") and rng.random() < 0.7
        if dense:  # the models of the dense kernels: UTR states (one-class species: see DESIGN.md 6 for the others) / two intergenic states
            if rng.random() < 0.6:
                species = rng.choice(["fly", "arabidopsis"] + (["human", "human"] if os.environ.get("AUGX_SOAK_DENSE") == "2" else []))  # (2: several GC classes as well)
                opts["UTR"] = "on"
                if rng.random() < 0.3:
                    opts["print_utr"] = "on"
                if rng.random() < 0.3:
                    opts["genemodel"] = "complete"
            else:
                opts["genemodel"] = rng.choice(["atleastone", "exactlyone"])
        elif rng.random() < 0.2:
            opts["genemodel"] = rng.choice(["intronless", "complete"])
        if rng.random() < 0.25:
            opts["strand"] = rng.choice(["forward", "backward"])
        if rng.random() < 0.2:
            opts["noInFrameStop"] = "true"
        if rng.random() < 0.2:
            opts["introns"] = "on"
        if opts["sample"] != "0":
            if rng.random() < 0.3:
                opts["alternatives-from-sampling"] = "true"
                if rng.random() < 0.5:
                    opts["maxtracks"] = rng.choice(["1", "2", "3"])
            if rng.random() < 0.3:
                opts["minexonintronprob"] = rng.choice(["0.1", "0.3"])
                opts["minmeanexonintronprob"] = rng.choice(["0.2", "0.5"])
                opts["keep_viterbi"] = rng.choice(["true", "false"])
        fa = os.path.join(d, "c%d.fa" % seed)
        write_fasta(fa, recs)
        if os.environ.get("AUGX_SOAK_KEEP"):  # (to look at a case: its FASTA file stays)
            write_fasta(os.environ["AUGX_SOAK_KEEP"], recs)
        ref = subprocess.run([REF_AUGUSTUS, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
        verdict = None
        try:
            m = ax.Model(CFG, species, **opts)
            ns = int(m.option("sample") or 0)
            ns = 0 if 0 < ns < 10 else ns
            soft = opts.get("softmasking", "1") != "0"
            res = emu_decode(m.tables_ptr, [s if soft else s.upper() for _, s in recs], m.n_states, samples=max(ns - 1, 0))
            if ref.returncode != 0 or any(r[0] != 0 for r in res):
                verdict = "OK" if ref.returncode != 0 and any(r[0] != 0 for r in res) else "FAIL rc %d status %s" % (ref.returncode, [r[0] for r in res])
            else:
                paths = [[(b, e, st, emu_state_type(m.tables_ptr, st)) for b, e, st in r[2]] for r in res]
                mine = format_gff_sampled(m, recs, paths, [r[7] for r in res]) if ns else format_gff(m, recs, paths)
                want = gff_body(ref.stdout)
                if mine == want:
                    verdict = "OK"
                else:
                    norm = lambda ls: sorted(re.sub(r"(g\d+)\.t\d+", r"\1.t", l) for l in ls if not l.startswith("#"))
                    verdict = "OK but for the order of equals" if norm(mine) == norm(want) else "FAIL"
                    if verdict == "FAIL":
                        import difflib
                        print("\n".join(list(difflib.unified_diff(want, mine, lineterm="", n=0))[:14]))
        except Exception as e:
            verdict = "FAIL exception %s" % str(e)[:200]
        print("seed", seed, species, opts, [len(s) for _, s in recs], verdict, flush=True)
        if verdict.startswith("FAIL"):
            fails += 1
        else:
            os.remove(fa)
    return fails


if __name__ == "__main__":
    sys.exit(main())
