#!/usr/bin/env python3
"""(build container only: needs /root/reference and oracle/_ref)   python tests/tie_order_probe.py FIRST_SEED N_CASES
PROBE=species,UTR,sample,records,length fixes the shape of the cases; PROBE_FA=file only writes the input and prints the command line.
Which order of EQUAL mean state probabilities does the reference's heap give?  Cases with --alternatives-from-sampling through the
emulator + gene stage, once with the transcripts of a gene in reverse order of creation (the product's rule), once in the order of
creation (AUGX_TIE_ASC).  ORDER = the same transcripts under other t-numbers (and, with them, another rounding of the gene's float sum)."""
import sys, os, re, subprocess, random
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import *
import augustus_amd as ax
import soak_cli
first, count = int(sys.argv[1]), int(sys.argv[2])
d, g = soak_cli.real_dna()
CFG = "/root/reference/config/"
env = dict(os.environ, AUGUSTUS_CONFIG_PATH=CFG)
tot = {"asc": 0, "desc": 0, "both": 0, "neither": 0}
for seed in range(first, first + count):
    rng = random.Random(seed)
    species = rng.choice(["human", "fly", "arabidopsis", "saccharomyces", "caenorhabditis"])
    utr = rng.random() < 0.5 and species in ("human", "fly", "arabidopsis", "caenorhabditis")
    recs = []
    for k in range(rng.randint(1, 3)):
        L = rng.choice([8000, 15000, 30000])
        st = rng.randrange(0, len(g) - L)
        recs.append(("r%d" % k, g[st:st + L]))
    opts = {"UTR": "on" if utr else "off", "sample": rng.choice(["30", "100"]), "alternatives-from-sampling": "true"}
    if os.environ.get("PROBE"):  # PROBE=species,UTR,sample,records,length
        sp, u, smp, nr, ln = os.environ["PROBE"].split(",")
        species, opts["UTR"], opts["sample"] = sp, u, smp
        recs = []
        for k in range(int(nr)):
            st = rng.randrange(0, len(g) - int(ln))
            recs.append(("r%d" % k, g[st:st + int(ln)]))
    if rng.random() < 0.3:
        opts["softmasking"] = "0"
    fa = os.path.join(d, "t%d.fa" % seed)
    write_fasta(fa, recs)
    if os.environ.get("PROBE_FA"):  # only the input and the command line
        write_fasta(os.environ["PROBE_FA"], recs)
        print("--species=" + species, " ".join("--%s=%s" % kv for kv in opts.items()))
        continue
    ref = subprocess.run([REF_AUGUSTUS, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
    want = gff_body(ref.stdout)
    m = ax.Model(CFG, species, **opts)
    ns = int(m.option("sample"))
    soft = opts.get("softmasking", "1") != "0"
    res = emu_decode(m.tables_ptr, [s if soft else s.upper() for _, s in recs], m.n_states, samples=ns - 1)
    paths = [[(b, e, st, emu_state_type(m.tables_ptr, st)) for b, e, st in r[2]] for r in res]
    out = {}
    for mode in ("asc", "desc"):
        os.environ.pop("AUGX_TIE_ASC", None)
        if mode == "asc":
            os.environ["AUGX_TIE_ASC"] = "1"
        mine = format_gff_sampled(m, recs, paths, [r[7] for r in res])
        def norm(ls):
            out = []
            for l in ls:
                if l.startswith("#"):
                    continue
                f = l.split("\t")
                if len(f) > 5 and f[2] == "gene":
                    f[5] = "*"
                out.append(re.sub(r"(g\d+)\.t\d+", r"\1.t", "\t".join(f)))
            return sorted(out)
        out[mode] = "OK" if mine == want else ("ORDER" if norm(mine) == norm(want) else "FAIL")
    ntx = sum(1 for l in want if "\ttranscript\t" in l)
    key = "both" if out["asc"] == out["desc"] == "OK" else "asc" if out["asc"] == "OK" else "desc" if out["desc"] == "OK" else "neither"
    tot[key] += 1
    print("seed", seed, species, opts, [len(s) for _, s in recs], ntx, "tx", out, flush=True)
print(tot)
