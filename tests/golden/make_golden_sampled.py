#!/usr/bin/env python3
"""Golden vectors of posterior sampling (--sample=100) from the REAL reference (oracle/_ref).  Run in the build container:
    python tests/golden/make_golden_sampled.py [big | cfg ...]      (no argument: every configuration)

  golden_sampled_<cfg>.gff          the reference binary's GFF (prediction part) with posterior probabilities in the score
                                    columns, for helpers.SAMPLED_CFGS (inputs.fa, or its single-GC-class records for human)
  golden_sampled_paths_<cfg>.json   per record the first sampled state paths of ref_harness --dumpsamples (5 per record; the
                                    draws are one rand() stream over the run, so they pin generator, option order and draw rule)
  golden_big_fly_sampled.gff        [big] examples/autoAug/genome.fa, --species=fly --UTR=off --softmasking=0 at the species'
                                    default --sample=100: BASELINE config 2 without its --sample=0 (1 Mbp, 200 kb pieces)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import *  # noqa


def main():
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        txt = subprocess.run([REF_AUGUSTUS, "--species=fly", "--UTR=off", "--softmasking=0", "/root/reference/examples/autoAug/genome.fa"],
                             capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        open(os.path.join(HERE, "golden_big_fly_sampled.gff"), "w").write("\n".join(gff_body(txt.stdout)) + "\n")
        return
    import tempfile
    d = tempfile.mkdtemp()
    for cfg, (species, opts, names) in SAMPLED_CFGS.items():
        if len(sys.argv) > 1 and cfg not in sys.argv[1:]:
            continue
        recs = sampled_records(cfg)
        fa = os.path.join(d, cfg + ".fa")
        write_fasta(fa, recs)
        extra = ["--%s=%s" % kv for kv in opts.items()]
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species] + extra + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        body = gff_body(txt.stdout)
        open(os.path.join(HERE, "golden_sampled_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        smp = ref_samples(fa, species, extra, 5, cfg="/root/reference/config/")
        json.dump({"species": species, "records": [{"name": n, "samples": s} for (n, _), s in zip(recs, smp)]},
                  open(os.path.join(HERE, "golden_sampled_paths_%s.json" % cfg), "w"))
        print(cfg, len(recs), "records", len(body), "gff lines")


if __name__ == "__main__":
    main()
