#!/usr/bin/env python3
"""Golden vectors for --noInFrameStop from the REAL reference (oracle/_ref).  Run in the build container:
    python tests/golden/make_golden_noinframestop.py

  inframe_stop.fa.gz                 20 kb of examples/autoAug/genome.fa (625000..645000) and its reverse complement: with the fly
                                     model the Viterbi path holds a gene whose CDS has a stop codon put together by a long intron
  golden_noinframestop_<cfg>.gff     the reference binary's GFF for helpers.NOINFRAMESTOP_CFGS (the option off: the gene is there;
                                     on: it is dropped and the numbering moves up; on with the single-strand model: both runs)
"""
import gzip
import os
import subprocess
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import *  # noqa


def main():
    d = tempfile.mkdtemp()
    with tarfile.open(os.path.join(HERE, "big_inputs.tar.gz")) as t:
        t.extractall(d)
    g = read_fasta(os.path.join(d, "genome.fa"))[0][1][625000:645000]
    recs = [("slice", g), ("slice_rc", g[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca")))]
    with gzip.open(os.path.join(HERE, "inframe_stop.fa.gz"), "wt") as f:
        for n, s in recs:
            f.write(">%s\n%s\n" % (n, s))
    fa = os.path.join(d, "x.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    for cfg, opts in NOINFRAMESTOP_CFGS.items():
        txt = subprocess.run([REF_AUGUSTUS, "--species=fly"] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        body = gff_body(txt.stdout)
        open(os.path.join(HERE, "golden_noinframestop_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        print(cfg, len(body), "gff lines,", sum("\tgene\t" in l for l in body), "genes")


if __name__ == "__main__":
    main()
