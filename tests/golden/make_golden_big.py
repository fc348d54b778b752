#!/usr/bin/env python3
"""Mb-scale golden vectors from the REAL reference (oracle/_ref, built by oracle/Makefile from /root/reference) for the
BASELINE.json configurations.  Run in the build container:  python tests/golden/make_golden_big.py

Inputs
  big_inputs.tar.gz         examples/autoAug/genome.fa of the reference (chrI, 1.0 Mbp of real, soft-masked DNA: the
                            in-container stand-in for examples/chr2L) -- DATA of the reference, not source
  the first bench contig    bench.synth_contigs(1, 1000000, 12345) (BASELINE config 3), regenerated from its seed
Outputs
  golden_big_<cfg>.gff      the reference binary's GFF (prediction part) for
      fly         genome.fa --species=fly --UTR=off --sample=0 --softmasking=0   (200 kb pieces: cut chain, config 2)
      fly_sm      genome.fa --species=fly --UTR=off --sample=0                   (soft-masking bonus across the cuts)
      human       genome.fa --species=human --softmasking=0                      (two GC classes inside one 1 Mbp piece)
      human_sm    genome.fa --species=human                                      (default flags)
      synth       the bench contig, --species=human                              (config 3)
  golden_big_paths.json     ln Viterbi (%.17g) + raw state path of the single-piece cases (human, synth) from ref_harness
"""
import json
import os
import subprocess
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import *  # noqa

BIG_CFGS = {
    "fly": ("genome", "fly", ["--UTR=off", "--sample=0", "--softmasking=0"]),
    "fly_sm": ("genome", "fly", ["--UTR=off", "--sample=0"]),
    "human": ("genome", "human", ["--softmasking=0"]),
    "human_sm": ("genome", "human", []),
    "synth": ("synth", "human", []),
    "human_intronless": ("genome", "human", ["--genemodel=intronless"]),          # 3-state model, default flags, one 1 Mbp piece
    "fly_intronless": ("genome", "fly", ["--genemodel=intronless", "--UTR=off", "--sample=100", "--softmasking=0"]),  # + sampling
    "human_sampled": ("genome", "human", ["--sample=100"]),   # one 1 Mbp piece, two GC classes with ten steps, soft-masking, sampling
    "fly_single": ("genome", "fly", ["--singlestrand=true", "--UTR=off", "--sample=0"]),    # 24-state model, both runs of five 200 kb pieces
    # BASELINE config 4 stand-in: the 71-state model with UTR states at the fly model's own 200 kb pieces
    "fly_utr": ("genome", "fly", ["--sample=0"]),            # UTR on (the species' default), soft-masking bonus, cut chain
    "fly_default": ("genome", "fly", []),                    # every default of the species: UTR on, sample 100, soft-masking
    "human_utr": ("genome", "human", ["--UTR=on"]),          # one 1 Mbp piece with two GC classes
    "human_utr_sampled": ("genome", "human", ["--UTR=on", "--sample=100"]),  # ... with the forward pass and 99 sampled paths
    "synth_sampled": ("synth", "human", ["--sample=100"]),   # the bench contig with sampling: 99 paths of a 1 Mbp piece, 10^8 draws of the one generator
}


def genome_like_records(g):
    import bench
    rc = g[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
    big = g + rc[100000:900000] + bench.synth_contigs(1, 1300000, 777)[0].decode() + g[300000:1000000] + rc[:400000]
    return [("chrBig", big), ("chrI_tail", g[650000:])]


def main():
    ref_genome = "/root/reference/examples/autoAug/genome.fa"
    tar = os.path.join(HERE, "big_inputs.tar.gz")
    with tarfile.open(tar, "w:gz") as t:
        t.add(ref_genome, arcname="genome.fa")
    import bench
    d = tempfile.mkdtemp()
    synth = os.path.join(d, "synth.fa")
    write_fasta(synth, [("rand000", bench.synth_contigs(1, 1000000, 12345)[0].decode())])
    files = {"genome": ref_genome, "synth": synth}
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    paths = {}
    only = [a for a in sys.argv[1:] if a in BIG_CFGS]
    for cfg, (inp, species, extra) in BIG_CFGS.items():
        if only and cfg not in only:
            continue
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species] + extra + [files[inp]], capture_output=True, text=True, env=env)
        assert txt.returncode == 0 and txt.stderr == "", txt.stderr
        body = gff_body(txt.stdout)
        open(os.path.join(HERE, "golden_big_%s.gff" % cfg), "w").write("\n".join(body) + "\n")
        print(cfg, len(body), "gff lines")
        if cfg in ("human", "synth"):
            res, err = ref_harness(files[inp], species, extra, cfg="/root/reference/config/")
            assert len(res) == 1, err
            paths[cfg] = {"lnv": repr(res[0]["lnv"]), "path": res[0]["path"], "n": res[0]["n"]}
    if only:
        return
    json.dump(paths, open(os.path.join(HERE, "golden_big_paths.json"), "w"))
    # a genome-like input for the human model at its own maxDNAPieceSize (2 Mbp): one 4.2 Mbp record (three pieces, two cut
    # points found in 150 kb exam windows) + one short record; soft-masked real DNA in both orientations and random DNA
    import gzip
    fa = os.path.join(d, "genome_like.fa")
    write_fasta(fa, genome_like_records(read_fasta(ref_genome)[0][1]))
    txt = subprocess.run([REF_AUGUSTUS, "--species=human", "--progress=true", fa], capture_output=True, text=True, env=env)
    assert txt.returncode == 0, txt.stderr
    with gzip.open(os.path.join(HERE, "golden_big_genome_like.gff.gz"), "wt") as f:
        f.write("\n".join([l for l in txt.stderr.splitlines() if l.startswith("examining piece")] + gff_body(txt.stdout)) + "\n")
    print("genome_like", len(gff_body(txt.stdout)), "gff lines")


def more_species():
    """two more species (config_more.tar.gz): nasonia (5 GC classes) and rice (4), 200 kb pieces"""
    g = read_fasta("/root/reference/examples/autoAug/genome.fa")[0][1]
    byname = dict(golden_inputs())
    recs = [("chrI_300k", g[100000:400000]), ("multigc_levels", byname["multigc_levels"]), ("HS04636", byname["HS04636"])]
    fa = os.path.join(HERE, "inputs_more.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH="/root/reference/config")
    for cfg, (species, opts) in MORE_CFGS.items():
        extra = ["--%s=%s" % kv for kv in opts.items()]
        res, err = ref_harness(fa, species, extra, cfg="/root/reference/config/")
        assert len(res) == len(recs), err
        json.dump({"species": species, "records": [{"name": r["name"], "n": r["n"], "lnv": repr(r["lnv"]), "path": r["path"]} for r in res]},
                  open(os.path.join(HERE, "golden_more_paths_%s.json" % cfg), "w"))
        txt = subprocess.run([REF_AUGUSTUS, "--species=" + species, "--progress=true"] + extra + [fa], capture_output=True, text=True, env=env)
        assert txt.returncode == 0, txt.stderr
        open(os.path.join(HERE, "golden_more_%s.gff" % cfg), "w").write(
            "\n".join([l for l in txt.stderr.splitlines() if l.startswith("examining piece")] + gff_body(txt.stdout)) + "\n")
        print(cfg, len(res), "records")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "more":
        more_species()
        sys.exit(0)
    main()
