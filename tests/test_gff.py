"""Gene-structure + GFF stage (augustus_amd/csrc/genes.cc) against the reference binary's GFF (golden files)."""
import pytest

import augustus_amd as ax
from helpers import *


@pytest.mark.parametrize("cfg", list(GOLDEN_CFGS))
def test_gff_byte_identical_to_reference(cfg):
    species, opts = GOLDEN_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    recs = golden_inputs()
    paths = [twin_decode(m.tables_ptr, s, m.n_states)[2] for _, s in recs]
    assert format_gff(m, recs, paths) == golden_gff(cfg)


@pytest.mark.parametrize("cfg", ["off", "on"])
def test_noinframestop_drops_the_gene_with_a_stop_codon_in_its_cds(cfg):
    """--noInFrameStop=true (reference filterGenePrediction / Gene::hasInFrameStop, src/gene.cc:1422-1438, 2482-2486): the gene of
    the Viterbi path whose CDS holds a stop codon put together by a long intron is dropped, on either strand, and the numbering
    of the genes moves up -- the reference binary's GFF"""
    opts = NOINFRAMESTOP_CFGS[cfg]
    m = ax.Model(config_path(), "fly", **opts)
    recs = inframe_stop_records()
    paths = [twin_decode(m.tables_ptr, s, m.n_states)[2] for _, s in recs]
    gold = open(os.path.join(GOLDEN, "golden_noinframestop_%s.gff" % cfg)).read().splitlines()
    assert format_gff(m, recs, paths) == gold
    assert sum("\tgene\t" in l for l in gold) == (9 if cfg == "off" else 8)


@needs_ref
def test_reference_gene_objects_fall_in_address_through_the_sampling_loop(tmp_path):
    """The evidence behind the order of alternatives with EQUAL mean state probability (genes.cc: groupToGenes):
    the reference sorts Transcript POINTERS (src/gene.cc:3196), and under glibc the `Gene` objects of one iteration of its sampling
    loop (src/namgene.cc:832-870) lie BELOW those of the iteration before.  Counted here with an LD_PRELOAD shim around malloc
    (tests/golden/gene_alloc_trace.c: the 384-byte allocations = sizeof(Gene)) under the reference binary on the record of the
    `human_utr_alt` golden: of the steps from one iteration's first Gene to the next one's at least nine in ten fall (measured: all)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc for the malloc shim")
    so = str(tmp_path / "gene_alloc_trace.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(GOLDEN, "gene_alloc_trace.c"), "-ldl"], check=True)
    species, opts, names = SAMPLED_CFGS["human_utr_alt"]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, sampled_records("human_utr_alt"))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path(), LD_PRELOAD=so)
    r = subprocess.run([REF_AUGUSTUS, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0
    assert gff_body(r.stdout) == golden_sampled_gff("human_utr_alt")       # (the shim changes nothing)
    allocs = [(int(w[1]), int(w[2], 16)) for w in (l.split() for l in r.stderr.splitlines()) if len(w) == 3 and w[0] == "M"]
    # an iteration of the sampling loop = allocations more than 1000 mallocs after the one before (a sampled path lies between);
    # the clones of the printing stage follow one another within a few dozen
    firsts = [p for i, (n, p) in enumerate(allocs) if i > 0 and n - allocs[i - 1][0] > 1000]
    n_iter = int(opts["sample"]) - 1
    assert len(firsts) >= n_iter
    firsts = firsts[:n_iter]
    falls = sum(1 for a, b in zip(firsts, firsts[1:]) if b < a)
    assert falls >= 0.9 * (n_iter - 1), (falls, n_iter)
    assert allocs[0][1] > max(firsts)                                      # the Viterbi path's first Gene lies above them all
