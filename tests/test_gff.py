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
