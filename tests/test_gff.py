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
