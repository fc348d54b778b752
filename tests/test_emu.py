"""The device kernel bodies (augustus_amd/csrc/device/kernels.h) executed by the lane-loop emulator must be
bit-identical to the oracle: every trellis cell, the score and the path.  (CPU-only; the same comparison runs on
the real GPU in test_gpu_parity.py.)"""
import numpy as np
import pytest

import augustus_amd as ax
from helpers import *


@pytest.mark.parametrize("cfg", list(GOLDEN_CFGS))
def test_emulated_kernels_bit_identical_to_oracle(cfg):
    species, opts = GOLDEN_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    S = m.n_states
    recs = golden_inputs()
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
    for (name, seq), (st, lnv, path, V, cls) in zip(recs, res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        if st == ax.AUGX_E_UNSUPPORTED:  # multi-GC-class piece: not decoded by this version (fails loudly)
            assert len(set(gc.tolist())) > 1 or cls == -1
            continue
        assert st == 0 and rc == 0, name
        assert lnv == lnv2, name
        assert path == [(b, e, s) for b, e, s, t in path2], name
        if set(seq.upper()) != {"N"}:
            assert np.array_equal(V, V2), name  # -inf == -inf holds, no NaNs are produced


def test_emulated_interior_piece_kinds():
    m = ax.Model(config_path(), "human")
    S = m.n_states
    seq = random_dna(12000, 31337)
    for ik, tk in [(1, 1), (0, 1), (1, 0)]:
        (st, lnv, path, V, cls), = emu_decode(m.tables_ptr, [seq], S, cells=True, init_kind=ik, term_kind=tk)
        rc, lnv2, path2, V2, _ = twin_decode(m.tables_ptr, seq, S, cells=True, init_kind=ik, term_kind=tk)
        assert st == 0 and lnv == lnv2 and np.array_equal(V, V2)
        assert path == [(b, e, s) for b, e, s, t in path2]


@pytest.mark.parametrize("species", ["human", "fly", "arabidopsis"])
def test_emulated_ragged_lengths(species):
    """Edge lengths around the tile (64) and block (8) sizes, a one-base piece, and a ragged batch."""
    m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
    S = m.n_states
    seqs = [random_dna(n, 100 + n) for n in (1, 2, 7, 8, 9, 63, 64, 65, 127, 129, 600, 1031)]
    res = emu_decode(m.tables_ptr, seqs, S, cells=True)
    for seq, (st, lnv, path, V, cls) in zip(seqs, res):
        rc, lnv2, path2, V2, _ = twin_decode(m.tables_ptr, seq, S, cells=True)
        assert st == rc or (st == ax.AUGX_E_NOPATH and rc != 0), len(seq)
        if rc == 0:
            assert lnv == lnv2 and np.array_equal(V, V2), len(seq)
            assert path == [(b, e, s) for b, e, s, t in path2], len(seq)
