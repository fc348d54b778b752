"""C-ABI: libaugx.so loads, exports every symbol include/augx.h declares, loads models, and refuses to decode
without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import augustus_amd as ax
from helpers import *


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "augx.h")).read()
    names = set(re.findall(r"\b(augx_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 18
    L = ax.lib()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_model_tables_human():
    m = ax.Model(config_path(), "human")
    assert m.n_states == 47
    assert m.option("maxDNAPieceSize") == "2000000"


def test_unsupported_features_fail_loudly():
    for opts in ({"singlestrand": "true", "genemodel": "atleastone"}, {"hintsfile": "x.gff"}, {"genemodel": "bacterium"},
                 {"mea": "1"}, {"contentmodels": "false"}, {"nc": "on", "UTR": "on"}):
        with pytest.raises(ax.AugxError) as e:
            ax.Model(config_path(), "human", **opts)
        assert e.value.code == ax.AUGX_E_UNSUPPORTED
    with pytest.raises(ax.AugxError):
        ax.Model(config_path(), "no_such_species")
    # the single-strand and the intron-less model load (24 and 3 states)
    assert ax.Model(config_path(), "human", singlestrand="true").n_states == 24
    assert ax.Model(config_path(), "human", genemodel="intronless").n_states == 3
    # the model with untranslated regions (71 states) and the ones with two intergenic states (48)
    assert ax.Model(config_path(), "human", UTR="on").n_states == 71
    assert ax.Model(config_path(), "human", genemodel="exactlyone").n_states == 48
    with pytest.raises(ax.AugxError) as e:  # (the reference's own error: src/properties.cc:363-365)
        ax.Model(config_path(), "human", UTR="on", genemodel="exactlyone")
    assert e.value.code == -2 and "UTR only implemented" in str(e.value)  # AUGX_E_CONFIG


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    m = ax.Model(config_path(), "human")
    with pytest.raises(ax.AugxError) as e:
        ax.Decoder(m, 0)
    assert e.value.code == ax.AUGX_E_NODEVICE


def test_rand_is_glibc_rand_also_after_skips():
    """augx_rand restates glibc's rand() (the reference never seeds it: its draws are srand(1)'s); augx_rand_skip spends draws in
    blocks -- the next value after any skip is the one glibc gives after as many calls"""
    import ctypes
    L = ax.lib()
    libc = ctypes.CDLL("libc.so.6")
    L.augx_rand_create.restype = ctypes.c_void_p
    L.augx_rand_next.argtypes = [ctypes.c_void_p]
    L.augx_rand_skip.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.augx_rand_destroy.argtypes = [ctypes.c_void_p]
    r = L.augx_rand_create(1)
    libc.srand(1)
    try:
        for n in [0, 1, 5, 95, 96, 97, 200, 3071, 3072, 3073, 3075, 10000, 6144, 3, 31, 64, 100000]:
            L.augx_rand_skip(r, n)
            for _ in range(n):
                libc.rand()
            for _ in range(3):
                assert L.augx_rand_next(r) == libc.rand(), n
    finally:
        L.augx_rand_destroy(r)


def test_rand_buffers_filled_in_parts_are_glibc_rand():
    """From 393 216 values on a buffer of the generator is cut into parts, each started from the 31 values before it -- the
    matrix power A^PART times those before the buffer -- and filled by helper threads (sampler.h: augx_rand::refill): 60 million
    draws, one in about twenty looked at, against glibc (the emulator library carries the comparison loop: 60 million calls
    through ctypes would take minutes), and a second generator in the same process"""
    import ctypes
    E = ctypes.CDLL(EMU_LIB)
    E.emu_rand_check.restype = ctypes.c_longlong
    E.emu_rand_check.argtypes = [ctypes.c_uint, ctypes.c_longlong, ctypes.c_int]
    assert E.emu_rand_check(1, 60_000_000, 40) == -1
    assert E.emu_rand_check(20260927, 8_000_000, 1) == -1


def test_stop_thresholds_are_the_draw_rule():
    """At a stop of a chain state the sampler compares rand() with an integer threshold instead of evaluating
    z = rand() / RAND_MAX * total * 0.99999 < p(first option) (reference OptionsList::sample, src/vitmatrix.cc:295-320): the
    threshold is found by bisection over that very expression, so for every pair (total, p) the draws below it take the first
    option and the threshold itself does not -- 200 000 random pairs, incl. p = total, p = total * 0.99999 and p tiny"""
    import ctypes
    E = ctypes.CDLL(EMU_LIB)
    E.emu_stay_threshold_check.argtypes = [ctypes.c_uint, ctypes.c_int]
    assert E.emu_stay_threshold_check(1, 200000) == 0

