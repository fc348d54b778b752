"""The decoder's default mode on pieces with several GC classes against the oracle, bit for bit (GPU; through the C ABI).

Two independent routes to the reference's short-intron contents near a class step: the oracle restates the SnippetProbs cache inside
its sequential loop (oracle/ghmm_twin.cc: snipGet, pinned against every cell of the real reference by test_oracle.py); the device
replays it from what the first trellis run left at the donor sites and runs the trellis again (device/snipmemo.h)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import augustus_amd as ax
from helpers import *


@pytest.mark.parametrize("species", ["human", "nasonia", "rice"])
def test_gpu_default_mode_is_the_oracle_bit_for_bit_on_multiclass_pieces(monkeypatch, species):
    from test_emu import _multiclass_records
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    monkeypatch.delenv("AUGX_EXACT_MULTICLASS", raising=False)
    m = ax.Model(config_path(), species, softmasking="0", UTR="off")
    d = ax.Decoder(m, 0)
    recs = _multiclass_records({"human": 11, "nasonia": 12, "rice": 13}[species])
    if species == "human":
        byname = dict(golden_inputs())
        recs += [(k, byname[k].upper()) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
        recs.append(("path_case", multiclass_path_case()[0].upper()))
    b = ax.Batch(d, [s for _, s in recs])
    b.decode()
    multi = differs = 0
    for i, ((name, seq), r) in enumerate(zip(recs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=True)
        assert rc == 0 and r.status == 0, name
        multi += int(len(set(gc.tolist())) > 1)
        assert r.ln_viterbi == lnv and r.states == path, name
        assert np.array_equal(b.cells(i), V), name
        V0 = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=False)[3]
        differs += int(not np.array_equal(V0, V))
    assert multi >= 2 and differs > 0, (multi, differs)


def test_gpu_default_mode_full_size_multiclass_piece_is_the_oracle(monkeypatch, tmp_path):
    """the 1 Mbp real-DNA piece (human model, two GC classes, ten class steps inside the piece): every cell, the score and the path"""
    import tarfile
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    monkeypatch.delenv("AUGX_EXACT_MULTICLASS", raising=False)
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(str(tmp_path))
    (name, seq), = read_fasta(str(tmp_path / "genome.fa"))
    m = ax.Model(config_path(), "human", softmasking="0")
    d = ax.Decoder(m, 0)
    b = ax.Batch(d, [seq])
    b.decode()
    r, = b.paths()
    rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=True)
    assert rc == 0 and r.status == 0 and len(set(gc.tolist())) > 1
    assert r.ln_viterbi == lnv and r.states == path
    assert np.array_equal(b.cells(0), V)


@pytest.mark.parametrize("cfg", list(NOINFRAMESTOP_CFGS))
def test_cli_noinframestop_matches_reference(tmp_path, cfg):
    """--noInFrameStop through the executable: off, on, on with the single-strand model (the second run is filtered on the reverse
    complement it was made on) and on with sampling -- the reference binary's GFF"""
    import subprocess
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, inframe_stop_records())
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([exe, "--species=fly"] + ["--%s=%s" % kv for kv in NOINFRAMESTOP_CFGS[cfg].items()] + [fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    assert gff_body(r.stdout) == open(os.path.join(GOLDEN, "golden_noinframestop_%s.gff" % cfg)).read().splitlines()


@pytest.mark.parametrize("opts", [{"UTR": "on"}, {"genemodel": "exactlyone", "UTR": "off"}, {"genemodel": "atleastone", "UTR": "off"}])
def test_gpu_dense_kernels_default_mode_is_the_oracle_bit_for_bit_on_multiclass_pieces(monkeypatch, opts):
    """the dense kernels (UTR states; two intergenic states) on pieces with several GC classes, default mode: the snippet cache of the
    reference replayed from the dense ln V matrix (device/snipmemo.h, `dense`), the terms rebuilt, the Viterbi pass run again -- every
    cell, score and path equal to the twin that runs the cache inside its loop; and the replay is what does it"""
    from test_emu import _multiclass_records
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    monkeypatch.delenv("AUGX_EXACT_MULTICLASS", raising=False)
    m = ax.Model(config_path(), "human", softmasking="0", sample="0", **opts)
    d = ax.Decoder(m, 0)
    byname = dict(golden_inputs())
    recs = _multiclass_records(21)[:4] + [(k, byname[k].upper()) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
    b = ax.Batch(d, [s for _, s in recs])
    b.decode()
    multi = differs = 0
    for i, ((name, seq), r) in enumerate(zip(recs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=True)
        assert rc == r.status, name
        if rc != 0:
            continue
        multi += int(len(set(gc.tolist())) > 1)
        assert r.ln_viterbi == lnv and r.states == path, name
        assert np.array_equal(b.cells(i), V), name
        V0 = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=False)[3]
        differs += int(not np.array_equal(V0, V))
    assert multi >= 2 and differs > 0, (multi, differs)
