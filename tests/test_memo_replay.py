"""UTR states on pieces with several GC classes: the two call-history caches of the reference that only UTR states read --
tssProbsPlus (src/utrmodel.cc:748-790,1788-1790: a forward TSS window keeps the value of the class current when a 5' UTR state FIRST
asked for it) and the memo of IntronModel::aSSProb (src/intronmodel.cc:1120-1135,1182-1186: first asker; emptied by the next call
once it holds more than 1000 sites).  The oracle twin restates both inside its sequential loop; the product replays them after a
first run of the dense kernel from which cells are alive (dense.h: k1TssReplay; assmemo.h) -- two independent routes that must
agree bit for bit, and with the live reference to 1e-9."""
import os
import struct

import numpy as np
import pytest

import augustus_amd as ax
from helpers import *

OPTS = {"UTR": "on", "softmasking": "0"}


def _records():
    byname = dict(golden_inputs())
    return [(k, byname[k].upper()) for k in ("multigc_two", "multigc_levels")] + gc_step_records(3, 7)   # (seed 7: 73 sites change their value during the sweep)


@needs_ref
def test_twin_restates_the_caches_every_cell_of_the_live_reference(tmp_path):
    recs = _records()
    fa = str(tmp_path / "x.fa")
    write_fasta(fa, recs)
    m = ax.Model(config_path(), "human", **OPTS)
    S = m.n_states
    cells = str(tmp_path / "cells.bin")
    res, err = ref_harness(fa, "human", ["--%s=%s" % kv for kv in OPTS.items()], cells_file=cells)
    assert len(res) == len(recs), err
    f = open(cells, "rb")
    for (name, seq), r in zip(recs, res):
        n, S2 = struct.unpack("ii", f.read(8))
        vref = np.frombuffer(f.read(n * S2 * 8), dtype=np.float64).reshape(n, S2)
        f.read(n * 4)
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, S, cells=True, cache=True)
        assert len(set(gc.tolist())) > 1, name
        assert S2 == S and np.array_equal(np.isfinite(V), np.isfinite(vref)), name
        both = np.isfinite(V)
        assert np.all(np.abs(V[both] - vref[both]) <= 1e-9 * np.abs(vref[both]) + 5e-9), name
        assert [(b, e, t) for b, e, s, t in path] == r["path"], name


@pytest.mark.parametrize("slow", [False, True])
def test_emulated_replay_is_the_oracle_bit_for_bit(monkeypatch, slow):
    """emulator (first run, replay, second run) == twin (caches inside the loop): every cell, score and path; the walk that skips
    what it knows to be in the memo (assmemo.h) gives what the call-by-call restatement gives; and the replay changes cells"""
    if slow:
        monkeypatch.setenv("AUGX_MEMO_SLOW", "1")
    recs = _records()[1:]
    m = ax.Model(config_path(), "human", sample="0", **OPTS)
    S = m.n_states
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
    if not slow:
        monkeypatch.setenv("AUGX_NO_ASSMEMO", "1")
        plain = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
        assert sum(int(not np.array_equal(a[3], b[3])) for a, b in zip(res, plain)) >= 2
    for (name, seq), r in zip(recs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, S, cells=True, cache=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path], name
        assert np.array_equal(r[3], V), name


@needs_ref
def test_emulated_memo_lives_on_through_the_sampled_paths(tmp_path, monkeypatch):
    """the first record of soak case 11059 (117 kb of real and GC-shifted DNA, two classes): the forward matrix is the reference's
    to 1e-9 -- and still the 17th sampled path was another one: every step of a traced-back path through a longass state or a UTR
    exon that begins at an acceptor site asks aSSProb again, the memo is emptied on the way, and a site is valued with the class
    of the step's end base from then on (reference NAMGene::getSampledPath, src/namgene.cc:399-406; sampler.h: memoStep).
    All 99 sampled paths of the live reference, state by state; without the late memo they run apart."""
    import soak_cli
    monkeypatch.setenv("AUGX_SOAK_DENSE", "2")
    d, g = soak_cli.real_dna()
    recs, species, opts = soak_cli.make_case(11059, g)
    recs = recs[:1]
    fa = str(tmp_path / "c.fa")
    write_fasta(fa, recs)
    extra = ["--UTR=on", "--softmasking=0"]
    smp = ref_samples(fa, "human", extra, n=99)
    m = ax.Model(config_path(), "human", sample="100", **OPTS)
    res = emu_decode(m.tables_ptr, [recs[0][1]], m.n_states, forward=True, samples=99)
    assert [[tuple(x) for x in q] for q in smp[0]] == [list(p) for p in res[0][7]]
    monkeypatch.setenv("AUGX_NO_LATE_MEMO", "1")
    res2 = emu_decode(m.tables_ptr, [recs[0][1]], m.n_states, forward=True, samples=99)
    assert [list(p) for p in res2[0][7]] != [list(p) for p in res[0][7]]
