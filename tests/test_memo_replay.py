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


@needs_ref
def test_tss_window_at_base_0_is_answered_from_the_sequence_before(tmp_path):
    """Entry 0 of the reference's tssProbsPlus / tssProbsMinus lives on from sequence to sequence while the sequences keep ONE length
    (UtrModel::updateToLocalGC clears [from, to) with from = 1, src/utrmodel.cc:779-781; initAlgorithms re-allocates only for another
    length, :744-747): the second of two pieces of 20 000 bases is decoded with the value the first computed for ITS base 0.  Found by
    the soak (seed 28004, fly --UTR=on --sample=30 --maxDNAPieceSize=20000: 48 posterior probabilities of the second piece differed).
    The live reference (oracle/ref_harness --kindlist: both pieces in one run, with their own initial / terminal kinds) against the twin
    that carries the entry, and against the emulator that is handed the value (augx_tss0 -> BatchView::tss0): every sampled path of
    both pieces state by state -- and without the value most paths of the second piece differ."""
    import soak_cli
    os.environ.update(SOAK_NRUNS="1", AUGX_SOAK_DENSE="2", SOAK_REAL="1")
    try:
        d, g = soak_cli.real_dna()
        recs, species, opts = soak_cli.make_case(28004, g)
    finally:
        for k in ("SOAK_NRUNS", "AUGX_SOAK_DENSE", "SOAK_REAL"):
            del os.environ[k]
    assert species == "fly" and opts["UTR"] == "on" and opts["maxDNAPieceSize"] == "20000" and len(recs) == 1 and len(recs[0][1]) == 40000
    seq = recs[0][1]
    p1, p2 = seq[:20000], seq[20000:]
    m = ax.Model(config_path(), "fly", UTR="on", sample="30")
    S = m.n_states
    fa = str(tmp_path / "k.fa")
    write_fasta(fa, [("x0", p1), ("x1", p2)])
    rs = ref_samples(fa, "fly", ["--UTR=on", "--kindlist=0:1,1:0"], n=29)
    theirs = [[[tuple(x) for x in p] for p in r] for r in rs]
    t0 = ax.tss0(m, p1)
    assert t0[0] > -np.inf
    with_v = emu_decode(m.tables_ptr, [p1, p2], S, init_kind=[0, 1], term_kind=[1, 0], samples=29, tss0=[None, t0])
    without = emu_decode(m.tables_ptr, [p1, p2], S, init_kind=[0, 1], term_kind=[1, 0], samples=29)
    for k in range(2):
        assert [[tuple(x) for x in p] for p in with_v[k][7]] == theirs[k], k
    ours = [[tuple(x) for x in p] for p in without[1][7]]
    nd = [i for i, (a, b) in enumerate(zip(ours, theirs[1])) if a != b]
    assert len(nd) >= 5  # (the begin of a first state that is cut off by the piece start, for most paths)
    # the twin that carries entry 0 from call to call: its cells of the second piece are the emulator's with the value, and differ from a fresh decode
    twin_tss0_carry(True)
    try:
        twin_decode(m.tables_ptr, p1, S, init_kind=0, term_kind=1)
        carried = twin_decode(m.tables_ptr, p2, S, cells=True, init_kind=1, term_kind=0)
    finally:
        twin_tss0_carry(False)
    fresh = twin_decode(m.tables_ptr, p2, S, cells=True, init_kind=1, term_kind=0)
    e2 = emu_decode(m.tables_ptr, [p2], S, cells=True, init_kind=[1], term_kind=[0], tss0=[t0])
    assert np.array_equal(e2[0][3], carried[3]) and not np.array_equal(carried[3], fresh[3])
