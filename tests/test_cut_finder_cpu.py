"""The cut finder (driver.cc: findCutPoints, C ABI augx_find_cuts; reference NAMGene::getNextCutEndPoint, src/namgene.cc:973-1133) decodes its
exam windows AHEAD of the serial chain of cuts, from a forecast.  The forecast may only decide which windows are decoded early, never a
cut: with the decode function of the test (the lane-loop emulator of the kernel source) the pieces must be those of the plain chain, one
window per round, whatever the forecast guessed.  (On the device the same is checked against the reference binary's own cuts:
test_cli_cut_finder_ladder_matches_reference, test_cli_piece_cutting_matches_reference.)"""
import ctypes
import os

import pytest

import augustus_amd as ax
from helpers import config_path, emu_decode, emu_state_type, golden_inputs, random_dna

libc = ctypes.CDLL("libc.so.6")
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]


def emulator_decode_fn(model, log):
    S = model.n_states
    tp = model.tables_ptr

    def fn(user, pieces, n, out):
        seqs = [ctypes.string_at(pieces[i].seq, pieces[i].len) for i in range(n)]
        res = emu_decode(tp, seqs, S, init_kind=[pieces[i].init_kind for i in range(n)], term_kind=[pieces[i].term_kind for i in range(n)])
        log.append([(len(s), pieces[i].init_kind, pieces[i].term_kind) for i, s in enumerate(seqs)])
        for i, r in enumerate(res):
            status, lnv, path = r[0], r[1], r[2]
            out[i].status, out[i].ln_viterbi, out[i].n_states = status, lnv, len(path)
            mem = libc.malloc(max(1, len(path)) * ctypes.sizeof(ax._State))  # (augx_path_free releases it with free())
            arr = ctypes.cast(mem, ctypes.POINTER(ax._State))
            for k, (b, e, st) in enumerate(path):
                arr[k].begin, arr[k].end, arr[k].state, arr[k].type = b, e, st, emu_state_type(tp, st)
            out[i].states = arr
        return 0
    return ax.DECODE_FN(fn)


def records():
    ex = dict(golden_inputs())["HS04636"]
    sm = list(random_dna(260000, 77))
    for a, b in [(15000, 48000), (55000, 61000), (70000, 125000), (139000, 140500), (170000, 259000)]:
        for i in range(a, b):
            sm[i] = sm[i].lower()
    return [random_dna(520000, 555),
            random_dna(50000, 557) + ex + random_dna(45000, 558) + ex + random_dna(30000, 559) + ex[700:9000] * 6 + random_dna(90000, 560),
            random_dna(30000, 556),            # needs no cut
            "".join(sm)]                       # soft-masked runs across the cuts and the window ends


@pytest.mark.parametrize("species,opts", [("human", {"maxDNAPieceSize": "60000"}), ("fly", {"UTR": "off", "sample": "0", "maxDNAPieceSize": "70000"}),
                                          ("fly", {"sample": "0", "maxDNAPieceSize": "60000"})])   # (fly's default: UTR states, the dense kernels, five guesses per window)
def test_windows_decoded_ahead_give_the_cuts_of_the_serial_chain(monkeypatch, species, opts):
    m = ax.Model(config_path(), species, **opts)
    recs = records()
    if m.n_states > 48:  # (the emulated 71-state decode takes 16 s per Mbp: fewer records, smaller batches of guesses)
        monkeypatch.setenv("AUGX_CUT_ASK", "16")
        recs = [recs[0][:330000], recs[2], recs[3][:150000]]
    log0, log1 = [], []
    serial, st0 = ax.find_cuts(m, recs, emulator_decode_fn(m, log0), scout=0)
    ahead, st1 = ax.find_cuts(m, recs, emulator_decode_fn(m, log1), scout=1)
    assert serial == ahead
    # the pieces tile every record, cut kinds as the piece loop sets them (src/namgene.cc:584-603)
    for r, s in enumerate(recs):
        ps = [c for c in serial if c[0] == r]
        assert ps[0][2] == 0 and ps[-1][3] == len(s) - 1 and all(a[3] + 1 == b[2] for a, b in zip(ps, ps[1:]))
        assert all(c[1] == 0 and c[4] == (0 if c[2] == 0 else 1) and c[5] == (0 if c[3] == len(s) - 1 else 1) for c in ps)
        assert all(c[3] - c[2] + 1 <= int(opts["maxDNAPieceSize"]) for c in ps)
    assert len([c for c in serial if c[0] == 0]) >= 5 and len([c for c in serial if len(recs[c[0]]) <= int(opts["maxDNAPieceSize"])]) == 1
    # serial: every batch holds at most one window per unfinished record, all of them used; ahead: a scout decode, then fewer batches
    assert st0["scout_tiles"] == 0 and st0["windows_decoded"] == st0["windows_used"] and all(len(b) <= 3 for b in log0)
    assert st1["scout_tiles"] > 0 and st1["windows_used"] == st0["windows_used"]
    assert st1["batches"] < st0["batches"], (st0, st1)
    assert st1["windows_decoded"] <= int(os.environ.get("AUGX_CUT_ASK", "256")) * st1["batches"]   # (the forecast's breadth is bounded per batch)


def test_decode_failure_is_reported():
    m = ax.Model(config_path(), "human", maxDNAPieceSize="60000")
    fn = ax.DECODE_FN(lambda user, pieces, n, out: ax.AUGX_E_HIP)
    with pytest.raises(ax.AugxError):
        ax.find_cuts(m, [random_dna(200000, 1)], fn, scout=0)


def all_intergenic_decode_fn(log):
    """a decode function that calls every piece one intergenic run (state 0, type 0): the chain logic alone"""
    def fn(user, pieces, n, out):
        log.append(n)
        for i in range(n):
            out[i].status, out[i].ln_viterbi, out[i].n_states = 0, -1.0, 1
            arr = ctypes.cast(libc.malloc(ctypes.sizeof(ax._State)), ctypes.POINTER(ax._State))
            arr[0].begin, arr[0].end, arr[0].state, arr[0].type = 0, pieces[i].len - 1, 0, 0
            out[i].states = arr
        return 0
    return ax.DECODE_FN(fn)


@pytest.mark.parametrize("scout", [0, 1, -1])
@pytest.mark.parametrize("ask,recs", [(None, ["A" * 1000] * 300 + ["C" * 200000]),      # more open records than a batch has room for
                                      ("1", ["A" * 150000, "C" * 200000, "G" * 500]),   # room for one window, two long records
                                      (None, ["A" * 1000] * 200 + ["C" * 400000])])    # short records must not eat the long one's share
def test_every_open_record_gets_a_window(monkeypatch, scout, ask, recs):
    """round-4 advisor finding: `room = maxAsk / nOpen` was 0 with more than maxAsk unfinished records (short ones included), no
    window was asked, and the long records were left without pieces while the call returned success"""
    if ask:
        monkeypatch.setenv("AUGX_CUT_ASK", ask)
    m = ax.Model(config_path(), "human", maxDNAPieceSize="60000")
    log = []
    cuts, st = ax.find_cuts(m, recs, all_intergenic_decode_fn(log), scout=scout)
    for r, s in enumerate(recs):
        ps = [c for c in cuts if c[0] == r]
        assert ps and ps[0][2] == 0 and ps[-1][3] == len(s) - 1 and all(a[3] + 1 == b[2] for a, b in zip(ps, ps[1:])), r
        assert all(c[3] - c[2] + 1 <= 60000 for c in ps)
    assert st["windows_used"] >= sum(len(s) // 60000 for s in recs if len(s) > 60000)
    if ask is None and scout == 0 and len(recs) == 201:
        assert st["batches"] == st["windows_decoded"] == st["windows_used"] == 10   # (one window per round for the one open record: cuts every 35 kb)
