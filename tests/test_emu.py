"""The device kernel bodies (augustus_amd/csrc/device/kernels.h) executed by the lane-loop emulator must be
bit-identical to the oracle: every trellis cell, the score and the path.  (CPU-only; the same comparison runs on
the real GPU in test_gpu_parity.py.)"""
import numpy as np
import pytest

import augustus_amd as ax
from helpers import *


@pytest.fixture(autouse=True)
def _one_class_per_end_base(monkeypatch):
    """The kernels score a short-intron interior with the class of its end base; the replay of the reference's snippet cache on
    pieces with several GC classes (exact mode, the product's default) sits on top of that.  The tests here check the first pass
    on its own: exact mode off, and with it the twin's restatement of the cache (helpers.twin_decode follows the same switch).
    The tests that switch it on compare the whole with the real reference and with the twin's cache."""
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "0")


@pytest.mark.parametrize("cfg", list(GOLDEN_CFGS))
def test_emulated_kernels_bit_identical_to_oracle(cfg):
    species, opts = GOLDEN_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    S = m.n_states
    recs = golden_inputs()
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
    most_classes = 0
    for (name, seq), (st, lnv, path, V, cls) in zip(recs, res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        most_classes = max(most_classes, len(set(gc.tolist())))
        assert st == 0 and rc == 0, name
        assert lnv == lnv2, name
        assert path == [(b, e, s) for b, e, s, t in path2], name
        if set(seq.upper()) != {"N"}:
            assert np.array_equal(V, V2), name  # -inf == -inf holds, no NaNs are produced
    # the multigc_* records switch GC class inside the piece (class-dependent tables follow the end base of each state)
    assert most_classes >= {"human": 2, "human_nosm": 2, "saccharomyces": 3}.get(cfg, 1)


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


def adversarial_cases():
    """dense splice sites (more candidates per tile than the LDS staging holds), 20 kb open reading frames on both strands
    (exon candidates far outside every LDS window), start-codon repeats, purine / pyrimidine tracts"""
    rng8, rng9 = np.random.default_rng(8), np.random.default_rng(9)
    return {
        "aggt": "AGGT" * 3000,
        "gtag_rand": random_dna(2000, 1) + "GTAG" * 1500 + random_dna(2000, 2),
        "polyGCC": random_dna(1000, 3) + "ATG" + "GCC" * 7000 + "TAA" + random_dna(1000, 4),
        "polyGGC_rev": random_dna(1000, 5) + "TTA" + "GGC" * 7000 + "CAT" + random_dna(1000, 6),
        "atg_rep": "ATG" * 4000 + random_dna(3000, 7),
        "ag_rich": "".join(rng8.choice(list("AG"), size=20000)),
        "ct_rich": "".join(rng9.choice(list("CT"), size=20000)),
    }


@pytest.mark.parametrize("species", ["human", "fly"])
def test_emulated_adversarial_sequences(species):
    m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
    S = m.n_states
    cases = adversarial_cases()
    res = emu_decode(m.tables_ptr, list(cases.values()), S, cells=True)
    decoded = 0
    for (name, seq), (st, lnv, path, V, cls) in zip(cases.items(), res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        decoded += 1
        assert st == 0 and rc == 0 and lnv == lnv2 and np.array_equal(V, V2), name
        assert path == [(b, e, s) for b, e, s, t in path2], name
    assert decoded == len(cases)


def test_emulated_content_stairs_smoothing():
    """GC-content stairs (reference ContentStairs::computeStairs, src/motif.cc:543-616), settled from the device's window
    classes run by run (layout.h: stairsPlanes) vs the oracle's loop over the positions: all cells must be equal."""
    import random

    def gc_dna(n, gc, seed):
        rng = random.Random(seed)
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))

    m = ax.Model(config_path(), "human", softmasking="0")
    S = m.n_states
    seqs = [gc_dna(40000, 0.445, 11),                                                # GC content at the class boundary: the window
                                                                                      # classes flicker, the stairs keep steps >= 1000 apart
            gc_dna(9000, 0.30, 4) + gc_dna(6000, 0.65, 5) + gc_dna(9000, 0.30, 6),   # a real step up and down
            gc_dna(2000, 0.65, 7) + gc_dna(9000, 0.30, 8)]                            # a step close to the start
    res = emu_decode(m.tables_ptr, seqs, S, cells=True)
    n_steps = []
    for seq, (st, lnv, path, V, cls) in zip(seqs, res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        n_steps.append(int((gc[1:] != gc[:-1]).sum()))
        assert st == 0 and rc == 0 and lnv == lnv2 and np.array_equal(V, V2)
        assert path == [(b, e, s) for b, e, s, t in path2]
    assert n_steps[0] >= 3 and n_steps[1] == 2 and n_steps[2] >= 1, n_steps


def test_emulated_planes_beyond_lds_table():
    """a piece may have more GC classes than the candidate kernel keeps transition terms for in LDS (8): the others are read
    from the model's table.  The emulator built with room for ONE plane takes that path on the two- and three-class records."""
    for cfg in ("human_nosm", "saccharomyces"):
        species, opts = GOLDEN_CFGS[cfg]
        m = ax.Model(config_path(), species, **opts)
        recs = [(n, s) for n, s in golden_inputs() if n.startswith("multigc")]
        res = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, cells=True, lib=os.path.join(ROOT, "build", "libaugx_emu_pl1.so"))
        for (name, seq), (st, lnv, path, V, cls) in zip(recs, res):
            rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True)
            assert st == 0 and rc == 0 and lnv == lnv2 and np.array_equal(V, V2), name
            assert path == [(b, e, s) for b, e, s, t in path2], name


@pytest.mark.parametrize("blk", ["4", "2"])
def test_emulated_small_block_sizes(monkeypatch, blk):
    """the candidate / trellis kernels are templates over the block size (8, 4 or 2 bases per step of the wavefront pipeline,
    layout.h: chooseBlockSize); the smaller instantiations, forced here, must give the same cells"""
    monkeypatch.setenv("AUGX_BLK", blk)
    m = ax.Model(config_path(), "human")
    S = m.n_states
    recs = [(n, s) for n, s in golden_inputs() if n in ("HS04636", "withN", "short100", "softmask_gene", "multigc_gene", "trunc_both")]
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
    for (name, seq), (st, lnv, path, V, cls) in zip(recs, res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        assert st == 0 and rc == 0 and lnv == lnv2 and np.array_equal(V, V2), name
        assert path == [(b, e, s) for b, e, s, t in path2], name



def _segment_cases():
    import bench
    import tarfile
    import tempfile
    d = tempfile.mkdtemp()
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(d)
    (name, g), = read_fasta(os.path.join(d, "genome.fa"))
    return [bench.synth_contigs(1, 400000, 12345)[0].decode(),        # uniform-random DNA (the bench workload)
            random_dna(90000, 5),                                     # too short to be cut
            g[100000:430000],                                         # real DNA, soft-masked, two GC classes
            "N" * 250000,                                             # no nucleotide at all
            random_dna(200000, 9) + "N" * 150000 + random_dna(100000, 10)]  # a dead start inside an N run cannot converge: gives up


import os


@pytest.mark.parametrize("env", [{"AUGX_SEG_LEN": "100000"}, {"AUGX_SEG_LEN": "100000", "AUGX_SEG_CHECK_TILES": "100000"}, {}])
def test_emulated_segment_parallel_trellis(monkeypatch, env):
    """pieces cut into segments (pass 1: every segment at once, all but the first from a dead start; pass 2: fix-ups until the
    retired values differ from pass 1 by one constant; pass 3: continuation where a fix-up gave up -- forced for every segment by
    an unreachable check length): every cell, the score and the path equal the sequential oracle bit for bit"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = ax.Model(config_path(), "human")
    S = m.n_states
    seqs = _segment_cases()
    import ctypes
    from helpers import EMU_LIB
    E = ctypes.CDLL(EMU_LIB)
    E.emu_quiet_tiles.restype = ctypes.c_longlong
    E.emu_jump_tiles.restype = ctypes.c_longlong
    q0, j0 = E.emu_quiet_tiles(), E.emu_jump_tiles()
    res = emu_decode(m.tables_ptr, seqs, S, cells=True)
    # (the run of 150 000 N: a dozen chain-only tiles, then a jump to its end -- by pass 1 and again by the fix-ups and continuations
    #  that cannot converge inside it; with one workgroup per piece one jump, with segments one per segment that lies in it)
    assert E.emu_quiet_tiles() - q0 >= 10 and E.emu_jump_tiles() - j0 >= 2000
    for seq, (st, lnv, path, V, cls) in zip(seqs, res):
        rc, lnv2, path2, V2, _ = twin_decode(m.tables_ptr, seq, S, cells=True)
        assert st == 0 and rc == 0 and lnv == lnv2, len(seq)
        assert path == [(b, e, s) for b, e, s, t in path2], len(seq)
        if set(seq) != {"N"}:
            assert np.array_equal(V, V2), len(seq)


def _forward_records():
    byname = dict(golden_inputs())
    return [(k, byname[k]) for k in ("HS04636", "HS08198", "rand20k_b", "withN", "short7", "short100", "short600", "iupac", "trunc_left",
                                     "trunc_right", "trunc_both", "revcomp", "softmask_rand")]


@needs_ref
@pytest.mark.parametrize("cfg", ["human_nosm", "human", "fly", "arabidopsis"])
def test_emulated_forward_matches_reference(tmp_path, cfg):
    """the forward algorithm (groundwork of posterior sampling; device/kernels.h: forwardPiece) against every forward variable
    of the REAL reference (NAMGene::getForwardVariables): the same cells are alive, ln F within 1e-9 relative (+ the absolute
    slack of the first bases, see test_oracle.py), on single-class records of three species incl. soft-masked ones"""
    species, opts = GOLDEN_CFGS[cfg]
    recs = _forward_records()
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    extra = ["--%s=%s" % kv for kv in opts.items() if kv[0] != "sample"]
    Fref = ref_forward(fa, species, extra)
    m = ax.Model(config_path(), species, **opts)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, forward=True)
    for (name, seq), fr, r in zip(recs, Fref, res):
        F = r[5]
        assert np.array_equal(np.isfinite(F), np.isfinite(fr)), name
        both = np.isfinite(F)
        assert np.all(np.abs(F[both] - fr[both]) <= 1e-9 * np.abs(fr[both]) + 5e-9), name
        # ln P(sequence) = the terminal-weighted sum of the last column
        assert r[6] >= r[1] and r[6] - r[1] < 0.01 * len(seq) + 5


@needs_ref
def test_emulated_exact_mode_viterbi_cells_with_several_gc_classes(tmp_path, monkeypatch):
    """exact mode (augx_decoder_set_exact; AUGX_EXACT_MULTICLASS in the emulator): after a first trellis run the reference's snippet
    cache around the class steps is replayed from which donor-site values are alive, the candidate terms concerned are rebuilt and
    the trellis runs again -- every Viterbi variable of the real reference to 1e-9 on the records with several GC classes (without:
    18 to 361 cells per record off, by up to 5.4)"""
    import struct
    byname = dict(golden_inputs())
    recs = [(k, byname[k]) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    cells = str(tmp_path / "cells.bin")
    res, err = ref_harness(fa, "human", ["--softmasking=0"], cells_file=cells)
    m = ax.Model(config_path(), "human", softmasking="0")
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "1")
    em = emu_decode(m.tables_ptr, [s.upper() for _, s in recs], m.n_states, cells=True)
    with open(cells, "rb") as f:
        for (name, seq), r, e in zip(recs, res, em):
            n, S = struct.unpack("ii", f.read(8))
            vref = np.frombuffer(f.read(n * S * 8), dtype=np.float64).reshape(n, S)
            f.read(n * 4)
            V = e[3]
            assert np.array_equal(np.isfinite(V), np.isfinite(vref)), name
            both = np.isfinite(V)
            assert np.all(np.abs(V[both] - vref[both]) <= 1e-9 * np.abs(vref[both]) + 5e-9), name
            assert [(b, e2, emu_state_type(m.tables_ptr, st)) for b, e2, st in e[2]] == r["path"], name


@needs_ref
@pytest.mark.parametrize("seed", [901, 909, 916])
def test_emulated_exact_mode_randomised(tmp_path, monkeypatch, seed):
    """random records of GC-shifted stretches, real DNA and N runs (two to four GC classes per record) for the human and the
    saccharomyces model, exact mode: every Viterbi and every forward variable and the path of the live reference (a soak of 30 such
    cases incl. rice and nasonia, 4 classes, was clean)"""
    import random
    import struct
    import tarfile
    rng = random.Random(seed)
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(str(tmp_path))
    g = read_fasta(str(tmp_path / "genome.fa"))[0][1]

    def gc_dna(n, gc):
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))

    def mk():
        parts = []
        for k in range(rng.randint(3, 6)):
            L = rng.choice([1500, 3000, 4000, 6000])
            r = rng.random()
            if r < 0.6:
                parts.append(gc_dna(L, rng.choice([0.25, 0.33, 0.38, 0.42, 0.47, 0.52, 0.58, 0.65, 0.72])))
            elif r < 0.85:
                st = rng.randrange(0, len(g) - L)
                parts.append(g[st:st + L].upper())
            else:
                parts.append(gc_dna(L // 2, 0.4) + "N" * rng.choice([1, 30, 400]) + gc_dna(L // 2, 0.6))
        return "".join(parts)
    species = rng.choice(["human", "saccharomyces"])
    recs = [("r%d" % k, mk()) for k in range(2)]
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    extra = ["--softmasking=0", "--UTR=off"]
    cells = str(tmp_path / "cells.bin")
    res, err = ref_harness(fa, species, extra, cells_file=cells)
    Fref = ref_forward(fa, species, extra)
    m = ax.Model(config_path(), species, softmasking="0", UTR="off", sample="0")
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "1")
    em = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, cells=True, forward=True)
    with open(cells, "rb") as f:
        for (name, seq), r, e, fr in zip(recs, res, em, Fref):
            n, S = struct.unpack("ii", f.read(8))
            vref = np.frombuffer(f.read(n * S * 8), dtype=np.float64).reshape(n, S)
            f.read(n * 4)
            for X, ref in ((e[3], vref), (e[5], fr)):
                assert np.array_equal(np.isfinite(X), np.isfinite(ref)), name
                both = np.isfinite(X)
                assert np.all(np.abs(X[both] - ref[both]) <= 1e-9 * np.abs(ref[both]) + 5e-9), name
            assert [(b, e2, emu_state_type(m.tables_ptr, st)) for b, e2, st in e[2]] == r["path"], name


def _multiclass_records(seed, n=5):
    """GC-shifted stretches, a gene, N runs: two to five GC classes per record"""
    import random
    rng = random.Random(seed)
    gene = dict(golden_inputs())["HS04636"].upper()

    def gc_dna(L, gc):
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(L))
    recs = []
    for k in range(n):
        parts = []
        for _ in range(rng.randint(3, 6)):
            r = rng.random()
            L = rng.choice([1500, 2500, 4000, 6000])
            if r < 0.65:
                parts.append(gc_dna(L, rng.choice([0.25, 0.33, 0.38, 0.42, 0.47, 0.52, 0.58, 0.65, 0.72])))
            elif r < 0.85:
                st = rng.randrange(0, len(gene) - L)
                parts.append(gene[st:st + L])
            else:
                parts.append(gc_dna(L // 2, 0.4) + "N" * rng.choice([1, 30, 400]) + gc_dna(L // 2, 0.6))
        recs.append(("mc%d_%d" % (seed, k), "".join(parts)))
    return recs


@pytest.mark.parametrize("species", ["human", "nasonia", "rice"])
def test_emulated_exact_mode_is_the_oracle_bit_for_bit(monkeypatch, species):
    """The oracle restates the reference's SnippetProbs cache (oracle/ghmm_twin.cc: snipGet; pinned against every cell of the real
    reference in test_oracle.py); the device replays it from what the first trellis run left (device/snipmemo.h).  Two independent
    routes to the same numbers: every cell, score and path bit-identical, on records with two to five GC classes."""
    m = ax.Model(config_path(), species, softmasking="0", UTR="off")
    recs = _multiclass_records({"human": 11, "nasonia": 12, "rice": 13}[species])
    if species == "human":
        byname = dict(golden_inputs())
        recs += [(k, byname[k].upper()) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
        recs.append(("path_case", multiclass_path_case()[0].upper()))
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "1")
    em = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, cells=True)
    differs = multi = 0
    for (name, seq), e in zip(recs, em):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True)
        assert rc == 0 and e[0] == 0, name
        multi += int(len(set(gc.tolist())) > 1)
        assert np.array_equal(e[3], V), name
        assert e[1] == lnv and [tuple(x) for x in e[2]] == [(b, en, st) for b, en, st, t in path], name
        rc, lnv0, path0, V0, _ = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=False)
        differs += int(not np.array_equal(V0, V))
    assert multi >= 2 and differs > 0, (multi, differs) # (and the cache is what the comparison is about: without it the cells of some record are different)


def test_emulated_exact_mode_decides_a_path(monkeypatch):
    """the record a randomised end-to-end soak found: with one class per end base the optimal path takes another acceptor site
    (ln V off by 0.03) than the reference; with the reference's snippet cache replayed (exact mode, the product's default) path and
    score are the reference's"""
    seq, opts, lnv, path = multiclass_path_case()
    m = ax.Model(config_path(), "human", sample="0", **opts)
    r0, = emu_decode(m.tables_ptr, [seq.upper()], m.n_states)
    assert [(b, e, emu_state_type(m.tables_ptr, st)) for b, e, st in r0[2]] != path and abs(r0[1] - lnv) > 0.01
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "1")
    r1, = emu_decode(m.tables_ptr, [seq.upper()], m.n_states)
    assert [(b, e, emu_state_type(m.tables_ptr, st)) for b, e, st in r1[2]] == path and abs(r1[1] - lnv) <= 1e-9 * abs(lnv)


@pytest.mark.parametrize("cfg", ["fly", "arabidopsis", "human1", "human1_sm"])
def test_emulated_sampling_matches_reference_paths(cfg):
    """posterior sampling of state paths (device/sampler.h on the emulator's forward matrix) against the REAL reference's
    NAMGene::getSampledPath (tests/golden/make_golden_sampled.py: 5 paths per record, one rand() stream over the records):
    every sampled path is the reference's, state by state -- the generator, the order of the options and the draw rule agree"""
    species, opts, _ = SAMPLED_CFGS[cfg]
    recs = sampled_records(cfg)
    gold = golden_sampled_paths(cfg)
    m = ax.Model(config_path(), species, **opts)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, samples=5)
    for (name, seq), r, g in zip(recs, res, gold):
        assert len(g) == 5
        for it in range(5):
            assert r[7][it] == g[it], (name, it)


@pytest.mark.parametrize("cfg", ["fly", "fly_sm", "human1", "fly_filter", "human_all", "fly_alt", "human_alt", "human_utr_alt"])
def test_emulated_sampling_gff_is_the_reference_binarys(cfg):
    """--sample=100 end to end on the CPU: emulator decode + forward + 99 sampled paths per record, the host gene stage
    (genes.cc: posteriorTranscripts) -> the GFF with posterior probabilities of genes, transcripts and CDS is byte-identical
    to the reference binary's (fly: sample = 100 is the species default; fly_sm: with the soft-masking bonus; human_all: incl. the
    records with several GC classes in a piece, where the reference's snippet cache is replayed; fly_alt, human_alt:
    --alternatives-from-sampling=true, the sampled transcripts as alternatives of the genes, with --maxtracks; human_utr_alt: UTR
    states, where alternatives of EQUAL mean state probability are common and their order is the one the reference's heap gives)"""
    species, opts, _ = SAMPLED_CFGS[cfg]
    n = int(opts.get("sample", 100))
    recs = sampled_records(cfg)
    m = ax.Model(config_path(), species, **opts)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, samples=n - 1)
    paths = [[(b, e, st, emu_state_type(m.tables_ptr, st)) for b, e, st in r[2]] for r in res]
    assert format_gff_sampled(m, recs, paths, [r[7] for r in res]) == golden_sampled_gff(cfg)


def test_emulated_sampling_with_noinframestop_is_the_reference_binarys():
    """--noInFrameStop=true with sampling on (--sample=30): the transcripts with a stop codon inside their CDS are dropped after the
    posterior probabilities were estimated -- the reference binary's GFF"""
    opts = NOINFRAMESTOP_CFGS["on_sampled"]
    recs = inframe_stop_records()
    m = ax.Model(config_path(), "fly", **opts)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], m.n_states, samples=29)
    paths = [[(b, e, st, emu_state_type(m.tables_ptr, st)) for b, e, st in r[2]] for r in res]
    gold = open(os.path.join(GOLDEN, "golden_noinframestop_on_sampled.gff")).read().splitlines()
    assert format_gff_sampled(m, recs, paths, [r[7] for r in res]) == gold


@needs_ref
def test_emulated_forward_with_several_gc_classes_matches_reference(tmp_path, monkeypatch):
    """pieces with several GC classes: near a class step the reference's short-intron interiors are products of cached chunks scored
    under different classes (SnippetProbs is not emptied at a step); device/snipmemo.h replays the cache from which cells are alive.
    With it every forward variable is the reference's to 1e-9 relative (without: up to 155 000 cells of a record off, by up to 4.7)"""
    byname = dict(golden_inputs())
    recs = [(k, byname[k]) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    Fref = ref_forward(fa, "human", ["--softmasking=0"])
    m = ax.Model(config_path(), "human", softmasking="0")
    res = emu_decode(m.tables_ptr, [s.upper() for _, s in recs], m.n_states, forward=True)
    for (name, seq), fr, r in zip(recs, Fref, res):
        F = r[5]
        assert np.array_equal(np.isfinite(F), np.isfinite(fr)), name
        both = np.isfinite(F)
        assert np.all(np.abs(F[both] - fr[both]) <= 1e-9 * np.abs(fr[both]) + 5e-9), name
    monkeypatch.setenv("AUGX_NO_MEMO", "1") # (and the replay is what does it)
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "0")
    res = emu_decode(m.tables_ptr, [recs[2][1].upper()], m.n_states, forward=True)
    both = np.isfinite(res[0][5])
    assert np.sum(np.abs(res[0][5][both] - Fref[2][both]) > 1e-9 * np.abs(Fref[2][both]) + 5e-9) > 1000


@pytest.mark.parametrize("seed", [74, 3, 58, 1007])
def test_emulated_segments_randomised(monkeypatch, seed):
    """random pieces (real and random DNA, N runs, GC-shifted stretches), random segment lengths, species and init / term kinds:
    every cell, score and path equal the sequential oracle.  (Seed 74 is the case a randomised soak found: a continuation that
    converged exactly where a later fix-up had stopped; seed 1007 has several N runs a dead start cannot converge in.)"""
    import random
    import tarfile
    import tempfile
    import bench
    rng = random.Random(seed)
    monkeypatch.setenv("AUGX_SEG_LEN", str(rng.choice([78000, 90000, 110000, 150000])))
    if rng.random() < 0.2:
        monkeypatch.setenv("AUGX_SEG_CHECK_TILES", "100000")
    d = tempfile.mkdtemp()
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(d)
    g = read_fasta(os.path.join(d, "genome.fa"))[0][1]
    sp = rng.choice(["human", "fly", "human_nosm", "arabidopsis"])
    species, opts = GOLDEN_CFGS[sp]
    m = ax.Model(config_path(), species, **opts)
    S = m.n_states

    def gc_dna(n, gc, r):
        return "".join(r.choice("GC") if r.random() < gc else r.choice("AT") for _ in range(n))

    seqs = []
    for i in range(rng.randint(1, 3)):
        parts, total = [], rng.randint(160000, 420000)
        while sum(map(len, parts)) < total:
            k = rng.random()
            if seed >= 1000:  # many long N runs
                parts.append(bench.synth_contigs(1, rng.randint(15000, 70000), rng.randint(0, 10**6))[0].decode())
                if rng.random() < 0.7:
                    parts.append("N" * rng.randint(20000, 90000))
            elif k < 0.35:
                a = rng.randint(0, 900000); parts.append(g[a:a + rng.randint(20000, 150000)])
            elif k < 0.7:
                parts.append(bench.synth_contigs(1, rng.randint(20000, 200000), rng.randint(0, 10**6))[0].decode())
            elif k < 0.8:
                parts.append("N" * rng.randint(1, 60000))
            else:
                parts.append(gc_dna(rng.randint(5000, 60000), rng.choice([0.3, 0.4, 0.5, 0.6, 0.7]), rng))
        s = "".join(parts)[:total]
        if rng.random() < 0.3:
            s = s.upper()
        seqs.append(s)
    ik, tk = rng.choice([(0, 0), (1, 1), (0, 1), (1, 0)])
    res = emu_decode(m.tables_ptr, seqs, S, cells=True, init_kind=ik, term_kind=tk)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True, init_kind=ik, term_kind=tk)
        assert (r[0] == 0) == (rc == 0)
        if rc == 0:
            assert r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
            assert set(s) == {"N"} or np.array_equal(r[3], V)


@needs_ref
@pytest.mark.parametrize("species,opts", [("human", {"UTR": "on", "softmasking": "0"}), ("fly", {})])
def test_emulated_utr_forward_and_sampling_match_the_reference(tmp_path, species, opts):
    """the 71-state model through the dense kernels (device/dense.h): every forward variable of the REAL reference within 1e-9
    (the same cells alive) and its sampled state paths, draw for draw -- UTR exon candidates are evaluated by the kernel and, for
    the sampler, on the host from the same site lists"""
    m = ax.Model(config_path(), species, sample="100", **opts)
    S = m.n_states
    assert S == 71
    ex = dict(golden_inputs())
    recs = [(k, ex[k]) for k in ("HS04636", "HS08198", "short600", "trunc_both", "trunc_right", "iupac")] + [("rnd", random_dna(12000, 77))]
    fa = str(tmp_path / "x.fa")
    write_fasta(fa, recs)
    extra = ["--%s=%s" % kv for kv in opts.items()]
    mats = ref_forward(fa, species, extra)
    smp = ref_samples(fa, species, extra, n=4)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, forward=True, samples=4)
    for (name, seq), R, rs, e in zip(recs, mats, smp, res):
        F, esm = e[5], e[7]
        assert np.array_equal(np.isfinite(R[1:]), np.isfinite(F[1:])), name
        both = np.isfinite(R) & np.isfinite(F)
        assert np.all(np.abs(R[both] - F[both]) <= 1e-9 * np.abs(R[both]) + 5e-9), name
        assert [[tuple(x) for x in r] for r in rs] == [list(p) for p in esm], name


@pytest.mark.parametrize("cfg", list(GENEMODEL_CFGS))
def test_emulated_two_intergenic_states(cfg):
    """--genemodel=atleastone / exactlyone through the dense kernels: every cell, score, path and status equal to the oracle's
    (records without a feasible path included), and the GFF of the sampled run the reference binary's"""
    species, opts = GENEMODEL_CFGS[cfg]
    m = ax.Model(config_path(), species, **dict(opts, sample="0"))
    S = m.n_states
    recs = golden_inputs()
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, cells=True)
    for (name, seq), (st, lnv, path, V, cls) in zip(recs, res):
        rc, lnv2, path2, V2, gc = twin_decode(m.tables_ptr, seq, S, cells=True)
        assert st == rc and (rc != 0 or (lnv == lnv2 and path == [(b, e, s) for b, e, s, t in path2])), name
        assert np.array_equal(V, V2), name
    assert sum(r[0] == ax.AUGX_E_NOPATH for r in res) == 3
    # the species' defaults through the host stage (fly: sample = 100)
    m2 = ax.Model(config_path(), species, **opts)
    ns = int(m2.option("sample") or 0)
    recs = genemodel_records()
    res = emu_decode(m2.tables_ptr, [s for _, s in recs], S, samples=max(ns - 1, 0)) if ns > 1 else emu_decode(m2.tables_ptr, [s for _, s in recs], S)
    tys = [emu_state_type(m2.tables_ptr, s) for s in range(S)]
    paths = [[(b, e, s, tys[s]) for b, e, s in r[2]] for r in res]
    gold = open(os.path.join(GOLDEN, "golden_genemodel_%s.gff" % cfg)).read().splitlines()
    out = format_gff_sampled(m2, recs, paths, [r[7] for r in res]) if ns > 1 else format_gff(m2, recs, paths)
    assert out == gold


@needs_ref
@pytest.mark.parametrize("species,maxdiff", [("fly", None), ("human", None), ("human", "0")])
def test_emulated_utr_alternatives_drop_almost_identical_transcripts(tmp_path, species, maxdiff):
    """--UTR=on --alternatives-from-sampling=true: alternatives with the coding exons of a more probable one and a transcription
    start / end within /Constant/almost_identical_maxdiff bases of its are dropped (AltGene::deleteSuboptimalTranscripts); the
    same transcripts as the reference binary's (human: 329 of 393 survive), in its order but for alternatives of EQUAL mean state
    probability"""
    import re, subprocess
    opts = {"UTR": "on", "alternatives-from-sampling": "true", "sample": "100", "softmasking": "0"}
    if maxdiff is not None:
        opts["/Constant/almost_identical_maxdiff"] = maxdiff
    m = ax.Model(config_path(), species, **opts)
    S = m.n_states
    ex = dict(golden_inputs())
    recs = [(k, ex[k]) for k in ("HS04636", "HS08198")] + [("rnd", random_dna(30000, 5))]
    fa = str(tmp_path / "x.fa")
    write_fasta(fa, recs)
    res = emu_decode(m.tables_ptr, [s for _, s in recs], S, samples=99)
    tys = [emu_state_type(m.tables_ptr, s) for s in range(S)]
    paths = [[(b, e, s, tys[s]) for b, e, s in r[2]] for r in res]
    out = format_gff_sampled(m, recs, paths, [r[7] for r in res])
    ref = subprocess.run([REF_AUGUSTUS, "--AUGUSTUS_CONFIG_PATH=" + config_path(), "--species=" + species] +
                         ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True).stdout
    refl = gff_body(ref)
    norm = lambda ls: sorted(re.sub(r"g(\d+)\.t\d+", r"g\1.tX", l) for l in ls if not l.startswith("#"))
    assert sum("\ttranscript\t" in l for l in out) >= 10
    if species == "fly":
        assert out == refl
    else:
        assert norm(out) == norm(refl)


@needs_ref
@pytest.mark.parametrize("opts", [{"genemodel": "exactlyone", "softmasking": "0"}, {"UTR": "on", "softmasking": "0"}])
def test_emulated_dense_kernels_with_several_gc_classes_against_the_reference(tmp_path, opts):
    """the dense kernels on pieces with several GC classes, with the reference's call-history caches replayed from the dense matrix:
    the same cells alive as in the REAL reference, every forward variable within 1e-9, and its sampled state paths, draw for draw.
    Two intergenic states: the snippet cache (snipmemo.h `dense`; before: up to 4e-4 off, another 14th path).  UTR states: also
    tssProbsPlus and the aSSProb memo (dense.h: k1TssReplay, assmemo.h; before round 6: 0 / 26 / 83 777 / 4 forward variables of
    the four records off by up to 1.6e-4 relative)."""
    byname = dict(golden_inputs())
    recs = [(k, byname[k]) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    extra = ["--%s=%s" % kv for kv in opts.items()]
    Fref = ref_forward(fa, "human", extra)
    smp = ref_samples(fa, "human", extra, n=6)
    m = ax.Model(config_path(), "human", sample="100", **opts)
    res = emu_decode(m.tables_ptr, [s.upper() for _, s in recs], m.n_states, forward=True, samples=6)
    for (name, seq), fr, rs, r in zip(recs, Fref, smp, res):
        F = r[5]
        assert np.array_equal(np.isfinite(F[1:]), np.isfinite(fr[1:])), name
        both = np.isfinite(F) & np.isfinite(fr)
        rel = np.abs(F[both] - fr[both]) / (np.abs(fr[both]) + 1e-300)
        assert np.all(np.abs(F[both] - fr[both]) <= 1e-9 * np.abs(fr[both]) + 5e-9), (name, float(rel.max()))
        assert [[tuple(x) for x in q] for q in rs] == [list(p) for p in r[7]], name


def test_emulated_dense_exact_mode_is_the_oracle_bit_for_bit(monkeypatch):
    """default mode of the dense kernels on multi-class records: emulator (replay after the run, from the dense matrix) == twin
    (cache inside the loop), every cell; and the replay changes cells"""
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "1")
    byname = dict(golden_inputs())
    seqs = [byname[k].upper() for k in ("multigc_gene", "multigc_rand")] + [s for _, s in _multiclass_records(5)[:2]]
    for opts in ({"UTR": "on", "sample": "0", "softmasking": "0"}, {"genemodel": "atleastone", "sample": "0", "softmasking": "0"}):
        m = ax.Model(config_path(), "human", **opts)
        S = m.n_states
        res = emu_decode(m.tables_ptr, seqs, S, cells=True)
        differs = 0
        for s, r in zip(seqs, res):
            rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True, cache=True)
            assert r[0] == rc
            if rc == 0:
                assert r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path] and np.array_equal(r[3], V)
                differs += int(not np.array_equal(twin_decode(m.tables_ptr, s, S, cells=True, cache=False)[3], V))
        assert differs > 0


def test_emulated_near_tie_counter_flags_the_soak_case():
    """the exam window of soak case 5010 (tests/soak_cli.py): two copies of one single-exon gene 164 bases apart -- in exact arithmetic a tie
    between staying intergenic and coming out of the second copy; the reference's rounding and ours decide it differently.  The chain
    wavefront flags the cell, the back-trace counts it; the reference's examples and random DNA have no such cell"""
    import ctypes
    import soak_cli
    _, g = soak_cli.real_dna()
    recs, species, opts = soak_cli.make_case(5010, g)
    m = ax.Model(config_path(), "human", softmasking="0")
    ex = dict(golden_inputs())
    seqs = [recs[1][1][:50000].upper(), ex["HS04636"], ex["HS08198"], random_dna(60000, 3)]
    emu_decode(m.tables_ptr, seqs, m.n_states)
    E = ctypes.CDLL(EMU_LIB)
    assert [E.emu_near_ties(i) for i in range(4)] == [1, 0, 0, 0]
    # the same stretch through the dense kernels (two intergenic states): flagged in their chain runs, counted by their back-trace
    m2 = ax.Model(config_path(), "human", softmasking="0", genemodel="atleastone", sample="0")
    res = emu_decode(m2.tables_ptr, [seqs[0][8000:16000], ex["HS04636"]], m2.n_states)
    assert [E.emu_near_ties(i) for i in range(2)] == [1, 0]
    assert res[0][2] == [(b, e, s) for b, e, s, t in twin_decode(m2.tables_ptr, seqs[0][8000:16000], m2.n_states)[2]]


@pytest.mark.parametrize("species", ["Vitrella_brassicaformis", "maize"])
def test_emulated_47_state_models_the_trellis_layout_was_not_built_for(species):
    """a 47-state model whose windows the wavefront layout of the trellis kernel refuses (Vitrella: an equalD state that looks back 63
    bases; maize: an acceptor window of 64 bases) goes to the state-graph driven dense kernels (layout.h: modelIsDense): every cell,
    score and path equal to the oracle twin (round 6; against the reference binary: tests/test_gpu_parity.py, tests/sweep_species.py)"""
    m = ax.Model(config_path(), species, UTR="off", sample="0", softmasking="0")
    S = m.n_states
    assert S == 47
    ex = dict(golden_inputs())
    seqs = [ex[k].upper() for k in ("HS04636", "withN", "trunc_both", "multigc_levels")] + [random_dna(15000, 91)]
    res = emu_decode(m.tables_ptr, seqs, S, cells=True)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
        assert np.array_equal(r[3], V)


def test_emulated_gc_donor_sites():
    """/IntronModel/allow_dss_consensus_gc (chlamy2011; reference Constant::dss_gc_allowed, include/geneticcode.hh:47-54): a donor site may
    read gc as well as gt, scored with the pattern probability times non_gt_dss_prob before the binning (src/intronmodel.cc:1232-1239):
    emulator == twin, every cell -- and the switch matters (round 6; against the reference binary: tests/test_gpu_parity.py)"""
    m = ax.Model(config_path(), "chlamy2011", UTR="off", sample="0", softmasking="0")
    assert m.n_states == 47
    ex = dict(golden_inputs())
    seqs = [ex[k].upper() for k in ("HS04636", "withN", "trunc_both", "multigc_levels")] + [random_dna(15000, 92)]
    res = emu_decode(m.tables_ptr, seqs, m.n_states, cells=True)
    gc_live = 0
    De = 4  # (chlamy2011: /Constant/dss_end 4 -- a longdss state that ends at j has its dinucleotide at j - De - 1, j - De)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
        assert np.array_equal(r[3], V)
        for st in (10, 15, 20):  # longdss0..2 (config/model/states_shadow.cfg)
            for j in np.nonzero(np.isfinite(V[:, st]))[0]:
                if j == 0:  # (column 0: the initial probabilities)
                    continue
                assert s[j - De - 1:j - De + 1] in ("GT", "GC"), (j, st)
                gc_live += s[j - De - 1:j - De + 1] == "GC"
    assert gc_live > 100  # (live donor-site cells on gc: the switch is in effect)


def test_emulated_utr_content_order_below_the_intron_order():
    """chlamy2011 with its own --UTR=on: UTR content tables of order 3 beside exon / intron content of order 4 (augx_tables::utr_k; the
    reference's index-for-index mixing with the intron table and the shifted intron pattern of the UTR intron states,
    src/utrmodel.cc:681-688,1255-1262): emulator == twin, every cell; the twin against the live reference: tests/test_oracle.py"""
    m = ax.Model(config_path(), "chlamy2011", sample="0", softmasking="0")
    assert m.n_states == 71
    ex = dict(golden_inputs())
    seqs = [ex[k].upper() for k in ("HS04636", "withN", "trunc_both")] + [random_dna(12000, 93)]
    res = emu_decode(m.tables_ptr, seqs, m.n_states, cells=True)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
        assert np.array_equal(r[3], V)


@pytest.mark.parametrize("species,table,opts", [("human", "6", {}), ("fly", "12", {"sample": "0"}), ("tetrahymena", "6", {})])
def test_emulated_translation_table(species, table, opts):
    """--translation_table: the stop codons and the start codons of the chosen genetic code in the kernels (dp.h: stopCodon3 /
    DevTables::stopMask, startMask) -- emulator == twin, every cell (the twin against the live reference: tests/test_oracle.py).
    tetrahymena: its own table 6 and intron content of order 3 beside exon content of order 4 (DevTables::kIn)"""
    m = ax.Model(config_path(), species, softmasking="0", translation_table=table, **opts)
    ex = dict(golden_inputs())
    seqs = [ex[k].upper() for k in ("HS04636", "withN", "trunc_both")] + [random_dna(12000, 94)]
    res = emu_decode(m.tables_ptr, seqs, m.n_states, cells=True)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
        assert np.array_equal(r[3], V)


def test_emulated_gc_class_of_windows_without_a_nucleotide():
    """kernels.h: k1WindowClass -- a GC window inside a long run of N takes the composition of the piece's first window (the reference's
    BaseCount keeps its relative frequencies when the counts sum to 0): emulator == twin, classes and every cell (the twin against the
    live reference: tests/test_oracle.py)"""
    m = ax.Model(config_path(), "human", softmasking="0")
    seqs = [n_window_record(5)[1], n_window_record(6, (0.62, 0.36), 15000)[1]]
    res = emu_decode(m.tables_ptr, seqs, m.n_states, cells=True)
    for s, r in zip(seqs, res):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r[0] == rc == 0 and r[1] == lnv and r[2] == [(b, e, st) for b, e, st, t in path]
        assert np.array_equal(r[3], V)
        assert gc[s.index("N") + s.count("N") // 2] == gc[0] and len(set(np.asarray(gc).tolist())) >= 2
