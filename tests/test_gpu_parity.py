"""GPU parity tests (run by the driver with -m gpu on a real MI355X).  Everything goes through the C ABI of
libaugx.so; the oracle (oracle/ghmm_twin.cc) and the golden vectors from the real reference are the checkers."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import augustus_amd as ax
from helpers import *


@pytest.fixture(autouse=True)
def _one_class_per_end_base(monkeypatch):
    """the kernels score a short-intron interior with the class of its end base; the replay of the reference's snippet cache on pieces
    with several GC classes (exact mode, the decoder's default) sits on top of that.  The tests here check the first pass on its own:
    decoders are created with exact mode off, and the twin's restatement of the cache is off with it (helpers.twin_decode follows the
    same switch).  The default mode against the twin's cache: tests/test_gpu_zz_exact_oracle.py."""
    monkeypatch.setenv("AUGX_EXACT_MULTICLASS", "0")


@pytest.fixture(scope="module")
def human():
    m = ax.Model(config_path(), "human")
    return m, ax.Decoder(m, 0)


@pytest.mark.parametrize("cfg", list(GOLDEN_CFGS))
def test_gpu_matches_golden_reference(cfg):
    """score within 1e-9 relative, integer coordinates bit-exact, against vectors produced by the REAL reference"""
    species, opts = GOLDEN_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    recs = golden_inputs()
    gold = golden_paths(cfg)["records"]
    res = d.decode([s for _, s in recs])
    n_checked = 0
    for (name, seq), r, g in zip(recs, res, gold):
        assert r.status == 0, name
        assert abs(r.ln_viterbi - g["lnv"]) <= 1e-9 * abs(g["lnv"]), name
        assert [(b, e, t) for b, e, s, t in r.states] == g["path"], name
        n_checked += 1
    assert n_checked == len(recs)


@pytest.mark.parametrize("cfg", ["human", "saccharomyces"])
def test_gpu_cells_bit_identical_to_oracle(monkeypatch, cfg):
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    m = ax.Model(config_path(), GOLDEN_CFGS[cfg][0], **GOLDEN_CFGS[cfg][1])
    d = ax.Decoder(m, 0)
    S = m.n_states
    # every golden input (real genes on both strands, truncated genes, N runs, IUPAC codes, records with two (human) or
    # three (saccharomyces) GC classes inside the piece) plus random pieces: the path alone can hide a wrong cell, so all
    # S x n cells are compared
    seqs = [s for _, s in golden_inputs()] + [random_dna(30000, 1), random_dna(5000, 2).lower(), random_dna(100, 3)]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True)
        assert r.status == 0 and r.ln_viterbi == lnv and r.states == path, i
        if set(s.upper()) != {"N"}:
            assert np.array_equal(b.cells(i), V), i


def test_gpu_interior_piece_kinds(human):
    m, d = human
    seq = random_dna(20000, 31337)
    for ik, tk in [(1, 1), (0, 1), (1, 0)]:
        r, = d.decode([seq], init_kind=ik, term_kind=tk)
        rc, lnv, path, _, _ = twin_decode(m.tables_ptr, seq, m.n_states, init_kind=ik, term_kind=tk)
        assert r.status == 0 and r.ln_viterbi == lnv and r.states == path


def test_gpu_full_size_contig_properties(human):
    """BASELINE config size (1 Mbp contigs): score bit-equal to the oracle, path tiles the sequence, batch order
    and batch composition do not change any result (pieces are independent)."""
    m, d = human
    big = random_dna(1000000, 2024)
    small = [random_dna(50000, 7), random_dna(70000, 8)]
    r1 = d.decode([big] + small)
    r2 = d.decode(small[::-1] + [big])
    assert r1[0].ln_viterbi == r2[2].ln_viterbi and r1[0].states == r2[2].states
    assert r1[1].states == r2[1].states and r1[2].states == r2[0].states
    rc, lnv, path, _, _ = twin_decode(m.tables_ptr, big, m.n_states)
    assert r1[0].ln_viterbi == lnv and r1[0].states == path
    pos = 1
    for b, e, s, t in r1[0].states:  # records tile [1, n-1] without gaps (negative-length states aside)
        assert b == pos or b > e
        pos = e + 1
    assert pos == len(big)


def test_cli_gff_identical_to_reference(tmp_path):
    """the augustus-compatible executable end to end: GFF byte-identical to the reference binary's output"""
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    fa = os.path.join(GOLDEN, "inputs.fa")
    for cfg, (species, opts) in GOLDEN_CFGS.items():
        args = [exe, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + ["--AUGUSTUS_CONFIG_PATH=" + config_path(), fa]
        r = subprocess.run(args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert gff_body(r.stdout) == golden_gff(cfg) and r.stderr == ""


@needs_ref
def test_cli_piece_cutting_matches_reference(tmp_path):
    """contig longer than maxDNAPieceSize: cut chain + synch-state pieces + global gene numbering"""
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    fa = str(tmp_path / "long.fa")
    ex = dict(golden_inputs())["HS04636"]
    # two records with cut chains of different length (the exam windows of all unfinished records are decoded together),
    # one of them with genes at and between its cut regions, and a record that needs no cut
    write_fasta(fa, [("long1", random_dna(260000, 555)),
                     ("long2", random_dna(50000, 557) + ex + random_dna(45000, 558) + ex + random_dna(30000, 559)),
                     ("tail", random_dna(30000, 556))])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=human", "--maxDNAPieceSize=60000", fa]
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([exe] + args, capture_output=True, text=True, env=env)
    assert ref.returncode == 0 and ours.returncode == 0, ours.stderr
    assert gff_body(ours.stdout) == gff_body(ref.stdout)


@pytest.mark.parametrize("extra", [["--strand=forward"], ["--strand=backward"], ["--predictionStart=2001", "--predictionEnd=8000"],
                                   ["--predictionStart=3000"], ["--gff3=on", "--introns=on", "--strand=minus"], ["--gff3=on", "--introns=on", "--strand=backward"],
                                   ["--softmasking=0", "--codingseq=on", "--exonnames=on"]])
@needs_ref
def test_cli_options_match_reference(tmp_path, extra):
    """command-line options of the path (strand filter, prediction range with coordinate offset, output variants):
    stdout of the prediction part byte-identical to the reference binary, empty stderr"""
    exe = os.path.join(ROOT, "augustus_amd", "bin", "augustus")
    recs = golden_inputs()
    byname = dict(recs)
    fa = str(tmp_path / "opt.fa")
    write_fasta(fa, [("HS04636", byname["HS04636"]), ("softmask_gene", byname["softmask_gene"]), ("revcomp", byname["revcomp"])])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=human"] + extra + [fa]
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([exe] + args, capture_output=True, text=True, env=env)
    assert ref.returncode == 0 and ours.returncode == 0, ours.stderr
    assert gff_body(ours.stdout) == gff_body(ref.stdout)
    assert ours.stderr == ref.stderr

@pytest.mark.parametrize("species", ["human", "fly", "arabidopsis"])
def test_gpu_ragged_lengths(species):
    """Edge lengths around the tile (64) and block (8) sizes, a one-base piece, ragged batch: bit-identical to the oracle."""
    m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
    d = ax.Decoder(m)
    seqs = [random_dna(n, 100 + n) for n in (1, 2, 7, 8, 9, 63, 64, 65, 127, 129, 600, 1031, 4099)]
    res = d.decode(seqs)
    for seq, r in zip(seqs, res):
        rc, lnv, path, _, _ = twin_decode(m.tables_ptr, seq, m.n_states)
        if rc == 0:
            assert r.status == 0 and r.ln_viterbi == lnv and r.states == path, len(seq)
        else:
            assert r.status != 0, len(seq)


@pytest.mark.parametrize("species", ["human", "fly"])
def test_gpu_adversarial_sequences(species):
    """dense splice sites, 20 kb open reading frames, repeats (see tests/test_emu.py): cells, score and path bit-identical"""
    from test_emu import adversarial_cases
    os.environ["AUGX_DEBUG_CELLS"] = "1"
    try:
        m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
        d = ax.Decoder(m, 0)
        cases = adversarial_cases()
        b = ax.Batch(d, list(cases.values()))
        b.decode()
        decoded = 0
        for i, ((name, seq), r) in enumerate(zip(cases.items(), b.paths())):
            rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True)
            decoded += 1
            assert r.status == 0 and r.ln_viterbi == lnv and r.states == path, name
            assert np.array_equal(b.cells(i), V), name
        assert decoded == len(cases)
    finally:
        del os.environ["AUGX_DEBUG_CELLS"]


@pytest.mark.parametrize("cfg", ["human", "fly", "arabidopsis", "saccharomyces"])
def test_gpu_randomised_pieces(cfg):
    """ragged batches of random composition -- real genes on both strands, N runs, soft-masked halves, stretches of
    different GC content (several GC classes inside a piece), both init/term kinds: score and path bit-identical to the oracle"""
    import random
    recs = dict(golden_inputs())
    ex = recs["HS04636"].upper()
    rc_ex = ex[::-1].translate(str.maketrans("ACGT", "TGCA"))

    def gc_dna(n, gc, rng):
        return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))

    species, opts = GOLDEN_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    rng = random.Random(sum(map(ord, cfg)))
    seqs = []
    for i in range(24):
        parts = []
        for k in range(rng.randint(1, 5)):
            r = rng.random()
            if r < 0.25:
                parts.append(ex if rng.random() < 0.5 else rc_ex)
            elif r < 0.35:
                parts.append("N" * rng.randint(1, 3000))
            else:
                parts.append(gc_dna(rng.randint(50, 30000), rng.choice([0.3, 0.4, 0.445, 0.5, 0.6, 0.7]), rng))
        s = "".join(parts)
        if rng.random() < 0.3:
            s = s.lower() if rng.random() < 0.3 else s[:len(s) // 2] + s[len(s) // 2:].lower()
        seqs.append(s)
    multi = 0
    for ik, tk in ((0, 0), (1, 1)):
        for s, r in zip(seqs, d.decode(seqs, init_kind=ik, term_kind=tk)):
            rc, lnv, path, _, gc = twin_decode(m.tables_ptr, s, m.n_states, init_kind=ik, term_kind=tk)
            multi += len(set(gc.tolist())) > 1
            assert (r.status == 0 and rc == 0 and r.ln_viterbi == lnv and r.states == path) or (r.status == ax.AUGX_E_NOPATH and rc != 0), len(s)
    assert multi > 0 or cfg in ("fly", "arabidopsis")  # (one GC class)


@pytest.mark.parametrize("blk", ["4", "2"])
def test_gpu_small_block_sizes(monkeypatch, blk):
    """block sizes 4 and 2 of the candidate / trellis kernels (species with short signal windows), forced on the human model:
    cells, score and path bit-identical to the oracle"""
    monkeypatch.setenv("AUGX_BLK", blk)
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    m = ax.Model(config_path(), "human")
    d = ax.Decoder(m, 0)
    recs = [(n, s) for n, s in golden_inputs() if n in ("HS04636", "withN", "rand60k", "softmask_gene", "multigc_gene", "multigc_two", "trunc_both")]
    seqs = [s for _, s in recs] + [random_dna(50000, 77)]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r.status == 0 and r.ln_viterbi == lnv and r.states == path, i
        assert np.array_equal(b.cells(i), V), i



@pytest.mark.parametrize("env", [{"AUGX_SEG_LEN": "100000"}, {"AUGX_SEG_LEN": "100000", "AUGX_SEG_CHECK_TILES": "100000"}, {}])
@pytest.mark.parametrize("species", ["human", "fly", "arabidopsis"])
def test_gpu_segment_parallel_trellis(monkeypatch, env, species):
    """pieces cut into segments decoded at once (tests/test_emu.py: test_emulated_segment_parallel_trellis), the give-up path
    forced, and the planner's own choice: every cell, the score and the path equal the sequential oracle bit for bit"""
    from test_emu import _segment_cases
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
    d = ax.Decoder(m, 0)
    seqs = _segment_cases()
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (seq, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True)
        assert r.status == 0 and rc == 0 and r.ln_viterbi == lnv and r.states == path, i
        if set(seq) != {"N"}:
            assert np.array_equal(b.cells(i), V), i
    b.decode()  # a batch decoded again (the bench does): the same result
    for i, (seq, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, _, _ = twin_decode(m.tables_ptr, seq, m.n_states)
        assert r.ln_viterbi == lnv and r.states == path, i


@needs_ref
@pytest.mark.parametrize("cfg", ["human_nosm", "fly"])
def test_gpu_forward_matches_reference(tmp_path, cfg):
    """the forward algorithm on the device (augx_batch_forward; groundwork of posterior sampling) against every forward variable
    of the REAL reference, run live: the same cells alive, ln F within 1e-9 relative"""
    from test_emu import _forward_records
    species, opts = GOLDEN_CFGS[cfg]
    byname = dict(golden_inputs())
    recs = _forward_records() + [("rand60k", byname["rand60k"])]
    if cfg == "human_nosm": # pieces with several GC classes: the reference's snippet cache around the class steps is replayed (snipmemo.h)
        recs += [(k, byname[k]) for k in ("multigc_gene", "multigc_two", "multigc_rand", "multigc_levels")]
    fa = str(tmp_path / "f.fa")
    write_fasta(fa, recs)
    Fref = ref_forward(fa, species, ["--%s=%s" % kv for kv in opts.items() if kv[0] != "sample"])
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    b = ax.Batch(d, [s for _, s in recs])
    b.decode()
    b.forward()
    for i, ((name, seq), fr, r) in enumerate(zip(recs, Fref, b.paths())):
        F, lnp = b.forward_cells(i)
        assert np.array_equal(np.isfinite(F), np.isfinite(fr)), name
        both = np.isfinite(F)
        assert np.all(np.abs(F[both] - fr[both]) <= 1e-9 * np.abs(fr[both]) + 5e-9), name
        assert lnp >= r.ln_viterbi


def test_gpu_exact_mode_decides_a_path():
    """the record whose optimal path depends on the reference's snippet cache across a GC-class step (tests/soak_cli.py, seed 4025):
    the decoder's default (exact mode) gives the reference's path and score, augx_decoder_set_exact(0) the one-class-per-end-base
    answer of the CPU twin"""
    seq, opts, lnv, path = multiclass_path_case()
    m = ax.Model(config_path(), "human", sample="0", **opts)
    d = ax.Decoder(m, 0)
    d.set_exact(True) # (the default of a decoder; the fixture of this file turned it off)
    b = ax.Batch(d, [seq.upper()])
    b.decode()
    r, = b.paths()
    assert [(bb, e, t) for bb, e, s, t in r.states] == path and abs(r.ln_viterbi - lnv) <= 1e-9 * abs(lnv)
    d.set_exact(False)
    b.decode()
    r0, = b.paths()
    rc, lnv0, path0, _, _ = twin_decode(m.tables_ptr, seq.upper(), m.n_states)
    assert r0.states == path0 and r0.ln_viterbi == lnv0 and [(bb, e, t) for bb, e, s, t in r0.states] != path


@pytest.mark.parametrize("opts,lo,hi", [({}, 0, 50000), ({"genemodel": "atleastone", "sample": "0"}, 8000, 16000)])
def test_gpu_near_ties_are_counted_through_the_c_abi(opts, lo, hi):
    """augx_decoder_count_near_ties / augx_decoder_near_ties on the device, both kernel families: the stretch of soak case 5010 with two
    overlapping copies of a single-exon gene has one near tie on its path (a chain-state decision), the reference's example none;
    the paths are the oracle's with the counter on"""
    import soak_cli
    _, g = soak_cli.real_dna()
    recs, _, _ = soak_cli.make_case(5010, g)
    m = ax.Model(config_path(), "human", softmasking="0", **opts)
    d = ax.Decoder(m, 0)
    d.count_near_ties(True)
    seqs = [recs[1][1][:50000].upper()[lo:hi], dict(golden_inputs())["HS04636"]]
    res = d.decode(seqs[:1])
    assert d.near_ties() == (1, 1)
    res += d.decode(seqs[1:])
    assert d.near_ties() == (1, 1)
    for s, r in zip(seqs, res):
        rc, lnv, path, _, _ = twin_decode(m.tables_ptr, s, m.n_states)
        assert r.status == 0 and r.ln_viterbi == lnv and r.states == path


@pytest.mark.parametrize("species", ["human", "fly"])
def test_gpu_role_specialised_equals_common_body(monkeypatch, species):
    """the default trellis kernel branches every wavefront into the instantiation of trellisPiece made for its role (k_trellis.hip:
    workgroup barriers in wave-divergent control flow -- sound only while all eight instantiations pass the same number of barriers);
    the build that counts near ties runs ONE common body.  Both on the device, cell for cell, on what the sequential emulator cannot
    judge: segments with fix-ups and continuations, runs of N that are jumped over (quiet tiles: three roles without flags between
    them), pieces with several GC classes."""
    from test_emu import _segment_cases
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    monkeypatch.setenv("AUGX_SEG_LEN", "100000")
    m = ax.Model(config_path(), *GOLDEN_CFGS[species][:1], **GOLDEN_CFGS[species][1])
    ex = dict(golden_inputs())
    seqs = _segment_cases() + [random_dna(120000, 41) + "N" * 150000 + random_dna(90000, 42) + "N" * 40000 + random_dna(60000, 43)]
    seqs += [s for _, s in gc_step_records(2, 11, parts=10, lo=15000, hi=40000)] + [ex["multigc_gene"], ex["multigc_rand"]]
    outs = []
    for ties in (False, True):
        d = ax.Decoder(m, 0)
        d.count_near_ties(ties)   # (batches created from now on run kTrellis<., ., true>: the common body)
        b = ax.Batch(d, seqs)
        b.decode()
        outs.append(([(r.status, r.ln_viterbi, r.states) for r in b.paths()], [b.cells(i) for i in range(len(seqs))]))
        d.close()
    assert outs[0][0] == outs[1][0]
    for i, (a, c) in enumerate(zip(outs[0][1], outs[1][1])):
        if set(seqs[i].upper()) != {"N"}:
            assert np.array_equal(a, c), i


@needs_ref
@pytest.mark.parametrize("species", ["Vitrella_brassicaformis", "maize", "chlamy2011"])
def test_cli_47_state_models_on_the_dense_kernels(tmp_path, monkeypatch, species):
    """two species the trellis kernel's wavefront layout refuses (equalD looking back 63 bases; a 64-base acceptor window) run on the
    dense kernel family instead (layout.h: modelIsDense, round 6): at their own defaults -- sample 100 -- with pieces cut at 30 kb, GFF
    byte-identical to the reference binary's, run live; cells of the device equal to the twin.  chlamy2011: the trellis kernel with
    donor sites that may read gc (/IntronModel/allow_dss_consensus_gc, round 6)."""
    import subprocess
    byname = dict(golden_inputs())
    recs = [(n, byname[n]) for n in ("HS04636", "multigc_levels", "softmask_gene", "trunc_both", "rand20k_b")]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=" + species, "--UTR=off", "--maxDNAPieceSize=30000", fa]
    ours = subprocess.run([os.path.join(ROOT, "augustus_amd", "bin", "augustus")] + args, capture_output=True, text=True, env=env)
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    assert ours.returncode == 0 and ref.returncode == 0 and ours.stderr == "", ours.stderr
    assert gff_body(ours.stdout) == gff_body(ref.stdout)
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")  # (chlamy2011: the trellis kernel keeps its cells only on request)
    m = ax.Model(config_path(), species, UTR="off", sample="0", softmasking="0")
    d = ax.Decoder(m, 0)
    seqs = [s.upper() for _, s in recs]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r.status == rc == 0 and r.ln_viterbi == lnv and r.states == path, i
        assert np.array_equal(b.cells(i), V), i


@needs_ref
@pytest.mark.parametrize("species,table,extra", [("human", "6", []), ("fly", "12", []), ("human", "4", ["--sample=20", "--UTR=on"]), ("tetrahymena", None, ["--maxDNAPieceSize=30000"])])
def test_cli_translation_table(tmp_path, species, table, extra):
    """--translation_table (round 6; reference src/geneticcode.cc:146-170): the executable against the reference binary, run live, at the
    species' defaults (fly: UTR states and 99 sampled paths) -- GFF byte-identical incl. the protein lines, and the lines the reference
    prints first about the nonstandard code"""
    import subprocess
    byname = dict(golden_inputs())
    recs = [(n, byname[n]) for n in ("HS04636", "revcomp", "softmask_gene", "trunc_both", "rand20k_b")]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=" + species] + (["--translation_table=" + table] if table else []) + extra + [fa]  # (tetrahymena: table 6 is its own; intron content of order 3)
    ours = subprocess.run([os.path.join(ROOT, "augustus_amd", "bin", "augustus")] + args, capture_output=True, text=True, env=env)
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    assert ours.returncode == 0 and ref.returncode == 0 and ours.stderr == ref.stderr == "", ours.stderr
    assert gff_body(ours.stdout) == gff_body(ref.stdout)
    warn = lambda t: [l for l in t.splitlines() if l.startswith("# Warning: Using nonstandard genetic code")]
    assert warn(ours.stdout) == warn(ref.stdout) and len(warn(ref.stdout)) > 0
    assert "# protein sequence" in ref.stdout


@pytest.mark.parametrize("opts", [{}, {"UTR": "on"}])
def test_gpu_gc_class_of_windows_without_a_nucleotide(monkeypatch, opts):
    """a run of N longer than the GC window between two GC regimes (kernels.h: k1WindowClass takes the composition of the piece's first
    window for a window that holds no nucleotide, as the reference's BaseCount does; found by the soak in round 6, seed 26011): device
    cells == twin, both kernel families (the twin against the live reference: tests/test_oracle.py)"""
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    m = ax.Model(config_path(), "human", softmasking="0", **opts)
    d = ax.Decoder(m, 0)
    seqs = [n_window_record(5)[1], n_window_record(6, (0.62, 0.36), 15000)[1], n_window_record(7, (0.45, 0.55), 30000)[1]]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r.status == rc == 0 and r.ln_viterbi == lnv and r.states == path, i
        assert np.array_equal(b.cells(i), V), i
