"""GPU parity of the 71-state model with untranslated regions (--UTR=on: dense kernels, device/dense.h), through the C ABI.
Checkers: the oracle twin (oracle/ghmm_twin.cc, pinned to the real reference cell by cell in tests/test_oracle.py) and the
golden vectors made from the real reference (tests/golden/make_golden.py: the `human_utr`, `fly_utr`, ... configurations)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import augustus_amd as ax
from helpers import *


# (the decoder's default, exact mode: on pieces with several GC classes the reference's snippet cache is replayed for the dense kernels
#  too -- device/snipmemo.h from the dense ln V matrix -- and the oracle twin runs the same cache inside its loop)


@pytest.mark.parametrize("species,opts", [("human", {"UTR": "on"}), ("human", {"UTR": "on", "softmasking": "0"}), ("fly", {"sample": "0"})])
def test_gpu_utr_cells_bit_identical_to_oracle(species, opts):
    """every cell of the S x n matrix, the score and the state path of every golden input (genes on both strands, truncated genes,
    N runs, soft-masked records, records with several GC classes) and of random pieces"""
    m = ax.Model(config_path(), species, **opts)
    assert m.n_states == 71
    d = ax.Decoder(m, 0)
    S = m.n_states
    seqs = [s for _, s in golden_inputs()] + [random_dna(30000, 1), random_dna(5000, 2).lower(), random_dna(100, 3)]
    if species == "human":  # (GC steps every few kb: tssProbsPlus and the aSSProb memo of the reference, replayed -- tests/test_memo_replay.py)
        seqs += [s for _, s in gc_step_records(3, 7)]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True)
        assert r.status == rc == 0 and r.ln_viterbi == lnv and r.states == path, i
        assert np.array_equal(b.cells(i), V), i


@needs_ref
@pytest.mark.parametrize("species,opts,multi", [("fly", {}, False), ("human", {"UTR": "on", "softmasking": "0"}, True),
                                                ("human", {"genemodel": "exactlyone", "softmasking": "0"}, True)])
def test_gpu_dense_forward_matches_reference(tmp_path, species, opts, multi):
    """the forward pass of the dense kernels (kDense<BLK, 1>: UTR states; two intergenic states) on the device against every forward
    variable of the REAL reference, run live: the same cells alive, ln F within 1e-9 relative -- also on the records with several GC
    classes: the reference's call-history caches are replayed from the dense matrix (the snippet cache; with UTR states also
    tssProbsPlus and the aSSProb memo, round 6: records whose GC content steps every few kb, where sites change their value
    during the sweep)."""
    ex = dict(golden_inputs())
    names = ["HS04636", "HS08198", "short600", "trunc_both", "trunc_right", "iupac"] + (["multigc_gene", "multigc_rand", "multigc_two", "multigc_levels"] if multi else [])
    recs = [(k, ex[k]) for k in names] + [("rnd", random_dna(12000, 77))] + (gc_step_records(2, 7) if multi and "UTR" in opts else [])
    fa = str(tmp_path / "x.fa")
    write_fasta(fa, recs)
    Fref = ref_forward(fa, species, ["--%s=%s" % kv for kv in opts.items()])
    m = ax.Model(config_path(), species, sample="100", **opts)
    d = ax.Decoder(m, 0)
    b = ax.Batch(d, [s for _, s in recs])
    b.decode()
    b.forward()
    for i, ((name, seq), fr, r) in enumerate(zip(recs, Fref, b.paths())):
        if r.status != 0:
            continue
        F, lnp = b.forward_cells(i)
        assert np.array_equal(np.isfinite(F[1:]), np.isfinite(fr[1:])), name
        both = np.isfinite(F) & np.isfinite(fr)
        assert np.all(np.abs(F[both] - fr[both]) <= 1e-9 * np.abs(fr[both]) + 5e-9), name
        assert lnp >= r.ln_viterbi


def test_gpu_utr_interior_piece_kinds_and_batch_order():
    m = ax.Model(config_path(), "fly", sample="0", softmasking="0")
    d = ax.Decoder(m, 0)
    seqs = [random_dna(20000, 31337), random_dna(7000, 5), random_dna(33000, 6)]
    for ik, tk in [(1, 1), (0, 1), (1, 0)]:
        res = d.decode(seqs, init_kind=ik, term_kind=tk)
        for s, r in zip(seqs, res):
            rc, lnv, path, _, _ = twin_decode(m.tables_ptr, s, m.n_states, init_kind=ik, term_kind=tk)
            assert r.status == 0 and r.ln_viterbi == lnv and r.states == path
    a = d.decode(seqs)
    b2 = d.decode(seqs[::-1])[::-1]
    assert [(x.ln_viterbi, x.states) for x in a] == [(x.ln_viterbi, x.states) for x in b2]


def test_gpu_utr_full_size_piece():
    """a piece of the fly model's own size (200 kb of real DNA with its soft-masking): score and path equal to the oracle"""
    import tarfile, io
    tf = tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz"))
    name = [n for n in tf.getnames() if n.endswith("genome.fa")][0]
    seq = "".join(l.strip() for l in io.TextIOWrapper(tf.extractfile(name)) if not l.startswith(">"))[300000:500000]
    m = ax.Model(config_path(), "fly", sample="0")
    d = ax.Decoder(m, 0)
    r, = d.decode([seq])
    rc, lnv, path, _, _ = twin_decode(m.tables_ptr, seq, m.n_states)
    assert r.status == 0 and r.ln_viterbi == lnv and r.states == path


EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")


def _run_cli(args, fa):
    import subprocess
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE] + args + [fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stderr == ""
    return r.stdout


def test_cli_utr_reproduces_the_golden_file_the_reference_holds(tmp_path):
    """`augustus --species=human --UTR=on --softmasking=0 examples/example.fa`: the one golden file of this path in the reference's
    own test suite (tests/short/examples/expected_results/test_utr_on/aug_utr_on.gff, committed as ref_held_aug_utr_on.gff),
    compared as the reference's test does (everything from the first '# ----- prediction' line on)"""
    recs = golden_inputs()[:2]
    assert [r[0] for r in recs] == ["HS04636", "HS08198"]
    fa = str(tmp_path / "example.fa")
    write_fasta(fa, recs)
    out = _run_cli(["--species=human", "--UTR=on", "--softmasking=0"], fa).splitlines()
    i0 = [k for k, l in enumerate(out) if "# ----- prediction" in l][0]
    ours = out[i0:]
    held = open(os.path.join(GOLDEN, "ref_held_aug_utr_on.gff")).read().splitlines()
    assert len(ours) == len(held)
    for a, b in zip(ours, held):
        if a != b:  # the only line that may differ is the echoed command line (paths)
            assert b.startswith("# ") and "--species=human" in b and "--UTR=on" in b, (a, b)


@pytest.mark.parametrize("cfg", ["human_utr", "human_utr_nosm", "fly_utr", "fly_utr_print"])
def test_cli_utr_gff_identical_to_reference(tmp_path, cfg):
    """the executable with UTR prediction on all golden inputs: GFF byte-identical to the reference binary's (tss / tts / exon
    lines, UTR lines in their own format, GFF3, the evidence block with the UTR counts)"""
    species, opts = GOLDEN_CFGS[cfg]
    fa = os.path.join(GOLDEN, "inputs.fa")
    out = _run_cli(["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()], fa)
    assert gff_body(out) == golden_gff(cfg)


@pytest.mark.parametrize("cfg", list(GENEMODEL_CFGS))
def test_gpu_two_intergenic_states(tmp_path, cfg):
    """--genemodel=atleastone / exactlyone (dense kernels): cells, scores, paths and statuses equal to the oracle; the executable's
    GFF the reference binary's"""
    species, opts = GENEMODEL_CFGS[cfg]
    m = ax.Model(config_path(), species, **dict(opts, sample="0"))
    d = ax.Decoder(m, 0)
    S = m.n_states
    recs = golden_inputs()
    b = ax.Batch(d, [s for _, s in recs])
    b.decode()
    for i, ((name, s), r) in enumerate(zip(recs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True)
        assert r.status == rc and (rc != 0 or (r.ln_viterbi == lnv and r.states == path)), name
        assert np.array_equal(b.cells(i), V), name
    fa = str(tmp_path / "gm.fa")
    write_fasta(fa, genemodel_records())
    out = _run_cli(["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()], fa)
    assert gff_body(out) == open(os.path.join(GOLDEN, "golden_genemodel_%s.gff" % cfg)).read().splitlines()


def test_gpu_utr_descriptor_buffer_grows(monkeypatch):
    """the descriptors of the UTR exon cells (kUtrDesc) go into a buffer sized by an estimate; one that turns out too small (forced:
    16 entries) is reported by the kernel's counter and the kernel runs again with a buffer of the size it needs -- same cells,
    also when the batch is decoded again"""
    monkeypatch.setenv("AUGX_UD_CAP", "16")
    m = ax.Model(config_path(), "human", UTR="on", softmasking="0")
    S = m.n_states
    d = ax.Decoder(m, 0)
    seqs = [s for n, s in golden_inputs() if n in ("HS04636", "HS08198", "trunc_both")] + [random_dna(30000, 1)]
    b = ax.Batch(d, seqs)
    for rep in range(2):
        b.decode()
        for i, (s, r) in enumerate(zip(seqs, b.paths())):
            rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, S, cells=True)
            assert r.status == rc == 0 and r.ln_viterbi == lnv and r.states == path, (rep, i)
            assert np.array_equal(b.cells(i), V), (rep, i)


@needs_ref
@pytest.mark.parametrize("seed", [11037, 11059, 11073, 11088])
def test_cli_soak_seeds_with_utr_states_gc_steps_and_sampling(tmp_path, monkeypatch, seed):
    """four cases of the randomised soak (`AUGX_SOAK_DENSE=2 tests/soak_cli.py 11000 90`: human --UTR=on --sample=100 on records whose GC
    class steps every few kb) that printed ANOTHER SAMPLE than the reference until round 6 (`0.24` where the reference prints `0.35`): the
    aSSProb memo of the reference is replayed through the sweep (assmemo.h), and it lives on through the back-tracking of the Viterbi
    path and the 99 sampled paths (sampler.h: memoStep).  GFF byte-identical to the reference binary's, run live."""
    import soak_cli
    monkeypatch.setenv("AUGX_SOAK_DENSE", "2")
    d, g = soak_cli.real_dna()
    recs, species, opts = soak_cli.make_case(seed, g)
    assert species == "human" and opts["UTR"] == "on" and opts["sample"] == "100"
    fa = str(tmp_path / "c.fa")
    write_fasta(fa, recs)
    args = ["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()]
    import subprocess
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    ref = subprocess.run([REF_AUGUSTUS] + args + [fa], capture_output=True, text=True, env=env)
    assert ref.returncode == 0
    assert gff_body(_run_cli(args, fa)) == gff_body(ref.stdout)


@needs_ref
def test_cli_utr_content_order_below_the_intron_order(tmp_path):
    """chlamy2011 at its OWN defaults (UTR on, sample 100): its UTR content tables have order 3 where exon / intron / intergenic content
    has order 4 (the `k` lines of the species' utr_probs file: UtrModel::k; also chlamydomonas, culex).  The reference then mixes entry i
    of the UTR table with entry i of the order-4 INTRON table (src/utrmodel.cc:681-688) and scores a base of a utr5intron / utr3intron
    state with the intron pattern that begins 3 -- not 4 -- bases before it (:1255-1262,1389-1396): both restated (augx_tables::utr_k,
    round 6).  GFF byte-identical to the reference binary's, run live; every cell of the device equal to the twin."""
    import subprocess
    byname = dict(golden_inputs())
    recs = [(n, byname[n]) for n in ("HS04636", "rand20k_b", "trunc_both", "revcomp", "multigc_levels", "softmask_gene")]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    for args in (["--species=chlamy2011"], ["--species=chlamy2011", "--softmasking=0", "--maxDNAPieceSize=30000", "--sample=0"]):
        ref = subprocess.run([REF_AUGUSTUS] + args + [fa], capture_output=True, text=True, env=env)
        assert ref.returncode == 0
        assert gff_body(_run_cli(args, fa)) == gff_body(ref.stdout)
        assert "five_prime_utr" not in ref.stdout and "\t5'-UTR\t" in ref.stdout  # (UTR is this species' default: there are some to compare)
    m = ax.Model(config_path(), "chlamy2011", sample="0", softmasking="0")
    assert m.n_states == 71
    d = ax.Decoder(m, 0)
    seqs = [s.upper() for _, s in recs] + [random_dna(30000, 5)]
    b = ax.Batch(d, seqs)
    b.decode()
    for i, (s, r) in enumerate(zip(seqs, b.paths())):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, s, m.n_states, cells=True)
        assert r.status == rc == 0 and r.ln_viterbi == lnv and r.states == path, i
        assert np.array_equal(b.cells(i), V), i


@needs_ref
def test_cli_tss_window_at_base_0_follows_the_reference_from_sequence_to_sequence(tmp_path, monkeypatch):
    """entry 0 of the reference's TSS caches lives on while the sequences it decodes keep one length (tests/test_memo_replay.py has the
    mechanism): (1) soak seed 28004 -- fly --UTR=on --sample=30 --maxDNAPieceSize=20000, two pieces of exactly 20 000 bases: 48 posterior
    probabilities of the second piece differed; (2) four records of ONE length at fly's defaults: every record after the first reads
    what the first left.  GFF byte-identical to the reference binary's, run live (driver.cc: tss0Src; augx_tss0 / augx_tss0_override)."""
    import soak_cli, subprocess
    for k, v in (("SOAK_NRUNS", "1"), ("AUGX_SOAK_DENSE", "2"), ("SOAK_REAL", "1")):
        monkeypatch.setenv(k, v)
    d, g = soak_cli.real_dna()
    recs, species, opts = soak_cli.make_case(28004, g)
    assert species == "fly" and opts["maxDNAPieceSize"] == "20000" and len(recs[0][1]) == 40000
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    fa = str(tmp_path / "c.fa")
    write_fasta(fa, recs)
    args = ["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()]
    ref = subprocess.run([REF_AUGUSTUS] + args + [fa], capture_output=True, text=True, env=env)
    assert ref.returncode == 0
    assert gff_body(_run_cli(args, fa)) == gff_body(ref.stdout)
    same = [("e%d" % i, g[o:o + 30000]) for i, o in enumerate((100000, 350000, 610000, 820000))]
    fa2 = str(tmp_path / "same.fa")
    write_fasta(fa2, same)
    for args in (["--species=fly"], ["--species=fly", "--sample=0", "--softmasking=0"]):
        ref = subprocess.run([REF_AUGUSTUS] + args + [fa2], capture_output=True, text=True, env=env)
        assert ref.returncode == 0
        assert gff_body(_run_cli(args, fa2)) == gff_body(ref.stdout)
