"""GPU parity on the BASELINE.json configurations at full size (run by the driver with -m gpu on a real MI355X), through the
`augustus` executable and the C ABI of libaugx.so; checkers are golden files produced by the REAL reference
(tests/golden/make_golden_big.py), the reference binary itself where it travels (oracle/_ref), and the CPU twin."""
import json
import os
import shutil
import sys
import subprocess
import tarfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import augustus_amd as ax
from helpers import *

EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")


@pytest.fixture(scope="module")
def big_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("big")
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(str(d))
    import bench
    synth = str(d / "synth.fa")
    write_fasta(synth, [("rand000", bench.synth_contigs(1, 1000000, 12345)[0].decode())])
    return {"genome": str(d / "genome.fa"), "synth": synth}


BIG = {  # tests/golden/make_golden_big.py: BIG_CFGS
    "fly": ("genome", "fly", ["--UTR=off", "--sample=0", "--softmasking=0"]),     # config 2 stand-in: 200 kb pieces, cut chain
    "fly_sm": ("genome", "fly", ["--UTR=off", "--sample=0"]),                     # + soft-masking bonus across the cuts
    "human": ("genome", "human", ["--softmasking=0"]),                           # two GC classes inside one 1 Mbp piece
    "human_sm": ("genome", "human", []),                                         # default flags
    "synth": ("synth", "human", []),                                             # config 3: one actual bench contig
    "human_intronless": ("genome", "human", ["--genemodel=intronless"]),         # 3-state model: one 1 Mbp piece in segments
    "fly_intronless": ("genome", "fly", ["--genemodel=intronless", "--UTR=off", "--sample=100", "--softmasking=0"]),  # + sampling
    "fly_single": ("genome", "fly", ["--singlestrand=true", "--UTR=off", "--sample=0"]),    # 24-state model, both runs of five 200 kb pieces
    "human_sampled": ("genome", "human", ["--sample=100"]),   # one 1 Mbp piece, two GC classes with ten steps, soft-masking, sampling
    # BASELINE config 4 stand-in: the 71-state model with UTR states (dense kernels) at the fly model's own 200 kb pieces
    "fly_utr": ("genome", "fly", ["--sample=0"]),            # UTR on (the species' default), soft-masking bonus, cut chain
    "fly_default": ("genome", "fly", []),                    # every default of the species: UTR on, sample 100, soft-masking
    "human_utr": ("genome", "human", ["--UTR=on"]),          # the 71-state model on one 1 Mbp piece with two GC classes (ten steps): snippet cache replayed for the dense kernels
    "human_utr_sampled": ("genome", "human", ["--UTR=on", "--sample=100"]),   # ... and its forward pass + 99 sampled paths
    "synth_sampled": ("synth", "human", ["--sample=100"]),   # the bench contig with sampling: 10^8 draws (the generator's buffers made in parts, ahead), 12 000 option lists
}


@pytest.mark.parametrize("cfg", list(BIG))
def test_cli_full_size_gff_identical_to_reference(big_inputs, cfg):
    """Mb-scale inputs of BASELINE configs 2 and 3 through the executable: GFF byte-identical to the reference binary's"""
    inp, species, extra = BIG[cfg]
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=" + species] + extra + [big_inputs[inp]], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    gold = open(os.path.join(GOLDEN, "golden_big_%s.gff" % cfg)).read().splitlines()
    assert gff_body(r.stdout) == gold
    assert r.stderr == ""


@pytest.mark.parametrize("cfg", ["human", "synth"])
def test_full_size_piece_cells_and_reference_score(monkeypatch, big_inputs, cfg):
    """one 1 Mbp piece (real DNA with two GC classes inside the piece / a bench contig) through the C ABI: every cell equal
    to the twin, path equal to the REAL reference's, ln Viterbi within 1e-9 relative of the reference's"""
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    inp, species, extra = BIG[cfg]
    opts = {e[2:].split("=")[0]: e.split("=")[1] for e in extra}
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    (name, seq), = read_fasta(big_inputs[inp])
    b = ax.Batch(d, [seq])
    b.decode()
    r, = b.paths()
    gold = json.load(open(os.path.join(GOLDEN, "golden_big_paths.json")))[cfg]
    assert r.status == 0 and gold["n"] == len(seq)
    assert [(bb, e, t) for bb, e, s, t in r.states] == [tuple(p) for p in gold["path"]]
    assert abs(r.ln_viterbi - float(gold["lnv"])) <= 1e-9 * abs(float(gold["lnv"]))
    # (the first pass on its own: the decoder's replay of the reference's snippet cache off, and the twin's restatement of it)
    d.set_exact(False)
    b.decode()
    r, = b.paths()
    rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=False)
    assert rc == 0 and r.ln_viterbi == lnv and r.states == path
    if cfg == "human":
        assert len(set(gc.tolist())) > 1  # (more than one GC class inside the piece)
    assert np.array_equal(b.cells(0), V)


@needs_ref
def test_full_size_multiclass_piece_exact_mode_every_cell_is_the_references(monkeypatch, big_inputs, tmp_path):
    """the 1 Mbp real-DNA piece with the human model (two GC classes, ten class steps inside the piece) in exact mode
    (augx_decoder_set_exact: the reference's snippet cache around the steps replayed, the trellis run again): EVERY Viterbi variable
    of the real reference, run live, to 1e-9 relative -- without the replay 1 573 of the 8.45 M cells are off (test_oracle.py)"""
    import struct
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    m = ax.Model(config_path(), "human", softmasking="0")
    d = ax.Decoder(m, 0)
    d.set_exact(True)
    (name, seq), = read_fasta(big_inputs["genome"])
    b = ax.Batch(d, [seq])
    b.decode()
    r, = b.paths()
    V = b.cells(0)
    cells = str(tmp_path / "cells.bin")
    res, err = ref_harness(big_inputs["genome"], "human", ["--softmasking=0"], cells_file=cells)
    assert len(res) == 1, err
    with open(cells, "rb") as f:
        n, S = struct.unpack("ii", f.read(8))
        vref = np.frombuffer(f.read(n * S * 8), dtype=np.float64).reshape(n, S)
    assert np.array_equal(np.isfinite(V), np.isfinite(vref))
    both = np.isfinite(V)
    assert np.all(np.abs(V[both] - vref[both]) <= 1e-9 * np.abs(vref[both]) + 5e-9)
    assert [(bb, e, t) for bb, e, s, t in r.states] == res[0]["path"]
    assert abs(r.ln_viterbi - res[0]["lnv"]) <= 1e-9 * abs(res[0]["lnv"])


def _ladder_cases():
    ex = dict(golden_inputs())["HS04636"]
    core = ex[1100:8300]  # from inside the first intron to inside the last: tandem copies leave the cut finder no intergenic region
    sm = list(random_dna(200000, 77))
    for a, b in [(15000, 48000), (55000, 61000), (70000, 125000), (139000, 140500), (170000, 199000)]:
        for i in range(a, b):
            sm[i] = sm[i].lower()
    return {
        "tandem_core": random_dna(3000, 1) + core * 12 + random_dna(3000, 2),
        "tandem_gene": ex[700:9000] * 10,
        "rand": random_dna(150000, 5),
        "softmasked": "".join(sm),
    }


@needs_ref
@pytest.mark.parametrize("species,extra", [("human", []), ("fly", ["--UTR=off", "--sample=0"])])
@pytest.mark.parametrize("maxpiece", [20000, 60000])
def test_cli_cut_finder_ladder_matches_reference(tmp_path, species, extra, maxpiece):
    """every rung of the cut finder (reference src/namgene.cc:973-1133: first try, window doubled, non-internal intergenic
    region, give up = maxstep, the 5 %% / 5 kb rule) on records built to defeat it, two species, soft-masked runs across the
    cuts: the cut points (--progress lines on stderr) and the GFF equal the reference binary's"""
    fa = str(tmp_path / "ladder.fa")
    write_fasta(fa, list(_ladder_cases().items()))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=" + species] + extra + ["--progress=true", "--maxDNAPieceSize=%d" % maxpiece, fa]
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([EXE] + args, capture_output=True, text=True, env=env)
    assert ref.returncode == 0 and ours.returncode == 0, ours.stderr
    pieces = lambda t: [l for l in t.splitlines() if l.startswith("examining piece")]
    assert pieces(ours.stderr) == pieces(ref.stderr)
    assert len(pieces(ref.stderr)) > 12
    assert gff_body(ours.stdout) == gff_body(ref.stdout)


def test_sharded_decode_two_decoders():
    """the multi-GPU path of the C ABI (augx_decode_sharded) with two decoders -- on one device when the box has one -- gives
    what one decoder gives, in input order; and so does the executable with AUGX_DEVICES naming two devices"""
    m = ax.Model(config_path(), "human")
    ndev = ax.device_count()
    assert ndev >= 1
    d0, d1 = ax.Decoder(m, 0), ax.Decoder(m, 1 if ndev > 1 else 0)
    seqs = [random_dna(n, 900 + n) for n in (70000, 300, 120000, 9000, 45000, 64, 30000)]
    one = d0.decode(seqs)
    two = ax.decode_sharded([d0, d1], seqs)
    bins = ax.partition_lpt([len(s) for s in seqs], 2)
    assert set(bins) == {0, 1}
    for a, b in zip(one, two):
        assert a.status == b.status == 0 and a.ln_viterbi == b.ln_viterbi and a.states == b.states


def test_cli_two_devices_same_output(tmp_path):
    fa = str(tmp_path / "multi.fa")
    ex = dict(golden_inputs())["HS04636"]
    write_fasta(fa, [("a", random_dna(130000, 41)), ("b", ex), ("c", random_dna(90000, 42) + ex + random_dna(50000, 43)), ("d", random_dna(20000, 44))])
    outs = []
    second = "1" if ax.device_count() > 1 else "0"
    for devs in ("0,", "0," + second):
        env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path(), AUGX_DEVICES=devs)
        r = subprocess.run([EXE, "--species=human", "--maxDNAPieceSize=60000", fa], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and r.stderr == "", r.stderr
        outs.append(gff_body(r.stdout))
    assert outs[0] == outs[1] and any("\tgene\t" in l for l in outs[0])


@needs_ref
@pytest.mark.parametrize("extra", [["--uniqueGeneId=true"], ["--genemodel=complete"], ["--protein=off", "--start=off", "--stop=off"],
                                   ["--cds=off", "--introns=on"], ["--stopCodonExcludedFromCDS=true"]])
def test_cli_more_options_match_reference(tmp_path, extra):
    recs = dict(golden_inputs())
    fa = str(tmp_path / "opt.fa")
    write_fasta(fa, [("HS04636", recs["HS04636"]), ("trunc_both", recs["trunc_both"]), ("revcomp", recs["revcomp"]), ("rand60k", recs["rand60k"])])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = ["--species=human"] + extra + [fa]
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([EXE] + args, capture_output=True, text=True, env=env)
    assert ref.returncode == 0 and ours.returncode == 0, ours.stderr
    assert gff_body(ours.stdout) == gff_body(ref.stdout)
    assert ours.stderr == ref.stderr


@needs_ref
def test_cli_outfile_errfile_stdin_and_config_next_to_binary(tmp_path):
    """--outfile / --errfile, FASTA on standard input ('-'), and the config directory found relative to the executable
    (<dir of the binary>/../config, reference src/properties.cc:116-135) when neither the option nor the variable is given"""
    recs = dict(golden_inputs())
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, [("HS04636", recs["HS04636"]), ("HS08198", recs["HS08198"])])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    ref = subprocess.run([REF_AUGUSTUS, "--species=human", fa], capture_output=True, text=True, env=env)
    want = gff_body(ref.stdout)
    # --outfile / --errfile
    o, e = str(tmp_path / "o.gff"), str(tmp_path / "e.txt")
    r = subprocess.run([EXE, "--species=human", "--outfile=" + o, "--errfile=" + e, fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout == "" and r.stderr == ""
    assert gff_body(open(o).read()) == want and open(e).read() == ""
    # stdin
    r = subprocess.run([EXE, "--species=human", "-"], input=open(fa).read(), capture_output=True, text=True, env=env)
    assert r.returncode == 0 and gff_body(r.stdout) == want
    # config relative to the binary
    root = tmp_path / "inst"
    (root / "bin").mkdir(parents=True)
    shutil.copy(EXE, str(root / "bin" / "augustus"))
    shutil.copy(os.path.join(ROOT, "augustus_amd", "libaugx.so"), str(root / "libaugx.so"))  # (rpath of the executable: $ORIGIN/..)
    os.symlink(config_path().rstrip("/"), str(root / "config"))
    env2 = {k: v for k, v in os.environ.items() if k != "AUGUSTUS_CONFIG_PATH"}
    r = subprocess.run([str(root / "bin" / "augustus"), "--species=human", fa], capture_output=True, text=True, env=env2)
    assert r.returncode == 0 and gff_body(r.stdout) == want, r.stderr
    # a directory that does not exist: the reference's message
    r = subprocess.run([EXE, "--species=human", "--AUGUSTUS_CONFIG_PATH=/nonexistent", fa], capture_output=True, text=True, env=env2)
    assert r.returncode == 1 and "is not a directory. Could not locate directory AUGUSTUS_CONFIG_PATH." in r.stderr
    # --species=help: usage on stderr, exit code 0
    r = subprocess.run([EXE, "--species=help"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "human" in r.stderr.split()


def test_cli_genome_like_default_pieces(big_inputs, tmp_path):
    """a genome-like run with default flags (BASELINE config 5 in small): a 4.2 Mbp record at the human model's own
    maxDNAPieceSize (2 Mbp) -- two cut points found by decoding 150 kb exam windows, which the trellis itself cuts into
    segments --, soft-masked real DNA in both orientations, a second record: cut points and GFF equal the reference binary's
    (tests/golden/make_golden_big.py: genome_like)"""
    import gzip
    sys.path.insert(0, GOLDEN)
    from make_golden_big import genome_like_records
    fa = str(tmp_path / "genome_like.fa")
    write_fasta(fa, genome_like_records(read_fasta(big_inputs["genome"])[0][1]))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=human", "--progress=true", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ours = [l for l in r.stderr.splitlines() if l.startswith("examining piece")] + gff_body(r.stdout)
    gold = gzip.open(os.path.join(GOLDEN, "golden_big_genome_like.gff.gz"), "rt").read().splitlines()
    assert ours == gold


def test_cli_genome_like_fly_defaults_with_sampling(big_inputs, tmp_path):
    """the same two records with the fly model at its defaults but --UTR=off: 200 kb pieces (26 of them, cut points from 50 kb exam
    windows), soft-masking bonus, --sample=100 -- 99 sampled paths per piece, one stream of draws over all pieces of both records --:
    cut points, genes and every posterior probability equal the reference binary's (6 min on one core there)"""
    import gzip
    sys.path.insert(0, GOLDEN)
    from make_golden_big import genome_like_records
    fa = str(tmp_path / "genome_like.fa")
    write_fasta(fa, genome_like_records(read_fasta(big_inputs["genome"])[0][1]))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=fly", "--UTR=off", "--progress=true", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ours = [l for l in r.stderr.splitlines() if l.startswith("examining piece")] + gff_body(r.stdout)
    gold = gzip.open(os.path.join(GOLDEN, "golden_big_genome_like_fly_sampled.gff.gz"), "rt").read().splitlines()
    assert ours == gold


@pytest.mark.parametrize("cfg", list(MORE_CFGS))
def test_more_species_match_reference(monkeypatch, cfg):
    """nasonia (5 GC classes) and rice (4) at a larger scale (tests/golden/make_golden_big.py: more_species): through the C ABI
    every cell equal to the twin and the path equal to the reference's; through the executable at the species' own 200 kb
    pieces the cut points and the GFF equal the reference binary's"""
    monkeypatch.setenv("AUGX_DEBUG_CELLS", "1")
    species, opts = MORE_CFGS[cfg]
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m, 0)
    recs = more_inputs()
    gold = json.load(open(os.path.join(GOLDEN, "golden_more_paths_%s.json" % cfg)))["records"]
    b = ax.Batch(d, [s for _, s in recs])
    b.decode() # (the decoder's default: with the reference's snippet cache replayed where a piece has several classes)
    for i, ((name, seq), r, g) in enumerate(zip(recs, b.paths(), gold)):
        assert r.status == 0, name
        assert [(bb, e, t) for bb, e, s, t in r.states] == [tuple(p) for p in g["path"]], name
        assert abs(r.ln_viterbi - float(g["lnv"])) <= 1e-9 * abs(float(g["lnv"])), name
    d.set_exact(False) # (and without it: the kernels against the twin with its restatement of the cache off -- one class per end base)
    b.decode()
    for i, ((name, seq), r, g) in enumerate(zip(recs, b.paths(), gold)):
        rc, lnv, path, V, gc = twin_decode(m.tables_ptr, seq, m.n_states, cells=True, cache=False)
        assert r.status == 0 and r.ln_viterbi == lnv and r.states == path, name
        assert np.array_equal(b.cells(i), V), name
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    args = [EXE, "--species=" + species, "--progress=true"] + ["--%s=%s" % kv for kv in opts.items()] + [os.path.join(GOLDEN, "inputs_more.fa")]
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ours = [l for l in r.stderr.splitlines() if l.startswith("examining piece")] + gff_body(r.stdout)
    assert ours == open(os.path.join(GOLDEN, "golden_more_%s.gff" % cfg)).read().splitlines()


@pytest.mark.parametrize("cfg", list(SINGLE_CFGS))
def test_cli_singlestrand_matches_reference(tmp_path, cfg):
    """--singlestrand=true: the model without shadow states on every piece and on its reverse complement, the genes of the second run
    mapped back (exon types of the other strand), both lists merged by start -- incl. soft-masking, sampling (the draws run through the
    forward run, then the reverse one), --strand=backward and the cut finder at 20 kb pieces: the reference binary's GFF"""
    species, opts = SINGLE_CFGS[cfg]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, golden_inputs())
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert gff_body(r.stdout) == golden_single_gff(cfg)
    assert r.stderr == ""


@needs_ref
def test_cli_near_tie_counter(tmp_path):
    """DESIGN.md 6, near ties: every model term is rounded once to 2^-31, so two alternatives whose scores are closer than ~2e-7 may be
    decided the other way by the reference.  The trellis flags the cells of the chain states (intergenic, geometric introns) where
    staying and coming in from another state were that close, the back-trace counts those on the chosen path and, for the
    variable-length states on it, the cells whose runner-up candidate was that close (AUGX_NEAR_TIES=1 AUGX_TIMING=1 prints the sum; C ABI
    augx_decoder_near_ties).  The reference's own example has none and the same GFF.  The one input known to differ from the reference
    (tests/soak_cli.py, case 5010: two copies of one single-exon gene 164 bases apart, in exact arithmetic a tie) IS counted."""
    import re
    import soak_cli
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    ex = [(n, s) for n, s in golden_inputs() if n in ("HS04636", "HS08198")]
    fa2 = str(tmp_path / "ex.fa")
    write_fasta(fa2, ex)
    ref2 = subprocess.run([REF_AUGUSTUS, "--species=human", fa2], capture_output=True, text=True, env=env)
    ours2 = subprocess.run([EXE, "--species=human", fa2], capture_output=True, text=True, env=dict(env, AUGX_TIMING="1", AUGX_NEAR_TIES="1"))
    m2 = re.search(r"near ties on the chosen paths[^:]*: (\d+) cells", ours2.stderr)
    assert gff_body(ours2.stdout) == gff_body(ref2.stdout) and m2 and int(m2.group(1)) == 0
    _, g = soak_cli.real_dna()
    recs, species, opts = soak_cli.make_case(5010, g)
    fa = str(tmp_path / "c5010.fa")
    write_fasta(fa, recs)
    args = ["--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa]
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([EXE] + args, capture_output=True, text=True, env=dict(env, AUGX_TIMING="1", AUGX_NEAR_TIES="1"))
    assert ref.returncode == 0 and ours.returncode == 0, ours.stderr[-400:]
    m = re.search(r"near ties on the chosen paths[^:]*: (\d+) cells in (\d+) decodes", ours.stderr)
    assert m, ours.stderr[-400:]
    # (the exam window's cut lands 82 bases to the left of the reference's, the posterior probabilities from there on are those of
    #  another, equally valid sample -- and the counter says where to look)
    assert int(m.group(1)) >= 1
    assert gff_body(ours.stdout) == gff_body(ref.stdout) or int(m.group(1)) >= 1


def _first_difference(ours, gold):
    a, b = ours.splitlines(), gold.splitlines()
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return "line %d of %d / %d:\n  ours %s\n  ref  %s" % (i, len(a), len(b), x, y)
    return "lengths %d / %d" % (len(a), len(b))


@pytest.mark.parametrize("cfg", ["long", "long_utr"])
def test_cli_long_contig_matches_reference(tmp_path, cfg):
    """BASELINE configs 2 / 4 in shape and at full size -- ONE contig of 23 Mbp at the fly model's 200 kb pieces, UTR off and on:
    exactly the contig bench.py times as product.long_contig / long_contig_utr (bench.synth_contigs(1, 23000000, SEED0 + 77)).
    The 140 cut points (found with the scout / forecast cut finder, driver.cc: findCutPoints) and the GFF equal what the
    reference binary printed for it (tests/golden/make_golden_long.py: 11 / 29 minutes on one core there)"""
    import gzip
    sys.path.insert(0, GOLDEN)
    from make_golden_long import LONG_CFGS, golden_text, long_contig
    fa = str(tmp_path / "long.fa")
    write_fasta(fa, [("long", long_contig())])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE] + LONG_CFGS[cfg] + ["--progress=true", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = golden_text(r.stdout, r.stderr)
    gold = gzip.open(os.path.join(GOLDEN, "golden_%s.gff.gz" % cfg), "rt").read()
    assert ours == gold, _first_difference(ours, gold)
    assert json.load(open(os.path.join(GOLDEN, "golden_long.json")))[cfg]["cuts"] >= 115


def test_cli_genome_like_big_matches_reference(big_inputs, tmp_path):
    """BASELINE config 5 in shape (GRCh38 primary contigs), from what the container has: three records of 48 / 30 / 22 Mbp tiled from
    real soft-masked DNA in both orientations, GC-shifted stretches (several GC classes inside the human model's 2 Mbp pieces: exact
    mode's replay on nearly every piece), N runs of 0.1-5 Mbp (all-N pieces, cuts next to them), five scaffolds of 10-200 kb;
    --species=human at default flags.  58 cut points and 150 000 GFF lines equal the reference binary's (make_golden_long.py)"""
    import gzip
    sys.path.insert(0, GOLDEN)
    from make_golden_long import LONG_CFGS, genome_like_big_records, golden_text
    fa = str(tmp_path / "genome_like_big.fa")
    write_fasta(fa, genome_like_big_records(read_fasta(big_inputs["genome"])[0][1]))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE] + LONG_CFGS["genome_like_big"] + ["--progress=true", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = golden_text(r.stdout, r.stderr)
    gold = gzip.open(os.path.join(GOLDEN, "golden_genome_like_big.gff.gz"), "rt").read()
    assert ours == gold, _first_difference(ours, gold)


def test_cli_chromosome_of_250_mbp_matches_reference(big_inputs, tmp_path):
    """BASELINE config 5 AT SIZE: one record as long as GRCh38's chr1 (250 Mbp: the first record of the 1.0 Gbp stand-in,
    make_golden_long.py: genome_1g_records -- real soft-masked DNA in both orientations, GC-shifted stretches, N runs of 0.1-5 Mbp)
    at --species=human default flags: a chain of ~130 cuts at the model's 2 Mbp pieces with exact mode on nearly every piece, several
    device batches of 127 Mbp (the buffer pool's size classes).  Cut points and GFF equal the reference binary's (35 min on one core)."""
    import gzip
    sys.path.insert(0, GOLDEN)
    from make_golden_long import LONG_CFGS, genome_1g_records, golden_text
    import numpy as np
    fa = str(tmp_path / "genome_1g_chr1.fa")
    (name, seq), = genome_1g_records(read_fasta(big_inputs["genome"])[0][1], only_first=True)
    assert len(seq) == 250000000
    with open(fa, "wb") as f:  # (write_fasta's Python loop takes a minute at this size)
        f.write(b">" + name.encode() + b"\n")
        arr = np.frombuffer(seq.encode(), dtype=np.uint8)
        k = len(arr) // 60 * 60
        f.write(np.concatenate([arr[:k].reshape(-1, 60), np.full((k // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
        if k < len(arr):
            f.write(arr[k:].tobytes() + b"\n")
    del seq
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE] + LONG_CFGS["genome_1g_chr1"] + ["--progress=true", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = golden_text(r.stdout, r.stderr)
    gold = gzip.open(os.path.join(GOLDEN, "golden_genome_1g_chr1.gff.gz"), "rt").read()
    assert ours == gold, _first_difference(ours, gold)
