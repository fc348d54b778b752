"""GPU parity of posterior sampling (--sample > 0: the forward algorithm on the device, sampled paths, posterior probabilities in the
GFF score columns) against golden files of the REAL reference (tests/golden/make_golden_sampled.py)."""
import os
import subprocess
import tarfile

import pytest

pytestmark = pytest.mark.gpu

import augustus_amd as ax
from helpers import *

EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")


@pytest.mark.parametrize("cfg", list(SAMPLED_CFGS))
def test_cli_sampling_gff_identical_to_reference(tmp_path, cfg):
    """the executable with sampling on (fly: the species default): genes, transcripts and CDS with their posterior probabilities
    byte-identical to the reference binary's output"""
    species, opts, _ = SAMPLED_CFGS[cfg]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, sampled_records(cfg))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=" + species] + ["--%s=%s" % kv for kv in opts.items()] + [fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert gff_body(r.stdout) == golden_sampled_gff(cfg)
    assert r.stderr == ""


@pytest.mark.parametrize("cfg", ["fly", "human1_sm"])
def test_sampled_paths_are_the_references(cfg):
    """augx_decode_sampled through the C ABI: 5 sampled paths per record, one generator over the records -> the reference's
    NAMGene::getSampledPath paths, state by state"""
    species, opts, _ = SAMPLED_CFGS[cfg]
    recs = sampled_records(cfg)
    gold = golden_sampled_paths(cfg)
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m)
    soft = opts.get("softmasking", "1") != "0"
    res = ax.decode_sampled([d], [s if soft else s.upper() for _, s in recs], 5, ax.Rand(1))
    for (name, _), (dec, smp), g in zip(recs, res, gold):
        assert dec.status == 0, name
        assert [[(b, e, t) for b, e, _, t in sp] for sp in smp] == g, name


def test_sampling_batches_do_not_change_the_draws(monkeypatch):
    """batches of a few records each (AUGX_BATCH_BASES) and one batch give the same sampled paths: the draws follow the input order"""
    species, opts, _ = SAMPLED_CFGS["fly"]
    recs = sampled_records("fly")
    m = ax.Model(config_path(), species, **opts)
    d = ax.Decoder(m)
    seqs = [s.upper() for _, s in recs]
    one = ax.decode_sampled([d], seqs, 3, ax.Rand(1))
    monkeypatch.setenv("AUGX_BATCH_BASES", "30000")
    many = ax.decode_sampled([d], seqs, 3, ax.Rand(1))
    two = ax.decode_sampled([d, ax.Decoder(m)], seqs, 3, ax.Rand(1))
    for a, b, c in zip(one, many, two):
        assert a[1] == b[1] == c[1] and a[0].states == b[0].states == c[0].states


def test_cli_sampling_full_size_fly(tmp_path):
    """BASELINE config 2's input at the species default --sample=100: 1.0 Mbp of real DNA in 200 kb pieces, 99 sampled paths per
    piece (reference: 73 s on one core) -> byte-identical GFF"""
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(str(tmp_path))
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=fly", "--UTR=off", "--softmasking=0", str(tmp_path / "genome.fa")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert gff_body(r.stdout) == open(os.path.join(GOLDEN, "golden_big_fly_sampled.gff")).read().splitlines()


def test_cli_sample_too_low_is_the_references_message(tmp_path):
    """--sample below 10: the reference's message on stderr, no sampling (src/namgene.cc:58-62)"""
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, sampled_records("fly")[:2])
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, "--species=fly", "--UTR=off", "--softmasking=0", "--sample=5", fa], capture_output=True, text=True, env=env)
    assert r.returncode == 0
    assert r.stderr == "Error: Number of sample iterations is too low. (sample=5)\nI will not sample (sample=0) and will not estimate posterior probabilities.\n"
    r0 = subprocess.run([EXE, "--species=fly", "--UTR=off", "--softmasking=0", "--sample=0", fa], capture_output=True, text=True, env=env)
    assert gff_body(r.stdout) == gff_body(r0.stdout)
