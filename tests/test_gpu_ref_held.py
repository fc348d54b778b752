"""The golden files the REFERENCE'S OWN test suite holds for the single-genome ab-initio path, fed to the product executable exactly as
/root/reference/tests/short/examples/test_examples.py runs them (the files are committed byte for byte under tests/golden/ref_held_*):

  test_ab_initio_prediction  (:429-447)  augustus examples/autoAug/genome.fa --softmasking=1 --species=caenorhabditis
  test_format_and_error_out  (:449-472)  the same with --gff3=on --outfile=... --errfile=...
  (test_utr_on: tests/test_gpu_utr.py::test_cli_utr_reproduces_the_golden_file_the_reference_holds)

caenorhabditis' own defaults are `UTR on`, `sample 100`, `maxDNAPieceSize 200000`: the 71-state model (dense kernels), the forward pass +
99 sampled paths per piece, the cut finder's chain and the soft-masking bonus all at once, on 1 Mbp of real DNA.
The comparison is the reference's: everything from the first '# ----- prediction' line on, every line stripped
(tests/short/utils/aug_out_filter.py: pred), the files equal line for line -- but for the one line that echoes the command (paths)."""
import gzip
import os
import subprocess
import tarfile

import pytest

pytestmark = pytest.mark.gpu

from helpers import *

EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")


def ref_filter_pred(lines):
    """tests/short/utils/aug_out_filter.py: pred -- drop everything before the first line that holds '# ----- prediction', strip the rest"""
    i0 = [k for k, l in enumerate(lines) if "# ----- prediction" in l.strip()][0]
    return [l.strip() for l in lines[i0:]]


def held(name):
    return [l.strip() for l in gzip.open(os.path.join(GOLDEN, name), "rt").read().split("\n")[:-1]]


def same_but_for_the_command_line(ours, gold):
    assert len(ours) == len(gold), (len(ours), len(gold))
    for k, (a, b) in enumerate(zip(ours, gold)):
        if a != b:  # (the echoed command -- the line after '# command line:' -- holds the paths of the run)
            assert k > 0 and gold[k - 1] == "# command line:" and ours[k - 1] == "# command line:" and "--species=caenorhabditis" in a, (k, a, b)


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("autoAug")
    with tarfile.open(os.path.join(GOLDEN, "big_inputs.tar.gz")) as t:
        t.extractall(str(d))
    return str(d / "genome.fa")


def test_ab_initio_prediction(genome):
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    r = subprocess.run([EXE, genome, "--softmasking=1", "--species=caenorhabditis"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    same_but_for_the_command_line(ref_filter_pred(r.stdout.split("\n")[:-1]), held("ref_held_ab_initio_augustus.gff.gz"))


def test_format_and_error_out(genome, tmp_path):
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    out, err = str(tmp_path / "augustus_tmp.gff3"), str(tmp_path / "augustus.err")
    r = subprocess.run([EXE, genome, "--species=caenorhabditis", "--gff3=on", "--softmasking=1", "--outfile=" + out, "--errfile=" + err],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stderr == "" and r.stdout == "", (r.stdout[:200], r.stderr)
    assert os.path.isfile(out), "Output file was not created as expected!"
    same_but_for_the_command_line(ref_filter_pred(open(out).read().split("\n")[:-1]), held("ref_held_format_augustus.gff3.gz"))
    assert open(err).read() == ""   # (expected_results/test_format_and_error_out/augustus.err is empty)
