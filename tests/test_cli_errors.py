"""Command-line errors of the drop-in executable against the reference binary (no GPU needed: every case fails before a device
is asked for).  Same exit code, same stderr (with the program name normalised), nothing on stdout where the reference prints
nothing -- reference Properties::init, src/properties.cc:66-330, and main, src/augustus.cc:94-248."""
import os
import subprocess

import pytest

from helpers import *

EXE = os.path.join(ROOT, "augustus_amd", "bin", "augustus")

CASES = [
    ["--species=human", "/nonexistent/input.fa"],
    ["--species=no_such_species", "IN"],
    ["--species=human", "--bogus=1", "IN"],
    ["--species=human", "--UTR", "IN"],
    ["--species=human", "--bogus", "IN"],
    ["--species=", "IN"],
    ["--genemodel", "--species=human", "IN"],
    ["--species=human", "IN", "IN"],
    ["IN"],
    ["--species=human"],
    ["--species=human", "--genemodel=weird", "IN"],
    ["--species=human", "--maxDNAPieceSize=10", "IN"],
    ["--species=human", "--AUGUSTUS_CONFIG_PATH=/nonexistent/config", "IN"],
]


@needs_ref
@pytest.mark.parametrize("args", CASES, ids=[" ".join(c) for c in CASES])
def test_cli_errors_match_reference(tmp_path, args):
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, golden_inputs()[:1])
    args = [fa if a == "IN" else a for a in args]
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=config_path())
    ref = subprocess.run([REF_AUGUSTUS] + args, capture_output=True, text=True, env=env)
    ours = subprocess.run([EXE] + args, capture_output=True, text=True, env=env)
    assert ref.returncode != 0 and ours.returncode == ref.returncode
    assert ours.stderr.replace(EXE, "AUGUSTUS") == ref.stderr.replace(REF_AUGUSTUS, "AUGUSTUS")
    if ref.stdout == "":
        assert ours.stdout == ""
