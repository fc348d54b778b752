"""Shared test plumbing: config fixture, oracle (twin / reference harness) wrappers, FASTA I/O.

The oracle libraries (oracle/libghmm_twin.so, oracle/_ref/*) are loaded ONLY from here, i.e. from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from augustus_amd/.
"""
import ctypes
import os
import random
import subprocess
import tarfile
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TWIN_LIB = os.path.join(ROOT, "oracle", "libghmm_twin.so")
REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REF_AUGUSTUS = os.path.join(ROOT, "oracle", "_ref", "augustus_ref")
EMU_LIB = os.path.join(ROOT, "build", "libaugx_emu.so")

_cfg_dir = None


def config_path():
    """AUGUSTUS_CONFIG_PATH-style directory: extracted from the committed fixture (works on the GPU box)."""
    global _cfg_dir
    if _cfg_dir is None:
        d = os.path.join(tempfile.gettempdir(), "augx_config_%d" % os.getuid())
        marker = os.path.join(d, "config", "model", "states_shadow.cfg")
        if not os.path.exists(marker):
            os.makedirs(d, exist_ok=True)
            with tarfile.open(os.path.join(GOLDEN, "config_min.tar.gz")) as t:
                t.extractall(d)
        _cfg_dir = os.path.join(d, "config") + "/"
    return _cfg_dir


def read_fasta(fn):
    recs, name, seq = [], None, []
    for line in open(fn):
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split()[0], []
        else:
            seq.append("".join(ch for ch in line if ch.isalpha()))
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def write_fasta(fn, recs, width=60):
    with open(fn, "w") as f:
        for name, s in recs:
            f.write(">%s\n" % name)
            for k in range(0, len(s), width):
                f.write(s[k:k + width] + "\n")


def random_dna(n, seed, alphabet="ACGT"):
    rng = random.Random(seed)
    return "".join(rng.choice(alphabet) for _ in range(n))


class _St(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int32), ("end", ctypes.c_int32), ("state", ctypes.c_int16), ("type", ctypes.c_int16)]


_twin = None


def twin():
    global _twin
    if _twin is None:
        _twin = ctypes.CDLL(TWIN_LIB)
    return _twin


def twin_decode(tables_ptr, seq, S, cells=False, init_kind=0, term_kind=0):
    """CPU oracle (oracle/ghmm_twin.cc).  Returns (status, lnv, [(begin,end,state,type)], V or None, gc)."""
    n = len(seq)
    V = np.empty((n, S)) if cells else None
    gc = np.empty(n, dtype=np.int32)
    cap = max(1024, n // 4 + 16)
    sts = (_St * cap)()
    ns, lnv = ctypes.c_int32(), ctypes.c_double()
    rc = twin().twin_decode(tables_ptr, seq.encode() if isinstance(seq, str) else seq, ctypes.c_int64(n), init_kind, term_kind,
                            V.ctypes.data_as(ctypes.c_void_p) if cells else None, gc.ctypes.data_as(ctypes.c_void_p),
                            sts, cap, ctypes.byref(ns), ctypes.byref(lnv))
    path = [(sts[i].begin, sts[i].end, sts[i].state, sts[i].type) for i in range(ns.value)]
    return rc, lnv.value, path, V, gc


def ref_harness(fasta, species, extra=(), cells_file=None, cfg=None):
    """The REAL reference through oracle/_ref/ref_harness.  Returns list of dicts per record."""
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg or config_path())
    cmd = [REF_HARNESS, "--species=" + species] + list(extra)
    if cells_file:
        cmd.append("--dumpcells=" + cells_file)
    cmd.append(fasta)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env)
    res, cur = [], None
    for line in out.stdout.splitlines():
        w = line.split()
        if not w:
            continue
        if w[0] == "SEQ":
            cur = {"name": w[1], "n": int(w[2]), "path": [], "lnv": None}
        elif w[0] == "LNV":
            cur["lnv"] = float(w[1])
        elif w[0] == "ST":
            cur["path"].append((int(w[1]), int(w[2]), int(w[3])))
        elif w[0] == "ERR":
            cur["err"] = line
        elif w[0] == "END":
            res.append(cur)
    return res, out.stderr
