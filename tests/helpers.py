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
EMU_LIB = os.environ.get("AUGX_EMU_LIB") or os.path.join(ROOT, "build", "libaugx_emu.so")  # (AUGX_EMU_LIB: the emulator built with other build-time switches)

# tests that compare with the REAL reference (oracle/_ref, built by oracle/Makefile where /root/reference exists; the binaries
# travel to the GPU box with the working tree).  Without them such a test is skipped on the CPU -- and FAILS in the GPU suite or
# with AUGX_REQUIRE_REF=1 (tests/conftest.py): a parity run must not go green because its checker was missing.
import pytest
needs_ref = pytest.mark.needs_ref

_cfg_dir = None


def config_path():
    """AUGUSTUS_CONFIG_PATH-style directory: extracted from the committed fixture (works on the GPU box)."""
    global _cfg_dir
    if _cfg_dir is None:
        tar = os.path.join(GOLDEN, "config_min.tar.gz")
        more = os.path.join(GOLDEN, "config_more.tar.gz")  # two more species (nasonia: 5 GC classes, rice: 4)
        caeno = os.path.join(GOLDEN, "config_caeno.tar.gz")  # caenorhabditis (the reference's own test_ab_initio_prediction); Vitrella_brassicaformis, maize (47-state models the dense kernels take), chlamy2011 (gc donor sites, UTR tables of order 3), tetrahymena (translation table 6, intron content of order 3)
        d = os.path.join(tempfile.gettempdir(), "augx_config_%d_%d_%d_%d" % (os.getuid(), os.path.getsize(tar), os.path.getsize(more), os.path.getsize(caeno)))
        marker = os.path.join(d, "config", "model", "states_shadow.cfg")
        if not os.path.exists(marker):
            # several processes may get here at once (one rank per GPU, pytest-xdist): extract privately, publish with one
            # atomic rename; whoever loses the race uses the winner's copy
            import shutil
            tmp = tempfile.mkdtemp(prefix="augx_config_tmp_")
            for tf in (tar, more, caeno):
                with tarfile.open(tf) as t:
                    t.extractall(tmp)
            try:
                os.rename(tmp, d)
            except OSError:
                if os.path.exists(marker):
                    shutil.rmtree(tmp, ignore_errors=True)
                else:
                    d = tmp  # (a stale, incomplete directory is in the way: use the private copy)
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


def twin_tss0_carry(on):
    """consecutive twin_decode calls are the reference's consecutive sequences: entry 0 of its TSS caches lives on while they keep one
    length (oracle/ghmm_twin.cc: twin_set_tss0_carry); off: every call starts with empty caches"""
    twin().twin_set_tss0_carry(1 if on else 0)


def twin_decode(tables_ptr, seq, S, cells=False, init_kind=0, term_kind=0, cache=None):
    """CPU oracle (oracle/ghmm_twin.cc).  Returns (status, lnv, [(begin,end,state,type)], V or None, gc).
    cache: the restated SnippetProbs cache of the reference (multi-class pieces) on or off; by default it follows
    AUGX_EXACT_MULTICLASS like the emulator, so that a test that switches the device's exact mode off compares like with like."""
    n = len(seq)
    if cache is None:
        cache = os.environ.get("AUGX_EXACT_MULTICLASS", "1") != "0"
    twin().twin_set_snippet_cache(1 if cache else 0)
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


def ref_forward(fasta, species, extra=(), cfg=None):
    """ln of the REAL reference's forward variables (oracle/_ref/ref_harness --sample=100 --dumpforward): one [len, S] array
    per record, -inf where the cell is absent"""
    import struct
    dump = fasta + ".fwd.bin"
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg or config_path())
    out = subprocess.run([REF_HARNESS, "--species=" + species, "--sample=100"] + list(extra) + ["--dumpforward=" + dump, fasta],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    mats = []
    with open(dump, "rb") as f:
        while True:
            h = f.read(8)
            if len(h) < 8:
                break
            n, S = struct.unpack("ii", h)
            mats.append(np.frombuffer(f.read(n * S * 8), dtype=np.float64).reshape(n, S))
    os.remove(dump)
    return mats


def ref_samples(fasta, species, extra=(), n=5, cfg=None):
    """n sampled state paths per record from the REAL reference (oracle/_ref/ref_harness --dumpsamples: NAMGene::getSampledPath,
    rand() never seeded): [[(begin, end, type), ...] per sample] per record"""
    dump = fasta + ".smp.txt"
    env = dict(os.environ, AUGUSTUS_CONFIG_PATH=cfg or config_path())
    ex = [e for e in extra if not e.startswith("--sample=")]
    out = subprocess.run([REF_HARNESS, "--species=" + species, "--sample=100"] + ex + ["--dumpsamples=" + dump, "--nsamples=%d" % n, fasta],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    recs = []
    for l in open(dump):
        w = l.split()
        if w[0] == "SEQ":
            recs.append([])
        elif w[0] == "SAMPLE":
            recs[-1].append([])
        else:
            recs[-1][-1].append([int(x) for x in w[1:4]])
    os.remove(dump)
    return recs


# ---------------------------------------------------------------------------------------------------
# golden vectors (tests/golden/make_golden.py) and the lane-loop emulator of the device kernels
# ---------------------------------------------------------------------------------------------------
import json

GOLDEN_CFGS = {
    "human": ("human", {}),
    "human_nosm": ("human", {"softmasking": "0"}),
    "fly": ("fly", {"UTR": "off", "sample": "0", "softmasking": "0"}),
    "arabidopsis": ("arabidopsis", {"UTR": "off", "sample": "0", "softmasking": "0"}),
    "saccharomyces": ("saccharomyces", {"UTR": "off", "sample": "0", "softmasking": "0"}),
    "human_intronless": ("human", {"genemodel": "intronless", "softmasking": "0"}),          # 3 states; several GC classes in a piece
    "fly_intronless": ("fly", {"genemodel": "intronless", "UTR": "off", "sample": "0"}),     # with the soft-masking bonus
    # --UTR=on: the 71-state model with untranslated regions (dense kernels, device/dense.h)
    "human_utr": ("human", {"UTR": "on"}),
    "human_utr_nosm": ("human", {"UTR": "on", "softmasking": "0"}),
    "fly_utr": ("fly", {"sample": "0"}),                                                    # UTR on is the species' default
    "fly_utr_print": ("fly", {"sample": "0", "softmasking": "0", "print_utr": "on", "gff3": "on", "introns": "on"}),
}


# species pinned at a larger scale (tests/golden/make_golden_big.py: more_species): 300 kb of real DNA and records whose GC
# content runs through many classes, at the species' own maxDNAPieceSize (200 kb: one cut)
MORE_CFGS = {
    "nasonia": ("nasonia", {"UTR": "off", "sample": "0", "softmasking": "0"}),
    "rice": ("rice", {"UTR": "off", "sample": "0", "softmasking": "0"}),
    # two fungi whose equalD states look back 64 and 112 bases (dStateLen = d - splice windows): one to two tiles -- their predecessors
    # may lie in the tile before the current one, which is not in HBM yet when the current tile's long-lag values are staged
    "fusarium_graminearum": ("fusarium_graminearum", {"UTR": "off", "softmasking": "0"}),          # dStateLen 64; sample = 100 (default)
    "phanerochaete_chrysosporium": ("phanerochaete_chrysosporium", {"UTR": "off", "sample": "0", "softmasking": "0"}),  # 112
}


# gene models with two intergenic states (--genemodel=atleastone / exactlyone: states_shadow_2igenic.cfg, the synch state is the
# second intergenic state; dense kernels).  Records in which such a model has no feasible path (no room for a gene) are left out:
# the reference ends the whole run with an error there (tests/test_cli_errors.py)
GENEMODEL_CFGS = {
    "human_atleastone": ("human", {"genemodel": "atleastone"}),                                   # soft-masking bonus on
    "fly_exactlyone": ("fly", {"genemodel": "exactlyone", "UTR": "off", "softmasking": "0"}),    # sample = 100 (the species' default)
}


def gc_step_records(count, seed, parts=8, lo=2500, hi=6000):
    """records of uniform-random stretches whose GC content steps every few kb (two to five GC classes under most models): what the
    reference's call-history caches (SnippetProbs, tssProbsPlus, the aSSProb memo) are sensitive to"""
    rng = random.Random(seed)
    recs = []
    for i in range(count):
        p = []
        for k in range(parts):
            gc = rng.choice([0.35, 0.42, 0.5, 0.58, 0.65])
            p.append("".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(rng.randint(lo, hi))))
        recs.append(("gcsteps%d_%d" % (seed, i), "".join(p)))
    return recs


def n_window_record(seed=5, gcs=(0.36, 0.62), run=12000):
    """a stretch of low GC content, a run of N longer than the GC window (GCwinsize 10000), a stretch of high GC content: inside the run
    there are windows without a single nucleotide -- the reference classes them by the composition of the FIRST window of the piece
    (BaseCount::normalize leaves the relative frequencies alone when the counts sum to 0, src/motif.cc:204-212,561-575)"""
    rng = random.Random(seed)
    part = lambda n, gc: "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))
    return ("nwin%d" % seed, part(26000, gcs[0]) + "N" * run + part(24000, gcs[1]))


def genemodel_records():
    return [(n, s) for n, s in golden_inputs() if n not in ("allN", "short7", "short100")]


def more_inputs():
    return read_fasta(os.path.join(GOLDEN, "inputs_more.fa"))


# posterior sampling (tests/golden/make_golden_sampled.py): cfg -> (species, options, record names of inputs.fa or None = all).
# (human1: the records with one GC class; human_all: every record)
_ONE_CLASS = ("HS04636", "HS08198", "rand20k_b", "withN", "allN", "short7", "short100", "short600", "iupac", "trunc_left", "trunc_right",
              "trunc_both", "revcomp", "softmask_rand")
SAMPLED_CFGS = {
    "fly": ("fly", {"UTR": "off", "softmasking": "0"}, None),                 # sample = 100 is the species' default
    "fly_sm": ("fly", {"UTR": "off"}, None),
    "arabidopsis": ("arabidopsis", {"UTR": "off", "softmasking": "0", "sample": "100"}, None),
    "human1": ("human", {"sample": "100", "softmasking": "0"}, _ONE_CLASS),
    "human1_sm": ("human", {"sample": "50"}, _ONE_CLASS),
    # the probability filter (src/gene.cc:2489-2512) with the Viterbi transcripts not exempt: genes drop out, the numbering follows
    # all records incl. the ones with several GC classes in a piece (the reference's snippet cache around the class steps is replayed)
    "human_all": ("human", {"sample": "100", "softmasking": "0"}, None),
    "fly_filter": ("fly", {"UTR": "off", "softmasking": "0", "keep_viterbi": "false", "minexonintronprob": "0.3", "minmeanexonintronprob": "0.6"}, None),
    # --alternatives-from-sampling=true: the sampled transcripts that pass the filter stay, overlapping ones of one strand and reading
    # frame become the alternatives t1, t2, ... of a gene (sorted by mean state probability); --maxtracks bounds how many may overlap
    "fly_alt": ("fly", {"UTR": "off", "softmasking": "0", "alternatives-from-sampling": "true"}, None),
    "human_alt": ("human", {"sample": "100", "alternatives-from-sampling": "true", "maxtracks": "2"}, None),
    # UTR states + alternatives: transcripts that differ in a UTR end only often have EQUAL mean state probability (7 pairs among the
    # 18 of this record); which of two equals comes first follows the addresses of the reference's Gene objects -- falling in the
    # order of creation through the sampling loop of a record, restated as that in genes.cc: groupToGenes
    "human_utr_alt": ("human", {"UTR": "on", "sample": "30", "alternatives-from-sampling": "true"}, ("HS04636",)),
}


# --singlestrand=true (tests/golden/make_golden.py single): the 24-state model without shadow states, every piece decoded as it is and
# as its reverse complement, genes on opposite strands may overlap
SINGLE_CFGS = {
    "human": ("human", {"singlestrand": "true"}),                                              # default flags: soft-masking bonus
    "human_complete": ("human", {"singlestrand": "true", "genemodel": "complete", "softmasking": "0"}),
    "fly_sampled": ("fly", {"singlestrand": "true", "UTR": "off", "softmasking": "0"}),       # sample = 100: draws run forward, then reverse
    "fly_backward": ("fly", {"singlestrand": "true", "UTR": "off", "sample": "0", "strand": "backward"}),
    "fly_pieces": ("fly", {"singlestrand": "true", "UTR": "off", "sample": "0", "maxDNAPieceSize": "20000"}),  # cut finder + both runs per piece
    # alternatives from the sample in both runs: the genes of the reverse run are mirrored transcript by transcript
    "fly_alt": ("fly", {"singlestrand": "true", "UTR": "off", "softmasking": "0", "alternatives-from-sampling": "true", "maxtracks": "3"}),
}


# --noInFrameStop (tests/golden/make_golden_noinframestop.py): a gene whose CDS holds a stop codon put together by a long intron
NOINFRAMESTOP_CFGS = {
    "off": {"UTR": "off", "sample": "0", "softmasking": "0", "noInFrameStop": "false"},
    "on": {"UTR": "off", "sample": "0", "softmasking": "0", "noInFrameStop": "true"},
    "on_single": {"UTR": "off", "sample": "0", "softmasking": "0", "noInFrameStop": "true", "singlestrand": "true"},
    "on_sampled": {"UTR": "off", "softmasking": "0", "noInFrameStop": "true", "sample": "30"},
}


def inframe_stop_records():
    import gzip
    txt = gzip.open(os.path.join(GOLDEN, "inframe_stop.fa.gz"), "rt").read().split("\n")
    return [(txt[0][1:], txt[1]), (txt[2][1:], txt[3])]


def multiclass_path_case():
    """a 25 kb record (found by tests/soak_cli.py, seed 4025) whose OPTIMAL PATH under the human single-strand model depends on the
    reference's snippet cache across a GC-class step: (sequence, options, reference ln Viterbi, reference path)"""
    import gzip
    g = json.load(open(os.path.join(GOLDEN, "multiclass_path_case.json")))
    seq = gzip.open(os.path.join(GOLDEN, "multiclass_path_case.fa.gz"), "rt").read().split("\n")[1]
    return seq, g["options"], float(g["lnv"]), [tuple(p) for p in g["path"]]


def golden_single_gff(cfg):
    return open(os.path.join(GOLDEN, "golden_single_%s.gff" % cfg)).read().splitlines()


def sampled_records(cfg):
    names = SAMPLED_CFGS[cfg][2]
    recs = golden_inputs()
    if names is None:
        return recs
    byname = dict(recs)
    return [(k, byname[k]) for k in names]


def golden_sampled_gff(cfg):
    return open(os.path.join(GOLDEN, "golden_sampled_%s.gff" % cfg)).read().splitlines()


def gff_scores_apart(lines):
    """(lines with the score column masked, the scores) of GFF text: what is compared where the sample itself may differ"""
    struct, sc = [], []
    for l in lines:
        w = l.split("\t")
        if len(w) >= 9:
            sc.append(None if w[5] == "." else float(w[5]))
            w[5] = "*"
            struct.append("\t".join(w))
        else:
            struct.append(l)
    return struct, sc


def golden_sampled_paths(cfg):
    g = json.load(open(os.path.join(GOLDEN, "golden_sampled_paths_%s.json" % cfg)))
    return [[[tuple(st) for st in smp] for smp in r["samples"]] for r in g["records"]]


def golden_inputs():
    return read_fasta(os.path.join(GOLDEN, "inputs.fa"))


def golden_paths(cfg):
    g = json.load(open(os.path.join(GOLDEN, "golden_paths_%s.json" % cfg)))
    for r in g["records"]:
        r["lnv"] = float(r["lnv"])
        r["path"] = [tuple(p) for p in r["path"]]
    return g


def golden_gff(cfg):
    return open(os.path.join(GOLDEN, "golden_%s.gff" % cfg)).read().splitlines()


class _Piece(ctypes.Structure):
    _fields_ = [("seq", ctypes.c_char_p), ("len", ctypes.c_int64), ("init_kind", ctypes.c_int32), ("term_kind", ctypes.c_int32)]


_emu = None


def emu_decode(tables_ptr, seqs, S, cells=False, init_kind=0, term_kind=0, lib=None, forward=False, samples=0, seed=1, tss0=None):
    """Run the device kernel bodies on the CPU (tests/emu/emu.cc).  Returns [(status, lnv, path, V, cls)]
    (forward=True: [(status, lnv, path, V, cls, F, lnP)] with the ln forward matrix F and ln P(sequence);
    samples=n: [(status, lnv, path, V, cls, F, lnP, [n sampled paths of (begin, end, type)])], drawn from one rand() stream over seqs)."""
    forward = forward or samples > 0
    global _emu
    if lib is not None:
        _emu_lib = ctypes.CDLL(lib)
    elif _emu is None:
        _emu = ctypes.CDLL(EMU_LIB)
    n = len(seqs)
    P = (_Piece * n)()
    keep = [s.encode() if isinstance(s, str) else s for s in seqs]
    for i, b in enumerate(keep):
        P[i].seq, P[i].len = b, len(b)
        P[i].init_kind = init_kind[i] if isinstance(init_kind, (list, tuple)) else init_kind
        P[i].term_kind = term_kind[i] if isinstance(term_kind, (list, tuple)) else term_kind
    lnv = np.zeros(n)
    st = np.zeros(n, dtype=np.int32)
    cap = max(1024, max(len(s) for s in seqs) // 4 + 16)
    po = np.zeros((n, cap, 3), dtype=np.int32)
    pn = np.zeros(n, dtype=np.int32)
    cls = np.zeros(n, dtype=np.int32)
    tot = sum(len(s) for s in seqs)
    C = np.zeros(tot * S) if cells else None
    FW = np.zeros(tot * S) if forward else None
    lnF = np.zeros(n)
    E = _emu_lib if lib is not None else _emu
    E.emu_set_sampling(samples, seed)
    if tss0 is not None:  # [(forward, reverse) or None per piece]: the value of the TSS window at base 0 an earlier sequence left (BatchView::tss0)
        tv = np.array([[float('nan')] * 2 if t is None else list(t) for t in tss0], dtype=np.float64)
        E.emu_set_tss0(tv.ctypes.data_as(ctypes.c_void_p), len(tss0))
    else:
        E.emu_set_tss0(None, 0)
    rc = E.emu_decode(tables_ptr, P, n, lnv.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p),
                         po.ctypes.data_as(ctypes.c_void_p), cap, pn.ctypes.data_as(ctypes.c_void_p),
                         C.ctypes.data_as(ctypes.c_void_p) if cells else None, cls.ctypes.data_as(ctypes.c_void_p),
                         FW.ctypes.data_as(ctypes.c_void_p) if forward else None, lnF.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    out, w = [], 0
    for i, s in enumerate(seqs):
        V = C[w:w + len(s) * S].reshape(len(s), S) if cells else None
        Fm = FW[w:w + len(s) * S].reshape(len(s), S) if forward else None
        w += len(s) * S
        rec = (int(st[i]), float(lnv[i]), [tuple(int(x) for x in po[i, k]) for k in range(pn[i])], V, int(cls[i]))
        if samples:
            buf = np.zeros((max(1024, len(s) // 2 + 16), 3), dtype=np.int32)
            sm = []
            for it in range(samples):
                k = E.emu_sample_get(i, it, buf.ctypes.data_as(ctypes.c_void_p), len(buf))
                assert 0 <= k <= len(buf)
                sm.append([tuple(int(x) for x in buf[q]) for q in range(k)])
            out.append(rec + (Fm, float(lnF[i]), sm))
        else:
            out.append(rec + (Fm, float(lnF[i])) if forward else rec)
    E.emu_set_sampling(0, 1)
    return out


def emu_state_type(tables_ptr, s):
    return _emu.emu_state_type(tables_ptr, s)


def format_gff_sampled(model, recs, paths, samples):
    """GFF text of the product's gene-structure stage with posterior probabilities (augx_format_gff_sampled): paths[k] the Viterbi
    path [(begin, end, state, type)], samples[k] the sampled paths [(begin, end, type)] of record k"""
    import augustus_amd as ax
    L = ax.lib()
    out, gid = [], 1
    for k, ((name, seq), path, smp) in enumerate(zip(recs, paths, samples)):
        sts = (_St * max(1, len(path)))()
        for i, (b, e, st, t) in enumerate(path):
            sts[i].begin, sts[i].end, sts[i].state, sts[i].type = b, e, st, t
        arrs, ns, ptrs = [], (ctypes.c_int * max(1, len(smp)))(), (ctypes.POINTER(_St) * max(1, len(smp)))()
        for q, sp in enumerate(smp):
            a = (_St * max(1, len(sp)))()
            for i, (b, e, t) in enumerate(sp):
                a[i].begin, a[i].end, a[i].state, a[i].type = b, e, 0, t
            arrs.append(a)
            ns[q] = len(sp)
            ptrs[q] = ctypes.cast(a, ctypes.POINTER(_St))
        buf = ctypes.create_string_buffer(16 << 20)
        ng = ctypes.c_int()
        rc = L.augx_format_gff_sampled(model._h, name.encode(), seq.encode(), ctypes.c_int64(len(seq)), sts, len(path), len(smp), ptrs, ns,
                                       gid, buf, ctypes.c_int64(16 << 20), ctypes.byref(ng))
        assert rc == 0, ax.last_error() if hasattr(ax, "last_error") else rc
        out.append("# ----- prediction on sequence number %d (length = %d, name = %s) -----" % (k + 1, len(seq), name))
        out.append("#")
        st = model.option("strand")
        out.append("# Predicted genes for sequence number %d on %s" % (k + 1, "forward strand" if st == "forward" else "reverse strand" if st == "backward" else "both strands"))
        out += buf.value.decode().splitlines()
        if ng.value == 0:
            out.append("# (none)")
        gid += ng.value
        out.append("#")
    return out[:-1]


def format_gff(model, recs, paths):
    """GFF text of the product's gene-structure stage for externally supplied paths (augx_format_gff)."""
    import augustus_amd as ax
    L = ax.lib()
    L.augx_format_gff.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
    out, gid = [], 1
    for k, ((name, seq), path) in enumerate(zip(recs, paths)):
        sts = (_St * max(1, len(path)))()
        for i, (b, e, s, t) in enumerate(path):
            sts[i].begin, sts[i].end, sts[i].state, sts[i].type = b, e, s, t
        buf = ctypes.create_string_buffer(16 << 20)
        ng = ctypes.c_int()
        rc = L.augx_format_gff(model._h, name.encode(), seq.encode(), len(seq), sts, len(path), gid, buf, 16 << 20, ctypes.byref(ng))
        assert rc == 0, L.augx_last_error()
        out.append("# ----- prediction on sequence number %d (length = %d, name = %s) -----" % (k + 1, len(seq), name))
        out.append("#")
        st = model.option("strand")
        out.append("# Predicted genes for sequence number %d on %s" % (k + 1, "forward strand" if st == "forward" else "reverse strand" if st == "backward" else "both strands"))
        out += buf.value.decode().splitlines()
        if ng.value == 0:
            out.append("# (none)")
        gid += ng.value
        out.append("#")
    return out[:-1]


def gff_body(stdout_text):
    """Prediction part of an augustus stdout, as the reference's own test filter does (tests/short/utils/aug_out_filter.py)."""
    lines = stdout_text.splitlines()
    i0 = [k for k, l in enumerate(lines) if l.startswith("# ----- prediction")][0]
    return [l for l in lines[i0:] if not l.startswith("# command line")][:-1]
