"""augustus_amd -- MI355X-native ab-initio GHMM Viterbi decode (AUGUSTUS hot path), Python binding.

A thin ctypes layer over the C ABI of ``libaugx.so`` (``include/augx.h``).  The decode itself runs in
hand-written HIP kernels (``augustus_amd/csrc/device``); this module only marshals buffers.  There is no CPU
decode path: creating a :class:`Decoder` without a HIP device raises :class:`AugxError`.

Mirrors the reference's call sequence for this path
(``Properties::init`` .. ``NAMGene()`` .. ``StateModel::readAllParameters()`` ..
``NAMGene::doViterbiPiecewise``; reference ``src/augustus.cc:111-176,420``).
"""
import ctypes
import sys
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUGX_LIB") or os.path.join(_HERE, "libaugx.so")  # (AUGX_LIB: a developer build of the same library)

AUGX_E_NODEVICE = -3
AUGX_E_HIP = -4
AUGX_E_UNSUPPORTED = -5
AUGX_E_NOPATH = -6
AUGX_E_NOMEM = -7
AUGX_E_RANGE = -8


class AugxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("augx error %d: %s" % (code, msg))
        self.code = code


class _Piece(ctypes.Structure):
    _fields_ = [("seq", ctypes.c_char_p), ("len", ctypes.c_int64), ("init_kind", ctypes.c_int32),
                ("term_kind", ctypes.c_int32)]


class _State(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int32), ("end", ctypes.c_int32), ("state", ctypes.c_int16),
                ("type", ctypes.c_int16)]


class _Path(ctypes.Structure):
    _fields_ = [("states", ctypes.POINTER(_State)), ("n_states", ctypes.c_int32), ("status", ctypes.c_int32),
                ("ln_viterbi", ctypes.c_double)]


_lib = None


def lib():
    """Load libaugx.so (built in-tree by ``__graft_entry__.build()``); fails loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("augustus_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`"
                              % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.augx_last_error.restype = ctypes.c_char_p
        L.augx_version.restype = ctypes.c_char_p
        L.augx_model_tables.restype = ctypes.c_void_p
        L.augx_model_tables.argtypes = [ctypes.c_void_p]
        L.augx_model_option.restype = ctypes.c_char_p
        L.augx_model_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.augx_model_destroy.argtypes = [ctypes.c_void_p]
        L.augx_device_count.restype = ctypes.c_int
        L.augx_partition_lpt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.augx_decode_sharded.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.augx_decoder_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.augx_decoder_destroy.argtypes = [ctypes.c_void_p]
        L.augx_decoder_set_share.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.augx_decoder_set_exact.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.augx_decoder_batch_capacity.restype = ctypes.c_int64
        L.augx_decoder_batch_capacity.argtypes = [ctypes.c_void_p]
        L.augx_batch_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.augx_batch_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.augx_batch_sync.argtypes = [ctypes.c_void_p]
        L.augx_batch_paths.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.augx_batch_cells.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.augx_batch_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.augx_batch_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.augx_batch_forward_cells.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.augx_batch_destroy.argtypes = [ctypes.c_void_p]
        L.augx_rand_create.restype = ctypes.c_void_p
        L.augx_rand_create.argtypes = [ctypes.c_uint]
        L.augx_rand_next.argtypes = [ctypes.c_void_p]
        L.augx_rand_destroy.argtypes = [ctypes.c_void_p]
        L.augx_batch_sample.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.augx_decode_sampled.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]
        L.augx_format_gff_sampled.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64,
                                              ctypes.POINTER(ctypes.c_int)]
        L.augx_path_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise AugxError(rc, lib().augx_last_error().decode(errors="replace"))


class Model:
    """Species model = option store + flat ln tables (``augx_model_load``)."""

    def __init__(self, config_path, species, **opts):
        L = lib()
        self._h = ctypes.c_void_p()
        names = (ctypes.c_char_p * len(opts))(*[k.encode() for k in opts])
        vals = (ctypes.c_char_p * len(opts))(*[str(v).encode() for v in opts.values()])
        _check(L.augx_model_load(config_path.encode(), species.encode(), len(opts), names, vals, ctypes.byref(self._h)))
        self.species = species
        self.config_path = config_path

    @property
    def tables_ptr(self):
        return ctypes.c_void_p(lib().augx_model_tables(self._h))

    @property
    def n_states(self):
        return ctypes.cast(self.tables_ptr, ctypes.POINTER(ctypes.c_int32))[0]

    def option(self, name):
        v = lib().augx_model_option(self._h, name.encode())
        return None if v is None else v.decode()

    def close(self):
        if self._h:
            lib().augx_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        # (objects that live until the interpreter shuts down -- e.g. held by the traceback of a failed test -- are left to the
        #  operating system: by then the HIP runtime may have run its own exit handlers, and calling into it crashes)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class DecodedPiece:
    __slots__ = ("status", "ln_viterbi", "states")

    def __init__(self, status, lnv, states):
        self.status, self.ln_viterbi, self.states = status, lnv, states  # states: [(begin, end, state, type)]


class Batch:
    """A batch of pieces resident in HBM (``augx_batch_create``); decode() may be repeated and timed."""

    def __init__(self, decoder, seqs, init_kind=0, term_kind=0):
        L = lib()
        self.decoder = decoder
        self.n = len(seqs)
        self._keep = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        self.lens = [len(s) for s in self._keep]
        P = (_Piece * self.n)()
        iks = init_kind if isinstance(init_kind, (list, tuple)) else [init_kind] * self.n
        tks = term_kind if isinstance(term_kind, (list, tuple)) else [term_kind] * self.n
        for i, s in enumerate(self._keep):
            P[i].seq, P[i].len, P[i].init_kind, P[i].term_kind = s, len(s), iks[i], tks[i]
        self._h = ctypes.c_void_p()
        _check(L.augx_batch_create(decoder._h, P, self.n, ctypes.byref(self._h)))
        decoder._batches.add(self)   # (a decoder closes its live batches before it goes: they hold a pointer to it)

    def decode(self, sync=True):
        _check(lib().augx_batch_decode(self.decoder._h, self._h))
        if sync:
            _check(lib().augx_batch_sync(self.decoder._h))

    def kernel_ms(self):
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        _check(lib().augx_batch_kernel_ms(self.decoder._h, self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"prep_ms": a.value, "trellis_ms": b.value, "backtrace_ms": c.value}

    def paths(self):
        L = lib()
        out = (_Path * self.n)()
        _check(L.augx_batch_paths(self.decoder._h, self._h, out))
        res = []
        for i in range(self.n):
            st = [(out[i].states[k].begin, out[i].states[k].end, out[i].states[k].state, out[i].states[k].type)
                  for k in range(out[i].n_states)]
            res.append(DecodedPiece(out[i].status, out[i].ln_viterbi, st))
            L.augx_path_free(ctypes.byref(out[i]))
        return res

    def cells(self, piece):
        import numpy as np
        S = self.decoder.model.n_states
        V = np.empty((self.lens[piece], S), dtype=np.float64)
        _check(lib().augx_batch_cells(self.decoder._h, self._h, piece, V.ctypes.data_as(ctypes.c_void_p)))
        return V

    def forward(self):
        """run the forward algorithm on the decoded batch (``augx_batch_forward``)"""
        _check(lib().augx_batch_forward(self.decoder._h, self._h))

    def forward_cells(self, piece):
        """(ln F matrix [len, S], ln P(sequence)) of one piece"""
        import numpy as np
        S = self.decoder.model.n_states
        F = np.empty((self.lens[piece], S), dtype=np.float64)
        lnp = ctypes.c_double()
        _check(lib().augx_batch_forward_cells(self.decoder._h, self._h, piece, F.ctypes.data_as(ctypes.c_void_p), ctypes.byref(lnp)))
        return F, lnp.value

    def sample(self, piece, n, rand):
        """n state paths of one piece sampled from the forward matrix (``augx_batch_sample``; run forward() first); the draws
        come from ``rand`` (a Rand: glibc's rand() restated, one stream over a run).  Returns [[(begin, end, state, type)]]"""
        L = lib()
        out = (_Path * max(1, n))()
        _check(L.augx_batch_sample(self.decoder._h, self._h, piece, n, rand._h, out))
        res = []
        for i in range(n):
            if out[i].status != 0:
                raise AugxError(out[i].status, "sampling failed")
            res.append([(out[i].states[k].begin, out[i].states[k].end, out[i].states[k].state, out[i].states[k].type)
                        for k in range(out[i].n_states)])
            L.augx_path_free(ctypes.byref(out[i]))
        return res

    def close(self):
        if self._h:
            lib().augx_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        # (objects that live until the interpreter shuts down -- e.g. held by the traceback of a failed test -- are left to the
        #  operating system: by then the HIP runtime may have run its own exit handlers, and calling into it crashes)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Rand:
    """glibc's rand() after srand(seed), restated (``augx_rand``): the generator the reference samples with (never seeded: 1)."""

    def __init__(self, seed=1):
        self._h = ctypes.c_void_p(lib().augx_rand_create(seed))

    def next(self):
        return lib().augx_rand_next(self._h)

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            if self._h:
                lib().augx_rand_destroy(self._h)
                self._h = None
        except Exception:
            pass


def tss0(model, seq):
    """``augx_tss0``: (forward, reverse) ln value of the TSS window that begins at base 0 of `seq` -- what a later sequence of the same
    length is decoded with by the reference (include/augx.h)"""
    L = lib()
    L.augx_tss0.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    b = seq if isinstance(seq, bytes) else seq.encode()
    out = (ctypes.c_double * 2)()
    _check(L.augx_tss0(model._h, b, len(b), out))
    return (out[0], out[1])


def decode_sampled(decoders, seqs, n_samples, rand, init_kind=0, term_kind=0):
    """``augx_decode_sampled``: Viterbi path + n_samples sampled paths per piece, batches in input order over the decoders.
    Returns [(DecodedPiece, [sampled paths])]."""
    L = lib()
    n = len(seqs)
    keep = [s if isinstance(s, bytes) else s.encode() for s in seqs]
    P = (_Piece * n)()
    iks = init_kind if isinstance(init_kind, (list, tuple)) else [init_kind] * n
    tks = term_kind if isinstance(term_kind, (list, tuple)) else [term_kind] * n
    for i, s in enumerate(keep):
        P[i].seq, P[i].len, P[i].init_kind, P[i].term_kind = s, len(s), iks[i], tks[i]
    D = (ctypes.c_void_p * len(decoders))(*[d._h for d in decoders])
    out = (_Path * n)()
    smp = (_Path * max(1, n * n_samples))()
    _check(L.augx_decode_sampled(D, len(decoders), P, n, n_samples, rand._h, out, smp))
    res = []
    for i in range(n):
        st = [(out[i].states[k].begin, out[i].states[k].end, out[i].states[k].state, out[i].states[k].type) for k in range(out[i].n_states)]
        sm = []
        for q in range(n_samples):
            sp = smp[i * n_samples + q]
            sm.append([(sp.states[k].begin, sp.states[k].end, sp.states[k].state, sp.states[k].type) for k in range(sp.n_states)])
            L.augx_path_free(ctypes.byref(sp))
        res.append((DecodedPiece(out[i].status, out[i].ln_viterbi, st), sm))
        L.augx_path_free(ctypes.byref(out[i]))
    return res


class _Cut(ctypes.Structure):
    _fields_ = [("record", ctypes.c_int32), ("status", ctypes.c_int32), ("begin", ctypes.c_int64), ("end", ctypes.c_int64),
                ("init_kind", ctypes.c_int32), ("term_kind", ctypes.c_int32)]


class _CutStats(ctypes.Structure):
    _fields_ = [("scout_tiles", ctypes.c_int32), ("batches", ctypes.c_int32), ("windows_decoded", ctypes.c_int32), ("windows_used", ctypes.c_int32)]


DECODE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(_Piece), ctypes.c_int, ctypes.POINTER(_Path))


def find_cuts(model, seqs, decode_fn, scout=-1):
    """``augx_find_cuts``: the pieces [(record, status, begin, end, init_kind, term_kind)] the records are cut into, the exam windows
    decoded through ``decode_fn(user, pieces, n, out_paths) -> rc`` (a DECODE_FN); also returns the statistics of the run."""
    L = lib()
    L.augx_find_cuts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64), DECODE_FN, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.POINTER(ctypes.POINTER(_Cut)), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_CutStats)]
    L.augx_cuts_free.argtypes = [ctypes.POINTER(_Cut)]
    keep = [s.encode() if isinstance(s, str) else s for s in seqs]
    n = len(keep)
    c_seqs = (ctypes.c_char_p * n)(*keep)
    c_lens = (ctypes.c_int64 * n)(*[len(s) for s in keep])
    out = ctypes.POINTER(_Cut)()
    n_out = ctypes.c_int()
    st = _CutStats()
    _check(L.augx_find_cuts(model._h, n, c_seqs, c_lens, decode_fn, None, scout, ctypes.byref(out), ctypes.byref(n_out), ctypes.byref(st)))
    res = [(out[i].record, out[i].status, out[i].begin, out[i].end, out[i].init_kind, out[i].term_kind) for i in range(n_out.value)]
    L.augx_cuts_free(out)
    return res, {"scout_tiles": st.scout_tiles, "batches": st.batches, "windows_decoded": st.windows_decoded, "windows_used": st.windows_used}


def device_count():
    return lib().augx_device_count()


def partition_lpt(lens, n_bins):
    """Longest-first bin packing used by decode_sharded (host only)."""
    import numpy as np
    a = np.asarray(lens, dtype=np.int64)
    out = np.zeros(len(a), dtype=np.int32)
    _check(lib().augx_partition_lpt(a.ctypes.data_as(ctypes.c_void_p), len(a), n_bins, out.ctypes.data_as(ctypes.c_void_p)))
    return out.tolist()


def decode_sharded(decoders, seqs, init_kind=0, term_kind=0):
    """``augx_decode_sharded``: pieces fanned over several decoders (one per GPU), results in input order."""
    L = lib()
    n = len(seqs)
    keep = [s if isinstance(s, bytes) else s.encode() for s in seqs]
    P = (_Piece * n)()
    iks = init_kind if isinstance(init_kind, (list, tuple)) else [init_kind] * n
    tks = term_kind if isinstance(term_kind, (list, tuple)) else [term_kind] * n
    for i, s in enumerate(keep):
        P[i].seq, P[i].len, P[i].init_kind, P[i].term_kind = s, len(s), iks[i], tks[i]
    D = (ctypes.c_void_p * len(decoders))(*[d._h for d in decoders])
    out = (_Path * n)()
    _check(L.augx_decode_sharded(D, len(decoders), P, n, out))
    res = []
    for i in range(n):
        st = [(out[i].states[k].begin, out[i].states[k].end, out[i].states[k].state, out[i].states[k].type)
              for k in range(out[i].n_states)]
        res.append(DecodedPiece(out[i].status, out[i].ln_viterbi, st))
        L.augx_path_free(ctypes.byref(out[i]))
    return res


class Decoder:
    """Device context (``augx_decoder_create``).  Raises AugxError(AUGX_E_NODEVICE) without a HIP device."""

    def __init__(self, model, device=0):
        import weakref
        self.model = model
        self._batches = weakref.WeakSet()
        self._h = ctypes.c_void_p()
        _check(lib().augx_decoder_create(model._h, device, ctypes.byref(self._h)))

    def set_exact(self, on=True):
        """``augx_decoder_set_exact``: replay the reference's snippet cache on pieces with several GC classes for the Viterbi run, too"""
        _check(lib().augx_decoder_set_exact(self._h, 1 if on else 0))

    def count_near_ties(self, on=True):
        """``augx_decoder_count_near_ties``: batches created from now on count the near ties on their chosen paths"""
        _check(lib().augx_decoder_count_near_ties(self._h, 1 if on else 0))

    def near_ties(self):
        """``augx_decoder_near_ties``: (cells, pieces) summed over the paths fetched so far"""
        L = lib()
        L.augx_decoder_near_ties.restype = ctypes.c_int64
        L.augx_decoder_near_ties.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        np_ = ctypes.c_int64()
        return int(L.augx_decoder_near_ties(self._h, ctypes.byref(np_))), int(np_.value)

    def set_share(self, n):
        """n decoders (streams) work on this device at the same time: plan the trellis segments for 1/n of its compute units"""
        _check(lib().augx_decoder_set_share(self._h, n))

    def decode(self, seqs, init_kind=0, term_kind=0):
        b = Batch(self, seqs, init_kind, term_kind)
        try:
            b.decode()
            return b.paths()
        finally:
            b.close()

    def close(self):
        if self._h:
            for b in list(self._batches):
                b.close()
            lib().augx_decoder_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        # (objects that live until the interpreter shuts down -- e.g. held by the traceback of a failed test -- are left to the
        #  operating system: by then the HIP runtime may have run its own exit handlers, and calling into it crashes)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass
