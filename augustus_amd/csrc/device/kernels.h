// kernels.h -- bodies of the decode kernels, written once for gfx950 (hipcc) and for the lane-loop emulator
// (-DAUGX_EMU, tests only).  See DESIGN.md for the kernel decomposition:
//   K1  prep   : encode, site/stop/base-count prefix scans, GC class, fixed-point content prefix sums, signals
//   K2a candidates: transition * emission of every candidate of the variable-length states, one wavefront per tile of 64 bases
//   K2b trellis: one workgroup per piece, position-sequential, V columns in an LDS ring
//   K3  back   : back-pointer chase, one wavefront per piece
//
// Lane discipline: a FOR_LANES block is executed by every lane (concurrently on the device, one after the other
// in the emulator); lanes communicate only through LDS/global memory BETWEEN blocks, separated by WAVE_SYNC().
#pragma once
#include <type_traits>
#include "dp.h"

namespace augx {
namespace dev {

#ifdef AUGX_EMU
#define FOR_LANES(l) for (int l = 0; l < WAVE; ++l)
#define LV(T, name) T name[WAVE]
#define LV2(T, name, K) T name[K][WAVE]
#define LI l
#define LX(name) name[l]
#define WAVE_SYNC() ((void)0)
#define GLOBAL_SYNC() ((void)0)
#define AUGX_KFN inline
// trellis kernel: NWAVES wavefronts per workgroup.  FOR_THREADS runs every thread of the workgroup, FOR_WAVES /
// FOR_WLANES run "for each wavefront: its 64 lanes" (on the device each wavefront executes its own iteration only).
#define FOR_THREADS(t) for (int t = 0; t < NT; ++t)
#define FOR_WAVES(w) for (int w = 0; w < NWAVES; ++w)
#define FOR_WLANES(t, w) for (int t = (w) * WAVE; t < (w) * WAVE + WAVE; ++t)
#define TV(T, name) T name[NT]
#define TV2(T, name, K) T name[K][NT]
#define TI t
#define TX(name) name[t]
#define BLOCK_SYNC() ((void)0)
#define BLOCK_GLOBAL_SYNC() ((void)0)
#else
#define FOR_LANES(l) for (int l = (int)threadIdx.x, _once = 1; _once; _once = 0)
#define LV(T, name) T name[1]
#define LV2(T, name, K) T name[K][1]
#define LI 0
#define LX(name) name[0]
// One wavefront per workgroup: LDS instructions of a wave execute in issue order, so cross-lane LDS hand-offs only
// need the compiler not to reorder (and the LDS queue drained); no s_barrier and no wait for global stores.
#define WAVE_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __asm__ volatile("" ::: "memory"); } while (0)
// before re-reading global data this wave stored earlier (candidate lists, igenic column): drain the store queue
#define GLOBAL_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#define AUGX_KFN __device__ __forceinline__
#define FOR_THREADS(t) for (int t = (int)threadIdx.x, _once = 1; _once; _once = 0)
constexpr int RW = -1; // (shadowed by trellisPiece's template parameter where the wavefront's index is a compile-time constant: one instantiation per role)
#define FOR_WAVES(w) for (int w = (RW >= 0 ? RW : (int)(threadIdx.x >> 6)), _oncew = 1; _oncew; _oncew = 0)
#define FOR_WLANES(t, w) for (int t = (int)threadIdx.x, _once = 1; _once; _once = 0)
#define TV(T, name) T name[1]
#define TV2(T, name, K) T name[K][1]
#define TI 0
#define TX(name) name[0]
// workgroup barrier that only orders LDS traffic (no wait for outstanding global stores)
#define BLOCK_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); __asm__ volatile("" ::: "memory"); } while (0)
// workgroup barrier after which everything the workgroup stored to HBM is visible to its own later loads: producer and
// consumer share the CU's vector cache and L2, so waiting for the stores is enough (no L2 write-back)
#define BLOCK_GLOBAL_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0x0070); __syncthreads(); } while (0)
#endif
constexpr int NWAVES = 8, NT = NWAVES * WAVE;

// pointers loaded from the BatchView are generic to the compiler; the hot loops re-type them as global so that their
// loads and stores do not count against the LDS wait counter (flat instructions do)
#ifdef AUGX_EMU
template <class T> inline T *gp(T *p) { return p; }
template <class T> inline T *lp(T *p) { return p; }
inline double ldsLoadD(const double *p) { return *p; }
inline Item ldItem(const Item *p) { return *p; }
inline IntronStart ldIntronStart(const IntronStart *p) { return *p; }
#else
#define AUGX_GLOBAL __attribute__((address_space(1)))
// a pointer INTO the workgroup's LDS that the compiler only knows as generic (selected among several LDS arrays, or accessed
// as volatile): re-typed, so that the access is a ds_ instruction and not a flat one (slower, and counted against both
// the LDS and the memory wait counters)
#define AUGX_LDS __attribute__((address_space(3)))
__device__ __forceinline__ double ldsLoadD(const double *p) { return *(const AUGX_LDS double *)p; }
template <class T> __device__ __forceinline__ AUGX_GLOBAL T *gp(T *p) { return (AUGX_GLOBAL T *)p; }
template <class T> __device__ __forceinline__ AUGX_LDS T *lp(T *p) { return (AUGX_LDS T *)p; } // (a pointer into the LDS, re-typed)
__device__ __forceinline__ Item ldItem(const Item *p) { // one 16-byte global load
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = *(const AUGX_GLOBAL v4i *)p;
    Item I;
    I.te = __hiloint2double(r.y, r.x); I.kp = (uint32_t)r.z; I.src = (uint32_t)r.w;
    return I;
}
__device__ __forceinline__ IntronStart ldIntronStart(const IntronStart *p) { // one 16-byte global load
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = *(const AUGX_GLOBAL v4i *)p;
    IntronStart e;
    e.pos = r.x; e.ctx = (uint32_t)r.y; e.fx = ((uint64_t)(uint32_t)r.w << 32) | (uint32_t)r.z;
    return e;
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dppMov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xf, false); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dppMovD(double old, double x) {
    int lo = dppMov<CTRL, ROWMASK>(__double2loint(old), __double2loint(x)), hi = dppMov<CTRL, ROWMASK>(__double2hiint(old), __double2hiint(x));
    return __hiloint2double(hi, lo);
}
#endif

// ------------------------------------------------------------------------------------------------
// wave-wide argmax of (value, key): larger value wins, ties go to the larger key (= the candidate the
// reference's descending loop with strict '>' meets first: src/exonmodel.cc:1059,1115, src/intronmodel.cc:589,623)
// ------------------------------------------------------------------------------------------------
struct Best { double v; int key; int aux; };
AUGX_HD bool better(double v, int key, double bv, int bkey) { return v > bv || (v == bv && v > AUGX_NINF && key > bkey); }

#ifdef AUGX_EMU
inline Best waveArgMax(const double *v, const int *key, const int *aux) {
    Best b{AUGX_NINF, -2147483647, -1};
    for (int l = 0; l < WAVE; l++)
        if (better(v[l], key[l], b.v, b.key)) { b.v = v[l]; b.key = key[l]; b.aux = aux[l]; }
    return b;
}
#else
__device__ inline Best waveArgMax(const double *v, const int *key, const int *aux) {
    double bv = v[0];
    int bk = key[0], ba = aux[0];
    if (!(bv > AUGX_NINF)) { bk = -2147483647; ba = -1; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double ov = __shfl_xor(bv, o, 64);
        int ok = __shfl_xor(bk, o, 64);
        int oa = __shfl_xor(ba, o, 64);
        if (better(ov, ok, bv, bk)) { bv = ov; bk = ok; ba = oa; }
    }
    return Best{bv, bk, ba};
}
#endif

AUGX_HD Piece makePiece(const DevTables &T, const BatchView &B, int p) {
    Piece P;
    int64_t o = B.off[p];
    P.t = &T;
    P.n = B.len[p];
    P.c = B.cls[p];
    P.o = o;
    P.code = B.code + o + 1;
    P.fx = B.fx;
    P.nsm = B.nsm;
    P.sig = B.sig + (o + 1) * NSIG;
    return P;
}
// the same piece seen through plane pl: tables of that plane's GC class, content prefix sums of that plane
AUGX_HD Piece makePieceAt(const DevTables &T, const BatchView &B, int p, int pl) {
    Piece P = makePiece(T, B, p);
    if (pl > 0) {
        P.c = B.planeCls[p * MAXPL + pl];
        P.fx = B.fx + (int64_t)pl * B.N * NFX;
    }
    return P;
}

// =================================================================================================
// K1  prep kernels (one thread per slot unless noted).  g = global slot index.
// =================================================================================================
constexpr int NCNT = 11; // prefix-count fields: a c g t | atg | ag(LA) | ac(LR) | gt(LD) | ct(RD) | reverse stop codons | soft-masked
constexpr int CNT_ATG = 4, CNT_LA = 5, CNT_LR = 6, CNT_LD = 7, CNT_RD = 8, CNT_RS = 9, CNT_SOFT = 10;

AUGX_HD void k1Encode(const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t q = g - B.off[p] - 1;
    uint8_t c = 4;
    if (q >= 0 && q < B.len[p]) {
        char ch = B.raw[g];
        ch = (ch >= 'A' && ch <= 'Z') ? (char)(ch - 'A' + 'a') : ch; // reference lower-cases, src/extrinsicinfo.cc:1726
        c = ch == 'a' ? 0 : ch == 'c' ? 1 : ch == 'g' ? 2 : ch == 't' ? 3 : 4;
    }
    B.code[g] = c;
}

// site flags and stop codons -> terms of the count / max scans (in registers: the device scans recompute them in both of
// their passes instead of reading them back from HBM, decoder.hip kSiteScan*)
AUGX_HD void k1SiteTermsCalc(const DevTables &T, const BatchView &B, int64_t g, uint64_t cnt[NCNT], uint64_t ns[6],
                             const uint8_t *lcode = nullptr, int lLo = 0, int lHi = 0) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    Piece P;
    P.t = &T; P.n = B.len[p]; P.c = 0; P.o = o; P.code = B.code + o + 1; P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;
    for (int i = 0; i < NCNT; i++) cnt[i] = 0;
    for (int i = 0; i < 6; i++) ns[i] = 0;
    if (q >= 0 && q < P.n) {
    int c = P.b(q);
    cnt[0] = c == 0; cnt[1] = c == 1; cnt[2] = c == 2; cnt[3] = c == 3; // (no dynamic index: the arrays stay in registers)
    if (T.soft && B.raw[g] >= 'a' && B.raw[g] <= 'z') cnt[CNT_SOFT] = 1; // lower-case base = nonexonpart hint (src/extrinsicinfo.cc:1703-1720)
    // start codon with positive probability at q (a of atg)
    if (q < P.n - 2) { int pn = P.pat(q, 3); if (pn >= 0 && T.ln_startcodon[pn] > AUGX_NINF) cnt[CNT_ATG] = 1; }
    // (base 0 is also listed when an exon may begin right after it without a splice site: the reference asks for the
    //  site only if the biological exon begins at base >= 2 (src/exonmodel.cc:1468, 1522), and such a candidate then
    //  continues from column 0 -- the initial probabilities of the longass / rlongdss states)
    if (P.possASS(q - T.Ae) || (q == 0 && T.Ae <= 1)) cnt[CNT_LA] = 1;   // longass may end at q   (src/intronmodel.cc:705)
    if (P.possRDSS(q - T.Ds) || (q == 0 && T.Ds <= 1)) cnt[CNT_LR] = 1;  // rlongdss may end at q  (:709)
    if (P.possDSS(q - T.De - 2 + 1)) cnt[CNT_LD] = 1;               // longdss may end at q   (:693)
    if (P.possRASS(q - T.U - T.As - 2 + 1)) cnt[CNT_RD] = 1;        // rlongass may end at q  (:713)
    if (q <= P.n - 3) {
        const int r = q % 3;
        const uint64_t fs = P.isStop(q) ? (uint64_t)q + 1 : 0, rs = P.isRCStop(q) ? (uint64_t)q + 1 : 0;
        ns[0] = r == 0 ? fs : 0; ns[1] = r == 1 ? fs : 0; ns[2] = r == 2 ? fs : 0;
        ns[3] = r == 0 ? rs : 0; ns[4] = r == 1 ? rs : 0; ns[5] = r == 2 ? rs : 0;
        if (rs) cnt[CNT_RS] = 1;
    }
    }
}
AUGX_HD void k1SiteTerms(const DevTables &T, const BatchView &B, int64_t g) {
    uint64_t cnt[NCNT], ns[6];
    k1SiteTermsCalc(T, B, g, cnt, ns);
    for (int i = 0; i < NCNT; i++) B.cnt[fidx(g, i, NCNT)] = (uint32_t)cnt[i];
    for (int i = 0; i < 6; i++) B.nsm[fidx(g, i, 6)] = (uint32_t)ns[i];
}

// entries of the longest candidate list of piece p (site counts at its last base): sizes the list arrays
AUGX_HD void k1ListCount(const BatchView &B, int p) {
    const int64_t g = B.off[p] + B.len[p];
    uint64_t m = 0;
    for (int f = CNT_ATG; f <= CNT_RS; f++) { const uint64_t c = B.cnt[fidx(g, f, NCNT)]; m = c > m ? c : m; }
    B.listCnt[p] = (int32_t)m;
}

// GC class of the window starting at base s (one thread per slot; reference ContentStairs::computeStairs,
// src/motif.cc:543-616; the set of window classes decides whether the piece is single-class)
AUGX_HD int nearestClass(const DevTables &T, const double cnt[4]) {
    double r[4] = {0.25, 0.25, 0.25, 0.25};
    double sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    if (sum > 0.0)
        for (int i = 0; i < 4; i++) r[i] = cnt[i] / sum;
    double maxW = -1;
    int ret = -1;
    for (int c = 0; c < T.C; c++) {
        double w = 1;
        if (T.gc_weighing_type == 3) {
            double z[4], tmp[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4; i++) z[i] = r[i] - T.gc_zus[c][i];
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) tmp[j] += z[i] * T.gc_weight_matrix[i * 4 + j];
            double q = 0;
            for (int i = 0; i < 4; i++) q += tmp[i] * z[i];
            w = 1 + 9 * exp(-q);
        } else if (T.gc_weighing_type == 2) {
            double g1 = r[1] + r[2], g2 = T.gc_zus[c][1] + T.gc_zus[c][2];
            int c1 = g1 < .43 ? 0 : g1 < .51 ? 1 : g1 < .57 ? 2 : 3, c2 = g2 < .43 ? 0 : g2 < .51 ? 1 : g2 < .57 ? 2 : 3;
            w = c1 == c2 ? 1 : 0;
        }
        if (w > maxW) { maxW = w; ret = c; }
    }
    return ret;
}
// returns the class of window start s of piece p, or -1 if s is not a window start
AUGX_HD int k1WindowClass(const DevTables &T, const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int s = (int)(g - o - 1), n = B.len[p];
    int win = T.gc_win;
    if (win > n || win < 1) win = n;
    if (s < 0 || s > n - win) return -1;
    if (T.C == 1) { B.gcRaw[g] = 0; return 0; } // (a single class: nothing to decide)
    double cnt[4];
    for (int i = 0; i < 4; i++) cnt[i] = (double)(B.cnt[fidx(o + s + win, i, NCNT)] - B.cnt[fidx(o + s, i, NCNT)]);
    // a window without a single nucleotide (inside a run of N at least as long as the window): the reference's BaseCount keeps the
    // relative frequencies it was constructed with -- those of the FIRST window of the piece (normalize() leaves them alone when the
    // counts sum to 0, src/motif.cc:204-212, and computeStairs never refreshes them, :561-575); 1/4 each if that window is empty too
    if (cnt[0] + cnt[1] + cnt[2] + cnt[3] == 0.0)
        for (int i = 0; i < 4; i++) cnt[i] = (double)(B.cnt[fidx(o + win, i, NCNT)] - B.cnt[fidx(o, i, NCNT)]);
    const int c = nearestClass(T, cnt);
    B.gcRaw[g] = (uint8_t)c; // input of the content stairs, should the windows of the piece disagree (layout.h: stairsPlanes)
    return c;
}

// fixed-point terms of the 20 content prefix fields (false: the piece has no plane pl)
AUGX_HD bool k1FxTermsCalc(const DevTables &T, const BatchView &B, int64_t g, int pl, uint64_t out[NFX],
                           const uint8_t *lcode = nullptr, int lLo = 0, int lHi = 0) {
    int p = B.chunkPiece[g / CHUNK];
    if (pl > 0 && pl >= B.nPlanes[p]) return false; // (this piece has no such plane)
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    for (int i = 0; i < NFX; i++) out[i] = 0;
    Piece P;
    P.t = &T; P.n = B.len[p]; P.c = B.cls[p] < 0 ? -1 : B.planeCls[p * MAXPL + pl]; P.o = o; P.code = B.code + o + 1; P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;
    if (q >= 0 && q < B.len[p] && P.c >= 0) {
    const int k = T.k, NP = T.NP, c = P.c;
    int pn = q >= k ? P.pat(q - k, k + 1) : -1;
    int rn = P.rcpat(q, k + 1);
    const double *tabs[3] = {T.ex_emi + (int64_t)c * 3 * NP, T.ex_init + (int64_t)c * 3 * NP, T.ex_et + (int64_t)c * 3 * NP};
    for (int a = 0; a < 3; a++)
        for (int tb = 0; tb < 3; tb++) {
            // forward strand: frame (q + a) mod 3; reverse strand: frame (a - q) mod 3
            // (reference ExonModel::seqProb, src/exonmodel.cc:1957-1966)
            out[(0 * 3 + a) * 3 + tb] = toFx(pn >= 0 ? tabs[tb][mod3(q + a) * NP + pn] : T.ln_n_coding);
            out[(1 * 3 + a) * 3 + tb] = toFx(rn >= 0 ? tabs[tb][mod3(a - q) * NP + rn] : T.ln_n_coding);
        }
    const double *inE = T.in_emi + (int64_t)c * T.NPin;
    // (intron content; a soft-masked base adds the nonexonpart bonus, reference src/intronmodel.cc:1011-1036)
    const double softB = (T.soft && B.raw[g] >= 'a' && B.raw[g] <= 'z') ? T.lnSoft : 0.0;
    const int ki = T.kIn; // (the intron model's own order: the exon patterns serve where it is k)
    const int pni = ki == k ? pn : (q >= ki ? P.pat(q - ki, ki + 1) : -1);
    out[FX_INF] = toFx((pni >= 0 ? inE[pni] : T.ln_quarter) + softB);
    int rn2 = (q + ki < P.n) ? (ki == k ? rn : P.rcpat(q, ki + 1)) : -1;
    out[FX_INR] = toFx((rn2 >= 0 ? inE[rn2] : T.ln_quarter) + softB);
    }
    return true;
}
AUGX_HD void k1FxTerms(const DevTables &T, const BatchView &B, int64_t g, int pl) {
    uint64_t out[NFX];
    if (!k1FxTermsCalc(T, B, g, pl, out)) return;
    uint64_t *fx = B.fx + (int64_t)pl * B.N * NFX;
    for (int i = 0; i < NFX; i++) fx[fidx(g, i, NFX)] = out[i];
}

// per-base signal record + end-gate mask of the variable-length states
constexpr int NSITE = 4;
AUGX_HD void k1Signals(const DevTables &T, const BatchView &B, int64_t g, const uint8_t *lcode = nullptr, int lLo = 0, int lHi = 0) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    // (everything is computed into registers first and stored in one burst at the end: a store to the records between two table
    //  look-ups keeps the compiler from having the look-ups in flight together -- it cannot know that the tables and the records
    //  do not overlap -- and this kernel waits for its loads 78 % of the time, profiles/r05_sq.txt)
    double e[NSIG];
    for (int i = 0; i < NSIG; i++) e[i] = AUGX_NINF;
    int32_t stv[NSITE];
    for (int i = 0; i < NSITE; i++) stv[i] = -1;
    uint64_t gate = 0;
    int64_t atgIdx = -1;
    const bool inPiece = !(q < 0 || q >= B.len[p] || B.cls[p] < 0);
    int64_t lo = 0;
    if (inPiece) {
    Piece P = makePieceAt(T, B, p, B.gcPlane[g]); // everything ending at q is scored with the class of q
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;    // (device: the bases around the workgroup's slots, staged in LDS)
    const double softB = (T.soft && B.raw[g] >= 'a' && B.raw[g] <= 'z') ? T.lnSoft : 0.0; // (src/igenicmodel.cc:306-326)
    e[SIG_EIG] = q >= 1 ? eIg(P, q) + softB : AUGX_NINF;
    e[SIG_EIN] = eIn(P, q) + softB;
    if (T.utr && T.uk != T.k) e[SIG_EUIN] = eUin(P, q) + softB; // (dense.h: the UTR intron chain states read this one)
    // fixed-length intron states ending at q: gate && emission (reference src/intronmodel.cc:690-717,861-923)
    // (the splice-site records SIG_DSSF/DSSR/ASSF/ASSR are filled by k1SiteSignals, one thread per site instead of one
    //  per base: their motif loops would otherwise run with one lane in sixteen active.  SIG_TISF is not used: the
    //  start-codon records carry the translation-initiation term, see k1SiteConsts)
    e[SIG_TISR] = tisRev(P, q);
    e[SIG_STOPF] = exEndPart(P, AUGX_K_TERMINAL, 0, q, AUGX_NINF); // ln P(stop codon ending at q), -inf if none
    // list index of the site ending at q (prefix count - 1), -1 if q is not such a site
    uint64_t cn[NCNT], cp[NCNT];
    for (int i = CNT_ATG; i < NCNT; i++) { cn[i] = B.cnt[fidx(g, i, NCNT)]; cp[i] = B.cnt[fidx(g - 1, i, NCNT)]; }
    lo = listOff(B, p);
    for (int i = 0; i < NSITE; i++)
        stv[i] = cn[CNT_LA + i] != cp[CNT_LA + i] ? (int32_t)cn[CNT_LA + i] - 1 : -1;
    if (cn[CNT_ATG] != cp[CNT_ATG]) atgIdx = (int64_t)cn[CNT_ATG] - 1;
    // emission of the equalD states ending at q (reference IntronModel::seqProb, src/intronmodel.cc:1087-1107)
    if (q - T.dStateLen >= 0) e[SIG_EQD] = P.seg(FX_INF, q - T.dStateLen + 1, q);
    // end gates.  They depend on the state's kind group and (forward splice end) frame only: each is evaluated once
    if (q >= 1) {
        const bool stopOpen = e[SIG_STOPF] > AUGX_NINF && q - 3 >= 0;              // single, terminal: right = q - 3
        const bool tisOpen = e[SIG_TISR] > AUGX_NINF && q - T.W - 3 >= 0;           // rsingle, rinitial: right = q - W - 3
        bool fwdOpen[3];                                                             // initial, internal per frame
        for (int w2 = 0; w2 < 3; w2++) fwdOpen[w2] = exEndPart(P, AUGX_K_INTERNAL, w2, q, AUGX_NINF) > AUGX_NINF;
        const bool revOpen = exEndPart(P, AUGX_K_RINTERNAL, 0, q, AUGX_NINF) > AUGX_NINF; // rinternal, rterminal (no frame in the gate)
        const bool lessOpen = lessDGate(P, true, q), rlessOpen = lessDGate(P, false, q);
        for (int s = 0; s < T.S; s++) {
            if (!T.reachable[s]) continue;
            const int kind = T.kind[s], w2 = T.win[s];
            bool open = false;
            switch (kind) {
            case AUGX_K_SINGLE: case AUGX_K_TERMINAL: open = stopOpen; break;
            case AUGX_K_RSINGLE: case AUGX_K_RINITIAL: open = tisOpen; break;
            case AUGX_K_INITIAL: case AUGX_K_INTERNAL: open = w2 == 0 ? fwdOpen[0] : w2 == 1 ? fwdOpen[1] : fwdOpen[2]; break;
            case AUGX_K_RINTERNAL: case AUGX_K_RTERMINAL: open = revOpen; break;
            case AUGX_K_LESSD: open = lessOpen; break;
            case AUGX_K_RLESSD: open = rlessOpen; break;
            default: break;
            }
            if (open) gate |= 1ull << T.vbit[s];
        }
    }
    }
    // ---- the stores
    double *sg = B.sig + g * NSIG;
    for (int i = 0; i < NSIG; i++) sg[i] = e[i];
    B.gate[g] = gate;
    int32_t *st = B.site + g * NSITE;
    for (int i = 0; i < NSITE; i++) st[i] = stv[i];
    if (inPiece) {
        // the candidate lists are indexed by site; positions are known here, the trellis fills in the values
        if (stv[0] >= 0) B.laPos[lo + stv[0]] = q;
        if (stv[1] >= 0) B.lrPos[lo + stv[1]] = q;
        if (stv[2] >= 0) B.ldEnt[lo + stv[2]].pos = q;
        if (stv[3] >= 0) B.rdEnt[lo + stv[3]].pos = q;
        if (atgIdx >= 0) B.atgPos[lo + atgIdx] = q;
    }
}

// splice-site signal records: one thread per entry t of the four candidate lists (sel: 0 forward acceptor, 1 reverse
// donor, 2 forward donor, 3 reverse acceptor); reference IntronModel::aSSProb / dSSProb via emiProbUnderModel,
// src/intronmodel.cc:690-717,861-923.  Runs after k1Signals (which has reset the records and filled the positions).
AUGX_HD void k1SiteSignals(const DevTables &T, const BatchView &B, int64_t t, int sel) {
    if (t >= B.listCap) return;
    int p = 0; // the piece that owns entry t: the last one whose first entry is <= t
    for (int lo2 = 0, hi2 = B.nPieces - 1; lo2 <= hi2;) {
        const int mid = (lo2 + hi2) / 2;
        if (B.listOffs[mid] <= t) { p = mid; lo2 = mid + 1; } else hi2 = mid - 1;
    }
    if (B.cls[p] < 0) return;
    const int64_t o = B.off[p], lo = listOff(B, p), li = t - lo;
    const int n = B.len[p];
    if (li < 0 || li >= (int64_t)B.cnt[fidx(o + n, CNT_LA + sel, NCNT)]) return;
    const int q = sel == 0 ? B.laPos[lo + li] : sel == 1 ? B.lrPos[lo + li] : sel == 2 ? B.ldEnt[lo + li].pos : B.rdEnt[lo + li].pos;
    Piece P = makePieceAt(T, B, p, B.gcPlane[o + 1 + q]);
    const int dssWhole = T.Ds + 2 + T.De, assWhole = T.As + 2 + T.Ae;
    double *sg = B.sig + (o + 1 + q) * NSIG;
    // soft-masked bases inside the intronic part of the state's window [a, b] (reference src/intronmodel.cc:872-924,1011-1036)
    auto softIn = [&](int a, int b2) -> double {
        if (!T.soft) return 0.0;
        if (a < 0) a = 0;
        if (b2 < a) return 0.0;
        const uint64_t hi = B.cnt[fidx(o + 1 + b2, CNT_SOFT, NCNT)], lo2 = B.cnt[fidx(o + a, CNT_SOFT, NCNT)];
        return (double)(int64_t)(hi - lo2) * T.lnSoft;
    };
    if (sel == 2) { if (q - dssWhole >= 0) sg[SIG_DSSF] = dssProb(P, q - dssWhole + 1, true) + softIn(q - 2 - T.De + 1, q); }
    else if (sel == 1) { if (q - dssWhole >= 0) sg[SIG_DSSR] = dssProb(P, q - dssWhole + 1, false) + softIn(q - dssWhole + 1, q - T.Ds); }
    else if (sel == 0) { if (q - assWhole - T.U >= 0) sg[SIG_ASSF] = assProb(P, q - assWhole - T.U + 1, true) + softIn(q - assWhole - T.U + 1, q - T.Ae); }
    else { if (q - assWhole - T.U >= 0) sg[SIG_ASSR] = assProb(P, q - assWhole - T.U + 1, false) + softIn(q - assWhole - T.U + 1 + T.Ae, q); }
}

// candidate-side constants of the list entries (everything a candidate contributes that does not depend on Viterbi
// values).  Fast-path evaluation in the trellis combines them with end-side constants; the arithmetic is exactly that
// of exNotEndPart (same prefix differences, same order of additions).
// One set of constants per plane (GC class) of the piece -- the END of the candidate's state selects the plane: pl = plane.
AUGX_HD void k1SiteConsts(const DevTables &T, const BatchView &B, int64_t g, int pl, const uint8_t *lcode = nullptr, int lLo = 0, int lHi = 0) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    if (q < 0 || q >= B.len[p]) return;
    if (pl > 0 && (B.cls[p] < 0 || pl >= B.nPlanes[p])) return; // (this piece has no such plane)
    const int64_t pN = (int64_t)pl * B.N, pL = (int64_t)pl * B.listCap;
    double *pr = B.plsR + (pN + g) * 3;
    pr[0] = pr[1] = pr[2] = AUGX_NINF;
    if (B.cls[p] < 0) return;
    Piece P = makePieceAt(T, B, p, pl);
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;
    const int k = T.k, c = P.c, n = P.n;
    const int64_t lo = listOff(B, p);
    auto fxv = [&](int pos, int f) -> uint64_t { return pos < 0 ? 0 : P.fx[fidx(o + 1 + (pos < n ? pos : n - 1), f, NFX)]; };
    auto plsK = [&](int pn, int frame) { return pn >= 0 ? T.ex_pls[(((int64_t)c * (k + 1) + (k - 1)) * 3 + frame) * T.NP + pn] : k * T.ln_n_coding; };
    if (k > 0 && q - k + 1 >= 0) { // reverse strand initial pattern ending at q (src/exonmodel.cc:1598-1600)
        int pn = P.rcpat(q - k + 1, k);
        for (int fr = 0; fr < 3; fr++) pr[fr] = plsK(pn, fr);
    }
    uint64_t cn[NCNT], cp[NCNT];
    for (int i = CNT_ATG; i < NCNT; i++) { cn[i] = B.cnt[fidx(g, i, NCNT)]; cp[i] = B.cnt[fidx(g - 1, i, NCNT)]; }
    // (in every branch: all look-ups first, then the stores -- a store between two loads keeps them from being in flight together)
    if (cn[CNT_LA] != cp[CNT_LA]) { // forward acceptor candidate ending (as longass state) at q: exon inner part starts at bs = q+1
        int64_t idx = pL + lo + (int64_t)cn[CNT_LA] - 1;
        int bs = q + 1, eos = bs + k - 1, pn = P.pat(bs, k);
        double vP[3];
        uint64_t vF[3];
        for (int a = 0; a < 3; a++) { vP[a] = k == 0 ? 0.0 : plsK(pn, mod3(eos + a)); vF[a] = fxv(eos, (0 * 3 + a) * 3 + 0); }
        for (int a = 0; a < 3; a++) { B.laPls[idx * 3 + a] = vP[a]; B.laFx[idx * 3 + a] = vF[a]; }
    }
    if (cn[CNT_LR] != cp[CNT_LR]) { // reverse donor candidate
        int64_t idx = pL + lo + (int64_t)cn[CNT_LR] - 1;
        int bs = q + 1, eot = bs + T.Le - 1;
        double vE[3];
        uint64_t vF[3];
        for (int a = 0; a < 3; a++) {
            const int fb = (1 * 3 + a) * 3;
            vE[a] = eot < bs ? 0.0 : (double)(int64_t)(fxv(eot, fb + 2) - fxv(bs - 1, fb + 2)) * AUGX_FX_INV;
            vF[a] = fxv(eot, fb + 0);
        }
        for (int a = 0; a < 3; a++) { B.lrEt[idx * 3 + a] = vE[a]; B.lrFx[idx * 3 + a] = vF[a]; }
    }
    // short-intron starts: content prefix at q and the two bases before the biological intron (spliced-codon check)
    if (cn[CNT_LD] != cp[CNT_LD]) {
        IntronStart &e = B.ldEnt[pL + lo + (int64_t)cn[CNT_LD] - 1];
        const int bobi = q + 1 - T.De - 2;
        e.pos = q; e.fx = fxv(q, FX_INF); e.ctx = (uint32_t)P.b(bobi - 1) | ((uint32_t)P.b(bobi - 2) << 4);
    }
    if (cn[CNT_RD] != cp[CNT_RD]) {
        IntronStart &e = B.rdEnt[pL + lo + (int64_t)cn[CNT_RD] - 1];
        const int bobi = q + 1 - (T.U + T.As + 2);
        e.pos = q; e.fx = fxv(q, FX_INR); e.ctx = (uint32_t)P.b(bobi - 1) | ((uint32_t)P.b(bobi - 2) << 4);
    }
    if (cn[CNT_ATG] != cp[CNT_ATG]) { // start codon at q: bs = q+3, frame phase a = (-q) mod 3
        int64_t idx = pL + lo + (int64_t)cn[CNT_ATG] - 1;
        int bs = q + 3, eos = bs + k - 1, eoi = eos + T.Li, a = mod3(-q);
        const int fb = (0 * 3 + a) * 3;
        const double d0 = tisFwd(P, q), d1 = k == 0 ? 0.0 : plsK(P.pat(bs, k), mod3(eos + a)),
                     d2 = eoi > eos ? (double)(int64_t)(fxv(eoi, fb + 1) - fxv(eos, fb + 1)) * AUGX_FX_INV : 0.0;
        const uint64_t f0 = fxv(eoi, fb + 0);
        B.atgD[idx * 3 + 0] = d0; B.atgD[idx * 3 + 1] = d1; B.atgD[idx * 3 + 2] = d2;
        B.atgFx[idx] = f0;
    }
    if (cn[CNT_RS] != cp[CNT_RS]) { // reverse stop codon at q..q+2: bs = q+3
        int64_t idx = lo + (int64_t)cn[CNT_RS] - 1;
        if (pl == 0) {
            B.rsPos[idx] = q;
            B.rsBegin[idx] = (P.b(q) == 3 && P.b(q + 1) == 3) ? T.ln_stop_ochre : (P.b(q) == 1) ? T.ln_stop_amber : T.ln_stop_opal;
        }
        uint64_t vF[3];
        for (int a = 0; a < 3; a++) vF[a] = fxv(q + 2, (1 * 3 + a) * 3 + 0);
        for (int a = 0; a < 3; a++) B.rsFx[(pL + idx) * 3 + a] = vF[a];
    }
}

// =================================================================================================
// K2a  candidates of the variable-length states (coding exons, short introns).
//
// A variable-length state s ending at base j has many possible predecessors: every feasible exon start / intron
// start in a window that can be 15 kb long.  For each of them the score is
//        V[eop][a]  +  ( ln t(a->s) + ln emission(eop+1 .. j | s) )
// and only the first term depends on the trellis.  The second term ("te"), the tie-break key and the address of
// the first term are computed HERE, fully parallel over blocks of 8 (or 4) bases (one or two blocks per wavefront), and streamed
// to the trellis kernel, which is left with one addition and one segmented arg-max per candidate.
// The formulas are those of the reference loops (exon: src/exonmodel.cc:1059-1132; lessD:
// src/intronmodel.cc:585-629); the tie-break "larger key wins" is the reference's descending loop with strict '>'.
// =================================================================================================
#ifdef AUGX_EMU
static long long g_emuJumpProbes = 0, g_emuJumps = 0, g_emuQuietChecks = 0;
static long long g_emuQuietTiles = 0, g_emuJumpTiles = 0; // tiles the trellis took as chain-only tiles / jumped over in a run of N (trellisPiece)
static long long g_emuSlowA = 0, g_emuSlowB = 0, g_emuSlowVig = 0, g_emuSlowList = 0, g_emuSlowWaves = 0, g_emuItemWaves = 0; // emulator statistics: candidates taking the general evaluation path
#endif
struct VarDesc { // 64 bytes (kind, frame and geometry of the state are per-state constants: VarConst)
    int8_t pl;              // plane (GC class) of the end base j: selects every class-dependent array
    int8_t listSel;         // 0 LA, 1 LR, 2 LD, 3 RD, 4 ATG, 5 single reverse-stop candidate
    int8_t a;               // phase of the content prefix fields
    int8_t lenSel;          // length distribution of this exon type: 0 single, 1 initial, 2 internal, 3 terminal
    int8_t extra, fOR;
    int16_t cods;           // short intron: spliced-codon bases on the end side, 4 bits each (cod0 | cod1 << 4 | cod2 << 8)
    int32_t nList, total;
    int32_t i1;             // one past the newest list entry (piece-local index)
    int32_t eob, right, startMin; // exon geometry
    double endP;
    // end-side constants of the fast candidate evaluation (see k1SiteConsts)
    uint64_t eFx;           // content prefix at the end-side boundary
    double eD0, plsEnd;     // end-side content term (exon-terminal fwd / initial-content rev), reverse-strand ln P_ls
};
static_assert(sizeof(VarDesc) == 64, "VarDesc layout");

// wave-level bookkeeping primitives (device: cross-lane instructions; emulator: loops over the lane arrays)
#ifdef AUGX_EMU
inline void waveInclScan(int *v, int w) { for (int l = 1; l < WAVE; l++) v[w * WAVE + l] += v[w * WAVE + l - 1]; }
inline int waveRead(const int *v, int w, int lane) { return v[w * WAVE + lane]; }
#else
__device__ inline void waveInclScan(int *v, int) {
    int x = v[0];
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(x, o, 64); if (lane >= o) x += u; }
    v[0] = x;
}
__device__ inline int waveRead(const int *v, int, int lane) { return __shfl(v[0], lane, 64); }
#endif
#ifdef AUGX_EMU
inline int waveMin(const int *v, int w) { int m = v[w * WAVE]; for (int l = 1; l < WAVE; l++) m = v[w * WAVE + l] < m ? v[w * WAVE + l] : m; return m; }
#else
__device__ inline int waveMin(const int *v, int) {
    int x = v[0];
    for (int o = 32; o >= 1; o >>= 1) { const int y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
    return x;
}
#endif
AUGX_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }

// per-state constants of the variable-length states
struct VarConst { // indexed by the state's bit in the end-gate mask (DevTables::vbit; the state index itself while S <= 64)
    int kind, win, nanc, anc[4], ancWin[4], state;
    double tr[MAXPL_LDS][4]; // ln transition probability from each ancestor, per plane of the piece (the first MAXPL_LDS planes)
    ExGeom g;
};

AUGX_HD int longRow(const DevTables &T, int s) {
    int kind = T.kind[s];
    if (kind == AUGX_K_LONGDSS) return T.win[s];
    if (kind == AUGX_K_RLONGASS) return 3 + T.win[s];
    return -1;
}
AUGX_HD double lnT(const DevTables &T, int c, int a, int s) { return AUGX_GTAB(T.ln_trans)[((int64_t)c * T.S + a) * T.S + s]; }
AUGX_HD uint16_t bpFixed(int ai) { return (uint16_t)ai; }
AUGX_HD uint16_t bpVar(int ai, int dist) { return (uint16_t)((ai << 14) | (dist & 0x3FFF)); }
AUGX_HD uint32_t srcList(int ai, int sel, int frame, int64_t li) { return (SRC_LIST << 30) | ((uint32_t)ai << 28) | ((uint32_t)sel << 26) | ((uint32_t)frame << 24) | ((uint32_t)li & 0xFFFFFFu); }
AUGX_HD uint32_t srcVig(int ai, int eop) { return (SRC_VIG << 30) | ((uint32_t)ai << 28) | ((uint32_t)eop & 0xFFFFFFu); }
AUGX_HD uint32_t srcCol0(int ai, int a) { return (SRC_COL0 << 30) | ((uint32_t)ai << 28) | (uint32_t)a; }

AUGX_HD void fillVarConst(const DevTables &T, const BatchView &B, int p, int l, VarConst &VC) {
    const int kind = T.kind[l];
    VC.kind = kind; VC.win = T.win[l]; VC.nanc = T.n_anc[l] < 4 ? T.n_anc[l] : 4; VC.state = l;
    const int nPl = B.cls[p] < 0 ? 0 : B.nPlanes[p];
    for (int ai = 0; ai < 4; ai++) {
        int a = ai < VC.nanc ? T.anc[l][ai] : 0;
        VC.anc[ai] = a; VC.ancWin[ai] = T.win[a];
        VC.tr[0][ai] = AUGX_NINF;
        for (int pl = 0; pl < nPl && pl < MAXPL_LDS; pl++) VC.tr[pl][ai] = ai < VC.nanc ? lnT(T, B.planeCls[p * MAXPL + pl], a, l) : AUGX_NINF; // (planes >= nPl are never read)
    }
    VC.g = exGeom(T, (kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) ? kind : AUGX_K_INTERNAL);
}

#ifndef AUGX_DCAP
#define AUGX_DCAP 96
#endif
constexpr int DCAP = AUGX_DCAP; // descriptors of a tile that stay in LDS between the counting and the emitting pass (>= WAVE)
struct CandLds {
    VarConst vc[SP];
    VarDesc desc[NWAVES][DCAP];                          // per wavefront (= tile): descriptors of its (base, state) pairs
    uint8_t pairJ[NWAVES][WAVE], pairS[NWAVES][WAVE];    // the pairs of the current round: base offset in the tile, state
    int scan[NWAVES][3][WAVE];                           // inclusive lane scans of the pair counts of the three groups
    uint32_t cntItems[NWAVES][MAXNB], cntNonRT[NWAVES][MAXNB]; // per block: candidates (then: first candidate), candidates but RTERMINAL
    unsigned long long baseW[NWAVES][2];                 // first pair / first candidate of the tile in the batch's buffers
    uint8_t codes[NWAVES * WAVE + 2 * WAVE];             // the bases of the workgroup's tiles and 64 to either side (Piece::lcode)
    uint16_t queue[NWAVES][2][2 * WAVE];                 // candidates of the current round waiting to be evaluated with their kind: [0] exon
                                                         // states, [1] those that need the general formula (index in the round)
};

// read-only view of one piece for the candidate kernel (everything comes from HBM / L2)
struct CandCtx {
    const DevTables &T;
    const BatchView &B;
    const VarConst *vc;
    int p;
    Piece P;
    int64_t o;      // slot offset of the piece
    int64_t lo;     // list offset
    int n, c, S;
    AUGX_HD CandCtx(const DevTables &t, const BatchView &b, const VarConst *v, int pp) : T(t), B(b), vc(v), p(pp) {
        P = makePiece(T, B, p);
        o = B.off[p];
        lo = listOff(B, p);
        n = P.n; c = P.c; S = T.S;
    }
    AUGX_HD uint64_t cntAt(int q, int f) const { // number of sites of field f at bases <= q (q may be -1)
        if (q < 0) return 0;
        if (q > n - 1) q = n - 1;
        return B.cnt[fidx(o + 1 + q, f, NCNT)];
    }
    AUGX_HD int listPos(int sel, int64_t li) const { // sel: 0 LA, 1 LR, 2 LD, 3 RD; li piece-local index
        if (sel >= 2) return (sel == 2 ? B.ldEnt : B.rdEnt)[lo + li].pos;
        return (sel == 0 ? B.laPos : B.lrPos)[lo + li];
    }
    // class-dependent arrays: plane pl of [nPl][...]
    AUGX_HD int64_t pL(int pl) const { return (int64_t)pl * B.listCap + lo; }
    AUGX_HD double listC(int pl, int sel, int64_t li, int a) const { // LA: ln P_ls, LR: exon-terminal content
        return (sel == 0 ? B.laPls : B.lrEt)[(pL(pl) + li) * 3 + a];
    }
    AUGX_HD uint64_t listFx(int pl, int sel, int64_t li, int a) const {
        if (sel == 0) return B.laFx[(pL(pl) + li) * 3 + a];
        if (sel == 1) return B.lrFx[(pL(pl) + li) * 3 + a];
        return (sel == 2 ? B.ldEnt : B.rdEnt)[pL(pl) + li].fx;
    }
    AUGX_HD uint64_t fxAt(int pl, int q, int f) const { // content prefix field f up to and including base q (q < 0: empty)
        if (q < 0) return 0;
        return B.fx[(int64_t)pl * B.N * NFX + fidx(o + 1 + q, f, NFX)];
    }
    AUGX_HD Piece pieceAt(int pl) const { // the piece seen through plane pl (general evaluation paths)
        Piece Q = P;
        if (pl > 0) { Q.c = B.planeCls[p * MAXPL + pl]; Q.fx = B.fx + (int64_t)pl * B.N * NFX; }
        return Q;
    }
    AUGX_HD double lenAt(int sel, int len) const {
        return (sel == 0 ? T.len_single : sel == 1 ? T.len_initial : sel == 2 ? T.len_internal : T.len_terminal)[len];
    }
    AUGX_HD double lenIAt(int len) const { return T.len_intron[len]; }
    AUGX_HD double plsRAt(int pl, int q, int fr) const { return B.plsR[((int64_t)pl * B.N + o + 1 + q) * 3 + fr]; }
    AUGX_HD double sigAt(int q, int i) const { return B.sig[(o + 1 + q) * NSIG + i]; }
};

// descriptor of state s ending at base j: candidate range and end-side constants
// (s: the state's index in X.vc = its bit in the end-gate mask.  DENSE: the candidates go to the dense kernels of dense.h -- a
//  candidate names its predecessor by state, and a state whose predecessors are not told apart by the reading frame has one
//  candidate per predecessor end AND ancestor)
template <bool MULTI, bool DENSE = false>
AUGX_KFN void varDescribe(const CandCtx &X, int s, int j, VarDesc &D) {
    const DevTables &T = X.T;
    const Piece &P = X.P; // (sequence, stop tables and signal records only: nothing class-dependent is read through it here)
    const VarConst &VC = X.vc[s];
    const int kind = VC.kind, win = VC.win, n = X.n;
    const int pl = MULTI ? X.B.gcPlane[X.o + 1 + j] : 0; // (MULTI: the batch has a piece with more than one GC class)
    D.pl = pl;
    D.nList = 0; D.extra = 0; D.total = 0; D.listSel = 0; D.i1 = 0;
    D.eob = D.right = D.startMin = 0; D.fOR = 0; D.cods = 0x444; D.endP = AUGX_NINF;
    const ExGeom &Dg = VC.g;
    D.a = 0; D.eFx = 0; D.eD0 = 0.0; D.plsEnd = 0.0; D.lenSel = 2;
    if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
        const bool fwd = kind == AUGX_K_LESSD;
        const int f = win;
        const int eobi = fwd ? j + T.U + T.As + 2 : j + T.De + 2;
        const bool haveRight = eobi < n - 2;
        int cod0 = 4, cod1 = 4, cod2 = 4;
        if (fwd) {
            if (f == 1) { cod1 = haveRight ? P.b(eobi + 1) : 4; cod2 = haveRight ? P.b(eobi + 2) : 4; }
            if (f == 2) { cod2 = haveRight ? P.b(eobi + 1) : 4; }
        } else {
            if (f == 0) cod0 = (haveRight && P.b(eobi + 1) <= 3) ? 3 - P.b(eobi + 1) : 4;
            if (f == 1) {
                cod0 = (haveRight && P.b(eobi + 2) <= 3) ? 3 - P.b(eobi + 2) : 4;
                cod1 = (haveRight && P.b(eobi + 1) <= 3) ? 3 - P.b(eobi + 1) : 4;
            }
        }
        D.cods = (int16_t)(cod0 | (cod1 << 4) | (cod2 << 8));
        int left = j - T.dStateLen;
        if (left < 0) left = 0;
        const int fld = fwd ? CNT_LD : CNT_RD;
        const int64_t i0 = (int64_t)X.cntAt(left - 1, fld);
        D.i1 = (int32_t)X.cntAt(j - 1, fld);
        D.listSel = fwd ? 2 : 3;
        D.nList = (int)((int64_t)D.i1 - i0);
        D.extra = left == 0 ? 1 : 0; // eop = 0 reads column 0 (initial probabilities); not a splice site, so not listed
        D.total = D.nList + D.extra;
        D.endP = 0.0;
        D.eFx = X.fxAt(pl, j, fwd ? FX_INF : FX_INR);
        return;
    }
    const ExEnd e = exEnd(P, kind, win, j, Dg);
    D.eob = e.eob; D.right = e.right; D.fOR = (int8_t)e.fOR; D.startMin = e.startMin;
    D.endP = (kind == AUGX_K_SINGLE || kind == AUGX_K_TERMINAL) ? X.sigAt(j, SIG_STOPF)
             : (kind == AUGX_K_RSINGLE || kind == AUGX_K_RINITIAL) ? X.sigAt(j, SIG_TISR) : 0.0; // gate is open
    if (!(D.endP > AUGX_NINF) || e.right < 0 || e.startMax < e.startMin) return;
    {   // end-side constants of the fast candidate evaluation
        const int k = T.k, right = e.right;
        const int a = Dg.fwd ? mod3(e.fOR - right) : mod3(e.fOR + right);
        const int fb = ((Dg.fwd ? 0 : 1) * 3 + a) * 3;
        D.a = (int8_t)a;
        switch (kind) {
        case AUGX_K_INTERNAL: case AUGX_K_INITIAL:
            D.eFx = X.fxAt(pl, right - T.Le, fb + 0);
            D.eD0 = T.Le > 0 ? (double)(int64_t)(X.fxAt(pl, right, fb + 2) - X.fxAt(pl, right - T.Le, fb + 2)) * AUGX_FX_INV : 0.0;
            D.lenSel = kind == AUGX_K_INTERNAL ? 2 : 1;
            break;
        case AUGX_K_TERMINAL: case AUGX_K_SINGLE:
            D.eFx = X.fxAt(pl, right, fb + 0);
            D.lenSel = kind == AUGX_K_TERMINAL ? 3 : 0;
            break;
        default: {
            const int boip = right - (k - 1);
            D.plsEnd = (k > 0 && boip >= 0) ? X.plsRAt(pl, right, mod3(e.fOR + right - boip)) : 0.0;
            if (kind == AUGX_K_RINTERNAL || kind == AUGX_K_RTERMINAL) {
                D.eFx = X.fxAt(pl, boip - 1, fb + 0);
                D.lenSel = kind == AUGX_K_RINTERNAL ? 2 : 3;
            } else {
                const int boi = boip - T.Li;
                D.eFx = X.fxAt(pl, boi - 1, fb + 0);
                D.eD0 = T.Li > 0 ? (double)(int64_t)(X.fxAt(pl, boip - 1, fb + 1) - X.fxAt(pl, boi - 1, fb + 1)) * AUGX_FX_INV : 0.0;
                D.lenSel = kind == AUGX_K_RINITIAL ? 1 : 0;
            }
        }
        }
    }
    if (kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL) { // start codons with bob in [startMin-3, startMax-3]
        const int64_t i0 = (int64_t)X.cntAt(e.startMin - 3 - 1, CNT_ATG);
        D.i1 = (int32_t)X.cntAt(e.startMax - 3, CNT_ATG);
        D.listSel = 4;
        D.nList = (int)((int64_t)D.i1 - i0);
    } else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL) {
        D.listSel = 5; // single candidate bs = ORFleft+2 (src/exonmodel.cc:1044-1045)
        D.nList = 1;
    } else {
        const int fld = Dg.fwd ? CNT_LA : CNT_LR; // eop = bs - 1 in [startMin-1, startMax-1]
        const int64_t i0 = (int64_t)X.cntAt(e.startMin - 2, fld);
        D.i1 = (int32_t)X.cntAt(e.startMax - 1, fld);
        D.listSel = Dg.fwd ? 0 : 1;
        D.nList = (int)((int64_t)D.i1 - i0);
        D.extra = e.startMin == 0 ? 1 : 0; // bs = 0: left-truncated exon, predecessor column 0
    }
    D.total = D.nList + D.extra;
    if (DENSE && D.listSel >= 4) D.total *= VC.nanc;
}

// candidate number idx (0 = newest) of the state described by D: te = ln(transition * emission) (-inf: infeasible),
// tie-break key, and the address of the predecessor's Viterbi value
// FASTONLY: a candidate that needs the general emission formula (exNotEndPart: motif sums, short-exon cases -- 2 % of the
// candidates, thousands of cycles each) is not evaluated but reported in needSlow: the caller evaluates such candidates
// together, a wavefront full at a time, instead of stalling 63 lanes of every chunk for the one lane that needs it
// ONLY: 0 every kind, 1 short introns (lessD) only, 2 exon states only -- the caller has sorted the candidates by kind, the other
// kind's code is not even compiled into that call
template <bool MULTI, bool FASTONLY = false, int ONLY = 0, bool DENSE = false>
AUGX_KFN void varEvalItem(const CandCtx &X, int s, int j, const VarDesc &D, int idx, double &te, int &key, uint32_t &src, bool &needSlow) {
    needSlow = false;
    const DevTables &T = X.T;
    const BatchView &B = X.B;
    const VarConst &VC = X.vc[s];
    const int n = X.n, kind = VC.kind, win = VC.win, pl = MULTI ? D.pl : 0;
    const Piece P = X.pieceAt(pl); // (class-dependent reads of the general paths go through the plane of the end base)
    const ExGeom &Dg = VC.g;
    // (a piece rarely has more than MAXPL_LDS classes: those planes read the model's transition table)
    auto trOf = [&](int ai) -> double { return pl < MAXPL_LDS ? VC.tr[pl][ai] : lnT(T, B.planeCls[X.p * MAXPL + pl], VC.anc[ai], VC.state); };
    te = AUGX_NINF; key = 0; src = srcCol0(0, 0);
    if (ONLY != 2 && (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD)) {
        // written without early exits so that the loads of one candidate are all in flight together: the list entry
        // first, then the four sequence bytes, the content prefix and the length term
        const bool fwd = kind == AUGX_K_LESSD;
        const int f = win;
        const bool listed = idx < D.nList;
        const int64_t li = listed ? D.i1 - 1 - idx : 0;
        // a listed candidate is one 16-byte record: position, the two bases before the splice site, content prefix
        // (its splice-site dinucleotide is what put it on the list)
        IntronStart e;
        e.pos = 0; e.ctx = 0x44; e.fx = 0;
        if (listed) e = ldIntronStart((D.listSel == 2 ? B.ldEnt : B.rdEnt) + X.pL(pl) + li);
        const int eop = e.pos;
        const uint32_t sr = DENSE ? (uint32_t)VC.anc[0] : listed ? srcList(0, D.listSel, f, li) : srcCol0(0, VC.anc[0]);
        const int begin = eop + 1;
        const int bobi = fwd ? begin - T.De - 2 : begin - (T.U + T.As + 2);
        int bM1 = (int)(e.ctx & 15), bM2 = (int)((e.ctx >> 4) & 15);
        bool siteOk = true;
        if (!listed) { // eop = 0 (column 0): not a splice site; the reference still applies its gate (src/intronmodel.cc:595-600)
            const int bA = P.b(bobi), bB = P.b(bobi + 1);
            bM1 = P.b(bobi - 1); bM2 = P.b(bobi - 2);
            siteOk = bobi < 0 || (bobi >= 1 && bobi <= n - 2 && (fwd ? (bA == 2 && (bB == 3 || (T.dssGc && bB == 1))) : (bA == 1 && bB == 3)));
        }
        const uint64_t cFx = e.fx;
        const int eobi = fwd ? j + T.U + T.As + 2 : j + T.De + 2; // end of the biological intron
        int intronLength = eobi - bobi + 1;
        const bool lenOk = intronLength <= T.d;
        if (intronLength > T.d || intronLength < 0) intronLength = 0;
        const double lenI = X.lenIAt(intronLength);
        // stop codon across the splice (reference src/intronmodel.cc:935-958)
        const bool spliced = fwd ? (f != 0) : (f != 2);
        bool veto = false;
        if (spliced && bobi > 1) {
            int c0 = D.cods & 15, c1 = (D.cods >> 4) & 15, c2 = (D.cods >> 8) & 15;
            if (fwd) {
                if (f == 1) c0 = bM1;
                else { c0 = bM2; c1 = bM1; }
            } else {
                if (f == 0) { c1 = bM1 <= 3 ? 3 - bM1 : 4; c2 = bM2 <= 3 ? 3 - bM2 : 4; }
                else c2 = bM1 <= 3 ? 3 - bM1 : 4;
            }
            veto = stopCodon3(c0, c1, c2, T.stopMask);
        }
        const double restSeq = listed ? (double)(int64_t)(D.eFx - cFx) * AUGX_FX_INV : P.seg(fwd ? FX_INF : FX_INR, begin, j);
        const double emi = lenI + restSeq;
        if (siteOk && !veto && lenOk && emi > AUGX_NINF) {
            te = trOf(0) + emi;
            key = eop + KEY_BIAS; src = sr;
        }
        return;
    }
    if (ONLY == 1) return;
    if (D.listSel >= 4) { // predecessor is the igenic state (with UTR states: the 5' / 3' UTR states next to the gene)
        const int aiD = DENSE ? idx % VC.nanc : 0;
        if (DENSE) idx /= VC.nanc;
        const int a = VC.anc[aiD];
        int bs;
        double tisF = AUGX_NINF;
        double cPls = 0.0, cInit = 0.0;
        uint64_t cFxA = 0;
        if (D.listSel == 4) {
            const int64_t ai2 = D.i1 - 1 - idx;
            bs = B.atgPos[X.lo + ai2] + 3;
            const int64_t ap = X.pL(pl) + ai2;
            tisF = B.atgD[ap * 3 + 0]; cPls = B.atgD[ap * 3 + 1]; cInit = B.atgD[ap * 3 + 2];
            cFxA = B.atgFx[ap];
        } else
            bs = D.startMin;
        int eop = bs - Dg.bpl - 1;
        // eop == j reads the igenic cell of the CURRENT column (already final: the reference fills states in index
        // order and igenic is state 0); later columns do not exist yet
        if (!(eop < n && eop <= j)) return;
        // (dense kernels: a cell of the CURRENT column exists only for a state of lower index that is made before the candidates)
        if (DENSE && eop == j && !(T.kind[a] == AUGX_K_IGENIC && a < VC.state)) return;
        double nep;
        const int m = D.right - bs, k = T.k;
        const int bob = bs - Dg.ipo, len = D.eob - bob + 1;
        // length and reading-frame constraints first (reference src/exonmodel.cc:1716-1762 applies them last; a candidate
        // that fails them is infeasible whatever its content score): two of three start codons are out of frame
        if (len < 1 || len > T.max_exon_len) return;
        if ((kind == AUGX_K_SINGLE || kind == AUGX_K_RSINGLE) ? len % 3 != 0
            : kind == AUGX_K_INITIAL ? (len % 3 != win || len <= 2) : mod3(2 - len) != win) return;
        bool fast = false;
        if (D.listSel == 4) {
            const int eos = bs + k - 1, eoi = eos + T.Li;
            const bool ok = m > k && mod3(-bob) == D.a && (kind == AUGX_K_SINGLE ? D.right >= eoi : D.right - T.Le >= eoi);
            if (ok) {
                fast = true;
                double lenPart = AUGX_NINF;
                if (len >= 1 && len <= T.max_exon_len && (kind == AUGX_K_SINGLE ? len % 3 == 0 : (len % 3 == win && len > 2))) lenPart = X.lenAt(D.lenSel, len);
                if (!(lenPart > AUGX_NINF)) return;
                double seg1 = (double)(int64_t)(D.eFx - cFxA) * AUGX_FX_INV;
                double inner = kind == AUGX_K_SINGLE ? (cInit + seg1) : ((cInit + seg1) + D.eD0);
                nep = (tisF + (cPls + inner)) + lenPart;
            }
        } else {
            const int boip = D.right - (k - 1), boi = boip - T.Li;
            // the candidate must start with a reverse stop codon at bob (ORFleft may also be the max-length clamp or 0)
            const bool ok = m > k && (kind == AUGX_K_RTERMINAL || boi >= bs) && bob >= 0 &&
                            X.cntAt(bob, CNT_RS) != X.cntAt(bob - 1, CNT_RS);
            if (ok) {
                fast = true;
                double lenPart = AUGX_NINF;
                if (len >= 1 && len <= T.max_exon_len && (kind == AUGX_K_RSINGLE ? len % 3 == 0 : mod3(2 - len) == win)) lenPart = X.lenAt(D.lenSel, len);
                if (!(lenPart > AUGX_NINF)) return;
                const int64_t ri = (int64_t)X.cntAt(bob, CNT_RS) - 1; // the reverse stop codon at bob = bs-3
                double seg1 = (double)(int64_t)(D.eFx - B.rsFx[(X.pL(pl) + ri) * 3 + D.a]) * AUGX_FX_INV;
                double inner = kind == AUGX_K_RTERMINAL ? seg1 : (D.eD0 + seg1);
                nep = (B.rsBegin[X.lo + ri] + (D.plsEnd + inner)) + lenPart;
            }
        }
        if (FASTONLY && !fast) { needSlow = true; return; }
#ifdef AUGX_EMU
        if (!fast) g_emuSlowA++;
#endif
        if (!fast) nep = exNotEndPart(P, kind, win, bs, D.right, D.fOR, Dg, tisF);
        if (!(nep > AUGX_NINF)) return;
        te = (trOf(aiD) + D.endP) + nep;
        key = eop + KEY_BIAS;
        src = DENSE ? (uint32_t)a : eop <= 0 ? srcCol0(0, a) : srcVig(0, eop);
        return;
    }
    // predecessors are the three longass_f (forward) or rlongdss_f (reverse) states, listed per splice site
    const bool fwd = Dg.fwd;
    int eop;
    int64_t li = -1;
    // (the three loads of a listed candidate are issued together: position, content prefix and begin-side constant)
    uint64_t cFxL = 0;
    double cCL = 0.0;
    if (idx < D.nList) {
        li = D.i1 - 1 - idx;
        eop = X.listPos(D.listSel, li);
        cFxL = X.listFx(pl, D.listSel, li, D.a);
        cCL = X.listC(pl, D.listSel, li, D.a);
    } else eop = -1;
    int bs = eop + 1;
    int bob = bs - Dg.ipo, len = D.eob - bob + 1;
    if (len < 1 || len > T.max_exon_len || (kind == AUGX_K_RINITIAL && len <= 2)) return;
    double nep;
    {
        const int m = D.right - bs, k = T.k;
        bool fast = false;
        if (li >= 0 && m > k) {
            if (fwd) {
                if (kind == AUGX_K_TERMINAL || m >= k + T.Le - 1) {
                    fast = true;
                    double lenPart = (len >= 1 && len <= T.max_exon_len) ? X.lenAt(D.lenSel, len) : AUGX_NINF;
                    if (!(lenPart > AUGX_NINF)) return;
                    double seg1 = (double)(int64_t)(D.eFx - cFxL) * AUGX_FX_INV;
                    double inner = kind == AUGX_K_TERMINAL ? seg1 : (seg1 + D.eD0);
                    nep = (0.0 + (cCL + inner)) + lenPart;
                }
            } else {
                const int boip = D.right - (k - 1), eot = bs + T.Le - 1, boi = boip - T.Li;
                const bool ok = kind == AUGX_K_RINTERNAL ? (eot < boip) : (boi >= bs && eot < boi);
                if (ok) {
                    fast = true;
                    double lenPart = (len >= 1 && len <= T.max_exon_len && (kind == AUGX_K_RINTERNAL || len > 2)) ? X.lenAt(D.lenSel, len) : AUGX_NINF;
                    if (!(lenPart > AUGX_NINF)) return;
                    double seg1 = (double)(int64_t)(D.eFx - cFxL) * AUGX_FX_INV;
                    double cEt = cCL;
                    double inner = kind == AUGX_K_RINTERNAL ? (seg1 + cEt) : ((D.eD0 + seg1) + cEt);
                    nep = (0.0 + (D.plsEnd + inner)) + lenPart;
                }
            }
        }
        if (FASTONLY && !fast) { needSlow = true; return; }
#ifdef AUGX_EMU
        if (!fast) g_emuSlowB++;
#endif
        if (!fast) nep = exNotEndPart(P, kind, win, bs, D.right, D.fOR, Dg, AUGX_NINF);
    }
    if (!(nep > AUGX_NINF)) return;
    // exactly one of the (up to three) ancestors has the reading frame that fits the exon length
    for (int ai = 0; ai < VC.nanc; ai++) {
        if (win != mod3(fwd ? VC.ancWin[ai] + len : VC.ancWin[ai] - len)) continue;
        te = (trOf(ai) + D.endP) + nep;
        key = bs - Dg.bpl - 1 + KEY_BIAS;
        src = DENSE ? (uint32_t)VC.anc[ai] : li >= 0 ? srcList(ai, D.listSel, VC.ancWin[ai], li) : srcCol0(ai, VC.anc[ai]);
        break;
    }
}

// masks of the variable-length states: all but RTERMINAL / RTERMINAL (whose candidate may read the igenic cell of
// its own block and is therefore ordered after the chain states in the trellis)
AUGX_HD void varMasks(const DevTables &T, uint64_t &maskVar, uint64_t &maskRT) {
    maskVar = 0; maskRT = 0;
    for (int s2 = 0; s2 < T.S; s2++) {
        if (!T.reachable[s2]) continue;
        const int kind = T.kind[s2];
        if (kind == AUGX_K_RTERMINAL) maskRT |= 1ull << T.vbit[s2];
        else if ((kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) || kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) maskVar |= 1ull << T.vbit[s2];
    }
}

// LDS accumulators shared by the lanes of one wavefront
#ifdef AUGX_EMU
inline void ldsAdd(uint32_t *p, uint32_t v) { *p += v; }
#else
__device__ inline void ldsAdd(uint32_t *p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

// global allocation state of the candidate buffer (one per batch)
struct CandAlloc { unsigned long long pairs, items, descs; };

// Candidates of one tile of 64 bases (first base j0, first block gblk0 in the batch's block tables) of piece X.p, by ONE
// wavefront: lane = base while the pairs are collected, lane = pair while they are described, lane = candidate while
// they are evaluated.  The (base, state) pairs of a block are ordered: short introns, exons but RTERMINAL, RTERMINAL
// (each by base, then state); the candidates of the tile are contiguous in the batch's buffer, block after block.
//   pass 1: describe every pair (descriptors of the first DCAP pairs stay in LDS), count the candidates per block
//   reserve: one atomic add per tile hands out the range; the per-block tables (first candidate, counts) are written
//   pass 2: evaluate and store the candidates, 64 at a time; note the pair boundaries nearest to 1/3 and 2/3 of each
//           block's candidates (three trellis wavefronts share them)
template <int BLK, bool MULTI, bool DENSE = false>
AUGX_KFN void candTile(const CandCtx &X, CandLds &L, int w, int j0, int64_t gblk0, uint64_t maskLess, uint64_t maskVar, uint64_t maskRT) {
    constexpr int NB = WAVE / BLK;
    static_assert(NB <= MAXNB && DCAP >= WAVE, "tile layout");
    const BatchView &B = X.B;
    const int n = X.n;
    TV(uint64_t, gA); TV(uint64_t, gB); TV(uint64_t, gC);
    TV(int, nA); TV(int, nB); TV(int, nC);
    TV(int, iA); TV(int, iB); TV(int, iC);
    FOR_WLANES(t, w) {
        const int l = t & 63, j = j0 + l;
        const uint64_t gt = (j >= 1 && j < n) ? B.gate[X.o + 1 + j] : 0;
        TX(gA) = gt & maskLess; TX(gB) = gt & maskVar & ~maskLess; TX(gC) = gt & maskRT;
        TX(nA) = popc64(TX(gA)); TX(nB) = popc64(TX(gB)); TX(nC) = popc64(TX(gC));
        TX(iA) = TX(nA); TX(iB) = TX(nB); TX(iC) = TX(nC);
        if (l < NB) { L.cntItems[w][l] = 0; L.cntNonRT[w][l] = 0; }
    }
    waveInclScan(iA, w); waveInclScan(iB, w); waveInclScan(iC, w);
    FOR_WLANES(t, w) { const int l = t & 63; L.scan[w][0][l] = TX(iA); L.scan[w][1][l] = TX(iB); L.scan[w][2][l] = TX(iC); }
    WAVE_SYNC();
    const int totalPairs = waveRead(iA, w, WAVE - 1) + waveRead(iB, w, WAVE - 1) + waveRead(iC, w, WAVE - 1);
    // position (in the tile's pair order) of the first pair of each group of this lane's base
    TV(int, posA); TV(int, posB); TV(int, posC);
    FOR_WLANES(t, w) {
        const int l = t & 63, fb = (l / BLK) * BLK, lb = fb + BLK - 1;
        const int a0 = fb > 0 ? L.scan[w][0][fb - 1] : 0, a1 = L.scan[w][0][lb];
        const int b0 = fb > 0 ? L.scan[w][1][fb - 1] : 0, b1 = L.scan[w][1][lb];
        const int c0 = fb > 0 ? L.scan[w][2][fb - 1] : 0;
        const int P0 = a0 + b0 + c0;
        TX(posA) = P0 + (TX(iA) - TX(nA) - a0);
        TX(posB) = P0 + (a1 - a0) + (TX(iB) - TX(nB) - b0);
        TX(posC) = P0 + (a1 - a0) + (b1 - b0) + (TX(iC) - TX(nC) - c0);
    }
    // the pairs r0 .. r0+63 of the tile -> pairJ / pairS (every base scatters its own)
    auto expand = [&](int r0) {
        FOR_WLANES(t, w) {
            const int l = t & 63;
            int pos = TX(posA) - r0;
            for (uint64_t gg = TX(gA); gg; gg &= gg - 1, pos++)
                if (pos >= 0 && pos < WAVE) { L.pairJ[w][pos] = (uint8_t)l; L.pairS[w][pos] = (uint8_t)__builtin_ctzll(gg); }
            pos = TX(posB) - r0;
            for (uint64_t gg = TX(gB); gg; gg &= gg - 1, pos++)
                if (pos >= 0 && pos < WAVE) { L.pairJ[w][pos] = (uint8_t)l; L.pairS[w][pos] = (uint8_t)__builtin_ctzll(gg); }
            pos = TX(posC) - r0;
            for (uint64_t gg = TX(gC); gg; gg &= gg - 1, pos++)
                if (pos >= 0 && pos < WAVE) { L.pairJ[w][pos] = (uint8_t)l; L.pairS[w][pos] = (uint8_t)__builtin_ctzll(gg); }
        }
        WAVE_SYNC();
    };
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    const uint64_t cpA = B.prof ? clock64() : 0;
#endif
    // ---- pass 1: describe and count
    for (int r0 = 0; r0 < totalPairs; r0 += WAVE) {
        expand(r0);
        const int nPr = totalPairs - r0 < WAVE ? totalPairs - r0 : WAVE;
        FOR_WLANES(t, w) {
            const int l = t & 63;
            if (l < nPr) {
                const int dj = L.pairJ[w][l], s2 = L.pairS[w][l];
                VarDesc D;
                varDescribe<MULTI, DENSE>(X, s2, j0 + dj, D);
                if (r0 + l < DCAP) L.desc[w][r0 + l] = D;
                ldsAdd(&L.cntItems[w][dj / BLK], (uint32_t)D.total);
                if (!((maskRT >> s2) & 1)) ldsAdd(&L.cntNonRT[w][dj / BLK], (uint32_t)D.total);
            }
        }
        WAVE_SYNC();
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    const uint64_t cpB = B.prof ? clock64() : 0;
#endif
    // ---- reserve the tile's range; per-block tables
    TV(int, bItems); TV(int, bInc);
    FOR_WLANES(t, w) { const int l = t & 63; TX(bItems) = l < NB ? (int)L.cntItems[w][l] : 0; TX(bInc) = TX(bItems); }
    waveInclScan(bInc, w);
    const int tileItems = waveRead(bInc, w, WAVE - 1);
    FOR_WLANES(t, w) {
        if ((t & 63) == 0) {
#ifdef AUGX_EMU
            const unsigned long long bp0 = B.candAlloc->pairs, bi0 = B.candAlloc->items;
            B.candAlloc->pairs += (unsigned long long)totalPairs; B.candAlloc->items += (unsigned long long)tileItems;
#else
            const unsigned long long bp0 = atomicAdd(&B.candAlloc->pairs, (unsigned long long)totalPairs), bi0 = atomicAdd(&B.candAlloc->items, (unsigned long long)tileItems);
#endif
            L.baseW[w][0] = bp0; L.baseW[w][1] = bi0;
        }
    }
    WAVE_SYNC();
    const uint64_t pairBase = L.baseW[w][0], itemBase = L.baseW[w][1];
    if (itemBase + (uint64_t)tileItems > (uint64_t)B.itemCap) return; // the host re-runs the kernel with a buffer of the size the counters report
    FOR_WLANES(t, w) {
        const int l = t & 63;
        if (l < NB) {
            const int fb = l * BLK, lb = fb + BLK - 1;
            const int p0 = fb > 0 ? L.scan[w][0][fb - 1] + L.scan[w][1][fb - 1] + L.scan[w][2][fb - 1] : 0;
            const int p1 = L.scan[w][0][lb] + L.scan[w][1][lb] + L.scan[w][2][lb];
            const int64_t gblk = gblk0 + l;
            B.blkOff[gblk * 2] = pairBase + (uint64_t)p0; B.blkOff[gblk * 2 + 1] = itemBase + (uint64_t)(TX(bInc) - TX(bItems));
            B.blkCnt[gblk * 2] = (uint32_t)(p1 - p0); B.blkCnt[gblk * 2 + 1] = (uint32_t)TX(bItems);
            // the three trellis workers take a third each of the block's candidates but RTERMINAL (any split will do: a pair's
            // candidates may be spread over wavefronts, the cell is an atomic max)
            const uint32_t nr = L.cntNonRT[w][l];
            B.blkSplit[gblk * 3] = nr / 3; B.blkSplit[gblk * 3 + 1] = 2 * nr / 3; B.blkSplit[gblk * 3 + 2] = nr;
            L.cntItems[w][l] = (uint32_t)(TX(bInc) - TX(bItems)); // from here on: first candidate of the block, relative to the tile
        }
    }
    WAVE_SYNC();
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    const uint64_t cpC = B.prof ? clock64() : 0;
    if (B.prof && (threadIdx.x & 63) == 0) {
        unsigned long long *pp = (unsigned long long *)B.prof + (int64_t)B.nPieces * 56;
        atomicAdd(pp + 1, cpB - cpA); atomicAdd(pp + 2, cpC - cpB);
    }
#endif
    // ---- pass 2: evaluate and store
    uint32_t itemsDone = 0;
    TV(int, mnEop); // smallest predecessor position of a live candidate of the tile (BatchView::tileMinEop)
    FOR_WLANES(t, w) { TX(mnEop) = 0x7fffffff; }
    for (int r0 = 0; r0 < totalPairs; r0 += WAVE) {
        if (totalPairs > WAVE) expand(r0); // (a single round: pairJ / pairS still hold it)
        const int nPr = totalPairs - r0 < WAVE ? totalPairs - r0 : WAVE;
        // descriptor slot of pair r0 + q: the first DCAP pairs kept theirs, later ones are described again into the slots
        // of pairs already done
        TV(int, tot);
        FOR_WLANES(t, w) {
            const int l = t & 63;
            TX(tot) = 0;
            if (l < nPr) {
                if (r0 + l >= DCAP) varDescribe<MULTI, DENSE>(X, L.pairS[w][l], j0 + L.pairJ[w][l], L.desc[w][l]);
                TX(tot) = L.desc[w][r0 + l < DCAP ? r0 + l : l].total;
            }
        }
        WAVE_SYNC();
        TV(int, ibase); // inclusive prefix of the candidate counts
        FOR_WLANES(t, w) { TX(ibase) = TX(tot); }
        waveInclScan(ibase, w);
        const int totalItems = waveRead(ibase, w, WAVE - 1);
        // (the scan rows are free by now: row 0 takes the prefix, padded, for the candidates' search of their pair)
        FOR_WLANES(t, w) { const int l = t & 63; L.scan[w][0][l] = l < nPr ? TX(ibase) : 0x7fffffff; }
        WAVE_SYNC();
        // A chunk of 64 consecutive candidates mixes kinds: short-intron candidates (87 %, a short code path with one 16-byte
        // load) and, in small clusters at the end of every block, exon candidates (long paths with a dozen dependent loads), 2 %
        // of which need the general emission formula (thousands of cycles).  Evaluating a mixed chunk in place runs every path
        // with a few live lanes each.  So: short introns are evaluated in place, exon candidates queue up and are evaluated a
        // wavefront at a time, and from there the few that need the general formula queue up once more.
        int nQ[2] = {0, 0};               // (uniform) queued candidates of this round
        const bool canQueue = totalItems < 65536;
        // candidate `it` of the round by the lane of thread t; mode: 0 everything, 1 short introns only, 2 exon states (fast
        // formulas) only.  Returns true if the candidate needs the general formula (mode 2) and was not stored
        auto evalStore = [&](auto modeC, int t, int it, int q, int first) -> bool {
            constexpr int MODE = decltype(modeC)::value;
            const int dj = L.pairJ[w][q], s2 = L.pairS[w][q];
            double te; int key; uint32_t src;
            bool needSlow;
            varEvalItem<MULTI, MODE == 2, MODE, DENSE>(X, s2, j0 + dj, L.desc[w][r0 + q < DCAP ? r0 + q : q], it - first, te, key, src, needSlow);
            if (needSlow) return true;
            if (key < 0 || !(te > AUGX_NINF)) { te = AUGX_NINF; key = 0; }
            else if (key - KEY_BIAS < TX(mnEop)) TX(mnEop) = key - KEY_BIAS;
            Item I;
            // (the pair id carries the state: 6 bits of it while S <= 64, 7 in the records of the dense kernels)
            I.te = te; I.kp = ((uint32_t)(DENSE ? (((dj % BLK) << 7) | X.vc[s2].state) : (((dj % BLK) << 6) | s2)) << KEY_BITS) | ((uint32_t)key & KEY_MASK); I.src = src;
            B.items[itemBase + itemsDone + it] = I;
            return false;
        };
        auto pairOf = [&](int it, int &first) -> int { // the pair of candidate `it`: the number of pairs that end at or before it
            int pos = 0;
#pragma unroll
            for (int step = WAVE / 2; step >= 1; step >>= 1)
                if (L.scan[w][0][pos + step - 1] <= it) pos += step;
            first = pos > 0 ? L.scan[w][0][pos - 1] : 0;
            return pos;
        };
        auto push = [&](int qi, int *flag, int *itv) { // lanes with flag: their candidate joins queue qi, in lane order
            TV(int, inc);
            FOR_WLANES(t, w) { TX(inc) = flag[TI]; }
            waveInclScan(inc, w);
            const int nNew = waveRead(inc, w, WAVE - 1);
            if (nNew > 0) {
                FOR_WLANES(t, w) { if (flag[TI]) L.queue[w][qi][nQ[qi] + TX(inc) - 1] = (uint16_t)itv[TI]; }
                nQ[qi] += nNew;
                WAVE_SYNC();
            }
        };
        auto drop = [&](int qi, int cnt) { // the first cnt entries of queue qi are done: the rest moves to the front
            TV(int, mv);
            FOR_WLANES(t, w) { const int l = t & 63; TX(mv) = cnt + l < nQ[qi] ? (int)L.queue[w][qi][cnt + l] : -1; }
            WAVE_SYNC();
            FOR_WLANES(t, w) { const int l = t & 63; if (TX(mv) >= 0) L.queue[w][qi][l] = (uint16_t)TX(mv); }
            WAVE_SYNC();
            nQ[qi] -= cnt;
        };
        auto flushSlow = [&](int cnt) {
            FOR_WLANES(t, w) {
                const int l = t & 63;
                if (l < cnt) {
                    const int it = (int)L.queue[w][1][l];
                    int first;
                    const int q = pairOf(it, first);
                    evalStore(std::integral_constant<int, 0>{}, t, it, q, first);
                }
            }
            WAVE_SYNC();
            drop(1, cnt);
        };
        auto flushExon = [&](int cnt) {
            TV(int, slow); TV(int, itv);
            FOR_WLANES(t, w) {
                const int l = t & 63;
                TX(slow) = 0; TX(itv) = 0;
                if (l < cnt) {
                    const int it = (int)L.queue[w][0][l];
                    int first;
                    const int q = pairOf(it, first);
                    TX(itv) = it;
                    TX(slow) = evalStore(std::integral_constant<int, 2>{}, t, it, q, first);
                }
            }
            WAVE_SYNC();
            drop(0, cnt);
            push(1, slow, itv);
            if (nQ[1] >= WAVE) flushSlow(WAVE);
        };
        for (int base = 0; base < totalItems; base += WAVE) {
            TV(int, exon); TV(int, itv);
            FOR_WLANES(t, w) { // one candidate per lane
                const int it = base + (t & 63);
                TX(exon) = 0; TX(itv) = it;
                if (it < totalItems) {
                    int first;
                    const int q = pairOf(it, first);
                    const int kind = X.vc[L.pairS[w][q]].kind;
                    if (!canQueue) evalStore(std::integral_constant<int, 0>{}, t, it, q, first);
                    else if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) evalStore(std::integral_constant<int, 1>{}, t, it, q, first);
                    else TX(exon) = 1;
                }
            }
            push(0, exon, itv);
            if (nQ[0] >= WAVE) flushExon(WAVE);
        }
        if (nQ[0] > 0) flushExon(nQ[0]);
        if (nQ[1] > 0) flushSlow(nQ[1]);
        WAVE_SYNC();
        itemsDone += (uint32_t)totalItems;
    }
    if (B.tileMinEop) { // (batches with cut pieces: the fix-ups of the trellis size their check window by it)
        const int m = waveMin(mnEop, w);
        FOR_WLANES(t, w) { if ((t & 63) == 0) B.tileMinEop[gblk0 / NB] = m; }
    }
}

// one workgroup = NWAVES consecutive tiles of 64 bases (they belong to one piece: a chunk of CHUNK slots never spans
// pieces), one tile per wavefront; the wavefronts share the per-state constants and nothing else
template <int BLK, bool MULTI, bool DENSE = false>
AUGX_KFN void candWorkgroup(const DevTables &T, const BatchView &B, CandLds &L, int64_t wg) {
    constexpr int NB = WAVE / BLK;
    static_assert(CHUNK % (NWAVES * WAVE) == 0, "tiles of a workgroup lie in one chunk");
    const int64_t gtile0 = wg * NWAVES;
    if (gtile0 * WAVE >= B.N) return;
    const int p = B.chunkPiece[gtile0 * WAVE / CHUNK];
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    uint64_t cp0 = clock64(), cp1;
#endif
    FOR_THREADS(t) { if (t < T.S && T.vbit[t] < SP) fillVarConst(T, B, p, t, L.vc[T.vbit[t]]); }
    const int cLo = (int)(gtile0 * WAVE - B.off[p] - 1) - WAVE; // first staged base
    FOR_THREADS(t) {
        for (int i = t; i < NWAVES * WAVE + 2 * WAVE; i += NT) {
            const int q = cLo + i;
            L.codes[i] = (q >= 0 && q < B.len[p]) ? B.code[B.off[p] + 1 + q] : 4;
        }
    }
    BLOCK_SYNC();
    CandCtx X(T, B, L.vc, p);
    X.P.lcode = L.codes; X.P.lLo = cLo; X.P.lHi = cLo + NWAVES * WAVE + 2 * WAVE;
    uint64_t maskVar, maskRT;
    varMasks(T, maskVar, maskRT);
    uint64_t maskLess = 0;
    for (int s2 = 0; s2 < X.S; s2++)
        if (T.kind[s2] == AUGX_K_LESSD || T.kind[s2] == AUGX_K_RLESSD) maskLess |= 1ull << T.vbit[s2];
    maskLess &= maskVar;
    FOR_WAVES(w) {
        const int64_t gtile = gtile0 + w;
        candTile<BLK, MULTI, DENSE>(X, L, w, (int)(gtile * WAVE - X.o), gtile * NB, maskLess, maskVar, maskRT);
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    cp1 = clock64();
    if (B.prof && (threadIdx.x & 63) == 0) {
        unsigned long long *pp = (unsigned long long *)B.prof + (int64_t)B.nPieces * 56;
        atomicAdd(pp + 0, cp1 - cp0); atomicAdd(pp + 4, 1ull);
    }
#endif
}

// =================================================================================================
// K2b  trellis: one workgroup of NWAVES wavefronts per run of tiles of a piece (a segment).  Six wavefronts walk it block by
// block, synchronised by progress counters in LDS, with no workgroup barrier inside a tile of 64 bases: three workers (the near
// fixed-lag states of the block, then a third each of its candidates: the one hand-off of the cycle is among them), the chain
// wavefront (geometric intron states), the far wavefront (far fixed-lag states, cell resets, RTERMINAL candidates) and the igenic
// wavefront; two loader wavefronts stage the next tile (signal records, candidates) into the second half of the LDS buffers and
// retire the previous one (back pointers, igenic column, long-lag cells, list values) to HBM.  Every wavefront runs the
// instantiation of trellisPiece made for its role (k_trellis.hip: RW).  Runs of N are not walked: chain-only ("quiet") tiles, then
// a jump.  See trellisPiece for the schedule and DESIGN.md section 5 for the reasoning.
// =================================================================================================
constexpr int NWORK = 3, W_C = 3, W_X = 4, W_I = 5, W_LOAD = 6; // trellis workgroup: wavefronts 0..2 workers (near / late fixed-lag states + candidates), 3 geometric states, 4 far fixed-lag states, 5 igenic, 6.. loaders
constexpr int LOAD_T = (8 - W_LOAD) * WAVE;                     // threads of the loader wavefronts
constexpr int ITEM_CAP = AUGX_ITEM_CAP;   // candidates of one tile staged in LDS (the rest, if any, is read from HBM; a tile of random DNA has ~940)

struct TrellisLds {
    double ring[WAVE][SP];          // ln V of the last 64 columns, [j & 63][state]
    uint16_t bp[2][WAVE][SP];       // back pointers of the current / previous tile
    uint8_t bpc[2][WAVE][8];        // ... those of the chain states once more, compact (BatchView::bpChain)
    double sig[2][WAVE][NSIG];      // signal records of the current / next tile
    int32_t site[2][WAVE][NSITE];
    double eqPrev[2][WAVE][6];      // predecessor cells of the equalD states (lag dStateLen)
    double longW[2][WAVE][6];       // cells of the states equalD reads back at lag dStateLen (flushed to HBM tile by tile)
    uint64_t tileItem0[2];            // first candidate of the tile (index into the batch's candidate buffer)
    int32_t blkItem[2][MAXNB + 1];    // first candidate of each block relative to the tile ([NB] = end of the tile)
    uint32_t blkSplit[2][MAXNB][3]; // candidates: end of the first / second third, end of all states but RTERMINAL
    int32_t listTop[2][MAXNB][4];   // newest entry of each candidate list at the end of each block
    Item items[2][ITEM_CAP];
    double vigw[VIG_WIN];           // igenic column, newest VIG_WIN bases
    double lcVal[4][LIST_WIN][3];   // Viterbi values (three frames) of the newest LIST_WIN entries of the four lists
    double col0[SP];                // column 0 (initial probabilities)
    uint8_t gcw[2][WAVE];           // plane (GC class) of the bases of the current / next tile (multi-class pieces only)
    int flagSum;                     // sum of flagI[]: the workers are never more than one block apart, so flagSum >= NWORK * k <=> every flagI >= k
    int flagG, flagN, flagC, abortFlag; // (with flagSum: what the workers poll, side by side)
    int quietBad;                    // a tile without candidates: something in it or before it keeps it from being a chain-only tile (see trellisPiece)
    // jump over a run of N (trellisPiece): the chain states by slot -- state, signal record, ancestor index of the state itself,
    // transition term into itself, value in the column before the jump -- and the first thread whose stretch holds a nucleotide
    int jcState[8], jcSig[8], jcSelf[8], jumpFirst;
    double jcTr[8], jcV0[8];
    int flagF[NWORK], flagI[NWORK], flagNr, flagR, staged, rtPub; // blocks completed by the trellis wavefronts (see trellisPiece)
    // fix-up pass of a segment (trellisPiece<BLK, 1>): what pass 1 left at the end of the current tile, the offsets new - old
    // of the last tiles, and the last tile whose retired values did not all differ from the old ones by the tile's offset
    double oldCol[SP];
    double segDt[4];
    int lastBad;
    int needPos;                     // first base whose retired values the state at the end of the current tile still depends on
};

// loads of data this workgroup itself stored earlier (other wavefront, at least one tile barrier ago): workgroup-scope
// atomic loads, so that the compiler neither caches nor reorders them
#ifdef AUGX_EMU
inline double ldCoherent(const double *p) { return *p; }
#else
__device__ inline double ldCoherent(const double *p) {
    unsigned long long u = __hip_atomic_load((const AUGX_GLOBAL unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __longlong_as_double((long long)u);
}
#endif

#ifdef AUGX_EMU
inline bool waveAnyTrue(const int *flag, int w) { for (int l = w * WAVE; l < w * WAVE + WAVE; l++) if (flag[l]) return true; return false; }
#else
__device__ inline bool waveAnyTrue(const int *flag, int) { return __ballot(flag[0] != 0) != 0ull; }
#endif

// progress flags between the trellis wavefronts of one workgroup (all resident on one CU: spinning is safe)
#ifdef AUGX_EMU
inline void waitFlag(TrellisLds &L, const int *f, int target) { if (*f < target) { fprintf(stderr, "emu: trellis wavefront dependency violated\n"); abort(); } (void)L; }
inline void waitFlags4(TrellisLds &L, int tSum, int tG, int tN, int tC) { waitFlag(L, &L.flagSum, tSum); waitFlag(L, &L.flagG, tG); waitFlag(L, &L.flagN, tN); waitFlag(L, &L.flagC, tC); }
inline void setFlag(int *f, int v) { *f = v; }
inline void drainStores() {}
inline int readFlag(const int *f) { return *f; }
inline void addFlag(int *f) { *f += 1; }
inline void bumpFlag(int *f) { *f += 1; }
#else
__device__ inline void waitFlag(TrellisLds &L, const int *f, int target) {
    int spins = 0;
    while (*(const volatile AUGX_LDS int *)f < target && !*(const volatile AUGX_LDS int *)&L.abortFlag) {
        if (++spins > (1 << 22)) *(volatile AUGX_LDS int *)&L.abortFlag = 1; // never expected: turns a logic error into an error status
        __builtin_amdgcn_s_sleep(1);
    }
    __asm__ volatile("" ::: "memory");
}
// the workers' one wait of the block: all four counters with one trip to the LDS per poll
__device__ inline void waitFlags4(TrellisLds &L, int tSum, int tG, int tN, int tC) {
    int spins = 0;
    for (;;) {
        const int a = *(const volatile AUGX_LDS int *)&L.flagSum, b = *(const volatile AUGX_LDS int *)&L.flagG, c = *(const volatile AUGX_LDS int *)&L.flagN,
                  d = *(const volatile AUGX_LDS int *)&L.flagC, ab = *(const volatile AUGX_LDS int *)&L.abortFlag;
        if ((a >= tSum && b >= tG && c >= tN && d >= tC) || ab) break;
        if (++spins > (1 << 22)) *(volatile AUGX_LDS int *)&L.abortFlag = 1; // never expected: turns a logic error into an error status
        __builtin_amdgcn_s_sleep(1);
    }
    __asm__ volatile("" ::: "memory");
}
__device__ inline int readFlag(const int *f) {
    const int v = *(const volatile AUGX_LDS int *)f;
    __asm__ volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ inline void drainStores() { __builtin_amdgcn_s_waitcnt(0x0070); } // vmcnt(0) lgkmcnt(0): this wavefront's global stores are done
__device__ inline void addFlag(int *f) { // one count per wavefront, after everything it loaded has landed in LDS
    __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0) lgkmcnt(0)
    __asm__ volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void bumpFlag(int *f) { // right after a setFlag of the same wavefront (its LDS writes have been performed)
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void setFlag(int *f, int v) {
    __builtin_amdgcn_s_waitcnt(0xc07f); // the LDS writes of this wavefront have been performed
    __asm__ volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) *(volatile AUGX_LDS int *)f = v;
}
#endif

#if defined(AUGX_EMU) || !defined(AUGX_PROFILE)
#define PROF_MARK(X, sec) do {} while (0)
#define PROF_STAMP(X, gbk, slot) do {} while (0)
#define PROF_TSTAMP(X, cond, slot) do {} while (0)
#else
#define PROF_STAMP(X, gbk, slot) do { if ((X).B.prof && (gbk) == 1000 && (threadIdx.x & 63) == 0) (X).B.prof[(int64_t)(X).B.nPieces * 40 + (int64_t)(X).p * 16 + (slot)] = clock64(); } while (0)
#define PROF_TSTAMP(X, cond, slot) do { if ((X).B.prof && (cond) && (threadIdx.x & 63) == 0) (X).B.prof[(int64_t)(X).B.nPieces * 40 + (int64_t)(X).p * 16 + (slot)] = clock64(); } while (0)
#define PROF_MARK(X, sec) do { if ((X).B.prof) { uint64_t now_ = clock64(); (X).pacc[sec] += now_ - (X).plast; (X).plast = now_; } } while (0)
#endif
struct TrellisCtx {
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    uint64_t pacc[8], plast; // (developer build -DAUGX_PROFILE: cycle counters of the trellis wavefronts, printed with AUGX_PROF=1)
#endif
    const DevTables &T;
    const BatchView &B;
    TrellisLds &L;
    int p;
    int64_t o;      // slot offset of the piece
    int64_t lo;     // list offset
    int n, c, S;
    int vigLo;      // the igenic window holds bases > vigLo (and <= the newest chain base)
    bool multi;     // the piece has more than one GC class: transition terms follow the plane of the base
    // a segment that starts "dead" (pass 1, segment k >= 1): nothing before its first base is alive but the synch state at
    // base anchor = first base - 1 (value 0); list entries below listLo[sel] and long-lag cells before base segLo do not exist
    bool dead;
    int anchor, segLo, listLo[4];
    AUGX_HD TrellisCtx(const DevTables &t, const BatchView &b, TrellisLds &l, int pp) : T(t), B(b), L(l), p(pp) {
        o = B.off[p];
        lo = listOff(B, p);
        n = B.len[p]; c = B.cls[p]; S = T.S;
        vigLo = -1;
        multi = c >= 0 && B.nPlanes[p] > 1;
        dead = false; anchor = -(1 << 30); segLo = 0;
        for (int i = 0; i < 4; i++) listLo[i] = 0;
    }
};

#ifdef AUGX_EMU
inline void ldsMaxD(double *p, double v) { if (v > *p) *p = v; }
inline void ldsMaxU(uint32_t *p, uint32_t v) { if (v > *p) *p = v; }
inline void ldsMaxI(int *p, int v) { if (v > *p) *p = v; }
#else
__device__ inline void ldsMaxD(double *p, double v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_max_f64
__device__ inline void ldsMaxU(uint32_t *p, uint32_t v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // ds_max_u32
__device__ inline void ldsMaxI(int *p, int v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }           // ds_max_i32
#endif

// ---- staging of tile `tile` into LDS buffer `buf` by thread tid of nth (next tile: the loader wavefronts)
// CMP (fix-up pass): every value retired to HBM is first compared with what pass 1 left there; a value that is not the old one
// plus the offset of its tile marks the tile in L.lastBad
template <int BLK, bool CMP = false>
AUGX_KFN void loadTileThread(const TrellisCtx &X, int tile, int buf, int tid, int nth, bool flushOld) {
    constexpr int NB = WAVE / BLK; // blocks per tile
    // written in two phases -- every global load of the thread is issued before the first result is consumed -- so that a
    // tile costs the loaders a few memory round trips, not one per element (nth >= LOAD_T: the unroll bounds below)
    const BatchView &B = X.B;
    TrellisLds &L = X.L;
    const int n = X.n, j0 = tile * WAVE, dL = X.T.dStateLen;
    const int64_t o = X.o, g0 = o + 1 + j0;
    const int64_t gb0 = o / BLK + (int64_t)tile * NB;
    const int64_t gbL = gb0 + NB - 1 < B.nBlk ? gb0 + NB - 1 : B.nBlk - 1; // last block of the tile
    // (the candidate / pair ranges first: the loads that depend on them then overlap with everything else)
    const uint64_t firstI = gp(B.blkOff)[gb0 * 2 + 1], lastI = gp(B.blkOff)[gbL * 2 + 1] + gp(B.blkCnt)[gbL * 2 + 1];
    constexpr int KSIG = (WAVE * NSIG + LOAD_T - 1) / LOAD_T, KSITE = (WAVE * NSITE + LOAD_T - 1) / LOAD_T, KEQ = (WAVE * 6 + LOAD_T - 1) / LOAD_T;
    double vSig[KSIG], vEq[KEQ];
    int vSite[KSITE];
#pragma unroll
    for (int k = 0; k < KSIG; k++) {
        const int i = tid + k * nth;
        vSig[k] = (i < WAVE * NSIG && j0 + i / NSIG < n) ? gp(B.sig)[g0 * NSIG + i] : AUGX_NINF;
    }
#pragma unroll
    for (int k = 0; k < KSITE; k++) {
        const int i = tid + k * nth;
        vSite[k] = (i < WAVE * NSITE && j0 + i / NSITE < n) ? gp(B.site)[g0 * NSITE + i] : -1;
    }
#pragma unroll
    for (int k = 0; k < KEQ; k++) {
        const int i = tid + k * nth, q = j0 + i / 6;
        vEq[k] = (i < WAVE * 6 && dL >= WAVE && q - dL >= X.segLo && q < n) ? ldCoherent(&B.longV[(g0 - dL) * 6 + i]) : AUGX_NINF;
    }
    // block tables: thread i <= NB the first candidate of block i, i < 3 NB the split points, i < 4 NB the newest list entries
    int32_t vOff = 0;
    uint32_t vSplit = 0;
    int32_t vTop = 0;
    if (tid < NB + 1) { // first candidate of block tid relative to the tile (a tile is contiguous); [NB] = end of the tile
        int64_t gb = gb0 + tid;
        uint64_t extra = 0;
        if (tid == NB || gb >= B.nBlk) { gb = gbL; extra = gp(B.blkCnt)[gb * 2 + 1]; }
        vOff = (int32_t)(gp(B.blkOff)[gb * 2 + 1] + extra - firstI);
    }
    if (tid < NB * 3) vSplit = gb0 + tid / 3 < B.nBlk ? gp(B.blkSplit)[(gb0 + tid / 3) * 3 + tid % 3] : 0;
    if (tid < NB * 4) {
        int q = j0 + (tid / 4) * BLK + BLK - 1;
        if (q > n - 1) q = n - 1;
        vTop = (int32_t)gp(B.cnt)[fidx(o + 1 + q, CNT_LA + tid % 4, NCNT)] - 1;
    }
    uint8_t vGc = 0;
    if (X.multi && tid < WAVE) vGc = gp(B.gcPlane)[g0 + (j0 + tid < n ? tid : n - 1 - j0)];
    // (the candidate records go through registers in two halves: they are most of the tile's bytes)
    constexpr int KI = (ITEM_CAP + LOAD_T - 1) / LOAD_T, KH = (KI + 1) / 2;
    const int cntI = lastI - firstI < (uint64_t)ITEM_CAP ? (int)(lastI - firstI) : ITEM_CAP;
    Item vItem[KH];
    const Item *gi = B.items + firstI;
#pragma unroll
    for (int k = 0; k < KH; k++) { const int i = tid + k * nth; vItem[k] = ldItem(gi + (i < cntI ? i : 0)); }
    // ---- second phase
#pragma unroll
    for (int k = 0; k < KSIG; k++) {
        const int i = tid + k * nth;
        if (i < WAVE * NSIG) L.sig[buf][i / NSIG][i % NSIG] = vSig[k];
    }
#pragma unroll
    for (int k = 0; k < KSITE; k++) {
        const int i = tid + k * nth;
        if (i < WAVE * NSITE) {
            const int l = i / NSITE, sel = i % NSITE;
            if (flushOld) { // the buffer still holds the site indices of tile - 2: retire the list values of those sites to HBM
                const int si = L.site[buf][l][sel];
                if (si >= 0) {
                    double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
                    if constexpr (CMP) { // (the buffer held the sites of tile - 1)
                        const double dT = L.segDt[(tile - 2) & 3];
                        bool bad = false;
#pragma unroll
                        for (int f = 0; f < 3; f++) bad |= !(L.lcVal[sel][si & (LIST_WIN - 1)][f] == gp(a)[(X.lo + si) * 3 + f] + dT);
                        if (bad) ldsMaxI(&L.lastBad, tile - 2);
                    }
#pragma unroll
                    for (int f = 0; f < 3; f++) gp(a)[(X.lo + si) * 3 + f] = L.lcVal[sel][si & (LIST_WIN - 1)][f];
                }
            }
            L.site[buf][l][sel] = vSite[k];
        }
    }
#pragma unroll
    for (int k = 0; k < KEQ; k++) {
        const int i = tid + k * nth;
        if (i < WAVE * 6) L.eqPrev[buf][i / 6][i % 6] = vEq[k];
    }
    if (tid < NB + 1) L.blkItem[buf][tid] = vOff;
    if (tid == 0) L.tileItem0[buf] = firstI;
    if (tid < NB * 3) L.blkSplit[buf][tid / 3][tid % 3] = vSplit;
    if (tid < NB * 4) L.listTop[buf][tid / 4][tid % 4] = vTop;
    if (tid < WAVE) L.gcw[buf][tid] = vGc;
#pragma unroll
    for (int k = 0; k < KH; k++) { const int i = tid + k * nth; if (i < cntI) L.items[buf][i] = vItem[k]; }
#pragma unroll
    for (int k = 0; k < KH; k++) { const int i = tid + (KH + k) * nth; vItem[k] = ldItem(gi + (i < cntI ? i : 0)); }
#pragma unroll
    for (int k = 0; k < KH; k++) { const int i = tid + (KH + k) * nth; if (i < cntI) L.items[buf][i] = vItem[k]; }
}
// retire tile `tile` (LDS buffer buf) to HBM: back pointers (and reset of their buffer), igenic column, long-lag cells.
// (The trellis wavefronts themselves store to LDS only: a global store costs them hundreds of cycles.)
template <bool CMP = false>
AUGX_KFN void flushBpThread(const TrellisCtx &X, int tile, int buf, int tid, int nth) {
    const int j0 = tile * WAVE;
    bool bad = false;
    const double dT = CMP ? X.L.segDt[tile & 3] : 0.0;
    {   // the 64 x SP back pointers are contiguous in LDS and in HBM: move them as 64-bit words
        const uint64_t *src = (const uint64_t *)&X.L.bp[buf][0][0];
        uint64_t *srcW = (uint64_t *)&X.L.bp[buf][0][0];
        uint64_t *dst = (uint64_t *)(X.B.bp + (X.o + 1 + j0) * SP);
        constexpr int WPR = SP / 4; // words per row
        const uint64_t none = 0x0001000100010001ull * (uint64_t)BP_NONE;
        for (int i = tid; i < WAVE * WPR; i += nth) {
            if (j0 + i / WPR < X.n) gp(dst)[i] = src[i];
            srcW[i] = none;
        }
    }
    for (int i = tid; i < WAVE; i += nth) // compact back pointers of the chain states
        if (j0 + i < X.n) gp((uint64_t *)X.B.bpChain)[X.o + 1 + j0 + i] = *(const uint64_t *)&X.L.bpc[buf][i][0];
    for (int i = tid; i < WAVE; i += nth) // igenic column
        if (j0 + i >= 1 && j0 + i < X.n) {
            if constexpr (CMP) bad |= !(X.L.vigw[(j0 + i) & (VIG_WIN - 1)] == gp(X.B.vig)[X.o + 1 + j0 + i] + dT);
            gp(X.B.vig)[X.o + 1 + j0 + i] = X.L.vigw[(j0 + i) & (VIG_WIN - 1)];
        }
    for (int i = tid; i < WAVE * 6; i += nth) // cells read back at lag dStateLen
        if (j0 + i / 6 >= 1 && j0 + i / 6 < X.n) {
            if constexpr (CMP) bad |= !(X.L.longW[buf][i / 6][i % 6] == gp(X.B.longV)[(X.o + 1 + j0) * 6 + i] + dT);
            gp(X.B.longV)[(X.o + 1 + j0) * 6 + i] = X.L.longW[buf][i / 6][i % 6];
        }
    if constexpr (CMP) { if (bad) ldsMaxI(&X.L.lastBad, tile); }
}


// ---- candidates [lo, hi) (indices relative to the first candidate of the tile) of block blk of the trellis
//      wavefront: add the predecessor value, reduce per (base, state) pair, publish.  Candidates of one pair are
//      contiguous; a pair may span several chunks of 64.  Written branch-light: one LDS read per candidate.
AUGX_KFN void trellisItems(TrellisCtx &X, int w, int buf, int blk, int jb, int lo, int hi, int vigLo) {
    const BatchView &B = X.B;
    TrellisLds &L = X.L;
    const int S = X.S;
    const uint64_t tileItem0 = L.tileItem0[buf];
    // list entries at or below topK have (or may have: the far fixed-lag wavefront runs up to two blocks = LIST_AHEAD
    // entries ahead) left the LDS cache of the newest LIST_WIN entries
    constexpr int LIST_AHEAD = 32;
    const int top0 = L.listTop[buf][blk][0] - (LIST_WIN - LIST_AHEAD), top1 = L.listTop[buf][blk][1] - (LIST_WIN - LIST_AHEAD),
              top2 = L.listTop[buf][blk][2] - (LIST_WIN - LIST_AHEAD), top3 = L.listTop[buf][blk][3] - (LIST_WIN - LIST_AHEAD);
    // The cell of a (base, state) pair is the maximum over the pair's candidates, which may sit in any lanes of any chunks of
    // any of the wavefronts that share the block: every live candidate does one LDS floating-point atomic max on the cell
    // itself (reset to -inf by the far step of the block).  Nothing else is kept: which candidate won is found again by the
    // back-trace, for the few cells on the path (backtracePiece).
    PROF_MARK(X, 7);
    for (int base = lo; base < hi; base += WAVE) {
        const bool inLds = base + WAVE <= ITEM_CAP;
        FOR_WLANES(t, w) {
            const int l = t & 63, it = base + l;
            const bool valid = it < hi;
            Item I;
            if (inLds) I = L.items[buf][it];
            else if (valid) I = ldItem(B.items + tileItem0 + it);
            else { I.te = AUGX_NINF; I.kp = 0; I.src = srcCol0(0, 0); }
            const uint32_t sr = I.src, tag = sr >> 30;
            const int sel = (sr >> 26) & 3, fr = (sr >> 24) & 3, pay = (int)(sr & 0xFFFFFFu);
            const int top = sel == 0 ? top0 : sel == 1 ? top1 : sel == 2 ? top2 : top3;
            const double *ptr = tag == SRC_LIST ? &L.lcVal[sel][pay & (LIST_WIN - 1)][fr]
                                : tag == SRC_VIG ? &L.vigw[pay & (VIG_WIN - 1)] : &L.col0[sr & 0x3Fu];
            double pv = ldsLoadD(ptr);
            const bool slow = valid && ((tag == SRC_LIST && pay <= top) || (tag == SRC_VIG && pay <= vigLo));
#ifdef AUGX_EMU
            if (l == 0) g_emuItemWaves++;
            if (slow) { if (tag == SRC_LIST) g_emuSlowList++; else g_emuSlowVig++; }
#endif
            if (slow) { // the value left the LDS windows long ago: read it back from HBM
                if (tag == SRC_LIST) {
                    const double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
                    pv = ldCoherent(&a[(X.lo + pay) * 3 + fr]);
                } else
                    pv = ldCoherent(&B.vig[X.o + 1 + pay]);
            }
            if (X.dead) { // (uniform) nothing before the first base of the segment exists, but the synch state at the anchor base
                const int lLo = sel == 0 ? X.listLo[0] : sel == 1 ? X.listLo[1] : sel == 2 ? X.listLo[2] : X.listLo[3];
                if (tag == SRC_LIST ? pay < lLo : tag == SRC_VIG ? pay <= X.anchor : true) pv = (tag == SRC_VIG && pay == X.anchor) ? 0.0 : AUGX_NINF;
            }
            const double v = valid ? pv + I.te : AUGX_NINF;
            // the pair id is (base offset in the block, state)
            if (v > AUGX_NINF) ldsMaxD(&L.ring[(jb + (int)(I.kp >> (KEY_BITS + 6))) & 63][(I.kp >> KEY_BITS) & 63], v);
        }
        PROF_MARK(X, 4);
    }
    WAVE_SYNC();
    PROF_MARK(X, 5);
}

// MODE 0: pass 1 -- segment `sg` from its (true or dead) start to its end;  MODE 1: pass 2 -- fix-up of segment sg from the
// end state of the segment before it, until what it retires differs from the values of pass 1 by one constant;
// MODE 2: pass 3 -- sg is a PIECE: its first fix-up that gave up and has not been redone continues from where it stopped, still
// comparing, with no limit (one per launch; the piece's runs of this pass must not overlap);  MODE 3: such a fix-up continues
// to the end of the piece without comparing (what is left after the launches of pass 3).
template <int BLK, int MODE = 0, bool TIES = false, int RW = -1> // TIES: the chain wavefront flags near ties (dp.h: AUGX_NEAR_TIE) -- a build of its own, the default kernel pays nothing for it
AUGX_KFN void trellisPiece(const DevTables &T, const BatchView &B, TrellisLds &L, int sg) {
    constexpr int NB = WAVE / BLK;  // blocks per tile
    constexpr int SPR = WAVE / BLK; // state slots per round of 64 lanes: lane = (slot, base of the block)
    constexpr bool CMP = MODE == 1 || MODE == 2;
    int segIdx = sg, tStart = 0, tEnd = 0, ckSrc = -1; // ckSrc: ring checkpoint to start from ([seg] * 2 + slot), -1: none
    if (MODE >= 2) { // a fix-up of the piece that gave up and has not been continued: pass 3 takes the LAST one (a continuation
                     // reads only what lies behind it, so the ones further on must be settled first), the final pass the first
        segIdx = -1;
        for (int q = B.pieceSeg0[sg] + 1; q < B.pieceSeg0[sg + 1]; q++)
            if (B.segStop[q] <= -2 && B.segStop2[q] < 0) { segIdx = q; if (MODE == 3) break; }
        if (segIdx < 0) return;
        tStart = -2 - B.segStop[segIdx] + 1;
        ckSrc = segIdx * 2 + 1;
    }
    const SegDesc sd = B.segs[segIdx];
    const int p = sd.piece;
    TrellisCtx X(T, B, L, p);
    const int n = X.n, S = X.S, c = X.c;
    const int64_t o = X.o;
    const int nTiles = (n + WAVE - 1) / WAVE;
    if (MODE == 0) { tStart = sd.t0; tEnd = sd.t1; X.dead = sd.k > 0; }
    if (MODE == 1) {
        if (sd.k == 0) return;
        tStart = sd.t0; tEnd = sd.tlim + 1; ckSrc = (segIdx - 1) * 2;
    }
    if (MODE >= 2) tEnd = nTiles;
    const bool mayEndPiece = tEnd == nTiles; // a run that completes the last tile of the piece does the termination step
    int tResume = tStart; // first tile of the stretch being computed tile by tile (after a jump over a run of N: the tile it landed on)
    if (c < 0) { // multi-class piece: not decoded by this version
        FOR_THREADS(t) { if (t == 0 && mayEndPiece) { B.status[p] = AUGX_E_UNSUPPORTED; B.lnv[p] = AUGX_NINF; B.finalState[p] = -1; } }
        return;
    }
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U, dL = T.dStateLen;
    {   // ---- a piece without a single nucleotide is all intergenic (reference src/namgene.cc:205-226)
        uint64_t nuc = 0;
        for (int i = 0; i < 4; i++) nuc += B.cnt[fidx(o + n, i, NCNT)];
        if (nuc == 0 && !(MODE == 0 && sd.k == 0)) { // (segment 0 of pass 1 does the whole piece; nothing to fix up)
            FOR_THREADS(t) { if (t == 0 && MODE == 1) { B.segStop[segIdx] = sd.t0 - 1; B.segD[segIdx] = 0.0; } }
            return;
        }
        if (nuc == 0) {
            const int sy = T.synch;
            int selfAi = 0;
            for (int ai = 0; ai < T.n_anc[sy]; ai++)
                if (T.anc[sy][ai] == sy) selfAi = ai;
            FOR_THREADS(t) {
                int sySlot = 0; // chain slot of the synch state: chain states before it in state order
                for (int s2 = 0; s2 < sy; s2++)
                    if (T.reachable[s2] && (T.kind[s2] == AUGX_K_IGENIC || T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC)) sySlot++;
                for (int q = t; q < n; q += NT) {
                    for (int s2 = 0; s2 < SP; s2++) B.bp[(o + 1 + q) * SP + s2] = (s2 == sy && q >= 1) ? bpFixed(selfAi) : BP_NONE;
                    for (int c2 = 0; c2 < 8; c2++) B.bpChain[(o + 1 + q) * 8 + c2] = (c2 == sySlot && q >= 1) ? (uint8_t)selfAi : (uint8_t)0xFF;
                }
                if (t == 0) {
                    double v = B.initKind[p] == 0 ? T.ln_init[sy] : 0.0;
                    for (int q = 1; q < n; q++) v = v - T.ln4;
                    double tl = B.termKind[p] == 0 ? T.ln_term[sy] : 0.0;
                    B.lnv[p] = v + tl;
                    B.finalState[p] = (v + tl) > AUGX_NINF ? sy : -1;
                    B.status[p] = (v + tl) > AUGX_NINF ? 0 : AUGX_E_NOPATH;
                }
            }
            return;
        }
    }
    // ---- per-lane constants of the trellis wavefront.
    //   fixed-lag states (longdss, longass, equalD): FR rounds of 64 cells, lane = (state slot, base of the block)
    //   variable-length states: VR rounds of 64 cells to reset
    //   chain states (igenic, geometric introns): lane = (chain slot, base of the block)
    constexpr int FR = 3, VR = 4;
    int nNearStates = 0; // fixed-lag states with lag < 3 blocks (near, late); they occupy the rounds [0, nearRounds)
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int kind = T.kind[s2];
        if (kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) nNearStates += dssWhole < 3 * BLK;
        else if (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) nNearStates += assLag < 3 * BLK;
        else if (kind == AUGX_K_EQUALD || kind == AUGX_K_REQUALD) nNearStates += dL < 3 * BLK;
    }
    const int nearRounds = (nNearStates + SPR - 1) / SPR, farBase = nearRounds * SPR;
    bool geoLight = true; // every geometric intron state has at most two ancestors (the usual graph: equalD and itself)
    for (int s2 = 0; s2 < S; s2++)
        if (T.reachable[s2] && (T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC) && T.n_anc[s2] > 2) geoLight = false;
    // a geometric intron state fed by a NEAR fixed-lag state (equalD with a short dStateLen): its pass over a block then follows
    // the near step of that block, which the workers do (below); no shipped species with the standard graph at block size 8
    bool nearFeedsGeo = false;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2] || !(T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC)) continue;
        for (int ai = 0; ai < T.n_anc[s2]; ai++) {
            const int ak = T.kind[T.anc[s2][ai]];
            const int lagA = (ak == AUGX_K_LONGDSS || ak == AUGX_K_RLONGDSS) ? dssWhole : (ak == AUGX_K_LONGASS || ak == AUGX_K_RLONGASS) ? assLag
                             : (ak == AUGX_K_EQUALD || ak == AUGX_K_REQUALD) ? dL : 1 << 20;
            if (lagA < 3 * BLK) nearFeedsGeo = true;
        }
    }
    uint64_t chainMask = 0; // the single-base states (S <= SP = 48)
    for (int s2 = 0; s2 < S; s2++)
        if (T.kind[s2] == AUGX_K_IGENIC || T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC) chainMask |= 1ull << s2;
    // the far step of block b reads cells back to base b*BLK + BLK-1 - farMinLag; among them RTERMINAL cells, which exist
    // only once the igenic cells of their own block do: igenic must be complete up to that block (farNeedC)
    int farMinLag = 1 << 20;
    if (dssWhole >= 3 * BLK && dssWhole < farMinLag) farMinLag = dssWhole;
    if (assLag >= 3 * BLK && assLag < farMinLag) farMinLag = assLag;
    if (dL >= 3 * BLK && dL < farMinLag) farMinLag = dL;
    auto farNeedC = [&](int gb) { const int q = gb * BLK + BLK - 1 - farMinLag; return q < 0 ? 0 : q / BLK + 1; };
    TV2(int, fS, FR); TV2(int, fLag, FR); TV2(int, fSig, FR); TV2(int, fLong, FR); TV2(int, fNanc, FR);
    TV2(int, fAnc0, FR); TV2(int, fAnc1, FR); TV2(double, fTr0, FR); TV2(double, fTr1, FR);
    TV2(int, fLrow, FR); TV2(int, fList, FR); TV2(int, fFrame, FR); TV2(int, fLate, FR);
    TV2(int, vS, VR);
    TV(int, cS); TV(int, cSig); TV(int, cNanc); TV(int, cSelf); TV(int, cIsIg);
    TV2(int, cAnc, 5); TV2(double, cTr, 5);
    TV(int, fPl); TV(int, cPl); // plane (GC class) the transition terms of the lane currently belong to
    FOR_THREADS(t) {
        const int l = t & 63, slot = l / BLK;
        TX(fPl) = 0; TX(cPl) = 0;
#pragma unroll
        for (int r = 0; r < FR; r++) {
            fS[r][TI] = -1; fLag[r][TI] = 1; fSig[r][TI] = 0; fLong[r][TI] = 0; fNanc[r][TI] = 0; fAnc0[r][TI] = 0; fAnc1[r][TI] = 0;
            fTr0[r][TI] = AUGX_NINF; fTr1[r][TI] = AUGX_NINF; fLrow[r][TI] = -1; fList[r][TI] = -1; fFrame[r][TI] = 0; fLate[r][TI] = 0;
        }
#pragma unroll
        for (int r = 0; r < VR; r++) vS[r][TI] = -1;
        TX(cS) = -1; TX(cSig) = 0; TX(cNanc) = 0; TX(cSelf) = 5; TX(cIsIg) = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { cAnc[i][TI] = 0; cTr[i][TI] = AUGX_NINF; }
        int nfNear = 0, nfFar = 0, nv = 0, nc = 0;
        for (int s2 = 0; s2 < S; s2++) {
            if (!T.reachable[s2]) continue;
            const int kind = T.kind[s2];
            int lag = -1, sg = 0, lng = 0;
            switch (kind) {
            case AUGX_K_LONGDSS: lag = dssWhole; sg = SIG_DSSF; break;
            case AUGX_K_RLONGDSS: lag = dssWhole; sg = SIG_DSSR; break;
            case AUGX_K_LONGASS: lag = assLag; sg = SIG_ASSF; break;
            case AUGX_K_RLONGASS: lag = assLag; sg = SIG_ASSR; break;
            case AUGX_K_EQUALD: case AUGX_K_REQUALD: lag = dL; sg = SIG_EQD; lng = dL >= WAVE; break;
            default: break;
            }
            if (lag > 0) {
                // the near / late states (lag < 3 blocks) fill the first round(s), the far states the rest: each step of the
                // block loop then only runs its own rounds
                const int nf = lag < 3 * BLK ? nfNear++ : farBase + nfFar++;
                const int r = nf / SPR;
                if (r < FR && nf % SPR == slot) {
#pragma unroll
                    for (int rr = 0; rr < FR; rr++)
                        if (rr == r) {
                            fS[rr][TI] = s2; fLag[rr][TI] = lag; fSig[rr][TI] = sg; fLong[rr][TI] = lng;
                            fNanc[rr][TI] = T.n_anc[s2] < 2 ? T.n_anc[s2] : 2;
                            const int a0 = T.anc[s2][0], a1 = T.n_anc[s2] > 1 ? T.anc[s2][1] : 0;
                            fAnc0[rr][TI] = lng ? longRow(T, a0) : a0; fAnc1[rr][TI] = lng ? longRow(T, a1) : a1;
                            fTr0[rr][TI] = lnT(T, c, a0, s2); fTr1[rr][TI] = T.n_anc[s2] > 1 ? lnT(T, c, a1, s2) : AUGX_NINF;
                            fLrow[rr][TI] = longRow(T, s2);
                            fList[rr][TI] = kind == AUGX_K_LONGASS ? 0 : kind == AUGX_K_RLONGDSS ? 1 : kind == AUGX_K_LONGDSS ? 2 : kind == AUGX_K_RLONGASS ? 3 : -1;
                            fFrame[rr][TI] = T.win[s2];
                            // a short-lag state fed by a chain state needs the chain cells of the previous block: it is
                            // computed by the chain wavefront (rlongdss <- rgeometric at lag dss_whole)
                            bool chainAnc = false;
                            for (int ai = 0; ai < T.n_anc[s2] && ai < 2; ai++) {
                                const int ak = T.kind[T.anc[s2][ai]];
                                if (ak == AUGX_K_IGENIC || ak == AUGX_K_GEOMETRIC || ak == AUGX_K_RGEOMETRIC) chainAnc = true;
                            }
                            fLate[rr][TI] = (lag < 2 * BLK && chainAnc) ? 1 : lag >= 3 * BLK ? 2 : 0; // 0 near, 1 late, 2 far (reads blocks <= b-3 only)
                        }
                }
            } else if (kind == AUGX_K_IGENIC || kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC) {
                if (nc == slot) {
                    TX(cS) = s2; TX(cSig) = kind == AUGX_K_IGENIC ? SIG_EIG : SIG_EIN; TX(cIsIg) = kind == AUGX_K_IGENIC;
                    TX(cNanc) = T.n_anc[s2] < 5 ? T.n_anc[s2] : 5;
#pragma unroll
                    for (int ai = 0; ai < 5; ai++)
                        if (ai < TX(cNanc)) {
                            cAnc[ai][TI] = T.anc[s2][ai]; cTr[ai][TI] = lnT(T, c, T.anc[s2][ai], s2);
                            if (T.anc[s2][ai] == s2) TX(cSelf) = ai;
                        }
                }
                nc++;
            } else {
                const int r = nv / SPR;
                if (r < VR && nv % SPR == slot) {
#pragma unroll
                    for (int rr = 0; rr < VR; rr++) if (rr == r) vS[rr][TI] = s2;
                }
                nv++;
            }
        }
        // LDS tables
        if (t < SP) {
            double v = AUGX_NINF;
            if (t < S) v = B.initKind[p] == 0 ? T.ln_init[t] : (t == T.synch ? 0.0 : AUGX_NINF);
            L.col0[t] = v;
        }
        if (t == 0) { // (the progress counters count blocks / tiles of the PIECE: a run that starts at tile tStart starts them there)
            const int gb0 = tStart * NB;
            for (int i = 0; i < NWORK; i++) { L.flagF[i] = gb0; L.flagI[i] = gb0; }
            L.flagG = gb0; L.flagNr = gb0; L.flagC = gb0; L.flagN = gb0; L.flagR = tStart; L.staged = (NWAVES - W_LOAD) * tStart; L.rtPub = 0; L.abortFlag = 0; L.flagSum = NWORK * gb0;
            L.lastBad = tStart - 1;
            for (int i = 0; i < 4; i++) L.segDt[i] = 0.0;
        }
        for (int i = t; i < WAVE * SP; i += NT) {
            L.ring[i / SP][i % SP] = ckSrc >= 0 ? B.ckRing[((int64_t)ckSrc * WAVE + i / SP) * SP + i % SP] : AUGX_NINF;
            L.bp[0][i / SP][i % SP] = BP_NONE; L.bp[1][i / SP][i % SP] = BP_NONE;
        }
        for (int i = t; i < 2 * WAVE * 8; i += NT) L.bpc[i / (WAVE * 8)][(i / 8) % WAVE][i % 8] = 0xFF;
        for (int i = t; i < VIG_WIN; i += NT) L.vigw[i] = AUGX_NINF;
        for (int i = t; i < 4 * LIST_WIN * 3; i += NT) L.lcVal[i / (LIST_WIN * 3)][(i / 3) % LIST_WIN][i % 3] = AUGX_NINF;
    }
    BLOCK_SYNC();
    if (tStart > 0) { // ---- a run that does not start at the first base of the piece
        const int jA = tStart * WAVE - 1; // the base before its first one
        if (X.dead) { X.anchor = jA; X.segLo = jA + 1; }
        for (int sel = 0; sel < 4; sel++) { // entries of the candidate lists at or before jA
            const int cntA = (int)B.cnt[fidx(o + 1 + jA, CNT_LA + sel, NCNT)];
            if (X.dead) X.listLo[sel] = cntA;
            else {
                FOR_THREADS(t) { // the newest LIST_WIN entries, from what the run before left in HBM
                    const double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
                    for (int i = t; i < LIST_WIN * 3; i += NT) {
                        const int e = cntA - 1 - i / 3;
                        if (e >= 0) L.lcVal[sel][e & (LIST_WIN - 1)][i % 3] = a[(X.lo + e) * 3 + i % 3];
                    }
                }
            }
        }
        if (!X.dead) {
            FOR_THREADS(t) {
                for (int i = t; i < VIG_WIN; i += NT) {
                    const int q = jA - i;
                    if (q >= 0) L.vigw[q & (VIG_WIN - 1)] = B.vig[o + 1 + q];
                }
            }
        }
    }
    BLOCK_SYNC();
    if (X.dead) { // ---- the column before the first base of a dead start: the synch state, at value 0
        FOR_THREADS(t) {
            if (t == 0) { L.ring[X.anchor & 63][T.synch] = 0.0; L.vigw[X.anchor & (VIG_WIN - 1)] = 0.0; }
        }
    }
    // ---- column 0 = initial probabilities (reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150)
    FOR_THREADS(t) {
        if (t < S && tStart == 0) {
            const double v = L.col0[t];
            L.ring[0][t] = v;
            const int lr = longRow(T, t);
            if (lr >= 0) { B.longV[(o + 1) * 6 + lr] = v; L.longW[0][0][lr] = v; }
            if (B.cells) B.cells[(o + 1) * S + t] = v;
            if (T.kind[t] == AUGX_K_IGENIC) { B.vig[o + 1] = v; L.vigw[0] = v; }
            // base 0 as a list entry (k1SiteTerms): its values are the initial probabilities
            const int sel0 = T.kind[t] == AUGX_K_LONGASS ? 0 : T.kind[t] == AUGX_K_RLONGDSS ? 1 : -1;
            if (sel0 >= 0) {
                const int si = B.site[(o + 1) * NSITE + sel0];
                if (si >= 0) L.lcVal[sel0][si & (LIST_WIN - 1)][T.win[t]] = v;
            }
        }
        loadTileThread<BLK>(X, tStart, tStart & 1, t, NT, false);
    }
    const bool wantCells = B.cells != nullptr;
    const bool directLong = dL < 4 * WAVE; // short dStateLen: equalD would read a cell before its tile has been flushed
    BLOCK_GLOBAL_SYNC();
    // fixed-lag states of block jb (all loads first, then the two-way max); late selects the states described at fLate
    auto fixedStep = [&](int w, int buf, int jb, int late, int rlo, int rhi) { // classes in mask `late`, rounds [rlo, rhi)
        FOR_WLANES(t, w) {
            const int l = t & 63, dj = l % BLK, j = jb + dj;
            if (X.multi) { // a step of the content stairs: the transitions into the lane's states change with the class of base j
                const int pl = L.gcw[buf][j & 63];
                if (pl != TX(fPl)) {
                    TX(fPl) = pl;
                    const int cc = B.planeCls[p * MAXPL + pl];
#pragma unroll
                    for (int r = 0; r < FR; r++) {
                        const int s2 = fS[r][TI];
                        if (s2 < 0) continue;
                        fTr0[r][TI] = lnT(T, cc, T.anc[s2][0], s2);
                        fTr1[r][TI] = T.n_anc[s2] > 1 ? lnT(T, cc, T.anc[s2][1], s2) : AUGX_NINF;
                    }
                }
            }
            // written branch-light: every load of the selected rounds is issued before anything is computed
            double emi[FR], pv0[FR], pv1[FR];
            int si[FR];
#pragma unroll
            for (int r = 0; r < FR; r++) {
                emi[r] = AUGX_NINF; pv0[r] = AUGX_NINF; pv1[r] = AUGX_NINF; si[r] = -1;
                if (r < rlo || r >= rhi) continue;
                const int jp = j - fLag[r][TI];
                const bool lg = fLong[r][TI] != 0;
                // a long-lag cell (equalD: its predecessors dStateLen bases back) comes from eqPrev, which the loaders staged from HBM
                // while the tile before this one was being computed.  With dStateLen in [64, 128) the predecessor may lie IN that
                // tile -- not computed yet when eqPrev was staged: it is read from that tile's own long-lag rows, which stay in
                // LDS (the other half of the double buffer) until the next tile starts.  (The first tile of a run has no tile
                // before it in LDS; its eqPrev was staged before anything ran, from values an earlier run completed.)
                const bool inPrevTile = lg && dL < 2 * WAVE && jp >= (jb / WAVE) * WAVE - WAVE && jb / WAVE > tResume;
                const double *p0 = lg ? (inPrevTile ? &L.longW[buf ^ 1][jp & 63][fAnc0[r][TI]] : &L.eqPrev[buf][j & 63][fAnc0[r][TI]]) : &L.ring[jp & 63][fAnc0[r][TI]];
                const double *p1 = lg ? (inPrevTile ? &L.longW[buf ^ 1][jp & 63][fAnc1[r][TI]] : &L.eqPrev[buf][j & 63][fAnc1[r][TI]]) : &L.ring[jp & 63][fAnc1[r][TI]];
                emi[r] = L.sig[buf][j & 63][fSig[r][TI]];
                pv0[r] = ldsLoadD(p0); pv1[r] = ldsLoadD(p1);
                si[r] = L.site[buf][j & 63][fList[r][TI] & 3];
            }
#pragma unroll
            for (int r = 0; r < FR; r++) {
                if (r < rlo || r >= rhi) continue;
                const int s2 = fS[r][TI];
                const bool ok = j - fLag[r][TI] >= 0 && emi[r] > AUGX_NINF;
                const bool c0 = ok && pv0[r] > AUGX_NINF, c1 = ok && fNanc[r][TI] > 1 && pv1[r] > AUGX_NINF;
                const double v0 = pv0[r] + (fTr0[r][TI] + emi[r]), v1 = pv1[r] + (fTr1[r][TI] + emi[r]);
                double best = c0 ? v0 : AUGX_NINF;
                const bool take1 = c1 && v1 > best;
                best = take1 ? v1 : best;
                const uint16_t bp = take1 ? bpFixed(1) : c0 ? bpFixed(0) : BP_NONE;
                if (s2 >= 0 && j >= 1 && j < n && ((late >> fLate[r][TI]) & 1)) {
                    L.ring[j & 63][s2] = best;
                    L.bp[buf][j & 63][s2] = bp;
                    if (fLrow[r][TI] >= 0) {
                        L.longW[buf][j & 63][fLrow[r][TI]] = best;
                        if (directLong) gp(B.longV)[(o + 1 + j) * 6 + fLrow[r][TI]] = best; // short dStateLen: the tile-wise flush would come too late
                    }
                    if (wantCells) gp(B.cells)[(o + 1 + j) * S + s2] = best;
                    if (fList[r][TI] >= 0 && si[r] >= 0) {
                        L.lcVal[fList[r][TI]][si[r] & (LIST_WIN - 1)][fFrame[r][TI]] = best;
                    }
                }
            }
        }
    };
    // one pass over the lag-1 chain states (reference src/igenicmodel.cc:247-255, src/intronmodel.cc:757-786): igenic on the
    // block starting at jbIg, the geometric intron states on the block starting at jbGeo (-1: not in this pass).
    // Lane (slot, dj): tree arg-max over the ancestors before / after the state itself (ascending order, strict '>'),
    // then the 8-step recurrence along the block, one lane per base.
    // NAc: ancestors per lane that are looked at (5 in general; 2 when the pass serves geometric states with <= 2 ancestors only)
    auto chainPass = [&](auto NAc, int w, int buf, int jbIg, int jbGeo) {
        constexpr int NA = decltype(NAc)::value;
        TV(double, res);
        TV(double, prevRes);
        FOR_WLANES(t, w) { TX(res) = AUGX_NINF; TX(prevRes) = AUGX_NINF; }
        TV(double, bB); TV(double, bA); TV(double, teS); TV(double, psS);
        TV(int, aB); TV(int, aA); TV(int, rai); TV(int, jj);
        FOR_WLANES(t, w) {
            const int l = t & 63, dj = l % BLK;
            const int jbL = TX(cIsIg) ? jbIg : jbGeo;
            const int j = jbL >= 0 ? jbL + dj : -1;
            TX(jj) = j;
            if (X.multi && j >= 0 && TX(cS) >= 0) {
                const int pl = L.gcw[buf][j & 63];
                if (pl != TX(cPl)) {
                    TX(cPl) = pl;
                    const int cc = B.planeCls[p * MAXPL + pl];
#pragma unroll
                    for (int ai = 0; ai < 5; ai++)
                        if (ai < TX(cNanc)) cTr[ai][TI] = lnT(T, cc, cAnc[ai][TI], TX(cS));
                }
            }
            TX(bB) = AUGX_NINF; TX(bA) = AUGX_NINF; TX(aB) = -1; TX(aA) = -1; TX(teS) = AUGX_NINF; TX(psS) = AUGX_NINF; TX(rai) = -1;
            {
                const bool valid = TX(cS) >= 0 && j >= 1 && j < n;
                const double emiL = L.sig[buf][j & 63][TX(cSig)];
                double pv[5], v[5], te[5];
#pragma unroll
                for (int ai = 0; ai < 5; ai++) pv[ai] = ai < NA ? L.ring[(j - 1) & 63][cAnc[ai][TI]] : AUGX_NINF;
                const double emi = valid ? emiL : AUGX_NINF;
                const int self = TX(cSelf);
#pragma unroll
                for (int ai = 0; ai < 5; ai++) { te[ai] = ai < NA ? cTr[ai][TI] + emi : AUGX_NINF; v[ai] = pv[ai] + te[ai]; } // cTr = -inf beyond the last ancestor
                double vb[5], va[5];
#pragma unroll
                for (int ai = 0; ai < 5; ai++) { vb[ai] = ai < self ? v[ai] : AUGX_NINF; va[ai] = ai > self ? v[ai] : AUGX_NINF; }
                if constexpr (NA == 2) {
                    TX(teS) = self == 0 ? te[0] : self == 1 ? te[1] : AUGX_NINF;
                    TX(psS) = self == 0 ? pv[0] : self == 1 ? pv[1] : AUGX_NINF;
                    const bool t01 = vb[1] > vb[0];
                    TX(bB) = t01 ? vb[1] : vb[0];
                    TX(aB) = TX(bB) > AUGX_NINF ? (t01 ? 1 : 0) : -1;
                    TX(bA) = va[1];
                    TX(aA) = TX(bA) > AUGX_NINF ? 1 : -1;
                } else {
                TX(teS) = self == 0 ? te[0] : self == 1 ? te[1] : self == 2 ? te[2] : self == 3 ? te[3] : self == 4 ? te[4] : AUGX_NINF;
                TX(psS) = self == 0 ? pv[0] : self == 1 ? pv[1] : self == 2 ? pv[2] : self == 3 ? pv[3] : self == 4 ? pv[4] : AUGX_NINF;
                {
                    const bool t01 = vb[1] > vb[0], t23 = vb[3] > vb[2];
                    const double m01 = t01 ? vb[1] : vb[0], m23 = t23 ? vb[3] : vb[2];
                    const int i01 = t01 ? 1 : 0, i23 = t23 ? 3 : 2;
                    const bool tq = m23 > m01;
                    const double mq = tq ? m23 : m01;
                    const int iq = tq ? i23 : i01;
                    const bool t4 = vb[4] > mq;
                    TX(bB) = t4 ? vb[4] : mq;
                    TX(aB) = TX(bB) > AUGX_NINF ? (t4 ? 4 : iq) : -1;
                }
                {
                    const bool t01 = va[1] > va[0], t23 = va[3] > va[2];
                    const double m01 = t01 ? va[1] : va[0], m23 = t23 ? va[3] : va[2];
                    const int i01 = t01 ? 1 : 0, i23 = t23 ? 3 : 2;
                    const bool tq = m23 > m01;
                    const double mq = tq ? m23 : m01;
                    const int iq = tq ? i23 : i01;
                    const bool t4 = va[4] > mq;
                    TX(bA) = t4 ? va[4] : mq;
                    TX(aA) = TX(bA) > AUGX_NINF ? (t4 ? 4 : iq) : -1;
                }
                }
            }
        }
        // the value recurrence is one addition and one max per base (max(bB, self, bA) is what the three strict
        // comparisons leave); which ancestor won is decided afterwards, for all bases at once
        TV(double, mBA);
        FOR_WLANES(t, w) { TX(mBA) = TX(bB) > TX(bA) ? TX(bB) : TX(bA); }
#pragma unroll
        for (int d = 0; d < BLK; d++) {
#ifdef AUGX_EMU
            FOR_WLANES(t, w) { TX(prevRes) = (t & 63) > 0 ? res[t - 1] : AUGX_NINF; }
#else
            prevRes[0] = dppMovD<0x111, 0xf>(res[0], res[0]); // row_shr:1 (the 8 bases of a chain state share a row)
#endif
            FOR_WLANES(t, w) {
                const int l = t & 63, dj = l % BLK, j = TX(jj);
                if (dj == d) {
                    const double p0 = (d == 0 || j - 1 < 1) ? TX(psS) : TX(prevRes);
                    const double vs = p0 + TX(teS);
                    TX(psS) = p0;
                    TX(res) = vs > TX(mBA) ? vs : TX(mBA);
                }
            }
        }
        FOR_WLANES(t, w) {
            const double vs = TX(psS) + TX(teS);
            double best = TX(bB);
            int bai = TX(aB), gw = 0; // gw: which of the three won -- an ancestor before the state itself, the state itself, one after it
            if (vs > best) { best = vs; bai = TX(cSelf); gw = 1; }
            if (TX(bA) > best) { best = TX(bA); bai = TX(aA); gw = 2; }
            TX(rai) = bai;
            if (TIES && bai >= 0) { // near ties are being counted (dp.h: AUGX_NEAR_TIE): was the runner-up among (best way in from an earlier
                                         // ancestor, staying, best way in from a later ancestor) that close?  Bit 7 of the compact back pointer says so
                const double o1 = gw == 0 ? vs : TX(bB), o2 = gw == 2 ? vs : TX(bA);
                const double m2 = o1 > o2 ? o1 : o2;
                if (m2 > AUGX_NINF && best - m2 < AUGX_NEAR_TIE) TX(rai) = bai | 0x80;
            }
        }
        FOR_WLANES(t, w) {
            const int j = TX(jj);
            if (TX(cS) >= 0 && j >= 1 && j < n) {
                const int s2 = TX(cS);
                L.ring[j & 63][s2] = TX(res);
                L.bp[buf][j & 63][s2] = TX(res) > AUGX_NINF ? bpFixed(TX(rai) & 0x7F) : BP_NONE;
                L.bpc[buf][j & 63][(t & 63) / BLK] = TX(res) > AUGX_NINF ? (uint8_t)TX(rai) : (uint8_t)0xFF; // (chain slot = lane / BLK, see the set-up above; bit 7: near tie)
                if (wantCells) gp(B.cells)[(o + 1 + j) * S + s2] = TX(res);
                if (TX(cIsIg)) L.vigw[j & (VIG_WIN - 1)] = TX(res); // (HBM copy: flushed with the tile)
            }
        }
        WAVE_SYNC();
    };
    // candidates read igenic cells at least igSlack bases back; with two blocks of slack igenic may lag one block
    const int igSlackI = T.W + T.min_exon_len - T.Ds, igSlack = igSlackI < T.W ? igSlackI : T.W;
    const bool safeIg = igSlack < 2 * BLK;
    // RTERMINAL candidates (their single predecessor may be an igenic cell of their own block; nothing reads their cells
    // before lag > 2 blocks): done by the far wavefront for every block whose igenic cells are complete
    int rtNext = 0;  // (far wavefront) next block, global index, whose RTERMINAL candidates are due
    int farPre = 0;  // (far wavefront) blocks below this index already have their far step (pre-run across the tile boundary)
    // (debug / test builds only: the dense ln V matrix.  Cells of the variable-length states are final when every candidate
    //  of their block has been added: the chain wavefront, which waits for that anyway, writes them; RTERMINAL cells are
    //  written by whoever did their candidates)
    auto dumpVarCells = [&](int w, int jbD, bool rt) {
        FOR_WLANES(t, w) {
            const int j = jbD + (t & 63) % BLK;
#pragma unroll
            for (int r = 0; r < VR; r++) {
                const int s2 = vS[r][TI];
                if (s2 < 0 || j < 1 || j >= n || (T.kind[s2] == AUGX_K_RTERMINAL) != rt) continue;
                gp(B.cells)[(o + 1 + j) * S + s2] = L.ring[j & 63][s2];
            }
        }
    };
    auto doRT = [&](int w, int buf, int tile, int k, int jbNow) { // jbNow: no igenic cell at or beyond it exists yet
        const int bq = k - tile * NB;
        const int rt0 = L.blkItem[buf][bq] + (int)L.blkSplit[buf][bq][2], rt1 = L.blkItem[buf][bq + 1];
        const int vigLo = jbNow - 1 - VIG_WIN > -1 ? jbNow - 1 - VIG_WIN : -1;
        if (rt1 > rt0) trellisItems(X, w, buf, bq, k * BLK, rt0, rt1, vigLo);
        if (wantCells) dumpVarCells(w, k * BLK, true);
    };
    auto rtCatchUp = [&](int w, int buf, int tile, int igDone, int jbNow) {
        if (rtNext < tile * NB) rtNext = tile * NB;
        for (; rtNext < igDone && rtNext < (tile + 1) * NB && rtNext * BLK < n; rtNext++) doRT(w, buf, tile, rtNext, jbNow);
    };
    auto farStep = [&](int w, int buf, int jb) { // far fixed-lag states (class 2) and cell resets of the block starting at jb
        fixedStep(w, buf, jb, 4, nearRounds, FR);
        FOR_WLANES(t, w) {
            const int l = t & 63, dj = l % BLK, j = jb + dj;
            (void)l;
            // cells of variable-length states are absent unless a candidate survives
#pragma unroll
            for (int r = 0; r < VR; r++) {
                const int s2 = vS[r][TI];
                if (s2 < 0 || j < 1 || j >= n) continue;
                L.ring[j & 63][s2] = AUGX_NINF;
                if (wantCells) gp(B.cells)[(o + 1 + j) * S + s2] = AUGX_NINF;
            }
        }
    };
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    for (int i = 0; i < 8; i++) X.pacc[i] = 0;
    X.plast = clock64();
#endif
    int tLast = tStart - 1;  // last tile this run completed
    bool gaveUp = false;     // (fix-up pass) stopped at its limit without having converged
    bool quiet = false;      // the tile about to be computed is a chain-only tile (settled at the end of the tile before it)
    int quietRun = 0;        // ... and so many tiles before it were
    int jumpRetry = 0;       // no probe for a jump over a run of N before this tile (the last one found too little to jump over)
    for (int tile = tStart; tile < tEnd; tile++) {
        const int buf = tile & 1, j0 = tile * WAVE;
        FOR_WAVES(w) {
            if (w >= W_LOAD) {
                // ---- loader wavefronts: stage the next tile, retire the back pointers of the previous one
                FOR_WLANES(t, w) {
                    if (CMP) { // what pass 1 left at the end of this tile; how far back the state at its end reaches (candidates of
                               // this tile and the one before, of every later tile, the long-lag cells of equalD)
                        const int tid = t - W_LOAD * WAVE;
                        if (tid < SP) L.oldCol[tid] = gp(B.ckCol)[(o / WAVE + tile) * SP + tid];
                        if (tid == SP) {
                            const int64_t gt = o / WAVE + tile;
                            int m = gp(B.tileCross)[gt];
                            const int a = gp(B.tileMinEop)[gt], b2 = tile > 0 ? gp(B.tileMinEop)[gt - 1] : 0x7fffffff, c2 = tile * WAVE - dL - WAVE;
                            m = a < m ? a : m; m = b2 < m ? b2 : m; m = c2 < m ? c2 : m;
                            L.needPos = m;
                        }
                    }
                    if (tile + 1 < tEnd || (CMP && tile + 1 < nTiles)) loadTileThread<BLK, CMP>(X, tile + 1, buf ^ 1, t - W_LOAD * WAVE, NT - W_LOAD * WAVE, tile >= tResume + 1);
                    if (tile >= tResume + 1) flushBpThread<CMP>(X, tile - 1, buf ^ 1, t - W_LOAD * WAVE, NT - W_LOAD * WAVE);
                }
                addFlag(&L.staged); // (NWAVES - W_LOAD) counts per tile: the next tile is staged, its buffers are retired
            }
        }
        // ---- the trellis wavefronts walk the blocks of the tile, each at its own pace (progress flags in LDS).  Block b:
        //   far wavefront    (0) fixed-lag states that read blocks <= b-3 only (lag >= 24, equalD) and the cell resets of b;
        //                        runs up to two blocks ahead of the workers
        //   chain wavefront  (3) the geometric intron states of b (after (0): they are fed by far fixed-lag states and
        //                        themselves only), off the cycle
        //   igenic wavefront (3') igenic of b-1 (fed by the exon cells of b-1) -- igenic lags one block because no candidate
        //                        reads an igenic cell less than igSlack >= 2 blocks back (else: safe mode)
        //   workers 0..2     (1) the near and late fixed-lag states of b (longdss: exon cells at lag 9; rlongdss: also a chain
        //                        state at lag 9), EVERY worker for itself -- they read nothing of block b, so the three get the
        //                        same bits and store them to the same cells --, then (4) a third each of the candidates of b
        //   far wavefront        also does the RTERMINAL candidates (they may start at an igenic cell of their own block)
        //                        of every block whose igenic cells are complete
        // so that the cycle is  workers done with b (flagSum) -> near/late fixed-lag(b+1) -> candidates(b+1)  on the workers
        // alone: ONE hand-off per block.  (Rounds 1-4 had the near step on the chain wavefront: candidates(b) -> hand-off -> near
        // step -> hand-off -> candidates(b+1), 5.1 k cycles per block of which 2.5 k the near step and its two hand-offs.)
        PROF_MARK(X, 0);
        int nb = 0;
        // ---- a QUIET tile: no candidate ends in it, no fixed-lag state can emit in it (every splice-site signal is -inf: a run
        // of N, or of bases without GT / AG) and nothing but chain states (intergenic, geometric introns) is alive in the column
        // before it.  Then every cell of the tile but the chain states' is -inf whatever the wavefronts do, the chain states
        // follow themselves alone, and nothing in the tile waits for anything: the far wavefront writes the fixed-lag and
        // variable-length cells of all its blocks, the chain and igenic wavefronts run their states down the tile, no hand-offs (the
        // ring holds nothing else alive either: a slot read before its new column is written says the same, -inf) --
        // the same code as in the block loop below, so the same bits, at a quarter of the time.  (Runs of N are otherwise walked at
        // the pace of the hand-off cycle, 2 us per 8 bases -- by a fix-up that cannot converge inside them, alone on its piece:
        // 1.8 of 2.9 s of device time on a 100 Mbp genome with 11 % N, profiles/r05_genome_like_*.)
        // (whether this tile is one was settled at the end of the tile before, below)
        if (quiet) {
#ifdef AUGX_EMU
            g_emuQuietTiles++;
#endif
            for (int blk = 0; blk < NB && j0 + blk * BLK < n; blk++) {
                nb = blk + 1;
                const int jb = j0 + blk * BLK;
                FOR_WAVES(w) {
                    if (w == W_X) { // every fixed-lag state (near, late and far classes) and the cell resets of the block
                        fixedStep(w, buf, jb, 7, 0, FR);
                        FOR_WLANES(t, w) {
                            const int j = jb + (t & 63) % BLK;
#pragma unroll
                            for (int r = 0; r < VR; r++) {
                                const int s2 = vS[r][TI];
                                if (s2 < 0 || j < 1 || j >= n) continue;
                                L.ring[j & 63][s2] = AUGX_NINF;
                                if (wantCells) gp(B.cells)[(o + 1 + j) * S + s2] = AUGX_NINF;
                            }
                        }
                    }
                    if (w == W_C) {
                        if (geoLight) chainPass(std::integral_constant<int, 2>{}, w, buf, -1, jb);
                        else chainPass(std::integral_constant<int, 5>{}, w, buf, -1, jb);
                    }
                    if (w == W_I) chainPass(std::integral_constant<int, 5>{}, w, buf, jb, -1);
                }
            }
            const int gEnd = tile * NB + nb;
            rtNext = gEnd;
            FOR_WAVES(w) {
                if (w == W_X) { // the progress counters as a completed tile leaves them (nobody polls them before the barrier below)
                    if (wantCells) drainStores();
                    FOR_WLANES(t, w) {
                        if ((t & 63) == 0) {
                            for (int i = 0; i < NWORK; i++) { L.flagF[i] = gEnd; L.flagI[i] = gEnd; }
                            L.flagSum = NWORK * gEnd; L.flagG = gEnd; L.flagNr = gEnd; L.flagN = gEnd; L.flagC = gEnd; L.flagR = tile + 1; L.rtPub = gEnd;
                        }
                    }
                }
            }
        } else {
        for (int blk = 0; blk < NB && j0 + blk * BLK < n; blk++) {
            nb = blk + 1;
            const int jb = j0 + blk * BLK, gbk = tile * NB + blk;
            const int it0 = L.blkItem[buf][blk],
                      itA = it0 + (int)L.blkSplit[buf][blk][0], itB = it0 + (int)L.blkSplit[buf][blk][1], itS = it0 + (int)L.blkSplit[buf][blk][2];
            FOR_WAVES(w) {
                if (w == W_X && gbk >= farPre) { // (0) far fixed-lag states (lag >= 3 blocks, equalD) and cell resets of block b: may run two blocks ahead
                    waitFlag(L, &L.flagSum, NWORK * (gbk - 2));
                    PROF_MARK(X, 1);
                    waitFlag(L, &L.flagG, gbk - 2);       // (the chain cells it reads lie in blocks <= b-3)
                    waitFlag(L, &L.flagC, farNeedC(gbk)); // (lag 40 at block size 8: b-4, implied by the wait above; lag 39: b-3)
                    rtCatchUp(w, buf, tile, readFlag(&L.flagC), jb); // RTERMINAL candidates of the blocks whose igenic cells are complete
                    PROF_MARK(X, 3);
                    farStep(w, buf, jb);
                    if (wantCells) drainStores(); /* debug cells: keep the global stores of different wavefronts to one cell in order */ setFlag(&L.flagN, gbk + 1);
                    PROF_MARK(X, 2);
                }
            }
            FOR_WAVES(w) {
                if (w == W_C && wantCells && blk > 0) { // (debug cells) the variable-length cells of block b-1 are final
                    waitFlag(L, &L.flagSum, NWORK * gbk);
                    dumpVarCells(w, jb - BLK, false);
                }
            }
            auto geoStep = [&](int w) { // (3) geometric intron states of block b
                waitFlag(L, &L.flagN, gbk + 1);
                if (nearFeedsGeo) waitFlag(L, &L.flagNr, gbk + 1);
                PROF_MARK(X, 1);
                PROF_STAMP(X, gbk, 8);
                if (geoLight) chainPass(std::integral_constant<int, 2>{}, w, buf, -1, jb);
                else chainPass(std::integral_constant<int, 5>{}, w, buf, -1, jb);
                if (wantCells) drainStores(); /* debug cells: keep the global stores of different wavefronts to one cell in order */
                setFlag(&L.flagG, gbk + 1);
                PROF_STAMP(X, gbk, 9);
                PROF_MARK(X, 3);
            };
            FOR_WAVES(w) { if (w == W_C && !nearFeedsGeo) geoStep(w); }
            FOR_WAVES(w) {
                if (w == W_I) { // (3') igenic of block b-1, fed by its exon cells: off the cycle, on a wavefront of its own
                    waitFlag(L, &L.flagSum, NWORK * gbk);
                    if (blk > 0) chainPass(std::integral_constant<int, 5>{}, w, buf, jb - BLK, -1);
                    if (wantCells) drainStores();
                    setFlag(&L.flagC, gbk); // igenic is complete up to block b-1
                }
            }
            FOR_WAVES(w) {
                if (w < NWORK) { // (1) + (4)
                    // every worker is done with block b-1: the one hand-off of the cycle.  With it (one poll): the chain cells of
                    // blocks <= b-1 (late states), the far step of the block (cell resets; it runs ahead), igenic two blocks back
                    waitFlags4(L, NWORK * gbk, gbk, gbk + 1, safeIg ? -(1 << 30) : gbk - 1);
                    PROF_MARK(X, 1);
                    if (w == 0) PROF_STAMP(X, gbk, 6);
                    fixedStep(w, buf, jb, 3, 0, nearRounds); // near (class 0) and late (class 1) states
                    if (nearFeedsGeo && w == 0) setFlag(&L.flagNr, gbk + 1);
                    if (w == 0) PROF_STAMP(X, gbk, 7);
                    PROF_MARK(X, 2);
                    if (safeIg) waitFlag(L, &L.flagC, gbk); // igenic: no candidate reads a cell less than two blocks back (safe mode: one)
                    PROF_MARK(X, 1);
                    if (w < 2) PROF_STAMP(X, gbk, w == 0 ? 2 : 4);
                    const int vigLo = jb - 1 - VIG_WIN > -1 ? jb - 1 - VIG_WIN : -1;
                    const int lo2 = w == 0 ? it0 : w == 1 ? itA : itB, hi2 = w == 0 ? itA : w == 1 ? itB : itS;
                    if (hi2 > lo2) trellisItems(X, w, buf, blk, jb, lo2, hi2, vigLo);
                    if (wantCells) drainStores(); /* debug cells: keep the global stores of different wavefronts to one cell in order */ setFlag(&L.flagI[w], gbk + 1); bumpFlag(&L.flagSum);
                    if (w < 2) PROF_STAMP(X, gbk, w == 0 ? 3 : 5);
                    PROF_MARK(X, 2);
                }
            }
            FOR_WAVES(w) { if (w == W_C && nearFeedsGeo) geoStep(w); } // (after the near step of the block, which feeds it)
        }
        FOR_WAVES(w) { if (w == 0) PROF_TSTAMP(X, tile == 124, 10); }
        // ---- end of the tile: igenic of the last block, then the RTERMINAL candidates of the last two blocks
        FOR_WAVES(w) {
            if (w == W_C && nb > 0 && wantCells) {
                const int gLast = tile * NB + nb - 1, jbLast = j0 + (nb - 1) * BLK;
                waitFlag(L, &L.flagSum, NWORK * (gLast + 1));
                dumpVarCells(w, jbLast, false);
                drainStores();
            }
            if (w == W_I && nb > 0) {
                const int gLast = tile * NB + nb - 1, jbLast = j0 + (nb - 1) * BLK;
                waitFlag(L, &L.flagSum, NWORK * (gLast + 1));
                PROF_TSTAMP(X, tile == 124, 11);
                chainPass(std::integral_constant<int, 5>{}, w, buf, jbLast, -1);
                if (wantCells) drainStores(); /* debug cells: keep the global stores of different wavefronts to one cell in order */ setFlag(&L.flagC, gLast + 1);
                PROF_TSTAMP(X, tile == 124, 12);
            }
        }
        // the remaining RTERMINAL candidates of the tile (their back pointers live in this tile's buffer) are shared by
        // the workers and the far wavefront; before that the far wavefront already does its step for the first block of
        // the next tile, as soon as the loaders have staged it
        FOR_WAVES(w) {
            if (w == W_X && nb > 0) {
                const int gN = (tile + 1) * NB, jbN = j0 + WAVE;
                if (!CMP && tile + 1 < tEnd && jbN < n) { // (a fix-up may stop after any tile: it must not touch the next one)
                    waitFlag(L, &L.staged, (NWAVES - W_LOAD) * (tile + 1));
                    waitFlag(L, &L.flagSum, NWORK * (gN - 2));
                    waitFlag(L, &L.flagG, gN - 2);
                    waitFlag(L, &L.flagC, farNeedC(gN));
                    rtCatchUp(w, buf, tile, readFlag(&L.flagC), j0 + nb * BLK);
                    farStep(w, buf ^ 1, jbN);
                    if (wantCells) drainStores();
                    setFlag(&L.flagN, gN + 1);
                    farPre = gN + 1;
                }
                if (rtNext < tile * NB) rtNext = tile * NB;
                FOR_WLANES(t, w) { if ((t & 63) == 0) L.rtPub = rtNext; }
                if (wantCells) drainStores();
                setFlag(&L.flagR, tile + 1);
            }
        }
        FOR_WAVES(w) {
            if (w < NWORK && nb > 0) {
                const int gEnd = tile * NB + nb;
                waitFlag(L, &L.flagR, tile + 1);
                const int k = readFlag(&L.rtPub) + w;
                if (k < gEnd) { waitFlag(L, &L.flagC, gEnd); doRT(w, buf, tile, k, j0 + nb * BLK); }
                if (w == 0) PROF_TSTAMP(X, tile == 124, 13);
            }
        }
        FOR_WAVES(w) {
            if (w == W_X && nb > 0) {
                const int gEnd = tile * NB + nb;
                waitFlag(L, &L.flagC, gEnd);
                for (int k = rtNext + NWORK; k < gEnd; k++) doRT(w, buf, tile, k, j0 + nb * BLK);
                rtNext = gEnd;
                PROF_TSTAMP(X, tile == 124, 14);
            }
        }
        } // (not a quiet tile)
        BLOCK_GLOBAL_SYNC(); // stores of this tile are visible to later (coherent) loads; the staged tile is complete
        FOR_WAVES(w) { if (w == 0) PROF_TSTAMP(X, tile == 124, 15); }
        tLast = tile;
        if (B.ckCol) { // ---- pieces cut into segments: the column at the end of the tile (complete: every wavefront has passed the barrier)
            FOR_THREADS(t) {
                if (t < SP) {
                    const double nv = t < S ? L.ring[63][t] : AUGX_NINF;
                    if constexpr (CMP) { // the tile's offset = new - old of the synch state; a finite offset commutes with -inf = -inf
                        const double nSy = L.ring[63][T.synch], oSy = L.oldCol[T.synch];
                        const bool fin = nSy > AUGX_NINF && oSy > AUGX_NINF;
                        const double dT = fin ? nSy - oSy : 0.0;
                        if (!fin || !(nv == L.oldCol[t] + dT)) ldsMaxI(&L.lastBad, tile);
                        if (t == 0) {
                            if (tile > tStart && dT != L.segDt[(tile - 1) & 3]) ldsMaxI(&L.lastBad, tile - 1); // a run of verified tiles shares ONE offset
                            L.segDt[tile & 3] = dT;
                        }
                    }
                    gp(B.ckCol)[(o / WAVE + tile) * SP + t] = nv;
                }
            }
        }
        if constexpr (CMP) {
            BLOCK_SYNC();
            // tiles up to tile - 1 have been retired and compared (by the loaders, during this tile); converged when the last
            // segCheckTiles of them -- more than the longest look-back of any state -- all differ from pass 1 by one offset
            const int lb = L.lastBad;
            const int needTile = L.needPos < 0 ? -1 : L.needPos / WAVE; // the tile of that base: verified from its first base on
            BLOCK_SYNC();
            // converged: every tile from needTile to tile - 1 -- at least the last three -- has been verified with one offset.
            // (Then the ring at the end of this tile and every retired value a later cell can read are old + D, by induction
            // over the columns from the fully verified column at the end of tile - 2)
            const int win = B.segCheckTiles > 0x10000 ? B.segCheckTiles : 3; // (tests force the give-up path with a huge check length)
            bool atSeam = false;
            if (MODE == 2) { // a continuation crosses regions other runs have rewritten: the stored values change frame where such a
                             // run stopped.  It must not stop on top of such a seam -- its offset would belong to one side only
                for (int q = B.pieceSeg0[p] + 1; q < B.pieceSeg0[p + 1]; q++) {
                    const int e = q == segIdx ? -100 : B.segStop[q] >= -1 ? B.segStop[q] : B.segStop2[q];
                    if (e == tile || e == tile - 1) atSeam = true;
                }
            }
            if (lb < needTile && (tile - 1) - lb >= win && !atSeam) break;
            if (MODE == 1 && tile + 1 == tEnd) gaveUp = true;
        }
        // ---- is the NEXT tile a quiet one?  No candidate ends in it, no fixed-lag state can emit in it (every splice-site signal
        // is -inf, no list site: a run of N), equalD's staged predecessors are dead, and nothing but chain states is alive in the
        // ring (the 64 columns before it, whose slots its columns take over: the wavefronts of a quiet tile do not wait for one
        // another, so what a slot still holds must be what it will hold).  (Here, not at the head of the next iteration: every
        // wavefront is at this point anyway, and the loaders have nothing in flight.)
        quietRun = quiet ? quietRun + 1 : 0;
        quiet = false;
        if (tile + 1 < tEnd && L.blkItem[buf ^ 1][NB] == 0) { // (uniform: staged before the barrier of this tile)
            const int jn0 = j0 + WAVE;
#ifdef AUGX_EMU
            g_emuQuietChecks++;
#endif
            FOR_THREADS(t) { if (t == 0) L.quietBad = 0; }
            BLOCK_SYNC();
            FOR_THREADS(t) {
                bool bad = false;
                if (t < WAVE && jn0 + t < n) {
                    const double *sg = L.sig[buf ^ 1][t];
                    bad = sg[SIG_DSSF] > AUGX_NINF || sg[SIG_DSSR] > AUGX_NINF || sg[SIG_ASSF] > AUGX_NINF || sg[SIG_ASSR] > AUGX_NINF;
                    for (int k = 0; k < NSITE; k++) bad |= L.site[buf ^ 1][t][k] >= 0;
                    bool eqDead = dL >= 2 * WAVE; // (equalD reads its predecessors from the staged copy: all dead?)
                    for (int k = 0; k < 6; k++) eqDead &= !(L.eqPrev[buf ^ 1][t][k] > AUGX_NINF);
                    bad |= sg[SIG_EQD] > AUGX_NINF && !eqDead;
                }
                for (int i = t; i < WAVE * SP; i += NT) {
                    const int s2 = i % SP;
                    if (s2 < S && !((chainMask >> s2) & 1) && L.ring[i / SP][s2] > AUGX_NINF) bad = true;
                }
                if (bad) ldsMaxI(&L.quietBad, 1);
            }
            BLOCK_SYNC();
            quiet = L.quietBad == 0;
        }
        // ---- JUMP over a run of N.  Deep inside one -- the last jumpAfter tiles were quiet, so every look-back of the tiles ahead
        // (64 columns, dStateLen) lies in verified quiet tiles, and the prefix counts say no nucleotide follows for a stretch -- the
        // columns ahead are known without walking them: the chain states follow themselves, v(j) = v(j-1) + (ln t(s,s) + emission(j)),
        // every other cell is -inf.  All additions of the decode are exact, so the sums may be taken in any order: every
        // thread sums a stretch, the stretches are chained, and the threads write what the tiles would have retired -- back pointers,
        // the igenic column, the long-lag cells, the column at the end of every tile -- straight to HBM, 1 ms per Mbp instead of the
        // 170 ms of the block loop.  The run resumes tile by tile at least two tiles before the first nucleotide.  (A fix-up or a
        // continuation cannot converge inside a run of N -- the true run carries intron states through it that a dead start does
        // not have -- so it is they who walk the long runs, alone on their piece: 1.8 of 2.9 s of device time on a 100 Mbp genome
        // with 11 % N before this, profiles/r05_*.)  In the comparing passes the jumped tiles count as not verified.
        constexpr int JUMP_MIN = 32; // tiles worth the fixed cost of a jump
        const int jumpAfter = (dL + 3 * WAVE - 1) / WAVE + 1;
        if (quiet && quietRun >= jumpAfter && tile >= jumpRetry && tEnd - 1 - (tile + 1) >= JUMP_MIN) {
#ifdef AUGX_EMU
            g_emuJumpProbes++;
#endif
            const int jS = (tile + 1) * WAVE;
            const int hi = (tEnd - 1) * WAVE < n ? (tEnd - 1) * WAVE : n; // (the last tile of the run is always walked)
            const int step = (hi - jS + NT - 1) / NT;
            auto nucAt = [&](int q) { uint32_t c4 = 0; for (int i = 0; i < 4; i++) c4 += gp(B.cnt)[fidx(o + 1 + q, i, NCNT)]; return c4; };
            auto sitesAt = [&](int q) { uint32_t c4 = 0; for (int i = CNT_ATG; i <= CNT_RS; i++) c4 += gp(B.cnt)[fidx(o + 1 + q, i, NCNT)]; return c4; };
            FOR_THREADS(t) { if (t == 0) L.jumpFirst = -NT; }
            BLOCK_SYNC();
            FOR_THREADS(t) { // the first stretch with a nucleotide (or a list site: there is none without one, but the jump rests on it),
                             // or, in a piece with several GC classes, with a base of another plane than the one the run of N is in now
                             // (the transition terms follow the plane: a jump stays in one)
                const int qa = jS + t * step, qb = jS + (t + 1) * step < hi ? jS + (t + 1) * step : hi;
                bool stop = qa < qb && (nucAt(qb - 1) != nucAt(qa - 1) || sitesAt(qb - 1) != sitesAt(qa - 1));
                if (X.multi && qa < qb && !stop) {
                    const uint8_t pl0 = gp(B.gcPlane)[o + 1 + jS - 1];
                    for (int q = qa; q < qb; q++) stop |= gp(B.gcPlane)[o + 1 + q] != pl0;
                }
                if (stop) ldsMaxI(&L.jumpFirst, -t);
            }
            BLOCK_SYNC();
            const int fT = -L.jumpFirst;
            int jN = jS + fT * step; // [jS, jN) holds no nucleotide
            if (jN > hi) jN = hi;
            const int tJ = (jN - 2 * WAVE) / WAVE; // land two tiles before what may hold one
            BLOCK_SYNC();
            jumpRetry = tile + 16; // (no jump from here: the probe is not repeated at every tile of the run's tail)
            if (tJ - (tile + 1) >= JUMP_MIN) {
                const int jE = tJ * WAVE, len = jE - jS, per = (len + NT - 1) / NT;
                double *part = (double *)&L.items[0][0]; // [NT][8] sums of the stretches (the staged candidates are not needed: the run re-stages where it lands)
                static_assert(sizeof(L.items) >= sizeof(double) * NT * 8, "scratch of the jump");
                // retire the last tile walked (its back pointers, igenic cells, long-lag cells; quiet tiles have no list sites)
                FOR_THREADS(t) {
                    flushBpThread(X, tile, buf, t, NT);
                    if (t < WAVE && (t & 63) % BLK == 0 && (t & 63) / BLK < 8) { // the chain states by slot (lane = slot * BLK of any wavefront holds the slot's constants)
                        const int slot = (t & 63) / BLK, self = TX(cSelf), s2 = TX(cS);
                        const bool live = s2 >= 0 && self < 5;
                        // (the transition term of the plane the run of N lies in -- not the lane's copy: only the chain wavefronts keep theirs current)
                        const int cc = X.multi ? B.planeCls[p * MAXPL + (int)B.gcPlane[o + 1 + jS - 1]] : c;
                        L.jcState[slot] = live ? s2 : -1; L.jcSig[slot] = TX(cSig); L.jcSelf[slot] = self;
                        L.jcTr[slot] = live ? lnT(T, cc, s2, s2) : AUGX_NINF;
                        L.jcV0[slot] = live ? L.ring[(jS - 1) & 63][s2] : AUGX_NINF;
                    }
                }
                BLOCK_SYNC();
                constexpr int NC = 8;
                FOR_THREADS(t) { // sums of the stretches; with several GC classes the transition terms follow the plane: one plane only
                    const int qa = jS + t * per, qb = jS + (t + 1) * per < jE ? jS + (t + 1) * per : jE;
                    double acc[NC];
                    for (int c = 0; c < NC; c++) acc[c] = 0.0;
                    for (int q = qa; q < qb; q++) {
                        const double eIg = gp(B.sig)[(o + 1 + q) * NSIG + SIG_EIG], eIn = gp(B.sig)[(o + 1 + q) * NSIG + SIG_EIN];
                        for (int c = 0; c < NC; c++) acc[c] += L.jcTr[c] + (L.jcSig[c] == SIG_EIG ? eIg : eIn);
                    }
                    for (int c = 0; c < NC; c++) part[t * NC + c] = acc[c];
                }
                BLOCK_SYNC();
                {
#ifdef AUGX_EMU
                    g_emuJumpTiles += tJ - (tile + 1); g_emuJumps++;
#endif
                    FOR_THREADS(t) {
                        const int qa = jS + t * per, qb = jS + (t + 1) * per < jE ? jS + (t + 1) * per : jE;
                        double v[NC];
                        for (int c = 0; c < NC; c++) {
                            double a = L.jcV0[c];
                            for (int u = 0; u < t; u++) a += part[u * NC + c];
                            v[c] = L.jcState[c] >= 0 ? a : AUGX_NINF;
                        }
                        int igC = -1;
                        for (int c = 0; c < NC; c++) if (L.jcState[c] >= 0 && L.jcSig[c] == SIG_EIG) igC = c;
                        uint64_t rowW[SP / 4], bpcW = ~0ull;
                        int rowOf = -1;
                        for (int q = qa; q < qb; q++) {
                            const int64_t g = o + 1 + q;
                            const double eIg = gp(B.sig)[g * NSIG + SIG_EIG], eIn = gp(B.sig)[g * NSIG + SIG_EIN];
                            int alive = 0;
                            for (int c = 0; c < NC; c++) {
                                if (L.jcState[c] < 0) continue;
                                v[c] += L.jcTr[c] + (L.jcSig[c] == SIG_EIG ? eIg : eIn);
                                alive |= (v[c] > AUGX_NINF ? 1 : 0) << c;
                            }
                            if (alive != rowOf) { // the row of back pointers (8-byte words, as flushBpThread moves them) changes only when a state dies
                                rowOf = alive;
                                bpcW = ~0ull;
#pragma unroll
                                for (int wd = 0; wd < SP / 4; wd++) rowW[wd] = 0x0001000100010001ull * (uint64_t)BP_NONE;
                                for (int c = 0; c < NC; c++) {
                                    if (!((alive >> c) & 1)) continue;
                                    const int s2 = L.jcState[c];
                                    bpcW = (bpcW & ~(0xFFull << (8 * c))) | ((uint64_t)(uint8_t)L.jcSelf[c] << (8 * c));
#pragma unroll
                                    for (int wd = 0; wd < SP / 4; wd++)
                                        if (wd == s2 / 4) rowW[wd] = (rowW[wd] & ~(0xFFFFull << (16 * (s2 % 4)))) | ((uint64_t)bpFixed(L.jcSelf[c]) << (16 * (s2 % 4)));
                                }
                            }
#pragma unroll
                            for (int wd = 0; wd < SP / 4; wd++) gp((uint64_t *)B.bp)[g * (SP / 4) + wd] = rowW[wd];
                            gp((uint64_t *)B.bpChain)[g] = bpcW;
                            double vIg = AUGX_NINF;
#pragma unroll
                            for (int c = 0; c < NC; c++) vIg = c == igC ? v[c] : vIg;
                            gp(B.vig)[g] = vIg;
                            for (int k = 0; k < 6; k++) gp(B.longV)[g * 6 + k] = AUGX_NINF;
                            if (wantCells) {
                                for (int s2 = 0; s2 < S; s2++) gp(B.cells)[g * S + s2] = AUGX_NINF;
                                for (int c = 0; c < NC; c++) if (L.jcState[c] >= 0) gp(B.cells)[g * S + L.jcState[c]] = v[c];
                            }
                            if (B.ckCol && (q & 63) == 63) { // the column at the end of a tile
                                for (int s2 = 0; s2 < SP; s2++) gp(B.ckCol)[(o / WAVE + q / WAVE) * SP + s2] = AUGX_NINF;
                                for (int c = 0; c < NC; c++) if (L.jcState[c] >= 0) gp(B.ckCol)[(o / WAVE + q / WAVE) * SP + L.jcState[c]] = v[c];
                            }
                            if (q >= jE - WAVE) { // what the ring holds where the run lands
                                for (int s2 = 0; s2 < SP; s2++) L.ring[q & 63][s2] = AUGX_NINF;
                                for (int c = 0; c < NC; c++) if (L.jcState[c] >= 0) L.ring[q & 63][L.jcState[c]] = v[c];
                            }
                            if (q >= jE - VIG_WIN) L.vigw[q & (VIG_WIN - 1)] = vIg;
                        }
                        for (int i = t; i < 2 * WAVE * 6; i += NT) L.longW[i / (WAVE * 6)][(i / 6) % WAVE][i % 6] = AUGX_NINF;
                        if (t == 0) { // the progress counters as a run that starts at tile tJ has them; in the comparing passes the tiles jumped count as not verified
                            const int gbJ = tJ * NB;
                            for (int i = 0; i < NWORK; i++) { L.flagF[i] = gbJ; L.flagI[i] = gbJ; }
                            L.flagSum = NWORK * gbJ; L.flagG = gbJ; L.flagNr = gbJ; L.flagN = gbJ; L.flagC = gbJ; L.flagR = tJ; L.rtPub = 0;
                            L.staged = (NWAVES - W_LOAD) * tJ;
                            if (CMP && L.lastBad < tJ - 1) L.lastBad = tJ - 1;
                        }
                    }
                    BLOCK_GLOBAL_SYNC();
                    FOR_THREADS(t) { loadTileThread<BLK>(X, tJ, tJ & 1, t, NT, false); }
                    BLOCK_GLOBAL_SYNC();
                    rtNext = tJ * NB; farPre = 0;
                    tResume = tJ; tLast = tJ - 1;
                    quiet = false; quietRun = 0;
                    tile = tJ - 1; // (the loop goes on with tile tJ)
                }
            }
        }
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    if (B.prof && (threadIdx.x & 63) == 0 && threadIdx.x < 5 * WAVE)
        for (int i = 0; i < 8; i++) B.prof[((int64_t)p * 5 + (threadIdx.x >> 6)) * 8 + i] = X.pacc[i];
#endif
    const bool lastOfPiece = mayEndPiece && tLast == nTiles - 1;
    // ---- back pointers of the last tile; list values of the sites of the last two tiles (the back-trace reads them)
    FOR_THREADS(t) {
        if (tLast >= tStart) {
            flushBpThread(X, tLast, tLast & 1, t, NT);
            // (a fix-up has always staged the tile after its last one: the other buffer then holds THAT tile's sites, and the
            //  sites of tile tLast - 1 have been retired when it was staged)
            const bool otherIsNext = CMP && tLast + 1 < nTiles;
            for (int i = t; i < 2 * WAVE * NSITE; i += NT) {
                const int bsel = i / (WAVE * NSITE), l = (i / NSITE) % WAVE, sel = i % NSITE;
                if (bsel == 1 && (tLast - tResume + 1 < 2 || otherIsNext)) continue; // (the other buffer was never loaded)
                const int si = L.site[(tLast & 1) ^ bsel][l][sel];
                if (si >= 0) {
                    double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
                    for (int f = 0; f < 3; f++) a[(X.lo + si) * 3 + f] = L.lcVal[sel][si & (LIST_WIN - 1)][f];
                }
            }
        }
        // ---- segments: the ring where this run stopped (pass 1: for the fix-up of the next segment; a fix-up that gave up: for pass 3)
        if (B.ckRing && !lastOfPiece && (MODE == 0 || (MODE == 1 && gaveUp))) {
            const int64_t slot = (int64_t)segIdx * 2 + (MODE == 1 ? 1 : 0);
            for (int i = t; i < WAVE * SP; i += NT) B.ckRing[(slot * WAVE + i / SP) * SP + i % SP] = L.ring[i / SP][i % SP];
        }
        if (t == 0) {
            if (MODE == 1) { B.segStop[segIdx] = gaveUp ? -2 - tLast : tLast; B.segD[segIdx] = gaveUp ? 0.0 : L.segDt[tLast & 3]; }
            if (MODE >= 2) { // converged after tile tLast (or ran to the end of the piece: then it is in the frame it started in)
                B.segStop2[segIdx] = tLast; B.segD2[segIdx] = (CMP && !lastOfPiece) ? L.segDt[tLast & 3] : 0.0;
                B.pieceCovered[sg] = tLast;
            }
            if (B.segStatus && L.abortFlag) B.segStatus[segIdx] = 1;
        }
    }
    if (!lastOfPiece) return;
    // ---- termination (reference NAMGene::getViterbiPath, src/namgene.cc:442-457)
    FOR_THREADS(t) {
        if (t == 0) {
            double maxV = AUGX_NINF;
            int state = -1;
            for (int i = 0; i < S; i++) {
                double tl = B.termKind[p] == 0 ? T.ln_term[i] : (i == T.synch ? 0.0 : AUGX_NINF);
                double v = L.ring[(n - 1) & 63][i] + tl;
                if (v > maxV) { maxV = v; state = i; }
            }
            B.lnv[p] = maxV;
            B.finalState[p] = state;
            B.status[p] = L.abortFlag ? AUGX_E_HIP : state >= 0 ? 0 : AUGX_E_NOPATH;
        }
    }
}

// tileCross[t] = the smallest predecessor position any candidate ending in the tiles t+1 .. t+W of the same piece reads
AUGX_HD void tileCrossOne(const BatchView &B, int64_t gt) {
    const int p = B.chunkPiece[gt * WAVE / CHUNK];
    const int64_t last = B.off[p] / WAVE + (B.len[p] + WAVE - 1) / WAVE - 1; // last tile of the piece
    int m = 0x7fffffff;
    for (int64_t u = gt + 1; u <= gt + B.segCheckTiles && u <= last; u++) { const int v = B.tileMinEop[u]; m = v < m ? v : m; }
    B.tileCross[gt] = m;
}

// =================================================================================================
// Forward algorithm (reference NAMGene::viterbiAndForward with needForwardTable, src/namgene.cc:168-365; the per-state sums
// `fwdsum` of src/igenicmodel.cc:247-287, src/intronmodel.cc:585-629,757-858, src/exonmodel.cc:1059-1145): the same recurrences
// as the trellis with every maximum replaced by a sum -- in ln space, ln(sum_i exp(x_i)) around the largest term.  It runs only
// when posterior sampling is asked for, after the Viterbi decode, and reuses the candidates of K2a (their te is V-independent).
// Kept simple: one workgroup per piece, one block of BLK bases after the other with workgroup barriers between the stages
//   A fixed-lag states   B geometric chain   C candidates of the variable-length states but RTERMINAL   D igenic chain
//   E RTERMINAL candidates (they may start at an igenic cell of their own block)
// The dense ln F matrix goes to HBM (the host's sampler walks it); the last 64 columns also live in an LDS ring.
// Sums are not exact (exp / log round): the result agrees with the reference's LLDouble sums to ~1e-12 relative.
// =================================================================================================
// (not inlined: exp and log1p are a few hundred instructions, and the forward kernel has dozens of call sites -- inlined, a block
//  of 8 bases walked through more code than the instruction cache holds)
__attribute__((noinline)) AUGX_HD double lse2(double a, double b) { // ln(e^a + e^b)
    if (!(a > AUGX_NINF)) return b;
    if (!(b > AUGX_NINF)) return a;
    return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
#ifdef AUGX_EMU
inline void ldsAddU(unsigned long long *p, unsigned long long v) { *p += v; }
#else
__device__ inline void ldsAddU(unsigned long long *p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // ds_add_u64
#endif
// the sum over the candidates of a cell is taken in fixed point (terms in (0, 1], 2^-52 apart; < 2048 of them): integer adds
// commute, so the result does not depend on the order the wavefronts' atomics arrive in -- the same forward matrix, hence the
// same sampled paths, run after run
constexpr double FWD_FIX = 4503599627370496.0; // 2^52
struct FwdLds {
    double ring[WAVE][SP];   // ln F of the last 64 columns, [j & 63][state]
    double cmax[8][SP];      // variable-length cells of the current block: largest candidate ...
    unsigned long long csum[8][SP]; // ... and the sum of exp(candidate - largest), fixed point (FWD_FIX)
    double lt[SP][SP];       // ln transition a -> s of the piece's first class (a piece with one class never leaves it)
    double sg[2][8][NSIG];   // signal records of the block ([parity of the block]; the next block's are staged meanwhile)
    uint64_t bOff[2];        // candidates of the block: first record ...
    uint32_t bCnt[2][2];     // ... their number, and how many of them are not RTERMINAL  ([parity of the block])
    uint8_t anc[SP][4];      // ancestor ai of a variable-length state (the candidate records name it by index)
    uint8_t cellKind[SP];    // 1: variable-length state (all but RTERMINAL), 2: RTERMINAL, 0: no cell of the candidate steps
    // the single-base ("chain") states: slot 0..6 the geometric intron states, slot 7 the intergenic state
    int chS[8], chNa[8], chAnc[8][5];
    uint8_t chLive[8][5];    // the ancestor is a chain state itself (its cell of the base before is made in the same step)
    uint8_t chSelf[8], chOther[8]; // ... the state itself is among them / another chain state is
    int chNd[8], chDead[8][5];     // the ancestors that are not chain states, packed
    double oth[8][8];        // [slot][base of the block]: ln sum over the other ancestors' contributions
};
// One workgroup walks a piece block by block (BLK bases).  What a block needs from HBM is independent of the forward values and is
// in flight a block ahead: the candidate records in registers (the first NT of a block; more are rare and loaded when needed),
// the block's offsets and signal records in LDS.  A candidate's value is computed once and kept in a register between the
// max, the sum and the cell step.  The columns of a block are written to HBM as they are made; only columns older than the ring
// (64 bases) are read back from there, so the stores are fenced every fourth block, not every block.
template <int BLK>
AUGX_KFN void forwardPiece(const DevTables &T, const BatchView &B, FwdLds &L, int p) {
    const int n = B.len[p], S = T.S, c0 = B.cls[p];
    const int64_t o = B.off[p];
    double *F = B.fwd + (o + 1) * S; // F[q * S + s]
    // (the BatchView's pointers and the piece's scalars once, in registers: the stores to F could alias them for all the compiler
    //  knows, and every one of them would be fetched again after every store)
    const double *gSig = B.sig + (o + 1) * NSIG;
    const Item *gItems = B.items;
    const uint64_t *gBlkOff = B.blkOff;
    const uint32_t *gBlkCnt = B.blkCnt, *gBlkSplit = B.blkSplit;
    const uint8_t *gPlane = B.gcPlane + o + 1;
    const int32_t *gPlaneCls = B.planeCls + p * MAXPL;
    const int initKind = B.initKind[p], termKind = B.termKind[p], synch = T.synch;
    if (c0 < 0) { FOR_THREADS(t) { if (t == 0) B.lnFwd[p] = AUGX_NINF; } return; }
    const bool multi = B.nPlanes[p] > 1;
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U, dL = T.dStateLen;
    auto clsAt = [&](int j) __attribute__((always_inline)) { return multi ? (int)gp(gPlaneCls)[gp(gPlane)[j]] : c0; };
    // (two loads behind a branch, not one load through a selected pointer: that would be a flat load, slower than either)
    const double *gTrans = T.ln_trans;
    auto trn = [&](int cc, int a, int s2) __attribute__((always_inline)) -> double { if (multi) return gp(gTrans)[((int64_t)cc * S + a) * S + s2]; return ldsLoadD(&L.lt[a][s2]); };
    // value of state a at base q: from the ring while no base of the block being computed (first base jb) has taken its column,
    // from HBM before
    auto at = [&](int q, int a, int jb) __attribute__((always_inline)) -> double {
        if (q < 0) return AUGX_NINF;
        if (jb + BLK - 1 - q < WAVE) return ldsLoadD(&L.ring[q & 63][a]);
        return ldCoherent(&F[(int64_t)q * S + a]);
    };
    FOR_THREADS(t) {
        for (int i = t; i < WAVE * SP; i += NT) (*lp(&L.ring[i / SP][i % SP])) = AUGX_NINF;
        for (int i = t; i < SP * SP; i += NT) (*lp(&L.lt[i / SP][i % SP])) = (i / SP < S && i % SP < S) ? lnT(T, c0, i / SP, i % SP) : AUGX_NINF;
        if (t < SP) {
            const int k = t < S && T.reachable[t] ? T.kind[t] : -1;
            const bool var = (k >= AUGX_K_SINGLE && k <= AUGX_K_RTERMINAL) || k == AUGX_K_LESSD || k == AUGX_K_RLESSD;
            (*lp(&L.cellKind[t])) = !var ? 0 : k == AUGX_K_RTERMINAL ? 2 : 1;
            for (int ai = 0; ai < 4; ai++) (*lp(&L.anc[t][ai])) = (uint8_t)((t < S && ai < T.n_anc[t]) ? T.anc[t][ai] : 0);
        }
        for (int i = t; i < n * S; i += NT) gp(F)[i] = AUGX_NINF; // (absent cells stay -inf)
    }
    BLOCK_GLOBAL_SYNC();
    FOR_THREADS(t) { // column 0 = initial probabilities (reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150)
        if (t < S) {
            const double v = initKind == 0 ? T.ln_init[t] : (t == synch ? 0.0 : AUGX_NINF);
            (*lp(&L.ring[0][t])) = v;
            gp(F)[t] = v;
        }
    }
    BLOCK_GLOBAL_SYNC();
    // the states by group (uniform)
    int fixS[24], nFix = 0, geoS[8], nGeo = 0, igS = -1;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int k = T.kind[s2];
        if (k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS || k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS || k == AUGX_K_EQUALD || k == AUGX_K_REQUALD) { if (nFix < 24) fixS[nFix++] = s2; }
        else if (k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC) { if (nGeo < 8) geoS[nGeo++] = s2; }
        else if (k == AUGX_K_IGENIC) igS = s2;
    }
    const int nBlocks = (n + BLK - 1) / BLK;
    const int64_t gb0 = o / BLK;
    // per-thread constants of the fixed-lag step (thread = (state, base of the block)): state, lag, signal, ancestors -- read from the
    // tables once, not in every block
    TV(int, fS2); TV(int, fLag); TV(int, fSg); TV(int, fNa); TV2(int, fAn, 4);
    FOR_THREADS(t) {
        TX(fS2) = -1; TX(fLag) = 1; TX(fSg) = 0; TX(fNa) = 0;
        for (int ai = 0; ai < 4; ai++) fAn[ai][TI] = 0;
        if (t >= WAVE && t - WAVE < nFix * BLK) {
            const int s2 = fixS[(t - WAVE) / BLK], k = T.kind[s2];
            TX(fS2) = s2;
            TX(fLag) = (k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS) ? dssWhole : (k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS) ? assLag : dL;
            TX(fSg) = k == AUGX_K_LONGDSS ? SIG_DSSF : k == AUGX_K_RLONGDSS ? SIG_DSSR : k == AUGX_K_LONGASS ? SIG_ASSF : k == AUGX_K_RLONGASS ? SIG_ASSR : SIG_EQD;
            TX(fNa) = T.n_anc[s2] < 4 ? T.n_anc[s2] : 4;
            for (int ai = 0; ai < 4; ai++) if (ai < TX(fNa)) fAn[ai][TI] = T.anc[s2][ai];
        }
        if (t < 8) {
            const int s2 = t < nGeo ? geoS[t] : (t == 7 ? igS : -1);
            (*lp(&L.chS[t])) = s2;
            (*lp(&L.chNa[t])) = s2 >= 0 ? (T.n_anc[s2] < 5 ? T.n_anc[s2] : 5) : 0;
            for (int ai = 0; ai < 5; ai++) {
                const int a = (s2 >= 0 && ai < (*lp(&L.chNa[t]))) ? T.anc[s2][ai] : 0;
                const int ak = T.kind[a];
                (*lp(&L.chAnc[t][ai])) = a;
                (*lp(&L.chLive[t][ai])) = ak == AUGX_K_IGENIC || ak == AUGX_K_GEOMETRIC || ak == AUGX_K_RGEOMETRIC;
            }
            int nd = 0;
            bool self = false, other = false;
            for (int ai = 0; ai < 5; ai++) {
                if (!(s2 >= 0 && ai < (*lp(&L.chNa[t])))) continue;
                if ((*lp(&L.chLive[t][ai]))) { if ((*lp(&L.chAnc[t][ai])) == s2) self = true; else other = true; }
                else (*lp(&L.chDead[t][nd++])) = (*lp(&L.chAnc[t][ai]));
            }
            for (int k = nd; k < 5; k++) (*lp(&L.chDead[t][k])) = 0;
            (*lp(&L.chNd[t])) = nd; (*lp(&L.chSelf[t])) = self; (*lp(&L.chOther[t])) = other;
        }
    }
    // the candidate of this thread in the current block (cur*) and in the next one (nxt*): record, and whether there is one
    // Wavefront 0 does nothing but the runs of the chain states (register to register, LDS in and out): it never has a load or a
    // store to HBM in flight, so nothing in its runs waits for memory.  The other NTW threads (u = t - WAVE) share the rest.
    constexpr int NTW = NT - WAVE;
    TV(Item, curI); TV(Item, nxtI); TV(double, cv); TV(int, cdj); TV(int, cs2);
    FOR_THREADS(t) { // block 0: offsets now, its records below
        if (t == NT - 1) { (*lp(&L.bOff[0])) = gp(gBlkOff)[gb0 * 2 + 1]; (*lp(&L.bCnt[0][0])) = gp(gBlkCnt)[gb0 * 2 + 1]; (*lp(&L.bCnt[0][1])) = gp(gBlkSplit)[gb0 * 3 + 2]; }
        TX(cv) = AUGX_NINF; TX(cdj) = 0; TX(cs2) = 0;
    }
    BLOCK_SYNC();
    FOR_THREADS(t) {
        TX(nxtI) = (t >= WAVE && (uint32_t)(t - WAVE) < (*lp(&L.bCnt[0][0]))) ? ldItem(gItems + (*lp(&L.bOff[0])) + (t - WAVE)) : Item{AUGX_NINF, 0u, 0u};
        if (t >= NT - BLK * NSIG) { const int i = t - (NT - BLK * NSIG); (*lp(&L.sg[0][i / NSIG][i % NSIG])) = i / NSIG < n ? gp(gSig)[(int64_t)(i / NSIG) * NSIG + i % NSIG] : AUGX_NINF; }
    }
    BLOCK_SYNC();
    // a chain state at base j: first what reaches it from the states made in earlier steps of the block (all bases of the block
    // side by side: chainOthers), then, base after base, itself and the other chain states (chainRun)
    auto chainOthers = [&](int slot, int dj, int jb, int par) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot])), j = jb + dj;
        double f = AUGX_NINF;
        if (s2 >= 0 && j >= 1 && j < n) {
            const double emi = (*lp(&L.sg[par][dj][slot == 7 ? SIG_EIG : SIG_EIN]));
            const int cc = clsAt(j);
            const int nd = (*lp(&L.chNd[slot]));
            // all contributions first (-inf where there is none), then their ln-sum around the largest: no branch per ancestor
            double x[5], m = AUGX_NINF;
            int fin = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int a = (*lp(&L.chDead[slot][k]));
                const double pv = k < nd ? ldsLoadD(&L.ring[(j - 1) & 63][a]) : AUGX_NINF;
                x[k] = pv + (trn(cc, a, s2) + emi);
                if (!(pv > AUGX_NINF)) x[k] = AUGX_NINF; // (a transition of -inf next to a value of +... cannot happen; this guards 0 * inf forms)
                m = x[k] > m ? x[k] : m;
                fin += x[k] > AUGX_NINF;
            }
            f = m;
            if (fin > 1) {
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < 5; k++) sum += x[k] > AUGX_NINF ? exp(x[k] - m) : 0.0;
                f = m + log(sum);
            }
        }
        (*lp(&L.oth[slot][dj])) = f;
    };
    auto chainRun = [&](int slot, int jb, int par) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot]));
        if (s2 < 0) return;
        // what does not change along the block, once: which ancestors are chain states (as a rule only the state itself), the
        // block's signal records and the sums over the other ancestors -- the run itself then goes from register to register
        const int na = (*lp(&L.chNa[slot])), sgi = slot == 7 ? SIG_EIG : SIG_EIN;
        const bool self = (*lp(&L.chSelf[slot])) != 0, otherLive = (*lp(&L.chOther[slot])) != 0;
        double emiR[BLK], othR[BLK];
#pragma unroll
        for (int dj = 0; dj < BLK; dj++) { emiR[dj] = (*lp(&L.sg[par][dj][sgi])); othR[dj] = (*lp(&L.oth[slot][dj])); }
        const double trSelf = multi ? 0.0 : ldsLoadD(&L.lt[s2][s2]);
        double fprev = jb >= 1 ? ldsLoadD(&L.ring[(jb - 1) & 63][s2]) : AUGX_NINF;
        if (!multi && !otherLive && self && jb >= 1 && jb + BLK <= n) {
            // the usual block (one class, the state follows only itself, no piece end inside): straight-line code -- a taken
            // branch costs more than the arithmetic of a base; -inf takes care of itself in the sums and the max
#pragma unroll
            for (int dj = 0; dj < BLK; dj++) {
                const double x = fprev + (trSelf + emiR[dj]), o = othR[dj];
                double f = x > o ? x : o;
                if (o > AUGX_NINF && x > AUGX_NINF) f = lse2(o, x);
                (*lp(&L.ring[(jb + dj) & 63][s2])) = f;
                fprev = f;
            }
            return;
        }
#pragma unroll
        for (int dj = 0; dj < BLK; dj++) {
            const int j = jb + dj;
            if (j >= n) continue;
            if (j < 1) { fprev = ldsLoadD(&L.ring[0][s2]); continue; }
            const int cc = clsAt(j);
            double f = othR[dj];
            if (otherLive) {
#pragma unroll
                for (int ai = 0; ai < 5; ai++) {
                    if (ai < na && (*lp(&L.chLive[slot][ai])) && (*lp(&L.chAnc[slot][ai])) != s2) {
                        const int a = (*lp(&L.chAnc[slot][ai]));
                        const double pv = ldsLoadD(&L.ring[(j - 1) & 63][a]);
                        if (pv > AUGX_NINF) f = lse2(f, pv + (trn(cc, a, s2) + emiR[dj]));
                    }
                }
            }
            if (self && fprev > AUGX_NINF) { // (the call only when there are two terms: it waits for every store in flight)
                const double x = fprev + ((multi ? trn(cc, s2, s2) : trSelf) + emiR[dj]);
                f = f > AUGX_NINF ? lse2(f, x) : x;
            }
            (*lp(&L.ring[j & 63][s2])) = f;
            fprev = f;
        }
    };
    // the cells a run left in the ring go to HBM from another wavefront (u: thread of the rest, slot0: first slot of the run)
    auto chainFlush = [&](int u, int slot0, int nSlots, int jb) __attribute__((always_inline)) {
        if (u < nSlots * BLK) {
            const int slot = slot0 + u / BLK, j = jb + u % BLK, s2 = (*lp(&L.chS[slot]));
            if (s2 >= 0 && j >= 1 && j < n) {
                const double f = ldsLoadD(&L.ring[j & 63][s2]);
                if (f > AUGX_NINF) gp(F)[(int64_t)j * S + s2] = f;
            }
        }
    };
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    uint64_t fpa[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fpl = clock64();
    const bool doProf = B.prof != nullptr && threadIdx.x == 0;
#define FMARK(k) do { if (doProf) { const uint64_t now_ = clock64(); fpa[k] += now_ - fpl; fpl = now_; } } while (0)
#else
#define FMARK(k) do {} while (0)
#endif
    for (int b = 0; b < nBlocks; b++) {
        const int jb = b * BLK, par = b & 1;
        FMARK(0);
        const int64_t gb = gb0 + b;
        // ---- A: fixed-lag states; accumulators of the variable-length cells; the block's signal records; the next block's offsets
        FOR_THREADS(t) {
            TX(curI) = TX(nxtI);
            if (t < BLK * SP) { (*lp(&L.cmax[t / SP][t % SP])) = AUGX_NINF; (*lp(&L.csum[t / SP][t % SP])) = 0ull; }
            if (t == NT - 1 && b + 1 < nBlocks) {
                (*lp(&L.bOff[par ^ 1])) = gp(gBlkOff)[(gb + 1) * 2 + 1]; (*lp(&L.bCnt[par ^ 1][0])) = gp(gBlkCnt)[(gb + 1) * 2 + 1]; (*lp(&L.bCnt[par ^ 1][1])) = gp(gBlkSplit)[(gb + 1) * 3 + 2];
            }
            if (TX(fS2) >= 0) {
                const int s2 = TX(fS2), j = jb + (t - WAVE) % BLK;
                if (j >= 1 && j < n) {
                    const int lag = TX(fLag);
                    const double emi = (*lp(&L.sg[par][(t - WAVE) % BLK][TX(fSg)]));
                    double f = AUGX_NINF;
                    if (j - lag >= 0 && emi > AUGX_NINF) {
                        const int cc = clsAt(j);
#pragma unroll
                        for (int ai = 0; ai < 4; ai++) {
                            if (ai < TX(fNa)) {
                                const int a = fAn[ai][TI];
                                const double pv = at(j - lag, a, jb);
                                if (pv > AUGX_NINF) f = lse2(f, pv + (trn(cc, a, s2) + emi));
                            }
                        }
                    }
                    (*lp(&L.ring[j & 63][s2])) = f;
                    if (f > AUGX_NINF) gp(F)[(int64_t)j * S + s2] = f;
                }
            }
        }
        BLOCK_SYNC();
        FMARK(1);
        // ---- B: geometric intron states, base after base (fed by equalD of the base before and by themselves); the records of
        //      the next block set off
        FOR_THREADS(t) {
            if (b + 1 < nBlocks) TX(nxtI) = (t >= WAVE && (uint32_t)(t - WAVE) < (*lp(&L.bCnt[par ^ 1][0]))) ? ldItem(gItems + (*lp(&L.bOff[par ^ 1])) + (t - WAVE)) : Item{AUGX_NINF, 0u, 0u};
            if (b + 1 < nBlocks && t >= NT - BLK * NSIG) { // (the last wavefronts: the first one has the geometric states)
                const int i = t - (NT - BLK * NSIG), dj = i / NSIG, j = jb + BLK + dj;
                (*lp(&L.sg[par ^ 1][dj][i % NSIG])) = j < n ? gp(gSig)[(int64_t)j * NSIG + i % NSIG] : AUGX_NINF;
            }
            if (t >= WAVE && t - WAVE < 7 * BLK) chainOthers((t - WAVE) / BLK, (t - WAVE) % BLK, jb, par);
        }
        BLOCK_SYNC();
        FMARK(2);
        FOR_THREADS(t) { if (t < 7) chainRun(t, jb, par); }
        BLOCK_SYNC();
        // ---- C / E: candidates of the variable-length states (C: all but RTERMINAL, E: RTERMINAL), three steps each: the largest
        //      candidate of every cell, the sum around it, the cell
        const uint64_t i0 = (*lp(&L.bOff[par]));
        const uint32_t cntAll = (*lp(&L.bCnt[par][0])), cntNonRT = (*lp(&L.bCnt[par][1]));
        auto candValue = [&](const Item &I, int &dj, int &s2) __attribute__((always_inline)) -> double {
            dj = (int)(I.kp >> (KEY_BITS + 6)); s2 = (int)((I.kp >> KEY_BITS) & 63);
            if (!(I.te > AUGX_NINF)) return AUGX_NINF;
            const uint32_t tag = I.src >> 30;
            const int ai = (int)((I.src >> 28) & 3), eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
            double pv;
            if (tag == SRC_COL0) { const int a = (int)(I.src & 0x3Fu); pv = initKind == 0 ? T.ln_init[a] : (a == synch ? 0.0 : AUGX_NINF); }
            else pv = at(eop, tag == SRC_VIG ? igS : (int)(*lp(&L.anc[s2][ai])), jb);
            return pv + I.te;
        };
        auto candidates = [&](uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
            FOR_THREADS(t) {
                TX(cv) = AUGX_NINF;
                if (t >= WAVE && (uint32_t)(t - WAVE) >= lo && (uint32_t)(t - WAVE) < hi) { // this thread's own record: value kept for the next two steps
                    TX(cv) = candValue(TX(curI), TX(cdj), TX(cs2));
                    if (TX(cv) > AUGX_NINF) ldsMaxD(&L.cmax[TX(cdj)][TX(cs2)], TX(cv));
                }
                for (uint32_t it = (lo > (uint32_t)NTW ? lo : (uint32_t)NTW) + (uint32_t)(t - WAVE); t >= WAVE && it < hi; it += NTW) { // (a block with more than NT candidates)
                    int dj, s2;
                    const double v = candValue(ldItem(gItems + i0 + it), dj, s2);
                    if (v > AUGX_NINF) ldsMaxD(&L.cmax[dj][s2], v);
                }
            }
            BLOCK_SYNC();
            FOR_THREADS(t) {
                if (TX(cv) > AUGX_NINF) ldsAddU(&L.csum[TX(cdj)][TX(cs2)], (unsigned long long)(exp(TX(cv) - (*lp(&L.cmax[TX(cdj)][TX(cs2)]))) * FWD_FIX));
                for (uint32_t it = (lo > (uint32_t)NTW ? lo : (uint32_t)NTW) + (uint32_t)(t - WAVE); t >= WAVE && it < hi; it += NTW) {
                    int dj, s2;
                    const double v = candValue(ldItem(gItems + i0 + it), dj, s2);
                    if (v > AUGX_NINF) ldsAddU(&L.csum[dj][s2], (unsigned long long)(exp(v - (*lp(&L.cmax[dj][s2]))) * FWD_FIX));
                }
            }
            BLOCK_SYNC();
        };
        auto cells = [&](bool rt) __attribute__((always_inline)) {
            FOR_THREADS(t) {
                for (int i = t - WAVE; t >= WAVE && i < BLK * SP; i += NTW) {
                    const int dj = i / SP, s2 = i % SP, j = jb + dj;
                    if (j >= 1 && j < n) {
                        if ((*lp(&L.cellKind[s2])) == (rt ? 2 : 1)) {
                            const double f = (*lp(&L.csum[dj][s2])) > 0ull ? (*lp(&L.cmax[dj][s2])) + log((double)(*lp(&L.csum[dj][s2])) / FWD_FIX) : AUGX_NINF;
                            (*lp(&L.ring[j & 63][s2])) = f;
                            if (f > AUGX_NINF) gp(F)[(int64_t)j * S + s2] = f;
                        }
                    }
                }
            }
            BLOCK_SYNC();
        };
        FMARK(3);
        FOR_THREADS(t) { if (t >= WAVE) chainFlush(t - WAVE, 0, 7, jb); }
        candidates(0, cntNonRT);
        FMARK(4);
        cells(false);
        FMARK(5);
        // ---- D: the intergenic state, base after base (fed by itself and by the exon cells of the base before)
        FOR_THREADS(t) { if (t >= WAVE && t - WAVE < BLK) chainOthers(7, t - WAVE, jb, par); }
        BLOCK_SYNC();
        FMARK(6);
        FOR_THREADS(t) { if (t == 0) chainRun(7, jb, par); }
        BLOCK_SYNC();
        FMARK(7);
        FOR_THREADS(t) { if (t >= WAVE) chainFlush(t - WAVE, 7, 1, jb); }
        candidates(cntNonRT, cntAll);
        FMARK(8);
        cells(true);
        FMARK(9);
        // the columns of this block reach HBM before any later block reads them from there: that is 7 blocks away at the earliest
        if ((b & 3) == 3) BLOCK_GLOBAL_SYNC();
        FMARK(10);
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    if (B.prof && threadIdx.x == 0) for (int k = 0; k < 12; k++) B.prof[(int64_t)p * 56 + k] = fpa[k];
#endif
#undef FMARK
    BLOCK_SYNC();
    FOR_THREADS(t) { // ln P(sequence) (reference NAMGene::getSampledPath's option list over the last column, src/namgene.cc:385-392)
        if (t == 0) {
            double tot = AUGX_NINF;
            for (int i = 0; i < S; i++) {
                const double tl = termKind == 0 ? T.ln_term[i] : (i == synch ? 0.0 : AUGX_NINF);
                const double v = (*lp(&L.ring[(n - 1) & 63][i])) + tl;
                if (v > AUGX_NINF) tot = lse2(tot, v);
            }
            B.lnFwd[p] = tot;
        }
    }
}

// =================================================================================================
// after the passes of the segment-parallel trellis: the regions of piece p and the constant each one's stored values are off by
// (SegDesc, BatchView::brkPos / brkOff), the score of the piece in the true frame, the status of its runs
// =================================================================================================
AUGX_KFN void segFinalizePiece(const BatchView &B, int p) {
    const int s0 = B.pieceSeg0[p], K = B.pieceSeg0[p + 1] - s0, n = B.len[p];
    if (K <= 1) {
        B.brkPos[s0] = n - 1; B.brkOff[s0] = 0.0;
        // every addition of the decode is exact only while |ln V| < 2^(53 - AUGX_Q_BITS): a piece that improbable is an error
        if (B.status[p] == 0 && !(fabs(B.lnv[p]) < AUGX_EXACT_LIMIT)) B.status[p] = AUGX_E_RANGE;
        return;
    }
    // frame k = the values segment k computed in pass 1 (frame 0 = the true values).  Fix-up k rewrote the tiles up to segStop[k]
    // in the frame of what lies before it and measured D = (that frame) - (frame behind): every converged run ends a region,
    // and the region after it is off by D more.  A fix-up that gave up was continued by pass 3 (segStop2 / segD2) -- possibly
    // across later segments, whose own records inside the span it redid no longer exist.  A run that reached the end of the
    // piece leaves one last region in the frame it started in.
    double off = 0.0;
    int covered = -1, r = 0; // covered: last tile a continuation has redone so far
    bool abortAny = false;
    const int lastTile = (n + WAVE - 1) / WAVE - 1;
    for (int k = 0; k < K; k++)
        if (B.segStatus[s0 + k]) abortAny = true;
    for (int k = 1; k < K; k++) {
        const int st = B.segStop[s0 + k];
        // the seam this segment's run left: where it stopped, and the offset of what lies behind it.  A fix-up that gave up and
        // its continuation are one run in one frame: its seam is where the continuation stopped
        int endTile = st >= -1 ? st : B.segStop2[s0 + k];
        const double D = st >= -1 ? B.segD[s0 + k] : B.segD2[s0 + k];
        if (st < -1 && endTile < 0) { abortAny = true; endTile = lastTile; } // (never expected: the last pass leaves no fix-up behind)
        if (endTile < covered - 1) continue; // a later continuation ran over this seam: its own offset is relative to what lies behind
        if (st < -1 && endTile > covered) covered = endTile;
        if (endTile >= lastTile) break;     // the run reached the end of the piece: one last region in the frame it started in
        B.brkPos[s0 + r] = (endTile + 1) * WAVE - 1; B.brkOff[s0 + r] = off; r++;
        off = off + D;
    }
    for (; r < K; r++) { B.brkPos[s0 + r] = n - 1; B.brkOff[s0 + r] = off; }
    if (B.status[p] == 0) B.lnv[p] = B.lnv[p] + off;
    if (abortAny) B.status[p] = AUGX_E_HIP;
    if (B.status[p] == 0 && !(fabs(B.lnv[p]) < AUGX_EXACT_LIMIT)) B.status[p] = AUGX_E_RANGE;
}

// =================================================================================================
// K3  back-tracking: follow the back pointers from the best final state (reference NAMGene::getViterbiPath,
// src/namgene.cc:467-506).  Runs of the single-base chain states (igenic, geometric) are skipped 64 bases at
// a time by the whole wavefront and emitted as one merged record.  Records are written 3'->5'.
// =================================================================================================
#ifdef AUGX_EMU
inline int waveCount(const int *flag) { int c = 0; for (int l = 0; l < WAVE; l++) c += flag[l] != 0; return c; }
inline int waveFirstTrue(const int *flag) {
    for (int l = 0; l < WAVE; l++)
        if (flag[l]) return l;
    return WAVE;
}
#else
__device__ inline int waveCount(const int *flag) { return __popcll(__ballot(flag[0] != 0)); }
__device__ inline int waveFirstTrue(const int *flag) {
    unsigned long long m = __ballot(flag[0] != 0);
    return m ? (int)__ffsll((long long)m) - 1 : WAVE;
}
#endif

AUGX_KFN void backtracePiece(const DevTables &T, const BatchView &B, int p) {
    const int n = B.len[p];
    const int64_t o = B.off[p];
    const int64_t po = pathOff(B, p), cap = pathCap(B, p);
    int state = B.finalState[p];
    int base = n - 1;
    int count = 0, nearTies = 0;
    bool overflow = false;
    if (state < 0 || B.status[p] != 0) {
        FOR_LANES(l) { if (l == 0) { B.pathCount[p] = 0; if (B.nearTie) B.nearTie[p] = 0; } }
        return;
    }
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U;
    const bool segmented = B.pieceSeg0[p + 1] - B.pieceSeg0[p] > 1;
    while (base > 0) {
        const int kind = T.kind[state];
        int eop, ai;
        const bool chain = kind == AUGX_K_IGENIC || kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC;
        if (chain) {
            int selfAi = -1;
            for (int i = 0; i < T.n_anc[state]; i++)
                if (T.anc[state][i] == state) selfAi = i;
            int cur = base;
            uint16_t w = BP_NONE;
            int cslot = 0; // chain slot of the state (BatchView::bpChain)
            for (int s2 = 0; s2 < state; s2++)
                if (T.reachable[s2] && (T.kind[s2] == AUGX_K_IGENIC || T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC)) cslot++;
            for (;;) { // find the first base <= cur whose predecessor is not the state itself: 4 bases per lane, 256 per step
                LV(int, flag);
                LV(int, wv);
                LV(int, hit);
                FOR_LANES(l) {
                    LX(flag) = 0; LX(wv) = -1; LX(hit) = 4;
#pragma unroll
                    for (int k = 3; k >= 0; k--) { // (descending k: the hit nearest to cur is kept)
                        const int q = cur - 4 * l - k;
                        const int raw = q >= 1 ? (int)B.bpChain[(o + 1 + q) * 8 + cslot] : -1;
                        const int ww = raw < 0 || raw == 0xFF ? raw : (raw & 0x7F); // (bit 7: the decision of that cell was a near tie)
                        if (q < 1 || ww != selfAi) { LX(flag) = 1; LX(wv) = ww; LX(hit) = k; }
                    }
                }
                int first = waveFirstTrue(flag);
                if (first < WAVE) {
#ifdef AUGX_EMU
                    const int hk = hit[first], hw = wv[first];
#else
                    const int hk = __shfl(hit[0], first, 64), hw = __shfl(wv[0], first, 64);
#endif
                    cur -= 4 * first + hk;
                    w = cur >= 1 ? (uint16_t)hw : BP_NONE;
                    break;
                }
                cur -= 4 * WAVE;
            }
            // bases cur..base are in `state`; base `cur` was entered from another state (or cur < 1: sequence start)
            if (B.nearTie) { // the cells of this run whose decision (stay / come in from another state) was a near tie
                const int lo2 = cur < 1 ? 1 : cur;
                for (int q0 = lo2; q0 <= base; q0 += WAVE) {
                    LV(int, fl);
                    FOR_LANES(l) { const int q = q0 + l; LX(fl) = (q <= base && B.bpChain[(o + 1 + q) * 8 + cslot] != 0xFF && (B.bpChain[(o + 1 + q) * 8 + cslot] & 0x80)) ? 1 : 0; }
                    nearTies += waveCount(fl);
                }
            }
            if (cur < 1) { eop = 0; ai = -1; }
            else { eop = cur - 1; ai = w; }
        } else if ((kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) || kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
            // a variable-length state has no stored back pointer: the arg-max over the candidates of this one (base, state)
            // pair is taken again, from the candidate records of its block and the predecessor values the trellis left in
            // HBM -- the same additions, the same tie-break (larger key), a few hundred candidates per path element
            const int blkSz = B.blk;
            const int64_t gb = o / blkSz + base / blkSz;
            const uint64_t i0 = B.blkOff[gb * 2 + 1];
            const uint32_t cnt = B.blkCnt[gb * 2 + 1], pid = (uint32_t)(((base % blkSz) << 6) | state);
            Best best{AUGX_NINF, -2147483647, -1};
            double runnerUp = AUGX_NINF; // (pass 1, only when near ties are counted: the largest candidate that is not the winner)
            for (int pass = 0; pass < (B.nearTie ? 2 : 1); pass++)
            for (uint32_t c0 = 0; c0 < cnt; c0 += WAVE) {
                LV(double, cv);
                LV(int, ck);
                LV(int, ca);
                FOR_LANES(l) {
                    LX(cv) = AUGX_NINF; LX(ck) = -2147483647; LX(ca) = -1;
                    const uint32_t it = c0 + (uint32_t)l;
                    if (it < cnt) {
                        const Item I = B.items[i0 + it];
                        if ((I.kp >> KEY_BITS) == pid && I.te > AUGX_NINF) {
                            const uint32_t sr = I.src, tag = sr >> 30;
                            const int sel = (sr >> 26) & 3, fr = (sr >> 24) & 3, pay = (int)(sr & 0xFFFFFFu);
                            double pv;
                            // (a piece decoded in segments stores its values region by region up to a constant: frameOff)
                            if (tag == SRC_LIST) {
                                const double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
                                pv = a[(listOff(B, p) + pay) * 3 + fr];
                                if (segmented) {
                                    const int64_t e = listOff(B, p) + pay;
                                    const int pos = sel == 0 ? B.laPos[e] : sel == 1 ? B.lrPos[e] : sel == 2 ? B.ldEnt[e].pos : B.rdEnt[e].pos;
                                    pv = pv + frameOff(B, p, pos);
                                }
                            } else if (tag == SRC_VIG) {
                                pv = B.vig[o + 1 + pay];
                                if (segmented) pv = pv + frameOff(B, p, pay);
                            } else {
                                const int a0 = (int)(sr & 0x3Fu);
                                pv = B.initKind[p] == 0 ? T.ln_init[a0] : (a0 == T.synch ? 0.0 : AUGX_NINF);
                            }
                            LX(cv) = pv + I.te; LX(ck) = (int)(I.kp & KEY_MASK); LX(ca) = (int)((sr >> 28) & 3);
                            if (pass == 1 && LX(ck) == best.key && LX(ca) == best.aux) LX(cv) = AUGX_NINF; // (the winner itself)
                        }
                    }
                }
                const Best b2 = waveArgMax(cv, ck, ca);
                if (pass == 0) { if (better(b2.v, b2.key, best.v, best.key)) best = b2; }
                else if (b2.v > runnerUp) runnerUp = b2.v;
            }
            if (B.nearTie && runnerUp > AUGX_NINF && best.v - runnerUp < AUGX_NEAR_TIE && best.v != runnerUp) nearTies++; // (an exact tie is decided by the reference's own rule, the same in both)
            if (!(best.v > AUGX_NINF)) { overflow = true; break; }
            ai = best.aux;
            eop = best.key - KEY_BIAS;
#ifdef AUGX_EMU
            {   // (while the trellis still stores these back pointers: they must agree)
                const uint16_t w = B.bp[(o + 1 + base) * SP + state];
                if (w != BP_NONE && (ai != (w >> 14) || eop != base - (int)(w & 0x3FFF))) { fprintf(stderr, "emu: recomputed back pointer differs at base %d state %d\n", base, state); abort(); }
            }
#endif
        } else {
            uint16_t w = B.bp[(o + 1 + base) * SP + state];
            if (w == BP_NONE) { overflow = true; break; }
            ai = w;
            int lag = (kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) ? dssWhole
                      : (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) ? assLag : T.dStateLen;
            eop = base - lag;
        }
        if (count >= cap) { overflow = true; break; }
        FOR_LANES(l) {
            if (l == 0) {
                int32_t *r = B.pathRec + (po + count) * 3;
                r[0] = eop + 1; r[1] = base; r[2] = state;
            }
        }
        count++;
        base = eop;
        if (ai < 0 || ai >= T.n_anc[state]) { if (base > 0) overflow = true; break; }
        state = T.anc[state][ai];
    }
    FOR_LANES(l) {
        if (l == 0) {
            B.pathCount[p] = count;
            if (B.nearTie) B.nearTie[p] = nearTies;
            if (overflow) B.status[p] = AUGX_E_HIP;
        }
    }
}

} // namespace dev
} // namespace augx
