// kernels.h -- bodies of the decode kernels, written once for gfx950 (hipcc) and for the lane-loop emulator
// (-DAUGX_EMU, tests only).  See DESIGN.md for the kernel decomposition:
//   K1  prep   : encode, site/stop/base-count prefix scans, GC class, fixed-point content prefix sums, signals
//   K2  trellis: one 64-lane wavefront per piece, position-sequential, V columns in an LDS ring
//   K3  back   : back-pointer chase, one wavefront per piece
//
// Lane discipline: a FOR_LANES block is executed by every lane (concurrently on the device, one after the other
// in the emulator); lanes communicate only through LDS/global memory BETWEEN blocks, separated by WAVE_SYNC().
#pragma once
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
#define FOR_WAVES(w) for (int w = (int)(threadIdx.x >> 6), _oncew = 1; _oncew; _oncew = 0)
#define FOR_WLANES(t, w) for (int t = (int)threadIdx.x, _once = 1; _once; _once = 0)
#define TV(T, name) T name[1]
#define TV2(T, name, K) T name[K][1]
#define TI 0
#define TX(name) name[0]
// workgroup barrier that only orders LDS traffic (no wait for outstanding global stores)
#define BLOCK_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); __asm__ volatile("" ::: "memory"); } while (0)
#define BLOCK_GLOBAL_SYNC() __syncthreads()
#endif
constexpr int NWAVES = 8, NT = NWAVES * WAVE;

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

// =================================================================================================
// K1  prep kernels (one thread per slot unless noted).  g = global slot index.
// =================================================================================================
constexpr int NCNT = 10; // prefix-count fields: a c g t | atg | ag(LA) | ac(LR) | gt(LD) | ct(RD) | reverse stop codons
constexpr int CNT_ATG = 4, CNT_LA = 5, CNT_LR = 6, CNT_LD = 7, CNT_RD = 8, CNT_RS = 9;

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

// site flags and stop codons -> terms of the count / max scans
AUGX_HD void k1SiteTerms(const DevTables &T, const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    Piece P;
    P.t = &T; P.n = B.len[p]; P.c = 0; P.o = o; P.code = B.code + o + 1; P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
    uint64_t cnt[NCNT], ns[6];
    for (int i = 0; i < NCNT; i++) cnt[i] = 0;
    for (int i = 0; i < 6; i++) ns[i] = 0;
    if (q >= 0 && q < P.n) {
    int c = P.b(q);
    if (c < 4) cnt[c] = 1;
    // start codon with positive probability at q (a of atg)
    if (q < P.n - 2) { int pn = P.pat(q, 3); if (pn >= 0 && T.ln_startcodon[pn] > AUGX_NINF) cnt[CNT_ATG] = 1; }
    if (P.possASS(q - T.Ae)) cnt[CNT_LA] = 1;                       // longass may end at q   (src/intronmodel.cc:705)
    if (P.possRDSS(q - T.Ds)) cnt[CNT_LR] = 1;                      // rlongdss may end at q  (:709)
    if (P.possDSS(q - T.De - 2 + 1)) cnt[CNT_LD] = 1;               // longdss may end at q   (:693)
    if (P.possRASS(q - T.U - T.As - 2 + 1)) cnt[CNT_RD] = 1;        // rlongass may end at q  (:713)
    if (q <= P.n - 3) {
        if (P.isStop(q)) ns[q % 3] = (uint64_t)q + 1;
        if (P.isRCStop(q)) { ns[3 + q % 3] = (uint64_t)q + 1; cnt[CNT_RS] = 1; }
    }
    }
    for (int i = 0; i < NCNT; i++) B.cnt[fidx(g, i, NCNT)] = cnt[i];
    for (int i = 0; i < 6; i++) B.nsm[fidx(g, i, 6)] = ns[i];
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
    double cnt[4];
    for (int i = 0; i < 4; i++) cnt[i] = (double)(B.cnt[fidx(o + s + win, i, NCNT)] - B.cnt[fidx(o + s, i, NCNT)]);
    return nearestClass(T, cnt);
}

// fixed-point terms of the 20 content prefix fields
AUGX_HD void k1FxTerms(const DevTables &T, const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    uint64_t out[NFX];
    for (int i = 0; i < NFX; i++) out[i] = 0;
    Piece P;
    P.t = &T; P.n = B.len[p]; P.c = B.cls[p]; P.o = o; P.code = B.code + o + 1; P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
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
    const double *inE = T.in_emi + (int64_t)c * NP;
    out[FX_INF] = toFx(pn >= 0 ? inE[pn] : T.ln_quarter);
    int rn2 = (q + k < P.n) ? rn : -1;
    out[FX_INR] = toFx(rn2 >= 0 ? inE[rn2] : T.ln_quarter);
    }
    for (int i = 0; i < NFX; i++) B.fx[fidx(g, i, NFX)] = out[i];
}

// per-base signal record + end-gate mask of the variable-length states
constexpr int NSITE = 4;
AUGX_HD void k1Signals(const DevTables &T, const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    double *sg = B.sig + g * NSIG;
    for (int i = 0; i < NSIG; i++) sg[i] = AUGX_NINF;
    B.gate[g] = 0;
    int32_t *st = B.site + g * NSITE;
    for (int i = 0; i < NSITE; i++) st[i] = -1;
    if (q < 0 || q >= B.len[p] || B.cls[p] < 0) return;
    Piece P = makePiece(T, B, p);
    const int dssWhole = T.Ds + 2 + T.De, assWhole = T.As + 2 + T.Ae;
    sg[SIG_EIG] = q >= 1 ? eIg(P, q) : AUGX_NINF;
    sg[SIG_EIN] = eIn(P, q);
    // fixed-length intron states ending at q: gate && emission (reference src/intronmodel.cc:690-717,861-923)
    if (q - dssWhole >= 0 && P.possDSS(q - T.De - 2 + 1)) sg[SIG_DSSF] = dssProb(P, q - dssWhole + 1, true);
    if (q - dssWhole >= 0 && P.possRDSS(q - T.Ds)) sg[SIG_DSSR] = dssProb(P, q - dssWhole + 1, false);
    if (q - assWhole - T.U >= 0 && P.possASS(q - T.Ae)) sg[SIG_ASSF] = assProb(P, q - assWhole - T.U + 1, true);
    if (q - assWhole - T.U >= 0 && P.possRASS(q - T.U - T.As - 2 + 1)) sg[SIG_ASSR] = assProb(P, q - assWhole - T.U + 1, false);
    sg[SIG_TISF] = tisFwd(P, q);
    sg[SIG_TISR] = tisRev(P, q);
    sg[SIG_STOPF] = exEndPart(P, AUGX_K_TERMINAL, 0, q, AUGX_NINF); // ln P(stop codon ending at q), -inf if none
    // list index of the site ending at q (prefix count - 1), -1 if q is not such a site
    uint64_t cn[NCNT], cp[NCNT];
    for (int i = CNT_ATG; i < NCNT; i++) { cn[i] = B.cnt[fidx(g, i, NCNT)]; cp[i] = B.cnt[fidx(g - 1, i, NCNT)]; }
    const int64_t lo = listOff(B, p);
    for (int i = 0; i < NSITE; i++)
        st[i] = cn[CNT_LA + i] != cp[CNT_LA + i] ? (int32_t)cn[CNT_LA + i] - 1 : -1;
    // the candidate lists are indexed by site; positions are known here, the trellis fills in the values
    if (st[0] >= 0) B.laPos[lo + st[0]] = q;
    if (st[1] >= 0) B.lrPos[lo + st[1]] = q;
    if (st[2] >= 0) B.ldPos[lo + st[2]] = q;
    if (st[3] >= 0) B.rdPos[lo + st[3]] = q;
    if (cn[CNT_ATG] != cp[CNT_ATG]) B.atgPos[lo + cn[CNT_ATG] - 1] = q;
    // emission of the equalD states ending at q (reference IntronModel::seqProb, src/intronmodel.cc:1087-1107)
    if (q - T.dStateLen >= 0) sg[SIG_EQD] = P.seg(FX_INF, q - T.dStateLen + 1, q);
    // end gates
    uint64_t gate = 0;
    if (q >= 1)
        for (int s = 0; s < T.S; s++) {
            if (!T.reachable[s]) continue;
            int kind = T.kind[s];
            bool open = false;
            if (kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) {
                ExGeom gm = exGeom(T, kind);
                double endP = exEndPart(P, kind, T.win[s], q, sg[SIG_TISR]);
                int right = q + gm.baseOffset - gm.ipeo;
                open = endP > AUGX_NINF && right >= 0;
            } else if (kind == AUGX_K_LESSD)
                open = lessDGate(P, true, q);
            else if (kind == AUGX_K_RLESSD)
                open = lessDGate(P, false, q);
            if (open) gate |= 1ull << s;
        }
    B.gate[g] = gate;
}

// candidate-side constants of the list entries (everything a candidate contributes that does not depend on Viterbi
// values).  Fast-path evaluation in the trellis combines them with end-side constants; the arithmetic is exactly that
// of exNotEndPart (same prefix differences, same order of additions).
AUGX_HD void k1SiteConsts(const DevTables &T, const BatchView &B, int64_t g) {
    int p = B.chunkPiece[g / CHUNK];
    int64_t o = B.off[p];
    int q = (int)(g - o - 1);
    if (q < 0 || q >= B.len[p]) return;
    double *pr = B.plsR + g * 3;
    pr[0] = pr[1] = pr[2] = AUGX_NINF;
    if (B.cls[p] < 0) return;
    Piece P = makePiece(T, B, p);
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
    if (cn[CNT_LA] != cp[CNT_LA]) { // forward acceptor candidate ending (as longass state) at q: exon inner part starts at bs = q+1
        int64_t idx = lo + (int64_t)cn[CNT_LA] - 1;
        int bs = q + 1, eos = bs + k - 1, pn = P.pat(bs, k);
        for (int a = 0; a < 3; a++) {
            B.laPls[idx * 3 + a] = k == 0 ? 0.0 : plsK(pn, mod3(eos + a));
            B.laFx[idx * 3 + a] = fxv(eos, (0 * 3 + a) * 3 + 0);
        }
    }
    if (cn[CNT_LR] != cp[CNT_LR]) { // reverse donor candidate
        int64_t idx = lo + (int64_t)cn[CNT_LR] - 1;
        int bs = q + 1, eot = bs + T.Le - 1;
        for (int a = 0; a < 3; a++) {
            const int fb = (1 * 3 + a) * 3;
            B.lrEt[idx * 3 + a] = (double)(int64_t)(fxv(eot, fb + 2) - fxv(bs - 1, fb + 2)) * AUGX_FX_INV;
            if (eot < bs) B.lrEt[idx * 3 + a] = 0.0;
            B.lrFx[idx * 3 + a] = fxv(eot, fb + 0);
        }
    }
    if (cn[CNT_LD] != cp[CNT_LD]) B.ldFx[lo + (int64_t)cn[CNT_LD] - 1] = fxv(q, FX_INF);
    if (cn[CNT_RD] != cp[CNT_RD]) B.rdFx[lo + (int64_t)cn[CNT_RD] - 1] = fxv(q, FX_INR);
    if (cn[CNT_ATG] != cp[CNT_ATG]) { // start codon at q: bs = q+3, frame phase a = (-q) mod 3
        int64_t idx = lo + (int64_t)cn[CNT_ATG] - 1;
        int bs = q + 3, eos = bs + k - 1, eoi = eos + T.Li, a = mod3(-q);
        const int fb = (0 * 3 + a) * 3;
        B.atgD[idx * 3 + 0] = tisFwd(P, q);
        B.atgD[idx * 3 + 1] = k == 0 ? 0.0 : plsK(P.pat(bs, k), mod3(eos + a));
        B.atgD[idx * 3 + 2] = eoi > eos ? (double)(int64_t)(fxv(eoi, fb + 1) - fxv(eos, fb + 1)) * AUGX_FX_INV : 0.0;
        B.atgFx[idx] = fxv(eoi, fb + 0);
    }
    if (cn[CNT_RS] != cp[CNT_RS]) { // reverse stop codon at q..q+2: bs = q+3
        int64_t idx = lo + (int64_t)cn[CNT_RS] - 1;
        B.rsPos[idx] = q;
        B.rsBegin[idx] = (P.b(q) == 3 && P.b(q + 1) == 3) ? T.ln_stop_ochre : (P.b(q) == 1) ? T.ln_stop_amber : T.ln_stop_opal;
        for (int a = 0; a < 3; a++) B.rsFx[idx * 3 + a] = fxv(q + 2, (1 * 3 + a) * 3 + 0);
    }
}

// =================================================================================================
// K2  trellis: one wavefront per piece
// =================================================================================================
struct VarDesc {
    int kind, win, nList, extra, total, listSel; // listSel: 0 LA, 1 LR, 2 LD, 3 RD, 4 ATG, 5 single reverse-stop candidate
    int64_t i1;                                   // one past the newest list entry (piece-local index)
    int eob, right, fOR, startMin;                // exon geometry
    int eobi, cod0, cod1, cod2;                   // short intron: end of the biological intron, spliced-codon bases
    double endP;
    ExGeom g;
    // end-side constants of the fast candidate evaluation (see k1SiteConsts)
    int a;                  // phase of the content prefix fields
    uint64_t eFx;           // content prefix at the end-side boundary
    double eD0, plsEnd;     // end-side content term (exon-terminal fwd / initial-content rev), reverse-strand ln P_ls
    int lenSel;             // length distribution of this exon type: 0 single, 1 initial, 2 internal, 3 terminal
};

constexpr int BLK = 8;          // bases per block: smaller than every lag except the lag-1 chain states
constexpr int MAXPAIR = WAVE;   // gated (base, state) pairs handled per round (MAXPAIR / NWAVES per wavefront)
constexpr int MAXPW = MAXPAIR / NWAVES;

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
AUGX_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }

// per-state constants of the variable-length states (LDS copy: no table walks in the inner loops)
struct VarConst {
    int kind, win, nanc, anc[4], ancWin[4];
    double tr[4];
    ExGeom g;
};

struct TrellisLds {
    double ring[WAVE][SP];          // ln V of the last 64 columns, [j & 63][state]
    double eqPrev[WAVE][6];         // predecessor cells of the equalD states for the bases of the tile (lag dStateLen)
    double sig[WAVE][NSIG];         // tile of signal records for bases j0..j0+63
    uint64_t gate[WAVE];
    int32_t site[WAVE][NSITE];
    uint16_t bp[WAVE][SP];
    // windows over recent bases (served from LDS; older data falls back to HBM)
    uint8_t codew[CODE_WIN];
    uint32_t nsw[NS_WIN * 6];
    uint32_t cntw[CNT_WIN][6];      // site counts (ATG, LA, LR, LD, RD, RS) at bases <= q
    double lenw[4][LEN_WIN];        // ln(3 P(len)) of single / initial / internal / terminal exons, len < LEN_WIN
    double leniw[LENI_MAX];         // ln P(len) of short introns
    double vigw[VIG_WIN];           // igenic column
    uint64_t fxw[FX_WIN][NFX];      // content prefix sums of bases [j0-64, j0+64)
    double plsRw[FX_WIN][3];
    int32_t lcPos[4][LIST_WIN];     // newest LIST_WIN entries of the four splice-site candidate lists
    double lcVal[4][LIST_WIN][3];   //   Viterbi values of the three frames
    double lcC[2][LIST_WIN][3];     //   LA: ln P_ls; LR: exon-terminal content
    uint64_t lcFx[4][LIST_WIN][3];  //   content prefix at the candidate-side boundary (LD/RD: [0] only)
    int32_t aPos[ATG_WIN];          // newest start-codon entries
    double aD[ATG_WIN][3];
    uint64_t aFx[ATG_WIN];
    int chainIds[8], nChain;         // the lag-1 chain states (igenic + geometric introns)
    double chV[8][BLK][5];          // step 3 scratch: candidate value per (chain state, base, ancestor)
    double chPs[8][BLK], chTe[8][BLK], chB[8][BLK][2];
    int chA[8][BLK][2];
    VarConst vc[SP];
    VarDesc desc[MAXPAIR];          // descriptors of the gated (base, state) pairs of the current round
    int pairJ[MAXPAIR], pairS[MAXPAIR];
    double itVal[NWAVES][WAVE];     // one chunk of evaluated candidates per wavefront
    int itKey[NWAVES][WAVE], itAux[NWAVES][WAVE];
};

AUGX_HD int longRow(const DevTables &T, int s) {
    int kind = T.kind[s];
    if (kind == AUGX_K_LONGDSS) return T.win[s];
    if (kind == AUGX_K_RLONGASS) return 3 + T.win[s];
    return -1;
}
AUGX_HD double lnT(const DevTables &T, int c, int a, int s) { return T.ln_trans[((int64_t)c * T.S + a) * T.S + s]; }
AUGX_HD uint16_t bpFixed(int ai) { return (uint16_t)ai; }
AUGX_HD uint16_t bpVar(int ai, int dist) { return (uint16_t)((ai << 14) | (dist & 0x3FFF)); }

#if defined(AUGX_PROF) && !defined(AUGX_EMU)
#define PROF_MARK(X, sec) do { uint64_t now_ = clock64(); (X).pacc[sec] += now_ - (X).plast; (X).plast = now_; } while (0)
#else
#define PROF_MARK(X, sec) do {} while (0)
#endif
struct TrellisCtx {
#if defined(AUGX_PROF) && !defined(AUGX_EMU)
    uint64_t pacc[16], plast;
#endif
    const DevTables &T;
    const BatchView &B;
    TrellisLds &L;
    int p;
    Piece P;
    int64_t o;      // slot offset of the piece
    int64_t lo;     // list offset
    int n, c, S;
    int kwLo, kwHi; // cnt window
    int fwLo, fwHi; // content-prefix window
    int atgHi;      // newest start-codon entry in the LDS cache
    int vigLo;      // igenic ring holds bases > vigLo (and <= the newest chain base)
    int listHi0, listHi1, listHi2, listHi3;  // newest published entry of each splice-site list (piece-local index), -1 none
    AUGX_HD int listHi(int sel) const { return sel == 0 ? listHi0 : sel == 1 ? listHi1 : sel == 2 ? listHi2 : listHi3; }
    AUGX_HD TrellisCtx(const DevTables &t, const BatchView &b, TrellisLds &l, int pp) : T(t), B(b), L(l), p(pp) {
        P = makePiece(T, B, p);
        o = B.off[p];
        lo = listOff(B, p);
        n = P.n; c = P.c; S = T.S;
        kwLo = kwHi = 0; fwLo = fwHi = 0; atgHi = -1; vigLo = 0x7fffffff;
        listHi0 = listHi1 = listHi2 = listHi3 = -1;
    }
    AUGX_HD uint64_t cntAt(int q, int f) const { // number of sites of field f at bases <= q (q may be -1)
        if (q < 0) return 0;
        if (q > n - 1) q = n - 1;
        if (f >= CNT_ATG && q >= kwLo && q < kwHi) return L.cntw[q & (CNT_WIN - 1)][f - CNT_ATG];
        return B.cnt[fidx(o + 1 + q, f, NCNT)];
    }
    AUGX_HD int listPos(int sel, int64_t li) const { // sel: 0 LA, 1 LR, 2 LD, 3 RD; li piece-local index
        if (li <= listHi(sel) && li > listHi(sel) - LIST_WIN) return L.lcPos[sel][li & (LIST_WIN - 1)];
        const int32_t *a = sel == 0 ? B.laPos : sel == 1 ? B.lrPos : sel == 2 ? B.ldPos : B.rdPos;
        return a[lo + li];
    }
    AUGX_HD double listVal(int sel, int64_t li, int f) const {
        if (li <= listHi(sel) && li > listHi(sel) - LIST_WIN) return L.lcVal[sel][li & (LIST_WIN - 1)][f];
        const double *a = sel == 0 ? B.laVal : sel == 1 ? B.lrVal : sel == 2 ? B.ldVal : B.rdVal;
        return a[(lo + li) * 3 + f];
    }
    AUGX_HD double vigAt(int eop) const { return eop > vigLo ? L.vigw[eop & (VIG_WIN - 1)] : B.vig[o + 1 + eop]; }
    AUGX_HD bool listCached(int sel, int64_t li) const { return li <= listHi(sel) && li > listHi(sel) - LIST_WIN; }
    AUGX_HD double listC(int sel, int64_t li, int a) const { // LA: ln P_ls, LR: exon-terminal content
        if (listCached(sel, li)) return L.lcC[sel][li & (LIST_WIN - 1)][a];
        return (sel == 0 ? B.laPls : B.lrEt)[(lo + li) * 3 + a];
    }
    AUGX_HD uint64_t listFx(int sel, int64_t li, int a) const {
        if (listCached(sel, li)) return L.lcFx[sel][li & (LIST_WIN - 1)][a];
        if (sel == 0) return B.laFx[(lo + li) * 3 + a];
        if (sel == 1) return B.lrFx[(lo + li) * 3 + a];
        return (sel == 2 ? B.ldFx : B.rdFx)[lo + li];
    }
    AUGX_HD uint64_t fxAt(int q, int f) const { // content prefix field f up to and including base q (q < 0: empty)
        if (q < 0) return 0;
        if (q >= fwLo && q < fwHi) return L.fxw[q & (FX_WIN - 1)][f];
        return B.fx[fidx(o + 1 + q, f, NFX)];
    }
    AUGX_HD double lenAt(int sel, int len) const {
        if (len < LEN_WIN) return L.lenw[sel][len];
        return (sel == 0 ? T.len_single : sel == 1 ? T.len_initial : sel == 2 ? T.len_internal : T.len_terminal)[len];
    }
    AUGX_HD double lenIAt(int len) const { return T.d < LENI_MAX ? L.leniw[len] : T.len_intron[len]; }
    AUGX_HD double plsRAt(int q, int fr) const {
        if (q >= fwLo && q < fwHi) return L.plsRw[q & (FX_WIN - 1)][fr];
        return B.plsR[(o + 1 + q) * 3 + fr];
    }
};

// -------------------------------------------------------------------------------------------------
// variable-length states (coding exons, short introns) whose end gate is open.
// All gated (base, state) pairs of a block are evaluated together: one lane per pair builds a descriptor
// (candidate range + end-side constants), then all candidates are spread over the 64 lanes, and finally one
// lane per pair reduces its candidates.  The formulas are those of the reference loops
// (exon: src/exonmodel.cc:1059-1132; lessD: src/intronmodel.cc:585-629); the tie-break "larger key wins" is
// the reference's descending loop with strict '>'.
// -------------------------------------------------------------------------------------------------
AUGX_KFN void varDescribe(const TrellisCtx &X, int s, int j, VarDesc &D) {
    const DevTables &T = X.T;
    const Piece &P = X.P;
    const VarConst &VC = X.L.vc[s];
    const int kind = VC.kind, win = VC.win, n = X.n;
    D.kind = kind; D.win = win; D.nList = 0; D.extra = 0; D.total = 0; D.listSel = 0; D.i1 = 0;
    D.eob = D.right = D.fOR = D.startMin = 0; D.eobi = 0; D.cod0 = D.cod1 = D.cod2 = 4; D.endP = AUGX_NINF;
    D.g = VC.g;
    D.a = 0; D.eFx = 0; D.eD0 = 0.0; D.plsEnd = 0.0; D.lenSel = 2;
    if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
        const bool fwd = kind == AUGX_K_LESSD;
        const int f = win;
        const int eobi = fwd ? j + T.U + T.As + 2 : j + T.De + 2;
        const bool haveRight = eobi < n - 2;
        D.eobi = eobi;
        if (fwd) {
            if (f == 1) { D.cod1 = haveRight ? P.b(eobi + 1) : 4; D.cod2 = haveRight ? P.b(eobi + 2) : 4; }
            if (f == 2) { D.cod2 = haveRight ? P.b(eobi + 1) : 4; }
        } else {
            if (f == 0) D.cod0 = (haveRight && P.b(eobi + 1) <= 3) ? 3 - P.b(eobi + 1) : 4;
            if (f == 1) {
                D.cod0 = (haveRight && P.b(eobi + 2) <= 3) ? 3 - P.b(eobi + 2) : 4;
                D.cod1 = (haveRight && P.b(eobi + 1) <= 3) ? 3 - P.b(eobi + 1) : 4;
            }
        }
        int left = j - T.dStateLen;
        if (left < 0) left = 0;
        const int fld = fwd ? CNT_LD : CNT_RD;
        const int64_t i0 = (int64_t)X.cntAt(left - 1, fld);
        D.i1 = (int64_t)X.cntAt(j - 1, fld);
        D.listSel = fwd ? 2 : 3;
        D.nList = (int)(D.i1 - i0);
        D.extra = left == 0 ? 1 : 0; // eop = 0 reads column 0 (initial probabilities); not a splice site, so not listed
        D.total = D.nList + D.extra;
        D.endP = 0.0;
        D.eFx = X.fxAt(j, fwd ? FX_INF : FX_INR);
        return;
    }
    const ExEnd e = exEnd(P, kind, win, j, D.g);
    D.eob = e.eob; D.right = e.right; D.fOR = e.fOR; D.startMin = e.startMin;
    D.endP = (kind == AUGX_K_SINGLE || kind == AUGX_K_TERMINAL) ? X.L.sig[j & 63][SIG_STOPF]
             : (kind == AUGX_K_RSINGLE || kind == AUGX_K_RINITIAL) ? X.L.sig[j & 63][SIG_TISR] : 0.0; // gate is open
    if (!(D.endP > AUGX_NINF) || e.right < 0 || e.startMax < e.startMin) return;
    {   // end-side constants of the fast candidate evaluation
        const int k = T.k, right = e.right;
        const int a = D.g.fwd ? mod3(e.fOR - right) : mod3(e.fOR + right);
        const int fb = ((D.g.fwd ? 0 : 1) * 3 + a) * 3;
        D.a = a;
        switch (kind) {
        case AUGX_K_INTERNAL: case AUGX_K_INITIAL:
            D.eFx = X.fxAt(right - T.Le, fb + 0);
            D.eD0 = T.Le > 0 ? (double)(int64_t)(X.fxAt(right, fb + 2) - X.fxAt(right - T.Le, fb + 2)) * AUGX_FX_INV : 0.0;
            D.lenSel = kind == AUGX_K_INTERNAL ? 2 : 1;
            break;
        case AUGX_K_TERMINAL: case AUGX_K_SINGLE:
            D.eFx = X.fxAt(right, fb + 0);
            D.lenSel = kind == AUGX_K_TERMINAL ? 3 : 0;
            break;
        default: {
            const int boip = right - (k - 1);
            D.plsEnd = (k > 0 && boip >= 0) ? X.plsRAt(right, mod3(e.fOR + right - boip)) : 0.0;
            if (kind == AUGX_K_RINTERNAL || kind == AUGX_K_RTERMINAL) {
                D.eFx = X.fxAt(boip - 1, fb + 0);
                D.lenSel = kind == AUGX_K_RINTERNAL ? 2 : 3;
            } else {
                const int boi = boip - T.Li;
                D.eFx = X.fxAt(boi - 1, fb + 0);
                D.eD0 = T.Li > 0 ? (double)(int64_t)(X.fxAt(boip - 1, fb + 1) - X.fxAt(boi - 1, fb + 1)) * AUGX_FX_INV : 0.0;
                D.lenSel = kind == AUGX_K_RINITIAL ? 1 : 0;
            }
        }
        }
    }
    if (kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL) { // start codons with bob in [startMin-3, startMax-3]
        const int64_t i0 = (int64_t)X.cntAt(e.startMin - 3 - 1, CNT_ATG);
        D.i1 = (int64_t)X.cntAt(e.startMax - 3, CNT_ATG);
        D.listSel = 4;
        D.nList = (int)(D.i1 - i0);
    } else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL) {
        D.listSel = 5; // single candidate bs = ORFleft+2 (src/exonmodel.cc:1044-1045)
        D.nList = 1;
    } else {
        const int fld = D.g.fwd ? CNT_LA : CNT_LR; // eop = bs - 1 in [startMin-1, startMax-1]
        const int64_t i0 = (int64_t)X.cntAt(e.startMin - 2, fld);
        D.i1 = (int64_t)X.cntAt(e.startMax - 1, fld);
        D.listSel = D.g.fwd ? 0 : 1;
        D.nList = (int)(D.i1 - i0);
        D.extra = e.startMin == 0 ? 1 : 0; // bs = 0: left-truncated exon, predecessor column 0
    }
    D.total = D.nList + D.extra;
}

// candidate number idx (0 = newest) of the state described by D: value, tie-break key, predecessor index
AUGX_KFN void varEvalItem(const TrellisCtx &X, int s, int j, const VarDesc &D, int idx, double &val, int &key, int &aux) {
    const DevTables &T = X.T;
    const BatchView &B = X.B;
    const Piece &P = X.P;
    const VarConst &VC = X.L.vc[s];
    const int n = X.n, kind = D.kind, win = D.win;
    val = AUGX_NINF; key = -2147483647; aux = -1;
    auto col0 = [&](int a) { return (B.initKind[X.p] == 0) ? T.ln_init[a] : (a == T.synch ? 0.0 : AUGX_NINF); };
    if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
        const bool fwd = kind == AUGX_K_LESSD;
        const int f = win;
        const int a = VC.anc[0];
        int eop;
        double pv;
        if (idx < D.nList) {
            int64_t li = D.i1 - 1 - idx;
            eop = X.listPos(D.listSel, li);
            pv = X.listVal(D.listSel, li, f);
        } else { eop = 0; pv = col0(a); }
        if (!(pv > AUGX_NINF)) return;
        int begin = eop + 1;
        int bobi = fwd ? begin - T.De - 2 : begin - (T.U + T.As + 2);
        if (bobi >= 0 && !(fwd ? P.possDSS(bobi) : P.possRASS(bobi))) return;
        bool spliced = fwd ? (f != 0) : (f != 2);
        if (spliced && bobi > 1) {
            int c0 = D.cod0, c1 = D.cod1, c2 = D.cod2;
            if (fwd) {
                if (f == 1) c0 = P.b(bobi - 1);
                else { c0 = P.b(bobi - 2); c1 = P.b(bobi - 1); }
            } else {
                if (f == 0) { c1 = P.b(bobi - 1) <= 3 ? 3 - P.b(bobi - 1) : 4; c2 = P.b(bobi - 2) <= 3 ? 3 - P.b(bobi - 2) : 4; }
                else c2 = P.b(bobi - 1) <= 3 ? 3 - P.b(bobi - 1) : 4;
            }
            if (stopCodon3(c0, c1, c2)) return;
        }
        int intronLength = D.eobi - bobi + 1;
        if (intronLength > T.d) return;
        double restSeq = idx < D.nList ? (double)(int64_t)(D.eFx - X.listFx(D.listSel, D.i1 - 1 - idx, 0)) * AUGX_FX_INV
                                       : P.seg(fwd ? FX_INF : FX_INR, begin, j);
        double emi = X.lenIAt(intronLength) + restSeq;
        if (!(emi > AUGX_NINF)) return;
        val = pv + (VC.tr[0] + emi);
        key = eop; aux = 0;
        return;
    }
    if (D.listSel >= 4) { // predecessor is the igenic state
        const int a = VC.anc[0];
        int bs;
        double tisF = AUGX_NINF;
        double cPls = 0.0, cInit = 0.0;
        uint64_t cFxA = 0;
        if (D.listSel == 4) {
            const int64_t ai2 = D.i1 - 1 - idx;
            if (ai2 <= X.atgHi && ai2 > X.atgHi - ATG_WIN) {
                const int sl = (int)(ai2 & (ATG_WIN - 1));
                bs = X.L.aPos[sl] + 3; tisF = X.L.aD[sl][0]; cPls = X.L.aD[sl][1]; cInit = X.L.aD[sl][2]; cFxA = X.L.aFx[sl];
            } else {
                bs = B.atgPos[X.lo + ai2] + 3;
                tisF = B.atgD[(X.lo + ai2) * 3 + 0]; cPls = B.atgD[(X.lo + ai2) * 3 + 1]; cInit = B.atgD[(X.lo + ai2) * 3 + 2];
                cFxA = B.atgFx[X.lo + ai2];
            }
        } else
            bs = D.startMin;
        int eop = bs - D.g.bpl - 1;
        // eop == j reads the igenic cell of the CURRENT column (already final: the reference fills states in index
        // order and igenic is state 0); later columns do not exist yet
        if (!(eop < n && eop <= j)) return;
        double pv = eop <= 0 ? col0(a) : X.vigAt(eop);
        if (!(pv > AUGX_NINF)) return;
        double nep;
        const int m = D.right - bs, k = T.k;
        const int bob = bs - D.g.ipo, len = D.eob - bob + 1;
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
                double seg1 = (double)(int64_t)(D.eFx - B.rsFx[(X.lo + ri) * 3 + D.a]) * AUGX_FX_INV;
                double inner = kind == AUGX_K_RTERMINAL ? seg1 : (D.eD0 + seg1);
                nep = (B.rsBegin[X.lo + ri] + (D.plsEnd + inner)) + lenPart;
            }
        }
        if (!fast) nep = exNotEndPart(P, kind, win, bs, D.right, D.fOR, D.g, tisF);
        if (!(nep > AUGX_NINF)) return;
        double te = (VC.tr[0] + D.endP) + nep;
        val = pv + te;
        key = bs; aux = 0;
        return;
    }
    // predecessors are the three longass_f (forward) or rlongdss_f (reverse) states, listed per splice site
    const bool fwd = D.g.fwd;
    int eop;
    int64_t li = -1;
    if (idx < D.nList) { li = D.i1 - 1 - idx; eop = X.listPos(D.listSel, li); }
    else eop = -1;
    int bs = eop + 1;
    int bob = bs - D.g.ipo, len = D.eob - bob + 1;
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
                    double seg1 = (double)(int64_t)(D.eFx - X.listFx(0, li, D.a)) * AUGX_FX_INV;
                    double inner = kind == AUGX_K_TERMINAL ? seg1 : (seg1 + D.eD0);
                    nep = (0.0 + (X.listC(0, li, D.a) + inner)) + lenPart;
                }
            } else {
                const int boip = D.right - (k - 1), eot = bs + T.Le - 1, boi = boip - T.Li;
                const bool ok = kind == AUGX_K_RINTERNAL ? (eot < boip) : (boi >= bs && eot < boi);
                if (ok) {
                    fast = true;
                    double lenPart = (len >= 1 && len <= T.max_exon_len && (kind == AUGX_K_RINTERNAL || len > 2)) ? X.lenAt(D.lenSel, len) : AUGX_NINF;
                    if (!(lenPart > AUGX_NINF)) return;
                    double seg1 = (double)(int64_t)(D.eFx - X.listFx(1, li, D.a)) * AUGX_FX_INV;
                    double cEt = X.listC(1, li, D.a);
                    double inner = kind == AUGX_K_RINTERNAL ? (seg1 + cEt) : ((D.eD0 + seg1) + cEt);
                    nep = (0.0 + (D.plsEnd + inner)) + lenPart;
                }
            }
        }
        if (!fast) nep = exNotEndPart(P, kind, win, bs, D.right, D.fOR, D.g, AUGX_NINF);
    }
    if (!(nep > AUGX_NINF)) return;
    for (int ai = 0; ai < VC.nanc; ai++) {
        if (win != mod3(fwd ? VC.ancWin[ai] + len : VC.ancWin[ai] - len)) continue;
        double pv = li >= 0 ? X.listVal(D.listSel, li, VC.ancWin[ai]) : col0(VC.anc[ai]);
        if (!(pv > AUGX_NINF)) continue;
        double v2 = pv + ((VC.tr[ai] + D.endP) + nep);
        if (better(v2, bs, val, key)) { val = v2; key = bs; aux = ai; }
    }
}

// gated variable-length states of the bases [jb, jb+BLK) that belong to `mask`
// gated variable-length states of the bases [jb, jb+BLK) that belong to `mask`; wavefront w takes every NWAVES-th pair
AUGX_KFN void trellisVarBlock(TrellisCtx &X, int jb, uint64_t mask, int w, int stride) {
    // stride == NWAVES: the pairs of `mask` are dealt round-robin to the wavefronts; stride == 1: this wavefront takes all
    const BatchView &B = X.B;
    TrellisLds &L = X.L;
    const int n = X.n, S = X.S;
    uint64_t g[BLK];   // only ever indexed by fully unrolled loops: stays in registers
    int off[BLK + 1];
    off[0] = 0;
#pragma unroll
    for (int dj = 0; dj < BLK; dj++) {
        int j = jb + dj;
        g[dj] = (j >= 1 && j < n) ? (L.gate[j & 63] & mask) : 0;
        off[dj + 1] = off[dj] + popc64(g[dj]);
    }
    const int allPairs = off[BLK];
    const int first0 = stride == 1 ? 0 : w, perRound = MAXPW * stride;
    for (int done = 0; done < allPairs; done += perRound) {
        const int roundPairs = allPairs - done < perRound ? allPairs - done : perRound;
        const int nPairs = (roundPairs - first0 + stride - 1) / stride; // pairs done + first0, done + first0 + stride, ...
        if (nPairs <= 0) continue;
        TV(int, pj);
        TV(int, ps);
        TV(int, tot);
        FOR_WLANES(t, w) { // one lane per pair: locate the pair, build its descriptor
            const int l = t & 63;
            TX(pj) = -1; TX(ps) = 0; TX(tot) = 0;
            if (l < nPairs) {
                int want = done + l * stride + first0, dj = 0, first = 0;
                uint64_t gg = 0;
#pragma unroll
                for (int d2 = 0; d2 < BLK; d2++)
                    if (off[d2] <= want && want < off[d2 + 1]) { dj = d2; gg = g[d2]; first = off[d2]; }
                for (int k = want - first; k > 0; k--) gg &= gg - 1;
                const int s2 = __builtin_ctzll(gg | (1ull << 63));
                TX(pj) = jb + dj; TX(ps) = s2;
                L.pairJ[w * MAXPW + l] = jb + dj; L.pairS[w * MAXPW + l] = s2;
                varDescribe(X, s2, jb + dj, L.desc[w * MAXPW + l]);
                TX(tot) = (B.dbgFlags & 2) ? 0 : L.desc[w * MAXPW + l].total;
            }
        }
        WAVE_SYNC();
        PROF_MARK(X, 2);
        TV(int, ibase); // inclusive prefix of the candidate counts
        FOR_WLANES(t, w) { TX(ibase) = TX(tot); }
        waveInclScan(ibase, w);
        const int totalItems = waveRead(ibase, w, WAVE - 1);
        TV(double, rbv);
        TV(int, rbk);
        TV(int, rba);
        FOR_WLANES(t, w) { TX(rbv) = AUGX_NINF; TX(rbk) = -2147483647; TX(rba) = -1; }
        PROF_MARK(X, 3);
        for (int base = 0; base < totalItems; base += WAVE) {
            TV(int, myPair);
            TV(int, myFirst);
            FOR_WLANES(t, w) { TX(myPair) = 0; TX(myFirst) = 0; }
            for (int q = 0; q < nPairs; q++) { // pair of item `base + lane`: last pair whose first item is <= it
                const int first = q == 0 ? 0 : waveRead(ibase, w, q - 1); // all lanes active here (cross-lane read)
                FOR_WLANES(t, w) { if (first <= base + (t & 63)) { TX(myPair) = q; TX(myFirst) = first; } }
            }
            PROF_MARK(X, 4);
            FOR_WLANES(t, w) { // evaluate one candidate per lane
                const int l = t & 63;
                int it = base + l;
                if (it < totalItems) {
                    const int q = w * MAXPW + TX(myPair);
                    const int first = TX(myFirst);
                    double v; int k2, a2;
                    varEvalItem(X, L.pairS[q], L.pairJ[q], L.desc[q], it - first, v, k2, a2);
                    L.itVal[w][l] = v; L.itKey[w][l] = k2; L.itAux[w][l] = a2;
                }
            }
            WAVE_SYNC();
            PROF_MARK(X, 5);
            FOR_WLANES(t, w) { // the lane of each pair folds the candidates of this chunk that belong to it
                const int l = t & 63;
                if (l < nPairs) {
                    int lo2 = (TX(ibase) - TX(tot)) - base, hi2 = TX(ibase) - base;
                    if (lo2 < 0) lo2 = 0;
                    if (hi2 > WAVE) hi2 = WAVE;
                    for (int q = lo2; q < hi2; q++)
                        if (better(L.itVal[w][q], L.itKey[w][q], TX(rbv), TX(rbk))) { TX(rbv) = L.itVal[w][q]; TX(rbk) = L.itKey[w][q]; TX(rba) = L.itAux[w][q]; }
                }
            }
            WAVE_SYNC();
            PROF_MARK(X, 6);
        }
        FOR_WLANES(t, w) {
            const int l = t & 63;
            if (l < nPairs) {
                const int j = TX(pj), s2 = TX(ps);
                uint16_t bp = BP_NONE;
                if (TX(rbv) > AUGX_NINF) {
                    const int kind = L.vc[s2].kind;
                    int eop = (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) ? TX(rbk) : TX(rbk) - L.vc[s2].g.bpl - 1;
                    bp = bpVar(TX(rba), j - eop);
                }
                L.ring[j & 63][s2] = TX(rbv);
                L.bp[j & 63][s2] = bp;
                if (B.cells) B.cells[(X.o + 1 + j) * S + s2] = TX(rbv);
            }
        }
        WAVE_SYNC();
        PROF_MARK(X, 7);
    }
}

// One workgroup of NWAVES wavefronts per piece.  Thread t: st = t & 63 is "its" state, q4 = t >> 6 its quarter.
AUGX_KFN void trellisPiece(const DevTables &T, const BatchView &B, TrellisLds &L, int p) {
    TrellisCtx X(T, B, L, p);
    const int n = X.n, S = X.S, c = X.c;
    const int64_t o = X.o;
    if (c < 0) { // multi-class piece: not decoded by this version
        FOR_THREADS(t) { if (t == 0) { B.status[p] = AUGX_E_UNSUPPORTED; B.lnv[p] = AUGX_NINF; B.finalState[p] = -1; } }
        return;
    }
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U, dL = T.dStateLen;
    {   // ---- a piece without a single nucleotide is all intergenic (reference src/namgene.cc:205-226)
        uint64_t nuc = 0;
        for (int i = 0; i < 4; i++) nuc += B.cnt[fidx(o + n, i, NCNT)];
        if (nuc == 0) {
            const int sy = T.synch;
            int selfAi = 0;
            for (int ai = 0; ai < T.n_anc[sy]; ai++)
                if (T.anc[sy][ai] == sy) selfAi = ai;
            FOR_THREADS(t) {
                for (int q = t; q < n; q += NT)
                    for (int s2 = 0; s2 < SP; s2++) B.bp[(o + 1 + q) * SP + s2] = (s2 == sy && q >= 1) ? bpFixed(selfAi) : BP_NONE;
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
    // ---- per-thread constants of state st = t & 63.  cls: 0 chain (lag 1), 1 fixed lag, 2 variable, 3 RTERMINAL, -1 off
    TV(int, hCls);
    TV(int, hLag);
    TV(int, hSig);
    TV(int, hLong);
    TV(int, hNanc);
    TV2(int, hAnc, 5);
    TV2(double, hTr, 5);
    TV(int, hLrow);
    TV(int, hList);
    TV(int, hFrame);
    TV(int, hIgenic);
    // each wavefront specialises on one family of variable-length states (its instruction stream then only runs the
    // branches of that family): 0 internal/terminal, 1 initial/single, 2 short introns, 3 reverse exons but RTERMINAL
    uint64_t maskVar = 0, maskRT = 0;
    uint64_t maskW[NWAVES];
#pragma unroll
    for (int i = 0; i < NWAVES; i++) maskW[i] = 0;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int kind = T.kind[s2];
        const uint64_t bit = 1ull << s2;
        int fam = -1;
        switch (kind) {
        case AUGX_K_RTERMINAL: maskRT |= bit; break;
        case AUGX_K_INTERNAL: fam = 0; break;
        case AUGX_K_TERMINAL: fam = 1; break;
        case AUGX_K_INITIAL: fam = 2; break;
        case AUGX_K_SINGLE: case AUGX_K_RSINGLE: fam = 3; break;
        case AUGX_K_LESSD: fam = 4; break;
        case AUGX_K_RLESSD: fam = 5; break;
        case AUGX_K_RINTERNAL: fam = 6; break;
        case AUGX_K_RINITIAL: fam = 7; break;
        default: break;
        }
#pragma unroll
        for (int i = 0; i < NWAVES; i++) if (fam >= 0 && i == fam % NWAVES) maskW[i] |= bit;
    }
#pragma unroll
    for (int i = 0; i < NWAVES; i++) maskVar |= maskW[i];
    FOR_THREADS(t) {
        const int l = t & 63;
        TX(hCls) = -1; TX(hLag) = -1; TX(hSig) = 0; TX(hLong) = 0; TX(hNanc) = 0; TX(hLrow) = -1; TX(hList) = -1; TX(hFrame) = 0; TX(hIgenic) = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { hAnc[i][TI] = 0; hTr[i][TI] = AUGX_NINF; }
        if (l < S && T.reachable[l]) {
            const int kind = T.kind[l];
            TX(hCls) = kind == AUGX_K_RTERMINAL ? 3 : 2;
            switch (kind) {
            case AUGX_K_IGENIC: TX(hLag) = 1; TX(hSig) = SIG_EIG; TX(hCls) = 0; break;
            case AUGX_K_GEOMETRIC: case AUGX_K_RGEOMETRIC: TX(hLag) = 1; TX(hSig) = SIG_EIN; TX(hCls) = 0; break;
            case AUGX_K_LONGDSS: TX(hLag) = dssWhole; TX(hSig) = SIG_DSSF; TX(hCls) = 1; break;
            case AUGX_K_RLONGDSS: TX(hLag) = dssWhole; TX(hSig) = SIG_DSSR; TX(hCls) = 1; break;
            case AUGX_K_LONGASS: TX(hLag) = assLag; TX(hSig) = SIG_ASSF; TX(hCls) = 1; break;
            case AUGX_K_RLONGASS: TX(hLag) = assLag; TX(hSig) = SIG_ASSR; TX(hCls) = 1; break;
            case AUGX_K_EQUALD: case AUGX_K_REQUALD: TX(hLag) = dL; TX(hSig) = SIG_EQD; TX(hLong) = dL >= WAVE; TX(hCls) = 1; break;
            default: break;
            }
            if (TX(hLag) > 0) {
                TX(hNanc) = T.n_anc[l] < 5 ? T.n_anc[l] : 5;
#pragma unroll
                for (int ai = 0; ai < 5; ai++) {
                    if (ai >= TX(hNanc)) continue;
                    int a = T.anc[l][ai];
                    hAnc[ai][TI] = TX(hLong) ? longRow(T, a) : a;
                    hTr[ai][TI] = lnT(T, c, a, l);
                }
            }
            TX(hLrow) = longRow(T, l);
            TX(hFrame) = T.win[l];
            TX(hIgenic) = kind == AUGX_K_IGENIC;
            TX(hList) = kind == AUGX_K_LONGASS ? 0 : kind == AUGX_K_RLONGDSS ? 1 : kind == AUGX_K_LONGDSS ? 2 : kind == AUGX_K_RLONGASS ? 3 : -1;
            if (t < WAVE) { // LDS copy of the constants of the variable-length states
                VarConst &VC = L.vc[l];
                VC.kind = kind; VC.win = T.win[l]; VC.nanc = T.n_anc[l] < 4 ? T.n_anc[l] : 4;
                for (int ai = 0; ai < 4; ai++) {
                    int a = ai < VC.nanc ? T.anc[l][ai] : 0;
                    VC.anc[ai] = a; VC.ancWin[ai] = T.win[a]; VC.tr[ai] = ai < VC.nanc ? lnT(T, c, a, l) : AUGX_NINF;
                }
                VC.g = exGeom(T, (kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) ? kind : AUGX_K_INTERNAL);
            }
        }
    }
    // ---- step-3 work items: one (chain state, ancestor, base of the block) triple per thread
    FOR_THREADS(t) {
        if (t == 0) {
            int nc = 0;
            for (int s2 = 0; s2 < S && nc < 8; s2++)
                if (T.reachable[s2] && (T.kind[s2] == AUGX_K_IGENIC || T.kind[s2] == AUGX_K_GEOMETRIC || T.kind[s2] == AUGX_K_RGEOMETRIC)) L.chainIds[nc++] = s2;
            L.nChain = nc;
        }
    }
    BLOCK_SYNC();
    TV(int, ciCs);   // chain-state slot, -1: no item
    TV(int, ciAi);
    TV(int, ciDj);
    TV(int, ciAnc);
    TV(int, ciSig);
    TV(int, ciSelf);
    TV(double, ciTr);
    const int nChain = L.nChain;
    FOR_THREADS(t) {
        TX(ciCs) = -1; TX(ciAi) = 0; TX(ciDj) = 0; TX(ciAnc) = 0; TX(ciSig) = 0; TX(ciSelf) = 0; TX(ciTr) = AUGX_NINF;
        int base = 0;
        for (int cs = 0; cs < nChain; cs++) {
            const int s2 = L.chainIds[cs];
            const int na = T.n_anc[s2] < 5 ? T.n_anc[s2] : 5;
            if (t >= base && t < base + na * BLK) {
                const int ai = (t - base) / BLK;
                const int a = T.anc[s2][ai];
                TX(ciCs) = cs; TX(ciAi) = ai; TX(ciDj) = (t - base) % BLK; TX(ciAnc) = a; TX(ciSelf) = a == s2;
                TX(ciSig) = T.kind[s2] == AUGX_K_IGENIC ? SIG_EIG : SIG_EIN;
                TX(ciTr) = lnT(T, c, a, s2);
            }
            base += na * BLK;
        }
    }
    FOR_THREADS(t) { // length distributions into LDS
        for (int i = t; i < LEN_WIN; i += NT) {
            const bool in = i <= T.max_exon_len;
            L.lenw[0][i] = in ? T.len_single[i] : AUGX_NINF; L.lenw[1][i] = in ? T.len_initial[i] : AUGX_NINF;
            L.lenw[2][i] = in ? T.len_internal[i] : AUGX_NINF; L.lenw[3][i] = in ? T.len_terminal[i] : AUGX_NINF;
        }
        for (int i = t; i < LENI_MAX; i += NT) L.leniw[i] = i <= T.d ? T.len_intron[i] : AUGX_NINF;
    }
    // ---- column 0 = initial probabilities (reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150)
    FOR_THREADS(t) {
        if (t < SP) {
            const int l = t;
            double v = AUGX_NINF;
            if (l < S) v = B.initKind[p] == 0 ? T.ln_init[l] : (l == T.synch ? 0.0 : AUGX_NINF);
            L.ring[0][l] = v;
            L.bp[0][l] = BP_NONE;
            if (l < S) {
                if (TX(hLrow) >= 0) B.longV[(o + 1) * 6 + TX(hLrow)] = v;
                if (B.cells) B.cells[(o + 1) * S + l] = v;
                if (TX(hIgenic)) { B.vig[o + 1] = v; L.vigw[0] = v; }
            }
        }
    }
    X.vigLo = -1;
    X.P.wcode = L.codew;
    X.P.wns = L.nsw;
    BLOCK_SYNC();
#if defined(AUGX_PROF) && !defined(AUGX_EMU)
    for (int i = 0; i < 16; i++) X.pacc[i] = 0;
    X.plast = clock64();
#endif
    for (int j0 = 0; j0 < n; j0 += WAVE) {
        BLOCK_GLOBAL_SYNC(); // everything this workgroup stored to HBM so far is visible to its own later loads
        // ---- load the tile of per-base records for bases j0..j0+63 and advance the LDS windows; the four
        //      wavefronts share the work by record type (every row of 64 loads is coalesced)
        FOR_THREADS(t) {
            const int l = t & 63, q4 = (B.dbgFlags & 64) ? 99 : (t >> 6);
            const int q = j0 + l;
            const int64_t gq = o + 1 + q;
            if (q4 == 0) {
                for (int i = 0; i < NSIG; i++) L.sig[l][i] = B.sig[gq * NSIG + i];
                L.gate[l] = B.gate[gq];
                for (int i = 0; i < NSITE; i++) L.site[l][i] = B.site[gq * NSITE + i];
                if (q != 0) for (int s2 = 0; s2 < SP; s2++) L.bp[l][s2] = BP_NONE;
            } else if (q4 == 1) {
                // code window: bases [j0+64, j0+128) (first tile: also [0, 64))
                for (int r = (j0 == 0 ? 0 : 1); r < 2; r++) {
                    int qq = j0 + r * WAVE + l;
                    if (qq < n) L.codew[qq & (CODE_WIN - 1)] = B.code[o + 1 + qq];
                }
                // stop tables: bases [j0+16, j0+80) (first tile: also [0, 16))
                for (int r = (j0 == 0 ? 0 : 1); r < 2; r++) {
                    int qq = r == 0 ? l : j0 + 16 + l;
                    if (r == 0 && l >= 16) continue;
                    if (qq < n) for (int f = 0; f < 6; f++) L.nsw[(qq & (NS_WIN - 1)) * 6 + f] = (uint32_t)B.nsm[fidx(o + 1 + qq, f, 6)];
                }
                // site counts: bases [j0+64, j0+128) (first tile: also [0, 64))
                for (int r = (j0 == 0 ? 0 : 1); r < 2; r++) {
                    int qq = j0 + r * WAVE + l;
                    if (qq < n) for (int f = 0; f < 6; f++) L.cntw[qq & (CNT_WIN - 1)][f] = (uint32_t)B.cnt[fidx(o + 1 + qq, CNT_ATG + f, NCNT)];
                }
            } else if (q4 == 2) {
                // content prefix sums and reverse P_ls terms of the bases of this tile
                if (q < n) {
                    for (int f = 0; f < NFX; f++) L.fxw[q & (FX_WIN - 1)][f] = B.fx[fidx(gq, f, NFX)];
                    for (int f = 0; f < 3; f++) L.plsRw[q & (FX_WIN - 1)][f] = B.plsR[gq * 3 + f];
                }
            } else if (q4 == 3) {
                // predecessor cells of the equalD states (lag dStateLen >= 64: written long ago by this workgroup)
                for (int f = 0; f < 6; f++) L.eqPrev[l][f] = (dL >= WAVE && q - dL >= 0 && q < n) ? B.longV[(gq - dL) * 6 + f] : AUGX_NINF;
                // candidate-side constants of the list entries whose site lies in this tile
                if (q < n) {
                    for (int sel = 0; sel < 4; sel++) {
                        int si = B.site[gq * NSITE + sel];
                        if (si < 0) continue;
                        const int sl = si & (LIST_WIN - 1);
                        L.lcPos[sel][sl] = q;
                        for (int a = 0; a < 3; a++) {
                            if (sel == 0) { L.lcC[0][sl][a] = B.laPls[(X.lo + si) * 3 + a]; L.lcFx[0][sl][a] = B.laFx[(X.lo + si) * 3 + a]; }
                            if (sel == 1) { L.lcC[1][sl][a] = B.lrEt[(X.lo + si) * 3 + a]; L.lcFx[1][sl][a] = B.lrFx[(X.lo + si) * 3 + a]; }
                        }
                        if (sel == 2) L.lcFx[2][sl][0] = B.ldFx[X.lo + si];
                        if (sel == 3) L.lcFx[3][sl][0] = B.rdFx[X.lo + si];
                    }
                    // start codons of this tile
                    uint32_t ac = (uint32_t)B.cnt[fidx(gq, CNT_ATG, NCNT)], ap = (uint32_t)B.cnt[fidx(gq - 1, CNT_ATG, NCNT)];
                    if (ac != ap) {
                        const int ai2 = (int)ac - 1, sl = ai2 & (ATG_WIN - 1);
                        L.aPos[sl] = q;
                        for (int a = 0; a < 3; a++) L.aD[sl][a] = B.atgD[(X.lo + ai2) * 3 + a];
                        L.aFx[sl] = B.atgFx[X.lo + ai2];
                    }
                }
            }
        }
        BLOCK_SYNC();
        {
            int hi = j0 + 2 * WAVE < n ? j0 + 2 * WAVE : n;
            X.P.wcHi = hi; X.P.wcLo = hi - CODE_WIN > 0 ? hi - CODE_WIN : 0;
            X.kwHi = hi; X.kwLo = hi - CNT_WIN > 0 ? hi - CNT_WIN : 0;
            int hn = j0 + 80 < n ? j0 + 80 : n;
            X.P.wnHi = hn; X.P.wnLo = hn - NS_WIN > 0 ? hn - NS_WIN : 0;
            int hf = j0 + WAVE < n ? j0 + WAVE : n;
            X.fwHi = hf; X.fwLo = hf - FX_WIN > 0 ? hf - FX_WIN : 0;
            // list entries of sites up to the end of this tile are in the LDS caches (values follow as the trellis advances)
            X.listHi0 = (int)X.cntAt(hf - 1, CNT_LA) - 1;
            X.listHi1 = (int)X.cntAt(hf - 1, CNT_LR) - 1;
            X.listHi2 = (int)X.cntAt(hf - 1, CNT_LD) - 1;
            X.listHi3 = (int)X.cntAt(hf - 1, CNT_RD) - 1;
            X.atgHi = (int)X.cntAt(hf - 1, CNT_ATG) - 1;
        }
        PROF_MARK(X, 0);
        for (int jb = j0; jb < j0 + WAVE && jb < n; jb += BLK) {
            // ---- step 1: fixed-length states with lag > BLK: thread = (state, quarter of the block)
            constexpr int CPT = BLK / NWAVES; // cells per thread
            FOR_THREADS(t) {
                const int l = t & 63, q4 = t >> 6;
                if (TX(hCls) == 1 && !(B.dbgFlags & 16)) {
                    const int lag = TX(hLag), nanc = TX(hNanc);
                    double emi[CPT], pv0[CPT], pv1[CPT];
#pragma unroll
                    for (int d = 0; d < CPT; d++) { // independent LDS loads first
                        const int j = jb + q4 * CPT + d, jp = j - lag;
                        emi[d] = L.sig[j & 63][TX(hSig)];
                        pv0[d] = AUGX_NINF; pv1[d] = AUGX_NINF;
                        if (jp >= 0) {
                            if (TX(hLong)) { pv0[d] = L.eqPrev[j & 63][hAnc[0][TI]]; if (nanc > 1) pv1[d] = L.eqPrev[j & 63][hAnc[1][TI]]; }
                            else { pv0[d] = L.ring[jp & 63][hAnc[0][TI]]; if (nanc > 1) pv1[d] = L.ring[jp & 63][hAnc[1][TI]]; }
                        }
                    }
#pragma unroll
                    for (int d = 0; d < CPT; d++) {
                        const int j = jb + q4 * CPT + d;
                        if (j < 1 || j >= n) continue;
                        double best = AUGX_NINF;
                        uint16_t bp = BP_NONE;
                        if (j - lag >= 0 && emi[d] > AUGX_NINF) {
                            if (pv0[d] > AUGX_NINF) { best = pv0[d] + (hTr[0][TI] + emi[d]); bp = bpFixed(0); }
                            if (nanc > 1 && pv1[d] > AUGX_NINF) {
                                double v2 = pv1[d] + (hTr[1][TI] + emi[d]);
                                if (v2 > best) { best = v2; bp = bpFixed(1); }
                            }
                        }
                        L.ring[j & 63][l] = best;
                        L.bp[j & 63][l] = bp;
                        if (TX(hLrow) >= 0) B.longV[(o + 1 + j) * 6 + TX(hLrow)] = best;
                        if (B.cells) B.cells[(o + 1 + j) * S + l] = best;
                        if (TX(hList) >= 0) {
                            int si = L.site[j & 63][TX(hList)];
                            if (si >= 0) {
                                double *lval = TX(hList) == 0 ? B.laVal : TX(hList) == 1 ? B.lrVal : TX(hList) == 2 ? B.ldVal : B.rdVal;
                                lval[(X.lo + si) * 3 + TX(hFrame)] = best;
                                L.lcVal[TX(hList)][si & (LIST_WIN - 1)][TX(hFrame)] = best;
                            }
                        }
                    }
                } else if (TX(hCls) >= 2 || (l < SP && TX(hCls) < 0)) {
                    // cells of variable-length states are absent unless their gate is open and a candidate survives
#pragma unroll
                    for (int d = 0; d < CPT; d++) {
                        const int j = jb + q4 * CPT + d;
                        if (j < 1 || j >= n) continue;
                        L.ring[j & 63][l] = AUGX_NINF;
                        if (B.cells && l < S) B.cells[(o + 1 + j) * S + l] = AUGX_NINF;
                    }
                }
            }
            BLOCK_SYNC();
            PROF_MARK(X, 1);
            // ---- step 2: variable-length states (all but RTERMINAL): they depend on fixed-state cells (just published)
            //      and on igenic cells at least BLK bases back.  Pairs are dealt round-robin to the wavefronts.
            uint64_t anyVar = 0, anyRT = 0;
            for (int dj = 0; dj < BLK; dj++) {
                int j = jb + dj;
                if (j >= 1 && j < n) { uint64_t gg = L.gate[j & 63]; anyVar |= gg & maskVar; anyRT |= gg & maskRT; }
            }
            // (HBM re-reads only touch data at least one tile old -- the LDS caches cover 64 sites / 512 bases -- and
            //  the store queue is drained at every tile boundary, so no wait is needed here)
            PROF_MARK(X, 12);
            if (anyVar && !(B.dbgFlags & 1)) {
                FOR_WAVES(w) {
                    uint64_t mw = 0;
#pragma unroll
                    for (int i = 0; i < NWAVES; i++) if (i == w) mw = maskW[i];
                    trellisVarBlock(X, jb, mw, w, 1);
                }
                PROF_MARK(X, 13);
                BLOCK_SYNC();
                PROF_MARK(X, 8);
            }
            // ---- step 3: the lag-1 chain states (igenic, geometric introns).  Three short phases:
            //   3a  one thread per (state, ancestor, base): candidate value  V[j-1][a] + (t(a->s) + e_s(j))
            //   3b  one thread per (state, base): best ancestor before / after the state itself, in ascending ancestor
            //       order with strict '>' (reference src/igenicmodel.cc:247-255, src/intronmodel.cc:757-786)
            //   3c  one thread per state: the sequential part, two additions and two compares per base
            if (!(B.dbgFlags & 8)) {
            FOR_THREADS(t) {
                if (TX(ciCs) >= 0) {
                    const int j = jb + TX(ciDj);
                    const bool valid = j >= 1 && j < n;
                    const double emi = valid ? L.sig[j & 63][TX(ciSig)] : AUGX_NINF;
                    const double pv = L.ring[(j - 1) & 63][TX(ciAnc)];
                    const double te = TX(ciTr) + emi;
                    L.chV[TX(ciCs)][TX(ciDj)][TX(ciAi)] = TX(ciSelf) ? AUGX_NINF : pv + te;
                    if (TX(ciSelf)) { L.chPs[TX(ciCs)][TX(ciDj)] = pv; L.chTe[TX(ciCs)][TX(ciDj)] = te; }
                }
            }
            BLOCK_SYNC();
            FOR_THREADS(t) {
                if (t < nChain * BLK) {
                    const int cs = t / BLK, dj = t % BLK, s2 = L.chainIds[cs];
                    const int na = L.vc[s2].nanc; // (chain states have at most 5 ancestors; vc keeps 4, igenic's 5th read below)
                    (void)na;
                    double bB = AUGX_NINF, bA = AUGX_NINF;
                    int aB = -1, aA = -1;
                    bool seenSelf = false;
                    const int nanc = T.n_anc[s2] < 5 ? T.n_anc[s2] : 5;
#pragma unroll
                    for (int ai = 0; ai < 5; ai++) {
                        if (ai >= nanc) continue;
                        const bool self = T.anc[s2][ai] == s2;
                        const double v = L.chV[cs][dj][ai];
                        if (self) seenSelf = true;
                        else if (!seenSelf) { if (v > bB) { bB = v; aB = ai; } }
                        else { if (v > bA) { bA = v; aA = ai; } }
                    }
                    L.chB[cs][dj][0] = bB; L.chB[cs][dj][1] = bA;
                    L.chA[cs][dj][0] = aB; L.chA[cs][dj][1] = aA;
                }
            }
            BLOCK_SYNC();
            FOR_THREADS(t) {
                if (t < nChain) {
                    const int cs = t, l = L.chainIds[cs];
                    int selfAi = 5;
                    const int nanc = T.n_anc[l] < 5 ? T.n_anc[l] : 5;
#pragma unroll
                    for (int ai = 0; ai < 5; ai++) if (ai < nanc && T.anc[l][ai] == l) selfAi = ai;
                    double bB[BLK], bA[BLK], te[BLK], ps[BLK];
                    int aB[BLK], aA[BLK];
#pragma unroll
                    for (int dj = 0; dj < BLK; dj++) {
                        bB[dj] = L.chB[cs][dj][0]; bA[dj] = L.chB[cs][dj][1]; aB[dj] = L.chA[cs][dj][0]; aA[dj] = L.chA[cs][dj][1];
                        te[dj] = L.chTe[cs][dj]; ps[dj] = L.chPs[cs][dj];
                    }
                    double res[BLK];
                    int rai[BLK];
                    double prev = AUGX_NINF;
#pragma unroll
                    for (int dj = 0; dj < BLK; dj++) {
                        const int j = jb + dj;
                        const double p0 = (dj == 0 || j - 1 < 1) ? ps[dj] : prev;
                        const double vs = p0 + te[dj];
                        double best = bB[dj];
                        int bai = aB[dj];
                        if (vs > best) { best = vs; bai = selfAi; }
                        if (bA[dj] > best) { best = bA[dj]; bai = aA[dj]; }
                        res[dj] = best; rai[dj] = bai;
                        prev = best;
                    }
                    const bool isIg = T.kind[l] == AUGX_K_IGENIC;
#pragma unroll
                    for (int dj = 0; dj < BLK; dj++) {
                        const int j = jb + dj;
                        if (j < 1 || j >= n) continue;
                        L.ring[j & 63][l] = res[dj];
                        L.bp[j & 63][l] = res[dj] > AUGX_NINF ? bpFixed(rai[dj]) : BP_NONE;
                        if (B.cells) B.cells[(o + 1 + j) * S + l] = res[dj];
                        if (isIg) { B.vig[o + 1 + j] = res[dj]; L.vigw[j & (VIG_WIN - 1)] = res[dj]; }
                    }
                }
            }
            }
            {
                int jl = jb + BLK - 1 < n - 1 ? jb + BLK - 1 : n - 1;
                X.vigLo = jl - VIG_WIN > -1 ? jl - VIG_WIN : -1;
            }
            BLOCK_SYNC();
            PROF_MARK(X, 9);
            // ---- step 4: RTERMINAL exons (their single candidate may start at an igenic cell of this very block)
            if (anyRT && !(B.dbgFlags & 1)) {
                FOR_WAVES(w) { trellisVarBlock(X, jb, maskRT, w, NWAVES); }
                BLOCK_SYNC();
                PROF_MARK(X, 10);
            }
        }
        // ---- flush the back-pointer tile
        FOR_THREADS(t) {
            const int l = t & 63, q4 = t >> 6;
            for (int r = q4; r < WAVE && !(B.dbgFlags & 32); r += NWAVES) {
                int q = j0 + r;
                if (q < n && l < SP) B.bp[(o + 1 + q) * SP + l] = L.bp[r][l];
            }
        }
        BLOCK_SYNC();
        PROF_MARK(X, 11);
    }
#if defined(AUGX_PROF) && !defined(AUGX_EMU)
    if ((threadIdx.x & 63) == 0 && B.prof)
        for (int i = 0; i < 16; i++) B.prof[((int64_t)p * NWAVES + (threadIdx.x >> 6)) * 16 + i] = X.pacc[i];
#endif
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
            B.status[p] = state >= 0 ? 0 : AUGX_E_NOPATH;
        }
    }
}

// =================================================================================================
// K3  back-tracking: follow the back pointers from the best final state (reference NAMGene::getViterbiPath,
// src/namgene.cc:467-506).  Runs of the single-base chain states (igenic, geometric) are skipped 64 bases at
// a time by the whole wavefront and emitted as one merged record.  Records are written 3'->5'.
// =================================================================================================
#ifdef AUGX_EMU
inline int waveFirstTrue(const int *flag) {
    for (int l = 0; l < WAVE; l++)
        if (flag[l]) return l;
    return WAVE;
}
#else
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
    int count = 0;
    bool overflow = false;
    if (state < 0 || B.status[p] != 0) {
        FOR_LANES(l) { if (l == 0) B.pathCount[p] = 0; }
        return;
    }
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U;
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
            for (;;) { // find the first base <= cur whose predecessor is not the state itself
                LV(int, flag);
                LV(int, wv);
                FOR_LANES(l) {
                    int q = cur - l;
                    int ww = q >= 1 ? (int)B.bp[(o + 1 + q) * SP + state] : -1;
                    LX(wv) = ww;
                    LX(flag) = (q < 1) || ww != selfAi;
                }
                int first = waveFirstTrue(flag);
                if (first < WAVE) {
                    cur -= first;
#ifdef AUGX_EMU
                    w = cur >= 1 ? (uint16_t)wv[first] : BP_NONE;
#else
                    w = cur >= 1 ? (uint16_t)__shfl(wv[0], first, 64) : BP_NONE;
#endif
                    break;
                }
                cur -= WAVE;
            }
            // bases cur..base are in `state`; base `cur` was entered from another state (or cur < 1: sequence start)
            if (cur < 1) { eop = 0; ai = -1; }
            else { eop = cur - 1; ai = w; }
        } else if ((kind >= AUGX_K_SINGLE && kind <= AUGX_K_RTERMINAL) || kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
            uint16_t w = B.bp[(o + 1 + base) * SP + state];
            ai = w >> 14;
            eop = base - (int)(w & 0x3FFF);
            if (w == BP_NONE) { overflow = true; break; }
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
            if (overflow) B.status[p] = AUGX_E_HIP;
        }
    }
}

} // namespace dev
} // namespace augx
