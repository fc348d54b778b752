// densev.h -- the Viterbi pass of the dense kernels (dense.h), software-pipelined over the blocks of a piece.
//
// densePiece<BLK, 0> (dense.h) evaluates the candidates of a block inside the block: site record -> length term and predecessor
// cell -> LDS atomic, three dependent trips to HBM / L2 while the whole workgroup waits at the block's first barrier (14 k of the
// 25 k cycles of a block of 4 bases, profiles/r03_utr_*).  Nothing of that depends on a cell of the block itself except for the few
// candidates whose predecessor ends inside the previous block.  Here the candidates of block b+1 -- the records of kCand and the
// units of the UTR exon descriptors alike -- are evaluated WHILE block b runs, one level of loads per stage of block b (the
// barriers of the stages wait for LDS traffic only, global loads stay in flight across them).  The kernel has no register to spare
// (256 VGPRs, 8 wavefronts): a loaded value that waits in a register gets spilled, and a spill waits for the load.  So the loads go
// from HBM STRAIGHT INTO LDS (`global_load_lds_dword / _dwordx4`, gfx950: lane i of a wavefront lands at base + i * size), into
// landing pads of the issuing wavefront, and the wavefront waits for them (vmcnt) two stages later, when they have long arrived:
//     stage 1 of block b   issue: item records, site records of block b+1; the long-lag predecessor of its fixed-lag states;
//                          offsets / counts / signal records of block b+2 into their stage buffers
//     stage 3              records -> predecessor end; issue: predecessor cells (end < first base of block b: final; a cell still
//                          in the ring is copied pad to pad), length terms; the descriptors of block b+2
//     stage 5              value -> LDS atomic max into the accumulators of block b+1
// and a candidate whose predecessor ends in block b or b+1 (NEAR) goes, with its finished term, onto a small queue in LDS that block
// b+1 empties from the LDS ring of the newest columns.  What is left on the critical path of a block is LDS traffic.  The maximum is
// idempotent and order-free, so the result is bit-identical to densePiece<BLK, 0> whatever is evaluated when -- a block whose queue
// overflows, that has more descriptors than the LDS stage holds or more records than threads simply evaluates everything again in
// place (`needSync`; also block 0 and the last block).  The forward pass (a sum needs its largest term first) stays with
// densePiece<BLK, 1>.  Reference: UtrModel::viterbiForwardAndSampling, src/utrmodel.cc:796-1064; NAMGene::viterbiAndForward,
// src/namgene.cc:168-365.  Compiled for gfx950 and, with -DAUGX_EMU, for the lane-loop emulator (tests only).
#pragma once
#include "dense.h"

namespace augx {
namespace dev {

// This kernel runs with FOUR wavefronts (one per SIMD): a wavefront may then use 512 registers (256 VGPRs + 256 AGPRs), nothing is
// spilled to scratch -- and a reload from scratch would wait for every load in flight (vmcnt counts in order), which is what the
// landing pads are there to avoid.
constexpr int VNW = 4, VNT = VNW * WAVE;
constexpr int VSL = 3;   // UTR units a wavefront has in flight (slots)
#ifdef AUGX_EMU
#define FORV_THREADS(t) for (int t = 0; t < VNT; ++t)
#define FORV_WAVES(w) for (int w = 0; w < VNW; ++w)
#define WV2(T, name, K) T name[K][NWAVES]
#define WX2(name, k) name[k][w]
#else
#define FORV_THREADS(t) FOR_THREADS(t)
#define FORV_WAVES(w) FOR_WAVES(w)
#define WV2(T, name, K) T name[K][1]
#define WX2(name, k) name[k][0]
#endif

constexpr int DQCAP = 128; // NEAR candidates a block may queue
#ifdef AUGX_EMU // (tests: smaller limits, so that the blocks that fall back to evaluating in place are exercised)
inline int emuKnob(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#define DQLIM emuKnob("AUGX_EMU_DQCAP", DQCAP)
#define ITEMLIM ((uint32_t)emuKnob("AUGX_EMU_ITEMCAP", 2 * VNT))
#define UDLIM ((uint32_t)emuKnob("AUGX_EMU_UDCAP", UDCAP))
#else
#define DQLIM DQCAP
#define ITEMLIM ((uint32_t)(2 * VNT))
#define UDLIM ((uint32_t)UDCAP)
#endif
// a queue entry: the finished term and (end of the predecessor, base of the block, state, predecessor state, kind) in one word;
// kind 0: record of kCand (predecessor state src), 1: UTR exon (every ancestor), 2: RTERMINAL record
AUGX_HD uint64_t dqPack(int eop, int dj, int s2, int src, int kind) { return (uint64_t)(uint32_t)eop | ((uint64_t)dj << 32) | ((uint64_t)s2 << 40) | ((uint64_t)src << 48) | ((uint64_t)kind << 56); }

// ---- loads from HBM straight into LDS.  `row`: wave-uniform; lane i lands at row + i (dmaW, 4 bytes) or row + 4 i (dmaQ, 16 bytes)
#ifdef AUGX_EMU
inline void dmaW(uint32_t *row, int lane, const void *g) { memcpy(row + lane, g, 4); }
inline void dmaQ(uint32_t *row, int lane, const void *g) { memcpy(row + 4 * lane, g, 16); }
inline void dmaWait() {}
#else
__device__ __forceinline__ void dmaW(uint32_t *row, int, const void *g) { __builtin_amdgcn_global_load_lds((const AUGX_GLOBAL void *)g, (AUGX_LDS void *)row, 4, 0, 0); }
__device__ __forceinline__ void dmaQ(uint32_t *row, int, const void *g) { __builtin_amdgcn_global_load_lds((const AUGX_GLOBAL void *)g, (AUGX_LDS void *)row, 16, 0, 0); }
__device__ __forceinline__ void dmaWait() { __builtin_amdgcn_s_waitcnt(0x0F70); __asm__ volatile("" ::: "memory"); } // vmcnt(0)
#endif
AUGX_HD double dwPair(uint32_t lo, uint32_t hi) { union { uint64_t u; double d; } x; x.u = ((uint64_t)hi << 32) | lo; return x.d; }
constexpr int VUH = 2;   // candidates a lane of a UTR unit has in flight
constexpr int VPA = 3;   // ancestors of a UTR exon state whose cells are fetched ahead (no state of the reference's models has more)
constexpr int VFR = 2;   // fixed-lag (state, base) pairs a thread of the first wavefronts may have
constexpr int VFW = 3;   // ... and the number of those wavefronts

template <int BLK> struct DenseLdsV {
    // landing pads, [wavefront][...][lane]
    uint32_t pSite[VNW - 1][VSL][VUH][3][WAVE];     // site record of a candidate: predecessor end, (begin signal - prefix) lo / hi
    uint32_t pLen[VNW - 1][VSL][VUH][2][WAVE];      // its length term
    uint32_t pPv[VNW - 1][VSL][VUH][VPA][2][WAVE];  // the cells of its predecessor
    uint32_t pItem[VNW][2][WAVE * 4];               // two records of kCand per thread
    uint32_t pItemPv[VNW][2][2][WAVE];
    uint32_t pFix[VFW][VFR][2][2][WAVE];     // long-lag predecessor cells of the next block's fixed-lag states: [pair][ancestor][lo / hi]
    double ring[WAVE][SPX];          // the newest 64 columns, [j & 63][state]
    double cmax[3][BLK][SPX];        // accumulators of the variable-length cells: [block % 3][base of the block][state]
    double tr[SPX][AUGX_MAX_ANC];    // ln t(ancestor ai -> s) of the piece's first class
    uint8_t anc[SPX][AUGX_MAX_ANC], nanc[SPX];
    uint8_t cellKind[SPX];           // 1: candidates from the records of kCand, 2: reverse terminal exon, 3: UTR exon, 0: none
    double sg[3][BLK][NSIG];         // signal records, [block % 3]
    uint64_t bOff[3];
    uint32_t bCnt[3][2];
    uint64_t uOff[3];
    uint32_t uCnt[3];
    int chS[DCH], chNa[DCH], chAnc[DCH][AUGX_MAX_ANC], chNd[DCH], chDead[DCH][AUGX_MAX_ANC], chDeadAi[DCH][AUGX_MAX_ANC], chSgi[DCH];
    uint8_t chLive[DCH][AUGX_MAX_ANC], chEarly[DCH];
    double oth[DCH][BLK];
    uint8_t othAi[DCH][BLK];
    uint64_t ud[3][UDCAP * UDW];     // descriptors of the block's UTR exon cells, [block % 3]
    const double *lenTab[9];
    int lenMax[9];
    const USite *siteTab[6];
    double qTe[2][DQCAP];            // NEAR candidates of the block, [block % 2]
    uint64_t qMeta[2][DQCAP];
    int qn[2], needSync[2];
};

template <int BLK>
AUGX_KFN double denseAtV(const DenseLdsV<BLK> &L, const double *M, int S, int q, int a, int jbCur) {
    if (q < 0) q = 0;
    if (jbCur + BLK - 1 - q < WAVE) return ldsLoadD(&L.ring[q & 63][a]);
    return ldCoherent(&M[(int64_t)q * S + a]);
}

#if defined(AUGX_EMU) || !defined(AUGX_PROFILE)
#define VPROF(k) do {} while (0)
#else
#define VPROF(k) do { if (B.prof && threadIdx.x == WAVE) { const uint64_t now_ = clock64(); dpAcc[k] += now_ - dpLast; dpLast = now_; } } while (0)
#endif

template <int BLK>
AUGX_KFN void densePieceV(const DevTables &T, const BatchView &B, DenseLdsV<BLK> &L, int p) {
    const int n = B.len[p], S = T.S, c0 = B.cls[p];
    const int64_t o = B.off[p];
    double *M = B.cells + (o + 1) * S;
    uint8_t *BP = B.bpD ? B.bpD + (o + 1) * S : nullptr;
    const double *gSig = B.sig + (o + 1) * NSIG;
    const uint64_t *gUdOff = B.udOff;
    const uint32_t *gUdCnt = B.udCnt;
    const uint64_t *gUd = (const uint64_t *)B.ud;
    const Item *gItems = B.items;
    const uint64_t *gBlkOff = B.blkOff;
    const uint32_t *gBlkCnt = B.blkCnt, *gBlkSplit = B.blkSplit;
    const uint8_t *gPlane = B.gcPlane + o + 1;
    const int32_t *gPlaneCls = B.planeCls + p * MAXPL;
    const int initKind = B.initKind[p], termKind = B.termKind[p], synch = T.synch;
    if (c0 < 0) { FORV_THREADS(t) { if (t == 0) { B.lnv[p] = AUGX_NINF; B.status[p] = AUGX_E_HIP; B.finalState[p] = -1; } } return; }
    const bool multi = B.nPlanes[p] > 1;
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U, dL = T.dStateLen;
    auto clsAt = [&](int j) __attribute__((always_inline)) { return multi ? (int)gp(gPlaneCls)[gp(gPlane)[j]] : c0; };
    const double *gTrans = T.ln_trans;
    auto trn = [&](int cc, int s2, int ai) __attribute__((always_inline)) -> double {
        if (multi) return gp(gTrans)[((int64_t)cc * S + (*lp(&L.anc[s2][ai]))) * S + s2];
        return ldsLoadD(&L.tr[s2][ai]);
    };
    UCtx UX(T, B, p);
    bool anyNuc = false;
    FORV_THREADS(t) {
        for (int i = t; i < WAVE * SPX; i += VNT) (*lp(&L.ring[i / SPX][i % SPX])) = AUGX_NINF;
        for (int i = t; i < 3 * BLK * SPX; i += VNT) (*lp(&L.cmax[i / (BLK * SPX)][(i / SPX) % BLK][i % SPX])) = AUGX_NINF;
        if (t < SPX) {
            const int k = t < S && T.reachable[t] ? T.kind[t] : -1;
            (*lp(&L.cellKind[t])) = k < 0 ? 0 : k == AUGX_K_RTERMINAL ? 2 : isItemKind(k) ? 1 : isUtrExonKind(k) ? 3 : 0;
            (*lp(&L.nanc[t])) = (uint8_t)(t < S ? T.n_anc[t] : 0);
            for (int ai = 0; ai < AUGX_MAX_ANC; ai++) {
                const int a = (t < S && ai < T.n_anc[t]) ? T.anc[t][ai] : 0;
                (*lp(&L.anc[t][ai])) = (uint8_t)a;
                (*lp(&L.tr[t][ai])) = (t < S && ai < T.n_anc[t]) ? lnT(T, c0, a, t) : AUGX_NINF;
            }
        }
        if (t == 0) { (*lp(&L.qn[0])) = 0; (*lp(&L.qn[1])) = 0; (*lp(&L.needSync[0])) = 1; (*lp(&L.needSync[1])) = 0; } // (block 0 has nobody before it)
        for (int64_t i = t; i < (int64_t)n * S; i += VNT) gp(M)[i] = AUGX_NINF; // (absent cells stay -inf)
        if (BP) for (int64_t i = t; i < (int64_t)n * S; i += VNT) gp(BP)[i] = 0xFF;
    }
    for (int q = 0; q < n && !anyNuc; q++) anyNuc = B.code[o + 1 + q] < 4;
    BLOCK_GLOBAL_SYNC();
    FORV_THREADS(t) { // column 0 = initial probabilities (reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150)
        if (t < S) {
            const double v = initLn(T, initKind, t);
            (*lp(&L.ring[0][t])) = v;
            gp(M)[t] = v;
        }
    }
    BLOCK_GLOBAL_SYNC();
    int fixS[DFIX], nFix = 0, chS[DCH], nCh = 0, nEarly = 0;
    const int nUv = T.nUv;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int k = T.kind[s2];
        if (isFixedKind(k)) { if (nFix < DFIX) fixS[nFix++] = s2; }
        else if (isEarlyChainKind(k)) { if (nCh < DCH) chS[nCh++] = s2; }
    }
    nEarly = nCh;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int k = T.kind[s2];
        if (isChainKind(k) && !isEarlyChainKind(k)) { if (nCh < DCH) chS[nCh++] = s2; }
    }
    if (!anyNuc) { // all N: everything is intergenic (reference src/namgene.cc:205-226)
        FORV_THREADS(t) {
            if (t == 0) {
                double v = initLn(T, initKind, synch);
                for (int j = 1; j < n; j++) { v = v - T.ln4; gp(M)[(int64_t)j * S + synch] = v; if (BP) gp(BP)[(int64_t)j * S + synch] = 0xFE; }
                const double tl = termKind == 0 ? T.ln_term[synch] : 0.0;
                B.lnv[p] = v + tl; B.finalState[p] = (v + tl) > AUGX_NINF ? synch : -1; B.status[p] = (v + tl) > AUGX_NINF ? 0 : AUGX_E_NOPATH;
            }
        }
        return;
    }
    const int nBlocks = (n + BLK - 1) / BLK;
    const int64_t gb0 = o / BLK;
    constexpr int UW0 = BLK == 8 ? 3 : 1, A1T = UW0 * WAVE, FR = (DFIX * BLK + A1T - 1) / A1T, NUW = VNW - UW0;
    static_assert(BLK * NSIG <= A1T, "roles of stage 1");
    constexpr int UH = 2; // candidates a lane of a UTR unit has in flight
    TV2(int, fS2, FR); TV2(int, fLag, FR); TV2(int, fSg, FR);
    TV(int, cS2); TV(int, cSelf); TV(int, cFast); TV(int, cSgi);
    FORV_THREADS(t) {
        TX(cS2) = -1; TX(cSelf) = -1; TX(cFast) = 0; TX(cSgi) = SIG_EIN;
        for (int r = 0; r < FR; r++) {
            const int f = t + r * A1T;
            fS2[r][TI] = -1; fLag[r][TI] = 1; fSg[r][TI] = 0;
            if (t < A1T && f < nFix * BLK) {
                const int s2 = fixS[f / BLK], k = T.kind[s2];
                fS2[r][TI] = s2;
                fLag[r][TI] = (k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS) ? dssWhole : (k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS) ? assLag : dL;
                fSg[r][TI] = k == AUGX_K_LONGDSS ? SIG_DSSF : k == AUGX_K_RLONGDSS ? SIG_DSSR : k == AUGX_K_LONGASS ? SIG_ASSF : k == AUGX_K_RLONGASS ? SIG_ASSR : SIG_EQD;
            }
        }
        if (t == VNT - 1 && T.utr) {
            const double *lt[9] = {T.len5s, T.len5i, T.len5n, T.len5t, T.len3s, T.len3i, T.len3n, T.len3t, T.tail3s};
            const int lm[9] = {T.uML, T.uML, T.uML, T.uML, T.uM3S, T.uML, T.uML, T.uM3T, T.uM3S};
            for (int i = 0; i < 9; i++) { (*lp(&L.lenTab[i])) = lt[i]; (*lp(&L.lenMax[i])) = lm[i]; }
            for (int i = 0; i < 6; i++) (*lp(&L.siteTab[i])) = UX.list(i);
        }
        if (t < DCH) {
            const int s2 = t < nCh ? chS[t] : -1;
            (*lp(&L.chS[t])) = s2;
            const int na = s2 >= 0 ? T.n_anc[s2] : 0;
            (*lp(&L.chNa[t])) = na;
            (*lp(&L.chEarly[t])) = t < nEarly;
            (*lp(&L.chSgi[t])) = (s2 >= 0 && T.kind[s2] == AUGX_K_IGENIC) ? SIG_EIG : SIG_EIN;
            int nd = 0, nLive = 0, selfAi = -1;
            bool onlySelf = true;
            for (int ai = 0; ai < AUGX_MAX_ANC; ai++) {
                const int a = ai < na ? T.anc[s2][ai] : 0;
                const bool live = ai < na && isChainKind(T.kind[a]) && (isEarlyChainKind(T.kind[a]) == (t < nEarly));
                (*lp(&L.chAnc[t][ai])) = a;
                (*lp(&L.chLive[t][ai])) = live;
                if (live) { nLive++; if (a == s2) selfAi = ai; else onlySelf = false; }
                if (ai < na && !live) { (*lp(&L.chDead[t][nd])) = a; (*lp(&L.chDeadAi[t][nd])) = ai; nd++; }
            }
            for (int k2 = nd; k2 < AUGX_MAX_ANC; k2++) { (*lp(&L.chDead[t][k2])) = 0; (*lp(&L.chDeadAi[t][k2])) = 0; }
            (*lp(&L.chNd[t])) = nd;
            TX(cS2) = s2; TX(cSelf) = selfAi; TX(cFast) = s2 >= 0 && onlySelf && nLive <= 1;
            TX(cSgi) = (s2 >= 0 && T.kind[s2] == AUGX_K_IGENIC) ? SIG_EIG : SIG_EIN;
        }
    }
    // offsets, signal records and descriptors of block bb into the stage buffers of index bb % 3 (threads [t0, t0 + nt) of the workgroup)
    auto stageBlock = [&](int t, int t0, int nt, int bb) __attribute__((always_inline)) {
        if (t < t0 || t >= t0 + nt) return;
        const int u = t - t0, ix = bb % 3;
        const int64_t gb = gb0 + bb;
        if (bb >= nBlocks) { if (u == 0) { (*lp(&L.bCnt[ix][0])) = 0; (*lp(&L.bCnt[ix][1])) = 0; (*lp(&L.uCnt[ix])) = 0; } return; }
        if (u == nt - 1) { (*lp(&L.bOff[ix])) = gp(gBlkOff)[gb * 2 + 1]; (*lp(&L.bCnt[ix][0])) = gp(gBlkCnt)[gb * 2 + 1]; (*lp(&L.bCnt[ix][1])) = gp(gBlkSplit)[gb * 3 + 2]; }
        if (u < BLK * NSIG) { const int dj = u / NSIG, j = bb * BLK + dj; (*lp(&L.sg[ix][dj][u % NSIG])) = j < n ? gp(gSig)[(int64_t)j * NSIG + u % NSIG] : AUGX_NINF; }
        if (nUv > 0) {
            const uint64_t uo = gp(gUdOff)[gb];
            const uint32_t uc = gp(gUdCnt)[gb];
            if (u == 0) { (*lp(&L.uOff[ix])) = uo; (*lp(&L.uCnt[ix])) = uc; }
            const uint32_t words = (uc < (uint32_t)UDCAP ? uc : (uint32_t)UDCAP) * UDW;
            for (uint32_t i = (uint32_t)u; i < words; i += (uint32_t)nt) (*lp(&L.ud[ix][i])) = gp(gUd)[uo * UDW + i];
        } else if (u == 0) (*lp(&L.uCnt[ix])) = 0;
    };
    FORV_THREADS(t) { stageBlock(t, 0, VNT / 2, 0); stageBlock(t, VNT / 2, VNT / 2, 1); }
    BLOCK_SYNC();
    auto chainOthers = [&](int slot, int dj, int jb, int cur, bool acc) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot])), j = jb + dj;
        double f = AUGX_NINF;
        int fa = 0xFF;
        if (s2 >= 0 && j >= 1 && j < n) {
            const double emi = (*lp(&L.sg[cur][dj][(*lp(&L.chSgi[slot]))]));
            const int cc = clsAt(j), nd = (*lp(&L.chNd[slot]));
            double m = AUGX_NINF;
            for (int k2 = 0; k2 < AUGX_MAX_ANC; k2++) {
                if (k2 < nd) {
                    const int a = (*lp(&L.chDead[slot][k2])), ai = (*lp(&L.chDeadAi[slot][k2]));
                    double pv, x = AUGX_NINF;
                    if (acc && dj >= 1 && j - 1 >= 1) pv = (*lp(&L.cmax[cur][dj - 1][a]));
                    else pv = ldsLoadD(&L.ring[(j - 1) & 63][a]);
                    if (pv > AUGX_NINF) x = pv + (trn(cc, s2, ai) + emi);
                    if (x > m) { m = x; fa = ai; } // (ascending ancestors, strict '>': the first of equals, as the reference)
                }
            }
            f = m;
        }
        (*lp(&L.oth[slot][dj])) = f;
        (*lp(&L.othAi[slot][dj])) = (uint8_t)fa;
    };
    auto chainRun = [&](int slot, int jb, int cur) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot]));
        if (s2 < 0) return;
        const int na = (*lp(&L.chNa[slot])), sgi = (*lp(&L.chSgi[slot]));
        for (int dj = 0; dj < BLK; dj++) {
            const int j = jb + dj;
            if (j >= n || j < 1) continue;
            const int cc = clsAt(j);
            const double emi = (*lp(&L.sg[cur][dj][sgi]));
            double f = (*lp(&L.oth[slot][dj]));
            int fa = (*lp(&L.othAi[slot][dj]));
            for (int ai = 0; ai < na; ai++) {
                if (!(*lp(&L.chLive[slot][ai]))) continue;
                const int a = (*lp(&L.chAnc[slot][ai]));
                const double pv = ldsLoadD(&L.ring[(j - 1) & 63][a]);
                if (!(pv > AUGX_NINF)) continue;
                const double x = pv + (trn(cc, s2, ai) + emi);
                if (x > f || (x == f && ai < fa)) { f = x; fa = ai; }
            }
            (*lp(&L.ring[j & 63][s2])) = f;
            if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
        }
    };
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    uint64_t dpAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dpLast = clock64();
#endif
    auto chainRunSelf = [&](int slot, int s2, int selfAi, int sgi, int jb, int cur) __attribute__((always_inline)) {
        double em[BLK], ot[BLK];
        int oa[BLK];
        _Pragma("unroll") for (int dj = 0; dj < BLK; dj++) { em[dj] = (*lp(&L.sg[cur][dj][sgi])); ot[dj] = (*lp(&L.oth[slot][dj])); oa[dj] = (*lp(&L.othAi[slot][dj])); }
        const int jp = (jb >= 1 ? jb : 1) - 1;
        double prev = ldsLoadD(&L.ring[jp & 63][s2]);
        const double tSelf = (selfAi >= 0 && !multi) ? ldsLoadD(&L.tr[s2][selfAi]) : 0.0;
        _Pragma("unroll") for (int dj = 0; dj < BLK; dj++) {
            const int j = jb + dj;
            if (j >= n || j < 1) continue;
            double f = ot[dj];
            int fa = oa[dj];
            if (selfAi >= 0 && prev > AUGX_NINF) {
                const double x = prev + ((multi ? trn(clsAt(j), s2, selfAi) : tSelf) + em[dj]);
                if (x > f || (x == f && selfAi < fa)) { f = x; fa = selfAi; }
            }
            (*lp(&L.ring[j & 63][s2])) = f;
            if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
            prev = f;
        }
    };
    bool lateAcc = true;
    for (int slot = nEarly; slot < nCh; slot++) {
        const int s2 = chS[slot];
        for (int ai = 0; ai < T.n_anc[s2]; ai++) {
            const int a = T.anc[s2][ai], ka = T.kind[a];
            const bool live = isChainKind(ka) && !isEarlyChainKind(ka);
            if (!live && !((isItemKind(ka) && ka != AUGX_K_RTERMINAL) || isUtrExonKind(ka))) lateAcc = false;
        }
    }
    constexpr int NTW = VNT - WAVE;
    // ---- per-thread state of the work done ahead (for block b + 1 while block b runs): computed values only -- what is on its way
    //      from HBM sits in the landing pads
    static_assert(FR <= VFR && UW0 <= VFW && UH == VUH, "landing pads");
    TV2(int, iFl, 2);
    TV2(int, uXi, VSL * UH); TV2(int, uEop, VSL * UH); TV2(double, uSig, VSL * UH); TV2(int, uFl, VSL * UH);
    WV2(int, myD, VSL); WV2(int, myC, VSL);
    FORV_THREADS(t) { iFl[0][TI] = 0; iFl[1][TI] = 0; for (int h = 0; h < VSL * UH; h++) uFl[h][TI] = 0; }
    FORV_WAVES(w) { for (int k = 0; k < VSL; k++) { WX2(myD, k) = -1; WX2(myC, k) = 0; } }
    // a NEAR candidate onto the queue of the next block
    auto enqueue = [&](int qx, double te, int eop, int dj, int s2, int src, int kind) __attribute__((always_inline)) {
#ifdef AUGX_EMU
        const int at = L.qn[qx]++;
#else
        const int at = __hip_atomic_fetch_add(lp(&L.qn[qx]), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        if (at < DQLIM) { (*lp(&L.qTe[qx][at])) = te; (*lp(&L.qMeta[qx][at])) = dqPack(eop, dj, s2, src, kind); }
        else (*lp(&L.needSync[qx])) = 1;
    };
    // a cell of the matrix onto a landing pad: from HBM, or -- a column the ring still holds -- copied within LDS
    auto fetchCell = [&](uint32_t *lo, uint32_t *hi, int lane, int q, int a, int jbCur) __attribute__((always_inline)) {
        if (q < 0) q = 0;
        if (jbCur + BLK - 1 - q < WAVE) {
            union { double d; uint32_t w[2]; } x;
            x.d = ldsLoadD(&L.ring[q & 63][a]);
            (*lp(&lo[lane])) = x.w[0]; (*lp(&hi[lane])) = x.w[1];
        } else {
            const uint32_t *g = (const uint32_t *)&M[(int64_t)q * S + a];
            dmaW(lo, lane, g); dmaW(hi, lane, g + 1);
        }
    };
    // the three levels of a UTR unit (descriptor D of the next block, unit c of it), lane t of wavefront w.
    // jbCur: first base of the block that is running (a predecessor that ends before it is final)
    auto utrL1 = [&](int w, int t, int sl, const UDesc &D, int c) __attribute__((always_inline)) {
        const USite *sites = (*lp(&L.siteTab[D.list]));
        const int lane = t & 63;
        _Pragma("unroll") for (int h = 0; h < UH; h++) {
            const int idx = c * UH * WAVE + h * WAVE + lane;
            int li, xi;
            utrCandIndex(D, idx, li, xi);
            const bool act = idx < D.total, pre = act && xi < 0 && li < D.nPre;
            uXi[sl * UH + h][TI] = xi;
            uFl[sl * UH + h][TI] = (act ? 1 : 0) | (pre ? 4 : 0) | (li << 8);
            if (act && xi < 0 && !pre) { // (the field of the record is chosen by address)
                const USite *e = sites + ((int64_t)D.i1 - 1 - li);
                const uint32_t *gb2 = (const uint32_t *)&e->b[D.bsel];
                dmaW(&L.pSite[w - 1][sl][h][0][0], lane, &e->pos);
                dmaW(&L.pSite[w - 1][sl][h][1][0], lane, gb2);
                dmaW(&L.pSite[w - 1][sl][h][2][0], lane, gb2 + 1);
            }
        }
    };
    auto utrL2 = [&](int w, int t, int sl, const UDesc &D, int jbCur) __attribute__((always_inline)) {
        const int s2 = D.s, na = (*lp(&L.nanc[s2])), lane = t & 63;
        _Pragma("unroll") for (int h = 0; h < UH; h++) {
            const int fl = uFl[sl * UH + h][TI];
            bool act = fl & 1;
            const bool pre = fl & 4;
            const int li = fl >> 8;
            double sig = AUGX_NINF;
            int eop = 0, len = 0;
            bool tail3 = false, haveLen = false;
            if (pre) { // (evaluated by the descriptor kernel: the whole term)
                sig = li == 0 ? D.preTe[0] : li == 1 ? D.preTe[1] : D.preTe[2];
                eop = li == 0 ? D.preEop[0] : li == 1 ? D.preEop[1] : D.preEop[2];
                act = sig > AUGX_NINF;
            } else if (act) {
                int sPos = 0;
                double sB = AUGX_NINF;
                if (uXi[sl * UH + h][TI] < 0) { sPos = (int)(*lp(&L.pSite[w - 1][sl][h][0][lane])); sB = dwPair((*lp(&L.pSite[w - 1][sl][h][1][lane])), (*lp(&L.pSite[w - 1][sl][h][2][lane]))); }
                act = utrCandPre(UX, D, uXi[sl * UH + h][TI], sPos, sB, sig, len, tail3, eop, false) == 1;
                if (act) {
                    const int ti = tail3 ? 8 : D.len;
                    if (len >= 0 && len <= (*lp(&L.lenMax[ti]))) {
                        const uint32_t *g = (const uint32_t *)&(*lp(&L.lenTab[ti]))[len];
                        dmaW(&L.pLen[w - 1][sl][h][0][0], lane, g); dmaW(&L.pLen[w - 1][sl][h][1][0], lane, g + 1);
                        haveLen = true;
                    } else act = false;
                }
            }
            const bool near = act && eop >= jbCur;
            uSig[sl * UH + h][TI] = sig; uEop[sl * UH + h][TI] = eop;
            uFl[sl * UH + h][TI] = (act ? 1 : 0) | (near ? 2 : 0) | (haveLen ? 8 : 0);
            if (act && !near) {
                _Pragma("unroll") for (int ai = 0; ai < VPA; ai++)
                    if (ai < na) fetchCell(&L.pPv[w - 1][sl][h][ai][0][0], &L.pPv[w - 1][sl][h][ai][1][0], lane, eop, (*lp(&L.anc[s2][ai])), jbCur);
            }
        }
    };
    auto utrL3 = [&](int w, int t, int sl, const UDesc &D, int jbCur, int jbNext, int nxt, int qx) __attribute__((always_inline)) {
        const int s2 = D.s, dj = D.j - jbNext, cc = clsAt(D.j), na = (*lp(&L.nanc[s2])), lane = t & 63;
        double vmax = AUGX_NINF;
        _Pragma("unroll") for (int h = 0; h < UH; h++) {
            const int fl = uFl[sl * UH + h][TI];
            uFl[sl * UH + h][TI] = 0;
            if (!(fl & 1)) continue;
            const double lnLen = (fl & 8) ? dwPair((*lp(&L.pLen[w - 1][sl][h][0][lane])), (*lp(&L.pLen[w - 1][sl][h][1][lane]))) : 0.0;
            if (!(lnLen > AUGX_NINF)) continue;
            const double te = uSig[sl * UH + h][TI] + lnLen;
            if (!(te > AUGX_NINF)) continue;
            const int eop = uEop[sl * UH + h][TI];
            if (fl & 2) { enqueue(qx, te, eop, dj, s2, 0, 1); continue; }
            _Pragma("unroll") for (int ai = 0; ai < VPA; ai++) {
                if (ai >= na) continue;
                const double pv = dwPair((*lp(&L.pPv[w - 1][sl][h][ai][0][lane])), (*lp(&L.pPv[w - 1][sl][h][ai][1][lane])));
                if (!(pv > AUGX_NINF)) continue;
                const double v = pv + (trn(cc, s2, ai) + te);
                vmax = v > vmax ? v : vmax;
            }
            for (int ai = VPA; ai < na; ai++) { // (not in the reference's models)
                const double pw = denseAtV<BLK>(L, M, S, eop, (*lp(&L.anc[s2][ai])), jbCur);
                if (!(pw > AUGX_NINF)) continue;
                const double v = pw + (trn(cc, s2, ai) + te);
                vmax = v > vmax ? v : vmax;
            }
        }
        if (vmax > AUGX_NINF) ldsMaxD(&L.cmax[nxt][dj][s2], vmax);
    };
    for (int b = 0; b < nBlocks; b++) {
        const int jb = b * BLK, cur = b % 3, nxt = (b + 1) % 3, rst = (b + 2) % 3, qc = b & 1, qx = (b + 1) & 1, jbN = jb + BLK;
        const bool hasNext = b + 1 < nBlocks;
        // the block of the last base evaluates in place (one of its cells is made twice, below), and so does the one before it
        const bool lastBlock = nUv > 0 && jb + BLK > n - 1;
        VPROF(0);
        const bool sync = (*lp(&L.needSync[qc])) != 0 || lastBlock;
        const uint64_t i0 = (*lp(&L.bOff[cur]));
        const uint32_t cntAll = (*lp(&L.bCnt[cur][0])), cntNonRT = (*lp(&L.bCnt[cur][1]));
        const uint64_t i0N = (*lp(&L.bOff[nxt]));
        const uint32_t cntAllN = hasNext ? (*lp(&L.bCnt[nxt][0])) : 0u, cntNonRTN = (*lp(&L.bCnt[nxt][1]));
        const uint32_t ndN = hasNext && nUv > 0 ? (uint32_t)uni((int)(*lp(&L.uCnt[nxt]))) : 0u;
        const bool aheadItems = cntAllN <= ITEMLIM, aheadUtr = ndN <= UDLIM;
        auto candValue = [&](const Item &I, int &dj, int &s2) __attribute__((always_inline)) -> double {
            dj = (int)(I.kp >> (KEY_BITS + 7)); s2 = (int)((I.kp >> KEY_BITS) & 127);
            if (!(I.te > AUGX_NINF)) return AUGX_NINF;
            const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
            const double pv = denseAtV<BLK>(L, M, S, eop, (int)(I.src & 127u), jb);
            return pv + I.te;
        };
        // every record [lo2, hi2) of this block, in place (the block that could not be prepared ahead)
        auto itemPass = [&](uint32_t lo2, uint32_t hi2) __attribute__((always_inline)) {
            FORV_THREADS(t) {
                for (uint32_t it = lo2 + (uint32_t)(t - WAVE); t >= WAVE && it < hi2; it += NTW) {
                    int dj, s2;
                    const double v = candValue(ldItem(gItems + i0 + it), dj, s2);
                    if (v > AUGX_NINF) ldsMaxD(&L.cmax[cur][dj][s2], v);
                }
            }
        };
        // every UTR exon candidate of this block, in place (only >= 0: that one descriptor only)
        auto utrPass = [&](int only) __attribute__((always_inline)) {
            const uint32_t nd = (uint32_t)uni((int)(*lp(&L.uCnt[cur])));
            const uint64_t uo = (*lp(&L.uOff[cur]));
            FORV_WAVES(w) {
                if (w >= UW0) {
                    uint32_t wi = 0;
                    for (uint32_t d = only >= 0 ? (uint32_t)only : 0u; d < (only >= 0 ? (uint32_t)only + 1u : nd); d++) {
                        const int total = uni(d < (uint32_t)UDCAP ? ((const UDesc *)&L.ud[cur][d * UDW])->total : B.ud[uo + d].total);
                        const int nun = (total + UH * WAVE - 1) / (UH * WAVE);
                        const int mine = (int)((uint32_t)(w - UW0 + NUW - (int)(wi % NUW)) % NUW);
                        wi += (uint32_t)nun;
                        if (mine >= nun) continue;
                        const UDesc D = d < (uint32_t)UDCAP ? *(const UDesc *)&L.ud[cur][d * UDW] : B.ud[uo + d];
                        const int s2 = D.s, dj = D.j - jb, cc = clsAt(D.j), na = (*lp(&L.nanc[s2]));
                        const USite *sites = (*lp(&L.siteTab[D.list]));
                        for (int c = mine; c < nun; c += NUW) {
                            FOR_WLANES(t, w) {
                                double vmax = AUGX_NINF;
                                for (int h = 0; h < UH; h++) {
                                    const int idx = c * UH * WAVE + h * WAVE + (t & 63);
                                    if (idx >= total) continue;
                                    int li, xi, eop = 0, len = 0;
                                    utrCandIndex(D, idx, li, xi);
                                    double sig = AUGX_NINF, lnLen = AUGX_NINF;
                                    bool tail3 = false;
                                    if (xi < 0 && li < D.nPre) {
                                        sig = li == 0 ? D.preTe[0] : li == 1 ? D.preTe[1] : D.preTe[2];
                                        eop = li == 0 ? D.preEop[0] : li == 1 ? D.preEop[1] : D.preEop[2];
                                        lnLen = 0.0;
                                        if (!(sig > AUGX_NINF)) continue;
                                    } else {
                                        USite e;
                                        e.pos = 0; e.pad = 0; e.b[0] = e.b[1] = e.b[2] = AUGX_NINF;
                                        if (xi < 0) e = ldUSite(sites + ((int64_t)D.i1 - 1 - li));
                                        if (utrCandPre(UX, D, xi, e.pos, D.bsel == 0 ? e.b[0] : D.bsel == 1 ? e.b[1] : e.b[2], sig, len, tail3, eop, false) != 1) continue;
                                        const int ti = tail3 ? 8 : D.len;
                                        lnLen = (len >= 0 && len <= (*lp(&L.lenMax[ti]))) ? gp((*lp(&L.lenTab[ti])))[len] : AUGX_NINF;
                                    }
                                    if (!(lnLen > AUGX_NINF)) continue;
                                    const double te = sig + lnLen;
                                    if (!(te > AUGX_NINF)) continue;
                                    for (int ai = 0; ai < na; ai++) {
                                        const double pw = denseAtV<BLK>(L, M, S, eop, (*lp(&L.anc[s2][ai])), jb);
                                        if (!(pw > AUGX_NINF)) continue;
                                        const double v = pw + (trn(cc, s2, ai) + te);
                                        vmax = v > vmax ? v : vmax;
                                    }
                                }
                                if (vmax > AUGX_NINF) ldsMaxD(&L.cmax[cur][dj][s2], vmax);
                            }
                        }
                    }
                }
            }
        };
        // the NEAR candidates of this block from its queue: predecessor values from the ring (kinds 0, 1: stage 2; kind 2: stage 5)
        auto queuePass = [&](bool rt) __attribute__((always_inline)) {
            int nq = (*lp(&L.qn[qc]));
            if (nq > DQLIM) nq = DQLIM;
            FORV_THREADS(t) {
                for (int i = t; i < nq; i += VNT) {
                    const uint64_t mt = (*lp(&L.qMeta[qc][i]));
                    const double te = (*lp(&L.qTe[qc][i]));
                    const int eop = (int)(uint32_t)mt, dj = (int)((mt >> 32) & 255), s2 = (int)((mt >> 40) & 255), src = (int)((mt >> 48) & 255), kind = (int)(mt >> 56);
                    if ((kind == 2) != rt) continue;
                    const int q = eop < 0 ? 0 : eop;
                    if (kind != 1) {
                        const double pv = ldsLoadD(&L.ring[q & 63][src]);
                        if (pv > AUGX_NINF) ldsMaxD(&L.cmax[cur][dj][s2], pv + te);
                    } else {
                        const int na = (*lp(&L.nanc[s2])), cc = clsAt(jb + dj);
                        double vmax = AUGX_NINF;
                        for (int ai = 0; ai < na; ai++) {
                            const double pw = ldsLoadD(&L.ring[q & 63][(*lp(&L.anc[s2][ai]))]);
                            if (!(pw > AUGX_NINF)) continue;
                            const double v = pw + (trn(cc, s2, ai) + te);
                            vmax = v > vmax ? v : vmax;
                        }
                        if (vmax > AUGX_NINF) ldsMaxD(&L.cmax[cur][dj][s2], vmax);
                    }
                }
            }
        };
        auto cellsOf = [&](int t, int kindMask) __attribute__((always_inline)) {
            for (int i = t - WAVE; i < BLK * SPX; i += NTW) {
                const int dj = i / SPX, s2 = i % SPX, j = jb + dj;
                if (j >= 1 && j < n && ((kindMask >> (*lp(&L.cellKind[s2]))) & 1)) {
                    const double f = (*lp(&L.cmax[cur][dj][s2]));
                    (*lp(&L.ring[j & 63][s2])) = f;
                    if (f > AUGX_NINF) gp(M)[(int64_t)j * S + s2] = f;
                }
            }
        };
        // ---- 1: AHEAD, level 1: records and site records of block b + 1 into their landing pads (first: the loads then have the
        //         stage); the fixed-lag states of this block; the stage buffers of block b + 2
        FORV_WAVES(w) { // the units of the next block that are this wavefront's: unit u (in the order of the descriptors) is slot u / NUW of wavefront UW0 + u % NUW
            for (int k = 0; k < VSL; k++) { WX2(myD, k) = -1; WX2(myC, k) = 0; }
            if (w >= UW0 && aheadUtr) {
                uint32_t wi = 0;
                for (uint32_t d = 0; d < ndN && wi < (uint32_t)(VSL * NUW); d++) {
                    const int total = uni(((const UDesc *)&L.ud[nxt][d * UDW])->total);
                    const int nun = (total + UH * WAVE - 1) / (UH * WAVE);
                    for (int c = 0; c < nun; c++) {
                        const uint32_t u = wi + (uint32_t)c;
                        if ((int)(u % NUW) == w - UW0 && u / NUW < (uint32_t)VSL) { WX2(myD, u / NUW) = (int)d; WX2(myC, u / NUW) = c; }
                    }
                    wi += (uint32_t)nun;
                }
                for (int k = 0; k < VSL; k++)
                    if (WX2(myD, k) >= 0) {
                        const UDesc D = *(const UDesc *)&L.ud[nxt][WX2(myD, k) * UDW];
                        FOR_WLANES(t, w) { utrL1(w, t, k, D, WX2(myC, k)); }
                    }
            }
        }
        FORV_THREADS(t) {
            const int wv = t >> 6, lane = t & 63;
            _Pragma("unroll") for (int k = 0; k < 2; k++) { // two records of the next block
                iFl[k][TI] = 0;
                const uint32_t it = (uint32_t)t + (uint32_t)k * VNT;
                if (hasNext && aheadItems && it < cntAllN) { dmaQ(&L.pItem[wv][k][0], lane, gItems + i0N + it); iFl[k][TI] = 1; }
            }
            if (t == 0 && hasNext && (!aheadItems || !aheadUtr)) (*lp(&L.needSync[qx])) = 1;
            for (int i = t; t < A1T && i < BLK * SPX; i += A1T) (*lp(&L.cmax[rst][i / SPX][i % SPX])) = AUGX_NINF;
            _Pragma("unroll") for (int r = 0; r < FR; r++)
            if (fS2[r][TI] >= 0) {
                const int s2 = fS2[r][TI], dj = (t + r * A1T) % BLK, j = jb + dj, lag = fLag[r][TI];
                if (j >= 1 && j < n) {
                    const double emi = (*lp(&L.sg[cur][dj][fSg[r][TI]]));
                    double f = AUGX_NINF;
                    int fa = 0xFF;
                    if (j - lag >= 0 && emi > AUGX_NINF) {
                        const int cc = clsAt(j), na = (*lp(&L.nanc[s2]));
                        for (int ai = 0; ai < na; ai++) {
                            // (a lag beyond the ring: the cell was fetched while the block before ran -- block 0 has no block before it)
                            const double pv = (lag >= WAVE - 2 * BLK && ai < 2 && b > 0) ? dwPair((*lp(&L.pFix[wv][r][ai][0][lane])), (*lp(&L.pFix[wv][r][ai][1][lane])))
                                                                                          : denseAtV<BLK>(L, M, S, j - lag, (*lp(&L.anc[s2][ai])), jb);
                            if (!(pv > AUGX_NINF)) continue;
                            const double x = pv + (trn(cc, s2, ai) + emi);
                            if (x > f) { f = x; fa = ai; }
                        }
                    }
                    (*lp(&L.ring[j & 63][s2])) = f;
                    if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
                }
                // the long-lag predecessors of the same state in the next block (after the pad has been read)
                const int jn = j + BLK;
                if (lag >= WAVE - 2 * BLK && jn < n && jn - lag >= 0) {
                    _Pragma("unroll") for (int ai = 0; ai < 2; ai++)
                        if (ai < (*lp(&L.nanc[s2]))) {
                            const uint32_t *g = (const uint32_t *)&M[(int64_t)(jn - lag) * S + (*lp(&L.anc[s2][ai]))];
                            dmaW(&L.pFix[wv][r][ai][0][0], lane, g); dmaW(&L.pFix[wv][r][ai][1][0], lane, g + 1);
                        }
                }
            }
            if (t < WAVE) { // the stage buffers of block b + 2, first half: offsets, counts, signal records
                if (b + 2 < nBlocks) {
                    const int64_t gb2 = gb0 + b + 2;
                    if (t < 2) dmaW((uint32_t *)&L.bOff[rst], t, (const uint32_t *)&gBlkOff[gb2 * 2 + 1] + t);
                    if (t < 2) dmaW((uint32_t *)&L.bCnt[rst][0], t, t == 0 ? (const uint32_t *)&gBlkCnt[gb2 * 2 + 1] : (const uint32_t *)&gBlkSplit[gb2 * 3 + 2]);
                    if (nUv > 0) {
                        if (t < 2) dmaW((uint32_t *)&L.uOff[rst], t, (const uint32_t *)&gUdOff[gb2] + t);
                        if (t < 1) dmaW((uint32_t *)&L.uCnt[rst], t, (const uint32_t *)&gUdCnt[gb2]);
                    } else if (t == 0) (*lp(&L.uCnt[rst])) = 0;
                    if (t < BLK * NSIG / 2) { // 16 bytes = two fields of a record; a record beyond the piece reads as absent
                        const int j2 = (b + 2) * BLK + t / (NSIG / 2);
                        if (j2 < n) dmaQ((uint32_t *)&L.sg[rst][0][0], t, (const char *)(gSig + (int64_t)(b + 2) * BLK * NSIG) + 16 * t);
                        else { (*lp(&L.sg[rst][t / (NSIG / 2)][2 * (t % (NSIG / 2))])) = AUGX_NINF; (*lp(&L.sg[rst][t / (NSIG / 2)][2 * (t % (NSIG / 2)) + 1])) = AUGX_NINF; }
                    }
                } else if (t == 0) { (*lp(&L.bCnt[rst][0])) = 0; (*lp(&L.bCnt[rst][1])) = 0; (*lp(&L.uCnt[rst])) = 0; }
            }
        }
        VPROF(1);
        BLOCK_SYNC();
        VPROF(2);
        // ---- 2: the candidates of this block that could not be evaluated ahead; what reaches the early chain states
        if (sync) { itemPass(0, cntNonRT); if (nUv > 0) utrPass(-1); }
        else queuePass(false);
        FORV_THREADS(t) { if (t >= WAVE && t - WAVE < nEarly * BLK) chainOthers((t - WAVE) / BLK, (t - WAVE) % BLK, jb, cur, false); }
        VPROF(3);
        BLOCK_SYNC();
        VPROF(4);
        // ---- 3: AHEAD, level 2 (what level 1 fetched has arrived): predecessor ends; predecessor cells and length terms on their
        //         way; the descriptors of block b + 2.  The cells of the variable-length states; the early chain states over the block
        FORV_THREADS(t) {
            const int wv = t >> 6, lane = t & 63;
            dmaWait();
            _Pragma("unroll") for (int k = 0; k < 2; k++)
            if (iFl[k][TI] & 1) {
                const uint32_t *pi = &L.pItem[wv][k][lane * 4];
                const double te = dwPair((*lp(&pi[0])), (*lp(&pi[1])));
                const uint32_t kp = (*lp(&pi[2])), src = (*lp(&pi[3]));
                const int eop = (int)(kp & KEY_MASK) - KEY_BIAS;
                const bool ok = te > AUGX_NINF, near = eop >= jb;
                iFl[k][TI] = ok ? (near ? 3 : 1) : 0;
                if (ok && !near) fetchCell(&L.pItemPv[wv][k][0][0], &L.pItemPv[wv][k][1][0], lane, eop, (int)(src & 127u), jb);
            }
            if (t < WAVE && nUv > 0 && b + 2 < nBlocks) { // the stage buffers of block b + 2, second half: its descriptors, 16 bytes a lane
                const uint64_t uo2 = (*lp(&L.uOff[rst]));
                const uint32_t uc2 = (*lp(&L.uCnt[rst])), chunks = (uc2 < (uint32_t)UDCAP ? uc2 : (uint32_t)UDCAP) * (UDW / 2);
                for (uint32_t c0 = 0; c0 < chunks; c0 += WAVE)
                    if (c0 + (uint32_t)t < chunks) dmaQ((uint32_t *)&L.ud[rst][0] + c0 * 4, t, (const char *)(gUd + uo2 * UDW) + 16 * (c0 + (uint32_t)t));
            }
        }
        FORV_WAVES(w) {
            if (w >= UW0)
                for (int k = 0; k < VSL; k++)
                    if (WX2(myD, k) >= 0) {
                        const UDesc D = *(const UDesc *)&L.ud[nxt][WX2(myD, k) * UDW];
                        FOR_WLANES(t, w) { utrL2(w, t, k, D, jb); }
                    }
        }
        FORV_THREADS(t) {
            if (t >= WAVE) cellsOf(t, (1 << 1) | (1 << 3));
            if (t < nEarly) { if (TX(cFast)) chainRunSelf(t, TX(cS2), TX(cSelf), TX(cSgi), jb, cur); else chainRun(t, jb, cur); }
            if (lateAcc && !lastBlock && t >= VNT - (nCh - nEarly) * BLK) { const int u = t - (VNT - (nCh - nEarly) * BLK); chainOthers(nEarly + u / BLK, u % BLK, jb, cur, true); }
        }
        VPROF(5);
        BLOCK_SYNC();
        VPROF(6);
        // the right-truncated 3' UTR exon at the last base of the piece may begin anywhere up to that base (src/utrmodel.cc:880-884):
        // its predecessors of this very block exist only now -- the cell is made once more, from all of them
        if (nUv > 0 && jb + BLK > n - 1 && jb <= n - 1) {
            const uint32_t nd = (*lp(&L.uCnt[cur]));
            const uint64_t uo = (*lp(&L.uOff[cur]));
            for (uint32_t d = 0; d < nd; d++) {
                const UDesc D = d < (uint32_t)UDCAP ? *(const UDesc *)&L.ud[cur][d * UDW] : B.ud[uo + d];
                if (D.kind != AUGX_K_UTR3SINGLE || D.j != n - 1 || D.total == 0) continue;
                const int s2 = D.s;
                FORV_THREADS(t) { if (t == 0) (*lp(&L.cmax[cur][n - 1 - jb][s2])) = AUGX_NINF; }
                BLOCK_SYNC();
                utrPass((int)d);
                BLOCK_SYNC();
                FORV_THREADS(t) {
                    if (t == 0) {
                        const double f = (*lp(&L.cmax[cur][n - 1 - jb][s2]));
                        (*lp(&L.ring[(n - 1) & 63][s2])) = f;
                        gp(M)[(int64_t)(n - 1) * S + s2] = f;
                    }
                }
                BLOCK_SYNC();
            }
        }
        // ---- 4: late chain states (intergenic, UTR introns: fed by the exon cells of the base before and by themselves)
        if (!(lateAcc && !lastBlock)) {
            FORV_THREADS(t) { if (t >= WAVE && t - WAVE < (nCh - nEarly) * BLK) chainOthers(nEarly + (t - WAVE) / BLK, (t - WAVE) % BLK, jb, cur, false); }
            BLOCK_SYNC();
        }
        FORV_THREADS(t) {
            if (t >= nEarly && t < nCh) { if (TX(cFast)) chainRunSelf(t, TX(cS2), TX(cSelf), TX(cSgi), jb, cur); else chainRun(t, jb, cur); }
        }
        BLOCK_SYNC();
        VPROF(7);
        // ---- 5: AHEAD, level 3 (what level 2 fetched has arrived): the values of block b + 1 into its accumulators, its NEAR
        //         candidates onto its queue; the units beyond a wavefront's first, level after level.  Reverse terminal exons of this
        //         block (they may start from a cell of their own block)
        FORV_THREADS(t) {
            const int wv = t >> 6, lane = t & 63;
            dmaWait();
            _Pragma("unroll") for (int k = 0; k < 2; k++)
            if (iFl[k][TI] & 1) {
                const uint32_t *pi = &L.pItem[wv][k][lane * 4];
                const double te = dwPair((*lp(&pi[0])), (*lp(&pi[1])));
                const uint32_t kp = (*lp(&pi[2])), src = (*lp(&pi[3]));
                const int dj = (int)(kp >> (KEY_BITS + 7)), s2 = (int)((kp >> KEY_BITS) & 127);
                if (iFl[k][TI] & 2) enqueue(qx, te, (int)(kp & KEY_MASK) - KEY_BIAS, dj, s2, (int)(src & 127u), (uint32_t)t + (uint32_t)k * VNT >= cntNonRTN ? 2 : 0);
                else {
                    const double pv = dwPair((*lp(&L.pItemPv[wv][k][0][lane])), (*lp(&L.pItemPv[wv][k][1][lane])));
                    if (pv > AUGX_NINF) ldsMaxD(&L.cmax[nxt][dj][s2], pv + te);
                }
                iFl[k][TI] = 0;
            }
        }
        FORV_WAVES(w) {
            if (w >= UW0 && aheadUtr) {
                for (int k = 0; k < VSL; k++)
                    if (WX2(myD, k) >= 0) {
                        const UDesc D = *(const UDesc *)&L.ud[nxt][WX2(myD, k) * UDW];
                        FOR_WLANES(t, w) { utrL3(w, t, k, D, jb, jbN, nxt, qx); }
                    }
                // (a block with more units than slots: the others one after the other, level after level, through slot 0)
                uint32_t wi = 0;
                for (uint32_t d = 0; d < ndN; d++) {
                    const int total = uni(((const UDesc *)&L.ud[nxt][d * UDW])->total);
                    const int nun = (total + UH * WAVE - 1) / (UH * WAVE);
                    if (wi + (uint32_t)nun > (uint32_t)(VSL * NUW)) {
                        const UDesc D = *(const UDesc *)&L.ud[nxt][d * UDW];
                        for (int c = 0; c < nun; c++) {
                            const uint32_t u = wi + (uint32_t)c;
                            if ((int)(u % NUW) != w - UW0 || u / NUW < (uint32_t)VSL) continue;
                            FOR_WLANES(t, w) { utrL1(w, t, 0, D, c); dmaWait(); utrL2(w, t, 0, D, jb); dmaWait(); utrL3(w, t, 0, D, jb, jbN, nxt, qx); }
                        }
                    }
                    wi += (uint32_t)nun;
                }
            }
        }
        if (sync) itemPass(cntNonRT, cntAll);
        else queuePass(true);
        BLOCK_SYNC();
        VPROF(8);
        FORV_THREADS(t) {
            if (t >= WAVE) cellsOf(t, 1 << 2);
            if (t == 0) { (*lp(&L.qn[qc])) = 0; (*lp(&L.needSync[qc])) = 0; } // (this block's queue: filled again while block b + 1 runs)
        }
        BLOCK_SYNC();
        VPROF(9);
        // the columns of this block reach HBM before any later block reads them from there (the ring covers 64 bases)
        if (((b + 1) * BLK) % 32 == 0) BLOCK_GLOBAL_SYNC();
        VPROF(10);
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    if (B.prof && threadIdx.x == WAVE) for (int k = 0; k < 12; k++) B.prof[(int64_t)p * 56 + k] = dpAcc[k];
#endif
    BLOCK_SYNC();
    FORV_THREADS(t) { // termination (reference NAMGene::getViterbiPath, src/namgene.cc:442-462)
        if (t == 0) {
            double tot = AUGX_NINF;
            int fin = -1;
            for (int i = 0; i < S; i++) {
                const double tl = termKind == 0 ? T.ln_term[i] : (i == synch ? 0.0 : AUGX_NINF);
                const double v = (*lp(&L.ring[(n - 1) & 63][i])) + tl;
                if (!(v > AUGX_NINF)) continue;
                if (v > tot) { tot = v; fin = i; }
            }
            B.lnv[p] = tot; B.finalState[p] = fin; B.status[p] = fin < 0 ? AUGX_E_NOPATH : fabs(tot) < AUGX_EXACT_LIMIT ? 0 : AUGX_E_RANGE;
        }
    }
}

} // namespace dev
} // namespace augx
