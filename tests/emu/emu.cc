// emu.cc -- lane-loop emulator of the HIP decode kernels.  TEST INFRASTRUCTURE ONLY.
//
// Compiles augustus_amd/csrc/device/kernels.h with -DAUGX_EMU: every kernel body runs on the CPU with the
// 64 lanes of a wavefront executed one after the other, the grid as nested loops and the prefix scans as
// plain sequential loops.  It exists because the build container has no GPU: it lets the CPU test-suite
// exercise the very same source the device runs (bit-identical arithmetic, -ffp-contract=off).  It is not a
// fallback: libaugx never links it and the product path fails without a HIP device.
#define AUGX_EMU 1
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "../../augustus_amd/csrc/device/kernels.h"
#include "../../augustus_amd/csrc/device/dense.h"
#include "../../augustus_amd/csrc/device/assmemo.h"
#include "../../augustus_amd/csrc/device/layout.h"
#include "../../augustus_amd/csrc/device/sampler.h"
#include "../../augustus_amd/csrc/device/snipmemo.h"
static std::vector<int32_t> g_nearTies; // near ties on the chosen path of every piece of the last decode (dp.h: AUGX_NEAR_TIE)

using namespace augx::dev;

namespace {
template <class T> T *zalloc(int64_t n) { return (T *)calloc((size_t)(n > 0 ? n : 1), sizeof(T)); }

// piece-local inclusive scans over the AoS field arrays (the device does the same with chunked kernels)
template <bool MAX, class E> void scanFields(E *a, int nf, const BatchLayout &L) {
    for (int p = 0; p < L.nPieces; p++) {
        std::vector<E> acc(nf, 0);
        for (int64_t g = L.off[p]; g < L.off[p + 1]; g++)
            for (int f = 0; f < nf; f++) {
                E v = a[fidx(g, f, nf)];
                acc[f] = MAX ? (v > acc[f] ? v : acc[f]) : acc[f] + v;
                a[fidx(g, f, nf)] = acc[f];
            }
    }
}
} // namespace



// posterior sampling in the emulator: emu_set_sampling(n, seed) before emu_decode(..., fwd_out != NULL); the generator lives on
// across emu_decode calls (one stream over the run, as in the reference); emu_sample_get reads sample `it` of piece p
static int g_nsamples = 0;
static std::vector<double> g_tss0; // emu_set_tss0: [n][2] for the next emu_decode (BatchView::tss0), empty: none
static augx_rand *g_rand = nullptr;
static std::vector<std::vector<std::vector<augx_state>>> g_samples;

// ---- models decoded by the dense kernels (device/dense.h: the 71-state model with UTR states): the same prep kernels, the UTR
//      prefix / signal / site-list kernels, the candidate records of kCand in their dense form, densePiece, denseBacktracePiece
static int emu_decode_dense(const augx_tables *t, const augx_piece *pieces, int n, double *lnv, int32_t *status, int32_t *path_out,
                            int32_t path_cap, int32_t *path_n, double *cells_out, int32_t *cls_out, double *fwd_out, double *lnfwd_out) {
    int blk = 4;
    try {
        blk = chooseDenseBlock(*t);
    } catch (std::exception &e) {
        fprintf(stderr, "emu: %s\n", e.what());
        return AUGX_E_UNSUPPORTED;
    }
    DevTables T;
    try { fillDevTablesScalars(*t, T); } catch (std::exception &e) { fprintf(stderr, "emu: %s\n", e.what()); return AUGX_E_UNSUPPORTED; }
    for (auto &sp : tableSpans(*t, T)) *sp.dst = sp.src;
    BatchLayout L;
    L.build(pieces, n);
    BatchSizes Z(L);
    BatchView B;
    memset(&B, 0, sizeof B);
    B.tss0 = (int)g_tss0.size() == 2 * n ? g_tss0.data() : nullptr;
    B.nPieces = n; B.N = L.N; B.nChunks = L.nChunks;
    B.off = L.off.data(); B.len = L.len.data(); B.initKind = L.initKind.data(); B.termKind = L.termKind.data();
    B.chunkPiece = L.chunkPiece.data();
    std::vector<int32_t> cls(n, -1), clsMM(2 * n);
    B.cls = cls.data(); B.clsMinMax = clsMM.data();
    std::vector<void *> bufs;
    auto za = [&](int64_t count, size_t elem) { void *p = calloc((size_t)(count > 0 ? count : 1), elem); bufs.push_back(p); return p; };
    char *raw = (char *)za(Z.N, 1);
    for (int p = 0; p < n; p++) memcpy(raw + L.off[p] + 1, pieces[p].seq, (size_t)L.len[p]);
    B.raw = raw;
    B.code = (uint8_t *)za(Z.N, 1);
    B.cnt = (uint32_t *)za(Z.N * NCNT, 4);
    B.nsm = (uint32_t *)za(Z.N * 6, 4);
    B.sig = (double *)za(Z.N * NSIG, 8);
    B.gate = (uint64_t *)za(Z.N, 8);
    B.site = (int32_t *)za(Z.N * NSITE, 4);
    B.cells = (double *)za(Z.N * t->S, 8);
    B.bpD = (uint8_t *)za(Z.N * t->S, 1);
    B.gcRaw = (uint8_t *)za(Z.N, 1); B.gcPlane = (uint8_t *)za(Z.N, 1);
    B.ufx = (uint64_t *)za(Z.N * NUFX, 8); B.ucnt = (uint32_t *)za(Z.N * NUCNT, 4); B.usig = (double *)za(Z.N * NUSIG, 8);
    std::vector<int32_t> nPlanes(n, 1), planeCls((size_t)n * MAXPL, 0);
    B.nPlanes = nPlanes.data(); B.planeCls = planeCls.data();
    B.nPl = 1;
    std::vector<int32_t> listCnt(n);
    std::vector<int64_t> listOffs;
    B.listCnt = listCnt.data();
    std::vector<double> lnvv(n);
    std::vector<int32_t> st(n), fin(n), pc(n);
    B.lnv = lnvv.data(); B.status = st.data(); B.finalState = fin.data(); B.pathCount = pc.data();
    B.pathRec = (int32_t *)za(Z.pathCap * 3, 4);
    g_nearTies.assign((size_t)n, 0);
    B.nearTie = g_nearTies.data();
    for (int64_t g = 0; g < B.N; g++) k1Encode(B, g);
    for (int64_t g = 0; g < B.N; g++) k1SiteTerms(T, B, g);
    scanFields<false>(B.cnt, NCNT, L);
    scanFields<true>(B.nsm, 6, L);
    for (int p = 0; p < n; p++) { clsMM[2 * p] = 1 << 30; clsMM[2 * p + 1] = -1; }
    for (int64_t g = 0; g < B.N; g++) {
        int c = k1WindowClass(T, B, g);
        if (c >= 0) {
            int p = B.chunkPiece[g / CHUNK];
            if (c < clsMM[2 * p]) clsMM[2 * p] = c;
            if (c > clsMM[2 * p + 1]) clsMM[2 * p + 1] = c;
        }
    }
    for (int p = 0; p < n; p++) {
        cls[p] = clsMM[2 * p] == clsMM[2 * p + 1] ? clsMM[2 * p] : -1;
        planeCls[(size_t)p * MAXPL] = cls[p];
        if (cls[p] < 0) {
            std::vector<uint8_t> plane;
            const int np = stairsPlanes(B.gcRaw + L.off[p] + 1, L.len[p], t->gc_win, plane, &planeCls[(size_t)p * MAXPL]);
            if (np < 0) continue;
            cls[p] = planeCls[(size_t)p * MAXPL];
            nPlanes[p] = np;
            if (np > 1) memcpy(B.gcPlane + L.off[p] + 1, plane.data(), (size_t)L.len[p]);
            if (np > B.nPl) B.nPl = np;
        }
    }
    // UTR content prefix sums and site counts (one scan), then the list sizes
    for (int64_t g = 0; g < B.N; g++) k1UtrTerms(T, B, g);
    scanFields<false>(B.ufx, NUFX, L);
    scanFields<false>(B.ucnt, NUCNT, L);
    for (int p = 0; p < n; p++) { k1ListCount(B, p); k1UtrListCount(B, p); }
    const int64_t listCap = listOffsets(listCnt.data(), n, listOffs);
    B.listOffs = listOffs.data(); B.listCap = listCap;
    B.laPos = (int32_t *)za(listCap, 4); B.lrPos = (int32_t *)za(listCap, 4); B.atgPos = (int32_t *)za(listCap, 4);
    B.rsPos = (int32_t *)za(listCap, 4); B.rsBegin = (double *)za(listCap, 8);
    B.tfSite = (USite *)za(listCap, sizeof(USite)); B.laSite = (USite *)za(listCap, sizeof(USite)); B.fsSite = (USite *)za(listCap, sizeof(USite));
    B.lrSite = (USite *)za(listCap, sizeof(USite)); B.tmSite = (USite *)za(listCap, sizeof(USite)); B.rtSite = (USite *)za(listCap, sizeof(USite));
    const int64_t nPl = B.nPl;
    B.fx = (uint64_t *)za(nPl * Z.N * NFX, 8);
    B.plsR = (double *)za(nPl * Z.N * 3, 8);
    B.ldEnt = (IntronStart *)za(nPl * listCap, sizeof(IntronStart)); B.rdEnt = (IntronStart *)za(nPl * listCap, sizeof(IntronStart));
    B.laPls = (double *)za(nPl * listCap * 3, 8); B.laFx = (uint64_t *)za(nPl * listCap * 3, 8);
    B.lrEt = (double *)za(nPl * listCap * 3, 8); B.lrFx = (uint64_t *)za(nPl * listCap * 3, 8);
    B.atgD = (double *)za(nPl * listCap * 3, 8); B.atgFx = (uint64_t *)za(nPl * listCap, 8);
    B.rsFx = (uint64_t *)za(nPl * listCap * 3, 8);
    for (int pl = 0; pl < nPl; pl++) {
        for (int64_t g = 0; g < B.N; g++) k1FxTerms(T, B, g, pl);
        scanFields<false>(B.fx + (int64_t)pl * Z.N * NFX, NFX, L);
    }
    for (int64_t g = 0; g < B.N; g++) k1Signals(T, B, g);
    for (int sel = 0; sel < 4; sel++)
        for (int64_t t2 = 0; t2 < B.listCap; t2++) k1SiteSignals(T, B, t2, sel);
    for (int pl = 0; pl < B.nPl; pl++)
        for (int64_t g = 0; g < B.N; g++) k1SiteConsts(T, B, g, pl);
    for (int64_t g = 0; g < B.N; g++) k1UtrSignals(T, B, g);
    // candidate records of the coding exons and short introns, in their dense form
    B.blk = blk;
    B.nBlk = B.N / blk;
    B.blkCnt = (uint32_t *)za(B.nBlk * 2, 4);
    B.blkSplit = (uint32_t *)za(B.nBlk * 3, 4);
    B.blkOff = (uint64_t *)za(B.nBlk * 2, 8);
    CandAlloc ca;
    B.candAlloc = &ca;
    CandLds *cl = new CandLds();
    const int64_t nWg = B.N / (WAVE * NWAVES);
    B.itemCap = 64;
    B.items = zalloc<Item>(B.itemCap + 1);
    for (int attempt = 0; attempt < 2; attempt++) {
        ca.pairs = 0; ca.items = 0; ca.descs = 0;
        for (int64_t wg = 0; wg < nWg; wg++) {
            if (B.nPl > 1) { if (blk == 8) candWorkgroup<8, true, true>(T, B, *cl, wg); else if (blk == 4) candWorkgroup<4, true, true>(T, B, *cl, wg); else candWorkgroup<2, true, true>(T, B, *cl, wg); }
            else { if (blk == 8) candWorkgroup<8, false, true>(T, B, *cl, wg); else if (blk == 4) candWorkgroup<4, false, true>(T, B, *cl, wg); else candWorkgroup<2, false, true>(T, B, *cl, wg); }
        }
        if ((int64_t)ca.items <= B.itemCap) break;
        free(B.items);
        B.itemCap = (int64_t)ca.items;
        B.items = zalloc<Item>(B.itemCap + 1);
    }
    delete cl;
    // descriptors of the UTR exon cells (kUtrDesc): counted, then written
    B.udOff = (uint64_t *)za(B.nBlk, 8);
    B.udCnt = (uint32_t *)za(B.nBlk, 4);
    B.udCap = 0;
    B.ud = nullptr;
    if (T.utr) {
        UDescLds *ul = new UDescLds();
        const int64_t nGrp = B.N / (NT / 16);
        for (int attempt = 0; attempt < 2; attempt++) {
            ca.descs = 0;
            for (int64_t wg = 0; wg < nGrp; wg++) { if (blk == 8) utrDescGroup<8>(T, B, *ul, wg); else if (blk == 4) utrDescGroup<4>(T, B, *ul, wg); else utrDescGroup<2>(T, B, *ul, wg); }
            if ((int64_t)ca.descs <= B.udCap) break;
            B.udCap = (int64_t)ca.descs;
            B.ud = (UDesc *)za(B.udCap, sizeof(UDesc));
        }
        delete ul;
        if (getenv("AUGX_EMU_STATS")) {
            long long cands = 0, chunks = 0, byKind[40] = {0}, dk[40] = {0};
            for (int64_t i = 0; i < (int64_t)ca.descs; i++) { cands += B.ud[i].total; chunks += (B.ud[i].total + 63) / 64; byKind[B.ud[i].kind - AUGX_K_UTR5SINGLE] += B.ud[i].total; dk[B.ud[i].kind - AUGX_K_UTR5SINGLE]++; }
            fprintf(stderr, "emu stats (UTR): %lld descriptors, %lld candidates, %lld chunks over N=%lld\n", (long long)ca.descs, cands, chunks, (long long)B.N);
            for (int k = 0; k < 40; k++) if (dk[k]) fprintf(stderr, "   kind %d: %lld descriptors, %lld candidates\n", k + AUGX_K_UTR5SINGLE, dk[k], byKind[k]);
        }
    }
    if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu stats (dense): N=%lld block %d pairs=%lld records=%lld\n", (long long)B.N, blk, (long long)ca.pairs, (long long)ca.items);
    std::vector<int32_t> seg0(n + 1);
    for (int p = 0; p <= n; p++) seg0[p] = p;
    B.pieceSeg0 = seg0.data(); B.nSegs = n;
    DenseLds *dl = new DenseLds();
    // the reference's snippet cache around the class steps of a piece (snipmemo.h), from which cells of the matrix `mat` are alive:
    // the candidate terms concerned are rebuilt in place; true: some were, the pass has to run once more
    auto snippetReplay = [&](int p, const double *mat) -> bool {
        if (B.nPlanes[p] <= 1) return false;
        SnippetReplay R;
        const int len = L.len[p], S = t->S;
        const int64_t o = L.off[p];
        R.t = t; R.n = len; R.S = S; R.blk = blk; R.d = t->d; R.dense = true;
        R.F = mat + (o + 1) * S;
        R.plane = B.gcPlane + o + 1;
        R.planeCls = B.planeCls + (int64_t)p * MAXPL;
        R.nPlanes = B.nPlanes[p];
        const int nBlocks = (len + blk - 1) / blk;
        const int64_t gb0 = o / blk;
        R.blkOff = B.blkOff + gb0 * 2; R.blkCnt = B.blkCnt + gb0 * 2;
        uint64_t lo = ~0ull;
        for (int q = 0; q < nBlocks; q++) if (R.blkCnt[(size_t)q * 2 + 1] && R.blkOff[(size_t)q * 2 + 1] < lo) lo = R.blkOff[(size_t)q * 2 + 1];
        if (lo == ~0ull) lo = 0;
        R.item0 = lo; R.items = B.items + lo;
        R.fxF.assign((size_t)R.nPlanes, {}); R.fxR.assign((size_t)R.nPlanes, {});
        for (int pl = 0; pl < R.nPlanes; pl++) {
            R.fxF[pl].resize((size_t)len + 1); R.fxR[pl].resize((size_t)len + 1);
            const uint64_t *fx = B.fx + (int64_t)pl * B.N * NFX;
            for (int g = 0; g <= len; g++) { R.fxF[pl][g] = fx[fidx(o + g, FX_INF, NFX)]; R.fxR[pl][g] = fx[fidx(o + g, FX_INR, NFX)]; }
        }
        R.run();
        if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu stats (dense): piece %d: %zu candidate terms rebuilt from the reference's snippet cache\n", p, R.patches.size());
        return !R.patches.empty();
    };
    // the two other call-history caches of the reference, read by UTR states only (tssProbsPlus: dense.h k1TssReplay; the aSSProb memo:
    // assmemo.h): the values concerned are rebuilt in place; true: some were -- the descriptors hold values made from them (kUtrDesc
    // again), and the pass has to run once more
    std::vector<LaSw> laSw;
    std::vector<AssSwIn> laSwIn;
    std::vector<char> memoDone((size_t)n, 0);
    std::vector<std::shared_ptr<AssMemoReplay>> memoOf((size_t)n);
    auto memoReplay = [&](int p, const double *mat) -> bool {
        if (B.nPlanes[p] <= 1 || !T.utr || memoDone[(size_t)p]) return false;
        memoDone[(size_t)p] = 1;
        const int len = L.len[p], S = t->S;
        const int64_t o = L.off[p];
        const double *M = mat + (o + 1) * S;
        int nTss = 0;
        const int nTf = (int)B.ucnt[fidx(o + len, UCNT_TF, NUCNT)];
        for (int li = 0; li < nTf; li++) nTss += k1TssReplay(T, B, p, li, M);
        memoOf[(size_t)p] = std::make_shared<AssMemoReplay>();
        AssMemoReplay &R = *memoOf[(size_t)p];
        R.T = &T; R.n = len; R.plane = B.gcPlane + o + 1;
        R.requesters();
        const int nList = (int)B.cnt[fidx(o + len, CNT_LA, NCNT)];
        for (int li = 0; li < nList + T.Ae; li++) {
            uint8_t alive;
            const int q = memoAssSite(T, B, p, li, M, R.reqS, R.nReq, alive);
            if (q < 0) continue;
            R.siteQ.push_back(q); R.siteLi.push_back(li); R.siteAlive.push_back(alive);
        }
        R.gateOwn.resize((size_t)len);
        for (int j = 0; j < len; j++) R.gateOwn[(size_t)j] = memoGateBits(T, B, p, j, R.reqS, R.nReq);
        R.gate = R.gateOwn.data();
        R.run(getenv("AUGX_MEMO_SLOW") != nullptr);
        std::vector<AssPatch> pt;
        const size_t sw0 = laSwIn.size();
        const int extras = R.patches(nList, pt, laSwIn);
        laSw.resize(laSwIn.size());
        B.laSw = laSw.data();
        for (const AssPatch &A : pt) k1AssPatch(T, B, p, A, laSwIn.data(), laSw.data());
        if (getenv("AUGX_EMU_STATS"))
            fprintf(stderr, "emu stats (dense): piece %d: %d TSS windows and %zu acceptor sites (%zu changes of value during the sweep, %d sites past the end left) rebuilt from the reference's caches; aSSProb memo: %lld calls walked, emptied %lld times\n",
                    p, nTss, pt.size(), laSwIn.size() - sw0, extras, R.calls, R.flushes);
        return nTss > 0 || !pt.empty();
    };
    auto describeAgain = [&]() { // (kUtrDesc: the leading candidates of a descriptor are evaluated there, from the site values)
        if (!T.utr) return;
        UDescLds *ul = new UDescLds();
        const int64_t nGrp = B.N / (NT / 16);
        ca.descs = 0;
        for (int64_t wg = 0; wg < nGrp; wg++) { if (blk == 8) utrDescGroup<8>(T, B, *ul, wg); else if (blk == 4) utrDescGroup<4>(T, B, *ul, wg); else utrDescGroup<2>(T, B, *ul, wg); }
        delete ul;
    };
    const bool exact = !getenv("AUGX_EXACT_MULTICLASS") || atoi(getenv("AUGX_EXACT_MULTICLASS")) != 0; // (augx_decoder_set_exact, on by default)
    auto viterbiPiece = [&](int p) {
        if (blk == 8) densePiece<8, 0, true>(T, B, *dl, p); else if (blk == 4) densePiece<4, 0, true>(T, B, *dl, p); else densePiece<2, 0, true>(T, B, *dl, p);
    };
    for (int p = 0; p < n; p++) {
        viterbiPiece(p);
        if (exact) {
            const bool memo = !getenv("AUGX_NO_ASSMEMO") && memoReplay(p, B.cells);
            if (memo) describeAgain();
            if (snippetReplay(p, B.cells) || memo) viterbiPiece(p);
        }
        denseBacktracePiece(T, B, p);
    }
    if (fwd_out) {
        B.fwd = (double *)za(Z.N * t->S, 8);
        std::vector<double> lnF(n);
        B.lnFwd = lnF.data();
        auto fwdPiece = [&](int p) { if (blk == 8) densePiece<8, 1>(T, B, *dl, p); else if (blk == 4) densePiece<4, 1>(T, B, *dl, p); else densePiece<2, 1>(T, B, *dl, p); };
        for (int p = 0; p < n; p++) {
            fwdPiece(p);
            if (!getenv("AUGX_NO_MEMO")) { // (the forward pass always replays; the site values may have been rebuilt by the Viterbi pass already)
                const bool memo = !getenv("AUGX_NO_ASSMEMO") && memoReplay(p, B.fwd);
                if (memo) describeAgain();
                if (snippetReplay(p, B.fwd) || memo) fwdPiece(p);
            }
        }
        int64_t w = 0;
        for (int p = 0; p < n; p++) {
            memcpy(fwd_out + w, B.fwd + (L.off[p] + 1) * t->S, sizeof(double) * (size_t)L.len[p] * t->S);
            w += (int64_t)L.len[p] * t->S;
            if (lnfwd_out) lnfwd_out[p] = lnF[p];
        }
        g_samples.assign((size_t)n, {});
        for (int p = 0; p < n && g_nsamples > 0; p++) {
            SamplePiece P;
            const int len = L.len[p], S = t->S;
            const int64_t o = L.off[p];
            P.t = t; P.S = S; P.n = len; P.blk = blk; P.cls0 = cls[p]; P.nPlanes = B.nPlanes[p]; P.termKind = L.termKind[p];
            P.dense = true; P.hT = &T; P.hB = &B; P.hp = p;
            if (memoOf[(size_t)p] && !getenv("AUGX_NO_LATE_MEMO")) { // (the aSSProb memo lives on through the back-tracking and the sampled paths)
                P.memo = memoOf[(size_t)p].get();
                const int64_t po = pathOff(B, p);
                for (int i = pc[p] - 1; i >= 0; i--) { const int32_t *r = B.pathRec + (po + i) * 3; P.vitPath.push_back({r[0], r[1], (int16_t)r[2], (int16_t)t->state_type[r[2]]}); }
            }
            P.F = B.fwd + (o + 1) * S;
            P.sig.assign(B.sig + (o + 1) * NSIG, B.sig + (o + 1 + len) * NSIG);
            if (P.nPlanes > 1) {
                P.plane.assign(B.gcPlane + o + 1, B.gcPlane + o + 1 + len);
                P.planeCls.assign(B.planeCls + (int64_t)p * MAXPL, B.planeCls + (int64_t)(p + 1) * MAXPL);
            }
            const int nBlocks = (len + blk - 1) / blk;
            const int64_t gb0 = o / blk;
            P.blkOff.assign(B.blkOff + gb0 * 2, B.blkOff + (gb0 + nBlocks) * 2);
            P.blkCnt.assign(B.blkCnt + gb0 * 2, B.blkCnt + (gb0 + nBlocks) * 2);
            P.item0 = P.blkOff[1];
            const uint64_t itemEnd = P.blkOff[(size_t)(nBlocks - 1) * 2 + 1] + P.blkCnt[(size_t)(nBlocks - 1) * 2 + 1];
            P.items.assign(B.items + P.item0, B.items + itemEnd);
            P.anyNuc = false;
            for (int q = 0; q < len && !P.anyNuc; q++) P.anyNuc = B.code[o + 1 + q] < 4;
            std::vector<int> sst;
            const auto ts0 = std::chrono::steady_clock::now();
            prepareStops(P);
            const auto ts1 = std::chrono::steady_clock::now();
            samplePaths(P, g_nsamples, *g_rand, g_samples[p], sst);
            if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu sampler: piece %d (%d bases): stops %.3f s, %d paths drawn in %.3f s (the generator's buffers so far: %.3f s)\n", p, P.n, std::chrono::duration<double>(ts1 - ts0).count(), g_nsamples,
                                                  std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count(), g_rand->refillSeconds);
            if (getenv("AUGX_EMU_STATS") && P.memo) fprintf(stderr, "emu sampler: piece %d: aSSProb memo carried on: %lld calls in all, emptied %lld times; candidates of the Viterbi path's UTR exon steps the back-tracking values under another class: %ld\n",
                                                            p, P.memo->calls, P.memo->flushes, P.memoVitDiffs);
        }
    }
    delete dl;
    for (int p = 0; p < n; p++) {
        lnv[p] = lnvv[p];
        status[p] = st[p];
        if (cls_out) cls_out[p] = cls[p];
        int cnt = pc[p];
        path_n[p] = cnt;
        int64_t po = pathOff(B, p);
        for (int i = 0; i < cnt && i < path_cap; i++) {
            const int32_t *r = B.pathRec + (po + (cnt - 1 - i)) * 3;
            int32_t *o = path_out + ((int64_t)p * path_cap + i) * 3;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
        }
    }
    if (cells_out) {
        int64_t w = 0;
        for (int p = 0; p < n; p++) {
            memcpy(cells_out + w, B.cells + (L.off[p] + 1) * t->S, sizeof(double) * (size_t)L.len[p] * t->S);
            w += (int64_t)L.len[p] * t->S;
        }
    }
    free(B.items);
    for (void *p : bufs) free(p);
    return 0;
}

extern "C" {
long long emu_jump_probes() { return g_emuJumpProbes; }
long long emu_jumps() { return g_emuJumps; }
long long emu_quiet_checks() { return g_emuQuietChecks; }
long long emu_jump_tiles() { return g_emuJumpTiles; }   // (tests: runs of N were jumped over)
long long emu_quiet_tiles() { return g_emuQuietTiles; } // (tests: the chain-only path of trellisPiece was taken)
int emu_near_ties(int p) { return p >= 0 && p < (int)g_nearTies.size() ? g_nearTies[p] : -1; }
// values of the TSS window at base 0 of the pieces of the NEXT emu_decode, [n][2] forward / reverse, NaN: the piece's own (n = 0: none)
void emu_set_tss0(const double *v, int n) { g_tss0.assign(v, v + (v ? 2 * (size_t)n : 0)); }
void emu_set_sampling(int n, unsigned seed) {
    g_nsamples = n;
    delete g_rand;
    g_rand = n > 0 ? new augx_rand(seed) : nullptr;
}
int emu_state_type(const augx_tables *t, int s) { return s >= 0 && s < t->S ? t->state_type[s] : -1; }
// stayThreshold (sampler.h) against the expression it stands for: for n random (total, p0) pairs the draws below the threshold take the
// first option and the threshold itself does not; returns the number of pairs where that fails
int emu_stay_threshold_check(unsigned seed, int n) {
    augx_rand R(seed);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        // totals between 1 and 3 (the largest option has probability 1 by construction), p0 anywhere in (0, total]; some pairs at the edges
        const double cum = 1.0 + 2.0 * (double)R.next() / 2147483647.0;
        double p0 = (i % 7 == 0) ? cum : (i % 11 == 0) ? cum * 0.99999 : cum * (double)(R.next() + 1) / 2147483648.0;
        if (i % 13 == 0) p0 = 1e-300;
        const uint32_t thr = augx::dev::stayThreshold(cum, p0);
        auto takesFirst = [&](uint32_t r) { return (double)(int)r / 2147483647.0 * cum * 0.99999 < p0; };
        if (thr > 0 && !takesFirst(thr - 1)) bad++;
        if (thr < 2147483648u && takesFirst(thr)) bad++;
        if (thr > 1 && !takesFirst(thr / 2)) bad++; // (monotone: everything below takes it)
    }
    return bad;
}
// augx_rand against glibc's rand() after srand(seed): n draws, every `stride`-th one looked at, the others spent with skip();
// returns the index of the first draw that differs, -1: none
long long emu_rand_check(unsigned seed, long long n, int stride) {
    augx_rand R(seed);
    srand(seed);
    for (long long i = 0; i < n;) {
        const int gap = stride > 1 ? (int)((i * 7 + 3) % stride) : 0;
        for (int q = 0; q < gap; q++) (void)rand();
        R.skip(gap);
        i += gap;
        if (R.next() != rand()) return i;
        i++;
    }
    return -1;
}
int emu_sample_get(int p, int it, int32_t *out, int cap) {
    if (p < 0 || p >= (int)g_samples.size() || it < 0 || it >= (int)g_samples[p].size()) return -1;
    const std::vector<augx_state> &v = g_samples[p][it];
    for (int i = 0; i < (int)v.size() && i < cap; i++) { out[i * 3] = v[i].begin; out[i * 3 + 1] = v[i].end; out[i * 3 + 2] = v[i].type; }
    return (int)v.size();
}
// decode a batch on the emulator.  Outputs: lnv[n], status[n]; paths as (begin,end,state) triples in 5'->3'
// order into path_out (capacity path_cap triples per piece, counts in path_n); cells_out (optional) receives
// the dense ln V matrices piece after piece (len*S doubles each).
// fwd_out (optional): the dense ln F matrices of the forward algorithm, piece after piece; lnfwd_out[n]: ln P(sequence)
int emu_decode(const augx_tables *t, const augx_piece *pieces, int n, double *lnv, int32_t *status, int32_t *path_out,
               int32_t path_cap, int32_t *path_n, double *cells_out, int32_t *cls_out, double *fwd_out, double *lnfwd_out) {
    if (modelIsDense(*t)) return emu_decode_dense(t, pieces, n, lnv, status, path_out, path_cap, path_n, cells_out, cls_out, fwd_out, lnfwd_out);
    int blk = 8;
    try {
        blk = chooseBlockSize(*t);
    } catch (std::exception &e) {
        fprintf(stderr, "emu: %s\n", e.what());
        return AUGX_E_UNSUPPORTED;
    }
    DevTables T;
    fillDevTablesScalars(*t, T);
    for (auto &sp : tableSpans(*t, T)) *sp.dst = sp.src;
    BatchLayout L;
    L.build(pieces, n);
    BatchSizes Z(L);
    BatchView B;
    memset(&B, 0, sizeof B);
    B.nPieces = n; B.N = L.N; B.nChunks = L.nChunks;
    B.off = L.off.data(); B.len = L.len.data(); B.initKind = L.initKind.data(); B.termKind = L.termKind.data();
    B.chunkPiece = L.chunkPiece.data();
    std::vector<int32_t> cls(n, -1), clsMM(2 * n);
    B.cls = cls.data(); B.clsMinMax = clsMM.data();
    char *raw = zalloc<char>(Z.N);
    for (int p = 0; p < n; p++) memcpy(raw + L.off[p] + 1, pieces[p].seq, (size_t)L.len[p]);
    B.raw = raw;
    B.code = zalloc<uint8_t>(Z.N);
    B.cnt = zalloc<uint32_t>(Z.N * NCNT);
    B.nsm = zalloc<uint32_t>(Z.N * 6);
    B.sig = zalloc<double>(Z.N * NSIG);
    B.gate = zalloc<uint64_t>(Z.N);
    B.site = zalloc<int32_t>(Z.N * NSITE);
    B.bp = zalloc<uint16_t>(Z.N * SP);
    B.bpChain = zalloc<uint8_t>(Z.N * 8);
    B.cells = cells_out ? zalloc<double>(Z.N * t->S) : nullptr;
    B.vig = zalloc<double>(Z.N);
    B.longV = zalloc<double>(Z.N * 6);
    B.gcRaw = zalloc<uint8_t>(Z.N); B.gcPlane = zalloc<uint8_t>(Z.N);
    std::vector<int32_t> nPlanes(n, 1), planeCls((size_t)n * MAXPL, 0);
    B.nPlanes = nPlanes.data(); B.planeCls = planeCls.data();
    B.nPl = 1;
    std::vector<int32_t> listCnt(n);
    std::vector<int64_t> listOffs;
    B.listCnt = listCnt.data();
    std::vector<double> lnvv(n);
    std::vector<int32_t> st(n), fin(n), pc(n);
    B.lnv = lnvv.data(); B.status = st.data(); B.finalState = fin.data(); B.pathCount = pc.data();
    B.pathRec = zalloc<int32_t>(Z.pathCap * 3);
    g_nearTies.assign((size_t)n, 0);
    B.nearTie = g_nearTies.data();

    // ---- K1
    for (int64_t g = 0; g < B.N; g++) k1Encode(B, g);
    for (int64_t g = 0; g < B.N; g++) k1SiteTerms(T, B, g);
    scanFields<false>(B.cnt, NCNT, L);
    scanFields<true>(B.nsm, 6, L);
    for (int p = 0; p < n; p++) { clsMM[2 * p] = 1 << 30; clsMM[2 * p + 1] = -1; }
    for (int64_t g = 0; g < B.N; g++) {
        int c = k1WindowClass(T, B, g);
        if (c >= 0) {
            int p = B.chunkPiece[g / CHUNK];
            if (c < clsMM[2 * p]) clsMM[2 * p] = c;
            if (c > clsMM[2 * p + 1]) clsMM[2 * p + 1] = c;
        }
    }
    for (int p = 0; p < n; p++) {
        cls[p] = clsMM[2 * p] == clsMM[2 * p + 1] ? clsMM[2 * p] : -1;
        planeCls[(size_t)p * MAXPL] = cls[p];
        if (cls[p] < 0) { // the windows disagree: content stairs of the piece, classes -> planes (as augx_batch_decode does)
            std::vector<uint8_t> plane;
            const int np = stairsPlanes(B.gcRaw + L.off[p] + 1, L.len[p], t->gc_win, plane, &planeCls[(size_t)p * MAXPL]);
            if (np < 0) continue;
            cls[p] = planeCls[(size_t)p * MAXPL];
            nPlanes[p] = np;
            if (np > 1) memcpy(B.gcPlane + L.off[p] + 1, plane.data(), (size_t)L.len[p]);
            if (np > B.nPl) B.nPl = np;
        }
    }
    // candidate lists: sized from the counted sites (as augx_batch_decode does)
    for (int p = 0; p < n; p++) k1ListCount(B, p);
    const int64_t listCap = listOffsets(listCnt.data(), n, listOffs);
    B.listOffs = listOffs.data(); B.listCap = listCap;
    B.laPos = zalloc<int32_t>(listCap); B.laVal = zalloc<double>(listCap * 3);
    B.lrPos = zalloc<int32_t>(listCap); B.lrVal = zalloc<double>(listCap * 3);
    B.ldVal = zalloc<double>(listCap * 3);
    B.rdVal = zalloc<double>(listCap * 3);
    B.atgPos = zalloc<int32_t>(listCap);
    B.rsPos = zalloc<int32_t>(listCap); B.rsBegin = zalloc<double>(listCap);
    // class-dependent arrays: one plane per class of the most varied piece
    const int64_t nPl = B.nPl;
    B.fx = zalloc<uint64_t>(nPl * Z.N * NFX);
    B.plsR = zalloc<double>(nPl * Z.N * 3);
    B.ldEnt = zalloc<IntronStart>(nPl * listCap); B.rdEnt = zalloc<IntronStart>(nPl * listCap);
    B.laPls = zalloc<double>(nPl * listCap * 3); B.laFx = zalloc<uint64_t>(nPl * listCap * 3);
    B.lrEt = zalloc<double>(nPl * listCap * 3); B.lrFx = zalloc<uint64_t>(nPl * listCap * 3);
    B.atgD = zalloc<double>(nPl * listCap * 3); B.atgFx = zalloc<uint64_t>(nPl * listCap);
    B.rsFx = zalloc<uint64_t>(nPl * listCap * 3);
    for (int pl = 0; pl < nPl; pl++) {
        for (int64_t g = 0; g < B.N; g++) k1FxTerms(T, B, g, pl);
        scanFields<false>(B.fx + (int64_t)pl * Z.N * NFX, NFX, L);
    }
    for (int64_t g = 0; g < B.N; g++) k1Signals(T, B, g);
    for (int sel = 0; sel < 4; sel++)
        for (int64_t t2 = 0; t2 < B.listCap; t2++) k1SiteSignals(T, B, t2, sel);
    for (int pl = 0; pl < B.nPl; pl++)
        for (int64_t g = 0; g < B.N; g++) k1SiteConsts(T, B, g, pl);
    // ---- K2a: candidates, tile by tile (first with buffers that are too small, to exercise the re-run path)
    B.blk = blk;
    B.nBlk = B.N / blk;
    B.blkCnt = zalloc<uint32_t>(B.nBlk * 2);
    B.blkSplit = zalloc<uint32_t>(B.nBlk * 3);
    B.blkOff = zalloc<uint64_t>(B.nBlk * 2);
    B.tileMinEop = zalloc<int32_t>(Z.N / WAVE + 1); B.tileCross = zalloc<int32_t>(Z.N / WAVE + 1);
    CandAlloc ca;
    B.candAlloc = &ca;
    CandLds *cl = new CandLds();
    const int64_t nWg = B.N / (WAVE * NWAVES); // one wavefront per tile of 64 bases
    B.itemCap = 64;
    B.items = zalloc<Item>(B.itemCap + 1);
    for (int attempt = 0; attempt < 2; attempt++) {
        ca.pairs = 0; ca.items = 0;
        for (int64_t wg = 0; wg < nWg; wg++) {
            if (B.nPl > 1) { if (blk == 8) candWorkgroup<8, true>(T, B, *cl, wg); else if (blk == 4) candWorkgroup<4, true>(T, B, *cl, wg); else candWorkgroup<2, true>(T, B, *cl, wg); }
            else { if (blk == 8) candWorkgroup<8, false>(T, B, *cl, wg); else if (blk == 4) candWorkgroup<4, false>(T, B, *cl, wg); else candWorkgroup<2, false>(T, B, *cl, wg); }
        }
        if ((int64_t)ca.items <= B.itemCap) break;
        free(B.items);
        B.itemCap = (int64_t)ca.items;
        B.items = zalloc<Item>(B.itemCap + 1);
    }
    delete cl;
    if (getenv("AUGX_EMU_STATS")) {
        int64_t dead = 0, byTag[3] = {0, 0, 0};
        for (int64_t i = 0; i < (int64_t)ca.items; i++) { if (!(B.items[i].te > AUGX_NINF)) dead++; else byTag[B.items[i].src >> 30]++; }
        fprintf(stderr, "emu stats: N=%lld pairs=%lld items=%lld dead=%lld live list=%lld vig=%lld col0=%lld\n", (long long)B.N, (long long)ca.pairs,
                (long long)B.itemCap, (long long)dead, (long long)byTag[0], (long long)byTag[1], (long long)byTag[2]);
        fprintf(stderr, "emu stats: general-path evaluations: igenic-pred %lld, list exon %lld\n", g_emuSlowA, g_emuSlowB);
        {   // how often does a trellis worker need a second chunk of 64 candidates?  (pair-aligned split over k workers)
            long long hist[3][6] = {{0}};
            for (int64_t gb = 0; gb < B.nBlk; gb++) {
                const uint64_t i0 = B.blkOff[gb * 2 + 1];
                const uint32_t nonRT = B.blkSplit[gb * 3 + 2];
                if (B.blkCnt[gb * 2] == 0) continue;
                std::vector<uint32_t> bnd; // inclusive prefix at pair boundaries
                uint32_t run = 0;
                for (uint32_t i = 0; i < nonRT; i++) {
                    run++;
                    if (i + 1 == nonRT || (B.items[i0 + i].kp >> KEY_BITS) != (B.items[i0 + i + 1].kp >> KEY_BITS)) bnd.push_back(i + 1);
                }
                (void)run;
                for (int kw = 3; kw <= 5; kw++) {
                    uint32_t prev = 0; int worst = 0;
                    for (int q = 1; q <= kw; q++) {
                        uint32_t target = (uint32_t)((uint64_t)nonRT * q / kw), best = nonRT, bd = 0xffffffffu;
                        if (q == kw) best = nonRT;
                        else for (uint32_t x : bnd) { uint32_t d = x > target ? x - target : target - x; if (d < bd) { bd = d; best = x; } }
                        if (best < prev) best = prev;
                        int chunks = (int)((best - prev + 63) / 64);
                        if (chunks > worst) worst = chunks;
                        prev = best;
                    }
                    hist[kw - 3][worst < 5 ? worst : 5]++;
                }
            }
            for (int kw = 3; kw <= 5; kw++)
                fprintf(stderr, "emu stats: %d workers: blocks by chunks of the busiest worker: 0:%lld 1:%lld 2:%lld 3:%lld 4:%lld 5+:%lld\n", kw, hist[kw - 3][0], hist[kw - 3][1], hist[kw - 3][2], hist[kw - 3][3], hist[kw - 3][4], hist[kw - 3][5]);
        }
    }
    // ---- K2b, K3.  Segments (AUGX_SEG_LEN, else the planner's choice for 256 slots): pass 1 over all segments, pass 2 = the
    //      fix-ups, pass 3 = continuation of pieces with a fix-up that gave up, then the region offsets (as augx_batch_decode does)
    SegPlan plan = planSegments(L, *t, 256);
    B.nSegs = (int)plan.segs.size();
    B.segs = plan.segs.data();
    B.pieceSeg0 = plan.pieceSeg0.data();
    B.segCheckTiles = plan.checkTiles;
    if (const char *e = getenv("AUGX_SEG_CHECK_TILES")) B.segCheckTiles = atoi(e); // (tests of the give-up path: an unreachable check length)
    std::vector<int32_t> segStop(B.nSegs, -1), segStop2(B.nSegs, -1), segStatus(B.nSegs, 0), brkPos(B.nSegs, 0), pieceCovered(n, -1);
    std::vector<double> segD(B.nSegs, 0.0), segD2(B.nSegs, 0.0), brkOff(B.nSegs, 0.0);
    B.segStop = segStop.data(); B.segStatus = segStatus.data(); B.segD = segD.data(); B.brkPos = brkPos.data(); B.brkOff = brkOff.data();
    B.segStop2 = segStop2.data(); B.segD2 = segD2.data(); B.pieceCovered = pieceCovered.data();
    if (plan.cut()) {
        for (int64_t gt = 0; gt < B.N / WAVE; gt++) tileCrossOne(B, gt);
        B.ckRing = zalloc<double>((int64_t)B.nSegs * 2 * WAVE * SP);
        B.ckCol = zalloc<double>(Z.N / WAVE * SP);
    }
    TrellisLds *lds = new TrellisLds();
#define EMU_TRELLIS(MODE_, idx) do { if (blk == 8) trellisPiece<8, MODE_, true>(T, B, *lds, idx); else if (blk == 4) trellisPiece<4, MODE_, true>(T, B, *lds, idx); else trellisPiece<2, MODE_, true>(T, B, *lds, idx); } while (0)
    auto runTrellis = [&]() {
        std::fill(segStop.begin(), segStop.end(), -1); std::fill(segStop2.begin(), segStop2.end(), -1); std::fill(segStatus.begin(), segStatus.end(), 0);
        std::fill(pieceCovered.begin(), pieceCovered.end(), -1);
        for (int sg = 0; sg < B.nSegs; sg++) EMU_TRELLIS(0, sg);
        if (plan.cut()) {
            for (int sg = B.nSegs - 1; sg >= 0; sg--) EMU_TRELLIS(1, sg); // (any order: the fix-ups are independent of each other)
            for (int round = 0; round < SEG_CONT_ROUNDS; round++)
                for (int p = 0; p < n; p++) EMU_TRELLIS(2, p);
            for (int p = 0; p < n; p++) EMU_TRELLIS(3, p);
        }
    };
    runTrellis();
    if (!getenv("AUGX_EXACT_MULTICLASS") || atoi(getenv("AUGX_EXACT_MULTICLASS")) != 0) { // (augx_decoder_set_exact, on by default) the reference's snippet cache on pieces with several GC classes, then again
        size_t nPatched = 0;
        for (int p = 0; p < n; p++) {
            if (B.nPlanes[p] <= 1) continue;
            SnippetReplay R;
            const int len = L.len[p], S = t->S;
            const int64_t o = L.off[p], lo2 = B.listOffs[p];
            std::vector<double> col0((size_t)S);
            for (int s2 = 0; s2 < S; s2++) col0[s2] = L.initKind[p] == 0 ? t->ln_init[s2] : (s2 == t->synch_state ? 0.0 : -INFINITY);
            R.t = t; R.n = len; R.S = S; R.blk = blk; R.d = t->d;
            R.ldVal = B.ldVal + lo2 * 3; R.rdVal = B.rdVal + lo2 * 3; R.col0 = col0.data();
            R.plane = B.gcPlane + o + 1;
            R.planeCls = B.planeCls + (int64_t)p * MAXPL;
            R.nPlanes = B.nPlanes[p];
            const int nBlocks = (len + blk - 1) / blk;
            const int64_t gb0 = o / blk;
            R.blkOff = B.blkOff + gb0 * 2; R.blkCnt = B.blkCnt + gb0 * 2;
            uint64_t lo = ~0ull;
            for (int q = 0; q < nBlocks; q++) if (R.blkCnt[(size_t)q * 2 + 1] && R.blkOff[(size_t)q * 2 + 1] < lo) lo = R.blkOff[(size_t)q * 2 + 1];
            if (lo == ~0ull) lo = 0;
            R.item0 = lo; R.items = B.items + lo;
            R.fxF.assign((size_t)R.nPlanes, {}); R.fxR.assign((size_t)R.nPlanes, {});
            for (int pl = 0; pl < R.nPlanes; pl++) {
                R.fxF[pl].resize((size_t)len + 1); R.fxR[pl].resize((size_t)len + 1);
                const uint64_t *fx = B.fx + (int64_t)pl * B.N * NFX;
                for (int g = 0; g <= len; g++) { R.fxF[pl][g] = fx[fidx(o + g, FX_INF, NFX)]; R.fxR[pl][g] = fx[fidx(o + g, FX_INR, NFX)]; }
            }
            const auto tr0 = std::chrono::steady_clock::now();
            R.run();
            if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu stats: piece %d (%d bases): snippet replay %.3f s, %zu terms rebuilt\n", p, len, std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count(), R.patches.size());
            nPatched += R.patches.size();
        }
        if (nPatched) runTrellis();
    }
#undef EMU_TRELLIS
    int nGaveUp = 0;
    for (int sg = 0; sg < B.nSegs; sg++) nGaveUp += segStop[sg] <= -2;
    if (getenv("AUGX_EMU_STATS")) {
        fprintf(stderr, "emu stats: trellis candidates read back from HBM: igenic %lld, list %lld, in %lld chunks of 64\n", g_emuSlowVig, g_emuSlowList, g_emuItemWaves);

        fprintf(stderr, "emu stats: %d segments for %d pieces, check %d tiles, %d fix-ups gave up;", B.nSegs, n, B.segCheckTiles, nGaveUp);
        for (int sg = 0; sg < B.nSegs && sg < 24; sg++) fprintf(stderr, " [%d:%d..%d stop %d D %.3f cont %d]", plan.segs[sg].piece, plan.segs[sg].t0, plan.segs[sg].t1, segStop[sg], segD[sg], segStop2[sg]);
        fprintf(stderr, "\n");
    }
    for (int p = 0; p < n; p++) {
        segFinalizePiece(B, p);
        backtracePiece(T, B, p);
    }
    delete lds;
    if (fwd_out) { // ---- forward algorithm (posterior sampling), after the Viterbi decode
        B.fwd = zalloc<double>(Z.N * t->S);
        std::vector<double> lnF(n);
        B.lnFwd = lnF.data();
        FwdLds *fl = new FwdLds();
        auto fwdPiece = [&](int p) { if (blk == 8) forwardPiece<8>(T, B, *fl, p); else if (blk == 4) forwardPiece<4>(T, B, *fl, p); else forwardPiece<2>(T, B, *fl, p); };
        for (int p = 0; p < n; p++) {
            fwdPiece(p);
            if (B.nPlanes[p] > 1 && !getenv("AUGX_NO_MEMO")) { // the reference's snippet cache around the class steps (snipmemo.h), then once more
                SnippetReplay R;
                const int len = L.len[p], S = t->S;
                const int64_t o = L.off[p];
                R.t = t; R.n = len; R.S = S; R.blk = blk; R.d = t->d;
                R.F = B.fwd + (o + 1) * S;
                R.plane = B.gcPlane + o + 1;
                R.planeCls = B.planeCls + (int64_t)p * MAXPL;
                R.nPlanes = B.nPlanes[p];
                const int nBlocks = (len + blk - 1) / blk;
                const int64_t gb0 = o / blk;
                R.blkOff = B.blkOff + gb0 * 2; R.blkCnt = B.blkCnt + gb0 * 2;
                uint64_t lo = ~0ull;
                for (int q = 0; q < nBlocks; q++) if (R.blkCnt[(size_t)q * 2 + 1] && R.blkOff[(size_t)q * 2 + 1] < lo) lo = R.blkOff[(size_t)q * 2 + 1];
                if (lo == ~0ull) lo = 0;
                R.item0 = lo; R.items = B.items + lo;
                R.fxF.assign((size_t)R.nPlanes, {}); R.fxR.assign((size_t)R.nPlanes, {});
                for (int pl = 0; pl < R.nPlanes; pl++) {
                    R.fxF[pl].resize((size_t)len + 1); R.fxR[pl].resize((size_t)len + 1);
                    const uint64_t *fx = B.fx + (int64_t)pl * B.N * NFX;
                    for (int g = 0; g <= len; g++) { R.fxF[pl][g] = fx[fidx(o + g, FX_INF, NFX)]; R.fxR[pl][g] = fx[fidx(o + g, FX_INR, NFX)]; }
                }
                R.run();
                if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu stats: piece %d: %zu candidate terms rebuilt from the reference's snippet cache\n", p, R.patches.size());
                if (!R.patches.empty()) fwdPiece(p);
            }
        }
        delete fl;
        int64_t w = 0;
        for (int p = 0; p < n; p++) {
            memcpy(fwd_out + w, B.fwd + (L.off[p] + 1) * t->S, sizeof(double) * (size_t)L.len[p] * t->S);
            w += (int64_t)L.len[p] * t->S;
            if (lnfwd_out) lnfwd_out[p] = lnF[p];
        }
        g_samples.assign((size_t)n, {});
        for (int p = 0; p < n && g_nsamples > 0; p++) {
            SamplePiece P;
            const int len = L.len[p], S = t->S;
            const int64_t o = L.off[p];
            P.t = t; P.S = S; P.n = len; P.blk = blk; P.cls0 = cls[p]; P.nPlanes = B.nPlanes[p]; P.termKind = L.termKind[p];
            P.F = B.fwd + (o + 1) * S;
            P.sig.assign(B.sig + (o + 1) * NSIG, B.sig + (o + 1 + len) * NSIG);
            if (P.nPlanes > 1) {
                P.plane.assign(B.gcPlane + o + 1, B.gcPlane + o + 1 + len);
                P.planeCls.assign(B.planeCls + (int64_t)p * MAXPL, B.planeCls + (int64_t)(p + 1) * MAXPL);
            }
            const int nBlocks = (len + blk - 1) / blk;
            const int64_t gb0 = o / blk;
            P.blkOff.assign(B.blkOff + gb0 * 2, B.blkOff + (gb0 + nBlocks) * 2);
            P.blkCnt.assign(B.blkCnt + gb0 * 2, B.blkCnt + (gb0 + nBlocks) * 2);
            P.item0 = P.blkOff[1];
            const uint64_t itemEnd = P.blkOff[(size_t)(nBlocks - 1) * 2 + 1] + P.blkCnt[(size_t)(nBlocks - 1) * 2 + 1];
            P.items.assign(B.items + P.item0, B.items + itemEnd);
            P.anyNuc = false;
            for (int q = 0; q < len && !P.anyNuc; q++) P.anyNuc = B.code[o + 1 + q] < 4;
            std::vector<int> sst;
            const auto ts0 = std::chrono::steady_clock::now();
            prepareStops(P);
            const auto ts1 = std::chrono::steady_clock::now();
            samplePaths(P, g_nsamples, *g_rand, g_samples[p], sst);
            if (getenv("AUGX_EMU_STATS")) fprintf(stderr, "emu sampler: piece %d (%d bases): stops %.3f s, %d paths drawn in %.3f s (the generator's buffers so far: %.3f s)\n", p, P.n, std::chrono::duration<double>(ts1 - ts0).count(), g_nsamples,
                                                  std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count(), g_rand->refillSeconds);
        }
        free(B.fwd);
    }
    for (int p = 0; p < n; p++) {
        lnv[p] = lnvv[p];
        status[p] = st[p];
        if (cls_out) cls_out[p] = cls[p];
        int cnt = pc[p];
        path_n[p] = cnt;
        int64_t po = pathOff(B, p);
        for (int i = 0; i < cnt && i < path_cap; i++) {
            const int32_t *r = B.pathRec + (po + (cnt - 1 - i)) * 3;
            int32_t *o = path_out + ((int64_t)p * path_cap + i) * 3;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
        }
    }
    if (cells_out) {
        int64_t w = 0;
        for (int p = 0; p < n; p++) {
            memcpy(cells_out + w, B.cells + (L.off[p] + 1) * t->S, sizeof(double) * (size_t)L.len[p] * t->S);
            if (B.pieceSeg0[p + 1] - B.pieceSeg0[p] > 1) // regions of a piece decoded in segments are stored up to a constant
                for (int q = 0; q < L.len[p]; q++) {
                    const double off = frameOff(B, p, q);
                    for (int s2 = 0; s2 < t->S; s2++) cells_out[w + (int64_t)q * t->S + s2] += off;
                }
            w += (int64_t)L.len[p] * t->S;
        }
    }
    free(B.ckRing); free(B.ckCol); free(B.tileMinEop); free(B.tileCross);
    free(B.blkCnt); free(B.blkSplit); free(B.blkOff); free(B.items);
    free(raw); free(B.code); free(B.cnt); free(B.nsm); free(B.fx); free(B.sig); free(B.gate); free(B.site); free(B.bp); free(B.bpChain);
    free(B.cells); free(B.vig); free(B.longV); free(B.laPos); free(B.laVal); free(B.lrPos); free(B.lrVal); free(B.ldEnt); free(B.ldVal);
    free(B.rdEnt); free(B.rdVal); free(B.atgPos); free(B.pathRec);
    free(B.laPls); free(B.laFx); free(B.lrEt); free(B.lrFx); free(B.atgD); free(B.atgFx); free(B.rsPos); free(B.rsBegin); free(B.rsFx); free(B.plsR); free(B.gcRaw); free(B.gcPlane);
    return 0;
}
}
