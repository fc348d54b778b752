// layout.h -- host-side helpers shared by the HIP decoder and the test emulator: slot layout of a batch,
// DevTables construction from augx_tables, buffer size bookkeeping.
#pragma once
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "dp.h"

namespace augx {
namespace dev {

struct BatchLayout {
    int nPieces = 0;
    int64_t N = 0;
    int nChunks = 0;
    std::vector<int64_t> off;       // [nPieces+1]
    std::vector<int32_t> len, initKind, termKind, chunkPiece;
    void build(const augx_piece *pieces, int n) {
        nPieces = n;
        off.assign(n + 1, 0);
        len.resize(n); initKind.resize(n); termKind.resize(n);
        for (int p = 0; p < n; p++) {
            // (ln V falls by ~1.39 per base; it must stay above -2^(52-AUGX_Q_BITS) for the fp64 additions to be exact)
            if (pieces[p].len < 1 || pieces[p].len >= 2900000) throw std::runtime_error("augx: piece length out of range (device pieces are shorter than 2.9 Mbp; lower --maxDNAPieceSize)");
            len[p] = (int32_t)pieces[p].len;
            initKind[p] = pieces[p].init_kind;
            termKind[p] = pieces[p].term_kind;
            int64_t slots = (int64_t)len[p] + 1 + 8;               // before-first slot + a few pad slots
            slots = (slots + CHUNK - 1) / CHUNK * CHUNK;
            off[p + 1] = off[p] + slots;
        }
        N = off[n];
        nChunks = (int)(N / CHUNK);
        chunkPiece.resize(nChunks);
        for (int p = 0; p < n; p++)
            for (int64_t ch = off[p] / CHUNK; ch < off[p + 1] / CHUNK; ch++) chunkPiece[ch] = p;
    }
};

// smoothed GC-content stairs of one piece from the classes of its windows (reference ContentStairs::computeStairs,
// src/motif.cc:543-616): wc[s] = class of the window of `win` bases starting at base s, s in [0, n - win] (win already
// clamped to n).  Base i has the class of the window centred on it, the ends that of the first / last window; a step
// shorter than 1000 bases that returns to the class before it is removed.  The classes are then numbered by first
// appearance (planes).  Returns the number of planes (plane[] filled, planeCls[k] = class of plane k), or -1 if the
// piece has more than MAXPL classes.
inline int stairsPlanes(const uint8_t *wc, int n, int win, std::vector<uint8_t> &plane, int32_t planeCls[MAXPL]) {
    if (win > n || win < 1) win = n;
    const int half = win / 2, last = n - win;
    // runs of equal window class -> runs of positions (classes change every few kb at most: work per run, not per base)
    std::vector<int> st, cl; // first position and class of each run
    for (int s0 = 0; s0 <= last;) {
        const uint8_t c = wc[s0];
        int s1 = s0 + 1;
        const uint64_t pat = 0x0101010101010101ull * c;
        while (s1 + 8 <= last + 1) { // eight windows at a time while they agree
            uint64_t w;
            memcpy(&w, wc + s1, 8);
            if (w != pat) break;
            s1 += 8;
        }
        while (s1 <= last && wc[s1] == c) s1++;
        st.push_back(s0 == 0 ? 0 : s0 + half); // window s is centred on position s + half; the ends take the first / last window
        cl.push_back(c);
        s0 = s1;
    }
    // smoothing, in the order of the reference's loop over the positions: when run k begins, run k-1 is dissolved into its
    // neighbours if it is shorter than 1000, does not start the piece, and the run before it has the class of run k
    const int R = (int)st.size();
    for (int k = 2; k < R; k++)
        if (st[k] - st[k - 1] < 1000 && st[k - 1] > 0 && cl[k - 2] == cl[k]) cl[k - 1] = cl[k];
    int nPl = 0, map[256];
    for (int i = 0; i < 256; i++) map[i] = -1;
    plane.resize((size_t)n);
    for (int k = 0; k < R; k++) {
        int &m = map[cl[k]];
        if (m < 0) {
            if (nPl >= MAXPL) return -1;
            planeCls[nPl] = cl[k];
            m = nPl++;
        }
        const int e = k + 1 < R ? st[k + 1] : n;
        memset(plane.data() + st[k], m, (size_t)(e - st[k]));
    }
    return nPl;
}

// first list entry of every piece from the counted sites (k1ListCount); returns the total = entries per plane
inline int64_t listOffsets(const int32_t *listCnt, int nPieces, std::vector<int64_t> &offs) {
    offs.assign((size_t)nPieces + 1, 0);
    for (int p = 0; p < nPieces; p++) offs[p + 1] = offs[p] + (((int64_t)listCnt[p] + 1 + 63) / 64) * 64; // (+1: index i1 may point one past)
    return offs[nPieces] + 64;
}

// ---- segments of the trellis (dp.h: SegDesc).  A fix-up ends after `checkTiles` consecutive verified tiles: they must cover
// the longest look-back of any state (a candidate of a variable-length state reaches back at most one maximal exon or intron
// plus its signal windows; equalD reads at lag dStateLen), plus the tiles whose retired values are still on their way.
constexpr int SEG_CONT_ROUNDS = 3; // launches of pass 3 (one continuation per piece each); what is left goes sequentially to the end
struct SegPlan {
    std::vector<SegDesc> segs;
    std::vector<int32_t> pieceSeg0;
    int checkTiles = 0;
    bool cut() const { return segs.size() + 1 > pieceSeg0.size(); } // some piece has more than one segment
};
inline int segCheckTiles(const augx_tables &t) {
    int reach = t.max_exon_len > t.d ? t.max_exon_len : t.d;
    reach += t.W + t.U + 64;
    return (reach + WAVE - 1) / WAVE + 4;
}
// can pieces of this model be cut at all?  (the synch state must be the intergenic state, whose column anchors a dead start;
// equalD must read its long-lag cells through the tile-wise flush, not the direct path of short dStateLen)
inline bool segmentsSupported(const augx_tables &t) {
    const int dL = t.d - 2 - t.De - t.As - 2 - t.U;
    return t.state_kind[t.synch_state] == AUGX_K_IGENIC && dL >= 4 * WAVE && t.reachable[t.synch_state];
}
// segTiles: tiles per segment wanted (0: choose so that `slots` workgroups are busy; < 0: never cut).  AUGX_SEG_LEN (bases)
// overrides it (tests; 0 = never cut).
inline SegPlan planSegments(const BatchLayout &L, const augx_tables &t, int slots, int segTiles = 0) {
    SegPlan P;
    P.checkTiles = segCheckTiles(t);
    const int n = L.nPieces;
    if (const char *e = getenv("AUGX_SEG_LEN")) { const long v = atol(e); segTiles = v <= 0 ? -1 : (int)((v + WAVE - 1) / WAVE); }
    const int minSeg = 5 * P.checkTiles; // a segment holds its own fix-up (convergence + checkTiles) and the look-back of the next one
    std::vector<int> tiles(n);
    int64_t total = 0;
    int maxTiles = 0;
    for (int p = 0; p < n; p++) { tiles[p] = (L.len[p] + WAVE - 1) / WAVE; total += tiles[p]; if (tiles[p] > maxTiles) maxTiles = tiles[p]; }
    if (!segmentsSupported(t)) segTiles = -1;
    auto countFor = [&](int st, int p) { // nearest count, no segment shorter than minSeg
        int k = (tiles[p] + st / 2) / st;
        while (k > 1 && tiles[p] / k < minSeg) k--;
        return k < 1 ? 1 : k;
    };
    if (segTiles == 0) {
        // estimate of the time of the three passes for a candidate segment length: rounds of `slots` concurrent workgroups,
        // each as long as the longest segment / a typical fix-up (convergence within ~400 tiles on random DNA, then the check)
        const int64_t fixLen = 450; // (convergence ~400 tiles on random DNA + the check window, which follows the candidates that really cross the stop point)
        int64_t bestCost = -1;
        int best = -1;
        for (int k = 1; k <= 64; k++) {
            int st = (maxTiles + k - 1) / k;
            if (k > 1 && st < minSeg) break;
            int64_t nSeg = 0, nFix = 0, longest = 0;
            for (int p = 0; p < n; p++) {
                const int kp = k == 1 ? 1 : countFor(st, p);
                nSeg += kp; nFix += kp - 1;
                const int64_t len = (tiles[p] + kp - 1) / kp;
                if (len > longest) longest = len;
            }
            int64_t cost = (nSeg + slots - 1) / slots * longest + (nFix + slots - 1) / slots * fixLen;
            // (the fix-ups are extra work and their length has a heavy tail: cutting must promise a clear gain to be chosen)
            if (k > 1) cost += cost / 6;
            if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = k == 1 ? -1 : st; }
        }
        segTiles = best;
    }
    if (segTiles > 0 && segTiles < minSeg) segTiles = minSeg;
    std::vector<int> kOf(n, 1);
    if (segTiles > 0) {
        int64_t nSeg = 0;
        for (int p = 0; p < n; p++) { kOf[p] = countFor(segTiles, p); nSeg += kOf[p]; }
        // whole rounds: when the segments nearly fill a last round of `slots` workgroups, a few pieces get one segment more
        // (100 pieces x 5 = 500 segments on 256 compute units -> 12 pieces with 6: two full rounds of shorter segments)
        int extra = nSeg > slots ? (int)((slots - nSeg % slots) % slots) : 0;
        if (extra > 0 && extra <= n / 4 + 1)
            for (int p = 0; p < n && extra > 0; p++)
                if (kOf[p] > 1 && tiles[p] / (kOf[p] + 1) >= minSeg) { kOf[p]++; extra--; }
    }
    P.pieceSeg0.assign((size_t)n + 1, 0);
    for (int p = 0; p < n; p++) {
        const int kp = kOf[p];
        P.pieceSeg0[p] = (int32_t)P.segs.size();
        for (int k = 0; k < kp; k++) {
            SegDesc d;
            d.piece = p; d.k = k;
            d.t0 = (int32_t)((int64_t)tiles[p] * k / kp);
            d.t1 = (int32_t)((int64_t)tiles[p] * (k + 1) / kp);
            d.tlim = d.t1 - P.checkTiles - 2;
            d.pad = 0;
            P.segs.push_back(d);
        }
    }
    P.pieceSeg0[n] = (int32_t)P.segs.size();
    return P;
}

// element counts of every device buffer of a batch (bytes = count * sizeof(element))
struct BatchSizes {
    int64_t N, nChunks, nPieces, pathCap;
    explicit BatchSizes(const BatchLayout &L) {
        N = L.N; nChunks = L.nChunks; nPieces = L.nPieces;
        pathCap = N / 8 + 64 * (int64_t)L.nPieces + 64;
    }
};

inline void checkModelSupported(const augx_tables &t, int BLK) {
    if (t.S > SP) throw std::runtime_error("augx: model has more than 48 states: not for the wavefront layout of the trellis kernel (models with UTR states or two intergenic states go to the dense kernels, modelIsDense below)");
    int dL = t.d - 2 - t.De - t.As - 2 - t.U;
    if (dL >= LONG_RING || dL <= BLK || (dL > WAVE - BLK && dL < WAVE)) throw std::runtime_error("augx: intron d out of the supported range");
    // (a near fixed-lag state of block b is computed when the candidates of block b-1 are done: its lag must reach back past
    //  the block, lag >= BLK)
    if (t.Ds + 2 + t.De < BLK || t.As + 2 + t.Ae + t.U < BLK) throw std::runtime_error("augx: splice-site windows shorter than a trellis block");
    if (t.As + 2 + t.Ae + t.U > 63 || t.Ds + 2 + t.De > 63) throw std::runtime_error("augx: splice-site windows too long");
    if (t.max_exon_len + t.W + 64 > 0x3FFF) throw std::runtime_error("augx: maxexonlength too large for 14-bit back pointers");
    if (t.d > 0x3FFF) throw std::runtime_error("augx: intron d too large");
    int nFixed = 0, nVar = 0, nChain = 0;
    for (int s = 0; s < t.S; s++) {
        int kind = t.state_kind[s];
        if (t.reachable[s]) {
            bool fixedLag = kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS || kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS ||
                            kind == AUGX_K_EQUALD || kind == AUGX_K_REQUALD;
            bool chain = kind == AUGX_K_IGENIC || kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC;
            if (fixedLag) { nFixed++; if (t.n_anc[s] > 2) throw std::runtime_error("augx: fixed-length intron state with more than 2 ancestors"); }
            else if (chain) { nChain++; if (t.n_anc[s] > 5) throw std::runtime_error("augx: single-base state with more than 5 ancestors"); }
            else nVar++;
        }
        bool var3 = kind == AUGX_K_INTERNAL || kind == AUGX_K_TERMINAL || kind == AUGX_K_RINTERNAL || kind == AUGX_K_RINITIAL;
        if (var3 && t.n_anc[s] > 3) throw std::runtime_error("augx: exon state with more than 3 ancestors");
        if (var3)
            for (int a = 0; a < t.n_anc[s]; a++)
                for (int b2 = 0; b2 < a; b2++)
                    if (t.state_win[t.anc[s][a]] == t.state_win[t.anc[s][b2]])
                        throw std::runtime_error("augx: exon state with two ancestors of the same reading frame");
        if ((kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD || kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL ||
             kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL) && t.n_anc[s] != 1)
            throw std::runtime_error("augx: unexpected ancestor count (non-standard transition file)");
    }
    // the trellis kernel finishes the RTERMINAL cells of a block when the igenic cells of that block exist, one or two
    // blocks later: their readers must be fixed-lag states of the far class (lag >= 3 blocks), whose step waits for them
    for (int s = 0; s < t.S; s++)
        for (int a = 0; a < t.n_anc[s]; a++)
            if (t.reachable[s] && t.state_kind[t.anc[s][a]] == AUGX_K_RTERMINAL) {
                int kind = t.state_kind[s];
                int lag = (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) ? t.As + 2 + t.Ae + t.U
                          : (kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) ? t.Ds + 2 + t.De : 0;
                if (lag < 3 * BLK) throw std::runtime_error("augx: unexpected successor of the reverse terminal exon state");
            }
    // scheduling assumptions of the trellis kernel (device/kernels.h, trellisPiece)
    {
        auto isChain = [&](int k) { return k == AUGX_K_IGENIC || k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC; };
        auto isFixed = [&](int k) { return k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS || k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS || k == AUGX_K_EQUALD || k == AUGX_K_REQUALD; };
        for (int s = 0; s < t.S; s++) {
            if (!t.reachable[s]) continue;
            const int kind = t.state_kind[s];
            for (int a = 0; a < t.n_anc[s]; a++) {
                const int ak = t.state_kind[t.anc[s][a]];
                // igenic lags one block behind: no fixed-lag state may read it; the geometric states run ahead of the
                // candidates of their block: they may only be fed by fixed-lag states
                if (isFixed(kind) && ak == AUGX_K_IGENIC) throw std::runtime_error("augx: fixed-length intron state fed by the intergenic state");
                if ((kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC) && t.anc[s][a] != s && !isFixed(ak))
                    throw std::runtime_error("augx: geometric intron state fed by a state that is not a fixed-length intron state");
                if (kind == AUGX_K_IGENIC && isChain(ak) && t.anc[s][a] != s) throw std::runtime_error("augx: intergenic state fed by an intron state");
            }
        }
        // single / initial exons reach back to an igenic cell at least this far (reference src/exonmodel.cc:1042-1054)
        int slack = t.W + t.min_exon_len - t.Ds;
        if (t.W < slack) slack = t.W;
        if (slack < BLK) throw std::runtime_error("augx: trans_init_window too short for the block size of the trellis kernel");
    }
    {   // the fixed-lag states are laid out in rounds of 8: near / late states (lag < 3 blocks) first, then the far ones
        int nNear = 0, nFar = 0;
        for (int s = 0; s < t.S; s++) {
            if (!t.reachable[s]) continue;
            int kind = t.state_kind[s];
            int lag = (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) ? t.As + 2 + t.Ae + t.U
                      : (kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) ? t.Ds + 2 + t.De
                      : (kind == AUGX_K_EQUALD || kind == AUGX_K_REQUALD) ? dL : 0;
            if (lag > 0) (lag < 3 * BLK ? nNear : nFar)++;
        }
        if ((nNear + 7) / 8 + (nFar + 7) / 8 > 3) throw std::runtime_error("augx: too many fixed-length intron states for the trellis wavefront layout");
    }
    if (nFixed > 24 || nVar > 32 || nChain > 8) throw std::runtime_error("augx: state graph too large for the trellis wavefront layout");
}

// models decoded by the dense kernels (dense.h): everything the wavefront layout of the trellis kernel was not built for
inline int chooseDenseBlock(const augx_tables &t);
inline bool modelIsDense(const augx_tables &t) {
    int nIg = 0; // (two intergenic states: --genemodel=atleastone / exactlyone)
    for (int s = 0; s < t.S; s++) nIg += t.state_kind[s] == AUGX_K_IGENIC;
    if (t.utr != 0 || nIg > 1) return true;
    // a 47-state model whose windows the wavefront layout of the trellis kernel was not built for (an equalD state that looks back 57-63
    // bases; splice-site windows of more than 63 bases; a translation-initiation window shorter than the smallest block) goes to the
    // state-graph driven dense kernels as well, where they take it (round 6: Vitrella_brassicaformis, maize, Micromonas_pusilla,
    // Rhopilema_esculentum -- slower there, but the reference's result instead of a refusal)
    try { checkModelSupported(t, 2); return false; } catch (std::exception &) {}
    for (int b = 4; b <= 8; b *= 2) { try { checkModelSupported(t, b); return false; } catch (std::exception &) {} }
    try { (void)chooseDenseBlock(t); return true; } catch (std::exception &) { return false; } // (neither: chooseBlockSize reports the trellis kernel's reason)
}
// block size of the dense kernels (dense.h): no variable-length or fixed-lag state may read a cell of its own block but through
// the stage order of densePiece (fixed-lag states, early chains, candidates, late chains, reverse terminal exons)
inline int chooseDenseBlock(const augx_tables &t) {
    if (t.S > SPX) throw std::runtime_error("augx: too many states");
    const int dssWhole = t.Ds + 2 + t.De, assLag = t.As + 2 + t.Ae + t.U, dL = t.d - 2 - t.De - t.As - 2 - t.U;
    int lag = dssWhole < assLag ? dssWhole : assLag;
    if (dL < lag) lag = dL;
    int slack = t.W + t.min_exon_len - t.Ds; // single / initial exons reach back to their predecessor at least this far
    if (t.W < slack) slack = t.W;
    if (slack < lag) lag = slack;
    if (t.utr) {
        DevTables D;
        memset(&D, 0, sizeof D);
        D.W = t.W; D.U = t.U; D.As = t.As; D.Ae = t.Ae; D.Ds = t.Ds; D.De = t.De; D.tss_upwin = t.tss_upwin; D.tss_end = t.tss_end;
        D.dpc = t.d_polyasig_cleavage; D.boxlen = t.aataaa_boxlen; D.uML = t.utr_max_exon_len; D.uM3S = t.utr_max3single; D.uM3T = t.utr_max3term;
        const int ul = utrMinLag(D);
        if (ul < lag) lag = ul;
        if (t.tss_upwin + 2 > KEY_BIAS || t.aataaa_boxlen + t.d_polyasig_cleavage + 2 > KEY_BIAS) throw std::runtime_error("augx: UTR signal windows too long for the candidate keys");
    }
    int nChain = 0, nFix = 0, nUv = 0;
    for (int s = 0; s < t.S; s++) {
        if (!t.reachable[s]) continue;
        const int k = t.state_kind[s];
        const bool chain = k == AUGX_K_IGENIC || k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC || isUtrIntronKind(k);
        const bool fixed = k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS || k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS || k == AUGX_K_EQUALD || k == AUGX_K_REQUALD;
        nChain += chain; nFix += fixed; nUv += isUtrExonKind(k);
        for (int a = 0; a < t.n_anc[s]; a++) {
            const int ak = t.state_kind[t.anc[s][a]];
            // the early chains (geometric introns) run before the candidates of their block
            if ((k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC) && t.anc[s][a] != s &&
                !(ak == AUGX_K_LONGDSS || ak == AUGX_K_RLONGDSS || ak == AUGX_K_LONGASS || ak == AUGX_K_RLONGASS || ak == AUGX_K_EQUALD || ak == AUGX_K_REQUALD))
                throw std::runtime_error("augx: geometric intron state fed by a state that is not a fixed-length intron state");
            if (chain && ak == AUGX_K_RTERMINAL) throw std::runtime_error("augx: single-base state fed by the reverse terminal exon state");
        }
    }
    if (nChain > 16 || nFix > 24 || nUv > 16) throw std::runtime_error("augx: state graph too large for the dense kernels");
    int b = 8;
    if (const char *e = getenv("AUGX_BLK")) b = atoi(e);
    while (b > 1 && b > lag) b /= 2;
    if (b < 2) throw std::runtime_error("augx: signal windows too short for the dense kernels");
    return b;
}

// block size of the candidate / trellis kernels for this model: 8 where the species' windows allow it, else 4 or 2
// (override for tests: AUGX_BLK=4).  Throws what checkModelSupported throws for the smallest size.
inline int chooseBlockSize(const augx_tables &t) {
    if (modelIsDense(t)) return chooseDenseBlock(t);
    if (const char *e = getenv("AUGX_BLK")) {
        const int b = atoi(e);
        if (b != 8 && b != 4 && b != 2) throw std::runtime_error("augx: AUGX_BLK must be 8, 4 or 2");
        checkModelSupported(t, b);
        return b;
    }
    for (int b = 8; b > 2; b /= 2) {
        try {
            checkModelSupported(t, b);
            return b;
        } catch (std::exception &) {
        }
    }
    checkModelSupported(t, 2);
    return 2;
}

// fill the scalar part of DevTables; the caller sets the table pointers (device or host)
inline void fillDevTablesScalars(const augx_tables &t, DevTables &D) {
    memset(&D, 0, sizeof D);
    D.S = t.S; D.C = t.n_classes; D.k = t.k; D.NP = 1 << (2 * (t.k + 1)); D.kIn = t.k_in; D.NPin = 1 << (2 * (t.k_in + 1));
    D.W = t.W; D.U = t.U; D.As = t.As; D.Ae = t.Ae; D.Ds = t.Ds; D.De = t.De; D.Li = t.Li; D.Le = t.Le; D.d = t.d;
    D.dStateLen = t.d - 2 - t.De - t.As - 2 - t.U;
    D.max_exon_len = t.max_exon_len; D.min_exon_len = t.min_exon_len;
    D.tis_n = t.tis_n; D.tis_k = t.tis_k; D.ass_n = t.ass_n; D.ass_k = t.ass_k; D.tis_nbins = t.tis_nbins; D.tis_mem = t.tis_mem;
    D.synch = t.synch_state; D.gc_win = t.gc_win; D.gc_weighing_type = t.gc_weighing_type;
    D.soft = t.softmasking; D.lnSoft = t.ln_soft_bonus;
    D.dssGc = t.dss_gc; D.stopMask = t.stop_mask; D.startMask = t.start_mask;
    D.utr = t.utr; D.tss_upwin = t.tss_upwin; D.tss_start = t.tss_start; D.tss_end = t.tss_end; D.tata_start = t.tata_start; D.tata_end = t.tata_end;
    D.d_tss_tata_min = t.d_tss_tata_min; D.d_tss_tata_max = t.d_tss_tata_max; D.dpc = t.d_polyasig_cleavage; D.boxlen = t.aataaa_boxlen;
    D.tts_spacing = t.tts_spacing; D.uML = t.utr_max_exon_len; D.uM3S = t.utr_max3single; D.uM3T = t.utr_max3term; D.tssup_k = t.tssup_k;
    D.uk = t.utr ? t.utr_k : t.k; D.uNP = 1 << (2 * (D.uk + 1));
    D.tss_n = t.tss_n; D.tss_k = t.tss_k; D.tsstata_n = t.tsstata_n; D.tsstata_k = t.tsstata_k; D.tata_n = t.tata_n; D.tata_k = t.tata_k;
    D.tts_n = t.tts_n; D.tts_k = t.tts_k; D.ln_tts_rand = t.ln_tts_rand; D.ln2 = t.ln2;
    D.dense = modelIsDense(t);
    {   // end-gate bits of the variable-length states: the state index while it fits a 64-bit mask, else a compact numbering
        int nb = 0;
        for (int s = 0; s < t.S; s++) {
            const int k = t.state_kind[s];
            const bool var = (k >= AUGX_K_SINGLE && k <= AUGX_K_RTERMINAL) || k == AUGX_K_LESSD || k == AUGX_K_RLESSD || isUtrExonKind(k);
            D.vbit[s] = t.S <= 64 ? s : (var ? nb++ : 63);
        }
        if (nb > SP) throw std::runtime_error("augx: more than 48 variable-length states");
    }
    for (int s = 0; s < t.S; s++) {
        D.kind[s] = t.state_kind[s]; D.win[s] = t.state_win[s]; D.type[s] = t.state_type[s]; D.reachable[s] = t.reachable[s];
        D.n_anc[s] = t.n_anc[s];
        for (int a = 0; a < AUGX_MAX_ANC; a++) D.anc[s][a] = t.anc[s][a];
        D.ln_init[s] = t.ln_init[s]; D.ln_term[s] = t.ln_term[s];
    }
    D.nUv = 0;
    for (int s = 0; s < 16; s++) D.uvS[s] = -1;
    for (int s = 0; s < t.S; s++)
        if (t.reachable[s] && isUtrExonKind(t.state_kind[s]) && D.nUv < 16) D.uvS[D.nUv++] = s;
    for (int i = 0; i < 64; i++) D.ln_startcodon[i] = t.ln_startcodon[i];
    D.ln_stop_ochre = t.ln_stop_ochre; D.ln_stop_amber = t.ln_stop_amber; D.ln_stop_opal = t.ln_stop_opal;
    D.ln_quarter = t.ln_quarter; D.ln_n_coding = t.ln_n_coding; D.ln4 = t.ln4; D.ass_pat_invalid = t.ass_pat_invalid;
    for (int c = 0; c < AUGX_MAX_CLASSES; c++)
        for (int i = 0; i < 4; i++) D.gc_zus[c][i] = t.gc_zus[c][i];
    for (int i = 0; i < 16; i++) D.gc_weight_matrix[i] = t.gc_weight_matrix[i];
}

// table element counts, in the order of tablePointers()
struct TableSpan { const double *src; int64_t count; TabPtr *dst; };
inline std::vector<TableSpan> tableSpans(const augx_tables &t, DevTables &D) {
    const int64_t C = t.n_classes, NP = 1 << (2 * (t.k + 1)), S = t.S;
    std::vector<TableSpan> v;
    v.push_back({t.ln_trans, C * S * S, &D.ln_trans});
    v.push_back({t.ig_emi, C * NP, &D.ig_emi});
    v.push_back({t.ig_short, C * (t.k + 1) * NP, &D.ig_short});
    v.push_back({t.in_emi, C * ((int64_t)1 << (2 * (t.k_in + 1))), &D.in_emi});
    v.push_back({t.ex_emi, C * 3 * NP, &D.ex_emi});
    v.push_back({t.ex_init, C * 3 * NP, &D.ex_init});
    v.push_back({t.ex_et, C * 3 * NP, &D.ex_et});
    v.push_back({t.ex_pls, C * (t.k + 1) * 3 * NP, &D.ex_pls});
    v.push_back({t.tis_motif, C * t.tis_n * (1 << (2 * (t.tis_k + 1))), &D.tis_motif});
    v.push_back({t.ass_motif, C * t.ass_n * (1 << (2 * (t.ass_k + 1))), &D.ass_motif});
    v.push_back({t.tis_bin_bounds, t.tis_nbins > 0 ? C * (t.tis_nbins - 1) : 0, &D.tis_bin_bounds});
    v.push_back({t.tis_bin_ln, t.tis_nbins > 0 ? C * t.tis_nbins : 0, &D.tis_bin_ln});
    v.push_back({t.ass_pat, (int64_t)1 << (2 * (t.As + t.Ae)), &D.ass_pat});
    v.push_back({t.dss_pat, ((int64_t)1 << (2 * (t.Ds + t.De))) * (t.dss_gc ? 2 : 1), &D.dss_pat});
    v.push_back({t.len_intron, t.d + 1, &D.len_intron});
    v.push_back({t.len_single, t.max_exon_len + 1, &D.len_single});
    v.push_back({t.len_initial, t.max_exon_len + 1, &D.len_initial});
    v.push_back({t.len_internal, t.max_exon_len + 1, &D.len_internal});
    v.push_back({t.len_terminal, t.max_exon_len + 1, &D.len_terminal});
    if (t.utr) {
        const int64_t uNP = (int64_t)1 << (2 * (t.utr_k + 1));
        v.push_back({t.utr5init_emi, C * uNP, &D.utr5init_emi}); v.push_back({t.utr5_emi, C * uNP, &D.utr5_emi}); v.push_back({t.utr3_emi, C * uNP, &D.utr3_emi});
        v.push_back({t.tssup_emi, C * ((int64_t)1 << (2 * (t.tssup_k + 1))), &D.tssup_emi});
        v.push_back({t.tss_motif, C * t.tss_n * ((int64_t)1 << (2 * (t.tss_k + 1))), &D.tss_motif});
        v.push_back({t.tsstata_motif, C * t.tsstata_n * ((int64_t)1 << (2 * (t.tsstata_k + 1))), &D.tsstata_motif});
        v.push_back({t.tata_motif, C * t.tata_n * ((int64_t)1 << (2 * (t.tata_k + 1))), &D.tata_motif});
        v.push_back({t.tts_motif, C * t.tts_n * ((int64_t)1 << (2 * (t.tts_k + 1))), &D.tts_motif});
        v.push_back({t.aataaa, (int64_t)1 << (2 * t.aataaa_boxlen), &D.aataaa});
        v.push_back({t.len5_single, t.utr_max_exon_len + 1, &D.len5s}); v.push_back({t.len5_initial, t.utr_max_exon_len + 1, &D.len5i});
        v.push_back({t.len5_internal, t.utr_max_exon_len + 1, &D.len5n}); v.push_back({t.len5_terminal, t.utr_max_exon_len + 1, &D.len5t});
        v.push_back({t.len3_single, t.utr_max3single + 1, &D.len3s}); v.push_back({t.len3_initial, t.utr_max_exon_len + 1, &D.len3i});
        v.push_back({t.len3_internal, t.utr_max_exon_len + 1, &D.len3n}); v.push_back({t.len3_terminal, t.utr_max3term + 1, &D.len3t});
        v.push_back({t.tail5_single, t.utr_max_exon_len + 1, &D.tail5s}); v.push_back({t.tail3_single, t.utr_max3single + 1, &D.tail3s});
    }
    return v;
}

} // namespace dev
} // namespace augx
