// The reference's content cache of the short-intron states, replayed -- host code shared by the device library and the lane-loop
// emulator.  Only pieces with several GC classes need it, and only with the forward algorithm / sampling on.
//
// A short intron (lessD state) ending at base j scores its interior with IntronModel::seqProb -> SnippetProbs::getSeqProb(right = j,
// len) (reference src/intronmodel.cc:1064-1067, src/statemodel.cc:312-342): a cache keyed by (end base, length) that is never
// emptied when the GC class changes (IntronModel::updateToLocalGC only swaps the emission table, src/intronmodel.cc:495-503).  A
// request is answered from the longest cached piece that fits plus a recursive request for the rest, and only what is not cached is
// computed -- with the table of the class current AT THAT TIME.  Within 2 d bases after a class step an interior may therefore
// be the product of chunks scored under different classes, which chunks depending on what was requested before: on which
// predecessor cells were alive (a request is only made for a live predecessor, src/intronmodel.cc:589-600) and in which order
// (base by base, lessD states in state order, predecessor positions from near to far).  The device scores the whole interior with
// the class of its end base; this replay finds, for every candidate in the window after a class step, the chunks the reference
// would have used, and the term of the candidate is rebuilt from them (fixed-point prefix differences per class: exact).
// What comes out are the forward values of the reference to 1e-9, hence its sampled paths.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <vector>
#include <algorithm>
#include <utility>
#include "dp.h"

namespace augx {
namespace dev {

struct MemoChunk { int32_t end, len; int32_t plane; };
struct MemoEntry { int32_t len; std::vector<MemoChunk> dec; };

struct MemoPatch { uint64_t item; double te; }; // global index of the candidate record (as in blkOff)

// everything of one piece the replay reads (host memory)
struct SnippetReplay {
    const augx_tables *t = nullptr;
    int n = 0, S = 0, blk = 8, d = 0;
    bool dense = false;                   // the candidate records are those of the dense kernels (dense.h): the pair is (base of the block << 7) | state,
                                          // `src` names the predecessor state, a start from column 0 has predecessor end 0; aliveness from the dense matrix F
    // which predecessor cells are alive, from a first run: the ln forward matrix F [n][S], or -- after a Viterbi run, which keeps no
    // matrix -- the values the trellis left at the donor sites (ldVal / rdVal: [entry][3 frames]) and the initial column col0 [S]
    const double *F = nullptr;
    const double *ldVal = nullptr, *rdVal = nullptr, *col0 = nullptr;
    const uint8_t *plane = nullptr;       // [n] plane of every base
    const int32_t *planeCls = nullptr;    // [MAXPL] class of a plane
    int nPlanes = 1;
    Item *items = nullptr;                // candidates of the piece, starting at global index item0 (te is rewritten in place)
    uint64_t item0 = 0;
    // Windowed mode (the device library): only what a window reads is fetched from HBM, by `fetch`, before the window is replayed --
    // the candidate records of its blocks (pool / blkPool), the rows fRow0.. of F, the prefix slots fx0.. .  Whole mode (the
    // emulator, whose arrays are host memory anyway): fetch is empty, everything above points at the whole piece.
    std::function<int(int, int)> fetch;   // (t0, t1) -> 0, or an error code that run() hands on
    // (optional) told every window of the piece before the first one is replayed -- {t0, t1} in the order run() takes them --, so
    // that what they read can come from HBM in one go; fetch then only has to pick the window's part
    std::function<int(const std::vector<std::pair<int, int>> &)> prefetch;
    std::vector<Item> pool;
    std::vector<int64_t> blkPool;         // [nBlocks] first record of the block in pool, -1: not fetched
    const double *F0 = nullptr;           // row 0 of F (the initial column), windowed mode
    int fRow0 = 0, fx0 = 0;
    Item *blockItems(int b) { return fetch ? (blkPool[(size_t)b] >= 0 ? pool.data() + blkPool[(size_t)b] : nullptr) : items + (blkOff[(size_t)b * 2 + 1] - item0); }
    const uint64_t *blkOff = nullptr;     // [nBlocks][2]
    const uint32_t *blkCnt = nullptr;     // [nBlocks][2]
    // fixed-point content prefix of the intron model: fx(plane, rev, g) with g = slot relative to the piece (0 = before the first base)
    std::vector<std::vector<uint64_t>> fxF, fxR; // [plane][n + 1]
    std::vector<MemoPatch> patches;

    int64_t segFx(int pl, bool rev, int l, int r) const { // fixed-point content of bases l..r under plane pl
        if (l > r) return 0;
        const std::vector<uint64_t> &a = rev ? fxR[pl] : fxF[pl];
        return (int64_t)(a[(size_t)(r + 1 - fx0)] - a[(size_t)(l - fx0)]);
    }

    // ---- the cache of one strand: lists[base] sorted by length (SnippetList).  The bases a window touches lie in [flatLo, flatLo + size):
    //      a vector indexed by base (round 6: the std::map it was cost a third of the replay); anything outside goes to the map
    std::map<int, std::vector<MemoEntry>> lists[2];
    std::vector<std::vector<MemoEntry>> flat[2];
    int flatLo = 0;
    std::vector<MemoEntry> *listAt(int st, int base, bool create) {
        const int64_t k = (int64_t)base - flatLo;
        if (k >= 0 && k < (int64_t)flat[st].size()) return &flat[st][(size_t)k];
        if (create) return &lists[st][base];
        auto f = lists[st].find(base);
        return f == lists[st].end() ? nullptr : &f->second;
    }

    void add(int st, int base, int len, const std::vector<MemoChunk> &dec) { // SnippetProbs::addProb, src/statemodel.cc:344-370
        std::vector<MemoEntry> &v = *listAt(st, base, true);
        size_t pos = 0;
        while (pos < v.size() && v[pos].len < len) pos++;
        if (pos < v.size() && v[pos].len == len) return; // ("tried to add snippet of same length": the reference keeps the old one)
        v.insert(v.begin() + (long)pos, MemoEntry{len, dec});
    }
    std::vector<MemoChunk> get(int st, int base, int len, int curPlane) { // SnippetProbs::getSeqProb, src/statemodel.cc:312-342
        std::vector<MemoChunk> dec;
        if (len == 0) return dec;
        std::vector<MemoEntry> *fv = listAt(st, base, false);
        if (fv && !fv->empty()) {
            std::vector<MemoEntry> &v = *fv;
            if (v.back().len < len) {
                const int l1 = v.back().len;
                const std::vector<MemoChunk> last = v.back().dec; // (copy: the recursion may add to other lists, not to this one)
                dec = get(st, base - l1, len - l1, curPlane);
                dec.insert(dec.end(), last.begin(), last.end());
                add(st, base, len, dec);
            } else { // SnippetList::getProb, src/statemodel.cc:378-393: the longest cached length <= len
                int partlen = 0;
                const std::vector<MemoChunk> *pd = nullptr;
                if (v.front().len <= len) {
                    size_t k = 0;
                    while (k + 1 < v.size() && v[k + 1].len < len) k++;
                    if (k + 1 < v.size() && v[k + 1].len == len) k++;
                    partlen = v[k].len;
                    pd = &v[k].dec;
                }
                if (partlen == len) dec = *pd;
                else if (partlen == 0) {
                    dec.push_back({base, len, curPlane});
                    add(st, base, len, dec);
                } else {
                    const std::vector<MemoChunk> part = *pd;
                    dec = get(st, base - partlen, len - partlen, curPlane);
                    dec.insert(dec.end(), part.begin(), part.end());
                }
            }
        } else {
            dec.push_back({base, len, curPlane});
            add(st, base, len, dec);
        }
        return dec;
    }

    // replay the requests of the bases t0..t1 (lists empty at t0); candidates ending at or after `from` whose chunks are not all of
    // their own plane get a new term
    void window(int t0, int t1, int from) {
        lists[0].clear(); lists[1].clear();
        flatLo = t0 - d - 64;
        for (int st = 0; st < 2; st++) { flat[st].clear(); flat[st].resize((size_t)(t1 - flatLo + 2 > 0 ? t1 - flatLo + 2 : 0)); }
        const int S2 = S;
        std::vector<int> lessF, lessR;
        for (int s = 0; s < S2; s++) {
            if (t->state_kind[s] == AUGX_K_LESSD) lessF.push_back(s);
            else if (t->state_kind[s] == AUGX_K_RLESSD) lessR.push_back(s);
        }
        struct Req { int eop; Item *item; uint64_t gidx; };
        std::vector<Req> reqs;
        // the candidates of a block, ordered by their (base, state) pair once per block (round 6: every one of the block's 8 x 6
        // (base, short-intron state) pairs went through all of the block's ~120 records)
        std::vector<uint32_t> byPair, pairFirst;
        int bucketBlock = -1;
        const int pidBits = dense ? 7 : 6, nPid = blk << pidBits;
        for (int j = (t0 < 1 ? 1 : t0); j <= t1 && j < n; j++) {
            const int pl = plane[j];
            const int b = j / blk;
            Item *bi = blockItems(b);
            const uint32_t cnt = bi ? blkCnt[(size_t)b * 2 + 1] : 0;
            if (b != bucketBlock) {
                bucketBlock = b;
                pairFirst.assign((size_t)nPid + 1, 0);
                for (uint32_t it = 0; it < cnt; it++) { const uint32_t pid = bi[it].kp >> KEY_BITS; if (pid < (uint32_t)nPid) pairFirst[(size_t)pid + 1]++; }
                for (int q = 0; q < nPid; q++) pairFirst[(size_t)q + 1] += pairFirst[(size_t)q];
                byPair.resize(cnt);
                std::vector<uint32_t> w(pairFirst.begin(), pairFirst.end() - 1);
                for (uint32_t it = 0; it < cnt; it++) { const uint32_t pid = bi[it].kp >> KEY_BITS; if (pid < (uint32_t)nPid) byPair[w[(size_t)pid]++] = it; }
            }
            for (int st = 0; st < 2; st++) {
                const std::vector<int> &states = st == 0 ? lessF : lessR;
                for (int s : states) {
                    const uint32_t pid = dense ? (uint32_t)(((j % blk) << 7) | s) : (uint32_t)(((j % blk) << 6) | s);
                    const int a = t->anc[s][0];
                    reqs.clear();
                    for (uint32_t bq = pairFirst[(size_t)pid]; bq < pairFirst[(size_t)pid + 1]; bq++) {
                        const uint32_t it = byPair[bq];
                        const Item &I = bi[it];
                        if ((I.kp >> KEY_BITS) != pid || !(I.te > -INFINITY)) continue;
                        const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
                        const uint32_t tag = dense ? (eop <= 0 ? SRC_COL0 : SRC_LIST) : I.src >> 30;
                        // (a request is made only where a predecessor cell is alive; column 0 holds the initial probabilities)
                        double pv;
                        if (dense) pv = eop <= 0 ? (F0 ? F0 : F)[(size_t)(I.src & 127u)] : (eop >= fRow0 ? F[(size_t)(eop - fRow0) * S2 + (I.src & 127u)] : -INFINITY);
                        else if (F) pv = tag == SRC_COL0 ? (F0 ? F0 : F)[(size_t)(I.src & 0x3Fu)] : (eop >= fRow0 ? F[(size_t)(eop - fRow0) * S2 + a] : -INFINITY);
                        else if (tag == SRC_COL0) pv = col0[I.src & 0x3Fu];
                        else pv = (((I.src >> 26) & 3) == 2 ? ldVal : rdVal)[(size_t)(I.src & 0xFFFFFFu) * 3 + ((I.src >> 24) & 3)];
                        if (!(pv > -INFINITY)) continue;
                        reqs.push_back({tag == SRC_COL0 ? 0 : eop, bi + it, blkOff[(size_t)b * 2 + 1] + it});
                    }
                    std::sort(reqs.begin(), reqs.end(), [](const Req &x, const Req &y) { return x.eop > y.eop; }); // from near to far
                    for (const Req &rq : reqs) {
                        const int len = j - rq.eop; // bases eop + 1 .. j
                        const std::vector<MemoChunk> dec = get(st, j, len, pl);
                        if (j < from) continue;
                        bool mixed = false;
                        for (const MemoChunk &c : dec) mixed = mixed || c.plane != pl;
                        if (!mixed) continue;
                        int64_t fx = 0;
                        for (const MemoChunk &c : dec) fx += segFx(c.plane, st == 1, c.end - c.len + 1, c.end);
                        // the term as K2a builds it (kernels.h: varEvalItem): transition + (length + content)
                        const bool fwd = st == 0;
                        const int begin = rq.eop + 1;
                        const int bobi = fwd ? begin - t->De - 2 : begin - (t->U + t->As + 2);
                        const int eobi = fwd ? j + t->U + t->As + 2 : j + t->De + 2;
                        int intronLength = eobi - bobi + 1;
                        if (intronLength > t->d || intronLength < 0) intronLength = 0;
                        const int cls = planeCls[pl];
                        const double tr = t->ln_trans[((int64_t)cls * S2 + a) * S2 + s];
                        const double te = tr + (t->len_intron[intronLength] + (double)fx * AUGX_FX_INV);
                        if (te != rq.item->te) { rq.item->te = te; patches.push_back({rq.gidx, te}); }
                    }
                }
            }
        }
    }

    // all windows of the piece: a class step at base b can leave its mark on candidates ending in [b, b + 2 d]; the caches that
    // matter then were started no earlier than b - d
    int run() {
        patches.clear();
        if (nPlanes <= 1) return 0;
        std::vector<int> steps;
        for (int j = 1; j < n; j++)
            if (plane[j] != plane[j - 1]) steps.push_back(j);
        std::vector<std::pair<int, int>> wins; // {first step, t1} of every window
        for (size_t i = 0; i < steps.size();) {
            const int first = steps[i];
            int t1 = first + 2 * d + 64;
            size_t k = i + 1;
            while (k < steps.size() && steps[k] - d - 64 <= t1) { t1 = steps[k] + 2 * d + 64; k++; }
            wins.push_back({first, t1});
            i = k;
        }
        if (prefetch) {
            std::vector<std::pair<int, int>> tt;
            for (auto &w : wins) tt.push_back({w.first - d - 64, w.second});
            const int rc = prefetch(tt);
            if (rc) return rc;
        }
        for (auto &w : wins) {
            if (fetch) { const int rc = fetch(w.first - d - 64, w.second); if (rc) return rc; }
            window(w.first - d - 64, w.second, w.first);
        }
        return 0;
    }
};

} // namespace dev
} // namespace augx
