// The reference's memo of acceptor-site values, replayed -- host code shared by the device library and the lane-loop emulator.
// Only pieces with several GC classes under a model with UTR states need it.
//
// IntronModel::aSSProb(base, forward strand) (reference src/intronmodel.cc:1116-1188) answers from a static map: a value is computed --
// with the tables of the GC class current THEN -- by whichever state asks first and kept until the map holds more than 1000 sites; the
// next call, whatever it asks for, empties it (:1122-1129).  In the 47-state model one state asks for a site, in one column.  With UTR
// states a site has many askers: the longass states in the column the intron would end in; utr5term up to W + Ae columns BEFORE that
// (a 5' UTR exon that overlaps the start codon, src/utrmodel.cc:861-866); utr5internal / utr5term / utr3internal / utr3term for every
// end base of an exon that may begin at the site, thousands of columns later (:1207,1231,1341,1361).  A request is only made for a
// live predecessor and an open end gate (:949-960).  Near a class step the class a value comes from therefore depends on the call
// history: on which cells are alive, in sweep order (columns ascending, states ascending, predecessor ends descending).
//
// The device scores a site with the class of the base its longass state would end at.  This replay walks the requests of a piece in
// the reference's order -- from the aliveness a first run of the dense kernel left -- and finds for every site the class of every
// (re-)computation: `hist`.  The caller rebuilds the values concerned (dense.h: k1AssPatch) and runs the kernel once more.
// The memo of the reverse strand needs no replay: every request for a reverse site is made in one column, its own.
// The memo is empty when the sweep of a piece begins (IntronModel::updateToLocalGCEach -> aSSProb(-1), :440-444).
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>
#include "dp.h"

namespace augx {
namespace dev {

struct AssHist { int32_t site; uint32_t key; int32_t pl; }; // the value of `site` is computed with the class of plane `pl` by the call of (column, state) = (key >> 7, key & 127)
AUGX_HD uint32_t assKey(int j, int s) { return ((uint32_t)j << 7) | (uint32_t)s; }

struct AssMemoReplay {
    static constexpr int MEMO_MAX = 1000; // memoF.size() > 1000 -> emptied
    const DevTables *T = nullptr;
    int n = 0;
    const uint8_t *plane = nullptr;       // [n] plane (= GC class, numbered by first appearance) of every base
    // the acceptor sites of the piece that pass isPossibleASS, by the end q of the longass state that belongs to them, ascending (the
    // site's predecessor end is q - U - As - 2 - Ae; sites with q >= n -- the AG inside the piece, the state's end not -- included),
    // and for requester r (below) bit r: one of ITS ancestors is alive at the predecessor end
    std::vector<int32_t> siteQ, siteLi;   // (siteLi: entry of the LA list; >= its length: a site with q >= n)
    std::vector<uint8_t> siteAlive;
    const uint8_t *gate = nullptr;        // [n] bit r: the UTR exon state of requester r passes its end gate at this base
    std::vector<uint8_t> planeOwn, gateOwn; // (where the replay outlives its caller's arrays: the device library)
    // the states that call aSSProb(., true), ascending: longass (asks for the site whose q is the column), the four UTR exon kinds
    int nReq = 0, reqS[8], reqKind[8];
    // result
    std::vector<AssHist> hist;            // in the order of the calls; a re-computation with the class the site had before is left out
    std::vector<int8_t> longCls;          // [site] plane the longass states of column q get their value from (-1: they do not ask)
    long long flushes = 0, calls = 0;
    // the memo as the sweep leaves it: it lives on through the back-tracking of the Viterbi path and the sampled paths of the piece
    // (reference NAMGene::getViterbiPath / getSampledPath, src/namgene.cc:432-510,366-424: every step through a longass state or one of
    // the four UTR exon kinds asks again, with the class of the step's end base; sampler.h)
    std::vector<int32_t> epochOf;
    std::vector<int8_t> clsOf;
    int E = 0, count = 0;
    std::vector<int32_t> histFirst;       // [site + 1] its entries of histBy (time order)
    std::vector<AssHist> histBy;
    int late(int i, int pl) {             // a call after the sweep: the plane the value comes from
        calls++;
        if (count > MEMO_MAX) { E++; count = 0; flushes++; }
        if (epochOf[(size_t)i] == E) return clsOf[(size_t)i];
        epochOf[(size_t)i] = E; count++; clsOf[(size_t)i] = (int8_t)pl;
        return pl;
    }
    int sweepPlane(int i, uint32_t key) const { // the plane the value of site i came from when (column, state) = key asked during the sweep (-1: never computed)
        int pl = -1;
        for (int h = histFirst[(size_t)i]; h < histFirst[(size_t)i + 1] && histBy[(size_t)h].key <= key; h++) pl = histBy[(size_t)h].pl;
        return pl;
    }
    int siteOfQ(int q) const {
        const auto it = std::lower_bound(siteQ.begin(), siteQ.end(), q);
        return it != siteQ.end() && *it == q ? (int)(it - siteQ.begin()) : -1;
    }

    static bool asks(int kind) { return kind == AUGX_K_LONGASS || kind == AUGX_K_UTR5INTERNAL || kind == AUGX_K_UTR5TERM || kind == AUGX_K_UTR3INTERNAL || kind == AUGX_K_UTR3TERM; }
    void requesters() {
        nReq = 0;
        for (int s = 0; s < T->S && nReq < 8; s++)
            if (T->reachable[s] && asks(T->kind[s])) { reqS[nReq] = s; reqKind[nReq] = T->kind[s]; nReq++; }
    }
    int clsAt(int j) const { return plane[j]; }

    // the sites whose values have to be rebuilt, from hist (nList: length of the LA list).  Returns the sites with q >= n among them
    // (they are not on the list: their value is made on the spot with the class of the last base, dense.h: utrCandPre -- left as it is)
    int patches(int nList, std::vector<AssPatch> &out, std::vector<AssSwIn> &sw) const {
        const int nS = (int)siteQ.size();
        const std::vector<int32_t> &first = histFirst;
        const std::vector<AssHist> &by = histBy;
        int extras = 0;
        for (int i = 0; i < nS; i++) {
            const int h0 = first[(size_t)i], h1 = first[(size_t)i + 1];
            if (h0 == h1) continue;
            const int q = siteQ[(size_t)i], nat = plane[q < n ? q : n - 1];
            bool differs = longCls[(size_t)i] >= 0 && longCls[(size_t)i] != nat;
            for (int h = h0; h < h1; h++) differs = differs || by[(size_t)h].pl != nat;
            if (!differs) continue;
            if (siteLi[(size_t)i] >= nList) { extras++; continue; }
            AssPatch A;
            A.li = siteLi[(size_t)i]; A.longPl = longCls[(size_t)i]; A.basePl = (int8_t)by[(size_t)h0].pl; A.nSw = (int16_t)(h1 - h0 - 1); A.swOff = (int32_t)sw.size();
            for (int h = h0 + 1; h < h1; h++) sw.push_back({by[(size_t)h].key, by[(size_t)h].pl});
            out.push_back(A);
        }
        return extras;
    }

    // slow: every call is made (no use of what is known to be in the memo) -- the plain restatement, for tests
    void run(bool slow = false) {
        hist.clear(); flushes = 0; calls = 0;
        const int nS = (int)siteQ.size(), off = T->U + T->As + 2 + T->Ae;
        longCls.assign((size_t)nS, -1);
        epochOf.assign((size_t)nS, -1);
        clsOf.assign((size_t)nS, -1);
        E = 0; count = 0;
        int cLo[8], cHi[8], cE[8];
        for (int r = 0; r < 8; r++) { cLo[r] = 0; cHi[r] = -1; cE[r] = -1; }
        int iLong = 0; // first site with q >= j
        for (int j = 1; j < n; j++) {
            while (iLong < nS && siteQ[(size_t)iLong] < j) iLong++;
            const bool haveLong = iLong < nS && siteQ[(size_t)iLong] == j;
            const unsigned g = gate[j];
            if (!g && !haveLong) continue;
            const int c = clsAt(j);
            for (int r = 0; r < nReq; r++) {
                auto call = [&](int i) {
                    calls++;
                    if (count > MEMO_MAX) { E++; count = 0; flushes++; }
                    if (epochOf[(size_t)i] == E) return;
                    epochOf[(size_t)i] = E; count++;
                    if (clsOf[(size_t)i] != c) { clsOf[(size_t)i] = (int8_t)c; hist.push_back({i, assKey(j, reqS[r]), c}); }
                };
                if (reqKind[r] == AUGX_K_LONGASS) {
                    if (!haveLong || !((siteAlive[(size_t)iLong] >> r) & 1)) continue;
                    call(iLong);
                    longCls[(size_t)iLong] = clsOf[(size_t)iLong];
                    continue;
                }
                if (!((g >> r) & 1)) continue;
                int lm, rm;
                utrWindow(*T, reqKind[r], j, n, lm, rm);
                if (rm < lm) continue;
                const int iLo = (int)(std::lower_bound(siteQ.begin(), siteQ.end(), lm + off) - siteQ.begin());
                const int iHi = (int)(std::upper_bound(siteQ.begin(), siteQ.end(), rm + off) - siteQ.begin()) - 1;
                if (iHi < iLo) continue;
                // predecessor ends descending.  What this requester has asked for since the memo was last emptied (sites cLo..cHi) is
                // in the memo: those calls change nothing -- unless the memo is full, then the first of them empties it
                int flushedAt = -1;
                for (int i = iHi; i >= iLo;) {
                    if (!slow && cE[r] == E && i >= cLo[r] && i <= cHi[r] && count <= MEMO_MAX) { i = cLo[r] - 1; continue; }
                    if ((siteAlive[(size_t)i] >> r) & 1) {
                        const int e0 = E;
                        call(i);
                        if (E != e0) flushedAt = i;
                    }
                    i--;
                }
                cLo[r] = iLo; cHi[r] = flushedAt >= 0 ? flushedAt : iHi; cE[r] = E;
            }
        }
        histFirst.assign((size_t)nS + 1, 0);
        for (const AssHist &h : hist) histFirst[(size_t)h.site + 1]++;
        for (int i = 0; i < nS; i++) histFirst[(size_t)i + 1] += histFirst[(size_t)i];
        histBy.resize(hist.size());
        { std::vector<int32_t> w(histFirst.begin(), histFirst.end() - 1); for (const AssHist &h : hist) histBy[(size_t)w[(size_t)h.site]++] = h; } // (stable: time order within a site)
    }
};

} // namespace dev
} // namespace augx
