/*
 * ghmm_twin.cc -- CPU restatement ("twin") of the reference's single-genome GHMM Viterbi decode.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; nothing under augustus_amd/ links, imports or executes it.
 *
 * What it is: a plain, position-sequential, brute-force restatement of
 *     NAMGene::viterbiAndForward      reference src/namgene.cc:168-365
 *     NAMGene::getViterbiPath         reference src/namgene.cc:432-510
 * and of the per-state scorers it drives (cited at each function below), with the reference's loop
 * structure and tie-breaking kept, but in log space: probabilities are fp64 natural logs, products are
 * sums, and every Markov-chain content product  prod_{p=l..r} e(p)  is the difference P[r]-P[l-1] of a
 * prefix-sum array (the same formulation the HIP kernels use).  Extrinsic evidence is absent (ab initio;
 * every malus/bonus factor of the reference is exactly 1, reference config/extrinsic/extrinsic.cfg).
 *
 * Parity pin: tests/test_oracle.py checks this file against the REAL reference built by
 * oracle/Makefile (oracle/_ref/ref_harness: ln Viterbi score to 1e-9 relative, state path exact, every
 * trellis cell to 1e-9) and against the committed golden vectors under tests/golden/.
 *
 * Multi-GC-class pieces: the reference answers the content of a short (lessD) intron from its SnippetProbs
 * cache (src/statemodel.cc:312-393), which is not emptied when the class changes, so that near a class
 * step an interior can be the product of pieces scored under different classes, in call-history order.
 * The cache is restated here (snipGet below) and is on by default; twin_set_snippet_cache(0) scores every
 * interior with the class of its END base instead -- what the first pass of the device computes, kept so
 * that this pass can be checked on its own.
 *
 * Known, documented deviation (not observed on any test input):
 *  - IntronModel::codon is a shared scratch buffer in the reference; for short introns that begin before
 *    sequence position 2 the spliced-codon stop test reads stale bytes (src/intronmodel.cc:935-958).
 *    Here the test is skipped for those (left-truncated) introns.
 */
#include <cmath>
#include <cstdint>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <vector>
#include "../include/augx.h"

namespace {

int g_snippetCache = 1; // twin_set_snippet_cache
// Entry 0 of the reference's tssProbsPlus / tssProbsMinus lives on from sequence to sequence while the sequences keep one length: it is
// neither cleared at a class step (updateToLocalGC clears [from, to) with from = 1, src/utrmodel.cc:779-781) nor re-allocated
// (initAlgorithms, :744-747).  twin_set_tss0_carry(1): consecutive twin_decode calls are the reference's consecutive sequences
// (off by default: every call starts with empty caches; twin_set_tss0_carry(0) also forgets what was carried)
int g_tss0Carry = 0;
long g_tss0Size = -1;
double g_tss0Val[2] = {0, 0};
int g_tss0Set[2] = {0, 0};

const double NINF = -std::numeric_limits<double>::infinity();
inline int mod3(int k) { return k >= 0 ? k % 3 : (k % 3 + 3) % 3; }

struct Twin {
    const augx_tables &t;
    int n;
    std::vector<uint8_t> code; // 0..3 acgt, 4 invalid
    std::vector<int> cls;      // GC class per position
    int S, k, NP;
    // prefix sums per class, allocated lazily; index [p+1], P[0]=0
    struct ClassArrays {
        bool built = false;
        // exactly-associative fixed-point prefix sums: term = llrint(ln p * 2^AUGX_FX_SHIFT), wrap-around uint64
        std::vector<uint64_t> inF, inR;     // intron fwd pattern / rc pattern (rlessD)
        std::vector<uint64_t> ex[2][3][3];  // [strand][table: 0 emi,1 init,2 et][phase]
    };
    std::vector<ClassArrays> ca;
    std::vector<int> nsF, nsR;              // nearestStopForward / Reverse
    std::vector<double> V;                  // [n][S]
    std::vector<int32_t> bpS, bpE;          // back pointers: pred state, endOfPred

    Twin(const augx_tables &tt, const char *seq, int len) : t(tt), n(len) {
        S = t.S; k = t.k; NP = 1 << (2 * (k + 1));
        code.resize(n + 16, 4);
        for (int i = 0; i < n; i++) {
            char c = (char)tolower((unsigned char)seq[i]); // reference lower-cases the whole sequence,
            code[i] = c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : c == 't' ? 3 : 4; // src/extrinsicinfo.cc:1726
        }
        ca.resize(t.n_classes);
        // soft-masking: lower-case bases are nonexonpart hints (reference src/extrinsicinfo.cc:1696-1724)
        soft.assign(n + 1, 0);
        softCnt.assign(n + 2, 0);
        for (int i = 0; i < n; i++) {
            soft[i] = (t.softmasking && seq[i] >= 'a' && seq[i] <= 'z') ? 1 : 0;
            softCnt[i + 1] = softCnt[i] + soft[i];
        }
    }
    std::vector<uint8_t> soft;   // soft-masked base?
    std::vector<int64_t> softCnt; // prefix count, softCnt[p+1] = number of soft-masked bases in [0, p]
    inline double softB(int p) const { return soft[p] ? t.ln_soft_bonus : 0.0; }
    // bonus of the soft-masked bases in [a, b] (reference src/intronmodel.cc:1011-1036: one factor per covered base)
    inline double softIn(int a, int b2) const {
        if (!t.softmasking) return 0.0;
        if (a < 0) a = 0;
        if (b2 < a) return 0.0;
        return (double)(softCnt[b2 + 1] - softCnt[a]) * t.ln_soft_bonus;
    }
    inline int b(int p) const { return (p >= 0 && p < n) ? code[p] : 4; }
    inline bool is2(int p, int x, int y) const { return b(p) == x && b(p + 1) == y; }
    // forward pattern of `len` bases starting at p (first base most significant); -1 if any invalid
    int pat(int p, int len) const {
        int r = 0;
        for (int i = 0; i < len; i++) {
            int c = b(p + i);
            if (c > 3) return -1;
            r = (r << 2) | c;
        }
        return r;
    }
    // Seq2Int::rc (reference include/geneticcode.hh:174-179): sum complement(s[i]) << 2i
    int rcpat(int p, int len) const {
        int r = 0;
        for (int i = 0; i < len; i++) {
            int c = b(p + i);
            if (c > 3) return -1;
            r |= (3 - c) << (2 * i);
        }
        return r;
    }
    // which codons end a reading frame follows the translation table (GeneticCode::isStopcodon = translate() == '*',
    // include/geneticcode.hh:327-329; OpenReadingFrame::isStopcodon, src/exonmodel.cc:204-219): t.stop_mask bit 0 taa, 1 tag, 2 tga
    bool stopCodon3(int c0, int c1, int c2) const {
        if (c0 > 3 || c1 > 3 || c2 > 3) return false; // translate() -> 'X' (src/geneticcode.cc:198-204)
        if (c0 != 3) return false;
        if (c1 == 0) return (c2 == 0 && (t.stop_mask & 1)) || (c2 == 2 && (t.stop_mask & 2));
        return c1 == 2 && c2 == 0 && (t.stop_mask & 4);
    }
    bool isStop(int p) const { return stopCodon3(b(p), b(p + 1), b(p + 2)); }
    bool isRCStop(int p) const { // the reverse complement of p..p+2
        const int c0 = b(p + 2), c1 = b(p + 1), c2 = b(p);
        return stopCodon3(c0 <= 3 ? 3 - c0 : 4, c1 <= 3 ? 3 - c1 : 4, c2 <= 3 ? 3 - c2 : 4);
    }

    // ------------------------------------------------------------------------------------------
    // GC-content stairs: reference ContentStairs::computeStairs, src/motif.cc:543-616, with
    // ContentDecomposition::getNearestBaseCountIndex :493-505 and BaseCount::doubleWeight :105-177
    // ------------------------------------------------------------------------------------------
    int nearestClass(const int cnt[4]) const {
        double r[4] = {0.25, 0.25, 0.25, 0.25};
        double sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (sum > 0.0)
            for (int i = 0; i < 4; i++) r[i] = cnt[i] / sum;
        double maxW = -1;
        int ret = -1;
        for (int c = 0; c < t.n_classes; c++) {
            double w = 1;
            if (t.gc_weighing_type == 3) {
                double z[4], tmp[4] = {0, 0, 0, 0};
                for (int i = 0; i < 4; i++) z[i] = r[i] - t.gc_zus[c][i];
                for (int j = 0; j < 4; j++)
                    for (int i = 0; i < 4; i++) tmp[j] += z[i] * t.gc_weight_matrix[i * 4 + j];
                double q = 0;
                for (int i = 0; i < 4; i++) q += tmp[i] * z[i];
                w = 1 + 9 * exp(-q);
            } else if (t.gc_weighing_type == 2) {
                auto gcc = [](double g) { return g < .43 ? 0 : g < .51 ? 1 : g < .57 ? 2 : 3; };
                w = gcc(r[1] + r[2]) == gcc(t.gc_zus[c][1] + t.gc_zus[c][2]) ? 1 : 0;
            }
            if (w > maxW) { maxW = w; ret = c; }
        }
        return ret;
    }
    void computeStairs() {
        cls.assign(n, -1);
        int win = t.gc_win;
        if (win > n || win < 1) win = n;
        int cnt[4] = {0, 0, 0, 0};
        for (int i = 0; i < win; i++)
            if (code[i] < 4) cnt[code[i]]++;
        // (a window without a single nucleotide: BaseCount::normalize leaves the relative frequencies alone when the counts sum to 0,
        //  :204-212, and the object of computeStairs was normalised once, for the first window: that window's frequencies answer)
        const int first[4] = {cnt[0], cnt[1], cnt[2], cnt[3]};
        auto classOf = [&](const int c4[4]) { return nearestClass(c4[0] + c4[1] + c4[2] + c4[3] == 0 ? first : c4); };
        int x = classOf(cnt);
        for (int i = 0; i <= win / 2 && i < n; i++) cls[i] = x;
        for (int i = win / 2 + 1; i <= n - (win + 1) / 2; i++) {
            int add = i + (win + 1) / 2 - 1, sub = i - win / 2 - 1;
            if (code[add] < 4) cnt[code[add]]++;
            if (code[sub] < 4) cnt[code[sub]]--;
            cls[i] = x = classOf(cnt);
        }
        for (int i = n - (win + 1) / 2 + 1; i < n; i++) cls[i] = x;
        const int totterywin = 1000;
        x = -2;
        int lastStep = 0;
        for (int i = 0; i < n; i++)
            if (cls[i] != x) {
                if (i - lastStep < totterywin && lastStep > 0 && cls[lastStep - 1] == cls[i])
                    for (int j = lastStep; j < i; j++) cls[j] = cls[i];
                lastStep = i;
                x = cls[i];
            }
    }

    // ------------------------------------------------------------------------------------------
    // OpenReadingFrame: reference src/exonmodel.cc:101-198
    // ------------------------------------------------------------------------------------------
    void buildORF() {
        nsF.assign(n, 0);
        nsR.assign(n, 0);
        for (int r = 0; r < 3; r++) {
            int sf = -1, sr = -1;
            for (int i = r; i <= n - 3; i += 3) {
                if (isStop(i)) sf = i;
                nsF[i] = sf;
                if (isRCStop(i)) sr = i;
                nsR[i] = sr;
            }
        }
        if (n > 5) {
            nsF[n - 2] = nsF[n - 5]; nsF[n - 1] = nsF[n - 4];
            nsR[n - 2] = nsR[n - 5]; nsR[n - 1] = nsR[n - 4];
        }
    }
    int leftmostExonBegin(int frame, int base, bool fwd) const {
        int pos;
        if (fwd) pos = (frame == 0 || frame == 1) ? base - frame - 3 : base - frame;
        else pos = (frame == 1 || frame == 2) ? base + frame - 5 : base - 2;
        if (pos >= n) pos -= 3 * ((pos - n + 3) / 3);
        int lmb = pos >= 0 ? (fwd ? nsF[pos] : nsR[pos]) + 1 : 0;
        int maxAllowed = t.max_exon_len - t.U - t.As - 2 - 2 - t.Ds;
        if (lmb < base - maxAllowed) lmb = base - maxAllowed;
        return lmb;
    }

    // ------------------------------------------------------------------------------------------
    // per-class prefix sums of the order-k emissions
    // ------------------------------------------------------------------------------------------
    void buildClass(int c) {
        ClassArrays &A = ca[c];
        if (A.built) return;
        A.built = true;
        const int ki = t.k_in, NPi = 1 << (2 * (ki + 1)); // (IntronModel::k: its own order, tetrahymena 3 beside 4)
        const double *inE = t.in_emi + (size_t)c * NPi;
        A.inF.assign(n + 1, 0);
        A.inR.assign(n + 1, 0);
        for (int p = 0; p < n; p++) {
            // forward: reference IntronModel::seqProb src/intronmodel.cc:1090-1101 / SnippetProbs fwd src/statemodel.cc:287-297
            int pn = p >= ki ? pat(p - ki, ki + 1) : -1;
            A.inF[p + 1] = A.inF[p] + fx((pn >= 0 ? inE[pn] : t.ln_quarter) + softB(p));
            // reverse snippet (rlessD only): src/statemodel.cc:298-309
            int rn = (p + ki < n) ? rcpat(p, ki + 1) : -1;
            A.inR[p + 1] = A.inR[p] + fx((rn >= 0 ? inE[rn] : t.ln_quarter) + softB(p));
        }
        const double *tabs[3] = {t.ex_emi + (size_t)c * 3 * NP, t.ex_init + (size_t)c * 3 * NP, t.ex_et + (size_t)c * 3 * NP};
        for (int tb = 0; tb < 3; tb++)
            for (int a = 0; a < 3; a++) {
                std::vector<uint64_t> &F = A.ex[0][tb][a], &R = A.ex[1][tb][a];
                F.assign(n + 1, 0);
                R.assign(n + 1, 0);
                for (int p = 0; p < n; p++) {
                    // forward strand: frame f(p) = (p + a) mod 3, pattern = bases p-k..p
                    // (reference ExonModel::seqProb, src/exonmodel.cc:1957-1966)
                    int pn = p >= k ? pat(p - k, k + 1) : -1;
                    F[p + 1] = F[p] + fx(pn >= 0 ? tabs[tb][mod3(p + a) * NP + pn] : t.ln_n_coding);
                    // reverse strand: frame f(p) = (a - p) mod 3, pattern = rc of bases p..p+k
                    int rn = rcpat(p, k + 1); // reading the terminating NUL / beyond = invalid
                    R[p + 1] = R[p] + fx(rn >= 0 ? tabs[tb][mod3(a - p) * NP + rn] : t.ln_n_coding);
                }
            }
    }
    static inline uint64_t fx(double lnp) { return (uint64_t)(int64_t)llrint(lnp * AUGX_FX_SCALE); }
    inline double seg(const std::vector<uint64_t> &P, int l, int r) const {
        return l > r ? 0.0 : (double)(int64_t)(P[r + 1] - P[l]) * AUGX_FX_INV;
    }

    // ------------------------------------------------------------------------------------------
    // content of a short intron's interior: reference IntronModel::seqProb -> SnippetProbs::getSeqProb(right, len),
    // src/intronmodel.cc:1063-1067, src/statemodel.cc:312-393.  One cache per strand, a list per end base, sorted by
    // length; values here are the fixed-point ln sums (exact, so that a value put together from cached pieces equals the
    // directly computed one whenever all pieces are of one class).  What is not cached is computed with the tables of
    // the class current at the time of the request (c) -- IntronModel::updateToLocalGC, src/intronmodel.cc:495-503.
    // ------------------------------------------------------------------------------------------
    struct Snip { int len; int64_t fx; };
    std::vector<std::vector<Snip>> snips[2];
    bool useSnips = false;
    int64_t snipElem(int st, int base, int len, int c) { // SnippetProbs::getElemSeqProb :283-310
        const std::vector<uint64_t> &P = st ? ca[c].inR : ca[c].inF;
        return (int64_t)(P[base + 1] - P[base - len + 1]);
    }
    void snipAdd(int st, int base, int len, int64_t v) { // SnippetProbs::addProb :344-370 (an entry of the same length stays)
        std::vector<Snip> &L = snips[st][base];
        size_t pos = 0;
        while (pos < L.size() && L[pos].len < len) pos++;
        if (pos < L.size() && L[pos].len == len) return;
        L.insert(L.begin() + (long)pos, Snip{len, v});
    }
    int64_t snipGet(int st, int base, int len, int c) { // SnippetProbs::getSeqProb :312-342
        if (len == 0) return 0;
        std::vector<Snip> &L = snips[st][base];
        if (L.empty()) {
            const int64_t v = snipElem(st, base, len, c);
            snipAdd(st, base, len, v);
            return v;
        }
        if (L.back().len < len) { // longer than everything cached: the longest piece + the rest, and the sum is cached
            const Snip last = L.back();
            const int64_t v = snipGet(st, base - last.len, len - last.len, c) + last.fx;
            snipAdd(st, base, len, v);
            return v;
        }
        int best = -1; // SnippetList::getProb :378-393: the longest cached length <= len
        for (size_t i = 0; i < L.size() && L[i].len <= len; i++) best = (int)i;
        if (best < 0) {
            const int64_t v = snipElem(st, base, len, c);
            snipAdd(st, base, len, v);
            return v;
        }
        const Snip part = L[(size_t)best];
        if (part.len == len) return part.fx;
        return snipGet(st, base - part.len, len - part.len, c) + part.fx; // (not cached, as in the reference)
    }

    // ------------------------------------------------------------------------------------------
    // igenic emission: reference IGenicModel::emiProbUnderModel, src/igenicmodel.cc:299-357
    // ------------------------------------------------------------------------------------------
    double eIg(int c, int p) const {
        if (p > k) {
            int pn = pat(p - k, k + 1);
            return pn >= 0 ? t.ig_emi[(size_t)c * NP + pn] : t.ln_quarter;
        }
        int bk = pat(0, p + 1);
        return bk >= 0 ? t.ig_short[((size_t)c * (k + 1) + p) * NP + bk] : t.ln_quarter;
    }
    // single-base intron emission: reference IntronModel::emiProbUnderModel geometric branch, src/intronmodel.cc:895-915
    double eIn(int c, int p) const {
        const int ki = t.k_in;
        int pn = p >= ki ? pat(p - ki, ki + 1) : -1;
        return pn >= 0 ? t.in_emi[((size_t)c << (2 * (ki + 1))) + pn] : t.ln_quarter;
    }

    // splice-site gates: reference include/statemodel.hh:98-117 (no hints: consensus dinucleotides only)
    // (onGenDSS / onGenRDSS, include/geneticcode.hh:47-54: with /IntronModel/allow_dss_consensus_gc also gc)
    bool possDSS(int pos) const { return pos >= 1 && pos <= n - 2 && (is2(pos, 2, 3) || (t.dss_gc && is2(pos, 2, 1))); }      // gt at pos
    bool possRDSS(int pos) const { return pos >= 1 && pos <= n - 2 && (is2(pos - 1, 0, 1) || (t.dss_gc && is2(pos - 1, 2, 1))); } // ac at pos-1
    bool possASS(int pos) const { return pos >= 1 && pos <= n - 2 && is2(pos - 1, 0, 2); }  // ag at pos-1
    bool possRASS(int pos) const { return pos >= 1 && pos <= n - 2 && is2(pos, 1, 3); }     // ct at pos

    // Motif::seqProb forward, reference src/motif.cc:308-331
    double motifF(const double *m, int mn, int mk, int start) const {
        double s = 0;
        int sz = 1 << (2 * (mk + 1));
        for (int i = 0; i < mn; i++) {
            int pn = pat(start + i - mk, mk + 1);
            s += pn >= 0 ? m[(size_t)i * sz + pn] : t.ln_quarter;
        }
        return s;
    }
    // Motif::seqProb reverse complement
    double motifRC(const double *m, int mn, int mk, int start) const {
        double s = 0;
        int sz = 1 << (2 * (mk + 1));
        for (int i = 0; i < mn; i++) {
            int pn = rcpat(start + i, mk + 1);
            s += pn >= 0 ? m[(size_t)(mn - 1 - i) * sz + pn] : t.ln_quarter;
        }
        return s;
    }
    double tisBin(int c, double lnp) const {
        if (t.tis_nbins < 1) return lnp;
        double p = exp(lnp);
        const double *bb = t.tis_bin_bounds + (size_t)c * (t.tis_nbins - 1);
        int a = 0, bq = t.tis_nbins - 1;
        while (a < bq) {
            int m = (a + bq) / 2;
            if (p < bb[m]) bq = m; else a = m + 1;
        }
        return t.tis_bin_ln[(size_t)c * t.tis_nbins + a];
    }
    // IntronModel::dSSProb, reference src/intronmodel.cc:1195-1248.  base = first position of the pattern
    double dssProb(int base, bool fwd) const {
        int pn;
        bool nonGt; // (:1216,1224: a gc site -- Constant::dss_gc_allowed -- takes the pattern probability times non_gt_dss_prob: the table's second half)
        if (fwd) {
            int dsspos = base + t.Ds;
            if (!possDSS(dsspos)) return NINF;
            nonGt = !is2(dsspos, 2, 3);
            int a = pat(base, t.Ds), bq = pat(dsspos + 2, t.De);
            if (a < 0 || bq < 0) return NINF;
            pn = (a << (2 * t.De)) | bq;
        } else {
            int dsspos = base + t.De;
            if (!possRDSS(dsspos + 1)) return NINF;
            nonGt = !is2(dsspos, 0, 1);
            // astr = rc(s[dsspos+2 .. +Ds)) followed by rc(s[base .. +De))
            int a = 0, bq = 0;
            for (int i = 0; i < t.Ds; i++) { int c = b(dsspos + 2 + t.Ds - 1 - i); if (c > 3) return NINF; a = (a << 2) | (3 - c); }
            for (int i = 0; i < t.De; i++) { int c = b(base + t.De - 1 - i); if (c > 3) return NINF; bq = (bq << 2) | (3 - c); }
            pn = (a << (2 * t.De)) | bq;
        }
        return t.dss_pat[pn + ((nonGt && t.dss_gc) ? (1 << (2 * (t.Ds + t.De))) : 0)];
    }
    // IntronModel::aSSProb, reference src/intronmodel.cc:1116-1188.  base = first position of the motif window (fwd)
    double assProb(int c, int base, bool fwd) const {
        const double *M = t.ass_motif + (size_t)c * t.ass_n * (1 << (2 * (t.ass_k + 1)));
        double motif, patl;
        int a = 0, bq = 0;
        bool valid = true;
        if (fwd) {
            int asspos = base + t.U + t.As;
            if (!possASS(asspos + 1)) return NINF;
            for (int i = 0; i < t.As; i++) { int cc = b(base + t.U + i); if (cc > 3) valid = false; a = (a << 2) | (cc & 3); }
            for (int i = 0; i < t.Ae; i++) { int cc = b(asspos + 2 + i); if (cc > 3) valid = false; bq = (bq << 2) | (cc & 3); }
            motif = base >= t.ass_k ? motifF(M, t.ass_n, t.ass_k, base) : NINF;
        } else {
            int asspos = base + t.Ae;
            if (!possRASS(asspos)) return NINF;
            for (int i = 0; i < t.As; i++) { int cc = b(asspos + 2 + t.As - 1 - i); if (cc > 3) valid = false; a = (a << 2) | ((3 - cc) & 3); }
            for (int i = 0; i < t.Ae; i++) { int cc = b(base + t.Ae - 1 - i); if (cc > 3) valid = false; bq = (bq << 2) | ((3 - cc) & 3); }
            int motifstart = base + t.As + 2 + t.Ae, motifend = motifstart + t.U;
            motif = motifend + t.ass_k < n ? motifRC(M, t.ass_n, t.ass_k, motifstart) : t.U * t.ln_quarter;
        }
        patl = valid ? t.ass_pat[(a << (2 * t.Ae)) | bq] : t.ass_pat_invalid;
        return motif + patl;
    }

    inline double &Vat(int j, int s) { return V[(size_t)j * S + s]; }
    inline double lnT(int c, int a, int s) const { return t.ln_trans[((size_t)c * S + a) * S + s]; }

    // ------------------------------------------------------------------------------------------
    // exon scorer: reference ExonModel::viterbiForwardAndSampling src/exonmodel.cc:899-1179,
    // endPartEmiProb :1272-1400, notEndPartEmiProb :1417-1859
    // ------------------------------------------------------------------------------------------
    struct ExGeom { int bpl, ipo, baseOffset, ipeo; bool fwd; };
    ExGeom geom(int kind) const { // reference ExonModel ctor / getBaseOffset / getInnerPartEndOffset, :231-279
        ExGeom g;
        g.fwd = kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL || kind == AUGX_K_INTERNAL || kind == AUGX_K_TERMINAL;
        if (kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL) { g.bpl = 3 + t.W; g.ipo = 3; }
        else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL) { g.bpl = g.ipo = 3; }
        else { g.bpl = 0; g.ipo = g.fwd ? t.Ae : t.Ds; }
        if (kind == AUGX_K_SINGLE || kind == AUGX_K_TERMINAL) { g.baseOffset = 0; g.ipeo = 3; }
        else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RINITIAL) { g.baseOffset = -t.W; g.ipeo = 3; }
        else { g.baseOffset = g.ipeo = g.fwd ? t.Ds : t.Ae; }
        return g;
    }
    double endPart(int kind, int win, int c, int end) const {
        switch (kind) {
        case AUGX_K_SINGLE: case AUGX_K_TERMINAL: {
            int stp = end - 2;
            if (stp < 0 || stp > n - 3 || !isStop(stp)) return NINF;
            if (b(stp + 1) == 0 && b(stp + 2) == 0) return t.ln_stop_ochre;
            if (b(stp + 1) == 0 && b(stp + 2) == 2) return t.ln_stop_amber;
            return t.ln_stop_opal;
        }
        case AUGX_K_RSINGLE: case AUGX_K_RINITIAL: {
            int startpos = end - t.W - 3 + 1;
            if (startpos < 0) return NINF;
            int pn = rcpat(startpos, 3);
            if (pn < 0 || t.ln_startcodon[pn] == NINF) return NINF;
            double p = t.ln_startcodon[pn];
            if (startpos + 3 + t.W - 1 + t.tis_mem < n) {
                const double *M = t.tis_motif + (size_t)c * t.tis_n * (1 << (2 * (t.tis_k + 1)));
                p = tisBin(c, p + motifRC(M, t.tis_n, t.tis_k, startpos + 3));
            } else
                p = (n - (startpos + 3)) * t.ln_quarter; // the reference REPLACES the start codon prob here (:1329)
            return p;
        }
        case AUGX_K_INITIAL: case AUGX_K_INTERNAL: {
            int dsspos = end + t.Ds + 1;
            if (end == n - 1) return 0.0;
            if ((dsspos + 2 - 1 < n && !possDSS(dsspos)) || end + t.Ds >= n ||
                leftmostExonBegin(win - 1, end + t.Ds, true) >= end)
                return NINF;
            return 0.0;
        }
        default: { // RTERMINAL, RINTERNAL
            int asspos = end + t.Ae + 1;
            if (end == n - 1) return 0.0;
            if (end + t.Ae + 2 < n && possRASS(asspos)) return 0.0;
            return NINF;
        }
        }
    }
    double notEndPart(int kind, int win, int c, int bs, int right, int fOR, const ExGeom &g) {
        ClassArrays &A = ca[c];
        const int st = g.fwd ? 0 : 1;
        double begin;
        int bob = bs - g.ipo;
        switch (kind) {
        case AUGX_K_SINGLE: case AUGX_K_INITIAL: {
            if (!(bob >= 0 && bob < n - 2)) return NINF;
            int pn = pat(bob, 3);
            if (pn < 0 || !((t.start_mask >> pn) & 1ull)) return NINF; // isStartcodon: the translation table's ({a,c,t}tg in table 1)
            begin = t.ln_startcodon[pn];
            if (begin == NINF) return NINF;
            int tis = bob - t.W;
            if (tis > t.tis_k) {
                const double *M = t.tis_motif + (size_t)c * t.tis_n * (1 << (2 * (t.tis_k + 1)));
                begin = tisBin(c, begin + motifF(M, t.tis_n, t.tis_k, tis));
            } else
                begin = begin + (bs - 3) * t.ln_quarter;
            break;
        }
        case AUGX_K_TERMINAL: case AUGX_K_INTERNAL:
            if (bs > 0) {
                if (bob < 0 || (bob - 2 >= 0 && !possASS(bob - 1))) return NINF;
                begin = 0.0;
            } else if (bs == 0) begin = 0.0;
            else return NINF;
            break;
        case AUGX_K_RSINGLE: case AUGX_K_RTERMINAL:
            if (bob < 0) return NINF;
            if (b(bob) == 3 && b(bob + 1) == 3 && b(bob + 2) == 0) begin = t.ln_stop_ochre;      // tta
            else if (b(bob) == 1 && b(bob + 1) == 3 && b(bob + 2) == 0) begin = t.ln_stop_amber; // cta
            else if (b(bob) == 3 && b(bob + 1) == 1 && b(bob + 2) == 0) begin = t.ln_stop_opal;  // tca
            else return NINF;
            if (begin == NINF) return NINF;
            break;
        default: // RINITIAL, RINTERNAL
            if (bs == 0) begin = 0.0;
            else if (bob < 0 || (bob - 2 > 0 && !possRDSS(bob - 1))) return NINF;
            else begin = 0.0;
        }
        // ---- restSeqProb, reference :1548-1711
        double rest;
        if (bs > right) {
            rest = (bs - right - 1) * t.ln4;
        } else if (right - bs <= k) {
            int l = right - bs;
            int pn = g.fwd ? pat(bs, l + 1) : rcpat(bs, l + 1);
            if (pn >= 0) {
                int f = g.fwd ? fOR : mod3(fOR + right - bs);
                rest = t.ex_pls[(((size_t)c * (k + 1) + l) * 3 + f) * NP + pn];
            } else
                rest = (l + 1) * t.ln_n_coding;
        } else {
            int endOfStart = bs + k - 1, beginOfInitP = right - (k - 1);
            if (k == 0) rest = 0;
            else if (g.fwd) {
                int pn = pat(bs, k);
                rest = pn >= 0 ? t.ex_pls[(((size_t)c * (k + 1) + (k - 1)) * 3 + mod3(fOR - right + endOfStart)) * NP + pn]
                               : k * t.ln_n_coding;
            } else {
                int pn = rcpat(beginOfInitP, k);
                rest = pn >= 0 ? t.ex_pls[(((size_t)c * (k + 1) + (k - 1)) * 3 + mod3(fOR + right - beginOfInitP)) * NP + pn]
                               : k * t.ln_n_coding;
            }
            // phase of the prefix arrays: fwd f(p) = (p + a) mod 3 with a = fOR - right; rev f(p) = (a - p), a = fOR + right
            const int a = g.fwd ? mod3(fOR - right) : mod3(fOR + right);
            const std::vector<uint64_t> &PX = A.ex[st][0][a], &PI = A.ex[st][1][a], &PT = A.ex[st][2][a];
            int endOfInitial, beginOfTerm, endOfTerm, beginOfInitial;
            double inner;
            switch (kind) {
            case AUGX_K_SINGLE:
                endOfInitial = endOfStart + t.Li;
                if (endOfInitial > right) endOfInitial = right;
                inner = seg(PI, endOfStart + 1, endOfInitial) + seg(PX, endOfInitial + 1, right);
                break;
            case AUGX_K_INITIAL:
                endOfInitial = endOfStart + t.Li;
                if (endOfInitial > right) { endOfInitial = right; beginOfTerm = right + 1; }
                else { beginOfTerm = right - t.Le + 1; if (beginOfTerm <= endOfInitial) beginOfTerm = right + 1; }
                inner = (seg(PI, endOfStart + 1, endOfInitial) + seg(PX, endOfInitial + 1, beginOfTerm - 1)) + seg(PT, beginOfTerm, right);
                break;
            case AUGX_K_INTERNAL:
                beginOfTerm = right - t.Le + 1;
                if (beginOfTerm <= endOfStart) beginOfTerm = right + 1;
                inner = seg(PX, endOfStart + 1, beginOfTerm - 1) + seg(PT, beginOfTerm, right);
                break;
            case AUGX_K_TERMINAL:
                inner = seg(PX, endOfStart + 1, right);
                break;
            case AUGX_K_RSINGLE:
                beginOfInitial = beginOfInitP - t.Li;
                if (beginOfInitial < bs) beginOfInitial = bs;
                inner = seg(PI, beginOfInitial, beginOfInitP - 1) + seg(PX, bs, beginOfInitial - 1);
                break;
            case AUGX_K_RINITIAL:
                beginOfInitial = beginOfInitP - t.Li;
                if (beginOfInitial < bs) { beginOfInitial = bs; endOfTerm = bs - 1; }
                else { endOfTerm = bs + t.Le - 1; if (endOfTerm >= beginOfInitial) endOfTerm = bs - 1; }
                inner = (seg(PI, beginOfInitial, beginOfInitP - 1) + seg(PX, endOfTerm + 1, beginOfInitial - 1)) + seg(PT, bs, endOfTerm);
                break;
            case AUGX_K_RINTERNAL:
                endOfTerm = bs + t.Le - 1;
                if (endOfTerm >= beginOfInitP) endOfTerm = bs - 1;
                inner = seg(PX, endOfTerm + 1, beginOfInitP - 1) + seg(PT, bs, endOfTerm);
                break;
            default: // RTERMINAL
                inner = seg(PX, bs, beginOfInitP - 1);
            }
            rest = rest + inner;
        }
        // ---- length part, reference :1716-1762
        int eob = right + g.ipeo;
        int len = eob - bob + 1;
        double lenPart;
        if (len < 1 || len > t.max_exon_len) return NINF; // beyond max_exon_len the reference tables end (ORF clamp keeps len below)
        switch (kind) {
        case AUGX_K_SINGLE: case AUGX_K_RSINGLE: lenPart = len % 3 == 0 ? t.len_single[len] : NINF; break;
        case AUGX_K_INITIAL: lenPart = (len % 3 == win && len > 2) ? t.len_initial[len] : NINF; break;
        case AUGX_K_RINITIAL: lenPart = len > 2 ? t.len_initial[len] : NINF; break;
        case AUGX_K_INTERNAL: case AUGX_K_RINTERNAL: lenPart = t.len_internal[len]; break;
        case AUGX_K_TERMINAL: lenPart = t.len_terminal[len]; break;
        default: lenPart = mod3(2 - len) == win ? t.len_terminal[len] : NINF; // RTERMINAL
        }
        if (lenPart == NINF) return NINF;
        return (begin + rest) + lenPart;
    }
    void exonCell(int s, int j) {
        const int kind = t.state_kind[s], win = t.state_win[s], c = cls[j];
        const ExGeom g = geom(kind);
        double endP = endPart(kind, win, c, j);
        int eob = j + g.baseOffset, right = eob - g.ipeo;
        if (endP == NINF || right < 0) return;
        int fOR = g.fwd ? mod3(win - (eob + 1) + right) : mod3(win + eob + 1 - right);
        int eons = (kind == AUGX_K_TERMINAL || kind == AUGX_K_SINGLE) ? eob - 3 : eob;
        if (eons > n - 1) eons = n - 1;
        int feons = g.fwd ? mod3(win - 1 - eob + eons) : mod3(win + 1 + eob - eons);
        int ORFleft = leftmostExonBegin(feons, eons, g.fwd);
        int startMax = eob + g.ipo - t.min_exon_len + 1, startMin;
        if (kind == AUGX_K_RTERMINAL || kind == AUGX_K_RSINGLE)
            startMin = startMax = ORFleft + 2;
        else {
            startMin = ORFleft <= 0 ? 0 : ORFleft + g.ipo;
            if (startMax > j + g.bpl) startMax = j + g.bpl;
        }
        double best = NINF;
        int ba = -1, be = 0;
        for (int bs = startMax; bs >= startMin; bs--) {
            int eop = bs - g.bpl - 1;
            double nep = notEndPart(kind, win, c, bs, right, fOR, g);
            if (nep == NINF || eop >= n) continue;
            int col = eop >= 0 ? eop : 0;
            int bob = bs - g.ipo, len = eob - bob + 1;
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                int a = t.anc[s][ai];
                double pv = Vat(col, a);
                if (pv == NINF) continue;
                bool ok = kind == AUGX_K_SINGLE || kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL || kind == AUGX_K_INITIAL ||
                          win == mod3(g.fwd ? t.state_win[a] + len : t.state_win[a] - len);
                if (!ok) continue;
                double te = (lnT(c, a, s) + endP) + nep;
                double val = pv + te;
                if (val > best) { best = val; ba = a; be = eop; }
            }
        }
        if (best > NINF) { Vat(j, s) = best; bpS[(size_t)j * S + s] = ba; bpE[(size_t)j * S + s] = be; }
    }

    // ------------------------------------------------------------------------------------------
    // intron scorer: reference IntronModel::viterbiForwardAndSampling src/intronmodel.cc:509-858,
    // emiProbUnderModel :861-1038
    // ------------------------------------------------------------------------------------------
    void intronCell(int s, int j) {
        const int kind = t.state_kind[s], c = cls[j], f = t.state_win[s];
        ClassArrays &A = ca[c];
        const int dStateLen = t.d - 2 - t.De - t.As - 2 - t.U;
        const int dssWhole = t.Ds + 2 + t.De, assWhole = t.As + 2 + t.Ae;
        double best = NINF;
        int ba = -1, be = 0;
        if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) {
            const bool fwd = kind == AUGX_K_LESSD;
            int eobi = fwd ? j + t.U + t.As + 2 : j + t.De + 2;
            if (fwd ? (eobi - 2 + 1 < n - 1 && !possASS(eobi)) : (eobi - 2 + 1 < n - 1 && !possRDSS(eobi))) return;
            int cod[3] = {4, 4, 4};
            bool haveRight = eobi < n - 2;
            // frame of the state = number of bases of the spliced codon before the intron (fwd), see :557-580
            if (fwd) {
                if (f == 1) { cod[1] = haveRight ? b(eobi + 1) : 4; cod[2] = haveRight ? b(eobi + 2) : 4; }
                if (f == 2) { cod[2] = haveRight ? b(eobi + 1) : 4; }
            } else {
                if (f == 0) { cod[0] = haveRight ? 3 - b(eobi + 1) : 4; if (haveRight && b(eobi + 1) > 3) cod[0] = 4; }
                if (f == 1) {
                    cod[0] = (haveRight && b(eobi + 2) <= 3) ? 3 - b(eobi + 2) : 4;
                    cod[1] = (haveRight && b(eobi + 1) <= 3) ? 3 - b(eobi + 1) : 4;
                }
            }
            int left = j - dStateLen;
            if (left < 0) left = 0;
            for (int eop = j - 1; eop >= left; eop--) {
                bool any = false;
                for (int ai = 0; ai < t.n_anc[s]; ai++)
                    if (Vat(eop, t.anc[s][ai]) > NINF) any = true;
                if (!any) continue;
                // emiProbUnderModel(eop+1, j), lessD branch :924-1000
                int begin = eop + 1;
                int bobi = fwd ? begin - t.De - 2 : begin - (t.U + t.As + 2);
                if (bobi >= 0 && !(fwd ? possDSS(bobi) : possRASS(bobi))) continue;
                bool spliced = fwd ? (f != 0) : (f != 2);
                if (spliced && bobi > 1) {
                    int cc[3] = {cod[0], cod[1], cod[2]};
                    if (fwd) {
                        if (f == 1) cc[0] = b(bobi - 1);
                        else { cc[0] = b(bobi - 2); cc[1] = b(bobi - 1); }
                    } else {
                        if (f == 0) { cc[1] = b(bobi - 1) <= 3 ? 3 - b(bobi - 1) : 4; cc[2] = b(bobi - 2) <= 3 ? 3 - b(bobi - 2) : 4; }
                        else cc[2] = b(bobi - 1) <= 3 ? 3 - b(bobi - 1) : 4;
                    }
                    if (stopCodon3(cc[0], cc[1], cc[2])) continue;
                }
                int intronLength = eobi - bobi + 1;
                if (intronLength > t.d) continue; // cannot happen without hints (j - eop <= dStateLen)
                double restSeq = useSnips ? (double)snipGet(fwd ? 0 : 1, j, j - begin + 1, c) * AUGX_FX_INV
                                          : (fwd ? seg(A.inF, begin, j) : seg(A.inR, begin, j));
                double emi = t.len_intron[intronLength] + restSeq;
                if (emi == NINF) continue;
                for (int ai = 0; ai < t.n_anc[s]; ai++) {
                    int a = t.anc[s][ai];
                    double pv = Vat(eop, a);
                    if (pv == NINF) continue;
                    double val = pv + (lnT(c, a, s) + emi);
                    if (val > best) { best = val; ba = a; be = eop; }
                }
            }
        } else {
            int eop;
            double emi;
            switch (kind) {
            case AUGX_K_LONGDSS:
                eop = j - dssWhole;
                if (eop < 0 || !possDSS(j - t.De - 2 + 1)) return;
                emi = dssProb(j - dssWhole + 1, true) + softIn(j - 2 - t.De + 1, j);
                break;
            case AUGX_K_RLONGDSS:
                eop = j - dssWhole;
                if (eop < 0 || !possRDSS(j - t.Ds)) return;
                emi = dssProb(j - dssWhole + 1, false) + softIn(j - dssWhole + 1, j - t.Ds);
                break;
            case AUGX_K_EQUALD: case AUGX_K_REQUALD:
                eop = j - dStateLen;
                if (eop < 0) return;
                emi = seg(A.inF, eop + 1, j);
                break;
            case AUGX_K_GEOMETRIC: case AUGX_K_RGEOMETRIC:
                eop = j - 1;
                emi = eIn(c, j) + softB(j);
                break;
            case AUGX_K_LONGASS: {
                eop = j - assWhole - t.U;
                if (eop < 0 || !possASS(j - t.Ae)) return;
                // (the value is asked for -- and, the first time since the memo was emptied, computed -- only for a live predecessor, :729-745)
                bool any = false;
                for (int ai = 0; ai < t.n_anc[s]; ai++) any = any || Vat(eop, t.anc[s][ai]) > NINF;
                if (!any) return;
                emi = assProb(assMemoClass(j - assWhole - t.U + 1), j - assWhole - t.U + 1, true) + softIn(j - assWhole - t.U + 1, j - t.Ae);
                break;
            }
            default: // RLONGASS
                eop = j - assWhole - t.U;
                if (eop < 0 || !possRASS(j - t.U - t.As - 2 + 1)) return;
                emi = assProb(c, j - assWhole - t.U + 1, false) + softIn(j - assWhole - t.U + 1 + t.Ae, j);
            }
            if (emi == NINF) return;
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                int a = t.anc[s][ai];
                double pv = Vat(eop, a);
                if (pv == NINF) continue;
                double val = pv + (lnT(c, a, s) + emi);
                if (val > best) { best = val; ba = a; be = eop; }
            }
        }
        if (best > NINF) { Vat(j, s) = best; bpS[(size_t)j * S + s] = ba; bpE[(size_t)j * S + s] = be; }
    }
    // reference IGenicModel::viterbiForwardAndSampling, src/igenicmodel.cc:231-287
    void igenicCell(int s, int j) {
        const int c = cls[j];
        double e = eIg(c, j) + softB(j), best = NINF;
        int ba = t.n_anc[s] ? t.anc[s][0] : -1;
        for (int ai = 0; ai < t.n_anc[s]; ai++) {
            int a = t.anc[s][ai];
            double pv = Vat(j - 1, a);
            if (pv == NINF) continue;
            double val = pv + (lnT(c, a, s) + e);
            if (val > best) { best = val; ba = a; }
        }
        if (best > NINF) { Vat(j, s) = best; bpS[(size_t)j * S + s] = ba; bpE[(size_t)j * S + s] = j - 1; }
    }


    // ------------------------------------------------------------------------------------------
    // UTR states: reference UtrModel::viterbiForwardAndSampling src/utrmodel.cc:796-1064, getEndPositions :1572-1643,
    // endPartEmiProb :1072-1161, notEndPartEmiProb :1167-1548, tssProb :1761-1833, computeTtsProbs :1840-1912,
    // SegProbs src/statemodel.cc:398-460.  Ab initio: every bonus / malus is 1 but the soft-masking bonus of the intronic
    // parts (nonexonpart hints, :1149-1157, :1526-1545).
    // ------------------------------------------------------------------------------------------
    // SegProbs: cumulative sums whose term at base i is taken from the class OF THAT BASE (UtrModel::updateToLocalGC refills only
    // the region of the new class, :766-790); P[i+1] = sum over bases 0..i, base 0 counts ln 1/4 (cumProds[0] = .25)
    std::vector<uint64_t> u5iF, u5iR, u5F, u5R, u3F, u3R;
    std::vector<double> ttsPlus, ttsMinus;          // ln ttsProbPlus / ttsProbMinus per aataaa box begin
    std::vector<double> tssC[2];                    // tssProbsPlus / tssProbsMinus: cached values ...
    std::vector<uint8_t> tssSet[2];                 // ... and whether there is one
    int curCls = 0, prevCls = -1;                   // class of the column being filled
    // (the UTR content tables have the order the species' UTR file states, UtrModel::k -- 3 against 4 for chlamydomonas, culex)
    double utrEmi1(const double *tab, int c, int p, bool fwd) const { // SegProbs::getSeqProb, from == to (:437-449): the CURRENT class
        const int k = t.utr_k, NP = 1 << (2 * (k + 1));
        if (fwd) {
            if (p < k) return t.ln_quarter;
            int pn = pat(p - k, k + 1);
            return pn >= 0 ? tab[(size_t)c * NP + pn] : t.ln_quarter;
        }
        int rn = rcpat(p, k + 1); // (reads up to the terminating NUL: invalid)
        return rn >= 0 ? tab[(size_t)c * NP + rn] : t.ln_quarter;
    }
    // a base of utr5intron / utr3intron (src/utrmodel.cc:1255-1262,1389-1396): s2i_intron -- IntronModel::k + 1 bases -- read from
    // pos - UtrModel::k on; with a UTR order below the intron order the pattern ends after pos; past the piece: the NUL, invalid
    double eUin(int c, int p) const {
        const int uk = t.utr_k;
        const int ki = t.k_in;
        int pn = (p >= uk && p - uk + ki < n) ? pat(p - uk, ki + 1) : -1;
        return pn >= 0 ? t.in_emi[((size_t)c << (2 * (ki + 1))) + pn] : t.ln_quarter;
    }
    void buildUtr() {
        const int k = t.utr_k, NP = 1 << (2 * (k + 1));
        auto build = [&](std::vector<uint64_t> &P, const double *tab, bool fwd) {
            P.assign(n + 2, 0);
            for (int i = 0; i <= n; i++) {
                double v;
                if (i == 0 || i >= n) v = t.ln_quarter; // cumProds[0] = .25; base n reads the terminator
                else if (fwd) { int pn = i >= k ? pat(i - k, k + 1) : -1; v = pn >= 0 ? tab[(size_t)cls[i] * NP + pn] : t.ln_quarter; }
                else { int rn = i < n - k ? rcpat(i, k + 1) : -1; v = rn >= 0 ? tab[(size_t)cls[i] * NP + rn] : t.ln_quarter; }
                P[i + 1] = P[i] + fx(v);
            }
        };
        build(u5iF, t.utr5init_emi, true); build(u5iR, t.utr5init_emi, false);
        build(u5F, t.utr5_emi, true); build(u5R, t.utr5_emi, false);
        build(u3F, t.utr3_emi, true); build(u3R, t.utr3_emi, false);
        // computeTtsProbs(from, to) over every class region [from, to]: each box begin 1..n-1 with the class of its own base
        ttsPlus.assign(n + 1, NINF); ttsMinus.assign(n + 1, NINF);
        const int bl = t.aataaa_boxlen, dc = t.d_polyasig_cleavage;
        for (int b0 = 1; b0 < n; b0++) {
            const int c = cls[b0];
            const double *M = t.tts_motif + (size_t)c * t.tts_n * (1 << (2 * (t.tts_k + 1)));
            int ttspos = b0 + bl + dc - 1;
            if (ttspos < n) {
                int pn = pat(b0, bl);
                double prob = pn >= 0 ? t.aataaa[pn] : NINF;
                if (b0 % t.tts_spacing == 0 && prob == NINF) prob = t.ln_tts_rand;
                if (prob > NINF) prob = prob + motifF(M, t.tts_n, t.tts_k, b0 + bl);
                ttsPlus[b0] = prob;
            }
            ttspos = b0 - dc;
            if (ttspos < 0 || b0 + bl - 1 >= n) ttsPlus[b0] = NINF; // (sic: the reference zeroes the PLUS entry here, :1873-1874)
            else {
                int rn = rcpat(b0, bl);
                double prob = rn >= 0 ? t.aataaa[rn] : NINF;
                if (b0 % t.tts_spacing == 0 && prob == NINF) prob = t.ln_tts_rand;
                if (prob > NINF) prob = prob + motifRC(M, t.tts_n, t.tts_k, ttspos);
                ttsMinus[b0] = prob;
            }
        }
        for (int st = 0; st < 2; st++) { tssC[st].assign(n + 1, NINF); tssSet[st].assign(n + 1, 0); }
        if (g_tss0Carry) { // (what an earlier sequence of this length left in entry 0)
            if (g_tss0Size != (long)n + 1) { g_tss0Size = (long)n + 1; g_tss0Set[0] = g_tss0Set[1] = 0; }
            for (int st = 0; st < 2; st++) { tssSet[st][0] = (uint8_t)g_tss0Set[st]; tssC[st][0] = g_tss0Val[st]; }
        }
    }
    // UtrModel::updateToLocalGC(from, to): the cached TSS values of [from, to) are forgotten (:779-780)
    void utrEnterRegion(int from) {
        int to = from;
        while (to + 1 < n && cls[to + 1] == cls[from]) to++;
        for (int i = from; i < to; i++) tssSet[0][i] = tssSet[1][i] = 0;
    }
    double tssupSeq(int c, int left, int right, bool rev) const { // UtrModel::tssupSeqProb :1733-1750
        const int uk = t.tssup_k;
        const double *E = t.tssup_emi + (size_t)c * (1 << (2 * (uk + 1)));
        double s = 0;
        for (int p = right; p >= left; p--) {
            int pn = -1;
            if (!rev && p - uk >= 0) pn = pat(p - uk, uk + 1);
            else if (rev && p >= 0 && p + uk < n) pn = rcpat(p, uk + 1);
            s += pn >= 0 ? E[pn] : t.ln_quarter;
        }
        return s;
    }
    double tssProb(int left, bool fwd) { // :1761-1833, with the class current at the time of the first request
        const int right = left + t.tss_upwin + t.tss_end - 1;
        if (right >= n) return NINF;
        if (left % t.tts_spacing != 0) return NINF;
        const int st = fwd ? 0 : 1;
        if (tssSet[st][left]) return tssC[st][left];
        const int c = curCls;
        const size_t sz0 = (size_t)1 << (2 * (t.tss_k + 1)), sz1 = (size_t)1 << (2 * (t.tsstata_k + 1)), sz2 = (size_t)1 << (2 * (t.tata_k + 1));
        const double *Mtss = t.tss_motif + (size_t)c * t.tss_n * sz0, *Mtt = t.tsstata_motif + (size_t)c * t.tsstata_n * sz1,
                     *Mta = t.tata_motif + (size_t)c * t.tata_n * sz2;
        const int maxpos = t.d_tss_tata_max - t.d_tss_tata_min - 1;
        double prob;
        if (fwd) {
            const int w0 = right - t.tss_end - t.d_tss_tata_max + 1;
            int rel = -1;
            for (int pos = 0; pos <= maxpos; pos++)
                if (b(w0 + pos) == 3 && b(w0 + pos + 1) == 0 && b(w0 + pos + 2) == 3 && b(w0 + pos + 3) == 0 && b(w0 + pos + 5) == 0) { rel = pos; break; }
            if (rel >= 0) {
                const int tatapos = w0 + rel;
                prob = motifF(Mtt, t.tsstata_n, t.tsstata_k, right - t.tss_end - t.tss_start + 1) + motifF(Mta, t.tata_n, t.tata_k, tatapos - t.tata_start) +
                       (tssupSeq(c, left, tatapos - t.tata_start - 1, false) + tssupSeq(c, tatapos + t.tata_end, right - t.tss_end - t.tss_start, false));
            } else
                prob = motifF(Mtss, t.tss_n, t.tss_k, right - t.tss_end - t.tss_start + 1) + tssupSeq(c, left, right - t.tss_end - t.tss_start, false);
        } else {
            const int w0 = left + t.tss_end + t.d_tss_tata_max - 1;
            int rel = 1;
            for (int pos = 0; pos >= -maxpos; pos--)
                if (b(w0 + pos) == 0 && b(w0 + pos - 1) == 3 && b(w0 + pos - 2) == 0 && b(w0 + pos - 3) == 3 && b(w0 + pos - 5) == 3) { rel = pos; break; }
            if (rel <= 0) {
                const int tatapos = w0 + rel;
                prob = motifRC(Mtt, t.tsstata_n, t.tsstata_k, left) + motifRC(Mta, t.tata_n, t.tata_k, tatapos - t.tata_end + 1) +
                       (tssupSeq(c, left + t.tata_end + t.tata_start - 1, tatapos - t.tata_end, true) + tssupSeq(c, tatapos + t.tata_start + 1, right, true));
            } else
                prob = motifRC(Mtss, t.tss_n, t.tss_k, left) + tssupSeq(c, left + t.tss_end + t.tss_start, right, true);
        }
        tssC[st][left] = prob; tssSet[st][left] = 1;
        return prob;
    }
    // SegProbs::getSeqProb(from, to) :437-460
    double segU(const std::vector<uint64_t> &P, const double *tab, bool fwd, int from, int to) const {
        if (from == to) return utrEmi1(tab, curCls, to, fwd);
        if (from > to) return 0.0;
        if (to > n) to = n;
        if (from < 1) return (double)(int64_t)(P[to + 1] - P[0]) * AUGX_FX_INV;
        return (double)(int64_t)(P[to + 1] - P[from]) * AUGX_FX_INV;
    }
    // aSSProb(base, forward strand) is answered from a memo (static map memoF, src/intronmodel.cc:1120-1135, 1182-1186): a value is
    // computed -- with the class current THEN -- by whichever state asks first (a 5' UTR exon overlapping the start codon asks up to
    // W + Ae columns before the longass state does; UTR exons that begin at the site ask for thousands of columns after it) and kept
    // until the memo holds more than 1000 sites: the next call, whatever it asks for, empties it.  The memo is empty when the sweep
    // of a piece begins (IntronModel::updateToLocalGCEach -> aSSProb(-1), :440-444).  Returns the class the value comes from.
    // (The reverse-strand memo needs no restating: every request for a reverse site is made in ONE column, its own.)
    // With the caches off (twin_set_snippet_cache(0): the first pass of the device): the class of the base the longass state ends at.
    std::map<int, int> memoF;
    bool useMemo = false;
    long long memoFlushes = 0, memoForeign = 0; // (test aid: times the memo was emptied; answers that came from another class than the natural one)
    int assMemoClass(int base) {
        int q = base + t.U + t.As + 2 + t.Ae - 1;
        if (q > n - 1) q = n - 1;
        if (q < 0) q = 0;
        if (!useMemo) return cls[q];
        if (memoF.size() > 1000) { memoF.clear(); memoFlushes++; }
        auto it = memoF.find(base);
        if (it != memoF.end()) { if (it->second != cls[q]) memoForeign++; return it->second; }
        if (!possASS(base + t.U + t.As + 1)) return curCls; // (no acceptor site: the value is 0, nothing is kept, :1140-1143)
        memoF[base] = curCls;
        if (curCls != cls[q]) memoForeign++;
        return curCls;
    }
    double assProbU(int base, bool fwd) { return assProb(fwd ? assMemoClass(base) : curCls, base, fwd); }
    // (checkF != NULL: instead of the Viterbi cell, ln of the SUM over the same candidates with predecessor values taken from
    //  checkF -- the forward recurrence of one cell, for tests of the forward matrices; result in checkOut)
    const double *checkF = nullptr;
    double checkOut = NINF;
    static double lse(double a, double b2) { if (a == NINF) return b2; if (b2 == NINF) return a; return a > b2 ? a + log1p(exp(b2 - a)) : b2 + log1p(exp(a - b2)); }
    void utrCell(int s, int j) {
        const int kind = t.state_kind[s], c = cls[j];
        const int W = t.W, U = t.U, up = t.tss_upwin, te = t.tss_end, dc = t.d_polyasig_cleavage, bl = t.aataaa_boxlen;
        const int assWhole = t.As + 2 + t.Ae, dssWhole = t.Ds + 2 + t.De;
        const int ML = t.utr_max_exon_len, M3S = t.utr_max3single, M3T = t.utr_max3term;
        double best = NINF;
        int ba = -1, be = 0;
        if (kind == AUGX_K_UTR5INTRONVAR || kind == AUGX_K_UTR3INTRONVAR || kind == AUGX_K_RUTR5INTRONVAR || kind == AUGX_K_RUTR3INTRONVAR)
            return; // only introns that match a hint (:985-1041)
        if (kind == AUGX_K_UTR5INTRON || kind == AUGX_K_UTR3INTRON || kind == AUGX_K_RUTR5INTRON || kind == AUGX_K_RUTR3INTRON) {
            const double emi = eUin(c, j) + softB(j); // (:1260-1271,1395-1406: the intron model's emission, strand does not matter)
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                int a = t.anc[s][ai];
                double pv = Vat(j - 1, a);
                if (pv == NINF) continue;
                double val = pv + (lnT(c, a, s) + emi);
                if (val > best) { best = val; ba = a; be = j - 1; }
            }
            if (best > NINF) { Vat(j, s) = best; bpS[(size_t)j * S + s] = ba; bpE[(size_t)j * S + s] = be; }
            return;
        }
        // ---- getEndPositions :1572-1643
        int boep, eobe; // beginOfEndPart, endOfBioExon
        switch (kind) {
        case AUGX_K_UTR5SINGLE: case AUGX_K_UTR5TERM: boep = j + 1; eobe = j + W; break;
        case AUGX_K_RUTR5SINGLE: case AUGX_K_RUTR5INIT: boep = j - up - te + 1; eobe = j - up; break;
        case AUGX_K_UTR5INIT: case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR3INIT: case AUGX_K_UTR3INTERNAL: boep = j - dssWhole + 1; eobe = j - t.De - 2; break;
        case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5TERM: case AUGX_K_RUTR3INTERNAL: case AUGX_K_RUTR3TERM: boep = j - assWhole - U + 1; eobe = j - U - t.As - 2; break;
        case AUGX_K_RUTR3SINGLE: case AUGX_K_RUTR3INIT: boep = j + 1; eobe = j; break;
        default: // UTR3SINGLE, UTR3TERM
            if (j != n - 1) { boep = j - dc - bl + 1; eobe = j; } else { boep = n; eobe = n - 1; }
        }
        // ---- window of predecessor ends :822-916
        int lm, rm;
        switch (kind) {
        case AUGX_K_UTR5SINGLE: lm = j - (ML - W + up); rm = j - up - te - 1 + W + te; if (rm > j - 1) rm = j - 1; break;
        case AUGX_K_RUTR5SINGLE: lm = j - (ML - W + up); rm = j - up - 1 + W; if (rm > j - 1) rm = j - 1; break;
        case AUGX_K_UTR5INIT: case AUGX_K_RUTR5INIT: lm = j - (ML + 2 + t.De + up); rm = j - up - te - dssWhole; break;
        case AUGX_K_UTR5INTERNAL: case AUGX_K_RUTR5INTERNAL: case AUGX_K_UTR3INTERNAL: case AUGX_K_RUTR3INTERNAL:
            lm = j - (ML + 2 + t.De + U + t.As + 2); rm = j - dssWhole - U - assWhole; break;
        case AUGX_K_UTR5TERM: case AUGX_K_RUTR5TERM:
            lm = j - (ML - W + U + t.As + 2); rm = j - U - assWhole;
            if (-U - assWhole + W + t.Ae < 0) rm = j - U - assWhole + W + t.Ae;
            break;
        case AUGX_K_UTR3SINGLE: lm = j - M3S; rm = j != n - 1 ? j - dc - bl : j - 1; break;
        case AUGX_K_RUTR3SINGLE: lm = j - M3S; rm = j - dc - bl; break;
        case AUGX_K_UTR3INIT: case AUGX_K_RUTR3INIT: lm = j - (ML + 2 + t.De); rm = j - t.De - 2; break;
        case AUGX_K_UTR3TERM: lm = j - (M3T + 2 + t.As + U); rm = j != n - 1 ? j - dc - bl - assWhole - U : j - assWhole - U; break;
        default: lm = j - (M3T + 2 + t.As + U); rm = j - dc - bl - assWhole - U; // RUTR3TERM
        }
        // ---- endPartEmiProb :1072-1161
        double endP = 0.0;
        if (boep >= 0) {
            switch (kind) {
            case AUGX_K_UTR5SINGLE: case AUGX_K_UTR5TERM:
                if (eobe + 3 <= n - 1) { int pn = pat(eobe + 1, 3); if (pn < 0 || !((t.start_mask >> pn) & 1ull)) endP = NINF; } // GeneticCode::isStartcodon (the translation table's)
                break;
            case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR5INIT: case AUGX_K_UTR3INTERNAL: case AUGX_K_UTR3INIT:
                endP = dssProb(j - dssWhole + 1, true);
                break;
            case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5TERM: case AUGX_K_RUTR3INTERNAL: case AUGX_K_RUTR3TERM:
                endP = assProb(c, j - U - assWhole + 1, false);
                break;
            case AUGX_K_RUTR5SINGLE: case AUGX_K_RUTR5INIT: endP = tssProb(boep, false); break;
            case AUGX_K_UTR3SINGLE: case AUGX_K_UTR3TERM:
                if (j == n - 1) endP = 0.0;
                else if (boep < 0 || boep + bl - 1 >= n) endP = NINF;
                else endP = ttsPlus[boep];
                break;
            default: // RUTR3SINGLE, RUTR3INIT
                if (j + 3 > n - 1 || !isRCStop(j + 1)) endP = NINF;
            }
            // the part of the intron that lies inside the state (:1144-1157)
            if (endP > NINF && kind != AUGX_K_UTR3SINGLE && kind != AUGX_K_UTR3TERM && kind != AUGX_K_RUTR5SINGLE && kind != AUGX_K_RUTR5INIT && eobe < j)
                endP = endP + softIn(eobe + 1, j);
        } else
            endP = NINF;
        if (endP == NINF) return;
        if (kind == AUGX_K_UTR5SINGLE || kind == AUGX_K_UTR5INIT) { if (lm < -up) lm = -up; }
        else if (kind == AUGX_K_RUTR3SINGLE || kind == AUGX_K_RUTR3TERM) { if (lm < -bl - dc) lm = -bl - dc; }
        else if (lm < 0) lm = 0;
        const int eom = boep - 1; // endOfMiddle
        for (int eop = rm; eop >= lm; eop--) {
            const int col = eop > 0 ? eop : 0;
            bool any = false;
            for (int ai = 0; ai < t.n_anc[s]; ai++)
                if ((checkF ? checkF[(size_t)col * S + t.anc[s][ai]] : Vat(col, t.anc[s][ai])) > NINF) any = true;
            if (!any) continue;
            // ---- notEndPartEmiProb :1167-1548
            const int begin = eop + 1;
            double bp = 0.0, mp = 0.0, lp = 0.0;
            int bom, bobe = -1; // beginOfMiddle, beginOfBioExon
            auto lenAt = [&](const double *d, int maxl, int len) { return (len >= 0 && len <= maxl) ? d[len] : NINF; };
            switch (kind) {
            case AUGX_K_UTR5SINGLE:
                bom = begin + up + te;
                mp = eom - bom + 1 >= 0 ? segU(u5iF, t.utr5init_emi, true, bom, eom) : -(eom - bom + 1) * t.ln2;
                bobe = begin + up;
                lp = lenAt(t.len5_single, ML, eobe - bobe + 1);
                if (begin >= 0) bp = tssProb(begin, true);
                else {
                    bp = (bom - 1) * t.ln_quarter;
                    if (begin + up == 0) lp = lenAt(t.tail5_single, ML, eom - begin + 1 + W - up);
                }
                break;
            case AUGX_K_UTR5INIT:
                bom = begin + up + te;
                mp = segU(u5iF, t.utr5init_emi, true, bom, eom);
                bobe = begin + up;
                lp = lenAt(t.len5_initial, ML, eobe - bobe + 1);
                if (begin >= 0) bp = tssProb(begin, true);
                else {
                    bp = (bom - 1) * t.ln_quarter;
                    if (begin + up == 0) lp = lenAt(t.tail5_single, ML, eobe - bobe + 1);
                }
                break;
            case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR3INTERNAL: case AUGX_K_UTR3TERM: case AUGX_K_UTR5TERM: {
                bobe = begin + U + t.As + 2;
                if (kind == AUGX_K_UTR5TERM && bobe >= n) bp = NINF;
                else bp = assProbU(begin, true);
                if (bp > NINF) {
                    bom = begin + U + assWhole;
                    const bool five = kind == AUGX_K_UTR5INTERNAL || kind == AUGX_K_UTR5TERM;
                    if (kind == AUGX_K_UTR5TERM && eom - bom + 1 < 0) mp = -(eom - bom + 1) * t.ln4;
                    else mp = five ? segU(u5F, t.utr5_emi, true, bom, eom) : segU(u3F, t.utr3_emi, true, bom, eom);
                    const int len = eobe - bobe + 1;
                    if (kind == AUGX_K_UTR5INTERNAL) lp = lenAt(t.len5_internal, ML, len);
                    else if (kind == AUGX_K_UTR5TERM) lp = lenAt(t.len5_terminal, ML, len);
                    else if (kind == AUGX_K_UTR3INTERNAL) lp = lenAt(t.len3_internal, ML, len);
                    else lp = eobe != n - 1 ? lenAt(t.len3_terminal, M3T, len) : lenAt(t.tail3_single, M3S, len);
                }
                break;
            }
            case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5INIT: case AUGX_K_RUTR3INIT: case AUGX_K_RUTR3INTERNAL: {
                bp = dssProb(begin, false);
                bobe = begin + t.De + 2;
                if (bp > NINF) {
                    bom = begin + dssWhole;
                    const int len = eobe - bobe + 1;
                    if (kind == AUGX_K_RUTR5INTERNAL) { mp = segU(u5R, t.utr5_emi, false, bom, eom); lp = lenAt(t.len5_internal, ML, len); }
                    else if (kind == AUGX_K_RUTR5INIT) { mp = segU(u5iR, t.utr5init_emi, false, bom, eom); lp = lenAt(t.len5_initial, ML, len); }
                    else if (kind == AUGX_K_RUTR3INIT) {
                        mp = eom - bom + 1 >= 0 ? segU(u3R, t.utr3_emi, false, bom, eom) : -(eom - bom + 1) * t.ln4;
                        lp = lenAt(t.len3_initial, ML, len);
                    } else { mp = segU(u3R, t.utr3_emi, false, bom, eom); lp = lenAt(t.len3_internal, ML, len); }
                }
                break;
            }
            case AUGX_K_RUTR5TERM:
                bom = begin; bobe = begin - W;
                mp = eom - bom + 1 >= 0 ? segU(u5R, t.utr5_emi, false, bom, eom) : -(eom - bom + 1) * t.ln4;
                lp = lenAt(t.len5_terminal, ML, eobe - bobe + 1);
                break;
            case AUGX_K_RUTR5SINGLE:
                bom = begin; bobe = begin - W;
                mp = eom - bom + 1 >= 0 ? segU(u5iR, t.utr5init_emi, false, bom, eom) : -(eom - bom + 1) * t.ln2;
                lp = lenAt(t.len5_single, ML, eobe - bobe + 1);
                break;
            case AUGX_K_UTR3SINGLE:
                bom = bobe = begin;
                mp = segU(u3F, t.utr3_emi, true, bom, eom);
                lp = eobe != n - 1 ? lenAt(t.len3_single, M3S, eobe - bobe + 1) : lenAt(t.tail3_single, M3S, eobe - bobe + 1);
                break;
            case AUGX_K_RUTR3SINGLE: case AUGX_K_RUTR3TERM:
                bobe = begin;
                bom = begin + bl + dc;
                if (begin > 0) {
                    bp = ttsMinus[begin + dc <= n ? begin + dc : n];
                    lp = kind == AUGX_K_RUTR3SINGLE ? lenAt(t.len3_single, M3S, eobe - bobe + 1) : 0.0;
                } else {
                    bp = (kind == AUGX_K_RUTR3TERM || bom > 0) ? (bom - 1) * t.ln_quarter : 0.0;
                    lp = kind == AUGX_K_RUTR3SINGLE ? lenAt(t.tail3_single, M3S, eobe - bobe + 1) : 0.0;
                }
                if (bp > NINF) {
                    mp = segU(u3R, t.utr3_emi, false, bom, eom);
                    if (kind == AUGX_K_RUTR3TERM) lp = lenAt(t.len3_terminal, M3T, eobe - bobe + 1);
                }
                break;
            default: // UTR3INIT
                bom = bobe = begin;
                mp = eom - bom + 1 >= 0 ? segU(u3F, t.utr3_emi, true, bom, eom) : -(eom - bom + 1) * t.ln4;
                lp = lenAt(t.len3_initial, ML, eobe - bobe + 1);
            }
            double nep = (bp + mp) + lp;
            if (!(nep > NINF)) continue;
            // the part of the preceding intron that lies inside the state (:1533-1545)
            if (kind == AUGX_K_UTR5INTERNAL || kind == AUGX_K_UTR5TERM || kind == AUGX_K_UTR3INTERNAL || kind == AUGX_K_UTR3TERM ||
                kind == AUGX_K_RUTR5INTERNAL || kind == AUGX_K_RUTR5INIT || kind == AUGX_K_RUTR3INTERNAL || kind == AUGX_K_RUTR3INIT)
                nep = nep + softIn(begin, bobe - 1);
            const double emi = nep + endP;
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                int a = t.anc[s][ai];
                double pv = checkF ? checkF[(size_t)col * S + a] : Vat(col, a);
                if (pv == NINF) continue;
                double val = pv + (lnT(c, a, s) + emi);
                if (checkF) { checkOut = lse(checkOut, val); continue; }
                if (val > best) { best = val; ba = a; be = eop; }
            }
        }
        if (checkF) return;
        if (best > NINF) { Vat(j, s) = best; bpS[(size_t)j * S + s] = ba; bpE[(size_t)j * S + s] = be; }
    }

    int run(int init_kind, int term_kind, double *lnv, std::vector<augx_state> &path) {
        V.assign((size_t)n * S, NINF);
        bpS.assign((size_t)n * S, -1);
        bpE.assign((size_t)n * S, -1);
        for (int i = 0; i < S; i++) // reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150
            Vat(0, i) = init_kind == 0 ? t.ln_init[i] : (i == t.synch_state ? 0.0 : NINF);
        computeStairs();
        buildORF();
        bool anyNuc = false;
        for (int i = 0; i < n; i++) anyNuc |= code[i] < 4;
        if (!anyNuc) { // reference src/namgene.cc:205-226
            for (int j = 1; j < n; j++) {
                Vat(j, t.synch_state) = Vat(j - 1, t.synch_state) - t.ln4;
                bpS[(size_t)j * S + t.synch_state] = t.synch_state;
                bpE[(size_t)j * S + t.synch_state] = j - 1;
            }
        } else {
            for (int j = 0; j < n; j++) buildClass(cls[j]);
            // (with one class in the piece the cache cannot change a value: left out for speed)
            useSnips = false;
            for (int j = 1; j < n && g_snippetCache; j++) useSnips = useSnips || cls[j] != cls[0];
            if (useSnips) { snips[0].assign((size_t)n, {}); snips[1].assign((size_t)n, {}); }
            if (t.utr) buildUtr();
            useMemo = useSnips && t.utr; memoF.clear();
            for (int j = 1; j < n; j++) {
                curCls = cls[j];
                if (t.utr && curCls != prevCls) utrEnterRegion(j);
                prevCls = curCls;
                for (int s = 0; s < S; s++) {
                    if (!t.reachable[s]) continue;
                    int kind = t.state_kind[s];
                    if (kind == AUGX_K_IGENIC) igenicCell(s, j);
                    else if (kind <= AUGX_K_RTERMINAL) exonCell(s, j);
                    else if (kind <= AUGX_K_RLONGASS) intronCell(s, j);
                    else utrCell(s, j);
                }
            }
        }
        if (g_tss0Carry && t.utr && anyNuc) for (int st = 0; st < 2; st++) { g_tss0Set[st] = tssSet[st][0]; g_tss0Val[st] = tssC[st][0]; }
        // termination + back-tracking: reference NAMGene::getViterbiPath, src/namgene.cc:432-510
        double maxV = NINF;
        int state = -1;
        for (int i = 0; i < S; i++) {
            double tl = term_kind == 0 ? t.ln_term[i] : (i == t.synch_state ? 0.0 : NINF);
            double v = Vat(n - 1, i) + tl;
            if (v > maxV) { maxV = v; state = i; }
        }
        *lnv = maxV;
        path.clear();
        if (state < 0) return AUGX_E_NOPATH;
        int base = n - 1;
        while (base > 0) {
            int a = bpS[(size_t)base * S + state], e = bpE[(size_t)base * S + state];
            if (a < 0) return AUGX_E_NOPATH;
            augx_state st;
            st.begin = e + 1; st.end = base; st.state = (int16_t)state; st.type = (int16_t)t.state_type[state];
            path.push_back(st);
            base = e;
            state = a;
        }
        // reverse into 5'->3' order and merge single-base igenic / geometric runs
        std::vector<augx_state> out;
        for (size_t i = path.size(); i-- > 0;) {
            const augx_state &st = path[i];
            int kind = t.state_kind[st.state];
            bool mergeable = kind == AUGX_K_IGENIC || kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC || kind == AUGX_K_UTR5INTRON ||
                             kind == AUGX_K_UTR3INTRON || kind == AUGX_K_RUTR5INTRON || kind == AUGX_K_RUTR3INTRON;
            if (mergeable && !out.empty() && out.back().state == st.state && out.back().end + 1 == st.begin)
                out.back().end = st.end;
            else
                out.push_back(st);
        }
        path.swap(out);
        return 0;
    }
};

} // namespace

extern "C" {
/* test aid: ln of the forward sum of UTR exon cell (j, s) over the twin's candidates, with predecessor values from F [len][S] */
double twin_utr_forward_cell(const augx_tables *t, const char *seq, int64_t len, const double *F, int s, int j) {
    Twin tw(*t, seq, (int)len);
    tw.V.assign((size_t)len * t->S, NINF); tw.bpS.assign((size_t)len * t->S, -1); tw.bpE.assign((size_t)len * t->S, -1);
    tw.computeStairs(); tw.buildORF();
    for (int q = 0; q < (int)len; q++) tw.buildClass(tw.cls[q]);
    tw.buildUtr();
    tw.curCls = tw.cls[j];
    tw.checkF = F; tw.checkOut = NINF;
    tw.utrCell(s, j);
    return tw.checkOut;
}
/* 1 (default): short-intron interiors through the restated SnippetProbs cache, as the reference; 0: class of the end base */
void twin_set_snippet_cache(int on) { g_snippetCache = on; }
void twin_set_tss0_carry(int on) { g_tss0Carry = on; g_tss0Size = -1; g_tss0Set[0] = g_tss0Set[1] = 0; }
/* decode one piece on the CPU.  V_out (len*S doubles) and gc_out (len int32) may be NULL.
 * states_out receives at most cap records; *n_states is the number available. */
int twin_decode(const augx_tables *t, const char *seq, int64_t len, int init_kind, int term_kind, double *V_out,
                int32_t *gc_out, augx_state *states_out, int32_t cap, int32_t *n_states, double *ln_viterbi) {
    if (!t || !seq || len < 1) return AUGX_E_ARG;
    Twin tw(*t, seq, (int)len);
    std::vector<augx_state> path;
    double lnv;
    int rc = tw.run(init_kind, term_kind, &lnv, path);
    if (ln_viterbi) *ln_viterbi = lnv;
    if (V_out) memcpy(V_out, tw.V.data(), sizeof(double) * (size_t)len * t->S);
    if (gc_out) for (int64_t i = 0; i < len; i++) gc_out[i] = tw.cls[i];
    if (n_states) *n_states = (int32_t)path.size();
    if (states_out)
        for (int32_t i = 0; i < cap && i < (int32_t)path.size(); i++) states_out[i] = path[i];
    return rc;
}
}
