// Posterior sampling of state paths from the forward matrix -- host code shared by the device library (decoder.hip fills
// SamplePiece from HBM) and the lane-loop emulator (tests/emu/emu.cc fills it from its host arrays).
// Reference: NAMGene::getSampledPath (src/namgene.cc:367-426), the doSampling branches of the state models
// (src/exonmodel.cc, src/intronmodel.cc:700-800, src/igenicmodel.cc) and OptionsList (src/vitmatrix.cc:270-320).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
#include <cstdint>
#include <deque>
#include <condition_variable>
#include <memory>
#include <mutex>
#include "assmemo.h"
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>
#include "dp.h"
#include "dense.h" // (the candidates of the UTR exon states are evaluated on the host too: utrDescribe / utrCand)

// glibc's rand(): the TYPE_3 additive-feedback generator r[i] = r[i-3] + r[i-31] of random_r.c, seeded by the Lehmer
// generator 16807 * x mod 2^31-1, the first 310 outputs discarded, the result shifted right by one
struct augx_rand {
    // The outputs are made a buffer at a time and handed out by position: a draw that is spent without being looked at (a sampled
    // path draws once per base of an intergenic run, the reference draws for a list of one option too) costs an increment.  Making
    // them is three independent additions per turn (the shortest lag is 3), 0.3 ns a value on one core -- with 99 paths per piece
    // that was half of the time of the one thread every sampled piece of a run waits for.  The recurrence is linear over Z / 2^32,
    // so the 31 values before any position follow from the 31 before the buffer by one matrix product (jumpMatrix: A^L by
    // squaring, 31 x 31): a large buffer is cut into parts that are filled side by side by helper threads of the generator's own
    // (started when the first large buffer is due), and it is filled AHEAD: while the draws of one buffer are handed out the
    // helpers make the next, so the drawing thread as a rule finds it ready (a helper that has to be woken takes 0.1-0.3 ms, as
    // long as the drawing thread takes for a third of a buffer).  The buffer grows from 6 144 values (a run that draws little
    // pays little, and starts no thread) to 1.5 M (6 MB: it stays in the caches -- one of 50 MB was measured slower than the
    // single-core loop, bound by memory).
    static constexpr int LAG = 31;
    static constexpr int64_t CH0 = 3 * 2048, CHMAX = 3 * 524288, PART = 3 * 32768;
    std::unique_ptr<uint32_t[]> buf, bufNext; // x[LAG + i]: the i-th output of the buffer; x[0 .. LAG): the 31 before them
    uint32_t *x = nullptr;                    // = buf.get()
    int64_t ch = 0, pos = 0;                  // outputs in the buffer / handed out
    std::vector<uint32_t> jump;               // A^PART, row-major 31 x 31 (made when the first buffer is cut into parts)
    std::vector<uint32_t> seeds;              // [parts][31]: the values before every part of the buffer being made
    // helper threads: `gen` counts the buffers asked for, `left` the helpers still at work on the one being made (0: it is ready)
    std::vector<std::thread> helpers;
    std::mutex mu;
    std::condition_variable cvGo, cvDone;
    uint64_t gen = 0;
    int left = 0;
    uint32_t *xMake = nullptr; // the buffer being made
    int64_t parts = 0;
    bool quit = false, ahead = false; // ahead: bufNext is being made or ready
    double refillSeconds = 0; // (AUGX_TIMING / AUGX_EMU_STATS: what the drawing thread spent on buffers so far)
    explicit augx_rand(unsigned seed) {
        int32_t st[34];
        st[0] = seed == 0 ? 1 : (int32_t)seed;
        for (int i = 1; i < 31; i++) {
            const int64_t hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;
            int64_t w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            st[i] = (int32_t)w;
        }
        for (int i = 31; i < 34; i++) st[i] = st[i - 31];
        // r[0 .. 34) = st; the next value is r[34] = r[3] + r[31]: the 31 values before it are r[3 .. 34)
        buf.reset(new uint32_t[LAG]);
        x = buf.get();
        for (int i = 0; i < LAG; i++) x[i] = (uint32_t)st[3 + i];
        ch = 0; pos = 0;
        skip(310); // (random_r.c discards the first 310 outputs)
    }
    ~augx_rand() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cvGo.notify_all();
        for (auto &t2 : helpers) t2.join();
    }
    augx_rand(const augx_rand &) = delete;
    augx_rand &operator=(const augx_rand &) = delete;
    // n outputs after the 31 values at p[-31 .. -1] (n a multiple of 3)
    static void fill(uint32_t *p, int64_t n) {
        for (int64_t i = 0; i < n; i += 3) { // (p[i + 2] reads p[i - 1]: written a turn ago)
            const uint32_t a = p[i - 31] + p[i - 3], b = p[i - 30] + p[i - 2], c = p[i - 29] + p[i - 1];
            p[i] = a; p[i + 1] = b; p[i + 2] = c;
        }
    }
    // A^n for the step (s[0..30]) -> (s[1..30], s[0] + s[28]) on the 31 values before a position
    static std::vector<uint32_t> jumpMatrix(int64_t n) {
        auto mul = [](const std::vector<uint32_t> &P, const std::vector<uint32_t> &Q) {
            std::vector<uint32_t> R((size_t)LAG * LAG, 0u);
            for (int i = 0; i < LAG; i++)
                for (int k = 0; k < LAG; k++) {
                    const uint32_t pv = P[(size_t)i * LAG + k];
                    if (!pv) continue;
                    for (int j = 0; j < LAG; j++) R[(size_t)i * LAG + j] += pv * Q[(size_t)k * LAG + j];
                }
            return R;
        };
        std::vector<uint32_t> A((size_t)LAG * LAG, 0u), R((size_t)LAG * LAG, 0u);
        for (int k = 0; k + 1 < LAG; k++) A[(size_t)k * LAG + k + 1] = 1u;
        A[(size_t)(LAG - 1) * LAG + 0] = 1u; A[(size_t)(LAG - 1) * LAG + 28] = 1u;
        for (int k = 0; k < LAG; k++) R[(size_t)k * LAG + k] = 1u;
        for (; n > 0; n >>= 1) {
            if (n & 1) R = mul(R, A);
            A = mul(A, A);
        }
        return R;
    }
    // the parts w, w + nth, ... of the buffer being made: each starts from its own 31 values (the part before is being written meanwhile)
    void fillParts(int w, int nth) {
        for (int64_t k = w; k < parts; k += nth) {
            uint32_t *p = xMake + LAG + k * PART;
            uint32_t h2[LAG + 33]; // the first 33 outputs of the part from its seed values alone (a multiple of 3 that covers the lag), the rest in place
            for (int q = 0; q < LAG; q++) h2[q] = seeds[(size_t)k * LAG + q];
            fill(h2 + LAG, 33);
            for (int i = 0; i < 33; i++) p[i] = h2[LAG + i];
            fill(p + 33, PART - 33);
        }
    }
    void helperLoop(int w) {
        uint64_t seen = 0;
        for (;;) {
            int nth;
            {
                std::unique_lock<std::mutex> lk(mu);
                cvGo.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                nth = (int)helpers.size();
            }
            fillParts(w, nth);
            std::lock_guard<std::mutex> lk(mu);
            if (--left == 0) cvDone.notify_one();
        }
    }
    // ask the helpers for the CHMAX outputs after the 31 values `last`, into bufNext
    void makeNext(const uint32_t *last) {
        parts = CHMAX / PART;
        seeds.resize((size_t)parts * LAG);
        for (int q = 0; q < LAG; q++) seeds[(size_t)q] = last[q];
        for (int64_t k = 1; k < parts; k++) // the 31 values before part k + 1 are A^PART times those before part k
            for (int i = 0; i < LAG; i++) {
                uint32_t acc = 0;
                for (int j = 0; j < LAG; j++) acc += jump[(size_t)i * LAG + j] * seeds[(size_t)(k - 1) * LAG + j];
                seeds[(size_t)k * LAG + i] = acc;
            }
        xMake = bufNext.get();
        for (int q = 0; q < LAG; q++) xMake[q] = last[q];
        {
            std::lock_guard<std::mutex> lk(mu);
            gen++;
            left = (int)helpers.size();
        }
        cvGo.notify_all();
        ahead = true;
    }
    void refill() {
        const auto t0 = std::chrono::steady_clock::now();
        refillBody();
        refillSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void refillBody() { // the next buffer of outputs from the last 31
        uint32_t last[LAG];
        for (int q = 0; q < LAG; q++) last[q] = x[ch + q];
        if (ahead) { // the buffer the helpers were asked for: wait for it (as a rule it is there), take it, ask for the one after
            {
                std::unique_lock<std::mutex> lk(mu);
                cvDone.wait(lk, [&] { return left == 0; });
            }
            buf.swap(bufNext);
            x = buf.get();
            pos = 0;
            for (int q = 0; q < LAG; q++) last[q] = x[ch + q];
            makeNext(last);
            return;
        }
        const int64_t next = ch == 0 ? CH0 : (ch * 4 < CHMAX ? ch * 4 : CHMAX);
        if (next == CHMAX && helpers.empty() && jump.empty()) { // the first buffer of full size: helpers, if there are cores for them
            jump = jumpMatrix(PART);
            const unsigned hw = std::thread::hardware_concurrency();
            const int want = hw >= 32 ? 8 : hw >= 8 ? 3 : hw >= 4 ? 1 : 0;
            for (int w = 0; w < want; w++) {
                try { helpers.emplace_back(&augx_rand::helperLoop, this, w); }
                catch (const std::system_error &) { break; } // (no more threads to be had: the parts are dealt to those there are)
            }
        }
        if (next != ch) { buf.reset(new uint32_t[(size_t)(LAG + next)]); x = buf.get(); ch = next; } // (not zeroed: every value is written below)
        for (int q = 0; q < LAG; q++) x[q] = last[q];
        pos = 0;
        fill(x + LAG, ch); // (on this thread: the sizes on the way up, and every buffer when there are no helpers)
        if (ch == CHMAX && !helpers.empty()) {
            bufNext.reset(new uint32_t[(size_t)(LAG + CHMAX)]);
            for (int q = 0; q < LAG; q++) last[q] = x[ch + q];
            makeNext(last);
        }
    }
    void skip(int64_t n) {
        while (n > 0) {
            if (pos == ch) refill();
            const int64_t m = n < ch - pos ? n : ch - pos;
            pos += m;
            n -= m;
        }
    }
    uint32_t step() {
        if (pos == ch) refill();
        return x[LAG + pos++];
    }
    int next() { return (int)(step() >> 1); }
    void prefetch(int64_t ahead2) const { // the output `ahead2` positions after the next one, if it lies in this buffer
        if (pos + ahead2 < ch) __builtin_prefetch(x + LAG + pos + ahead2);
    }
};

// cycle counter of the draw loop's developer timing (AUGX_TIMING_SAMPLER=1 turns it on before the first piece; off: the loop pays one
// predictable branch).  x86: the time-stamp counter; elsewhere: the steady clock in nanoseconds.
inline bool &samplerTimingOn() { static bool on = getenv("AUGX_TIMING_SAMPLER") != nullptr; return on; }
inline uint64_t samplerTicks() {
    if (!samplerTimingOn()) return 0;
#if defined(__x86_64__) || defined(__i386__)
    return __builtin_ia32_rdtsc();
#else
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
#endif
}
namespace augx {
namespace dev {
// the large arrays of a piece on the host (hundreds of MB): 2 MB pages where the kernel grants them -- in 4 KB pages the first
// touch and the release of a GB were each a tenth of a second of page-table work
template <class T> struct HugeAllocator {
    using value_type = T;
    HugeAllocator() = default;
    template <class U> HugeAllocator(const HugeAllocator<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void *q = nullptr;
        if (bytes < ((size_t)4 << 20)) q = malloc(bytes ? bytes : 1);
        else {
            const size_t rounded = ((bytes + ((size_t)2 << 20) - 1) >> 21) << 21;
            if (posix_memalign(&q, (size_t)2 << 20, rounded) != 0) q = nullptr;
            else (void)madvise(q, rounded, MADV_HUGEPAGE);
        }
        if (!q) throw std::bad_alloc();
        return (T *)q;
    }
    void deallocate(T *q, size_t) { free(q); }
    template <class U> bool operator==(const HugeAllocator<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAllocator<U> &) const { return false; }
};
// everything of one piece the sampler reads, on the host
struct SamplePiece {
    int n = 0, S = 0, blk = 8, nPlanes = 1;
    const augx_tables *t = nullptr;
    const double *F = nullptr;    // [n][S] ln forward
    std::shared_ptr<double> Fown;   // (the device library keeps its host copy here; the deleter hands the buffer back for the next piece)
    std::vector<double, HugeAllocator<double>> sig; // [n][NSIG]
    std::vector<uint8_t> plane;   // [n] (empty: one class)
    std::vector<int32_t> planeCls;
    int cls0 = 0;
    std::vector<Item, HugeAllocator<Item>> items;   // candidates of the piece
    std::vector<uint64_t> blkOff; // [nBlocks][2] (relative item offsets are blkOff[.][1] - item0)
    std::vector<uint32_t> blkCnt; // [nBlocks][2]
    uint64_t item0 = 0;
    int igS = -1, termKind = 0;
    bool anyNuc = true;
    // filled by prepareStops (may run on another thread, ahead of the sampling): see samplePaths
    std::vector<std::vector<int32_t>> stops;
    // per chain state and stop: the options of the stop, and -- flat, for the loop that runs down a chain state -- their total and
    // the draws that take the state itself at the base before: rand() < stopThr (0: the draw is looked at in full)
    std::vector<std::vector<struct OptList>> stopList;
    std::vector<std::vector<double>> stopCum;
    std::vector<std::vector<uint32_t>> stopThr;
    bool prepared = false;
    double buildSeconds = 0; // (AUGX_TIMING: option lists built while drawing)
    long nBuilt = 0, nStops = 0, nVar = 0;
    uint64_t tkChain = 0, tkVar = 0, tkTail = 0; // (TSC ticks: runs down chain states / draws at other states / paths put together)
    // models of the dense kernels (dense.h): candidate records name their predecessor by state; the candidates of the UTR exon
    // states are evaluated from a host view of the batch (hB: the emulator's own arrays, or the mirror of one piece below)
    bool dense = false;
    const DevTables *hT = nullptr;
    const BatchView *hB = nullptr;
    int hp = 0;
    struct UtrHost { // what utrDescribe / utrCand read of one piece, fetched from HBM, laid out as a batch of one piece
        DevTables T;
        BatchView B;
        std::vector<int64_t> off, listOffs;
        std::vector<int32_t> len, cls, nPlanes, planeCls, initKind, termKind, chunkPiece;
        std::vector<uint8_t> code, gcPlane;
        std::vector<uint32_t> cnt, ucnt;
        std::vector<uint64_t> ufx;
        std::vector<double> usig, sigAll;
        std::vector<USite> sites[6];
        std::vector<LaSw> laSw;
    };
    std::shared_ptr<UtrHost> uh;
    // (UTR states on a piece with several GC classes) the reference's memo of acceptor-site values as the sweep of the piece left it
    // (assmemo.h): it lives on through the back-tracking of the Viterbi path -- vitPath, 5'->3' -- and through the sampled paths
    struct AssMemoReplay *memo = nullptr;
    std::shared_ptr<void> memoOwner;
    std::vector<augx_state> vitPath;
    mutable long memoVitDiffs = 0; // candidates of UTR exon steps of the VITERBI path whose site the back-tracking would value with another class than
                                   // the sweep did (the device keeps the sweep's arg-max there; AUGX_TIMING_SAMPLER prints the count)
    double lnT(int j, int a, int s) const {
        const int c = plane.empty() ? cls0 : planeCls[plane[j]];
        return t->ln_trans[((int64_t)c * S + a) * S + s];
    }
};
struct Opt { int state, base; double lp; };
struct OptList { std::vector<Opt> o; std::vector<double> p; double cum = 0; };

inline void sortOptions(OptList &L);
// the signal field a chain state's one-base emission is read from (dense.h: chSgi)
inline int chainSig(const augx_tables &t, int kind) {
    return kind == AUGX_K_IGENIC ? SIG_EIG : (t.utr && t.utr_k != t.k && isUtrIntronKind(kind)) ? SIG_EUIN : SIG_EIN;
}

// the options of reaching (state s, base j), in the reference's order of listing, then sorted by probability (stable)
inline void buildOptions(const SamplePiece &P, int s, int j, OptList &L) {
    const augx_tables &t = *P.t;
    const int S = P.S, kind = t.state_kind[s];
    const int dssWhole = t.Ds + 2 + t.De, assLag = t.As + 2 + t.Ae + t.U, dL = t.d - 2 - t.De - t.As - 2 - t.U;
    auto Fat = [&](int q, int a) { return q < 0 ? -INFINITY : P.F[(size_t)q * S + a]; };
    L.o.clear();
    const bool chain = isChainKind(kind);
    const bool fixed = kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS || kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS ||
                       kind == AUGX_K_EQUALD || kind == AUGX_K_REQUALD;
    if (chain || fixed) {
        const int lag = chain ? 1 : (kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) ? dssWhole : (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) ? assLag : dL;
        const int sg = kind == AUGX_K_IGENIC ? SIG_EIG : chain ? chainSig(t, kind) : kind == AUGX_K_LONGDSS ? SIG_DSSF : kind == AUGX_K_RLONGDSS ? SIG_DSSR
                       : kind == AUGX_K_LONGASS ? SIG_ASSF : kind == AUGX_K_RLONGASS ? SIG_ASSR : SIG_EQD;
        const double emi = P.sig[(size_t)j * NSIG + sg];
        const int eop = j - lag;
        if (eop >= 0 && emi > -INFINITY)
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                const int a = t.anc[s][ai];
                const double lp = Fat(eop, a) + (P.lnT(j, a, s) + emi);
                if (lp > -INFINITY) L.o.push_back({a, eop, lp});
            }
    } else if (isUtrExonKind(kind)) { // UTR exon: the site list of its window, latest predecessor end first, ancestors in their order
        UCtx X(*P.hT, *P.hB, P.hp);
        UDesc D;
        utrDescribe(X, s, j, D);
        for (int idx = 0; idx < D.total; idx++) {
            double te; int eop;
            if (!utrCand(X, D, idx, te, eop)) continue;
            for (int ai = 0; ai < t.n_anc[s]; ai++) {
                const int a = t.anc[s][ai];
                const double lp = Fat(eop > 0 ? eop : 0, a) + (P.lnT(j, a, s) + te);
                if (lp > -INFINITY) L.o.push_back({a, eop, lp});
            }
        }
    } else if (P.dense) { // coding exon / short intron of a dense model: the records of its (base, state) pair
        const int b = j / P.blk;
        const uint64_t i0 = P.blkOff[(size_t)b * 2 + 1] - P.item0;
        const uint32_t cnt = P.blkCnt[(size_t)b * 2 + 1], pid = (uint32_t)(((j % P.blk) << 7) | s);
        for (uint32_t it = 0; it < cnt; it++) { // (the predecessor cells lie all over a matrix of hundreds of MB: asked for first, read in the loop below)
            const Item &I = P.items[i0 + it];
            if ((I.kp >> KEY_BITS) != pid || !(I.te > -INFINITY)) continue;
            const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
            __builtin_prefetch(&P.F[(size_t)(eop > 0 ? eop : 0) * S + (I.src & 127u)]);
        }
        for (uint32_t it = 0; it < cnt; it++) {
            const Item &I = P.items[i0 + it];
            if ((I.kp >> KEY_BITS) != pid || !(I.te > -INFINITY)) continue;
            const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS, a = (int)(I.src & 127u);
            const double lp = Fat(eop > 0 ? eop : 0, a) + I.te;
            if (lp > -INFINITY) L.o.push_back({a, eop, lp});
        }
        int pos[AUGX_MAX_STATES];
        for (int q = 0; q < AUGX_MAX_STATES; q++) pos[q] = AUGX_MAX_STATES;
        for (int ai = t.n_anc[s] - 1; ai >= 0; ai--) pos[t.anc[s][ai]] = ai;
        std::stable_sort(L.o.begin(), L.o.end(), [&](const Opt &x, const Opt &y) { return x.base != y.base ? x.base > y.base : pos[x.state] < pos[y.state]; });
    } else { // variable-length state: the candidates of its (base, state) pair, newest first (K2a's order = the reference's loops)
        const int b = j / P.blk;
        const uint64_t i0 = P.blkOff[(size_t)b * 2 + 1] - P.item0;
        const uint32_t cnt = P.blkCnt[(size_t)b * 2 + 1], pid = (uint32_t)(((j % P.blk) << 6) | s);
        for (uint32_t it = 0; it < cnt; it++) { // (the predecessor cells lie all over a matrix of hundreds of MB: asked for first, read in the loop below)
            const Item &I = P.items[i0 + it];
            if ((I.kp >> KEY_BITS) != pid || !(I.te > -INFINITY) || (I.src >> 30) == SRC_COL0) continue;
            const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
            if (eop >= 0) __builtin_prefetch(&P.F[(size_t)eop * S + ((I.src >> 30) == SRC_VIG ? P.igS : t.anc[s][(I.src >> 28) & 3])]);
        }
        for (uint32_t it = 0; it < cnt; it++) {
            const Item &I = P.items[i0 + it];
            if ((I.kp >> KEY_BITS) != pid || !(I.te > -INFINITY)) continue;
            const uint32_t tag = I.src >> 30;
            const int ai = (int)((I.src >> 28) & 3), eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
            int a;
            double pv;
            if (tag == SRC_COL0) { a = (int)(I.src & 0x3Fu); pv = P.F[a]; } // column 0 holds the initial probabilities
            else { a = tag == SRC_VIG ? P.igS : t.anc[s][ai]; pv = Fat(eop, a); }
            const double lp = pv + I.te;
            if (lp > -INFINITY) L.o.push_back({a, eop, lp});
        }
        // the reference lists them exon start after exon start, the latest first, and for each the ancestors in their order
        // (src/exonmodel.cc:1058-1100); K2a's order within a pair depends on how its queues were served.  The order decides
        // between options of exactly equal probability (e.g. the three intron phases at the start of a piece).
        int pos[AUGX_MAX_STATES];
        for (int q = 0; q < AUGX_MAX_STATES; q++) pos[q] = AUGX_MAX_STATES;
        for (int ai = t.n_anc[s] - 1; ai >= 0; ai--) pos[t.anc[s][ai]] = ai;
        std::stable_sort(L.o.begin(), L.o.end(), [&](const Opt &x, const Opt &y) { return x.base != y.base ? x.base > y.base : pos[x.state] < pos[y.state]; });
    }
    sortOptions(L);
}
// A step through (state s, base j) of a path that is traced back after the sweep -- the Viterbi path, a sampled path -- evaluates the
// state once more (reference doBacktracking / doSampling): a longass state asks aSSProb for its site, one of the four UTR exon kinds
// for every acceptor site of its window with a live predecessor, latest first -- from the memo as it is THEN, with the class of j
// where it has to be computed again (assmemo.h).  The options of such a UTR exon step (L != NULL) carry the values the memo gives.
inline void memoStep(const SamplePiece &P, int s, int j, OptList *L) {
    const augx_tables &t = *P.t;
    AssMemoReplay &R = *P.memo;
    const int S = P.S, kind = t.state_kind[s], pl = P.plane.empty() ? 0 : P.plane[(size_t)j];
    if (kind == AUGX_K_LONGASS) {
        const int i = R.siteOfQ(j);
        if (i >= 0) (void)R.late(i, pl);
        return;
    }
    const int off = t.U + t.As + 2 + t.Ae;
    UCtx X(*P.hT, *P.hB, P.hp);
    UDesc D;
    utrDescribe(X, s, j, D);
    if (L) L->o.clear();
    const uint32_t key = assKey(j, s);
    for (int idx = 0; idx < D.total; idx++) {
        int li, xi;
        utrCandIndex(D, idx, li, xi);
        USite e;
        e.pos = 0; e.pad = 0; e.b[0] = e.b[1] = e.b[2] = AUGX_NINF;
        int eop;
        if (xi < 0) { e = X.list(D.list)[D.i1 - 1 - li]; eop = e.pos; } else eop = D.xHi - xi;
        const int col = eop > 0 ? eop : 0;
        bool any = false;
        for (int ai = 0; ai < t.n_anc[s]; ai++) any = any || P.F[(size_t)col * S + t.anc[s][ai]] > -INFINITY;
        if (!any) continue;
        const int q = eop + off, site = R.siteOfQ(q);
        int now = -1, then = -1;
        if (site >= 0) {
            now = R.late(site, pl);
            then = xi < 0 ? R.sweepPlane(site, key) : (int)P.plane[(size_t)(P.n - 1)]; // (a site past the end of the piece is valued on the spot: the class of the last base)
        }
        if (!L) { if (site >= 0 && then >= 0 && now != then) P.memoVitDiffs++; continue; }
        double te; int eop2;
        if (!utrCandFrom(X, D, xi, e, te, eop2)) continue;
        if (site >= 0 && then >= 0 && now != then) te = te + (assSiteValue(*P.hT, *P.hB, P.hp, now, q) - assSiteValue(*P.hT, *P.hB, P.hp, then, q));
        for (int ai = 0; ai < t.n_anc[s]; ai++) {
            const int a = t.anc[s][ai];
            const double lp = P.F[(size_t)col * S + a] + (P.lnT(j, a, s) + te);
            if (lp > -INFINITY) L->o.push_back({a, eop2, lp});
        }
    }
    if (L) sortOptions(*L);
}
inline bool memoAsks(const SamplePiece &P, int kind) {
    return P.memo && (kind == AUGX_K_LONGASS || kind == AUGX_K_UTR5INTERNAL || kind == AUGX_K_UTR5TERM || kind == AUGX_K_UTR3INTERNAL || kind == AUGX_K_UTR3TERM);
}

// reference OptionsList::sample, src/vitmatrix.cc:295-320
inline const Opt *drawOption(const OptList &L, augx_rand &R) {
    if (L.o.empty() || !(L.cum > 0)) return nullptr;
    const double z = (double)R.next() / 2147483647.0 * L.cum * 0.99999;
    double cumsum = 0;
    for (size_t i = 0; i < L.o.size(); i++) {
        cumsum += L.p[i];
        if (z < cumsum) return &L.o[i];
    }
    return &L.o[0];
}

inline void sortOptions(OptList &L) {
    // OptionsList::add accumulates the sum in listing order; prepareSampling sorts with the largest first (std::list::sort: stable)
    double mx = -INFINITY;
    for (const Opt &x : L.o) mx = x.lp > mx ? x.lp : mx;
    L.p.resize(L.o.size());
    L.cum = 0;
    for (size_t i = 0; i < L.o.size(); i++) { L.p[i] = exp(L.o[i].lp - mx); L.cum += L.p[i]; }
    std::vector<size_t> ord(L.o.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b2) { return L.p[a] > L.p[b2]; });
    std::vector<Opt> o2(L.o.size());
    std::vector<double> p2(L.o.size());
    for (size_t i = 0; i < ord.size(); i++) { o2[i] = L.o[ord[i]]; p2[i] = L.p[ord[i]]; }
    L.o.swap(o2); L.p.swap(p2);
}

// Most steps of a path are a single-base state (intergenic region, long intron) following itself with no alternative: the
// draw is spent (the reference draws for a list of one, too) but decides nothing.  stops[s] = the bases, in increasing order,
// where chain state s has anything but that one option; between two of them a path runs through without looking.
// the draws r = rand() that take the first option of a stop -- r / RAND_MAX * total * 0.99999 < p0, the expression of
// OptionsList::sample as drawOption evaluates it, monotone in r -- are 0 .. stayThreshold - 1
inline uint32_t stayThreshold(double cum, double p0) {
    uint32_t lo = 0, hi = 2147483648u;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((double)(int)mid / 2147483647.0 * cum * 0.99999 < p0) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

inline void prepareStops(SamplePiece &P) {
    const augx_tables &t = *P.t;
    const int n = P.n, S = P.S;
    P.stops.assign((size_t)S, {});
    P.stopList.assign((size_t)S, {}); P.stopCum.assign((size_t)S, {});
    P.stopThr.assign((size_t)S, {});
    std::vector<int> chains;
    for (int s = 0; s < S && P.anyNuc; s++)
        if (isChainKind(t.state_kind[s])) chains.push_back(s);
    for (int j = 1; j < n; j++) // (one pass over the rows of F for all chain states: the matrix is 376 B a base, the pass is bound by reading it)
        for (int s : chains) {
            const int sg = chainSig(t, t.state_kind[s]);
            int cnt = 0;
            bool self = false;
            if (P.sig[(size_t)j * NSIG + sg] > -INFINITY)
                for (int ai = 0; ai < t.n_anc[s]; ai++) {
                    const int a = t.anc[s][ai];
                    if (P.F[(size_t)(j - 1) * S + a] > -INFINITY && P.lnT(j, a, s) > -INFINITY) { cnt++; self = self || a == s; }
                }
            if (!(cnt == 1 && self)) P.stops[s].push_back(j);
        }
    for (int s : chains) {
        std::vector<int32_t> &v = P.stops[s];
        // the options of every stop (the same for every sampled path; made here, ahead of the sampling and on another thread).
        // A stop whose most probable option is the state itself with everything else below 1e-5 of the total decides nothing
        // either: the draw is z = rand() / RAND_MAX * total * 0.99999 <= total * 0.99999, and the state itself is taken whenever
        // z < p(itself) -- with total * 0.99999 < p(itself) that is every draw (same expression, same rounding: the product is
        // monotone in rand()).  Such stops are dropped: the path runs through them as through any other base of the run.
        // At the others the same comparison is turned into a threshold on rand() itself (stayThreshold): the loop that runs down
        // a chain state compares two integers per stop.
        std::vector<int32_t> kept;
        for (size_t c = 0; c < v.size(); c++) {
            OptList L;
            buildOptions(P, s, v[c], L);
            const bool selfFirst = !L.o.empty() && L.o[0].state == s && L.o[0].base == v[c] - 1;
            const double p0 = L.p.empty() ? 0.0 : L.p[0];
            if (selfFirst && L.cum > 0 && 1.0 * L.cum * 0.99999 < p0) continue;
            kept.push_back(v[c]);
            P.stopCum[s].push_back(L.cum);
            P.stopThr[s].push_back(selfFirst && L.cum > 0 ? stayThreshold(L.cum, p0) : 0u);
            P.stopList[s].push_back(std::move(L));
        }
        v.swap(kept);
    }
    P.prepared = true;
#ifdef AUGX_EMU
    if (getenv("AUGX_EMU_STATS")) for (int s = 0; s < S; s++) if (!P.stops[s].empty()) fprintf(stderr, "emu sampler: chain state %d: %zu stops over %d bases\n", s, P.stops[s].size(), n);
#endif
}

// What the drawing thread keeps between the pieces of a run: the option lists of the (base, state) pairs a path came through --
// the same for every sample of a piece -- in two flat arrays behind an open-addressing table, and the work vectors.  Nothing is
// given back to the allocator between pieces: the thread every sampled piece waits for took page faults on fresh heap pages
// behind the helper threads that map and unmap the pieces' buffers (hundreds of MB each, one lock per process) -- up to 100 ms
// in a piece whose draws take 25.
struct SamplerScratch {
    struct Ent { uint64_t key; uint32_t first, cnt; double cum; }; // key: ((base << 8) | state) + 1, 0: free
    std::vector<Ent> table;
    std::vector<Opt> o;
    std::vector<double> p;
    size_t used = 0;
    OptList tmp;
    std::vector<augx_state> st;
    void reset() {
        if (table.empty()) table.assign(1u << 15, Ent{0, 0, 0, 0.0});
        else if (used) std::fill(table.begin(), table.end(), Ent{0, 0, 0, 0.0});
        used = 0; o.clear(); p.clear();
    }
    static size_t slotOf(uint64_t key, size_t mask) { return (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & mask; }
    Ent *find(uint64_t key) {
        const size_t mask = table.size() - 1;
        for (size_t i = slotOf(key, mask);; i = (i + 1) & mask) {
            if (table[i].key == key) return &table[i];
            if (table[i].key == 0) return nullptr;
        }
    }
    Ent *insert(uint64_t key, const OptList &L) {
        if ((used + 1) * 2 > table.size()) { // grow: twice the slots, every entry put again
            std::vector<Ent> old((size_t)table.size() * 2, Ent{0, 0, 0, 0.0});
            old.swap(table);
            const size_t mask = table.size() - 1;
            for (const Ent &e : old)
                if (e.key) { size_t i = slotOf(e.key, mask); while (table[i].key) i = (i + 1) & mask; table[i] = e; }
        }
        const size_t mask = table.size() - 1;
        size_t i = slotOf(key, mask);
        while (table[i].key) i = (i + 1) & mask;
        table[i] = Ent{key, (uint32_t)o.size(), (uint32_t)L.o.size(), L.cum};
        o.insert(o.end(), L.o.begin(), L.o.end());
        p.insert(p.end(), L.p.begin(), L.p.end());
        used++;
        return &table[i];
    }
};

// n_samples paths of one piece, 5'->3', runs of the single-base chain states merged (as the Viterbi path is delivered)
inline void samplePaths(SamplePiece &P, int n_samples, augx_rand &R, std::vector<std::vector<augx_state>> &paths, std::vector<int> &status) {
    const augx_tables &t = *P.t;
    const int n = P.n, S = P.S;
    for (int s2 = 0; s2 < S; s2++)
        if (t.state_kind[s2] == AUGX_K_IGENIC && P.igS < 0) P.igS = s2;
    paths.assign((size_t)n_samples, {});
    status.assign((size_t)n_samples, AUGX_OK);
    static thread_local SamplerScratch scratch; // (the options of a (base, state) pair are the same for every sample)
    SamplerScratch &M = scratch;
    M.reset();
    std::vector<augx_state> &st = M.st;
    if (!P.prepared) prepareStops(P);
    std::vector<std::vector<int32_t>> &stops = P.stops;
    std::vector<int64_t> cur((size_t)S, 0); // per sample: index of the last stop <= the base the path was last at in this state
    if (P.memo) // the Viterbi path was traced back before the first sample was drawn (reference NAMGene::findGenes, src/namgene.cc:790): 3'->5'
        for (size_t i = P.vitPath.size(); i-- > 0;)
            if (memoAsks(P, t.state_kind[P.vitPath[i].state]) && P.vitPath[i].end > 0) memoStep(P, P.vitPath[i].state, P.vitPath[i].end, nullptr);
    OptList memoList;
    for (int it = 0; it < n_samples; it++) {
        st.clear();
        // a piece without a nucleotide: one intergenic state, no draw (reference src/namgene.cc:380-384)
        if (!P.anyNuc) st.push_back({0, n - 1, (int16_t)t.synch_state, (int16_t)t.state_type[t.synch_state]});
        else {
            OptList last;
            for (int i = 0; i < S; i++) {
                const double tl = P.termKind == 0 ? t.ln_term[i] : (i == t.synch_state ? 0.0 : -INFINITY);
                const double lp = P.F[(size_t)(n - 1) * S + i] + tl;
                if (lp > -INFINITY) last.o.push_back({i, n - 1, lp});
            }
            sortOptions(last);
            const Opt *c = drawOption(last, R);
            if (!c) { status[it] = AUGX_E_NOPATH; continue; }
            int base = c->base, state = c->state;
            bool bad = false;
            for (int s2 = 0; s2 < S; s2++) cur[s2] = (int64_t)stops[s2].size() - 1;
            while (base > 0) {
                const int kd = t.state_kind[state];
                if (isChainKind(kd)) {
                    const std::vector<int32_t> &v = stops[state];
                    // the last stop <= base: bases only fall along a path, so the cursor of the state only moves down -- a step
                    // at a time while the path stays in the state, by bisection when it comes back to it further down
                    int64_t c = cur[state];
                    if (c >= (int64_t)v.size()) c = (int64_t)v.size() - 1;
                    if (c >= 0 && v[c] > base) { // (by doubling steps from where the cursor is: the path comes back a gene further down, a few dozen stops)
                        int64_t hi = c, stepDown = 1;
                        while (c >= 0 && v[c] > base) { hi = c; c -= stepDown; stepDown <<= 1; }
                        c = (int64_t)(std::upper_bound(v.begin() + (c < 0 ? 0 : c), v.begin() + hi, base) - v.begin()) - 1;
                    }
                    // down the state, stop after stop: between two stops a draw per base decides nothing; at a stop the draw is
                    // compared with the most probable option first -- as a rule the state itself, and the run goes on
                    const uint64_t tk0 = samplerTicks();
                    const int top = base;
                    const double *cumA = P.stopCum[state].data();
                    const uint32_t *thrA = P.stopThr[state].data();
                    const Opt *x = nullptr;
                    for (;;) {
                        const int stop = c < 0 ? 0 : v[c];
                        if (stop < base) { R.skip(base - stop); base = stop; }
                        if (base == 0) break;
                        // (the draws of the stops to come lie a cache line or more apart in the generator's buffer, which helper
                        //  threads have just written: the one eight stops down is asked for now)
                        if (c >= 8) R.prefetch((int64_t)(base - v[c - 8]));
                        int r = -1;
                        P.nStops++;
                        if (thrA[c]) { // (the stop has options and the state itself first among them)
                            r = R.next();
                            if ((uint32_t)r < thrA[c]) { base--; c--; if (base == 0) break; continue; }
                        }
                        const OptList &L = P.stopList[state][(size_t)c];
                        if (L.o.empty() || !(L.cum > 0)) { bad = true; break; }
                        if (r < 0) r = R.next();
                        const double z = (double)r / 2147483647.0 * cumA[c] * 0.99999; // (reference OptionsList::sample, as drawOption)
                        double cumsum = 0;
                        x = &L.o[0];
                        for (size_t i = 0; i < L.o.size(); i++) {
                            cumsum += L.p[i];
                            if (z < cumsum) { x = &L.o[i]; break; }
                        }
                        break;
                    }
                    cur[state] = c;
                    P.tkChain += samplerTicks() - tk0;
                    if (bad) break;
                    // (st runs 3'->5'; the steps of the run are one entry: the merged form the path is delivered in)
                    const int from = x ? x->base + 1 : 1;
                    if (!st.empty() && st.back().state == (int16_t)state && st.back().begin == top + 1) st.back().begin = from;
                    else st.push_back({from, top, (int16_t)state, (int16_t)t.state_type[state]});
                    if (x) { base = x->base; state = x->state; }
                    continue;
                }
                const uint64_t tk1 = samplerTicks();
                if (memoAsks(P, kd) && kd != AUGX_K_LONGASS) { // (the options of this step depend on what the memo holds now: made afresh)
                    memoStep(P, state, base, &memoList);
                    const Opt *x = drawOption(memoList, R);
                    if (!x) { bad = true; break; }
                    st.push_back({x->base + 1, base, (int16_t)state, (int16_t)t.state_type[state]});
                    const int nb = x->base, ns = x->state;
                    base = nb; state = ns;
                    P.nVar++;
                    P.tkVar += samplerTicks() - tk1;
                    continue;
                }
                if (memoAsks(P, kd)) memoStep(P, state, base, nullptr); // (a longass step: asks, its options do not depend on the answer)
                const uint64_t key = ((uint64_t)base << 8) | (uint64_t)state;
                SamplerScratch::Ent *f = M.find(key + 1);
                P.nVar++;
                if (!f) {
                    const auto tb0 = std::chrono::steady_clock::now();
                    buildOptions(P, state, base, M.tmp);
                    f = M.insert(key + 1, M.tmp);
                    P.buildSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
                    P.nBuilt++;
                }
                // (reference OptionsList::sample, as drawOption)
                const Opt *x = nullptr;
                if (f->cnt && f->cum > 0) {
                    const double z = (double)R.next() / 2147483647.0 * f->cum * 0.99999;
                    const double *pp = M.p.data() + f->first;
                    x = M.o.data() + f->first;
                    double cumsum = 0;
                    for (uint32_t i = 0; i < f->cnt; i++) {
                        cumsum += pp[i];
                        if (z < cumsum) { x = M.o.data() + f->first + i; break; }
                    }
                }
                if (!x) { bad = true; break; }
                st.push_back({x->base + 1, base, (int16_t)state, (int16_t)t.state_type[state]});
                base = x->base; state = x->state;
                P.tkVar += samplerTicks() - tk1;
            }
            if (bad) { status[it] = AUGX_E_NOPATH; continue; }
        }
        const uint64_t tk2 = samplerTicks();
        std::vector<augx_state> &m2 = paths[it];
        for (size_t i = st.size(); i-- > 0;) {
            const augx_state &x = st[i];
            const int k = t.state_kind[x.state];
            const bool chain = isChainKind(k);
            if (chain && !m2.empty() && m2.back().state == x.state && m2.back().end + 1 == x.begin) m2.back().end = x.end;
            else m2.push_back(x);
        }
        P.tkTail += samplerTicks() - tk2;
    }
}
} // namespace dev
} // namespace augx
