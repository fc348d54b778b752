// sharded.cc -- multi-GPU decode of independent pieces: one decoder (= one device, one HIP stream) and one host thread per
// GPU, pieces assigned by longest-processing-time-first bin packing (cost = length), results gathered in input order.
// There is no exchange step in the DP (contigs and pieces are independent), hence no collective: the
// "gather" is the host threads writing their slots of the caller's result array.
// Replaces nothing of the reference one to one (it is single-threaded); it is the fan-out of NAMGene::doViterbiPiecewise's
// piece loop (reference src/namgene.cc:575-676) and of the cut finder's exam windows over devices.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "capi_internal.h"

using namespace augx;

extern "C" {

// bin_of[i] = bin of item i.  Items sorted by decreasing length (ties: input order), each to the least loaded bin
// (ties: lowest index): deterministic.
int augx_partition_lpt(const int64_t *lens, int n, int n_bins, int32_t *bin_of) {
    if (!lens || !bin_of || n < 0 || n_bins < 1) { setLastError("augx_partition_lpt: bad argument"); return AUGX_E_ARG; }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    std::vector<int64_t> load(n_bins, 0);
    for (int i : order) {
        int best = 0;
        for (int b = 1; b < n_bins; b++)
            if (load[b] < load[best]) best = b;
        bin_of[i] = best;
        load[best] += lens[i] > 0 ? lens[i] : 0;
    }
    return AUGX_OK;
}

int augx_decode_sharded(augx_decoder *const *decs, int n_dec, const augx_piece *pieces, int n, augx_path *out) {
    if (!decs || n_dec < 1 || !pieces || n < 0 || !out) { setLastError("augx_decode_sharded: bad argument"); return AUGX_E_ARG; }
    for (int i = 0; i < n; i++) { out[i].states = nullptr; out[i].n_states = 0; out[i].status = AUGX_E_ARG; out[i].ln_viterbi = 0; }
    if (n == 0) return AUGX_OK;
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; i++) lens[i] = pieces[i].len;
    std::vector<int32_t> bin(n);
    int rc = augx_partition_lpt(lens.data(), n, n_dec, bin.data());
    if (rc) return rc;
    std::vector<int> rcs(n_dec, 0);
    std::vector<std::string> errs(n_dec);
    auto workBody = [&](int d) {
        // pieces of this device in input order, decoded in batches bounded by what the device's free memory holds
        std::vector<int> mine;
        for (int i = 0; i < n; i++)
            if (bin[i] == d) mine.push_back(i);
        if (mine.empty()) return;
        int64_t budget = augx_decoder_batch_capacity(decs[d]);
        if (const char *e = getenv("AUGX_BATCH_BASES")) budget = atol(e);
        size_t i = 0;
        while (i < mine.size()) {
            std::vector<augx_piece> batch;
            int64_t total = 0;
            size_t j = i;
            while (j < mine.size() && (batch.empty() || total + pieces[mine[j]].len <= budget)) {
                batch.push_back(pieces[mine[j]]);
                total += pieces[mine[j]].len;
                j++;
            }
            std::vector<augx_path> res(batch.size());
            int r = augx_decode_batch(decs[d], batch.data(), (int)batch.size(), res.data());
            if (r) { rcs[d] = r; errs[d] = augx_last_error(); return; }
            for (size_t k = 0; k < batch.size(); k++) out[mine[i + k]] = res[k];
            i = j;
        }
    };
    // (an exception on a worker thread -- allocation, thread creation -- becomes this device's error, not the end of the process)
    auto work = [&](int d) {
        try { workBody(d); }
        catch (const std::exception &e) { rcs[d] = AUGX_E_NOMEM; errs[d] = std::string("augx_decode_sharded: ") + e.what(); }
    };
    if (n_dec == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int d = 0; d < n_dec; d++) th.emplace_back(work, d);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < n_dec; d++)
        if (rcs[d]) {
            setLastError(errs[d]);
            for (int i = 0; i < n; i++) augx_path_free(&out[i]);
            return rcs[d];
        }
    return AUGX_OK;
}

// Decode + sample.  The draws come from ONE stream over the run, piece after piece in input order (the reference never seeds
// rand(), src/vitmatrix.cc:312), so the pieces are batched in input order, batch k on device k mod n_dec: the devices decode
// and run the forward algorithm of their batches concurrently, the host sampling of batch k waits for batch k-1's.
static int aheadPieces(int64_t bytesPerPiece) {
    int64_t avail = (int64_t)16 << 30; // (no /proc/meminfo: assume 16 GB to spare)
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            long long kb = 0;
            if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) { avail = (int64_t)kb << 10; break; }
        }
        fclose(f);
    }
    int64_t a = avail / 4 / (bytesPerPiece > 0 ? bytesPerPiece : 1);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && a > (int64_t)hw / 2) a = hw / 2;
    if (const char *e = getenv("AUGX_SAMPLE_AHEAD")) a = atol(e); // (developer aid)
    return (int)std::max<int64_t>(2, std::min<int64_t>(12, a));
}

int augx_decode_sampled(augx_decoder *const *decs, int n_dec, const augx_piece *pieces, int n, int n_samples, augx_rand *r,
                        augx_path *out, augx_path *samples) {
    if (!decs || n_dec < 1 || !pieces || n < 0 || n_samples < 0 || !r || !out || (n_samples && !samples)) {
        setLastError("augx_decode_sampled: bad argument");
        return AUGX_E_ARG;
    }
    for (int i = 0; i < n; i++) { out[i].states = nullptr; out[i].n_states = 0; out[i].status = AUGX_E_ARG; out[i].ln_viterbi = 0; }
    for (int64_t i = 0; i < (int64_t)n * n_samples; i++) { samples[i].states = nullptr; samples[i].n_states = 0; samples[i].status = AUGX_E_ARG; samples[i].ln_viterbi = 0; }
    if (n == 0) return AUGX_OK;
    // the forward matrix (S doubles per base) comes on top of what a decode needs
    int64_t budget = augx_decoder_sampled_capacity(decs[0]);
    for (int d = 1; d < n_dec; d++) budget = std::min<int64_t>(budget, augx_decoder_sampled_capacity(decs[d]));
    if (const char *e = getenv("AUGX_BATCH_BASES")) budget = atol(e);
    std::vector<std::pair<int, int>> batches; // [first, last)
    for (int i = 0; i < n;) {
        int j = i;
        int64_t total = 0;
        while (j < n && (j == i || total + pieces[j].len <= budget)) total += pieces[j++].len;
        batches.push_back({i, j});
        i = j;
    }
    std::mutex mu;
    std::condition_variable cv;
    int turn = 0;
    std::atomic<bool> failed{false}; // (read outside the lock by the workers)
    std::vector<int> rcs(n_dec, 0);
    std::vector<std::string> errs(n_dec);
    const bool timing = getenv("AUGX_TIMING") != nullptr; // (developer aid: the phases of every batch on stderr)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto workBody = [&](int d) {
        for (size_t k = (size_t)d; k < batches.size(); k += (size_t)n_dec) {
            const int first = batches[k].first, cnt = batches[k].second - batches[k].first;
            augx_batch *b = nullptr;
            const double t0 = timing ? now() : 0.0;
            int rc = augx_batch_create(decs[d], pieces + first, cnt, &b);
            if (!rc) rc = augx_batch_decode(decs[d], b);
            if (!rc) rc = augx_batch_paths(decs[d], b, out + first);
            const double t1 = timing ? now() : 0.0;
            if (!rc && n_samples) rc = augx_batch_forward(decs[d], b);
            if (!rc && timing) (void)augx_batch_sync(decs[d]);
            const double t2 = timing ? now() : 0.0;
            double tWaitTurn = 0, tWaitPrep = 0, tRun = 0, tFree = 0;
            if (rc) errs[d] = augx_last_error();
            // what the sampler reads of a piece is fetched and indexed on helper threads, a few pieces ahead of the sampling
            // (it does not depend on the draws), also while earlier batches are still being sampled
            // How many: a prepared piece holds its forward matrix and candidate records on the host (0.75 KB per base), and the draws
            // of a piece take tens of milliseconds where its preparation takes most of a second, and all but the forward matrix is copied while the forward kernel still runs -- up to 12 pieces (what they hold is touched once and then goes round: more of them is more page-table work at both ends), a quarter of
            // the memory the host has to spare and half of its cores
            int64_t longest = 1;
            for (int p = 0; p < cnt; p++) longest = std::max<int64_t>(longest, pieces[first + p].len);
            const int AHEAD = aheadPieces(longest * 768);
            struct Prep { int rc; augx_sample_prep *h; std::string err; }; // (augx_last_error is thread-local: the text travels with the result)
            std::vector<std::future<Prep>> prep((size_t)cnt);
            std::vector<std::future<void>> frees;
            auto launch = [&](int p) {
                if (p >= cnt || rc || !n_samples || out[first + p].status != AUGX_OK) return;
                auto job = [&, p]() {
                    augx_sample_prep *h = nullptr;
                    const int r2 = augx_batch_sample_prepare(decs[d], b, p, &h);
                    return Prep{r2, h, r2 ? std::string(augx_last_error()) : std::string()};
                };
                try {
                    prep[p] = std::async(std::launch::async, job);
                } catch (const std::system_error &) { // (no more threads to be had: fetched when its turn comes)
                    prep[p] = std::async(std::launch::deferred, job);
                }
            };
            for (int p = 0; p < AHEAD; p++) launch(p);
            {
                std::unique_lock<std::mutex> lk(mu);
                const double w0 = timing ? now() : 0.0;
                cv.wait(lk, [&] { return turn == (int)k || failed; });
                if (timing) tWaitTurn = now() - w0;
                for (int p = 0; p < cnt; p++) {
                    if (!prep[p].valid()) { launch(p + AHEAD); continue; } // (no path: nothing is drawn)
                    const double p0 = timing ? now() : 0.0;
                    Prep pr = prep[p].get();
                    const double p1 = timing ? now() : 0.0;
                    tWaitPrep += p1 - p0;
                    launch(p + AHEAD);
                    if (!rc && !failed) {
                        rc = pr.rc;
                        if (rc) errs[d] = pr.err;
                        if (!rc) {
                            rc = augx_sample_prep_run(pr.h, n_samples, r, samples + (int64_t)(first + p) * n_samples);
                            if (rc) errs[d] = augx_last_error();
                        }
                    }
                    const double p2 = timing ? now() : 0.0;
                    tRun += p2 - p1;
                    // (hundreds of MB per piece go back to the allocator: not on the thread every later piece waits for)
                    augx_sample_prep *hh = pr.h;
                    try { frees.push_back(std::async(std::launch::async, [hh] { augx_sample_prep_destroy(hh); })); }
                    catch (const std::system_error &) { augx_sample_prep_destroy(hh); }
                    if (timing) tFree += now() - p2;
                }
                if (rc) { rcs[d] = rc; failed = true; if (errs[d].empty()) errs[d] = "augx_decode_sampled: fetching a piece for the sampler failed"; }
                turn = (int)k + 1;
            }
            cv.notify_all();
            for (auto &f : frees) f.get();
            if (timing) {
                int64_t bases = 0;
                for (int p = 0; p < cnt; p++) bases += pieces[first + p].len;
                fprintf(stderr, "augx timing:   sampled batch on decoder %d: %d pieces, %lld bases: decode + paths %.3f s, forward %.3f s, wait for the batch before %.3f s, "
                                "wait for the sampler's inputs %.3f s, %d paths per piece drawn in %.3f s, inputs freed in %.3f s\n", d, cnt, (long long)bases, t1 - t0, t2 - t1, tWaitTurn, tWaitPrep, n_samples, tRun, tFree);
            }
            if (b) augx_batch_destroy(b);
            if (rc || failed) return;
        }
    };
    auto work = [&](int d) {
        try { workBody(d); }
        catch (const std::exception &e) {
            {
                std::unique_lock<std::mutex> lk(mu);
                rcs[d] = AUGX_E_NOMEM; errs[d] = std::string("augx_decode_sampled: ") + e.what();
                failed = true;
            }
            cv.notify_all();
        }
    };
    if (n_dec == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int d = 0; d < n_dec; d++) th.emplace_back(work, d);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < n_dec; d++)
        if (rcs[d]) {
            setLastError(errs[d]);
            for (int i = 0; i < n; i++) augx_path_free(&out[i]);
            for (int64_t i = 0; i < (int64_t)n * n_samples; i++) augx_path_free(&samples[i]);
            return rcs[d];
        }
    return AUGX_OK;
}
}
