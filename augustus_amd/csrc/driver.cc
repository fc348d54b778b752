// driver.cc -- whole-program driver: the `augustus` command line on top of the C ABI.
// Replaces main() / predictOnInputSequences (reference src/augustus.cc:94-248, 371-454) and
// NAMGene::doViterbiPiecewise / getNextCutEndPoint / tryFindCutEndPoint (src/namgene.cc:516-676, 973-1210)
// for the ab-initio path.  Host C++; every Viterbi decode (pieces AND cut-finding exam windows) goes through
// augx_decode_batch, i.e. runs on the GPU.
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <tuple>
#include <sstream>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "capi_internal.h"
#include "genes.h"

using namespace augx;

namespace {

struct Record { std::string name, seq; };

// reference readOneFastaSeq / readFastaHeader, src/fasta.cc:132-182: a record starts at '>' (name = first word of the header;
// a sequence before any header is "unnamed-N"), its sequence is every alphabetic character of the lines that follow.
// The whole input is read at once and scanned line by line with memchr (a genome is hundreds of megabytes).
bool readFasta(std::istream &in, std::vector<Record> &recs) {
    std::string data;
    {   // a seekable input (a file) in one read of its size; a pipe through the stream buffer
        const std::streampos p0 = in.tellg();
        bool done = false;
        if (p0 != std::streampos(-1) && in.seekg(0, std::ios::end)) {
            const std::streampos p1 = in.tellg();
            in.seekg(p0);
            if (p1 != std::streampos(-1) && p1 > p0) {
                data.resize((size_t)(p1 - p0));
                in.read(&data[0], (std::streamsize)data.size());
                data.resize((size_t)in.gcount());
                done = true;
            }
        }
        in.clear();
        if (!done) {
            std::ostringstream ss;
            ss << in.rdbuf();
            data = ss.str();
        }
    }
    size_t i = 0;
    const size_t n = data.size();
    while (i < n && isspace((unsigned char)data[i])) i++;
    if (i >= n || data[i] != '>') return false;
    int unnamed = 1;
    while (i < n) {
        while (i < n && isspace((unsigned char)data[i])) i++;
        if (i >= n) break;
        Record r;
        if (data[i] == '>') {
            const char *nl = (const char *)memchr(data.data() + i, '\n', n - i);
            const size_t e = nl ? (size_t)(nl - data.data()) : n;
            size_t w = i + 1;
            while (w < e && !isspace((unsigned char)data[w])) w++;
            r.name.assign(data, i + 1, w - i - 1);
            i = e < n ? e + 1 : n;
        } else
            r.name = "unnamed-" + std::to_string(unnamed++);
        // the record's lines: up to the next line that begins with '>'
        size_t j = i;
        while (j < n && data[j] != '>') {
            const char *nl = (const char *)memchr(data.data() + j, '\n', n - j);
            j = nl ? (size_t)(nl - data.data()) + 1 : n;
        }
        r.seq.reserve(j - i);
        for (size_t a = i; a < j;) {
            const char *nl = (const char *)memchr(data.data() + a, '\n', j - a);
            const size_t e = nl ? (size_t)(nl - data.data()) : j;
            // (the usual line: letters only -- appended in one piece; the test is a branch-free count the compiler vectorises)
            size_t letters = 0;
            const unsigned char *q = (const unsigned char *)data.data();
            for (size_t k = a; k < e; k++) letters += (unsigned char)((q[k] | 0x20) - 'a') < 26;
            const bool clean = letters == e - a;
            if (clean) r.seq.append(data, a, e - a);
            else
                for (size_t k = a; k < e; k++)
                    if (isalpha((unsigned char)data[k])) r.seq.push_back(data[k]);
            a = e + 1;
        }
        i = j;
        if (!r.seq.empty()) recs.push_back(std::move(r));
    }
    return true;
}

struct Decoded { std::vector<PathState> path; double lnv; int status; std::vector<std::vector<PathState>> samples; };

struct Session {
    augx_model *model = nullptr;
    std::vector<augx_decoder *> decs; // one per device in use (AUGX_DEVICES; default: every visible device)
    OutputOptions oo;
    int geneid = 1;
    std::string err;
    // posterior sampling (--sample=n >= 10): n - 1 paths are sampled per piece next to the Viterbi path, the draws of the whole
    // run come from one generator (reference src/namgene.cc:824-871, src/vitmatrix.cc:312)
    int sampleiterations = 0;
    augx_rand *rng = nullptr;

    void destroy() {
        if (rng) augx_rand_destroy(rng);
        rng = nullptr;
        for (augx_decoder *d : decs) augx_decoder_destroy(d);
        decs.clear();
        if (model) augx_model_destroy(model);
        model = nullptr;
    }
    // decode a set of pieces on the GPUs: longest-first bin packing over the devices, one host thread per device, batches
    // bounded by each device's free memory (sharded.cc)
    bool decode(const std::vector<augx_piece> &pieces, std::vector<Decoded> &out) {
        std::vector<augx_path> paths(pieces.size());
        int rc = augx_decode_sharded(decs.data(), (int)decs.size(), pieces.data(), (int)pieces.size(), paths.data());
        if (rc) { err = augx_last_error(); return false; }
        out.resize(pieces.size());
        for (size_t i = 0; i < pieces.size(); i++) {
            out[i].lnv = paths[i].ln_viterbi;
            out[i].status = paths[i].status;
            out[i].path.clear();
            out[i].samples.clear();
            for (int k = 0; k < paths[i].n_states; k++)
                out[i].path.push_back({paths[i].states[k].begin, paths[i].states[k].end, paths[i].states[k].type});
            augx_path_free(&paths[i]);
        }
        return true;
    }
    // the same with sampleiterations - 1 sampled paths per piece (pieces in input order: the order of the draws)
    bool decodeSampled(const std::vector<augx_piece> &pieces, std::vector<Decoded> &out) {
        const int ns = sampleiterations - 1;
        if (!rng) rng = augx_rand_create(1);
        std::vector<augx_path> paths(pieces.size()), smp(pieces.size() * (size_t)ns);
        int rc = augx_decode_sampled(decs.data(), (int)decs.size(), pieces.data(), (int)pieces.size(), ns, rng, paths.data(), smp.data());
        if (rc) { err = augx_last_error(); return false; }
        out.resize(pieces.size());
        for (size_t i = 0; i < pieces.size(); i++) {
            out[i].lnv = paths[i].ln_viterbi;
            out[i].status = paths[i].status;
            out[i].path.clear();
            for (int k = 0; k < paths[i].n_states; k++)
                out[i].path.push_back({paths[i].states[k].begin, paths[i].states[k].end, paths[i].states[k].type});
            augx_path_free(&paths[i]);
            out[i].samples.assign((size_t)ns, {});
            for (int q = 0; q < ns; q++) {
                augx_path &sp = smp[i * (size_t)ns + q];
                if (out[i].status == 0 && sp.status != 0) out[i].status = sp.status;
                for (int k = 0; k < sp.n_states; k++) out[i].samples[q].push_back({sp.states[k].begin, sp.states[k].end, sp.states[k].type});
                augx_path_free(&sp);
            }
        }
        return true;
    }
};

// reference NAMGene::tryFindCutEndPoint, src/namgene.cc:1145-1210, with the single group gap [gapStart, gapEnd]
long tryFindCutEndPoint(const std::vector<PathState> &path, long examStart, long examEnd, bool useGap, long gapStart,
                        long gapEnd, bool onlyInternalIR) {
    if (!useGap) { gapStart = 0; gapEnd = 0x7fffffff; }
    long maxirbegin = -1, maxirend = -1;
    for (size_t i = 0; i < path.size(); i++) {
        if (path[i].type != 0) continue; // igenic
        long irbegin = examStart + path[i].begin, irend = examStart + path[i].end;
        long lgbegin = -1, lgend = -1;
        if (gapStart < irbegin && gapEnd <= irend && gapEnd >= irbegin && gapEnd - irbegin > lgend - lgbegin) { lgbegin = irbegin; lgend = gapEnd; }
        else if (gapStart < irbegin && gapEnd > irend && irend - irbegin > lgend - lgbegin) { lgbegin = irbegin; lgend = irend; }
        else if (gapStart > irbegin && gapEnd < irend && gapEnd - gapStart > lgend - lgbegin) { lgbegin = gapStart; lgend = gapEnd; }
        else if (gapStart >= irbegin && gapStart <= irend && gapEnd >= irend && irend - gapStart > lgend - lgbegin) { lgbegin = gapStart; lgend = irend; }
        bool internal = i != 0 && i + 1 != path.size();
        if (lgend - lgbegin > maxirend - maxirbegin && (internal || !onlyInternalIR || lgend - lgbegin > (examEnd - examStart) / 2)) {
            maxirbegin = lgbegin;
            maxirend = lgend;
        }
    }
    return maxirend - maxirbegin > 0 ? (maxirend + maxirbegin) / 2 : -1;
}


// soft-masked runs [first, last] of a record (reference SequenceFeatureCollection::prepare, src/extrinsicinfo.cc:1696-1724:
// one nonexonpart hint of source RM per lower-case run)
std::vector<std::pair<long, long>> lowerRuns(const char *seq, long n) {
    std::vector<std::pair<long, long>> runs;
    for (long i = 0; i < n;) {
        if (seq[i] >= 'a' && seq[i] <= 'z') {
            long e = i;
            while (e + 1 < n && seq[e + 1] >= 'a' && seq[e + 1] <= 'z') e++;
            runs.push_back({i, e});
            i = e + 1;
        } else
            i++;
    }
    return runs;
}

// The reference builds the soft-masking hints on the whole record and hands a piece [from, to] only the features that END
// inside [from, to] (SequenceFeatureCollection(sfc, from, to), src/extrinsicinfo.cc:147-160; pieces: src/namgene.cc:607,
// exam windows: to = examEnd + 10000, src/namgene.cc:1051).  A lower-case run that continues beyond `limit` therefore
// gives no bonus inside the piece: the piece is decoded from a copy whose trailing run is upper-cased.
// Returns the pointer to decode from (the record itself, or the copy appended to `copies`).
const char *pieceSequence(const char *seq, long n, long begin, long end, long limit, std::vector<std::string> &copies) {
    auto lower = [&](long i) { return seq[i] >= 'a' && seq[i] <= 'z'; };
    if (end + 1 >= n || !lower(end) || !lower(end + 1)) return seq + begin;
    long e = end + 1;
    while (e + 1 < n && e <= limit && lower(e + 1)) e++;
    if (e <= limit) return seq + begin; // the run ends within the limit: its feature is kept
    copies.emplace_back(seq + begin, (size_t)(end - begin + 1));
    std::string &c = copies.back();
    for (long i = (long)c.size() - 1; i >= 0 && c[i] >= 'a' && c[i] <= 'z'; i--) c[i] = (char)(c[i] - 'a' + 'A');
    return c.data();
}

struct RecordView { const char *name; const char *seq; long len; };
struct PieceOut {
    int rec; long begin, end; const std::vector<PathState> *path; int status; const std::vector<std::vector<PathState>> *samples = nullptr;
    // --singlestrand=true: the run on the piece as it is (path: nullptr when --strand=backward leaves it out) and the run on its
    // reverse complement (pathR; nullptr with --strand=forward)
    const std::vector<PathState> *pathR = nullptr; const std::vector<std::vector<PathState>> *samplesR = nullptr;
    bool single = false;
};

// ---- gene structures + GFF for all records, in input order; gene ids are global and sequential over the run (reference
//      NAMGene::doViterbiPiecewise, src/namgene.cc:526,626-650; block headers src/augustus.cc:395-398).  `pieces` may
//      arrive in any order (they were decoded on several devices): they are gathered by (record, begin) first.
//      Returns 0, or 1 when the very first record failed (the reference then aborts the run, src/augustus.cc:425-440).
int formatRecords(const Model &M, const OutputOptions &oo, const std::vector<RecordView> &recs, std::vector<PieceOut> pieces,
                  int verbosity, int &geneid, std::string &out, std::string &err, std::string &fatal, int sampleiterations = 0) {
    std::stable_sort(pieces.begin(), pieces.end(), [](const PieceOut &a, const PieceOut &b) { return a.rec != b.rec ? a.rec < b.rec : a.begin < b.begin; });
    int successful = 0;
    const bool noInFrameStop = M.opt.getBool("noInFrameStop", false);
    // The gene structures of a piece (with sampling: the posterior probabilities from 99 paths) and its GFF text depend on nothing
    // but the piece -- except the gene numbers, which count through the run.  Four steps: the genes of every piece, side by side
    // on host threads; the numbers, in order; the text of every piece, side by side; the blocks put together in order.
    const size_t np = pieces.size();
    std::vector<std::vector<GeneOut>> pieceGenes(np);
    std::vector<std::string> pieceErr(np), pieceText(np);
    std::vector<int> firstId(np, 0);
    auto overPieces = [&](const std::function<void(size_t)> &job) {
        unsigned hw = std::thread::hardware_concurrency();
        const size_t nth = std::min<size_t>(np, std::min<size_t>(32, hw > 3 ? hw / 2 : 1));
        if (nth <= 1) { for (size_t i = 0; i < np; i++) job(i); return; }
        std::atomic<size_t> nextPiece{0};
        auto body = [&]() { for (size_t i; (i = nextPiece.fetch_add(1)) < np;) job(i); };
        std::vector<std::thread> th;
        for (size_t w = 1; w < nth; w++) {
            try { th.emplace_back(body); } catch (const std::system_error &) { break; } // (no more threads to be had: fewer of them)
        }
        body();
        for (auto &t2 : th) t2.join();
    };
    overPieces([&](size_t i) {
        const PieceOut &pr = pieces[i];
        if (pr.status != 0) return;
        const RecordView &rec = recs[(size_t)pr.rec];
        std::vector<GeneOut> &genes = pieceGenes[i];
        const long plen = pr.end - pr.begin + 1;
        auto run = [&](const std::vector<PathState> &path, const std::vector<std::vector<PathState>> *smp, bool anyStrand, const char *runSeq) {
            return groupToGenes(M, sampleiterations > 0 && smp ? filterTranscripts(M, posteriorTranscripts(M, path, *smp, plen, sampleiterations), anyStrand, runSeq)
                                                               : filterTranscripts(M, projectOntoGeneSequence(M, path, plen), anyStrand, runSeq));
        };
        try {
            if (!pr.single) genes = run(*pr.path, pr.samples, false, rec.seq + pr.begin);
            else { // reference NAMGene::doViterbiPiecewise, src/namgene.cc:611-626: the genes of the forward run, then those of the
                   // run on the reverse complement mapped back (reverseGeneList); sorted by coding start
                if (pr.path) genes = run(*pr.path, pr.samples, true, rec.seq + pr.begin);
                if (pr.pathR) {
                    std::string rc; // (the sequence of the second run is read by --noInFrameStop=true only)
                    if (noInFrameStop) {
                        rc.assign(rec.seq + pr.begin, (size_t)plen);
                        std::reverse(rc.begin(), rc.end());
                        for (char &c : rc) {
                            const char l = (char)tolower((unsigned char)c);
                            c = l == 'a' ? 't' : l == 'c' ? 'g' : l == 'g' ? 'c' : l == 't' ? 'a' : 'n';
                        }
                    }
                    std::vector<GeneOut> rv = run(*pr.pathR, pr.samplesR, true, noInFrameStop ? rc.c_str() : nullptr);
                    reverseGenes(rv, plen - 1);
                    for (GeneOut &g : rv) genes.push_back(std::move(g));
                }
                std::stable_sort(genes.begin(), genes.end(), [](const GeneOut &a, const GeneOut &b) { return a.mincodstart < b.mincodstart; });
            }
        } catch (std::exception &e) { pieceErr[i] = e.what(); if (pieceErr[i].empty()) pieceErr[i] = "error"; genes.clear(); }
    });
    for (size_t i = 0; i < np; i++) { firstId[i] = geneid; geneid += (int)pieceGenes[i].size(); }
    // (the hint groups of the evidence block need the soft-masked runs of the record: made once per record, by whichever piece asks first)
    std::vector<std::vector<std::pair<long, long>>> recRuns(recs.size());
    std::unique_ptr<std::once_flag[]> recRunsOnce(new std::once_flag[recs.size() ? recs.size() : 1]);
    overPieces([&](size_t i) {
        const PieceOut &pr = pieces[i];
        std::vector<GeneOut> &genes = pieceGenes[i];
        if (pr.status != 0 || !pieceErr[i].empty()) return;
        const RecordView &rec = recs[(size_t)pr.rec];
        char buf[256];
        int id = firstId[i];
        for (GeneOut &g : genes) {
            g.seqname = rec.name;
            if (oo.uniqueGeneId) { snprintf(buf, sizeof buf, "%.30s.g%d", rec.name, id); g.id = buf; }
            else g.id = "g" + std::to_string(id);
            int tid = 1;
            for (Transcript &t : g.transcripts) {
                t.shift(pr.begin);
                t.seqname = rec.name;
                t.id = "t" + std::to_string(tid++);
                t.geneid = g.id;
            }
            id++;
        }
        std::vector<std::pair<long, long>> pieceRuns;
        if (oo.evidence && oo.softmasking && !genes.empty()) {
            // the hint groups of the evidence block: the soft-masked runs that END inside the piece (see pieceSequence)
            std::call_once(recRunsOnce[(size_t)pr.rec], [&] { recRuns[(size_t)pr.rec] = lowerRuns(rec.seq, rec.len); });
            for (auto &ru : recRuns[(size_t)pr.rec])
                if (ru.second >= pr.begin && ru.second <= pr.end) pieceRuns.push_back(ru);
        }
        printGeneList(pieceText[i], genes, rec.seq, rec.len, oo, &pieceRuns);
    });
    size_t pi = 0;
    for (size_t r = 0; r < recs.size(); r++) {
        const RecordView &rec = recs[r];
        if (verbosity) {
            out += "#\n# ----- prediction on sequence number " + std::to_string(r + 1) + " (length = " + std::to_string(rec.len) + ", name = " + rec.name + ") -----\n#\n";
        }
        {
            const std::string st = M.opt.get("strand", "both");
            const bool fw = st == "forward", bw = st == "backward"; // (other values fall back to both, see genes.cc)
            out += "# Predicted genes for sequence number " + std::to_string(r + 1) + " on " + (fw ? "forward strand" : bw ? "reverse strand" : "both strands") + "\n";
            if (M.opt.getBool("singlestrand", false)) out += "# Overlapping genes on opposite strand are allowed.\n";
        }
        bool any = false;
        std::string errmsg;
        for (; pi < np && pieces[pi].rec == (int)r; pi++) {
            const PieceOut &pr = pieces[pi];
            if (!errmsg.empty()) continue; // (the reference leaves the record at its first error: nothing of the later pieces is printed)
            if (pr.status != 0) {
                errmsg = pr.status == AUGX_E_UNSUPPORTED ? "piece outside what the MI355X path decodes"
                         : pr.status == AUGX_E_NOPATH    ? "No feasible path found in HMM"
                         : pr.status == AUGX_E_RANGE     ? "piece too improbable for the exact arithmetic of the MI355X path; lower --maxDNAPieceSize"
                                                         : "device decode failed (HIP error or kernel abort; not a property of the input)";
                continue;
            }
            if (!pieceErr[pi].empty()) { errmsg = pieceErr[pi]; continue; }
            any = any || !pieceGenes[pi].empty();
            out += pieceText[pi];
        }
        if (!errmsg.empty()) {
            if (successful < 1) { fatal = errmsg; return 1; }
            err += "\n augustus: ERROR\n\t" + errmsg + "\n\n";
        } else
            successful++;
        if (!any && errmsg.empty()) out += "# (none)\n"; // (NAMGene::doViterbiPiecewise says so at its end, which an error does not reach, src/namgene.cc:672-673)
    }
    return 0;
}

// ---- the cut finder: where the records are cut into pieces (reference NAMGene::getNextCutEndPoint, src/namgene.cc:973-1133, as
//      called by the piece loop of doViterbiPiecewise, :575-603).  Inside a record the cuts are a serial chain: the exam window
//      of round k+1 starts where the piece of round k ended.  Every cut that is accepted comes from the decode of exactly the
//      window the reference decodes (same bases, same initial / terminal kinds) -- but the windows are decoded AHEAD of the
//      chain, many per batch: a scout decode of the long records in overlapping tiles tells where the intergenic regions lie,
//      the chain is run on that map as a FORECAST -- with the alternatives the map cannot decide: a gene cut by a window's edge
//      may vanish from the window's path or shrink to a variant that fits -- every window any forecast asks for is decoded in
//      one batch (windows are independent of each other), and the true chain then walks through the decoded windows until it
//      needs one that nobody foresaw.  A cut is the centre of an intergenic region, so the alternatives meet again after a round
//      or two and their number stays small.  The forecast decides only WHICH windows are decoded early, never a result.
struct PieceRef { int rec; long begin, end; int initKind, termKind; };
struct CutFinderStats { double scoutSeconds = 0; int tiles = 0, batches = 0, windows = 0, used = 0; };
// (scout = true: the paths only feed the cut finder's FORECAST -- which windows to decode early, never a cut --, so the decode may
//  skip what makes a multi-class piece exactly the reference's: the replay of the snippet cache and the second trellis run)
using DecodeFn = std::function<bool(const std::vector<augx_piece> &, std::vector<Decoded> &, bool scout)>;

bool findCutPoints(const Model &M, const std::vector<RecordView> &recs, long maxstep, bool soft, int scoutMode /* -1: decide here */, int nDevices,
                   const DecodeFn &decode, std::vector<std::vector<PieceRef>> &recPieces, std::vector<int> &failStatus, CutFinderStats &stats,
                   std::vector<std::vector<std::array<long, 3>>> *events = nullptr /* per record, in the reference's order: {1 piece | 0 exam window, begin, end} */) {
    if (events) events->assign(recs.size(), {});
    struct CutState {
        long beginPos = 0;
        int prevInit = 0, prevTerm = 0; // init/term kinds in effect while the exam window is decoded (state leak, src/namgene.cc:576 vs 594-603)
        int attempt = 0;
        long examChunk = 0, es = 0, ee = 0;
        bool done = false;
    };
    struct WinKey {
        int rec; long es, ee; int ik, tk;
        bool operator<(const WinKey &o) const { return std::tie(rec, es, ee, ik, tk) < std::tie(o.rec, o.es, o.ee, o.ik, o.tk); }
    };
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    recPieces.assign(recs.size(), {});
    failStatus.assign(recs.size(), 0);
    std::vector<CutState> cs(recs.size());
    // one step of the chain of record r on state c: the pieces it can emit without a window, then the window it needs next
    // (false: the record is finished)
    auto emitPiece = [&](size_t r, CutState &c, long endPos, std::vector<PieceRef> *emit) {
        const long seqlen = recs[r].len;
        PieceRef pr;
        pr.rec = (int)r; pr.begin = c.beginPos; pr.end = endPos;
        pr.initKind = c.beginPos == 0 ? 0 : 1;
        pr.termKind = endPos == seqlen - 1 ? 0 : 1;
        if (emit) emit->push_back(pr);
        c.prevInit = pr.initKind; c.prevTerm = pr.termKind;
        c.beginPos = endPos + 1;
        c.attempt = 0;
        if (c.beginPos >= seqlen) c.done = true;
    };
    auto nextWindow = [&](size_t r, CutState &c, std::vector<PieceRef> *emit) -> bool {
        const long seqlen = recs[r].len;
        while (!c.done && seqlen - c.beginPos <= maxstep) emitPiece(r, c, seqlen - 1, emit); // the rest fits one piece
        if (c.done) return false;
        if (c.attempt == 0) {
            c.examChunk = 50000;
            if (c.examChunk < 0.2 * maxstep) c.examChunk = (long)(0.2 * maxstep);
            if (c.examChunk > 150000) c.examChunk = 150000;
        } else { c.examChunk *= 2; if (c.examChunk > maxstep) c.examChunk = maxstep; }
        const long gapStart = 1, gapEnd = seqlen;
        const long center = (gapEnd - gapStart < c.examChunk) ? (gapEnd + gapStart) / 2 : gapEnd - c.examChunk / 2;
        if (c.attempt == 0 && c.examChunk > maxstep) { c.es = c.beginPos; c.ee = c.beginPos + maxstep - 1; }
        else {
            c.es = center - c.examChunk / 2;
            c.ee = center + c.examChunk / 2;
            if (c.ee >= c.beginPos + maxstep) { c.es -= (c.ee - (c.beginPos + maxstep - 1)); c.ee = c.beginPos + maxstep - 1; }
            if (c.es < c.beginPos) { c.ee += c.beginPos - c.es; c.es = c.beginPos; }
        }
        return true;
    };
    // what the chain does with the path of its window (attempt = 1 afterwards: the same round once more with a window twice as long)
    auto applyWindow = [&](size_t r, CutState &c, const std::vector<PathState> &path, std::vector<PieceRef> *emit) {
        const long gapStart = 1, gapEnd = recs[r].len;
        long cut = tryFindCutEndPoint(path, c.es, c.ee, true, gapStart, gapEnd, true);
        if (cut == -1 && c.attempt == 0) { c.attempt = 1; return; }
        if (cut == -1) {
            cut = tryFindCutEndPoint(path, c.es, c.ee, true, gapStart, gapEnd, false);
            if (cut == -1) cut = tryFindCutEndPoint(path, c.es, c.ee, false, 0, 0, false);
            if (cut == -1) cut = c.beginPos + maxstep - 1;
        }
        if (cut <= c.beginPos + 0.05 * maxstep || cut <= c.beginPos + 5000) cut = c.beginPos + maxstep - 1;
        emitPiece(r, c, cut, emit);
    };

    // ---- the scout: intergenic regions of the long records, from a decode in overlapping tiles.  Worth its time when the
    //      chains are long and few (one chromosome: 115 serial rounds on one compute unit each, against one pass over the
    //      record at the speed of the whole chip); not when many records' chains already run side by side.
    std::vector<std::vector<std::pair<long, long>>> igenic(recs.size());
    std::vector<char> scouted(recs.size(), 0);
    {
        long longBases = 0, maxRounds = 0;
        int nLong = 0;
        for (size_t r = 0; r < recs.size(); r++)
            if (recs[r].len > maxstep) { longBases += recs[r].len; nLong++; maxRounds = std::max(maxRounds, recs[r].len / maxstep); }
        long chunk = std::min<long>(150000, std::max<long>(50000, (long)(0.2 * maxstep)));
        if (chunk > maxstep) chunk = maxstep;
        int nIgenic = 0;
        for (int q = 0; q < M.t.S; q++) nIgenic += M.t.state_kind[q] == AUGX_K_IGENIC;
        const bool denseModel = M.t.utr != 0 || nIgenic > 1; // (the models of device/dense.h: one workgroup per piece)
        // serial: rounds x 1.5 windows (every other first try fails on gene-poor DNA) at the speed of one workgroup;
        // scout: the long records once at the speed of the chip
        const double perPiece = denseModel ? 0.36e6 : 3.5e6, chip = (denseModel ? 60e6 : 300e6) * (double)std::max(1, nDevices);
        const double tSerial = (double)maxRounds * 1.5 * (double)chunk / perPiece, tScout = (double)longBases / chip + 0.02;
        bool want = maxRounds >= 3 && tScout < 0.5 * tSerial;
        if (scoutMode >= 0) want = scoutMode != 0;
        if (want && nLong > 0) {
            const double t0 = now();
            const long margin = 25000;
            long tile = longBases / (256 * (long)std::max(1, nDevices));
            tile = std::max<long>(200000, std::min<long>(1000000, tile));
            std::vector<augx_piece> tp;
            struct TileRef { size_t rec; long b, e; };
            std::vector<TileRef> tr;
            for (size_t r = 0; r < recs.size(); r++) {
                const long n = recs[r].len;
                if (n <= maxstep) continue;
                for (long b = 0; b < n; b += tile) {
                    const long tb = std::max<long>(0, b - margin), te = std::min<long>(n - 1, b + tile - 1 + margin);
                    augx_piece p;
                    p.seq = recs[r].seq + tb; p.len = te - tb + 1; p.init_kind = tb == 0 ? 0 : 1; p.term_kind = te == n - 1 ? 0 : 1;
                    tp.push_back(p);
                    tr.push_back({r, tb, te});
                }
                scouted[r] = 1;
            }
            std::vector<Decoded> dd;
            if (!decode(tp, dd, true)) return false;
            for (size_t k = 0; k < tr.size(); k++) {
                if (dd[k].status != 0) continue; // (no forecast from this tile; the chain decodes its windows as it goes)
                const long n = recs[tr[k].rec].len;
                // the part of the record this tile speaks for: its own stretch without the margins
                const long lo = tr[k].b == 0 ? 0 : tr[k].b + margin, hi = tr[k].e == n - 1 ? n - 1 : tr[k].e - margin;
                auto &v = igenic[tr[k].rec];
                for (const PathState &st : dd[k].path) {
                    if (st.type != 0) continue;
                    const long b = std::max(lo, tr[k].b + st.begin), e = std::min(hi, tr[k].b + st.end);
                    if (b > e) continue;
                    if (!v.empty() && v.back().second + 1 >= b) v.back().second = std::max(v.back().second, e);
                    else v.push_back({b, e});
                }
            }
            stats.tiles = (int)tp.size();
            stats.scoutSeconds = now() - t0;
        }
    }
    // the paths a window [es, ee] of record r is EXPECTED to have, as far as the cut finder looks at them: intergenic runs
    // (type 0) and whatever lies between them (type 1).  A window begins (ends) in the intergenic state when its initial
    // (terminal) kind is the synchronisation state alone, and a gene the map shows across that edge cannot be in the window's
    // path: either it vanishes -- the run at the edge reaches to the next intergenic region of the map -- or a variant of it
    // that fits takes its place and the map's next intergenic region stays a run of its own.  Both, at both edges: up to four paths.
    auto forecasts = [&](size_t r, long es, long ee, int ik, int tk) {
        const auto &v = igenic[r];
        std::vector<std::pair<long, long>> runs;
        for (auto it = std::partition_point(v.begin(), v.end(), [&](const std::pair<long, long> &a) { return a.second < es; }); it != v.end() && it->first <= ee; ++it)
            runs.push_back({std::max(es, it->first), std::min(ee, it->second)});
        const bool openL = ik == 1 && (runs.empty() || runs.front().first > es), openR = tk == 1 && (runs.empty() || runs.back().second < ee);
        std::vector<std::vector<PathState>> out;
        // (the order is by how often each was right on uniform-random DNA: a variant of the gene at both edges first)
        for (int vl = openL ? 1 : 0; vl >= 0; vl--)
            for (int vr = openR ? 1 : 0; vr >= 0; vr--) {
                std::vector<std::pair<long, long>> ru = runs;
                if (ru.empty()) { // the window lies inside one gene of the map: all intergenic, or a variant of the gene between two edge runs
                    if (openL && openR && vl == 0 && vr == 0) ru.push_back({es, ee});
                    else { if (ik == 1) ru.push_back({es, es}); if (tk == 1) ru.push_back({ee, ee}); }
                } else {
                    if (openL) { if (vl == 0) ru.front().first = es; else ru.insert(ru.begin(), {es, es}); }
                    if (openR) { if (vr == 0) ru.back().second = ee; else ru.push_back({ee, ee}); }
                }
                std::vector<PathState> path;
                long pos = es;
                for (auto &x : ru) {
                    if (x.first < pos) continue; // (a window inside one gene of the map: the two edge runs)
                    if (x.first > pos) path.push_back({pos - es, x.first - 1 - es, 1});
                    path.push_back({x.first - es, x.second - es, 0});
                    pos = x.second + 1;
                }
                if (pos <= ee) path.push_back({pos - es, ee - es, 1});
                if (std::find_if(out.begin(), out.end(), [&](const std::vector<PathState> &o) {
                        return o.size() == path.size() && std::equal(o.begin(), o.end(), path.begin(), [](const PathState &a, const PathState &b) { return a.begin == b.begin && a.end == b.end && a.type == b.type; });
                    }) == out.end())
                    out.push_back(std::move(path));
            }
        return out;
    };

    std::map<WinKey, Decoded> cache;
    const bool cutDebug = getenv("AUGX_CUT_DEBUG") != nullptr;
    std::vector<std::string> cutDebugLines;
    // guesses followed per undecoded window.  Measured on 23 Mbp of uniform-random DNA, fly model: 47 states (a batch costs what its
    // bases cost): 1 -> 8 batches of ~55 windows, 5 -> 5 batches of ~215, slower.  71 states (one workgroup per window, a batch costs
    // what its longest window costs while there are compute units to spare): 1 -> 26 batches, 5 -> 18
    int nIgenicStates = 0;
    for (int q = 0; q < M.t.S; q++) nIgenicStates += M.t.state_kind[q] == AUGX_K_IGENIC;
    int breadth = (M.t.utr != 0 || nIgenicStates > 1) ? 5 : 1;
    if (const char *e = getenv("AUGX_CUT_BREADTH")) breadth = atoi(e);
    long nearShift = 3000;
    if (const char *e = getenv("AUGX_CUT_NEAR")) nearShift = atol(e);
    size_t maxAsk = 256 * (size_t)std::max(1, nDevices); // windows per batch: one wave of workgroups
    if (const char *e = getenv("AUGX_CUT_ASK")) maxAsk = (size_t)atol(e);
    // (round 6, measured and dropped: the long "retry" windows first in the batch and up to 2 (256 - L) short ones in their shadow --
    //  15 batches of 380 windows instead of 18 of 256, but 0.45 s instead of 0.30 s per batch: 6.8 against 5.3 s; profiles/EXPERIMENTS.md)
    // windows per batch (a bound on the forecast's breadth; the nearest rounds come first)
    for (;;) {
        std::vector<WinKey> keys;
        std::set<WinKey> asked;
        // the true chains first, as far as the decoded windows carry them ...
        for (size_t r = 0; r < recs.size(); r++) {
            CutState &c = cs[r];
            if (c.done) continue;
            for (;;) {
                CutState probe = c;
                std::vector<PieceRef> emitted;
                const bool more = nextWindow(r, probe, &emitted);
                auto hit = more ? cache.find(WinKey{(int)r, probe.es, probe.ee, probe.prevInit, probe.prevTerm}) : cache.end();
                if (more && hit == cache.end()) break;
                c = probe;
                recPieces[r].insert(recPieces[r].end(), emitted.begin(), emitted.end());
                if (events) for (const PieceRef &pe : emitted) (*events)[r].push_back({1, pe.begin, pe.end});
                if (!more) break;
                stats.used++;
                if (events) (*events)[r].push_back({0, c.es, c.ee});
                if (hit->second.status != 0) { failStatus[r] = hit->second.status; c.done = true; break; } // the record's error
                if (cutDebug && scouted[r]) { // developer aid: what the map's forecasts said about this window against what its decode says
                    CutState tr2 = c;
                    std::vector<PieceRef> tmp;
                    applyWindow(r, tr2, hit->second.path, &tmp);
                    std::string line = "F " + std::to_string(r) + " " + std::to_string(c.es) + " " + std::to_string(c.ee) + " " + std::to_string(c.attempt) + " true " + (tr2.attempt == 1 && c.attempt == 0 ? std::string("retry") : std::to_string(tr2.beginPos - 1));
                    for (auto &path : forecasts(r, c.es, c.ee, c.prevInit, c.prevTerm)) {
                        CutState g = c;
                        applyWindow(r, g, path, &tmp);
                        line += " " + (g.attempt == 1 && c.attempt == 0 ? std::string("retry") : std::to_string(g.beginPos - 1));
                    }
                    cutDebugLines.push_back(line);
                }
                const size_t before = recPieces[r].size();
                applyWindow(r, c, hit->second.path, &recPieces[r]);
                if (events) for (size_t q = before; q < recPieces[r].size(); q++) (*events)[r].push_back({1, recPieces[r][q].begin, recPieces[r][q].end});
            }
        }
        // ... then the batch's room is shared among the records that still wait for a window (short records finished above and
        // do not count; every open record gets at least one window, however many there are: a chain must always advance)
        size_t nOpen = 0;
        for (size_t r = 0; r < recs.size(); r++) nOpen += !cs[r].done;
        for (size_t r = 0; r < recs.size(); r++) {
            CutState &c = cs[r];
            if (c.done) continue;
            // the forecast chains from here: every window they ask for that has not been decoded yet.  Limited-discrepancy order:
            // first the chain of first guesses to the end of the record, then the chains that leave it once (nearest round first),
            // twice, ... while the batch has room -- a wrong guess at the nearest undecoded round is what ends a batch's use
            struct Sim { int disc, depth; CutState st; };
            auto later = [](const Sim &a, const Sim &b) { return a.disc != b.disc ? a.disc > b.disc : a.depth > b.depth; };
            std::priority_queue<Sim, std::vector<Sim>, decltype(later)> queue(later);
            queue.push({0, 0, c});
            std::set<std::tuple<long, int, int, int>> seen{{c.beginPos, c.attempt, c.prevInit, c.prevTerm}};
            const size_t room = keys.size() + std::max<size_t>(1, maxAsk / std::max<size_t>(1, nOpen));
            while (!queue.empty() && keys.size() < room) {
                Sim cur = queue.top();
                queue.pop();
                CutState sim = cur.st;
                if (!nextWindow(r, sim, nullptr)) continue;
                const WinKey k{(int)r, sim.es, sim.ee, sim.prevInit, sim.prevTerm};
                auto push = [&](const std::vector<PathState> &path, int extra) {
                    CutState nx = sim;
                    applyWindow(r, nx, path, nullptr);
                    if (!nx.done && seen.insert({nx.beginPos, nx.attempt, nx.prevInit, nx.prevTerm}).second) queue.push({cur.disc + extra, cur.depth + 1, nx});
                };
                auto hit = cache.find(k);
                if (hit != cache.end()) {
                    if (hit->second.status == 0) push(hit->second.path, 0);
                    continue;
                }
                if (asked.insert(k).second) keys.push_back(k);
                if (!scouted[r]) break; // no map: one window per round, as the reference goes
                int nv = 0;
                // a decoded window of the same size a few hundred bases away is the best guess there is (the chain left the forecast by
                // that much a round ago): its path, as far as it covers this window, with intergenic edges
                {
                    auto lo = cache.lower_bound(WinKey{(int)r, sim.es - nearShift, 0, 0, 0});
                    const Decoded *best = nullptr;
                    long bestShift = nearShift + 1, bestEs = 0;
                    for (auto it = lo; it != cache.end() && it->first.rec == (int)r && it->first.es <= sim.es + nearShift; ++it)
                        if (it->first.ee - it->first.es == sim.ee - sim.es && it->first.ik == sim.prevInit && it->first.tk == sim.prevTerm && it->second.status == 0 &&
                            std::labs(it->first.es - sim.es) < bestShift) { best = &it->second; bestShift = std::labs(it->first.es - sim.es); bestEs = it->first.es; }
                    if (best) {
                        std::vector<PathState> path;
                        for (const PathState &st : best->path) {
                            const long b2 = std::max(sim.es, bestEs + st.begin), e2 = std::min(sim.ee, bestEs + st.end);
                            if (b2 > e2) continue;
                            if (path.empty() && b2 > sim.es) path.push_back({0, b2 - 1 - sim.es, 0});
                            if (!path.empty() && path.back().type == 0 && st.type == 0) path.back().end = e2 - sim.es;
                            else path.push_back({b2 - sim.es, e2 - sim.es, st.type});
                        }
                        if (!path.empty() && path.back().end < sim.ee - sim.es) {
                            if (path.back().type == 0) path.back().end = sim.ee - sim.es; else path.push_back({path.back().end + 1, sim.ee - sim.es, 0});
                        }
                        if (!path.empty()) push(path, nv++ ? 1 : 0);
                    }
                }
                for (auto &path : forecasts(r, sim.es, sim.ee, sim.prevInit, sim.prevTerm)) { if (nv >= breadth) break; push(path, nv++ ? 1 : 0); }
            }
        }
        if (keys.empty()) {
            for (size_t r = 0; r < recs.size(); r++)
                if (!cs[r].done) { setLastError("cut finder: no exam window could be asked for record " + std::to_string(r) + " (internal error)"); return false; }
            break;
        }
        std::vector<augx_piece> ex;
        std::vector<std::string> copies;
        copies.reserve(keys.size());
        for (const WinKey &k : keys) {
            augx_piece p;
            p.seq = soft ? pieceSequence(recs[k.rec].seq, recs[k.rec].len, k.es, k.ee, k.ee + 10000, copies) : recs[k.rec].seq + k.es;
            p.len = k.ee - k.es + 1; p.init_kind = k.ik; p.term_kind = k.tk;
            ex.push_back(p);
        }
        std::vector<Decoded> dd;
        if (!decode(ex, dd, false)) return false;
        stats.batches++;
        stats.windows += (int)ex.size();
        for (size_t k = 0; k < keys.size(); k++) cache[keys[k]] = std::move(dd[k]);
    }
    if (const char *dbg = getenv("AUGX_CUT_DEBUG")) { // developer aid: the scout's map, every decoded window and the pieces, for work on the forecast
        if (FILE *f = fopen(dbg, "w")) {
            for (size_t r = 0; r < recs.size(); r++)
                for (auto &x : igenic[r]) fprintf(f, "M %zu %ld %ld\n", r, x.first, x.second);
            for (auto &kv : cache) {
                fprintf(f, "W %d %ld %ld %d %d %d %zu", kv.first.rec, kv.first.es, kv.first.ee, kv.first.ik, kv.first.tk, kv.second.status, kv.second.path.size());
                for (auto &st : kv.second.path) fprintf(f, " %ld %ld %d", st.begin, st.end, st.type);
                fprintf(f, "\n");
            }
            for (size_t r = 0; r < recs.size(); r++)
                for (auto &pr : recPieces[r]) fprintf(f, "C %zu %ld %ld\n", r, pr.begin, pr.end);
            for (auto &l : cutDebugLines) fprintf(f, "%s\n", l.c_str());
            fclose(f);
        }
    }
    return true;
}

} // namespace

extern "C" int augx_main(int argc, const char *const *argv) {
    Session S;
    // AUGX_TIMING=1: wall-clock of the phases on stderr (developer aid; stderr is otherwise empty on success)
    const bool timing = getenv("AUGX_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tPrev = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "augx timing: %-28s %8.3f s\n", what, t - tPrev); tPrev = t; } };
    std::string commandline;
    for (int i = 0; i < argc; i++) { commandline += argv[i]; if (i < argc - 1) commandline += " "; }
    auto fail = [&](const std::string &msg) {
        std::cerr << "\n" << (argc > 0 ? argv[0] : "augustus") << ": ERROR\n\t" << msg << "\n\n";
        S.destroy();
        return 1;
    };
    if (argc <= 1) {
        std::cout << "AUGUSTUS-MI355X (ab-initio GHMM Viterbi decode on gfx950; drop-in for AUGUSTUS 3.5.0 ab-initio prediction)\n\n"
                  << "usage:\naugustus [parameters] --species=SPECIES queryfilename\n";
        return 0;
    }
    // ---- command line (reference Properties::init, src/properties.cc:66-135)
    std::vector<std::pair<std::string, std::string>> cmd;
    std::vector<std::pair<std::string, bool>> plain; // every non-special --name in the reference's order of checking (right to left), with "has '='"
    std::string queryfile, species, configPath;
    for (int a = argc - 1; a >= 1; a--) {
        std::string s(argv[a]);
        if (s == "--version") { std::cerr << augx_version() << "\n"; return 0; }  // (reference: HelpException -> message on stderr, exit 0)
        if (s == "--help") { std::cerr << "usage:\naugustus [parameters] --species=SPECIES queryfilename\n"; return 0; }
        if (s.size() >= 2 && s.compare(0, 2, "--") == 0) {
            s.erase(0, 2);
            size_t pos = s.find('=');
            std::string name = s.substr(0, pos);
            // the parameters the reference takes from the command line in its first pass (src/properties.cc:93-108)
            const bool special = name == "genemodel" || name == "nc" || name == "singlestrand" || name == "species" || name == "extrinsicCfgFile" ||
                                 name == "AUGUSTUS_CONFIG_PATH" || name == "alnfile" || name == "treefile" || name == "dbaccess" ||
                                 name == "speciesfilenames" || name == "codonAlignmentFile" || name == "referenceFile";
            if (special && (pos == std::string::npos || pos >= s.size() - 1))
                return fail("Wrong argument format for " + name + ". Use: --argument=value");
            std::string value = pos == std::string::npos ? "" : s.substr(pos + 1);
            if (name == "species") species = value;
            else if (name == "AUGUSTUS_CONFIG_PATH") configPath = value;
            else {
                if (!special) plain.push_back({name, pos != std::string::npos});
                if (pos != std::string::npos) cmd.insert(cmd.begin(), {name, value});
            }
        } else if (queryfile.empty())
            queryfile = s;
        else
            return fail("Error: 2 query files given: " + queryfile + " and " + s + ".\nparameter names must start with '--'");
    }
    // config directory: command line > environment > relative to the executable, <dir of the binary>/../config
    // (reference src/properties.cc:116-135, findLocationOfSelfBinary :648-684)
    if (configPath.empty()) {
        const char *e = getenv("AUGUSTUS_CONFIG_PATH");
        if (e) configPath = e;
        else {
            char self[4096];
            ssize_t k = readlink("/proc/self/exe", self, sizeof self - 1);
            if (k <= 0) return fail("/proc/self/exe not found.\nPlease specify environment variable or parameter AUGUSTUS_CONFIG_PATH.");
            configPath.assign(self, (size_t)k);
            size_t pos = configPath.find_last_of('/');
            if (pos != std::string::npos && pos > 0) pos = configPath.find_last_of('/', pos - 1);
            if (pos != std::string::npos) configPath.resize(pos);
            configPath += "/config";
        }
    }
    if (configPath.back() != '/') configPath += '/';
    struct stat sb;
    if (stat(configPath.c_str(), &sb) == -1 || !S_ISDIR(sb.st_mode))
        return fail(configPath + " is not a directory. Could not locate directory AUGUSTUS_CONFIG_PATH.");
    if (species.empty()) return fail("No species specified. Type \"augustus --species=help\" to see available species.");
    if (species == "help") { // reference: HelpException(SPECIES_LIST) -> message on stderr, exit code 0 (src/augustus.cc:240-242)
        std::vector<std::string> names;
        if (DIR *d = opendir((configPath + "species").c_str())) {
            while (struct dirent *e = readdir(d)) {
                std::string nm = e->d_name;
                struct stat s2;
                if (nm[0] != '.' && stat((configPath + "species/" + nm + "/" + nm + "_parameters.cfg").c_str(), &s2) == 0) names.push_back(nm);
            }
            closedir(d);
        }
        std::sort(names.begin(), names.end());
        std::cerr << "usage:\naugustus [parameters] --species=SPECIES queryfilename\n\nwhere SPECIES is one of the following identifiers (parameter sets found in "
                  << configPath << "species)\n\n";
        for (auto &nm : names) std::cerr << nm << "\n";
        return 0;
    }
    {   // the species' parameter file must exist (reference Properties::readFile, src/properties.cc:35-60, prints its search on stderr)
        const std::string rel = "species/" + species + "/" + species + "_parameters.cfg";
        struct stat s2;
        if (stat((configPath + rel).c_str(), &s2) != 0) {
            std::cerr << "Could not find the config file " << configPath << rel << "." << std::endl;
            std::cerr << " Looking for " << species << "_parameters.cfg in the configuration directory instead ...";
            if (stat((configPath + species + "_parameters.cfg").c_str(), &s2) != 0) std::cerr << " not";
            std::cerr << " found." << std::endl;
        }
    }
    {   // a parameter without '=' and unknown parameters are errors, checked right to left (reference src/properties.cc:275-319,
        // names from config/parameters/aug_cmdln_parameters.json) -- after the species' files have been found
        struct stat s2;
        const bool haveSpecies = stat((configPath + "species/" + species + "/" + species + "_parameters.cfg").c_str(), &s2) == 0 ||
                                 stat((configPath + species + "_parameters.cfg").c_str(), &s2) == 0;
        std::ifstream pj((configPath + "parameters/aug_cmdln_parameters.json").c_str());
        if (pj && haveSpecies) {
            std::stringstream ss;
            ss << pj.rdbuf();
            const std::string js = ss.str();
            for (auto &kv : plain) {
                if (!kv.second) return fail("'=' missing for parameter: " + kv.first);
                if (js.find("\"name\": \"" + kv.first + "\"") == std::string::npos && js.find("\"name\":\"" + kv.first + "\"") == std::string::npos)
                    return fail("Unknown parameter: \"" + kv.first + "\". Type \"augustus\" for help.");
            }
        }
    }
    std::vector<const char *> names, values;
    for (auto &kv : cmd) { names.push_back(kv.first.c_str()); values.push_back(kv.second.c_str()); }
    int rc = augx_model_load(configPath.c_str(), species.c_str(), (int)cmd.size(), names.data(), values.data(), &S.model);
    if (rc) return fail(augx_last_error());
    lap("model load");
    const Model &M = S.model->m;
    S.oo.fromModel(M);
    if (queryfile.empty()) return fail("No query file specified. Type \"augustus\" for help.");
    if (queryfile != "-") { // (reference: the input file is opened before the header is printed)
        std::ifstream probe(queryfile.c_str());
        if (!probe) return fail("Could not open input file \"" + queryfile + "\"!");
    }
    { // reference NAMGene::NAMGene, src/namgene.cc:53-67
        int si = M.opt.getInt("sample", 0);
        if (si > 0 && si < 10) {
            std::cerr << "Error: Number of sample iterations is too low. (sample=" << si << ")" << std::endl
                      << "I will not sample (sample=0) and will not estimate posterior probabilities." << std::endl;
            si = 0;
        } else if (si >= 10 && si < 20)
            std::cerr << "Warning: Number of sample iterations 'sample'=" << si << " is low." << std::endl
                      << "Posterior probabilities will be only rough estimates." << std::endl;
        S.sampleiterations = si > 0 ? si : 0;
    }
    // redirect output if requested (reference src/augustus.cc:503-520)
    std::ofstream outfile, errfile;
    std::streambuf *coutbuf = std::cout.rdbuf(), *cerrbuf = std::cerr.rdbuf();
    if (M.opt.has("outfile")) { outfile.open(M.opt.get("outfile").c_str()); if (outfile) std::cout.rdbuf(outfile.rdbuf()); }
    if (M.opt.has("errfile")) { errfile.open(M.opt.get("errfile").c_str()); if (errfile) std::cerr.rdbuf(errfile.rdbuf()); }
    auto restore = [&]() { std::cout.flush(); std::cerr.flush(); std::cout.rdbuf(coutbuf); std::cerr.rdbuf(cerrbuf); };

    const int verbosity = M.opt.getInt("/augustus/verbosity", 1);
    // (what the reference says while it sets its parameters comes first: a genetic code other than the standard one, on the
    //  output stream; start codons of the species that the code does not know, on the error stream -- src/augustus.cc:526-528,
    //  src/geneticcode.cc:157-165,297-299)
    if (!M.codeWarnings.empty()) std::cout << M.codeWarnings;
    if (!M.stderrNotes.empty()) std::cerr << M.stderrNotes;
    if (S.oo.gff3) std::cout << "##gff-version 3" << std::endl;
    std::cout << "# This output was generated with AUGUSTUS-MI355X (GHMM Viterbi decode on gfx950; output format of AUGUSTUS 3.5.0).\n"
              << "# AUGUSTUS is a gene prediction tool written by M. Stanke (mario.stanke@uni-greifswald.de),\n"
              << "# O. Keller, S. K\xc3\xb6nig, L. Gerischer, L. Romoth, Katharina Hoff, Henry Mehlan and Daniel Honsel.\n"
              << "# Please cite: Mario Stanke, Mark Diekhans, Robert Baertsch, David Haussler (2008),\n"
              << "# Using native and syntenically mapped cDNA alignments to improve de novo gene finding\n"
              << "# Bioinformatics 24: 637-644, doi 10.1093/bioinformatics/btn013" << std::endl;
    if (verbosity) std::cout << "# No extrinsic information on sequences given." << std::endl;
    if (verbosity > 1) std::cout << "# Initializing the parameters using config directory " << configPath << " ..." << std::endl;
    std::cout << "# " << species << " version.";
    if (M.speciesSpecificTrans) std::cout << " Using species specific transition matrix: " << M.transFileUsed;
    else std::cout << " Using default transition matrix.";
    std::cout << std::endl;

    std::vector<Record> recs;
    {
        bool ok;
        if (queryfile == "-") ok = readFasta(std::cin, recs);
        else {
            std::ifstream in(queryfile.c_str());
            if (!in) { restore(); return fail("Could not open input file \"" + queryfile + "\"!"); }
            ok = readFasta(in, recs);
        }
        if (!ok) { restore(); return fail("File format of " + queryfile + " not recognized (only FASTA input is supported on the MI355X path)."); }
    }
    lap("FASTA read");
    if (verbosity > 2) {
        if (queryfile == "-") std::cout << "# Reading sequences from standard input. Assuming fasta format." << std::endl;
        else std::cout << "# Looks like " << queryfile << " is in fasta format." << std::endl;
    }
    if (verbosity > 0) std::cout << "# We have hints for 0 sequences and for 0 of the sequences in the input set." << std::endl;

    const long maxstep = M.opt.getInt("maxDNAPieceSize", 1000000);
    if (maxstep < 1000) { std::cerr << "maxDNAPieceSize is too small: " << maxstep << std::endl; restore(); S.destroy(); return 1; }
    {   // ---- devices: every visible GPU, or the ones named by AUGX_DEVICES ("0,2,5"; a single number N = the first N);
        //      AUGX_DEVICE (one index) is kept for single-device runs.  (Bringing the decoders up on a thread of their own while
        //      the input is read was tried: the decode that followed, on the main thread, took 1.6-2.4 s instead of 0.25 s.  Round 5
        //      tried it the other way round -- the input read on a helper thread, the decoders on this one: the same, 2.5 s, three
        //      batches instead of one.  Whatever a second thread touches first, the uploads that follow crawl; the two stay in sequence.)
        std::vector<int> devs;
        const int ndev = augx_device_count();
        if (const char *e = getenv("AUGX_DEVICES")) {
            std::string v(e);
            if (v.find(',') == std::string::npos) { int k = atoi(e); for (int i = 0; i < k; i++) devs.push_back(i); }
            else { std::stringstream ss(v); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) devs.push_back(atoi(tok.c_str())); }
        } else if (const char *e1 = getenv("AUGX_DEVICE"))
            devs.push_back(atoi(e1));
        else
            for (int i = 0; i < ndev; i++) devs.push_back(i);
        if (devs.empty()) devs.push_back(0); // (no device: augx_decoder_create reports it -- there is no CPU decode path)
        for (int dv : devs) {
            augx_decoder *d = nullptr;
            rc = augx_decoder_create(S.model, dv, &d);
            if (rc) { restore(); return fail(augx_last_error()); }
            S.decs.push_back(d);
        }
        // (a device named several times -- AUGX_DEVICES=0,0,...: the eight host threads of a node on one GPU -- is shared by that many
        //  decoders: each plans its trellis segments for its share of the compute units; the memory share is the decoder's own
        //  business, augx_decoder_batch_capacity)
        for (size_t i = 0; i < devs.size(); i++) {
            const int cnt = (int)std::count(devs.begin(), devs.end(), devs[i]);
            if (cnt > 1) (void)augx_decoder_set_share(S.decs[i], cnt);
        }
    }

    lap("device / decoder create");

    // --predictionStart / --predictionEnd: predict on a piece of the first sequence only and shift the printed coordinates
    // (reference cutRelevantPiece, src/augustus.cc:552-602)
    if ((M.opt.has("predictionStart") || M.opt.has("predictionEnd")) && !recs.empty()) {
        const long seqlen = (long)recs[0].seq.size();
        long ps = M.opt.has("predictionStart") ? M.opt.getInt("predictionStart", 1) - 1 : 0;
        long pe = M.opt.has("predictionEnd") ? M.opt.getInt("predictionEnd", 1) - 1 : seqlen - 1;
        if ((ps != 0 || pe != seqlen - 1) && !(pe < 0 && ps < 0)) {
            if (ps < 0) ps = 0;
            if (pe > seqlen - 1) pe = seqlen - 1;
            if (ps >= seqlen) {
                restore();
                return fail("predictionStart (" + std::to_string(ps + 1) + ") is larger than sequence length (" + std::to_string(seqlen) + "). No predictions made.");
            }
            if (pe < ps) {
                restore();
                return fail("predictionEnd (" + std::to_string(pe + 1) + ") is smaller than predictionStart (" + std::to_string(ps + 1) + "). No predictions made!");
            }
            if (recs.size() > 1) {
                std::cerr << "Warning: predictionStart or predictionEnd set but input consists of more than one sequence." << std::endl
                          << "Prediction will be made only on first sequence." << std::endl;
                recs.resize(1);
            }
            recs[0].seq = recs[0].seq.substr((size_t)ps, (size_t)(pe - ps + 1));
            S.oo.offset = ps;
        } else if (ps < 0 && pe < 0 && pe == ps)
            S.oo.offset = -ps - 1;
    }
    const bool soft = S.oo.softmasking;

    // ---- phase 1: find the cut points of all records (findCutPoints above; every exam window is decoded on the GPUs)
    std::vector<std::vector<PieceRef>> recPieces;
    std::vector<int> recFail;
    std::vector<std::vector<std::array<long, 3>>> decodeEvents; // per record: the sequences the reference decodes, in its order
    {
        std::vector<RecordView> views;
        for (auto &r : recs) views.push_back({r.name.c_str(), r.seq.data(), (long)r.seq.size()});
        int scoutMode = -1;
        if (const char *e = getenv("AUGX_SCOUT")) scoutMode = atoi(e) != 0;
        CutFinderStats st;
        std::string cfErr;
        auto decodeFn = [&](const std::vector<augx_piece> &ps, std::vector<Decoded> &dd, bool scout) {
            std::vector<int> was;
            if (scout) for (augx_decoder *d : S.decs) { was.push_back(augx_decoder_exact(d)); augx_decoder_set_exact(d, 0); }
            const bool ok = S.decode(ps, dd);
            if (scout) for (size_t i = 0; i < S.decs.size(); i++) augx_decoder_set_exact(S.decs[i], was[i]);
            if (ok) return true;
            cfErr = S.err;
            return false;
        };
        if (!findCutPoints(M, views, maxstep, soft, scoutMode, (int)S.decs.size(), decodeFn, recPieces, recFail, st, &decodeEvents)) { restore(); return fail(cfErr); }
        if (timing)
            fprintf(stderr, "augx timing:   cut finder: scout %.3f s (%d tiles), %d batches, %d windows decoded, %d used\n", st.scoutSeconds, st.tiles, st.batches, st.windows,
                    st.used);
    }
    lap("cut finder");
    std::vector<PieceRef> allPieces;
    // (a record whose chain of cuts stopped at an exam window without a feasible path: the pieces before that window are decoded and
    //  printed, then comes the error -- the reference prints a piece's genes before it looks for the next cut, src/namgene.cc:575-655)
    for (size_t r = 0; r < recs.size(); r++) allPieces.insert(allPieces.end(), recPieces[r].begin(), recPieces[r].end());
    if (M.opt.getBool("progress", false))
        for (auto &pr : allPieces)
            std::cerr << "examining piece " << pr.begin + S.oo.offset + 1 << ".." << pr.end + S.oo.offset + 1 << " (" << (pr.end - pr.begin + 1) << " bp)" << std::endl;

    // ---- phase 2: decode all pieces, sharded over the devices (longest first), each device in batches bounded by its memory
    // (--singlestrand=true: the model has no shadow states; every piece is decoded as it is and as its reverse complement --
    //  --strand picks the runs -- in this order, which is also the order of the draws when sampling)
    const bool single = M.opt.getBool("singlestrand", false);
    const std::string strandOpt = M.opt.get("strand", "both");
    const bool runF = !single || strandOpt != "backward", runR = single && strandOpt != "forward";
    const size_t per = single ? 2 : 1;
    std::vector<Decoded> decoded(allPieces.size() * per);
    // UTR states: the TSS window that begins at base 0 of a piece is answered from entry 0 of the reference's tssProbsPlus /
    // tssProbsMinus, which lives on while the sequences it decodes -- exam windows and pieces, record after record -- keep ONE length
    // (include/augx.h: augx_tss0; src/utrmodel.cc:744-747,779-781): a piece that follows such a sequence gets the value of the FIRST
    // sequence of that run of equal lengths.  (An exam window that reads a stale value is decoded with its own: its path only
    // places a cut.  --singlestrand=true runs every piece twice, in another order: left alone.)
    std::map<std::pair<int, long>, std::pair<int, long>> tss0Src; // (record, begin of the piece) -> (record, begin) of the sequence whose value it reads
    if (M.t.utr && !single) {
        long cacheSize = -1, srcBegin = 0;
        int srcRec = -1;
        for (size_t r = 0; r < recs.size(); r++)
            for (const auto &ev : decodeEvents[r]) {
                const long L = ev[2] - ev[1] + 1;
                if (L + 1 != cacheSize) { cacheSize = L + 1; srcRec = -1; }
                if (srcRec < 0) { srcRec = (int)r; srcBegin = ev[1]; }
                else if (ev[0] == 1 && !(srcRec == (int)r && srcBegin == ev[1])) tss0Src[{(int)r, ev[1]}] = {srcRec, srcBegin};
            }
    }
    std::vector<const char *> tss0Set;
    {
        std::vector<augx_piece> ps;
        std::vector<size_t> slot; // ps[k] is decoded[slot[k]]
        std::vector<std::string> copies, rcs;
        copies.reserve(allPieces.size());
        rcs.reserve(allPieces.size());
        for (size_t i = 0; i < allPieces.size(); i++) {
            const PieceRef &pr = allPieces[i];
            augx_piece p;
            p.seq = soft ? pieceSequence(recs[pr.rec].seq.data(), (long)recs[pr.rec].seq.size(), pr.begin, pr.end, pr.end, copies) : recs[pr.rec].seq.data() + pr.begin;
            p.len = pr.end - pr.begin + 1;
            p.init_kind = pr.initKind;
            p.term_kind = pr.termKind;
            if (runF) { ps.push_back(p); slot.push_back(i * per); }
            if (!tss0Src.empty()) {
                auto it = tss0Src.find({pr.rec, pr.begin});
                if (it != tss0Src.end()) {
                    const long L = pr.end - pr.begin + 1; // (the source has the piece's length: that is what makes it the source)
                    double v[2];
                    if (augx_tss0(S.model, recs[(size_t)it->second.first].seq.data() + it->second.second, L, v) == AUGX_OK && augx_tss0_override(p.seq, v) == AUGX_OK)
                        tss0Set.push_back(p.seq);
                }
            }
            if (runR) { // reverse complement, case kept (the soft-masked runs are the hints of this run, too); the initial and
                        // terminal probabilities are the piece's, not swapped (src/namgene.cc:594-603 precede both runs)
                rcs.emplace_back((size_t)p.len, 'n');
                std::string &rc = rcs.back();
                for (int64_t q = 0; q < p.len; q++) {
                    const char c = p.seq[p.len - 1 - q];
                    char d;
                    switch (c) {
                    case 'a': d = 't'; break; case 'c': d = 'g'; break; case 'g': d = 'c'; break; case 't': d = 'a'; break;
                    case 'A': d = 'T'; break; case 'C': d = 'G'; break; case 'G': d = 'C'; break; case 'T': d = 'A'; break;
                    default: d = c;
                    }
                    rc[(size_t)q] = d;
                }
                augx_piece q2 = p;
                q2.seq = rc.data();
                ps.push_back(q2);
                slot.push_back(i * per + 1);
            }
        }
        std::vector<Decoded> dd;
        const bool okDec = ps.empty() || (S.sampleiterations > 0 ? S.decodeSampled(ps, dd) : S.decode(ps, dd));
        for (const char *q : tss0Set) (void)augx_tss0_override(q, nullptr);
        if (!okDec) { restore(); return fail(S.err); }
        for (size_t k = 0; k < dd.size(); k++) decoded[slot[k]] = std::move(dd[k]);
    }

    if (timing && getenv("AUGX_NEAR_TIES") && atoi(getenv("AUGX_NEAR_TIES")) != 0) { // (counted by a kernel build of its own: asked for by name, not implied by AUGX_TIMING)
        int64_t nt = 0, np = 0;
        for (augx_decoder *d : S.decs) { int64_t q = 0; nt += augx_decoder_near_ties(d, &q); np += q; }
        fprintf(stderr, "augx timing:   near ties on the chosen paths (exam windows and pieces): %lld cells in %lld decodes\n", (long long)nt, (long long)np);
    }
    lap("decode of the pieces");
    // ---- phase 3: gene structures + GFF, in input order
    std::vector<RecordView> rv;
    for (auto &r : recs) rv.push_back({r.name.c_str(), r.seq.data(), (long)r.seq.size()});
    std::vector<PieceOut> po;
    static const std::vector<PathState> noPath;
    for (size_t i = 0; i < allPieces.size(); i++) {
        PieceOut o{allPieces[i].rec, allPieces[i].begin, allPieces[i].end, &decoded[i * per].path, decoded[i * per].status,
                   S.sampleiterations > 0 ? &decoded[i * per].samples : nullptr};
        if (single) {
            o.single = true;
            if (!runF) { o.path = nullptr; o.samples = nullptr; o.status = 0; }
            if (runR) {
                o.pathR = &decoded[i * per + 1].path;
                o.samplesR = S.sampleiterations > 0 ? &decoded[i * per + 1].samples : nullptr;
                if (o.status == 0) o.status = decoded[i * per + 1].status;
            }
        }
        po.push_back(o);
        // (the entry of the exam window that failed follows the record's pieces: formatRecords walks the entries record by record)
        const int r = allPieces[i].rec;
        if (recFail[(size_t)r] && (i + 1 == allPieces.size() || allPieces[i + 1].rec != r)) po.push_back(PieceOut{r, 0, (long)recs[(size_t)r].seq.size() - 1, &noPath, recFail[(size_t)r]});
    }
    {   // (failed records without a single piece)
        std::vector<char> has(recs.size(), 0);
        for (auto &pr : allPieces) has[(size_t)pr.rec] = 1;
        std::vector<PieceOut> merged;
        size_t k = 0;
        for (size_t r = 0; r < recs.size(); r++) {
            while (k < po.size() && po[k].rec == (int)r) merged.push_back(po[k++]);
            if (recFail[r] && !has[r]) merged.push_back(PieceOut{(int)r, 0, (long)recs[r].seq.size() - 1, &noPath, recFail[r]});
        }
        po.swap(merged);
    }
    std::string text, errText, fatal;
    if (formatRecords(M, S.oo, rv, po, verbosity, S.geneid, text, errText, fatal, S.sampleiterations)) {
        std::cout << text;
        std::cerr << errText;
        restore();
        return fail(fatal);
    }
    std::cout << text;
    std::cerr << errText;
    std::cout << "# command line:" << std::endl << "# " << commandline << std::endl;
    restore();
    lap("genes + GFF");
    S.destroy();
    lap("teardown");
    return 0;
}

// ---- test/diagnostic hook: format externally supplied state paths exactly like augx_main would (one record, one
//      piece).  Lets the CPU test-suite pin the gene-structure + GFF stage against the reference without a GPU.
extern "C" int augx_format_gff(const augx_model *m, const char *name, const char *seq, int64_t len, const augx_state *states,
                               int n_states, int first_gene_id, char *out, int64_t out_cap, int *n_genes) {
    if (!m || !name || !seq || !out) return AUGX_E_ARG;
    try {
        OutputOptions oo;
        oo.fromModel(m->m);
        std::vector<PathState> path;
        for (int i = 0; i < n_states; i++) path.push_back({states[i].begin, states[i].end, states[i].type});
        std::vector<GeneOut> genes = groupToGenes(m->m, filterTranscripts(m->m, projectOntoGeneSequence(m->m, path, (long)len), false, seq));
        int gid = first_gene_id;
        for (GeneOut &g : genes) {
            g.seqname = name;
            g.id = "g" + std::to_string(gid++);
            int tid = 1;
            for (Transcript &t : g.transcripts) { t.seqname = name; t.id = "t" + std::to_string(tid++); t.geneid = g.id; }
        }
        std::string text;
        printGeneList(text, genes, seq, (long)len, oo, nullptr);
        if (n_genes) *n_genes = (int)genes.size();
        if ((int64_t)text.size() + 1 > out_cap) { setLastError("augx_format_gff: output buffer too small"); return AUGX_E_ARG; }
        memcpy(out, text.c_str(), text.size() + 1);
        return AUGX_OK;
    } catch (std::exception &e) {
        setLastError(e.what());
        return AUGX_E_CONFIG;
    }
}

// ---- the same with sampled paths (n_samples of them, e.g. from augx_batch_sample): posterior probabilities in the score columns;
//      sampleiterations = n_samples + 1 as in the reference's --sample
extern "C" int augx_format_gff_sampled(const augx_model *m, const char *name, const char *seq, int64_t len, const augx_state *states,
                                       int n_states, int n_samples, const augx_state *const *sample_states, const int *sample_n,
                                       int first_gene_id, char *out, int64_t out_cap, int *n_genes) {
    if (!m || !name || !seq || !out || n_samples < 0 || (n_samples && (!sample_states || !sample_n))) return AUGX_E_ARG;
    try {
        OutputOptions oo;
        oo.fromModel(m->m);
        std::vector<PathState> path;
        for (int i = 0; i < n_states; i++) path.push_back({states[i].begin, states[i].end, states[i].type});
        std::vector<std::vector<PathState>> smp((size_t)n_samples);
        for (int q = 0; q < n_samples; q++)
            for (int i = 0; i < sample_n[q]; i++) smp[q].push_back({sample_states[q][i].begin, sample_states[q][i].end, sample_states[q][i].type});
        std::vector<GeneOut> genes = groupToGenes(m->m, filterTranscripts(m->m, posteriorTranscripts(m->m, path, smp, (long)len, n_samples + 1), false, seq));
        int gid = first_gene_id;
        for (GeneOut &g : genes) {
            g.seqname = name;
            g.id = "g" + std::to_string(gid++);
            int tid = 1;
            for (Transcript &t : g.transcripts) { t.seqname = name; t.id = "t" + std::to_string(tid++); t.geneid = g.id; }
        }
        std::string text;
        printGeneList(text, genes, seq, (long)len, oo, nullptr);
        if (n_genes) *n_genes = (int)genes.size();
        if ((int64_t)text.size() + 1 > out_cap) { setLastError("augx_format_gff_sampled: output buffer too small"); return AUGX_E_ARG; }
        memcpy(out, text.c_str(), text.size() + 1);
        return AUGX_OK;
    } catch (std::exception &e) {
        setLastError(e.what());
        return AUGX_E_CONFIG;
    }
}

// ---- the ordered gather of a sharded run on its own: pieces of several records, decoded anywhere and handed over in any
//      order, become the prediction part of the `augustus` output with global gene numbering in input order.
extern "C" int augx_format_records(const augx_model *m, int n_records, const char *const *names, const char *const *seqs,
                                   const int64_t *lens, int n_pieces, const augx_piece_result *pieces, char *out, int64_t out_cap) {
    if (!m || n_records < 0 || n_pieces < 0 || !out || (n_records && (!names || !seqs || !lens)) || (n_pieces && !pieces)) {
        setLastError("augx_format_records: bad argument");
        return AUGX_E_ARG;
    }
    try {
        OutputOptions oo;
        oo.fromModel(m->m);
        std::vector<RecordView> rv;
        for (int r = 0; r < n_records; r++) rv.push_back({names[r], seqs[r], (long)lens[r]});
        std::vector<std::vector<PathState>> paths(n_pieces);
        std::vector<PieceOut> po;
        for (int i = 0; i < n_pieces; i++) {
            if (pieces[i].record < 0 || pieces[i].record >= n_records) { setLastError("augx_format_records: bad record index"); return AUGX_E_ARG; }
            for (int k = 0; k < pieces[i].n_states; k++) paths[i].push_back({pieces[i].states[k].begin, pieces[i].states[k].end, pieces[i].states[k].type});
            po.push_back({pieces[i].record, (long)pieces[i].begin, (long)pieces[i].end, &paths[i], pieces[i].status});
        }
        int geneid = 1;
        std::string text, errText, fatal;
        int rc = formatRecords(m->m, oo, rv, po, 1, geneid, text, errText, fatal);
        if (rc) { setLastError(fatal); return AUGX_E_NOPATH; }
        if ((int64_t)text.size() + 1 > out_cap) { setLastError("augx_format_records: output buffer too small"); return AUGX_E_ARG; }
        memcpy(out, text.c_str(), text.size() + 1);
        return AUGX_OK;
    } catch (std::exception &e) {
        setLastError(e.what());
        return AUGX_E_CONFIG;
    }
}

// ---- the cut finder on its own, over any decode function (reference NAMGene::getNextCutEndPoint as driven by the piece loop of
//      doViterbiPiecewise, src/namgene.cc:575-603, 973-1133).  augx_main runs it over augx_decode_sharded; a caller that owns the
//      piece loop can run it over its own decoders.  scout: -1 = decide from the chain length, 0 = one window
//      per round as the reference goes, 1 = scout decode + windows decoded ahead of the chain (same cuts, fewer batches).
extern "C" int augx_find_cuts(const augx_model *m, int n_records, const char *const *seqs, const int64_t *lens, augx_decode_fn fn, void *user, int scout,
                              augx_cut **out, int *n_out, augx_cut_stats *stats) {
    if (!m || n_records < 0 || (n_records && (!seqs || !lens)) || !fn || !out || !n_out) { setLastError("augx_find_cuts: bad argument"); return AUGX_E_ARG; }
    *out = nullptr;
    *n_out = 0;
    try {
        const Model &M = m->m;
        const long maxstep = M.opt.getInt("maxDNAPieceSize", 1000000);
        if (maxstep < 1000) { setLastError("maxDNAPieceSize is too small"); return AUGX_E_CONFIG; }
        OutputOptions oo;
        oo.fromModel(M);
        std::vector<RecordView> views;
        for (int r = 0; r < n_records; r++) views.push_back({"", seqs[r], (long)lens[r]});
        int rcDecode = 0;
        auto decodeFn = [&](const std::vector<augx_piece> &ps, std::vector<Decoded> &dd, bool /*scout: the caller's decode function decides*/) {
            std::vector<augx_path> paths(ps.size());
            for (auto &p : paths) { p.states = nullptr; p.n_states = 0; p.status = AUGX_E_ARG; p.ln_viterbi = 0; }
            rcDecode = fn(user, ps.data(), (int)ps.size(), paths.data());
            if (rcDecode) return false;
            dd.resize(ps.size());
            for (size_t i = 0; i < ps.size(); i++) {
                dd[i].lnv = paths[i].ln_viterbi;
                dd[i].status = paths[i].status;
                dd[i].path.clear();
                for (int k = 0; k < paths[i].n_states; k++) dd[i].path.push_back({paths[i].states[k].begin, paths[i].states[k].end, paths[i].states[k].type});
                augx_path_free(&paths[i]);
            }
            return true;
        };
        std::vector<std::vector<PieceRef>> recPieces;
        std::vector<int> recFail;
        CutFinderStats st;
        if (!findCutPoints(M, views, maxstep, oo.softmasking, scout, 1, decodeFn, recPieces, recFail, st)) return rcDecode ? rcDecode : AUGX_E_HIP;
        // (a record whose exam window could not be decoded: its pieces so far, then one entry with that status and no range)
        for (int r = 0; r < n_records; r++)
            if (recFail[r]) recPieces[r].push_back(PieceRef{r, -1, -1, -recFail[r], 0});
        size_t n = 0;
        for (auto &v : recPieces) n += v.size();
        augx_cut *res = (augx_cut *)malloc(sizeof(augx_cut) * (n ? n : 1));
        if (!res) return AUGX_E_NOMEM;
        size_t k = 0;
        for (int r = 0; r < n_records; r++)
            for (auto &pr : recPieces[r]) res[k++] = pr.begin < 0 ? augx_cut{r, -pr.initKind, -1, -1, 0, 0} : augx_cut{r, 0, pr.begin, pr.end, pr.initKind, pr.termKind};
        *out = res;
        *n_out = (int)n;
        if (stats) { stats->scout_tiles = st.tiles; stats->batches = st.batches; stats->windows_decoded = st.windows; stats->windows_used = st.used; }
        return AUGX_OK;
    } catch (std::exception &e) {
        setLastError(e.what());
        return AUGX_E_CONFIG;
    }
}
extern "C" void augx_cuts_free(augx_cut *c) { free(c); }
