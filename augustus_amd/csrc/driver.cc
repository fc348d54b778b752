// driver.cc -- whole-program driver: the `augustus` command line on top of the C ABI.
// Replaces main() / predictOnInputSequences (reference src/augustus.cc:94-248, 371-454) and
// NAMGene::doViterbiPiecewise / getNextCutEndPoint / tryFindCutEndPoint (src/namgene.cc:516-676, 973-1210)
// for the ab-initio path.  Host C++; every Viterbi decode (pieces AND cut-finding exam windows) goes through
// augx_decode_batch, i.e. runs on the GPU.
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <sstream>
#include <string>
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
const char *pieceSequence(const std::string &seq, long begin, long end, long limit, std::vector<std::string> &copies) {
    const long n = (long)seq.size();
    auto lower = [&](long i) { return seq[i] >= 'a' && seq[i] <= 'z'; };
    if (end + 1 >= n || !lower(end) || !lower(end + 1)) return seq.data() + begin;
    long e = end + 1;
    while (e + 1 < n && e <= limit && lower(e + 1)) e++;
    if (e <= limit) return seq.data() + begin; // the run ends within the limit: its feature is kept
    copies.emplace_back(seq, (size_t)begin, (size_t)(end - begin + 1));
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
    size_t pi = 0;
    int successful = 0;
    const bool noInFrameStop = M.opt.getBool("noInFrameStop", false);
    char buf[256];
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
        std::vector<std::pair<long, long>> runs, pieceRuns;
        bool haveRuns = false;
        for (; pi < pieces.size() && pieces[pi].rec == (int)r; pi++) {
            const PieceOut &pr = pieces[pi];
            if (pr.status != 0) {
                errmsg = pr.status == AUGX_E_UNSUPPORTED ? "piece outside what the MI355X path decodes"
                         : pr.status == AUGX_E_NOPATH    ? "No feasible path found in HMM"
                         : pr.status == AUGX_E_RANGE     ? "piece too improbable for the exact arithmetic of the MI355X path; lower --maxDNAPieceSize"
                                                         : "device decode failed (HIP error or kernel abort; not a property of the input)";
                continue;
            }
            std::vector<GeneOut> genes;
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
            } catch (std::exception &e) { errmsg = e.what(); continue; }
            for (GeneOut &g : genes) {
                g.seqname = rec.name;
                if (oo.uniqueGeneId) { snprintf(buf, sizeof buf, "%.30s.g%d", rec.name, geneid); g.id = buf; }
                else g.id = "g" + std::to_string(geneid);
                int tid = 1;
                for (Transcript &t : g.transcripts) {
                    t.shift(pr.begin);
                    t.seqname = rec.name;
                    t.id = "t" + std::to_string(tid++);
                    t.geneid = g.id;
                }
                geneid++;
                any = true;
            }
            if (oo.evidence && oo.softmasking && !genes.empty()) {
                // the hint groups of the evidence block: the soft-masked runs that END inside the piece (see pieceSequence)
                if (!haveRuns) { runs = lowerRuns(rec.seq, rec.len); haveRuns = true; }
                pieceRuns.clear();
                for (auto &ru : runs)
                    if (ru.second >= pr.begin && ru.second <= pr.end) pieceRuns.push_back(ru);
            }
            printGeneList(out, genes, rec.seq, rec.len, oo, &pieceRuns);
        }
        if (!errmsg.empty()) {
            if (successful < 1) { fatal = errmsg; return 1; }
            err += "\n augustus: ERROR\n\t" + errmsg + "\n\n";
        } else
            successful++;
        if (!any) out += "# (none)\n";
    }
    return 0;
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
        //      the input is read was tried: the decode that followed, on the main thread, took 1.6-2.4 s instead of 0.25 s.)
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

    // ---- phase 1: find the cut points of all records (reference src/namgene.cc:973-1133).  Inside a record the cuts are a
    //      serial chain (the next exam window starts where the last piece ended), but records are independent: every round
    //      decodes the pending exam window of EVERY unfinished record, fanned over the devices.
    struct PieceRef { int rec; long begin, end; int initKind, termKind; };
    struct CutState {
        long beginPos = 0;
        int prevInit = 0, prevTerm = 0; // init/term kinds in effect while the exam window is decoded (state leak, src/namgene.cc:576 vs 594-603)
        int attempt = 0;
        long examChunk = 0, es = 0, ee = 0;
        bool done = false;
        int failStatus = 0;             // an exam window of this record could not be decoded: the record's error
    };
    std::vector<std::vector<PieceRef>> recPieces(recs.size());
    std::vector<CutState> cs(recs.size());
    {
        auto pushPiece = [&](size_t r, long endPos) {
            const long seqlen = (long)recs[r].seq.size();
            CutState &c = cs[r];
            PieceRef pr;
            pr.rec = (int)r; pr.begin = c.beginPos; pr.end = endPos;
            pr.initKind = c.beginPos == 0 ? 0 : 1;
            pr.termKind = endPos == seqlen - 1 ? 0 : 1;
            recPieces[r].push_back(pr);
            c.prevInit = pr.initKind; c.prevTerm = pr.termKind;
            c.beginPos = endPos + 1;
            c.attempt = 0;
            if (c.beginPos >= seqlen) c.done = true;
        };
        for (;;) {
            std::vector<augx_piece> ex;
            std::vector<size_t> who;
            std::vector<std::string> copies;
            copies.reserve(recs.size());
            for (size_t r = 0; r < recs.size(); r++) {
                CutState &c = cs[r];
                const long seqlen = (long)recs[r].seq.size();
                while (!c.done && seqlen - c.beginPos <= maxstep) pushPiece(r, seqlen - 1); // the rest fits one piece
                if (c.done) continue;
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
                augx_piece p;
                p.seq = soft ? pieceSequence(recs[r].seq, c.es, c.ee, c.ee + 10000, copies) : recs[r].seq.data() + c.es;
                p.len = c.ee - c.es + 1; p.init_kind = c.prevInit; p.term_kind = c.prevTerm;
                ex.push_back(p);
                who.push_back(r);
            }
            if (ex.empty()) break;
            std::vector<Decoded> dd;
            if (!S.decode(ex, dd)) { restore(); return fail(S.err); }
            for (size_t k = 0; k < who.size(); k++) {
                const size_t r = who[k];
                CutState &c = cs[r];
                const long seqlen = (long)recs[r].seq.size();
                const long gapStart = 1, gapEnd = seqlen;
                if (dd[k].status != 0) { c.failStatus = dd[k].status; c.done = true; continue; } // the record's error (reported in input order below)
                long cut = tryFindCutEndPoint(dd[k].path, c.es, c.ee, true, gapStart, gapEnd, true);
                if (cut == -1 && c.attempt == 0) { c.attempt = 1; continue; } // once more with a window twice as long
                if (cut == -1) {
                    cut = tryFindCutEndPoint(dd[k].path, c.es, c.ee, true, gapStart, gapEnd, false);
                    if (cut == -1) cut = tryFindCutEndPoint(dd[k].path, c.es, c.ee, false, 0, 0, false);
                    if (cut == -1) cut = c.beginPos + maxstep - 1;
                }
                if (cut <= c.beginPos + 0.05 * maxstep || cut <= c.beginPos + 5000) cut = c.beginPos + maxstep - 1;
                pushPiece(r, cut);
            }
        }
    }
    lap("cut finder");
    std::vector<PieceRef> allPieces;
    for (size_t r = 0; r < recs.size(); r++)
        if (!cs[r].failStatus) allPieces.insert(allPieces.end(), recPieces[r].begin(), recPieces[r].end());
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
    {
        std::vector<augx_piece> ps;
        std::vector<size_t> slot; // ps[k] is decoded[slot[k]]
        std::vector<std::string> copies, rcs;
        copies.reserve(allPieces.size());
        rcs.reserve(allPieces.size());
        for (size_t i = 0; i < allPieces.size(); i++) {
            const PieceRef &pr = allPieces[i];
            augx_piece p;
            p.seq = soft ? pieceSequence(recs[pr.rec].seq, pr.begin, pr.end, pr.end, copies) : recs[pr.rec].seq.data() + pr.begin;
            p.len = pr.end - pr.begin + 1;
            p.init_kind = pr.initKind;
            p.term_kind = pr.termKind;
            if (runF) { ps.push_back(p); slot.push_back(i * per); }
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
        if (!ps.empty() && !(S.sampleiterations > 0 ? S.decodeSampled(ps, dd) : S.decode(ps, dd))) { restore(); return fail(S.err); }
        for (size_t k = 0; k < dd.size(); k++) decoded[slot[k]] = std::move(dd[k]);
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
    }
    for (size_t r = 0; r < recs.size(); r++)
        if (cs[r].failStatus) { PieceOut o{(int)r, 0, (long)recs[r].seq.size() - 1, &noPath, cs[r].failStatus}; po.push_back(o); }
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
