// driver.cc -- whole-program driver: the `augustus` command line on top of the C ABI.
// Replaces main() / predictOnInputSequences (reference src/augustus.cc:94-248, 371-454) and
// NAMGene::doViterbiPiecewise / getNextCutEndPoint / tryFindCutEndPoint (src/namgene.cc:516-676, 973-1210)
// for the ab-initio path.  Host C++; every Viterbi decode (pieces AND cut-finding exam windows) goes through
// augx_decode_batch, i.e. runs on the GPU.
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "capi_internal.h"
#include "genes.h"

using namespace augx;

namespace {

struct Record { std::string name, seq; };

// reference readOneFastaSeq / readFastaHeader, src/fasta.cc:132-182
bool readFasta(std::istream &in, std::vector<Record> &recs) {
    std::string line;
    int unnamed = 1;
    in >> std::ws;
    if (!in || in.peek() != '>') return false;
    while (in) {
        in >> std::ws;
        if (!in) break;
        Record r;
        if (in.peek() == '>') {
            std::getline(in, line);
            size_t e = 1;
            while (e < line.size() && !isspace((unsigned char)line[e])) e++;
            r.name = line.substr(1, e - 1);
        } else
            r.name = "unnamed-" + std::to_string(unnamed++);
        while (in && in.peek() != '>') {
            if (std::getline(in, line))
                for (char c : line)
                    if (isalpha((unsigned char)c)) r.seq.push_back(c);
        }
        if (!r.seq.empty()) recs.push_back(std::move(r));
    }
    return true;
}

struct Decoded { std::vector<PathState> path; double lnv; int status; };

struct Session {
    augx_model *model = nullptr;
    augx_decoder *dec = nullptr;
    OutputOptions oo;
    int geneid = 1;
    std::string err;

    // decode a set of pieces on the GPU
    bool decode(const std::vector<augx_piece> &pieces, std::vector<Decoded> &out) {
        std::vector<augx_path> paths(pieces.size());
        int rc = augx_decode_batch(dec, pieces.data(), (int)pieces.size(), paths.data());
        if (rc) { err = augx_last_error(); return false; }
        out.resize(pieces.size());
        for (size_t i = 0; i < pieces.size(); i++) {
            out[i].lnv = paths[i].ln_viterbi;
            out[i].status = paths[i].status;
            out[i].path.clear();
            for (int k = 0; k < paths[i].n_states; k++)
                out[i].path.push_back({paths[i].states[k].begin, paths[i].states[k].end, paths[i].states[k].type});
            augx_path_free(&paths[i]);
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

} // namespace

extern "C" int augx_main(int argc, const char *const *argv) {
    Session S;
    std::string commandline;
    for (int i = 0; i < argc; i++) { commandline += argv[i]; if (i < argc - 1) commandline += " "; }
    auto fail = [&](const std::string &msg) {
        std::cerr << "\n" << (argc > 0 ? argv[0] : "augustus") << ": ERROR\n\t" << msg << "\n\n";
        if (S.dec) augx_decoder_destroy(S.dec);
        if (S.model) augx_model_destroy(S.model);
        return 1;
    };
    if (argc <= 1) {
        std::cout << "AUGUSTUS-MI355X (ab-initio GHMM Viterbi decode on gfx950; drop-in for AUGUSTUS 3.5.0 ab-initio prediction)\n\n"
                  << "usage:\naugustus [parameters] --species=SPECIES queryfilename\n";
        return 0;
    }
    // ---- command line (reference Properties::init, src/properties.cc:66-135)
    std::vector<std::pair<std::string, std::string>> cmd;
    std::string queryfile, species, configPath;
    for (int a = argc - 1; a >= 1; a--) {
        std::string s(argv[a]);
        if (s.size() > 2 && s.compare(0, 2, "--") == 0) {
            s.erase(0, 2);
            size_t pos = s.find('=');
            std::string name = s.substr(0, pos);
            if (pos == std::string::npos || pos >= s.size() - 1)
                return fail("Wrong argument format for " + name + ". Use: --argument=value");
            std::string value = s.substr(pos + 1);
            if (name == "species") species = value;
            else if (name == "AUGUSTUS_CONFIG_PATH") configPath = value;
            else cmd.insert(cmd.begin(), {name, value});
        } else if (queryfile.empty())
            queryfile = s;
        else
            return fail("Error: 2 query files given: " + queryfile + " and " + s + ".\nparameter names must start with '--'");
    }
    if (species.empty()) return fail("No species specified. Type \"augustus --species=help\" to see available species.");
    if (configPath.empty()) {
        const char *e = getenv("AUGUSTUS_CONFIG_PATH");
        if (e) configPath = e;
        else return fail("AUGUSTUS_CONFIG_PATH is not set and --AUGUSTUS_CONFIG_PATH was not given.");
    }
    if (configPath.back() != '/') configPath += '/';
    struct stat sb;
    if (stat(configPath.c_str(), &sb) == -1 || !S_ISDIR(sb.st_mode))
        return fail(configPath + " is not a directory. Could not locate directory AUGUSTUS_CONFIG_PATH.");
    {   // unknown parameters are an error (reference src/properties.cc:225-319, config/parameters/aug_cmdln_parameters.json)
        std::ifstream pj((configPath + "parameters/aug_cmdln_parameters.json").c_str());
        if (pj) {
            std::stringstream ss;
            ss << pj.rdbuf();
            const std::string js = ss.str();
            for (auto &kv : cmd)
                if (js.find("\"name\": \"" + kv.first + "\"") == std::string::npos && js.find("\"name\":\"" + kv.first + "\"") == std::string::npos)
                    return fail("Unknown parameter: \"" + kv.first + "\". Type \"augustus\" for help.");
        }
    }
    std::vector<const char *> names, values;
    for (auto &kv : cmd) { names.push_back(kv.first.c_str()); values.push_back(kv.second.c_str()); }
    int rc = augx_model_load(configPath.c_str(), species.c_str(), (int)cmd.size(), names.data(), values.data(), &S.model);
    if (rc) return fail(augx_last_error());
    const Model &M = S.model->m;
    const augx_tables &T = M.t;
    S.oo.fromModel(M);
    if (queryfile.empty()) return fail("No query file specified. Type \"augustus\" for help.");
    if (M.opt.getInt("sample", 0) > 0)
        return fail("sampling (--sample>0: forward algorithm + posterior probabilities) is not implemented on the MI355X path yet; "
                    "run with --sample=0 (the human default).");
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
            if (!in) { restore(); return fail("Could not open input file " + queryfile); }
            ok = readFasta(in, recs);
        }
        if (!ok) { restore(); return fail("File format of " + queryfile + " not recognized (only FASTA input is supported on the MI355X path)."); }
    }
    if (verbosity > 2) {
        if (queryfile == "-") std::cout << "# Reading sequences from standard input. Assuming fasta format." << std::endl;
        else std::cout << "# Looks like " << queryfile << " is in fasta format." << std::endl;
    }
    if (verbosity > 0) std::cout << "# We have hints for 0 sequences and for 0 of the sequences in the input set." << std::endl;

    int device = 0;
    if (const char *e = getenv("AUGX_DEVICE")) device = atoi(e);
    rc = augx_decoder_create(S.model, device, &S.dec);
    if (rc) { restore(); return fail(augx_last_error()); }

    const long maxstep = M.opt.getInt("maxDNAPieceSize", 1000000);
    if (maxstep < 1000) { std::cerr << "maxDNAPieceSize is too small: " << maxstep << std::endl; restore(); return 1; }
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

    // ---- phase 1: find the cut points of all records (reference src/namgene.cc:973-1133).  Inside a record the cuts are a
    //      serial chain (the next exam window starts where the last piece ended), but records are independent: every round
    //      decodes the pending exam window of EVERY unfinished record in one batch.
    struct PieceRef { int rec; long begin, end; int initKind, termKind; };
    struct CutState {
        long beginPos = 0;
        int prevInit = 0, prevTerm = 0; // init/term kinds in effect while the exam window is decoded (state leak, src/namgene.cc:576 vs 594-603)
        int attempt = 0;
        long examChunk = 0, es = 0, ee = 0;
        bool done = false;
    };
    std::vector<std::vector<PieceRef>> recPieces(recs.size());
    {
        std::vector<CutState> cs(recs.size());
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
                p.seq = recs[r].seq.data() + c.es; p.len = c.ee - c.es + 1; p.init_kind = c.prevInit; p.term_kind = c.prevTerm;
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
                if (dd[k].status != 0) { restore(); return fail("No feasible path found in HMM"); }
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
    std::vector<PieceRef> allPieces;
    for (auto &v : recPieces) allPieces.insert(allPieces.end(), v.begin(), v.end());

    // ---- phase 2: decode all pieces in batches bounded by a slot budget.  (Taking the batches in turn on two decoders /
    //      HIP streams, as bench.py does with its resident batches, was measured here and does not pay: genome pieces are
    //      maxDNAPieceSize long, the trellis kernel's time is set by the piece length, and two half-size batches in flight
    //      keep no more compute units busy than one full-size batch.)
    std::vector<Decoded> decoded(allPieces.size());
    {
        long budget = (long)augx_decoder_batch_capacity(S.dec); // bases per batch: what the free HBM holds, at most 128 Mbp
        if (const char *e = getenv("AUGX_BATCH_BASES")) budget = atol(e);
        size_t i = 0;
        while (i < allPieces.size()) {
            std::vector<augx_piece> batch;
            long total = 0;
            size_t j = i;
            while (j < allPieces.size() && (batch.empty() || total + (allPieces[j].end - allPieces[j].begin + 1) <= budget)) {
                augx_piece p;
                p.seq = recs[allPieces[j].rec].seq.data() + allPieces[j].begin;
                p.len = allPieces[j].end - allPieces[j].begin + 1;
                p.init_kind = allPieces[j].initKind;
                p.term_kind = allPieces[j].termKind;
                batch.push_back(p);
                total += p.len;
                j++;
            }
            std::vector<Decoded> dd;
            if (!S.decode(batch, dd)) { restore(); return fail(S.err); }
            for (size_t k = 0; k < dd.size(); k++) decoded[i + k] = std::move(dd[k]);
            i = j;
        }
    }

    // ---- phase 3: gene structures + GFF, in input order (gene ids are global and sequential, src/namgene.cc:526)
    size_t pi = 0;
    int successful = 0;
    for (size_t r = 0; r < recs.size(); r++) {
        const Record &rec = recs[r];
        if (verbosity)
            std::cout << "#\n# ----- prediction on sequence number " << (r + 1) << " (length = " << rec.seq.size()
                      << ", name = " << rec.name << ") -----" << std::endl << "#" << std::endl;
        {
            const std::string st = M.opt.get("strand", "both");
            const bool fw = st == "forward", bw = st == "backward"; // (other values fall back to both, see genes.cc)
            std::cout << "# Predicted genes for sequence number " << (r + 1) << " on " << (fw ? "forward strand" : bw ? "reverse strand" : "both strands") << std::endl;
        }
        bool any = false;
        std::string errmsg;
        for (; pi < allPieces.size() && allPieces[pi].rec == (int)r; pi++) {
            const PieceRef &pr = allPieces[pi];
            const Decoded &d = decoded[pi];
            if (d.status != 0) {
                errmsg = d.status == AUGX_E_UNSUPPORTED
                             ? "piece outside what the MI355X path decodes"
                             : "No feasible path found in HMM";
                continue;
            }
            std::vector<Transcript> txs;
            try {
                txs = filterTranscripts(M, projectOntoGeneSequence(M, d.path, pr.end - pr.begin + 1));
            } catch (std::exception &e) { errmsg = e.what(); continue; }
            std::vector<GeneOut> genes = groupToGenes(txs);
            for (GeneOut &g : genes) {
                g.seqname = rec.name;
                if (S.oo.uniqueGeneId) { char buf[64]; snprintf(buf, sizeof buf, "%.30s.g%d", rec.name.c_str(), S.geneid); g.id = buf; }
                else g.id = "g" + std::to_string(S.geneid);
                int tid = 1;
                for (Transcript &t : g.transcripts) {
                    t.shift(pr.begin);
                    t.seqname = rec.name;
                    t.id = "t" + std::to_string(tid++);
                    t.geneid = g.id;
                }
                S.geneid++;
                any = true;
            }
            std::string text;
            printGeneList(text, genes, rec.seq.data(), (long)rec.seq.size(), S.oo);
            std::cout << text;
        }
        if (!errmsg.empty()) {
            if (successful < 1) { restore(); return fail(errmsg); }
            std::cerr << "\n augustus: ERROR\n\t" << errmsg << "\n\n";
        } else
            successful++;
        if (!any) std::cout << "# (none)" << std::endl;
    }
    (void)T;
    std::cout << "# command line:" << std::endl << "# " << commandline << std::endl;
    restore();
    augx_decoder_destroy(S.dec);
    augx_model_destroy(S.model);
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
        std::vector<GeneOut> genes = groupToGenes(filterTranscripts(m->m, projectOntoGeneSequence(m->m, path, (long)len)));
        int gid = first_gene_id;
        for (GeneOut &g : genes) {
            g.seqname = name;
            g.id = "g" + std::to_string(gid++);
            int tid = 1;
            for (Transcript &t : g.transcripts) { t.seqname = name; t.id = "t" + std::to_string(tid++); t.geneid = g.id; }
        }
        std::string text;
        printGeneList(text, genes, seq, (long)len, oo);
        if (n_genes) *n_genes = (int)genes.size();
        if ((int64_t)text.size() + 1 > out_cap) { setLastError("augx_format_gff: output buffer too small"); return AUGX_E_ARG; }
        memcpy(out, text.c_str(), text.size() + 1);
        return AUGX_OK;
    } catch (std::exception &e) {
        setLastError(e.what());
        return AUGX_E_CONFIG;
    }
}
