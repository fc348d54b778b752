/*
 * ref_harness.cc -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
 *
 * A small driver of our own that links the REAL reference objects (built by oracle/Makefile from the
 * sources under /root/reference, nothing copied) and dumps what the `augustus` CLI never prints:
 *   - the full-precision ln of the Viterbi score of each decoded piece,
 *   - the raw (un-condensed) Viterbi state path,
 *   - optionally every live trellis cell  ln V[j][s]  and the GC-class stairs.
 *
 * It drives only entry points the reference itself uses on the hot path:
 *   Properties::init / Constant::init / ... as in main()            (reference src/augustus.cc:94-190)
 *   SequenceFeatureCollection::prepare                              (src/augustus.cc:409)
 *   NAMGene::getTrainViterbiPath = viterbiAndForward+getViterbiPath (src/namgene.cc:1218-1226)
 *   NAMGene::getViterbiVariables                                    (include/namgene.hh:65)
 * Interior-cut pieces (init/term = synch state only, src/namgene.cc:594-603) are reproduced by
 * patching NAMGene::initProbs/termProbs, which is why the private members are opened below.
 *
 * usage: ref_harness [--name=value ...] --species=SP [--dumpcells=FILE] [--dumpforward=FILE] [--initkind=0|1] [--termkind=0|1] in.fa
 * stdout (one block per FASTA record):
 *   SEQ <name> <len>
 *   LNV <%.17g>
 *   NSTATES <n>
 *   ST <begin> <end> <stateTypeId> <stateTypeName>        (5'->3', 0-based HMM-state coordinates)
 *   END
 * --dumpcells file (binary, little endian): per record  int32 len, int32 S, then len*S doubles
 *   (ln V[j][s], -inf where the cell is absent), then len int32 GC-class indices.
 */
#include <sstream>
#include <fstream>
#include <iostream>
#include <iomanip>
#include <string>
#include <vector>
#include <list>
#include <map>
#include <set>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <climits>
#include <limits>
#include <stdint.h>
#include <unordered_map>
#include <unordered_set>
#include <deque>
#include <stack>
#include <queue>
#include <bitset>
#include <exception>
#include <stdexcept>
#include <memory>
#include <functional>
#include <iterator>
#include <utility>
#include <cstdlib>
#include <cctype>
#include <cassert>
#include <ctime>
#include <typeinfo>
#include <numeric>
#include <complex>
#include <valarray>
#include <array>
#define private public
#define protected public
#include "namgene.hh"
#undef private
#undef protected
#include "types.hh"
#include "gene.hh"
#include "genbank.hh"
#include "evaluation.hh"
#include "statemodel.hh"
#include "extrinsicinfo.hh"
#include "properties.hh"
#include "motif.hh"
#include "pp_profile.hh"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

int verbosity = 0;          // augustus.cc:27-28 defines these in the real binary
bool mea_prediction = false;

int main(int argc, char *argv[]) {
    std::string dumpfile, fwdfile, smpfile;
    int nsamples = 0;
    int initkind = 0, termkind = 0;
    std::string kindlist, nosample, probecell;
    std::vector<char *> args;
    for (int i = 0; i < argc; i++) {
        if (strncmp(argv[i], "--dumpcells=", 12) == 0) dumpfile = argv[i] + 12;
        else if (strncmp(argv[i], "--dumpforward=", 14) == 0) fwdfile = argv[i] + 14; // (needs --sample > 0: the forward table is only filled then)
        else if (strncmp(argv[i], "--dumpsamples=", 14) == 0) smpfile = argv[i] + 14; // text: per record "SEQ name", then per sample "SAMPLE i n" + "ST b e type" lines
        else if (strncmp(argv[i], "--nsamples=", 11) == 0) nsamples = atoi(argv[i] + 11);
        else if (strncmp(argv[i], "--initkind=", 11) == 0) initkind = atoi(argv[i] + 11);
        else if (strncmp(argv[i], "--termkind=", 11) == 0) termkind = atoi(argv[i] + 11);
        else if (strncmp(argv[i], "--nosample=", 11) == 0) nosample = std::string(",") + (argv[i] + 11) + ","; // records (0-based, comma separated) that are decoded but not sampled: an exam window of the cut finder
        else if (strncmp(argv[i], "--probecell=", 12) == 0) probecell = argv[i] + 12; // "state:base:N": N draws of the options of that cell (doSampling), histogram of the predecessor ends on stdout
        else if (strncmp(argv[i], "--kindlist=", 11) == 0) kindlist = argv[i] + 11; // per record "ik:tk,ik:tk,..." (the pieces of one record, as NAMGene::doViterbiPiecewise sets them, src/namgene.cc:594-603)
        else args.push_back(argv[i]);
    }
    int nargs = (int)args.size();
    FILE *dump = dumpfile.empty() ? NULL : fopen(dumpfile.c_str(), "wb");
    FILE *sdump = smpfile.empty() ? NULL : fopen(smpfile.c_str(), "w");
    FILE *fdump = fwdfile.empty() ? NULL : fopen(fwdfile.c_str(), "wb");
    try {
        LLDouble::setOutputPrecision(3);
        Properties::init(nargs, args.data());
        Constant::init();
        Gene::init();
        GeneticCode::init();
        if (Properties::hasProperty("translation_table")) // (setParameters(), reference src/augustus.cc:526-528)
            GeneticCode::chooseTranslationTable(Properties::getIntProperty("translation_table"));
        StateModel::init();
        std::string filename = Properties::getProperty(INPUTFILE_KEY);
        GBProcessor gbank(filename);
        FeatureCollection extrinsicFeatures;
        std::streambuf *oldbuf = std::cout.rdbuf();
        std::ostringstream sink;
        std::cout.rdbuf(sink.rdbuf()); // readExtrinsicCFGFile and NAMGene() print "# ..." lines
        if (Constant::softmasking)
            extrinsicFeatures.readExtrinsicCFGFile();
        BaseCount::init();
        PP::initConstants();
        NAMGene namgene;
        StateModel::readAllParameters();
        std::cout.rdbuf(oldbuf);
        int S = namgene.statecount;
        int synch = 0;
        try { synch = Properties::getIntProperty("/NAMGene/SynchState"); } catch (...) {}
        for (int i = 0; i < S; i++) {
            if (initkind == 1) namgene.initProbs[i] = (i == synch) ? 1.0 : 0.0;
            if (termkind == 1) namgene.termProbs[i] = (i == synch) ? 1.0 : 0.0;
        }
        std::vector<double> origInit(S), origTerm(S);
        for (int i = 0; i < S; i++) { origInit[i] = namgene.initProbs[i].doubleValue(); origTerm[i] = namgene.termProbs[i].doubleValue(); }
        int recNo = 0;
        AnnoSequence *seq = gbank.getSequenceList();
        while (seq) {
            if (!kindlist.empty()) { // (this record's kinds)
                size_t at = 0;
                for (int k = 0; k < recNo && at != std::string::npos; k++) { at = kindlist.find(',', at); if (at != std::string::npos) at++; }
                if (at != std::string::npos && at + 2 < kindlist.size() + 1) {
                    const int ik = kindlist[at] - '0', tk = kindlist[at + 2] - '0';
                    for (int i = 0; i < S; i++) {
                        namgene.initProbs[i] = ik == 1 ? ((i == synch) ? 1.0 : 0.0) : origInit[i];
                        namgene.termProbs[i] = tk == 1 ? ((i == synch) ? 1.0 : 0.0) : origTerm[i];
                    }
                }
            }
            recNo++;
            AnnoSequence *cur = seq;
            seq = seq->next;
            cur->next = NULL;
            SequenceFeatureCollection &sfc = extrinsicFeatures.getSequenceFeatureCollection(cur->seqname);
            std::cout.rdbuf(sink.rdbuf());
            sfc.prepare(cur, false);
            sfc.setSeqLen(strlen(cur->sequence));
            sfc.makeGroups();
            sfc.prepareLocalMalus(cur->sequence);
            std::cout.rdbuf(oldbuf);
            int n = strlen(cur->sequence);
            printf("SEQ %s %d\n", cur->seqname, n);
            StatePath *p = NULL;
            try {
                p = namgene.getTrainViterbiPath(cur->sequence, &sfc);
            } catch (ProjectError &e) {
                printf("ERR %s\nEND\n", e.getMessage().c_str());
                continue;
            }
            printf("LNV %.17g\n", p->pathemiProb.log());
            // runs of single-base igenic / geometric-intron / UTR-intron states are merged into one record
            // (same merge StatePath::condenseStatePath does, src/gene.cc:977-1000); everything else is raw
            std::vector<State> recs;
            for (State *st = p->first; st; st = st->next) {
                bool mergeable = st->type == igenic || isGeometricIntron(st->type) || isRGeometricIntron(st->type) || st->type == utr5intron || st->type == utr3intron || st->type == rutr5intron || st->type == rutr3intron;
                if (mergeable && !recs.empty() && recs.back().type == st->type && recs.back().end + 1 == st->begin)
                    recs.back().end = st->end;
                else
                    recs.push_back(State(st->begin, st->end, st->type));
            }
            printf("NSTATES %d\n", (int)recs.size());
            for (size_t r = 0; r < recs.size(); r++)
                printf("ST %d %d %d %s\n", recs[r].begin, recs[r].end, (int)recs[r].type, stateTypeIdentifiers[recs[r].type]);
            printf("END\n");
            if (dump) {
                const ViterbiMatrixType &v = namgene.getViterbiVariables();
                int32_t hdr[2] = {n, S};
                fwrite(hdr, 4, 2, dump);
                std::vector<double> col(S);
                for (int j = 0; j < n; j++) {
                    for (int i = 0; i < S; i++) {
                        Double val = v[j].get(i);
                        col[i] = (val > 0) ? val.log() : -std::numeric_limits<double>::infinity();
                    }
                    fwrite(col.data(), 8, S, dump);
                }
                std::vector<int32_t> gc(n);
                for (int j = 0; j < n; j++) gc[j] = namgene.cs.idx[j];
                fwrite(gc.data(), 4, n, dump);
            }
            if (fdump) { // ln of the forward variables (reference NAMGene::getForwardVariables, include/namgene.hh:56), same layout
                const ViterbiMatrixType &v = namgene.getForwardVariables();
                int32_t hdr[2] = {n, S};
                fwrite(hdr, 4, 2, fdump);
                std::vector<double> col(S);
                for (int j = 0; j < n; j++) {
                    for (int i = 0; i < S; i++) {
                        Double val = v[j].get(i);
                        col[i] = (val > 0) ? val.log() : -std::numeric_limits<double>::infinity();
                    }
                    fwrite(col.data(), 8, S, fdump);
                }
            }
            if (!probecell.empty()) {
                int ps = 0, pb = 0, pn = 0;
                sscanf(probecell.c_str(), "%d:%d:%d", &ps, &pb, &pn);
                std::map<int, int> hist;
                for (int it = 0; it < pn; it++) {
                    OptionListItem oli;
                    namgene.states[ps]->viterbiForwardAndSampling(namgene.viterbi, namgene.forward, ps, pb, doSampling, oli);
                    hist[oli.base * 1000 + oli.state]++;
                }
                for (auto &kv : hist) printf("PROBE base %d state %d count %d\n", kv.first >= 0 ? kv.first / 1000 : -((-kv.first + 999) / 1000), ((kv.first % 1000) + 1000) % 1000, kv.second);
            }
            if (sdump) { // reference NAMGene::getSampledPath (src/namgene.cc:367): draws from rand(), one stream over the run
                fprintf(sdump, "SEQ %s\n", cur->seqname);
                for (int it = 0; it < nsamples && nosample.find("," + std::to_string(recNo - 1) + ",") == std::string::npos; it++) {
                    StatePath *sp = namgene.getSampledPath(cur->sequence, cur->seqname);
                    std::vector<State> sr;
                    for (State *st = sp->first; st; st = st->next) {
                        bool mergeable = st->type == igenic || isGeometricIntron(st->type) || isRGeometricIntron(st->type) || st->type == utr5intron || st->type == utr3intron || st->type == rutr5intron || st->type == rutr3intron;
                        if (mergeable && !sr.empty() && sr.back().type == st->type && sr.back().end + 1 == st->begin)
                            sr.back().end = st->end;
                        else
                            sr.push_back(State(st->begin, st->end, st->type));
                    }
                    fprintf(sdump, "SAMPLE %d %d\n", it, (int)sr.size());
                    for (size_t r = 0; r < sr.size(); r++) fprintf(sdump, "ST %d %d %d\n", sr[r].begin, sr[r].end, (int)sr[r].type);
                    delete sp;
                }
            }
            delete p;
        }
    } catch (ProjectError &err) {
        fprintf(stderr, "ref_harness: ERROR\n\t%s\n", err.getMessage().c_str());
        return 1;
    }
    if (dump) fclose(dump);
    if (fdump) fclose(fdump);
    if (sdump) fclose(sdump);
    return 0;
}
