// genes.cc -- see genes.h
#include "genes.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace augx {

static inline int mod3(int k) { return k >= 0 ? k % 3 : (k % 3 + 3) % 3; }
// type predicates, reference include/types.hh:520-640 (type ids: include/types.hh:492-512)
static inline bool isInitialExon(int t) { return t >= 2 && t <= 4; }
static inline bool isInternalExon(int t) { return t >= 5 && t <= 7; }
static inline bool isRInternalExon(int t) { return t >= 38 && t <= 40; }
static inline bool isRTerminalExon(int t) { return t >= 41 && t <= 43; }
static inline bool isCodingExon(int t) { return (t >= 1 && t <= 8) || (t >= 36 && t <= 43); }
static inline bool isCodingIntron(int t) { return (t >= 9 && t <= 23) || (t >= 44 && t <= 58); }
static inline bool is5UTR(int t) { return (t >= 24 && t <= 29) || (t >= 59 && t <= 64); }
static inline bool is5UTRIntron(int t) { return t == 26 || t == 27 || t == 61 || t == 62; }
static inline bool is3UTR(int t) { return (t >= 30 && t <= 35) || (t >= 65 && t <= 70); }
static inline bool is3UTRIntron(int t) { return t == 32 || t == 33 || t == 67 || t == 68; }
static inline bool isUtrExon(int t) { return (is5UTR(t) && !is5UTRIntron(t)) || (is3UTR(t) && !is3UTRIntron(t)); }
static inline bool isExonType(int t) { return isCodingExon(t) || isUtrExon(t); }
static inline bool isIntron(int t) { return isCodingIntron(t) || is5UTRIntron(t) || is3UTRIntron(t) || t == TYPE_INTRON || t == TYPE_RINTRON; }
enum { T_UTR5SINGLE = 24, T_UTR5INIT = 25, T_UTR5INTERNAL = 28, T_UTR5TERM = 29, T_UTR3SINGLE = 30, T_UTR3INIT = 31, T_UTR3INTERNAL = 34,
       T_UTR3TERM = 35, T_RUTR5SINGLE = 59, T_RUTR5INIT = 60, T_RUTR5INTERNAL = 63, T_RUTR5TERM = 64, T_RUTR3SINGLE = 65, T_RUTR3INIT = 66,
       T_RUTR3INTERNAL = 69, T_RUTR3TERM = 70 };
static inline bool isOnFStrand(int t) { return !((t >= 36 && t <= 70) || t == TYPE_RINTRON || (t >= 80 && t <= 85)); }
enum { T_SINGLE = 1, T_TERMINAL = 8, T_RSINGLE = 36, T_RINITIAL = 37 };

int BioState::frame() const { return mod3(winOfType(type) + framemod); }

// reference State::setTruncFlag, src/gene.cc:309-321
static int truncFlag(int type, long begin, long end, long dnalen) {
    int tr = 0;
    long predEnd = begin - 1;
    if (end == dnalen - 1 && (isInitialExon(type) || isInternalExon(type) || isRTerminalExon(type) || isRInternalExon(type) || isIntron(type) ||
                              type == T_UTR3SINGLE || type == T_UTR3TERM))
        tr |= TRUNC_RIGHT;
    if ((predEnd == -1 || predEnd == 0) &&
        (isInternalExon(type) || type == T_TERMINAL || isRInternalExon(type) || type == T_RINITIAL || isIntron(type) || isUtrExon(type)))
        tr |= TRUNC_LEFT;
    return tr;
}

// reference State::getBiologicalState, src/gene.cc:176-300 (coding exons and introns only)
static BioState bioState(const Model &m, long begin, long end, int type, int truncated) {
    const augx_tables &t = m.t;
    BioState b;
    int beginShift = 0, endShift = 0;
    if (type == T_SINGLE || isInitialExon(type)) beginShift = t.W;
    else if (isInternalExon(type) || type == T_TERMINAL) { if (!(truncated & TRUNC_LEFT)) beginShift = -t.Ae; }
    else if (isRInternalExon(type) || type == T_RINITIAL) { if (!(truncated & TRUNC_LEFT)) beginShift = -t.Ds; }
    else if (type == TYPE_INTRON) beginShift = !(truncated & TRUNC_LEFT) ? t.Ds : -1;
    else if (type == TYPE_RINTRON) beginShift = !(truncated & TRUNC_LEFT) ? t.Ae : -1;
    else if (type == T_UTR5SINGLE || type == T_UTR5INIT) beginShift = t.tss_upwin;
    else if (type == T_RUTR5SINGLE) beginShift = !(truncated & TRUNC_LEFT) ? -t.W : (int)-begin;
    else if (type == T_RUTR5INIT || type == T_RUTR5INTERNAL || type == T_RUTR3INIT || type == T_RUTR3INTERNAL) beginShift = t.De + 2;
    else if (type == T_UTR5INTERNAL || type == T_UTR3INTERNAL || type == T_UTR3TERM || type == T_UTR5TERM) beginShift = t.U + t.As + 2;
    else if (type == T_RUTR5TERM) beginShift = -t.W;
    else if (type == T_UTR3SINGLE) { if ((truncated & TRUNC_LEFT) && begin == 1) beginShift = -1; }
    else if (type == T_RUTR3SINGLE || type == T_RUTR3TERM) { if (begin < 0) beginShift = (int)-begin; }
    if (type == T_UTR5SINGLE || type == T_UTR5TERM) endShift = t.W;
    else if (type == T_RUTR5SINGLE || type == T_RUTR5INIT) endShift = -t.tss_upwin;
    else if (type == T_UTR5INIT || type == T_UTR5INTERNAL || type == T_UTR3INIT || type == T_UTR3INTERNAL) endShift = -t.De - 2;
    else if (type == T_RUTR5INTERNAL || type == T_RUTR5TERM || type == T_RUTR3INTERNAL || type == T_RUTR3TERM) endShift = -t.U - t.As - 2;
    if (type == T_RSINGLE || type == T_RINITIAL) endShift = -t.W;
    else if (isInitialExon(type) || isInternalExon(type)) {
        if (!(truncated & TRUNC_RIGHT)) endShift = t.Ds; else b.framemod = mod3(-t.Ds);
    } else if (isRTerminalExon(type) || isRInternalExon(type)) {
        if (!(truncated & TRUNC_RIGHT)) endShift = t.Ae; else b.framemod = mod3(t.Ae);
    } else if (type == TYPE_INTRON) { if (!(truncated & TRUNC_RIGHT)) endShift = -t.Ae; }
    else if (type == TYPE_RINTRON) { if (!(truncated & TRUNC_RIGHT)) endShift = -t.Ds; }
    b.begin = begin + beginShift;
    b.end = end + endShift;
    b.type = type;
    b.truncated = truncated;
    return b;
}

bool Transcript::completeCDS() const { // reference Gene::completeCDS, src/gene.cc:1977-1988
    if (exons.empty()) return false;
    int ft = exons.front().type, lt = exons.back().type;
    if (isInternalExon(ft) || isRInternalExon(ft) || isInternalExon(lt) || isRInternalExon(lt) || ft == T_TERMINAL ||
        isRTerminalExon(lt) || ft == T_RINITIAL || isInitialExon(lt))
        return false;
    if ((exons.front().truncated & TRUNC_LEFT) || (exons.back().truncated & TRUNC_RIGHT)) return false;
    return true;
}
void Transcript::shift(long d) {
    for (auto &e : exons) { e.begin += d; e.end += d; }
    for (auto &e : introns) { e.begin += d; e.end += d; }
    for (std::vector<BioState> *l : {&utr5exons, &utr3exons, &utr5introns, &utr3introns})
        for (auto &e : *l) { e.begin += d; e.end += d; }
    if (transstart >= 0) transstart += d;
    if (transend >= 0) transend += d;
    codingstart += d;
    codingend += d;
}

void OutputOptions::fromModel(const Model &m) { // reference Gene::init, src/gene.cc:2447-2459
    const Options &o = m.opt;
    transTable = m.transTable;
    print_start = o.getBool("start", true);
    print_stop = o.getBool("stop", true);
    print_introns = o.getBool("introns", false);
    print_cds = o.getBool("cds", true);
    print_exonnames = o.getBool("exonnames", false);
    gff3 = o.getBool("gff3", false);
    stopCodonExcludedFromCDS = o.getBool("stopCodonExcludedFromCDS", false);
    protein = o.getBool("protein", true);
    codingseq = o.getBool("codingseq", false);
    uniqueGeneId = o.getBool("uniqueGeneId", false);
    print_utr = o.getBool("print_utr", false);
    print_tss = o.getBool("tss", false);
    print_tts = o.getBool("tts", false);
    utr = m.t.utr != 0;
    // "# Evidence for and against" is printed when the hints machinery is on, i.e. with softmasking (default true,
    // reference src/types.cc:95, src/extrinsicinfo.cc:1722, src/gene.cc:3111) and printEvidence (default true)
    softmasking = o.getBool("softmasking", true);
    evidence = softmasking && o.getBool("printEvidence", true);
}

std::vector<Transcript> projectOntoGeneSequence(const Model &m, const std::vector<PathState> &path, long dnalen) {
    std::vector<Transcript> out;
    size_t i = 0;
    const size_t n = path.size();
    int genenumber = 1;
    bool haveGene = false;
    Transcript g;
    auto tr = [&](const PathState &s) { return truncFlag(s.type, s.begin, s.end, dnalen); };
    // incomplete gene beginning with an intron in the CDS (src/gene.cc:403-417)
    if (n > 0 && isCodingIntron(path[0].type)) {
        long b = path[0].begin;
        int trunc = tr(path[0]);
        bool fwd = isOnFStrand(path[0].type);
        while (i + 1 < n && isCodingIntron(path[i + 1].type)) i++;
        trunc |= tr(path[i]);
        g = Transcript();
        haveGene = true;
        g.introns.push_back(bioState(m, b, path[i].end, fwd ? TYPE_INTRON : TYPE_RINTRON, trunc));
        g.transstart = g.introns.back().begin;
        i++;
    }
    while (i < n) {
        while (i < n && !isExonType(path[i].type)) i++;
        if (i >= n) break;
        if (!haveGene) { g = Transcript(); haveGene = true; }
        g.plus = isOnFStrand(path[i].type);
        if (!g.plus) g.frame = 2;
        // the UTR left of the coding region: the 5' UTR of a forward gene, the 3' UTR of a reverse one (src/gene.cc:473-504)
        bool haveLeft5 = false, haveLeft3 = false;
        if (is5UTR(path[i].type)) {
            while (i < n && is5UTR(path[i].type)) {
                if (!haveLeft5) { g.complete5utr = path[i].type == T_UTR5SINGLE || path[i].type == T_UTR5INIT; haveLeft5 = true; }
                if (isExonType(path[i].type)) g.utr5exons.push_back(bioState(m, path[i].begin, path[i].end, path[i].type, tr(path[i])));
                i++;
            }
        } else if (is3UTR(path[i].type)) {
            while (i < n && is3UTR(path[i].type)) {
                if (!haveLeft3) { g.complete3utr = path[i].type == T_RUTR3SINGLE || path[i].type == T_RUTR3TERM; haveLeft3 = true; }
                if (isExonType(path[i].type)) g.utr3exons.push_back(bioState(m, path[i].begin, path[i].end, path[i].type, tr(path[i])));
                i++;
            }
        }
        bool last5 = false, last3 = false; // (a UTR was read to the RIGHT of the coding region: last5utrexon / last3utrexon of the reference)
        if (i < n && isExonType(path[i].type)) {
        const int ty = path[i].type;
        if (ty == T_SINGLE || ty == T_RSINGLE) {
            g.exons.push_back(bioState(m, path[i].begin, path[i].end, ty, tr(path[i])));
            i++;
        } else {
            if (!isInitialExon(ty) && !isRTerminalExon(ty)) g.complete = false;
            BioState first = bioState(m, path[i].begin, path[i].end, ty, tr(path[i]));
            g.exons.push_back(first);
            g.frame = g.plus ? mod3(first.frame() - first.length()) : mod3(first.frame() + first.length());
            if (ty == T_TERMINAL || ty == T_RINITIAL) {
                i++;
            } else {
                i++;
                while (i < n && path[i].type != T_TERMINAL && path[i].type != T_RINITIAL) {
                    int t2 = path[i].type;
                    if (isIntron(t2)) {
                        long b = path[i].begin;
                        bool fwd = isOnFStrand(t2);
                        while (i + 1 < n && isIntron(path[i + 1].type)) i++;
                        g.introns.push_back(bioState(m, b, path[i].end, fwd ? TYPE_INTRON : TYPE_RINTRON, tr(path[i])));
                        if (g.introns.back().end > g.transstart) g.transend = g.introns.back().end;
                    } else if (isInternalExon(t2) || isRInternalExon(t2)) {
                        g.exons.push_back(bioState(m, path[i].begin, path[i].end, t2, tr(path[i])));
                    } else
                        throw std::runtime_error("state path doesn't constitute a valid gene");
                    i++;
                }
                if (i >= n)
                    g.complete = false;
                else {
                    g.exons.push_back(bioState(m, path[i].begin, path[i].end, path[i].type, tr(path[i])));
                    i++;
                }
            }
        }
        // the UTR right of the coding region (src/gene.cc:565-596)
        if (i < n) {
            if (is5UTR(path[i].type)) {
                while (i < n && is5UTR(path[i].type)) {
                    if (!(i + 1 < n && is5UTR(path[i + 1].type))) g.complete5utr = path[i].type == T_RUTR5SINGLE || path[i].type == T_RUTR5INIT;
                    if (isExonType(path[i].type)) { g.utr5exons.push_back(bioState(m, path[i].begin, path[i].end, path[i].type, tr(path[i]))); last5 = true; }
                    i++;
                }
            } else if (is3UTR(path[i].type)) {
                while (i < n && is3UTR(path[i].type)) {
                    if (!(i + 1 < n && is3UTR(path[i + 1].type))) g.complete3utr = path[i].type == T_UTR3SINGLE || path[i].type == T_UTR3TERM;
                    if (isExonType(path[i].type)) { g.utr3exons.push_back(bioState(m, path[i].begin, path[i].end, path[i].type, tr(path[i]))); last3 = true; }
                    i++;
                }
            }
        }
        } else { // the (incomplete) gene consists just of UTR: not reported (Constant::reportUtrOnlyGenes = false, src/gene.cc:598-604)
            haveGene = false;
            continue;
        }
        // finish construction of the gene (src/gene.cc:609-676): UTR introns are the gaps between UTR exons
        for (size_t k = 0; k + 1 < g.utr5exons.size(); k++) { BioState in; in.begin = g.utr5exons[k].end + 1; in.end = g.utr5exons[k + 1].begin - 1; in.type = TYPE_INTRON; g.utr5introns.push_back(in); }
        for (size_t k = 0; k + 1 < g.utr3exons.size(); k++) { BioState in; in.begin = g.utr3exons[k].end + 1; in.end = g.utr3exons[k + 1].begin - 1; in.type = TYPE_INTRON; g.utr3introns.push_back(in); }
        g.clength = 0;
        for (auto &e : g.exons) g.clength += e.length();
        if (!g.plus) g.frame = mod3(g.frame - g.clength + 1);
        if (!g.utr5exons.empty() && (g.transstart < 0 || g.transstart > g.utr5exons.front().begin)) g.transstart = g.utr5exons.front().begin;
        if (!g.utr3exons.empty() && (g.transstart < 0 || g.transstart > g.utr3exons.front().begin)) g.transstart = g.utr3exons.front().begin;
        if (last5 && (g.transend < 0 || g.transend < g.utr5exons.back().end)) g.transend = g.utr5exons.back().end;
        if (last3 && (g.transend < 0 || g.transend < g.utr3exons.back().end)) g.transend = g.utr3exons.back().end;
        g.codingstart = g.exons.front().begin;
        g.codingend = g.exons.back().end;
        if (g.codingend > g.transend) g.transend = -1;
        if (g.codingstart >= 0 && g.codingstart < g.transstart) g.transstart = -1;
        g.id = "g" + std::to_string(genenumber++);
        out.push_back(g);
        haveGene = false;
    }
    return out;
}

// reference Transcript::meanStateProb, src/gene.cc:1241-1254: the geometric mean of the exon and intron probabilities
double Transcript::meanStateProb() const {
    if (!hasProbs) return 0.0;
    double p = 1.0;
    int k = 0;
    for (const std::vector<BioState> *l : {&exons, &introns, &utr5exons, &utr3exons}) // (Gene::getExInHeads, include/gene.hh:379)
        for (const BioState &e : *l) { p *= e.apostprob; k++; }
    return pow(p, 1.0 / k);
}

// reference Gene::hasInFrameStop, src/gene.cc:1422-1438: a stop codon in the reading frame of the CDS before its last codon (the
// stop codon of a gene can be put together by a long intron; short introns are kept from it in the trellis).  A codon with
// anything but acgt in it does not count.
static bool hasInFrameStop(const Transcript &t, const char *seq, int stopMask) {
    std::string cds;
    for (const BioState &e : t.exons) cds.append(seq + e.begin, (size_t)e.length());
    if (!t.plus) {
        std::string r(cds.rbegin(), cds.rend());
        for (char &c : r) {
            const char l = (char)tolower((unsigned char)c);
            c = l == 'a' ? 't' : l == 'c' ? 'g' : l == 'g' ? 'c' : l == 't' ? 'a' : 'n';
        }
        cds.swap(r);
    }
    for (size_t i = (size_t)mod3(-t.frame); i + 3 < cds.size(); i += 3) {
        const char a = (char)tolower((unsigned char)cds[i]), b = (char)tolower((unsigned char)cds[i + 1]), c = (char)tolower((unsigned char)cds[i + 2]);
        if (a == 't' && ((b == 'a' && ((c == 'a' && (stopMask & 1)) || (c == 'g' && (stopMask & 2)))) || (b == 'g' && c == 'a' && (stopMask & 4)))) return true; // (GeneticCode::isStopcodon: the translation table's)
    }
    return false;
}

std::vector<Transcript> filterTranscripts(const Model &m, const std::vector<Transcript> &txs, bool anyStrand, const char *seq) {
    std::vector<Transcript> out;
    const bool noInFrameStop = seq && m.opt.getBool("noInFrameStop", false);
    // --strand (reference src/augustus.cc:177-191, filterGenePrediction src/gene.cc:2474-2475).  Only the values listed as
    // possible_values in aug_cmdln_parameters.json reach the reference's parser: anything but forward / backward means both
    const std::string st = m.opt.get("strand", "both");
    const int want = anyStrand ? 0 : st == "forward" ? 1 : st == "backward" ? -1 : 0;
    const double minmean = m.opt.getDouble("minmeanexonintronprob", 0.0), minprob = m.opt.getDouble("minexonintronprob", 0.0);
    const bool keepViterbi = m.opt.getBool("keep_viterbi", false);
    for (const Transcript &g : txs) {
        bool keep = !(want == 1 && !g.plus) && !(want == -1 && g.plus);
        if (g.throwaway) keep = false;
        bool cc = g.completeCDS();
        if ((g.clength < m.t.min_coding_len && cc) || (g.clength < 4 && g.clength < m.t.min_coding_len && !cc)) keep = false;
        if (noInFrameStop && hasInFrameStop(g, seq, m.t.stop_mask)) keep = false;
        if (keep && g.hasProbs) { // src/gene.cc:2489-2514
            const bool kv = keepViterbi && g.viterbi;
            if (g.meanStateProb() < minmean && !kv) keep = false;
            for (const BioState &e : g.exons)
                if (e.apostprob < minprob && !kv) keep = false;
            for (const BioState &e : g.introns)
                if (e.apostprob < minprob && !kv) keep = false;
        }
        if (keep) out.push_back(g);
    }
    return out;
}

void reverseTranscript(Transcript &t, long endpos) {
    auto mirror = [&](std::vector<BioState> &v, bool exons) {
        for (BioState &e : v) {
            const long b = e.begin;
            e.begin = endpos - e.end;
            e.end = endpos - b;
            if (!exons) continue;
            const int ty = e.type; // (frame() and length() below are those of the type before the change, as in the reference)
            if (isInitialExon(ty)) e.type = T_RINITIAL;
            else if (ty == T_TERMINAL) e.type = 41 + mod3(2 - e.length());
            else if (isInternalExon(ty)) e.type = 38 + mod3(2 + e.frame() - e.length());
            else if (ty == T_SINGLE) e.type = T_RSINGLE;
            else if (isRTerminalExon(ty)) e.type = T_TERMINAL;
            else if (ty == T_RINITIAL) e.type = 2 + mod3(e.length());
            else if (isRInternalExon(ty)) e.type = 5 + mod3(1 + e.frame() + e.length());
            else if (ty == T_RSINGLE) e.type = T_SINGLE;
        }
        std::reverse(v.begin(), v.end());
    };
    mirror(t.exons, true);
    mirror(t.introns, false);
    const long ce = t.codingend;
    t.codingend = t.codingstart >= 0 ? endpos - t.codingstart : -1;
    t.codingstart = ce >= 0 ? endpos - ce : -1;
    const long te = t.transend;
    t.transend = t.transstart >= 0 ? endpos - t.transstart : -1;
    t.transstart = te >= 0 ? endpos - te : -1;
    t.plus = !t.plus;
    t.revRun = true;
}

// reference Transcript::operator==, src/gene.cc:1149-1175: the same exon and intron intervals (types and strand are not compared)
static bool sameIntervals(const Transcript &a, const Transcript &b) {
    const std::vector<BioState> *la[4] = {&a.exons, &a.introns, &a.utr5exons, &a.utr3exons}, *lb[4] = {&b.exons, &b.introns, &b.utr5exons, &b.utr3exons};
    for (int k = 0; k < 4; k++) {
        if (la[k]->size() != lb[k]->size()) return false;
        for (size_t i = 0; i < la[k]->size(); i++)
            if ((*la[k])[i].begin != (*lb[k])[i].begin || (*la[k])[i].end != (*lb[k])[i].end) return false;
    }
    return true;
}
// the four state lists the posterior probabilities run over (Gene::getExInHeads: CDS exons, CDS introns, 5' UTR exons, 3' UTR exons)
static std::vector<BioState> *exInList(Transcript &t, int k) { return k == 0 ? &t.exons : k == 1 ? &t.introns : k == 2 ? &t.utr5exons : &t.utr3exons; }
// reference Transcript::updatePostProb, src/gene.cc:1204-1235
static void mergeCount(std::vector<BioState> &x, std::vector<BioState> &y) {
    size_t i = 0, j = 0;
    while (i < x.size() && j < y.size()) {
        if (x[i].begin == y[j].begin && x[i].end == y[j].end && x[i].type == y[j].type) {
            x[i].apostprob += y[j].sampleCount;
            y[j].apostprob += x[i].sampleCount;
            i++; j++;
        } else if (x[i].begin < y[j].begin) i++;
        else j++;
    }
}

std::vector<Transcript> posteriorTranscripts(const Model &m, const std::vector<PathState> &viterbi,
                                             const std::vector<std::vector<PathState>> &samples, long dnalen, int sampleiterations) {
    std::vector<Transcript> all;
    const bool alternatives = m.opt.getBool("alternatives-from-sampling", false);
    int serial = 0;
    auto add = [&](const std::vector<PathState> &path, bool vit) {
        for (Transcript &g : projectOntoGeneSequence(m, path, dnalen)) {
            g.serial = serial++;
            g.apostprob = 1.0f;
            for (int k = 0; k < 4; k++)
                for (BioState &e : *exInList(g, k)) { e.apostprob = 1.0f; e.sampleCount = 1; e.hasScore = true; }
            g.hasProbs = true;
            g.viterbi = vit;
            g.throwaway = !vit && !alternatives; // (alternatives-from-sampling=false: a sampled transcript only adds to the counts)
            all.push_back(std::move(g));
        }
    };
    add(viterbi, true);
    for (const auto &sp : samples) add(sp, false);
    if (sampleiterations > 1) {
        std::stable_sort(all.begin(), all.end(), [](const Transcript &a, const Transcript &b) { return a.geneBegin() < b.geneBegin(); });
        // copies of one transcript are united, the count goes up instead (src/namgene.cc:876-893)
        std::vector<char> dead(all.size(), 0);
        for (size_t i = 0; i < all.size(); i++) {
            if (dead[i]) continue;
            for (size_t j = i + 1; j < all.size() && all[j].geneBegin() == all[i].geneBegin(); j++) {
                if (dead[j] || !sameIntervals(all[i], all[j])) continue;
                all[i].throwaway = all[i].throwaway && all[j].throwaway;
                all[i].viterbi = all[i].viterbi || all[j].viterbi;
                all[i].apostprob += 1.0f;
                for (int k = 0; k < 4; k++)
                    for (BioState &e : *exInList(all[i], k)) { e.sampleCount += 1; e.apostprob += 1.0f; }
                dead[j] = 1;
            }
        }
        {
            std::vector<Transcript> live;
            for (size_t i = 0; i < all.size(); i++)
                if (!dead[i]) live.push_back(std::move(all[i]));
            all.swap(live);
        }
        // exons and introns shared with overlapping transcripts (src/namgene.cc:898-903)
        for (size_t i = 0; i < all.size(); i++)
            for (size_t j = i + 1; j < all.size() && all[j].geneBegin() <= all[i].geneEnd(); j++) {
                if (all[j].geneBegin() > all[i].geneEnd() || all[i].geneBegin() > all[j].geneEnd()) continue;
                for (int k = 0; k < 4; k++) mergeCount(*exInList(all[i], k), *exInList(all[j], k));
            }
        const float n = (float)sampleiterations;
        for (Transcript &g : all) {
            g.apostprob /= n;
            for (int k = 0; k < 4; k++)
                for (BioState &e : *exInList(g, k)) e.apostprob /= n;
        }
    }
    return all;
}

// reference frame_compatible(State*, State*), src/gene.cc:163-166
static bool frameCompatible(const BioState &a, const BioState &b) {
    const bool fa = isOnFStrand(a.type), fb = isOnFStrand(b.type);
    return (fa && fb && mod3((int)(b.end - a.end) - b.frame() + a.frame()) == 0) ||
           (!fa && !fb && mod3((int)(b.end - a.end) + b.frame() - a.frame()) == 0);
}

namespace {
struct AltGeneBuild { // reference AltGene, src/gene.cc:2676-2731
    std::vector<const Transcript *> tx;
    bool plus = true;
    long mincodstart = 0, maxcodend = 0;
    float apostprob = 0;
    void add(const Transcript *t) { // AltGene::addGene (identical transcripts were united before)
        if (tx.empty()) { plus = t->plus; mincodstart = t->codingstart; maxcodend = t->codingend; }
        else { mincodstart = std::min(mincodstart, t->codingstart); maxcodend = std::max(maxcodend, t->codingend); }
        tx.push_back(t);
        apostprob += t->apostprob;
    }
    bool overlaps(const Transcript *t) const { // AltGene::overlaps: a coding exon in common, in the same reading frame
        if (t->exons.empty() || plus != t->plus || t->geneBegin() > maxcodend || t->geneEnd() < mincodstart) return false;
        for (const Transcript *a : tx)
            for (const BioState &ae : a->exons)
                for (const BioState &e : t->exons)
                    if (!(e.end < ae.begin || e.begin > ae.end) && frameCompatible(e, ae)) return true;
        return false;
    }
};
} // namespace

// What happens to the transcripts of one run between filterGenePrediction and the printing (reference
// SequenceFeatureCollection::joinGenesFromPredRuns with its one run of an ab initio prediction, src/extrinsicinfo.cc:1616-1657;
// NAMGene::doViterbiPiecewise, src/namgene.cc:627-650): --maxtracks, overlapping transcripts of one strand and reading frame
// become the alternatives of one gene, the genes are sorted by coding start, the transcripts of a gene by their mean state
// probability.  (AltGene::deleteSuboptimalTranscripts: see below; without UTR it drops nothing.)
// Where the reference's order rests on the addresses of its Transcript objects (list<Transcript*>::sort() without a comparison in
// groupTranscriptsToGenes, src/gene.cc:3196) the REVERSE order of creation is used -- the transcripts of the last sampled path
// first, those of the Viterbi path last: glibc hands the reference the Gene objects of a record's sampling loop at falling
// addresses (they are carved out of what the cleared Viterbi matrix left, src/namgene.cc:807; traced with an LD_PRELOAD counter
// of the 384-byte allocations: 30 of 30 iterations falling on soak seed 38005, one rise in 40 on an 80 kb arabidopsis record,
// mixed only in later records of a run).  It decides between transcripts of one gene with EQUAL mean state
// probability -- common with UTR states, where alternatives that differ in a UTR end share every probability -- and the rounding
// of the gene's float sum, nothing else.  (AUGX_TIE_ASC: the order of creation, tests/tie_order_probe.py measures both.)
std::vector<GeneOut> groupToGenes(const Model &m, const std::vector<Transcript> &txs) {
    std::vector<const Transcript *> list;
    for (const Transcript &t : txs) list.push_back(&t);
    const bool tieAsc = getenv("AUGX_TIE_ASC") != nullptr;
    std::stable_sort(list.begin(), list.end(), [tieAsc](const Transcript *a, const Transcript *b) {
        return a->geneBegin() != b->geneBegin() ? a->geneBegin() < b->geneBegin() : tieAsc ? a->serial < b->serial : a->serial > b->serial;
    });
    // Transcript::filterTranscriptsByMaxTracks, src/gene.cc:2533-2634
    int maxTracks = m.opt.getInt("maxtracks", -1);
    if (maxTracks >= 0) {
        const bool keepViterbi = m.opt.getBool("keep_viterbi", false);
        std::vector<const Transcript *> sorted, rest = list;
        while (!rest.empty()) { // by mean state probability, the first of equals; a Viterbi transcript first with keep_viterbi (the last)
            double best = -1.0;
            size_t at = 0;
            for (size_t i = 0; i < rest.size(); i++) {
                const double p = rest[i]->meanStateProb();
                if (p > best) { best = p; at = i; }
                if (rest[i]->viterbi && keepViterbi) { at = i; best = 1.0; }
            }
            sorted.push_back(rest[at]);
            rest.erase(rest.begin() + (long)at);
        }
        list.clear();
        for (const Transcript *t : sorted) { // kept while fewer than maxTracks kept ones cover any base of its range
            std::vector<std::pair<long, int>> ev; // sweep over the kept transcripts that meet the range
            for (const Transcript *k : list)
                if (!(k->geneEnd() < t->geneBegin() || k->geneBegin() > t->geneEnd())) {
                    ev.push_back({std::max(k->geneBegin(), t->geneBegin()), 1});
                    ev.push_back({std::min(k->geneEnd(), t->geneEnd()) + 1, -1});
                }
            std::sort(ev.begin(), ev.end());
            int cover = 0, most = 0;
            for (const auto &e : ev) { cover += e.second; most = std::max(most, cover); }
            if (most < maxTracks) list.push_back(t);
        }
    }
    // groupTranscriptsToGenes, src/gene.cc:3191-3240
    auto group = [tieAsc](std::vector<const Transcript *> &list) {
        std::stable_sort(list.begin(), list.end(), [tieAsc](const Transcript *a, const Transcript *b) { return tieAsc ? a->serial < b->serial : a->serial > b->serial; });
        std::vector<AltGeneBuild> agl;
        for (const Transcript *t : list) {
            long first = -1;
            for (size_t i = 0; i < agl.size();) {
                if (!agl[i].overlaps(t)) { i++; continue; }
                if (first < 0) { agl[i].add(t); first = (long)i; i++; }
                else { // the transcript ties another gene to the first one
                    for (const Transcript *o : agl[i].tx) agl[(size_t)first].add(o);
                    agl.erase(agl.begin() + (long)i);
                }
            }
            if (first < 0) { agl.emplace_back(); agl.back().add(t); }
        }
        return agl;
    };
    std::vector<AltGeneBuild> agl = group(list);
    // AltGene::deleteSuboptimalTranscripts, src/gene.cc:2789-2840 (src/extrinsicinfo.cc:1643-1651): without hints no transcript is
    // better supported than another; what goes is a transcript ALMOST identical to a more probable one of its gene -- the same
    // coding exons, the same UTR exons but for a transcription start / end within almost_identical_maxdiff bases -- or, with
    // --uniqueCDS, one with the coding exons of a more probable one.  Then the rest is grouped again.
    {
        const bool uniqueCDS = m.opt.getBool("uniqueCDS", false);
        const long maxdiff = m.opt.getInt("/Constant/almost_identical_maxdiff", 10);
        auto sameCDS = [](const Transcript &a, const Transcript &b) {
            if (a.exons.size() != b.exons.size()) return false;
            for (size_t i = 0; i < a.exons.size(); i++)
                if (a.exons[i].begin != b.exons[i].begin || a.exons[i].end != b.exons[i].end) return false;
            return true;
        };
        // (the end that may differ: the begin of the first exon of the UTR the transcript starts with on its strand, the end of
        // the last exon of the UTR it ends with)
        auto nearUtr = [&](const std::vector<BioState> &a, const std::vector<BioState> &b, bool firstBeginFree, bool lastEndFree) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); i++) {
                if (a[i].begin != b[i].begin && !(firstBeginFree && i == 0 && std::labs(a[i].begin - b[i].begin) <= maxdiff)) return false;
                if (a[i].end != b[i].end && !(lastEndFree && i + 1 == a.size() && std::labs(a[i].end - b[i].end) <= maxdiff)) return false;
            }
            return true;
        };
        auto almostIdentical = [&](const Transcript &a, const Transcript &b) { // Gene::almostIdenticalTo, src/gene.cc:1475-1512
            return a.plus == b.plus && sameCDS(a, b) && nearUtr(a.utr5exons, b.utr5exons, a.plus, !a.plus) &&
                   nearUtr(a.utr3exons, b.utr3exons, !a.plus, a.plus);
        };
        std::vector<const Transcript *> kept;
        bool dropped = false;
        for (const AltGeneBuild &ag : agl) {
            std::vector<char> dead(ag.tx.size(), 0);
            for (size_t i = 0; i < ag.tx.size(); i++)
                for (size_t j = 0; j < ag.tx.size(); j++) {
                    if (i == j) continue;
                    const Transcript &a = *ag.tx[i], &b = *ag.tx[j];
                    const double p1 = a.meanStateProb(), p2 = b.meanStateProb();
                    const long lengthdiff = (a.geneEnd() - a.geneBegin()) - (b.geneEnd() - b.geneBegin());
                    const bool better = p1 > p2 || (p1 == p2 && lengthdiff > 0);
                    if (better && (almostIdentical(a, b) || (uniqueCDS && sameCDS(a, b)))) dead[j] = 1;
                }
            for (size_t i = 0; i < ag.tx.size(); i++)
                if (dead[i]) dropped = true;
                else kept.push_back(ag.tx[i]);
        }
        if (dropped) agl = group(kept);
    }
    std::stable_sort(agl.begin(), agl.end(), [](const AltGeneBuild &a, const AltGeneBuild &b) { return a.mincodstart < b.mincodstart; });
    std::vector<GeneOut> genes;
    for (AltGeneBuild &ag : agl) {
        GeneOut g;
        g.plus = ag.plus;
        g.mincodstart = ag.mincodstart;
        g.maxcodend = ag.maxcodend;
        // AltGene::addGene: the sum of its transcripts' apostprob (src/gene.cc:2706); a Viterbi transcript enters with 1 when
        // nothing was sampled (src/namgene.cc:813-821)
        g.apostprob = ag.apostprob;
        g.hasProbs = true;
        // AltGene::sortTranscripts, src/gene.cc:2745-2778: by mean state probability (the running maximum is kept as a float)
        std::vector<const Transcript *> rest = ag.tx;
        while (!rest.empty()) {
            float best = -1.0f;
            size_t at = 0;
            if (ag.tx.size() > 1)
                for (size_t i = 0; i < rest.size(); i++) {
                    const double p = rest[i]->meanStateProb();
                    if (p > best) { best = (float)p; at = i; }
                }
            g.transcripts.push_back(*rest[at]);
            rest.erase(rest.begin() + (long)at);
        }
        genes.push_back(std::move(g));
    }
    return genes;
}

// the genes of a run on the reverse complement of a piece mapped back (reference reverseGeneList, src/gene.cc:3169-3185: every
// AltGene is built afresh from its mirrored transcripts -- hasProbs stays false, the gene line's score is '.')
void reverseGenes(std::vector<GeneOut> &genes, long endpos) {
    for (GeneOut &g : genes) {
        float sum = 0;
        bool first = true;
        for (Transcript &t : g.transcripts) {
            reverseTranscript(t, endpos);
            g.mincodstart = first ? t.codingstart : std::min(g.mincodstart, t.codingstart);
            g.maxcodend = first ? t.codingend : std::max(g.maxcodend, t.codingend);
            g.plus = t.plus;
            sum += t.apostprob;
            first = false;
        }
        g.apostprob = sum;
        g.hasProbs = false;
    }
}

static const char kTransTable1[] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
static int b2i(char c) {
    switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return -1; }
}
// reference getTranslation, src/gene.cc:2337-2354
std::string translateCDS(const char *cs, const char *table) {
    const char *g_transTable = (table && strlen(table) == 64) ? table : kTransTable1; // (--translation_table, Model::transTable)
    std::string result;
    while (cs[0] && cs[1] && cs[2]) {
        int a = b2i(cs[0]), b = b2i(cs[1]), c = b2i(cs[2]);
        char aa = (a < 0 || b < 0 || c < 0) ? 'X' : g_transTable[a * 16 + b * 4 + c];
        if (aa != '*') result.append(1, aa);
        else if (cs[3]) result.append("X");
        cs += 3;
    }
    return result;
}
static std::string exonicSequence(const Transcript &t, const char *seq) { // src/gene.cc:1400-1422
    std::string s;
    for (const BioState &e : t.exons) s.append(seq + e.begin, (size_t)e.length());
    for (char &c : s) c = (char)tolower((unsigned char)c);
    if (!t.plus) {
        std::string r(s.rbegin(), s.rend());
        for (char &c : r) c = c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c;
        return r;
    }
    return s;
}

static void appendf(std::string &out, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void appendf(std::string &out, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n > 0) out.append(buf, (size_t)std::min<int>(n, (int)sizeof buf - 1));
}

// reference Gene::printGFF, src/gene.cc:1998-2313 (coding genes without UTR)
static void printTranscriptGFF(std::string &out, const Transcript &t, const OutputOptions &o) {
    const char *seqname = t.seqname.c_str();
    const char *source = "AUGUSTUS";
    const char strand = t.plus ? '+' : '-';
    std::string transcript_id = t.geneid + "." + t.id;
    std::string parentstr = o.gff3 ? ("Parent=" + transcript_id) : ("transcript_id \"" + transcript_id + "\"; gene_id \"" + t.geneid + "\";");
    // score column: the posterior probability with setprecision(3), "." without sampling
    auto score = [&](const BioState &e) {
        if (!e.hasScore) return std::string(".");
        char b[32];
        snprintf(b, sizeof b, "%.3g", (double)e.apostprob);
        return std::string(b);
    };
    const std::vector<BioState> &rightUtr = t.plus ? t.utr3exons : t.utr5exons, &leftUtr = t.plus ? t.utr5exons : t.utr3exons;
    // the UTR left of the coding region (src/gene.cc:2008-2071)
    for (size_t k = 0; k < leftUtr.size(); k++) {
        const BioState &e = leftUtr[k];
        if (t.plus && k == 0 && t.complete5utr && o.print_tss)
            appendf(out, "%s\t%s\t%s\t%ld\t%ld\t.\t+\t.\t%s\n", seqname, source, o.gff3 ? "transcription_start_site" : "tss", e.begin + 1, e.begin + 1, parentstr.c_str());
        if (!t.plus && k == 0 && t.complete3utr && o.print_tts)
            appendf(out, "%s\t%s\t%s\t%ld\t%ld\t.\t-\t.\t%s\n", seqname, source, o.gff3 ? "transcription_end_site" : "tts", e.begin + 1, e.begin + 1, parentstr.c_str());
        if (o.print_utr) {
            if (e.end >= e.begin) // (a UTR exon has length 0 when the start codon comes right after the splice site)
                appendf(out, "%s\t%s\t%s\t%ld\t%ld\t%s\t%c\t.\t%s\n", seqname, source,
                        t.plus ? (o.gff3 ? "five_prime_utr" : "5'-UTR") : (o.gff3 ? "three_prime_utr" : "3'-UTR"), e.begin + 1, e.end + 1, score(e).c_str(), strand, parentstr.c_str());
        } else {
            long from = e.begin + 1, to = e.end + 1;
            if (k + 1 == leftUtr.size() && !t.exons.empty()) { // the last one runs on into the first coding exon
                to = t.exons.front().end + 1;
                if (t.exons.size() == 1 && !rightUtr.empty()) to = rightUtr.front().end + 1;
            }
            appendf(out, "%s\t%s\texon\t%ld\t%ld\t.\t%c\t.\t%s\n", seqname, source, from, to, strand, parentstr.c_str());
        }
    }
    if (!t.exons.empty()) {
        const BioState &f = t.exons.front();
        if (o.print_start && t.plus && (isInitialExon(f.type) || f.type == T_SINGLE))
            appendf(out, "%s\t%s\tstart_codon\t%ld\t%ld\t.\t+\t0\t%s\n", seqname, source, f.begin + 1, f.begin + 3, parentstr.c_str());
        if (o.print_stop && !t.plus && (f.type == T_TERMINAL || f.type == T_SINGLE || isRTerminalExon(f.type) || f.type == T_RSINGLE))
            appendf(out, "%s\t%s\tstop_codon\t%ld\t%ld\t.\t-\t0\t%s\n", seqname, source, f.begin + 1, f.begin + 3, parentstr.c_str());
    }
    auto frameCol = [&](const BioState &e) { return t.plus ? mod3(3 - (e.frame() - e.length())) : mod3(2 - e.frame()); };
    if (o.print_exonnames && !o.gff3)
        for (const BioState &e : t.exons) {
            const char *nm = (e.type == T_SINGLE || e.type == T_RSINGLE) ? "single"
                             : (isInitialExon(e.type) || e.type == T_RINITIAL) ? "initial"
                             : (e.type == T_TERMINAL || isRTerminalExon(e.type)) ? "terminal" : "internal";
            appendf(out, "%s\t%s\t%s\t%ld\t%ld\t%s\t%c\t%d\ttranscript_id \"%s.%s\"; gene_id \"%s\";\n", seqname, source, nm,
                    e.begin + 1, e.end + 1, score(e).c_str(), strand, frameCol(e), t.geneid.c_str(), t.id.c_str(), t.geneid.c_str());
        }
    if (o.print_introns)
        for (const BioState &e : t.introns)
            appendf(out, "%s\t%s\tintron\t%ld\t%ld\t%s\t%c\t.\t%s\n", seqname, source, e.begin + 1, e.end + 1, score(e).c_str(), strand, parentstr.c_str());
    for (size_t xi = 0; xi < t.exons.size(); xi++) {
        const BioState &e = t.exons[xi];
        if (o.print_cds) {
            int beginmod = 0, endmod = 0;
            if (o.stopCodonExcludedFromCDS) {
                if (e.type == T_TERMINAL || e.type == T_SINGLE) endmod = -3;
                if (isRTerminalExon(e.type) || e.type == T_RSINGLE) beginmod = 3;
            }
            if (e.begin + 1 + beginmod <= e.end + 1 + endmod) {
                appendf(out, "%s\t%s\tCDS\t%ld\t%ld\t%s\t%c\t%d\t", seqname, source, e.begin + 1 + beginmod, e.end + 1 + endmod, score(e).c_str(), strand, frameCol(e));
                if (o.gff3) appendf(out, "ID=%s.%s.cds;", t.geneid.c_str(), t.id.c_str());
                out += parentstr;
                out += "\n";
            }
        }
        // 'exon' lines: only with UTR prediction on, and when the UTRs are not printed in their own format (src/gene.cc:2166-2177)
        if (o.utr && !o.print_utr && (xi != 0 || leftUtr.empty())) {
            long from = e.begin + 1, to = e.end + 1;
            if (xi + 1 == t.exons.size() && !rightUtr.empty()) to = rightUtr.front().end + 1;
            appendf(out, "%s\t%s\texon\t%ld\t%ld\t.\t%c\t.\t%s\n", seqname, source, from, to, strand, parentstr.c_str());
        }
    }
    if (!t.exons.empty()) {
        const BioState &l = t.exons.back();
        if (o.print_stop && t.plus && (l.type == T_TERMINAL || l.type == T_SINGLE))
            appendf(out, "%s\t%s\tstop_codon\t%ld\t%ld\t.\t+\t0\t%s\n", seqname, source, l.end - 1, l.end + 1, parentstr.c_str());
        if (o.print_start && !t.plus && (isInitialExon(l.type) || l.type == T_SINGLE || l.type == T_RINITIAL || l.type == T_RSINGLE))
            appendf(out, "%s\t%s\tstart_codon\t%ld\t%ld\t.\t-\t0\t%s\n", seqname, source, l.end - 1, l.end + 1, parentstr.c_str());
    }
    // the UTR right of the coding region (src/gene.cc:2198-2250)
    for (size_t k = 0; k < rightUtr.size(); k++) {
        const BioState &e = rightUtr[k];
        if (o.print_utr) {
            if (e.end >= e.begin)
                appendf(out, "%s\t%s\t%s\t%ld\t%ld\t%s\t%c\t.\t%s\n", seqname, source,
                        t.plus ? (o.gff3 ? "three_prime_utr" : "3'-UTR") : (o.gff3 ? "five_prime_utr" : "5'-UTR"), e.begin + 1, e.end + 1, score(e).c_str(), strand, parentstr.c_str());
        } else if (k != 0)
            appendf(out, "%s\t%s\texon\t%ld\t%ld\t.\t%c\t.\t%s\n", seqname, source, e.begin + 1, e.end + 1, strand, parentstr.c_str());
        if (k + 1 == rightUtr.size() && t.plus && t.complete3utr && o.print_tts)
            appendf(out, "%s\t%s\t%s\t%ld\t%ld\t.\t+\t.\t%s\n", seqname, source, o.gff3 ? "transcription_end_site" : "tts", e.end + 1, e.end + 1, parentstr.c_str());
        if (k + 1 == rightUtr.size() && !t.plus && t.complete5utr && o.print_tss)
            appendf(out, "%s\t%s\t%s\t%ld\t%ld\t.\t-\t.\t%s\n", seqname, source, o.gff3 ? "transcription_start_site" : "tss", e.end + 1, e.end + 1, parentstr.c_str());
    }
}

void printGeneList(std::string &out, const std::vector<GeneOut> &genes, const char *seq, long seqlen, const OutputOptions &o,
                   const std::vector<std::pair<long, long>> *givenRuns) {
    // soft-masked runs of the input sequence = the hint groups of the evidence block
    std::vector<std::pair<long, long>> ownRuns;
    if (!givenRuns && o.evidence && o.softmasking && seq && !genes.empty())
        for (long i = 0; i < seqlen;) {
            if (seq[i] >= 'a' && seq[i] <= 'z') {
                long e = i;
                while (e + 1 < seqlen && seq[e + 1] >= 'a' && seq[e + 1] <= 'z') e++;
                ownRuns.push_back({i, e});
                i = e + 1;
            } else
                i++;
        }
    const std::vector<std::pair<long, long>> &rmRuns = givenRuns ? *givenRuns : ownRuns;
    for (const GeneOut &g : genes) {
        long minB = 0x7fffffffffffffffL, maxE = 0;
        for (const Transcript &t : g.transcripts) { minB = std::min(minB, t.geneBegin()); maxE = std::max(maxE, t.geneEnd()); }
        out += "# start gene " + g.id + "\n";
        // gene line: AltGene::hasProbs is true after joinGenesFromPredRuns, apostprob = sum of transcript apostprobs
        // (= 1 for the single Viterbi transcript), printed with setprecision(3)
        char score[32] = ".";
        if (g.hasProbs) snprintf(score, sizeof score, "%.3g", (double)g.apostprob);
        appendf(out, "%s\tAUGUSTUS\tgene\t%ld\t%ld\t%s\t%c\t.\t%s%s\n", g.seqname.c_str(), minB + 1 + o.offset, maxE + 1 + o.offset, score, g.plus ? '+' : '-',
                o.gff3 ? "ID=" : "", g.id.c_str());
        for (const Transcript &t : g.transcripts) {
            char tscore[32] = ".";
            if (t.hasProbs) snprintf(tscore, sizeof tscore, "%.3g", (double)t.apostprob);
            appendf(out, "%s\tAUGUSTUS\ttranscript\t%ld\t%ld\t%s\t%c\t.\t", g.seqname.c_str(), t.geneBegin() + 1 + o.offset, t.geneEnd() + 1 + o.offset, tscore, t.plus ? '+' : '-');
            if (o.gff3) out += "ID=" + g.id + "." + t.id + ";Parent=" + g.id + "\n";
            else out += g.id + "." + t.id + "\n";
            { // printed coordinates are shifted by the offset of --predictionStart (reference AnnoSequence::offset)
                Transcript tp = t;
                tp.shift(o.offset);
                printTranscriptGFF(out, tp, o);
            }
            if (seq) {
                std::string cds = exonicSequence(t, seq);
                if (o.codingseq) { // reference Gene::printCodingSeq, src/gene.cc:2315-2334
                    const int linelength = 100;
                    int cur = 21;
                    size_t offset = 0;
                    out += "# coding sequence = [";
                    while (offset < cds.size()) {
                        out += cds.substr(offset, (size_t)(linelength - cur));
                        offset += (size_t)(linelength - cur);
                        if (offset < cds.size()) { out += "\n# "; cur = 2; }
                    }
                    out += "]\n";
                }
                if (o.protein) { // reference Gene::printProteinSeq, src/gene.cc:2356-2383
                    const int linelength = 100;
                    const std::string prefix = "# protein sequence = [";
                    std::string trans = translateCDS(cds.c_str() + mod3(-t.frame), o.transTable.c_str());
                    size_t i2 = linelength - prefix.size();
                    out += prefix + trans.substr(0, i2);
                    while (i2 < trans.size()) {
                        out += "\n# " + trans.substr(i2, linelength - 2);
                        i2 += linelength - 2;
                    }
                    out += "]\n";
                }
            }
            if (o.evidence) { // reference Gene::printEvidence, src/gene.cc:2412-2445
                // the only hints of this path are the soft-masked runs (one nonexonpart hint group of source RM each,
                // reference src/extrinsicinfo.cc:1696-1724); a group that overlaps the transcript is obeyed when it lies
                // inside one intron (Gene::supportingFraction, src/gene.cc:1696-1720), else incompatible.  No hint
                // type that could support an exon or intron exists, so those counts stay 0.
                int obeyed = 0, incompatible = 0;
                for (const auto &run : rmRuns) {
                    if (run.second < t.geneBegin() || run.first > t.geneEnd()) continue;
                    bool inIntron = false;
                    for (const BioState &in : t.introns)
                        if (run.first >= in.begin && run.second <= in.end) inIntron = true;
                    for (const std::vector<BioState> *ul : {&t.utr5exons, &t.utr3exons}) // between two UTR exons (src/gene.cc:1736,1753)
                        for (size_t k = 1; k < ul->size(); k++)
                            if ((*ul)[k - 1].end + 1 <= run.first && (*ul)[k].begin - 1 >= run.second) inIntron = true;
                    (inIntron ? obeyed : incompatible)++;
                }
                out += "# Evidence for and against this transcript:\n";
                out += "# % of transcript supported by hints (any source): 0\n";
                appendf(out, "# CDS exons: 0/%zu\n# CDS introns: 0/%zu\n", t.exons.size(), t.introns.size());
                appendf(out, "# 5'UTR exons and introns: 0/%zu\n# 3'UTR exons and introns: 0/%zu\n", t.utr5exons.size() + t.utr5introns.size(), t.utr3exons.size() + t.utr3introns.size());
                appendf(out, "# hint groups fully obeyed: %d\n", obeyed);
                if (obeyed) appendf(out, "# %6s:%4d \n", "RM", obeyed);
                appendf(out, "# incompatible hint groups: %d\n", incompatible);
                if (incompatible) appendf(out, "# %6s:%4d \n", "RM", incompatible);
            }
        }
        out += "# end gene " + g.id + "\n###\n";
    }
}

} // namespace augx
