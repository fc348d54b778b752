// model.cc -- species-parameter loader producing the flat ln-tables of include/augx.h.
//
// File formats and arithmetic follow the reference readers (cited per function); the data structures
// are ours: every table is flattened, class-major, and stored as natural logs so the device only adds.
#include "model.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace augx {

static const double NEG_INF = -std::numeric_limits<double>::infinity();
static inline double lnp(double p) { return p > 0 ? std::log(p) : NEG_INF; }

// ---------------------------------------------------------------------------------------------------
// state types (reference include/types.hh:492-512, src/types.cc:157-189)
// ---------------------------------------------------------------------------------------------------
static const char *const kTypeNames[] = {
    "igenic", "single", "initial0", "initial1", "initial2", "internal0", "internal1", "internal2", "terminal",
    "lessD0", "longdss0", "equalD0", "geometric0", "longass0", "lessD1", "longdss1", "equalD1", "geometric1",
    "longass1", "lessD2", "longdss2", "equalD2", "geometric2", "longass2", "utr5single", "utr5init", "utr5intron",
    "utr5intronvar", "utr5internal", "utr5term", "utr3single", "utr3init", "utr3intron", "utr3intronvar",
    "utr3internal", "utr3term", "rsingle", "rinitial", "rinternal0", "rinternal1", "rinternal2", "rterminal0",
    "rterminal1", "rterminal2", "rlessD0", "rlongdss0", "requalD0", "rgeometric0", "rlongass0", "rlessD1",
    "rlongdss1", "requalD1", "rgeometric1", "rlongass1", "rlessD2", "rlongdss2", "requalD2", "rgeometric2",
    "rlongass2", "rutr5single", "rutr5init", "rutr5intron", "rutr5intronvar", "rutr5internal", "rutr5term",
    "rutr3single", "rutr3init", "rutr3intron", "rutr3intronvar", "rutr3internal", "rutr3term", "intron", "rintron",
    "exon", "ncsingle", "ncinit", "ncintron", "ncintronvar", "ncinternal", "ncterm", "rncsingle", "rncinit",
    "rncintron", "rncintronvar", "rncinternal", "rncterm"};
static const int kNumTypes = sizeof(kTypeNames) / sizeof(kTypeNames[0]);
static const int kReadingFrames[] = {0, 0, 0, 1, 2, 0, 1, 2, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 0, 1, 2, 0, 1, 2, 0, 0, 0, 0,
                                     0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

int stateTypeFromName(const std::string &name) {
    for (int i = 0; i < kNumTypes; i++)
        if (name == kTypeNames[i]) return i;
    return -1;
}
const char *stateTypeName(int type) { return (type >= 0 && type < kNumTypes) ? kTypeNames[type] : "unknown"; }
int winOfType(int type) { return (type >= 0 && type < kNumTypes) ? kReadingFrames[type] : 0; }
int kindOfType(int t) {
    if (t == 0) return AUGX_K_IGENIC;
    if (t == 1) return AUGX_K_SINGLE;
    if (t >= 2 && t <= 4) return AUGX_K_INITIAL;
    if (t >= 5 && t <= 7) return AUGX_K_INTERNAL;
    if (t == 8) return AUGX_K_TERMINAL;
    if (t >= 9 && t <= 23) {
        static const int m[5] = {AUGX_K_LESSD, AUGX_K_LONGDSS, AUGX_K_EQUALD, AUGX_K_GEOMETRIC, AUGX_K_LONGASS};
        return m[(t - 9) % 5];
    }
    if (t == 36) return AUGX_K_RSINGLE;
    if (t == 37) return AUGX_K_RINITIAL;
    if (t >= 38 && t <= 40) return AUGX_K_RINTERNAL;
    if (t >= 41 && t <= 43) return AUGX_K_RTERMINAL;
    if (t >= 44 && t <= 58) {
        static const int m[5] = {AUGX_K_RLESSD, AUGX_K_RLONGDSS, AUGX_K_REQUALD, AUGX_K_RGEOMETRIC, AUGX_K_RLONGASS};
        return m[(t - 44) % 5];
    }
    if (t >= 24 && t <= 35) return AUGX_K_UTR5SINGLE + (t - 24);   // utr5single .. utr3term
    if (t >= 59 && t <= 70) return AUGX_K_RUTR5SINGLE + (t - 59);  // rutr5single .. rutr3term
    return -1;
}

// ---------------------------------------------------------------------------------------------------
// Options
// ---------------------------------------------------------------------------------------------------
void Options::readFile(const std::string &path, const std::string &configPath) {
    std::ifstream in(path.c_str());
    if (!in) throw ConfigError("Could not open the this file: " + path);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string name, value;
        ls >> name >> value;
        if (name == "include")
            readFile(configPath + value, configPath);
        else if (!name.empty() && !value.empty() && name[0] != '#')
            kv[name] = value;
    }
}
const std::string &Options::get(const std::string &k) const {
    auto it = kv.find(k);
    if (it == kv.end()) throw ConfigError("Properties::getProperty(): no such key \"" + k + "\".");
    return it->second;
}
int Options::getInt(const std::string &k) const { return atoi(get(k).c_str()); }
double Options::getDouble(const std::string &k) const { return strtod(get(k).c_str(), nullptr); }
bool Options::getBool(const std::string &k) const {
    const std::string &v = get(k);
    if (v == "true" || v == "1" || v == "on" || v == "yes" || v == "True" || v == "TRUE") return true;
    if (v == "false" || v == "0" || v == "off" || v == "no" || v == "False" || v == "FALSE") return false;
    throw ConfigError("Properties::getBoolProperty(): invalid boolean value for \"" + k + "\": " + v);
}

// ---------------------------------------------------------------------------------------------------
// token reader over a whole .pbl file held in memory.  Equivalent of the reference's stream
// manipulators `comment` and `goto_line_after` (include/projectio.hh:27-82) and Seq2Int::read
// (include/geneticcode.hh:203-219).
// ---------------------------------------------------------------------------------------------------
class PblReader {
public:
    explicit PblReader(const std::string &path) : path_(path) {
        std::ifstream in(path.c_str(), std::ios::binary);
        if (!in) throw ConfigError("Couldn't open file " + path);
        std::stringstream ss;
        ss << in.rdbuf();
        buf_ = ss.str();
    }
    bool ok() const { return ok_; }
    size_t tell() const { return pos_; }
    void seek(size_t p) { pos_ = p; ok_ = true; }
    void skipWs() {
        while (pos_ < buf_.size() && isspace((unsigned char)buf_[pos_])) pos_++;
    }
    // skip whitespace and all lines starting with '#'
    void comment() {
        for (;;) {
            skipWs();
            if (pos_ < buf_.size() && buf_[pos_] == '#')
                skipLine();
            else
                return;
        }
    }
    int peek() {
        return pos_ < buf_.size() ? (unsigned char)buf_[pos_] : -1;
    }
    // position the cursor after the first line (from here on) that begins with `key`
    bool gotoLineAfter(const char *key) {
        size_t klen = strlen(key);
        while (pos_ < buf_.size()) {
            skipWs();
            if (pos_ >= buf_.size()) break;
            bool match = buf_.compare(pos_, klen, key) == 0;
            skipLine();
            if (match) return true;
        }
        ok_ = false;
        return false;
    }
    void need(const char *key) {
        if (!gotoLineAfter(key)) throw ConfigError("Error reading file " + path_ + ": section " + key + " not found");
    }
    double readDouble() {
        skipWs();
        const char *s = buf_.c_str() + pos_;
        char *e = nullptr;
        double v = strtod(s, &e);
        if (e == s) throw ConfigError("Error reading file " + path_ + ": number expected near offset " + std::to_string(pos_));
        pos_ += (size_t)(e - s);
        return v;
    }
    int readInt() {
        skipWs();
        const char *s = buf_.c_str() + pos_;
        char *e = nullptr;
        long v = strtol(s, &e, 10);
        if (e == s) throw ConfigError("Error reading file " + path_ + ": integer expected near offset " + std::to_string(pos_));
        pos_ += (size_t)(e - s);
        return (int)v;
    }
    // reads `size` characters as a base-4 pattern number
    int readPattern(int size) {
        int r = 0;
        for (int i = 0; i < size; i++) {
            if (pos_ >= buf_.size()) throw ConfigError("Error reading file " + path_ + ": truncated pattern");
            char c = (char)toupper((unsigned char)buf_[pos_++]);
            int b = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
            if (b < 0) throw ConfigError("Error reading file " + path_ + ": bad pattern character near offset " + std::to_string(pos_));
            r = (r << 2) | b;
        }
        return r;
    }
    std::string readWord() {
        skipWs();
        size_t b = pos_;
        while (pos_ < buf_.size() && !isspace((unsigned char)buf_[pos_])) pos_++;
        return buf_.substr(b, pos_ - b);
    }
    const std::string &path() const { return path_; }

private:
    void skipLine() {
        while (pos_ < buf_.size() && buf_[pos_] != '\n') pos_++;
        if (pos_ < buf_.size()) pos_++;
    }
    std::string path_, buf_;
    size_t pos_ = 0;
    bool ok_ = true;
};

struct Motif {
    int n = 0, k = 0;
    std::vector<double> p; // [n][4^(k+1)]
};
// reference Motif::read, src/motif.cc:388-411
static Motif readMotif(PblReader &r) {
    Motif m;
    r.comment();
    m.n = r.readInt();
    r.comment();
    m.k = r.readInt();
    int sz = ipow4(m.k + 1);
    m.p.assign((size_t)m.n * sz, 0.0);
    for (int i = 0; i < m.n; i++) {
        r.comment();
        (void)r.readInt();
        for (int j = 0; j < sz; j++) m.p[(size_t)i * sz + j] = r.readDouble();
    }
    return m;
}
struct Bins {
    int nbins = 0;
    std::vector<double> bb, av;
    // reference BinnedMMGroup::getIndex, src/merkmal.cc:155-168
    int index(double p) const {
        int a = 0, b = nbins - 1;
        while (a < b) {
            int m = (a + b) / 2;
            if (p < bb[m]) b = m; else a = m + 1;
        }
        return a;
    }
};
// reference BinnedMMGroup::read, src/merkmal.cc:186-197
static Bins readBins(PblReader &r) {
    Bins b;
    r.comment();
    b.nbins = r.readInt();
    if (b.nbins < 1) throw ConfigError("BinnedMMGroup:: less than one bin.");
    b.av.resize(b.nbins);
    b.bb.resize(b.nbins - 1);
    r.comment();
    b.av[0] = r.readDouble();
    for (int i = 1; i < b.nbins; i++) {
        b.bb[i - 1] = r.readDouble();
        b.av[i] = r.readDouble();
    }
    return b;
}

// tail of a length distribution; reference ExonModel::fillTailsOfLengthDistributions, src/exonmodel.cc:839-866.
// The reference multiplies in extended-range LLDouble; plain doubles are identical until the value leaves
// the normal range, from where we continue in log space.
static void fillTail(std::vector<double> &dist, std::vector<double> &lnd, int D, int maxLen, double numHuge, double num) {
    double a = dist[D];
    double p = 1.0 - a / ((numHuge + 1.0) / (num + 1.0));
    for (int i = 0; i <= D; i++) lnd[i] = lnp(dist[i]);
    bool inlog = false;
    for (int k = D + 1; k <= maxLen; k++) {
        if (!inlog) {
            dist[k] = p * dist[k - 1];
            lnd[k] = lnp(dist[k]);
            if (dist[k] != 0 && std::fabs(dist[k]) < 1e-290) inlog = true;
        } else {
            dist[k] = 0;
            lnd[k] = lnd[k - 1] + lnp(p);
        }
    }
}

void Model::bindPointers() {
    t.ln_trans = ln_trans.data();
    t.ig_emi = ig_emi.data();
    t.ig_short = ig_short.data();
    t.in_emi = in_emi.data();
    t.ex_emi = ex_emi.data();
    t.ex_init = ex_init.data();
    t.ex_et = ex_et.data();
    t.ex_pls = ex_pls.data();
    t.tis_motif = tis_motif.data();
    t.ass_motif = ass_motif.data();
    t.tis_bin_bounds = tis_bin_bounds.empty() ? nullptr : tis_bin_bounds.data();
    t.tis_bin_ln = tis_bin_ln.empty() ? nullptr : tis_bin_ln.data();
    t.ass_pat = ass_pat.data();
    t.dss_pat = dss_pat.data();
    t.len_intron = len_intron.data();
    t.len_single = len_single.data();
    t.len_initial = len_initial.data();
    t.len_internal = len_internal.data();
    t.len_terminal = len_terminal.data();
    auto ptr = [](const std::vector<double> &v) { return v.empty() ? nullptr : v.data(); };
    t.utr5init_emi = ptr(utr5init_emi); t.utr5_emi = ptr(utr5_emi); t.utr3_emi = ptr(utr3_emi); t.tssup_emi = ptr(tssup_emi);
    t.tss_motif = ptr(tss_motif); t.tsstata_motif = ptr(tsstata_motif); t.tata_motif = ptr(tata_motif); t.tts_motif = ptr(tts_motif);
    t.aataaa = ptr(aataaa);
    t.len5_single = ptr(len5_single); t.len5_initial = ptr(len5_initial); t.len5_internal = ptr(len5_internal);
    t.len5_terminal = ptr(len5_terminal); t.len3_single = ptr(len3_single); t.len3_initial = ptr(len3_initial);
    t.len3_internal = ptr(len3_internal); t.len3_terminal = ptr(len3_terminal);
    t.tail5_single = ptr(tail5_single); t.tail3_single = ptr(tail3_single);
}

void Model::load(const std::string &cfgPathIn, const std::string &sp,
                 const std::vector<std::pair<std::string, std::string>> &cmdline) {
    configPath = cfgPathIn;
    if (!configPath.empty() && configPath.back() != '/') configPath += '/';
    species = sp;
    speciesDir = "species/" + species + "/";
    const std::string full = configPath + speciesDir;

    // ---- options: species cfg, then command line, then the states cfg (reference src/properties.cc:193-394)
    try {
        opt.readFile(full + species + "_parameters.cfg", configPath);
    } catch (ConfigError &) {
        throw ConfigError("Species-specific configuration files not found in " + configPath +
                          "species/. Type \"augustus --species=help\" to see available species.");
    }
    for (auto &kv : cmdline) opt.set(kv.first, kv.second);
    opt.set("species", species);

    bool singleStrand = opt.getBool("singlestrand", false);
    std::string genemodel = opt.get("genemodel", "partial");
    if (genemodel != "partial" && genemodel != "complete" && genemodel != "atleastone" && genemodel != "exactlyone" &&
        genemodel != "intronless" && genemodel != "bacterium")
        throw ConfigError("Unknown value for parameter genemodel: " + genemodel);
    bool utr = false;
    if (opt.has("UTR")) {
        try { utr = opt.getBool("UTR"); } catch (ConfigError &) {
            throw ConfigError("Unknown option for parameter UTR. Use --UTR=on or --UTR=off.");
        }
    }
    bool nc = opt.getBool("nc", false);
    if (singleStrand && genemodel != "partial" && genemodel != "complete")
        throw UnsupportedError("--singlestrand=true with --genemodel=" + genemodel + " is outside the MI355X hot path (partial|complete only)");
    if (utr && (singleStrand || !(genemodel == "partial" || genemodel == "complete")))
        throw ConfigError("UTR only implemented with shadow and partial or complete."); // (reference src/properties.cc:363-365)
    if (nc) throw UnsupportedError("--nc=on is outside the MI355X hot path");
    if (genemodel == "bacterium")
        throw UnsupportedError("--genemodel=bacterium (overlapping genes, Constant::overlapmode) is outside the MI355X hot path");
    if (opt.has("hintsfile")) throw UnsupportedError("--hintsfile (extrinsic evidence) is outside the MI355X ab-initio hot path");
    if (opt.has("proteinprofile")) throw UnsupportedError("--proteinprofile (PPX) is outside the MI355X ab-initio hot path");
    if (opt.getBool("mea", false)) throw UnsupportedError("--mea=1 is outside the MI355X ab-initio hot path");
    // options that would change the prediction or the output and are not implemented here: fail loudly instead of
    // silently printing something else than the reference
    if (!opt.getBool("contentmodels", true)) throw UnsupportedError("--contentmodels=false is outside the MI355X hot path");
    for (const char *o2 : {"emiprobs", "exoncands", "printHints", "printSampled", "printOEs", "printGeneRangesBED", "printGeneRangesGFF",
                           "print_blocks", "printMEA"})
        if (opt.getBool(o2, false)) throw UnsupportedError(std::string("--") + o2 + " (extra output) is not implemented on the MI355X path");
    for (const char *o2 : {"speciesfilenames", "alnfile", "treefile", "dbaccess", "dbhints", "referenceFile", "refSpecies", "optCfgFile",
                           "codonAlignmentFile", "trainFeatureFile"})
        if (opt.has(o2)) throw UnsupportedError(std::string("--") + o2 + " (comparative / training mode) is outside the MI355X ab-initio hot path");
    std::string strandName = singleStrand ? "singlestrand" : "shadow";
    std::string transFile = "trans_" + strandName + "_" + genemodel + (utr ? "_utr" : "") + ".pbl";
    opt.set("/NAMGene/TransFile", transFile);
    // (reference src/properties.cc:376-392: two intergenic states for atleastone / exactlyone)
    const bool twoIgenic = genemodel == "atleastone" || genemodel == "exactlyone";
    opt.readFile(configPath + "model/states_" + strandName + (twoIgenic ? "_2igenic" : genemodel == "intronless" ? "_intronless" : utr ? "_utr" : "") + ".cfg", configPath);
    t.utr = utr ? 1 : 0;

    // ---- constants (reference Constant::init, src/types.cc:208-450; defaults src/types.cc:20-116)
    t.W = opt.getInt("/Constant/trans_init_window", 12);
    t.U = opt.getInt("/Constant/ass_upwindow_size", 20);
    t.As = opt.getInt("/Constant/ass_start", 2);
    t.Ae = opt.getInt("/Constant/ass_end", 2);
    t.Ds = opt.getInt("/Constant/dss_start", 2);
    t.De = opt.getInt("/Constant/dss_end", 5);
    t.Li = opt.getInt("/Constant/init_coding_len", 16);
    t.Le = opt.getInt("/Constant/intterm_coding_len", 5);
    t.n_classes = opt.getInt("/Constant/decomp_num_steps", 1);
    int decompAt = opt.getInt("/Constant/decomp_num_at", 1), decompGc = opt.getInt("/Constant/decomp_num_gc", 1);
    if (decompAt != 1 || decompGc != 1) throw UnsupportedError("decomp_num_at/decomp_num_gc != 1 not supported");
    if (t.n_classes < 1 || t.n_classes > AUGX_MAX_CLASSES) throw ConfigError("decomp_num_steps out of range");
    t.min_coding_len = opt.getInt("/Constant/min_coding_len", 102);
    t.max_exon_len = opt.getInt("/ExonModel/maxexonlength", 12000);
    t.min_exon_len = opt.getInt("/ExonModel/minexonlength", 1);
    t.d = opt.getInt("/IntronModel/d", 0);
    t.tis_mem = opt.getInt("/ExonModel/tis_motif_memory", 3);
    t.gc_win = opt.getInt("GCwinsize", 10000);
    // soft-masking: lower-case runs are nonexonpart hints of source RM (reference src/extrinsicinfo.cc:1696-1724); their bonus
    // comes from the extrinsic configuration.  Only the shipped shape of that file is supported: every bonus / malus 1
    // but the RM grade quotient of nonexonpart.
    t.softmasking = opt.getBool("softmasking", true) ? 1 : 0;
    t.ln_soft_bonus = 0.0;
    if (t.softmasking) {
        if (opt.has("extrinsicCfgFile")) throw UnsupportedError("--extrinsicCfgFile is outside the MI355X ab-initio hot path");
        const std::string ecfg = configPath + "extrinsic/extrinsic.cfg";
        std::ifstream ef(ecfg.c_str());
        if (!ef) throw ConfigError("Could not find the extrinsic config file " + ecfg + " (needed with --softmasking=1).");
        std::string line;
        bool general = false, found = false;
        while (std::getline(ef, line)) {
            std::istringstream ls(line);
            std::vector<std::string> w;
            for (std::string x; ls >> x;) w.push_back(x);
            if (w.empty() || w[0][0] == '#') continue;
            if (w[0][0] == '[') { general = w[0] == "[GENERAL]"; continue; }
            if (!general || w.size() < 3) continue;
            const double bonus = atof(w[1].c_str()), malus = atof(w[2].c_str());
            double rm = 1.0;
            for (size_t i = 3; i + 2 < w.size(); i++)
                if (w[i] == "RM") { // source columns: name, number of score classes, boundaries, grade quotients
                    if (atoi(w[i + 1].c_str()) != 1) throw UnsupportedError("extrinsic.cfg: more than one score class for source RM");
                    rm = atof(w[i + 2].c_str());
                }
            if (bonus != 1.0 || malus != 1.0 || (w[0] != "nonexonpart" && rm != 1.0))
                throw UnsupportedError("extrinsic.cfg with bonus/malus values other than the soft-masking nonexonpart bonus is outside the MI355X ab-initio hot path");
            if (w[0] == "nonexonpart") { t.ln_soft_bonus = std::log(bonus * rm); found = true; }
        }
        if (!found) throw ConfigError("extrinsic.cfg: no nonexonpart line in [GENERAL]");
    }
    double probNinCoding = opt.getDouble("/Constant/probNinCoding", 0.23);
    double opal = opt.getDouble("/Constant/opalprob", 0.333), amber = opt.getDouble("/Constant/amberprob", 0.333),
           ochre = opt.getDouble("/Constant/ochreprob", 0.333);
    double gcMin = opt.getDouble("/Constant/gc_range_min", 0.32), gcMax = opt.getDouble("/Constant/gc_range_max", 0.73);
    // (without an intron model -- genemodel=intronless -- the intergenic model keeps its own content model: reference
    //  IGenicModel::updateToLocalGC, src/igenicmodel.cc:71-79, IntronModel::GCemiprobs == NULL)
    bool tie = opt.getBool("tieIgenicIntron", true) && genemodel != "intronless";
    t.dss_gc = opt.getBool("/IntronModel/allow_dss_consensus_gc", false) ? 1 : 0; // (donor sites gc as well as gt: Constant::dss_gc_allowed, src/types.cc:427)
    t.ln_quarter = std::log(0.25);
    t.ln_n_coding = std::log(probNinCoding);
    t.ln4 = std::log(4.0);
    t.ln_stop_ochre = lnp(ochre);
    t.ln_stop_amber = lnp(amber);
    t.ln_stop_opal = lnp(opal);
    int kEx = opt.getInt("/ExonModel/k", 4), kIn = opt.getInt("/IntronModel/k", 4), kIg = opt.getInt("/IGenicModel/k", 4);
    if (kEx != kIg) throw UnsupportedError("exon/igenic Markov orders differ; not supported");
    t.k = kEx;
    t.k_in = kIn; // (the intron file's own `k` line decides, below)
    t.utr_k = kEx; // (loadUtr: the order the UTR parameter file states)
    const int k = t.k, NP = ipow4(k + 1), C = t.n_classes;
    if (t.d < 2 + t.De + t.U + t.As + 2)
        throw ConfigError("Inconsistent intron length parameters. Please increase /IntronModel/d or decrease /IntronModel/ass_motif_memory.");

    // ---- states (reference NAMGene::createStateModels src/namgene.cc:1537-1548 and the model ctors,
    //      src/exonmodel.cc:231-250)
    t.S = opt.getInt("/NAMGene/statecount");
    if (t.S > AUGX_MAX_STATES) throw ConfigError("too many states");
    t.synch_state = opt.getInt("/NAMGene/SynchState", 0);
    {
        int ne = 0, ni = 0, ng = 0, nu = 0;
        char key[64];
        for (int i = 0; i < t.S; i++) {
            snprintf(key, sizeof key, "/NAMGene/state%02d", i);
            std::string mdl = opt.get(key);
            std::string typeName;
            if (mdl == "exonmodel") { snprintf(key, sizeof key, "/ExonModel/type%02d", ne++); typeName = opt.get(key); }
            else if (mdl == "intronmodel") { snprintf(key, sizeof key, "/IntronModel/type%02d", ni++); typeName = opt.get(key); }
            else if (mdl == "igenicmodel") { snprintf(key, sizeof key, "/IGenicModel/type%02d", ng++); typeName = opt.get(key, "igenic"); }
            else if (mdl == "utrmodel" && utr) { snprintf(key, sizeof key, "/UtrModel/type%02d", nu++); typeName = opt.get(key); }
            else throw UnsupportedError("state model \"" + mdl + "\" is outside the MI355X hot path");
            int ty = stateTypeFromName(typeName);
            int kind = kindOfType(ty);
            if (ty < 0 || kind < 0) throw UnsupportedError("state type \"" + typeName + "\" not supported");
            t.state_type[i] = ty;
            t.state_kind[i] = kind;
            t.state_win[i] = winOfType(ty);
        }
    }

    // ---- transitions (reference NAMGene::readTransAndInitProbs, src/namgene.cc:1318-1392)
    std::vector<double> trans((size_t)t.S * t.S, 0.0), initP(t.S, 0.0), termP(t.S, 0.0);
    {
        std::string fname = full + species + "_" + transFile;
        std::ifstream probe(fname.c_str());
        if (probe && probe.peek() != EOF) {
            speciesSpecificTrans = true;
        } else {
            fname = configPath + "model/" + transFile;
        }
        transFileUsed = fname;
        PblReader r(fname);
        r.comment();
        int count = r.readInt();
        if (count != t.S) throw ConfigError("Incorrect state count in transition file!!!");
        r.need("[Initial]");
        r.comment();
        int ns = r.readInt();
        for (int i = 0; i < ns; i++) { r.comment(); int s = r.readInt(); initP.at(s) = r.readDouble(); }
        r.need("[Terminal]");
        r.comment();
        int ne = r.readInt();
        for (int i = 0; i < ne; i++) { r.comment(); int s = r.readInt(); termP.at(s) = r.readDouble(); }
        r.need("[Transition]");
        for (;;) {
            r.comment();
            if (r.peek() < 0) break;
            int i = r.readInt(), j = r.readInt();
            if (i >= t.S || j >= t.S) throw ConfigError("State number in transition file too large!");
            trans[(size_t)i * t.S + j] = r.readDouble();
        }
    }
    for (int i = 0; i < t.S; i++) { t.ln_init[i] = lnp(initP[i]); t.ln_term[i] = lnp(termP[i]); }
    // reachability closure (reference NAMGene::computeReachableStates, src/namgene.cc:1508-1535)
    {
        std::vector<int> reach(t.S);
        for (int i = 0; i < t.S; i++) reach[i] = initP[i] > 0.0;
        for (bool ex = true; ex;) {
            ex = false;
            for (int i = 0; i < t.S; i++)
                for (int j = 0; j < t.S; j++)
                    if (reach[i] && !reach[j] && trans[(size_t)i * t.S + j] > 0.0) reach[j] = ex = true;
        }
        for (int i = 0; i < t.S; i++) t.reachable[i] = reach[i];
    }

    // ---- exon parameters (reference ExonModel::readAllParameters, src/exonmodel.cc:604-792)
    for (int c = 0; c < 64; c++) t.ln_startcodon[c] = NEG_INF;
    {
        // translation table 1: {a,c,t}tg may start, only atg has probability 1 by default
        // (reference src/geneticcode.cc:15-19, GeneticCode::chooseTranslationTable :172-196)
        // --translation_table: the amino acid of every codon and which codons may start (the published NCBI genetic codes the
        // reference knows, codons in the order aaa, aac, aag, aat, aca ... ttt; an unknown number means table 1, src/geneticcode.cc:146-148)
        static const struct { int n; const char *aa, *start; } kCodes[] = {
            {1, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF", "--------------M---------------M-------------------------------M-"},
            {2, "KNKNTTTT*S*SMIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "------------MMMM------------------------------M-----------------"},
            {3, "KNKNTTTTRSRSMIMIQHQHPPPPRRRRTTTTEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "------------M-M-------------------------------------------------"},
            {4, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "------------MMMM--------------M---------------M-------------M-M-"},
            {5, "KNKNTTTTSSSSMIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "------------MMMM------------------------------M---------------M-"},
            {6, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVQYQYSSSS*CWCLFLF", "--------------M-------------------------------------------------"},
            {9, "NNKNTTTTSSSSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "--------------M-------------------------------M-----------------"},
            {10, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSCCWCLFLF", "--------------M-------------------------------------------------"},
            {11, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF", "------------MMMM--------------M---------------M---------------M-"},
            {12, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLSLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF", "--------------M---------------M---------------------------------"},
            {13, "KNKNTTTTGSGSMIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "------------M-M-------------------------------M---------------M-"},
            {14, "NNKNTTTTSSSSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVYY*YSSSSWCWCLFLF", "--------------M-------------------------------------------------"},
            {15, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*YQYSSSS*CWCLFLF", "--------------M-------------------------------------------------"},
            {16, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*YLYSSSS*CWCLFLF", "--------------M-------------------------------------------------"},
            {21, "NNKNTTTTSSSSMIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "--------------M-------------------------------M-----------------"},
            {22, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*YLY*SSS*CWCLFLF", "--------------M-------------------------------------------------"},
            {23, "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWC*FLF", "--------------MM------------------------------M-----------------"},
            {24, "KNKNTTTTSSKSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSSWCWCLFLF", "--------------M---------------M---------------M---------------M-"}};
        int tt = opt.has("translation_table") ? opt.getInt("translation_table", 1) : 1, ti = 0;
        for (int i = 0; i < (int)(sizeof kCodes / sizeof kCodes[0]); i++) if (kCodes[i].n == tt) ti = i;
        if (kCodes[ti].n != tt) tt = 1;
        transTable = kCodes[ti].aa;
        {   // "# Warning: Using nonstandard genetic code: ..." (chooseTranslationTable :157-165), one line per codon read differently
            static const char *kSym = "GDERKNQSTAVLIFYWHMCP";
            static const char *kName[20] = {"GLYCINE", "ASPARTIC ACID", "GLUTAMIC ACID", "ARGININE", "LYSINE", "ASPARAGINE", "GLUTAMINE", "SERINE", "THREONINE", "ALANINE",
                                            "VALINE", "LEUCINE", "ISOLEUCINE", "PHENYLALANINE", "TYROSINE", "TRYPTOPHAN", "HISTIDINE", "METHIONINE", "CYSTEINE", "PROLINE"};
            auto nameOf = [&](char a) -> std::string { if (a == '*') return "STOP"; const char *q = strchr(kSym, a); return q ? kName[q - kSym] : "?"; };
            codeWarnings.clear();
            for (int c = 0; c < 64; c++)
                if (transTable[(size_t)c] != kCodes[0].aa[c]) {
                    const char cod[4] = {"ACGT"[c >> 4], "ACGT"[(c >> 2) & 3], "ACGT"[c & 3], 0}; // (Seq2Int::INV prints capitals)
                    codeWarnings += std::string("# Warning: Using nonstandard genetic code: ") + cod + " coding for " + nameOf(transTable[(size_t)c]) + " instead of " +
                                    nameOf(kCodes[0].aa[c]) + ".\n";
                }
        }
        t.stop_mask = 0;
        for (int c = 0; c < 64; c++)
            if (transTable[(size_t)c] == '*') {
                if (c == 48) t.stop_mask |= 1;       // taa
                else if (c == 50) t.stop_mask |= 2;  // tag
                else if (c == 56) t.stop_mask |= 4;  // tga
                else throw UnsupportedError("translation_table " + std::to_string(tt) + " has a stop codon other than taa / tag / tga (the reference stops with an internal error at the first gene that ends in it); not supported");
            }
        double startProb[64] = {0};
        bool isStart[64] = {false};
        for (int c = 0; c < 64; c++) isStart[c] = kCodes[ti].start[c] != '-'; // (table 1: atg, ctg, ttg)
        t.start_mask = 0;
        for (int c = 0; c < 64; c++) if (isStart[c]) t.start_mask |= 1ull << c;
        startProb[14] = 1.0; // (GeneticCode::start_codon_probs: atg 1 until the species' file says otherwise, src/geneticcode.cc:18)
        PblReader r(full + opt.get("/ExonModel/infile"));
        size_t sp0 = r.tell();
        if (r.gotoLineAfter("[STARTCODONS]")) { // reference GeneticCode::readStart, src/geneticcode.cc:280-306
            r.comment();
            int n = r.readInt();
            r.comment();
            for (int i = 0; i < n; i++) {
                std::string cod = r.readWord();
                double p = r.readDouble();
                if (p < 0.0) throw ConfigError("Start codon probability is negative.");
                if (cod.size() != 3) throw ConfigError("Invalid start codon " + cod);
                int pn = 0;
                for (char ch : cod) {
                    ch = (char)tolower(ch);
                    int b = ch == 'a' ? 0 : ch == 'c' ? 1 : ch == 'g' ? 2 : ch == 't' ? 3 : -1;
                    if (b < 0) throw ConfigError("Invalid start codon " + cod);
                    pn = pn * 4 + b;
                }
                if (isStart[pn]) startProb[pn] = p;
                else stderrNotes += cod + " is not a start codon in the chosen translation table " + std::to_string(tt) + ". Ignoring it.\n"; // (GeneticCode::readStart :297-299)
            }
        } else
            r.seek(sp0);
        for (int c = 0; c < 64; c++) t.ln_startcodon[c] = (isStart[c] && startProb[c] > 0) ? std::log(startProb[c]) : NEG_INF;

        r.need("[LENGTH]");
        r.comment(); int exonLenD = r.readInt();
        r.comment(); (void)r.readDouble();
        r.comment(); (void)r.readInt();
        r.comment(); double numSingle = r.readInt(), numInitial = r.readInt(), numInternal = r.readInt(), numTerminal = r.readInt();
        r.comment(); double hugeSingle = r.readInt(), hugeInitial = r.readInt(), hugeInternal = r.readInt(), hugeTerminal = r.readInt();
        r.comment();
        const int ML = t.max_exon_len;
        if (exonLenD > ML) throw ConfigError("exonlengthD exceeds maxexonlength");
        std::vector<double> dS(ML + 1, 0.0), dI(ML + 1, 0.0), dN(ML + 1, 0.0), dT(ML + 1, 0.0);
        for (int i = 0; i <= exonLenD; i++) {
            (void)r.readInt();
            dS[i] = r.readDouble() / 1000;
            dI[i] = r.readDouble() / 1000;
            dN[i] = r.readDouble() / 1000;
            dT[i] = r.readDouble() / 1000;
        }
        for (int i = 0; i < t.min_coding_len && i <= ML; i++) dS[i] = 0;
        std::vector<double> lS(ML + 1), lI(ML + 1), lN(ML + 1), lT(ML + 1);
        fillTail(dS, lS, exonLenD, ML, hugeSingle, numSingle);
        fillTail(dI, lI, exonLenD, ML, hugeInitial, numInitial);
        fillTail(dN, lN, exonLenD, ML, hugeInternal, numInternal);
        fillTail(dT, lT, exonLenD, ML, hugeTerminal, numTerminal);
        // the reference multiplies 3*lenDist[len] at the use site (src/exonmodel.cc:1726-1754)
        auto fold3 = [&](std::vector<double> &out, const std::vector<double> &d, const std::vector<double> &l) {
            out.resize(ML + 1);
            for (int i = 0; i <= ML; i++) out[i] = d[i] != 0 ? std::log(3 * d[i]) : (l[i] == NEG_INF ? NEG_INF : std::log(3.0) + l[i]);
        };
        fold3(len_single, dS, lS);
        fold3(len_initial, dI, lI);
        fold3(len_internal, dN, lN);
        fold3(len_terminal, dT, lT);

        ex_emi.assign((size_t)C * 3 * NP, NEG_INF);
        ex_init.assign((size_t)C * 3 * NP, NEG_INF);
        ex_et.assign((size_t)C * 3 * NP, NEG_INF);
        ex_pls.assign((size_t)C * (k + 1) * 3 * NP, NEG_INF);
        t.tis_nbins = 0;
        std::vector<Bins> tisBins(C);
        for (int c = 0; c < C; c++) {
            char tag[16];
            snprintf(tag, sizeof tag, "[%d]", c + 1);
            r.need(tag);
            r.need("[P_ls]");
            r.comment();
            std::vector<double> plsK[3]; // order-k pattern probabilities (old-format files derive the emissions from them)
            for (int f = 0; f < 3; f++) plsK[f].assign(NP, 0.0);
            for (int l = 0; l <= k; l++) {
                r.comment();
                (void)r.readInt();
                int size = ipow4(l + 1);
                for (int j = 0; j < size; j++) {
                    r.comment();
                    int pn = r.readPattern(l + 1);
                    if (pn != j) throw ConfigError("ExonModel::readProbabilities: Error reading file " + r.path() + " at P_ls");
                    for (int f = 0; f < 3; f++) {
                        const double pr = r.readDouble();
                        ex_pls[(((size_t)c * (k + 1) + l) * 3 + f) * NP + j] = lnp(pr);
                        if (l == k) plsK[f][j] = pr;
                    }
                }
            }
            r.need("[TRANSINIT]");
            Motif m = readMotif(r);
            if (c == 0) {
                t.tis_n = m.n;
                t.tis_k = m.k;
                tis_motif.assign((size_t)C * m.n * ipow4(m.k + 1), NEG_INF);
            } else if (m.n != t.tis_n || m.k != t.tis_k)
                throw ConfigError("TRANSINIT motif shape differs between GC classes");
            for (size_t i = 0; i < m.p.size(); i++) tis_motif[(size_t)c * m.p.size() + i] = lnp(m.p[i]);
            size_t sp1 = r.tell();
            if (r.gotoLineAfter("[TRANSINITBIN]")) {
                tisBins[c] = readBins(r);
                t.tis_nbins = tisBins[c].nbins;
            } else
                r.seek(sp1);
            size_t sp2 = r.tell();
            if (!r.gotoLineAfter("[EMISSION]")) {
                // old parameter files: the emissions are the order-k pattern probabilities normalised over the last base
                // (reference StateModel::computeEmiFromPat, src/statemodel.cc:175-190; src/exonmodel.cc:730-737)
                r.seek(sp2);
                for (int f = 0; f < 3; f++)
                    for (int i = 0; i < NP; i += 4) {
                        const double sum = plsK[f][i] + plsK[f][i + 1] + plsK[f][i + 2] + plsK[f][i + 3];
                        for (int nuk = 0; nuk < 4; nuk++)
                            ex_emi[((size_t)c * 3 + f) * NP + i + nuk] = lnp(k > 0 ? plsK[f][i + nuk] / sum : plsK[f][i + nuk]);
                    }
            } else {
                r.comment(); (void)r.readInt();
                r.comment(); (void)r.readInt();
                r.comment(); (void)r.readDouble();
                for (int i = 0; i < NP; i++) {
                    r.comment();
                    int pn = r.readPattern(k + 1);
                    if (pn != i) throw ConfigError("ExonModel::readProbabilities: Error reading file " + r.path() + " at EMISSION");
                    for (int f = 0; f < 3; f++) ex_emi[((size_t)c * 3 + f) * NP + i] = lnp(r.readDouble());
                }
            }
            auto readSparse = [&](const char *sec, std::vector<double> &dst) {
                r.need(sec);
                r.comment(); (void)r.readInt();
                r.comment(); int kk = r.readInt();
                if (kk != k) throw ConfigError("ExonModel::readProbabilities: Mismatch in order of exon Markov chain.");
                r.comment(); (void)r.readDouble();
                // entries not listed keep probability 0 (reference resizes the vectors to `size`, :761-769)
                for (int f = 0; f < 3; f++)
                    for (int i = 0; i < NP; i++) dst[((size_t)c * 3 + f) * NP + i] = NEG_INF;
                for (;;) {
                    r.comment();
                    if (r.peek() < 0 || r.peek() == '[') break;
                    int pn = r.readPattern(k + 1);
                    for (int f = 0; f < 3; f++) dst[((size_t)c * 3 + f) * NP + pn] = lnp(r.readDouble());
                }
            };
            readSparse("[INITEMISSION]", ex_init);
            readSparse("[ETEMISSION]", ex_et);
        }
        if (t.tis_nbins > 0) {
            tis_bin_bounds.assign((size_t)C * (t.tis_nbins - 1), 0.0);
            tis_bin_ln.assign((size_t)C * t.tis_nbins, NEG_INF);
            for (int c = 0; c < C; c++) {
                if (tisBins[c].nbins != t.tis_nbins) throw ConfigError("TRANSINITBIN shape differs between GC classes");
                for (int i = 0; i < t.tis_nbins - 1; i++) tis_bin_bounds[(size_t)c * (t.tis_nbins - 1) + i] = tisBins[c].bb[i];
                for (int i = 0; i < t.tis_nbins; i++) tis_bin_ln[(size_t)c * t.tis_nbins + i] = lnp(tisBins[c].av[i]);
            }
        }
    }

    // ---- intron parameters (reference IntronModel::readAllParameters, src/intronmodel.cc:295-415)
    std::vector<double> probShort(C), mal(C), inEmiLinear;
    {
        PblReader r(full + opt.get("/IntronModel/infile"));
        const int assSize = ipow4(t.As + t.Ae), dssSize = ipow4(t.Ds + t.De);
        r.need("[ASS]");
        r.comment(); int size = r.readInt();
        r.comment(); double c_ass = r.readInt();
        r.comment(); double asspseudo = r.readDouble();
        if (size != assSize) throw ConfigError("IntronModel: [ASS] size does not match ass_start/ass_end");
        std::vector<double> assprobs(size, asspseudo / (c_ass + asspseudo * size));
        for (;;) {
            r.comment();
            if (r.peek() < 0 || r.peek() == '[') break;
            int pn = r.readPattern(t.As + t.Ae);
            assprobs[pn] = r.readDouble() / 1000;
        }
        Bins assBins, dssBins;
        size_t sp = r.tell();
        if (r.gotoLineAfter("[ASSBIN]")) assBins = readBins(r); else r.seek(sp);
        r.need("[DSS]");
        r.comment(); size = r.readInt();
        r.comment(); (void)r.readInt();
        r.comment(); (void)r.readDouble();
        if (size != dssSize) throw ConfigError("IntronModel: [DSS] size does not match dss_start/dss_end");
        std::vector<double> dssprobs(size, 0.0);
        r.comment();
        for (int pn = 0; pn < size; pn++) {
            int q = r.readPattern(t.Ds + t.De);
            if (q != pn) throw ConfigError("IntronModel::readProbabilities:  Error reading file " + r.path());
            dssprobs[pn] = r.readDouble() / 1000;
            r.comment();
        }
        sp = r.tell();
        if (r.gotoLineAfter("[DSSBIN]")) dssBins = readBins(r); else r.seek(sp);
        // fold the binning into the pattern tables (reference aSSProb src/intronmodel.cc:1167-1177,
        // dSSProb :1231-1239); without hints every site has the consensus dinucleotide
        ass_pat.resize(assSize);
        for (int i = 0; i < assSize; i++)
            ass_pat[i] = lnp(assBins.nbins >= 1 ? assBins.av[assBins.index(assprobs[i])] : assprobs[i]);
        dss_pat.resize(t.dss_gc ? 2 * dssSize : dssSize);
        for (int i = 0; i < dssSize; i++)
            dss_pat[i] = lnp(dssBins.nbins >= 1 ? dssBins.av[dssBins.index(dssprobs[i])] : dssprobs[i]);
        if (t.dss_gc) { // a gc donor site: the pattern probability times non_gt_dss_prob, THEN the binning (src/intronmodel.cc:1232-1239)
            const double nonGt = opt.getDouble("/IntronModel/non_gt_dss_prob", 0.001);
            for (int i = 0; i < dssSize; i++) {
                const double pr = dssprobs[i] * nonGt;
                dss_pat[dssSize + i] = lnp(dssBins.nbins >= 1 ? dssBins.av[dssBins.index(pr)] : pr);
            }
        }
        t.ass_pat_invalid = std::log(0.001 * std::pow(.25, (int)(t.As + t.Ae)));
        r.need("[LENGTH]");
        r.comment();
        int dd = r.readInt();
        t.d = dd;
        len_intron.resize(dd + 1);
        for (int i = 0; i <= dd; i++) { r.comment(); len_intron[i] = lnp(r.readDouble() / 1000); }
        int NPi = 0;
        for (int c = 0; c < C; c++) {
            char tag[16];
            snprintf(tag, sizeof tag, "[%d]", c + 1);
            r.need(tag);
            r.need("[TRANSITION]");
            r.comment(); probShort[c] = r.readDouble();
            r.comment(); mal[c] = r.readDouble();
            r.need("[EMISSION]");
            r.comment(); int sz = r.readInt();
            r.comment(); int kk = r.readInt();
            r.comment(); (void)r.readDouble();
            r.comment();
            if (c == 0) { // (the order of the intron content model is the one its file states, src/intronmodel.cc:358)
                if (kk < 0 || kk > k) throw UnsupportedError("intron Markov order above the exon order; not supported");
                t.k_in = kk; NPi = ipow4(kk + 1);
                in_emi.assign((size_t)C * NPi, NEG_INF);
                inEmiLinear.assign((size_t)C * NPi, 0.0);
            }
            if (kk != t.k_in || sz != NPi) throw ConfigError("IntronModel: emission order mismatch");
            for (int i = 0; i < sz; i++) {
                r.comment();
                int pn = r.readPattern(t.k_in + 1);
                inEmiLinear[(size_t)c * NPi + pn] = r.readDouble();
                in_emi[(size_t)c * NPi + pn] = lnp(inEmiLinear[(size_t)c * NPi + pn]);
            }
            r.need("[ASSMOTIF]");
            Motif m = readMotif(r);
            if (c == 0) {
                t.ass_n = m.n;
                t.ass_k = m.k;
                ass_motif.assign((size_t)C * m.n * ipow4(m.k + 1), NEG_INF);
            } else if (m.n != t.ass_n || m.k != t.ass_k)
                throw ConfigError("ASSMOTIF shape differs between GC classes");
            for (size_t i = 0; i < m.p.size(); i++) ass_motif[(size_t)c * m.p.size() + i] = lnp(m.p[i]);
        }
        if (t.ass_n != t.U) throw ConfigError("ASSMOTIF width differs from /Constant/ass_upwindow_size");
        if (t.tis_n != t.W) throw ConfigError("TRANSINIT motif width differs from /Constant/trans_init_window");
    }

    // ---- igenic parameters (reference IGenicModel::readAllParameters, src/igenicmodel.cc:150-225)
    {
        PblReader r(full + opt.get("/IGenicModel/infile"));
        ig_emi.assign((size_t)C * NP, NEG_INF);
        ig_short.assign((size_t)C * (k + 1) * NP, NEG_INF);
        for (int c = 0; c < C; c++) {
            char tag[16];
            snprintf(tag, sizeof tag, "[%d]", c + 1);
            r.need(tag);
            r.comment();
            int kk = r.readInt();
            if (kk != k) throw ConfigError("IGenicModel: order mismatch");
            r.need("[P_ls]");
            std::vector<std::vector<double>> pls(k + 1);
            for (int i = 0; i <= k; i++) {
                r.comment(); int l = r.readInt(); r.comment();
                int size = ipow4(l + 1);
                pls[i].assign(size, 0.0);
                for (int j = 0; j < size; j++) {
                    r.comment();
                    int pn = r.readPattern(i + 1);
                    if (pn != j) throw ConfigError("IgenicModel::readProbabilities: Error reading file " + r.path() + " at P_ls");
                    pls[i][j] = r.readDouble();
                }
            }
            // short-pattern emission used at sequence positions 1..k, including the reference's index
            // arithmetic basek/4 + i (src/igenicmodel.cc:346-351)
            for (int b = 1; b <= k; b++) {
                int size = ipow4(b + 1);
                for (int basek = 0; basek < size; basek++) {
                    double den = pls[b][basek / 4] + pls[b][basek / 4 + 1] + pls[b][basek / 4 + 2] + pls[b][basek / 4 + 3];
                    ig_short[((size_t)c * (k + 1) + b) * NP + basek] = lnp(pls[b][basek] / den);
                }
            }
            size_t sp = r.tell();
            if (!r.gotoLineAfter("[EMISSION]")) { // old parameter files (reference src/igenicmodel.cc:196-204, computeEmiFromPat)
                r.seek(sp);
                for (int i = 0; i < NP; i += 4) {
                    const double sum = pls[k][i] + pls[k][i + 1] + pls[k][i + 2] + pls[k][i + 3];
                    for (int nuk = 0; nuk < 4; nuk++) ig_emi[(size_t)c * NP + i + nuk] = lnp(k > 0 ? pls[k][i + nuk] / sum : pls[k][i + nuk]);
                }
            } else {
                r.comment(); (void)r.readInt();
                for (int j = 0; j < NP; j++) {
                    r.comment();
                    int pn = r.readPattern(k + 1);
                    if (pn != j) throw ConfigError("IgenicModel::readProbabilities: Error reading file " + r.path() + " at EMISSION");
                    ig_emi[(size_t)c * NP + j] = lnp(r.readDouble());
                }
            }
            if (tie && t.k_in == k) // reference IGenicModel::updateToLocalGC, src/igenicmodel.cc:69-81 (not where the orders of the two models differ, :74)
                for (int j = 0; j < NP; j++) ig_emi[(size_t)c * NP + j] = in_emi[(size_t)c * NP + j];
        }
    }

    if (utr && t.k_in != k) throw UnsupportedError("UTR states with an intron Markov order other than the exon order; not supported");
    if (utr) loadUtr(full, inEmiLinear);

    // ---- per-class transition matrices (reference IntronModel::updateToLocalGCEach, src/intronmodel.cc:439-488)
    ln_trans.assign((size_t)C * t.S * t.S, NEG_INF);
    for (int c = 0; c < C; c++) {
        std::vector<double> T = trans;
        auto at = [&](int i, int j) -> double & { return T[(size_t)i * t.S + j]; };
        for (int cur = 0; cur < t.S; cur++) {
            int kind = t.state_kind[cur];
            double factor = 0;
            if (kind == AUGX_K_LESSD || kind == AUGX_K_RLESSD) factor = probShort[c];
            else if (kind == AUGX_K_EQUALD || kind == AUGX_K_REQUALD) factor = 1 - probShort[c];
            if (factor > 0)
                for (int i = 0; i < t.S; i++)
                    if (at(i, cur) > 0) at(i, cur) = factor;
            if (kind == AUGX_K_GEOMETRIC || kind == AUGX_K_RGEOMETRIC) {
                if (mal[c] > 0.0) at(cur, cur) = 1 - 1 / mal[c];
                double sum = 0;
                for (int i = 0; i < t.S; i++)
                    if (i != cur) sum += at(cur, i);
                if (sum > 0)
                    for (int i = 0; i < t.S; i++)
                        if (i != cur) at(cur, i) /= mal[c] * sum;
            }
        }
        for (size_t i = 0; i < T.size(); i++) ln_trans[(size_t)c * t.S * t.S + i] = lnp(T[i]);
    }
    // ancestors in ascending index (reference StateModel::initPredecessors, src/statemodel.cc:46-51)
    for (int s = 0; s < t.S; s++) {
        t.n_anc[s] = 0;
        for (int a = 0; a < t.S; a++)
            if (trans[(size_t)a * t.S + s] != 0) {
                if (t.n_anc[s] >= AUGX_MAX_ANC) throw UnsupportedError("state with more than AUGX_MAX_ANC ancestors");
                t.anc[s][t.n_anc[s]++] = a;
            }
    }

    // ---- GC-class decomposition (reference ContentDecomposition::makeDecomposition, src/motif.cc:464-487;
    //      BaseCount::init / setWeightMatrix, src/motif.cc:47-72,180-198)
    for (int i = 0; i < C; i++) {
        double quot = 1.25;
        double gc = gcMin + (gcMax - gcMin) * (i + 1) / (C + 1);
        double atc = 1 - gc;
        double quot_at = (2 - quot) + (2 * (quot - 1)) * (0 + 1) / (1 + 1);
        double quot_cg = (2 - quot) + (2 * (quot - 1)) * (0 + 1) / (1 + 1);
        t.gc_zus[i][0] = atc / (1 + quot_at);     // ra
        t.gc_zus[i][3] = atc / (1 + 1 / quot_at); // rt
        t.gc_zus[i][1] = gc / (1 + quot_cg);      // rc
        t.gc_zus[i][2] = gc / (1 + 1 / quot_cg);  // rg
    }
    t.gc_weighing_type = opt.getInt("/BaseCount/weighingType", 1);
    for (int i = 0; i < 16; i++) t.gc_weight_matrix[i] = 0;
    if (t.gc_weighing_type == 3) {
        PblReader r(full + opt.get("/BaseCount/weightMatrixFile"));
        r.comment();
        for (int i = 0; i < 16; i++) t.gc_weight_matrix[i] = r.readDouble();
    }
    quantiseTables();
    bindPointers();
}

// ---- UTR parameters (reference UtrModel::init src/utrmodel.cc:149-264, readAllParameters :540-696,
//      fillTailsOfLengthDistributions :293-361).  inEmi: the intron emission probabilities (linear), [C][NP]
void Model::loadUtr(const std::string &full, const std::vector<double> &inEmi) {
    // (the order of the UTR content tables is the one the parameter FILE states: UtrModel::k is overwritten while reading, :620)
    const int NPin = ipow4(t.k_in + 1), C = t.n_classes;
    int k = opt.getInt("/UtrModel/k", 4), NP = ipow4(k + 1);
    t.tss_upwin = opt.getInt("/Constant/tss_upwindow_size", 0);
    t.tss_start = opt.getInt("/UtrModel/tss_start", 4);
    t.tss_end = opt.getInt("/UtrModel/tss_end", 4);
    t.tata_start = opt.getInt("/UtrModel/tata_start", 1);
    t.tata_end = opt.getInt("/UtrModel/tata_end", 10);
    t.d_tss_tata_min = opt.getInt("/UtrModel/d_tss_tata_min", 17);
    t.d_tss_tata_max = opt.getInt("/UtrModel/d_tss_tata_max", 40);
    t.d_polyasig_cleavage = opt.getInt("/UtrModel/d_polyasig_cleavage", 20); // Constant::d_polyasig_cleavage, src/types.cc:40,407
    t.tts_spacing = 10;                                                        // UtrModel::ttsSpacing, src/utrmodel.cc:122
    t.utr_max_exon_len = opt.getInt("/UtrModel/maxexonlength");
    t.utr_max3single = opt.getInt("/UtrModel/max3singlelength");
    t.utr_max3term = opt.getInt("/UtrModel/max3termlength");
    const std::string cons = opt.get("/UtrModel/polyasig_consensus", "aataaa");
    t.aataaa_boxlen = (int)cons.size();
    const double probPolya = opt.getDouble("/UtrModel/prob_polya", 0.9);
    const double w5 = opt.getDouble("/UtrModel/utr5patternweight", 0.0), w3 = opt.getDouble("/UtrModel/utr3patternweight", 0.0);
    if (t.d_tss_tata_max + t.tata_start > t.tss_upwin)
        throw ConfigError("Inconsistent UTR training parameters. Must have d_tss_tata_max <= tss_upwindow_size - tata_start");
    if (t.d_tss_tata_min < t.tata_end + t.tss_start)
        throw ConfigError("Inconsistent UTR training parameters. Must have d_tss_tata_min >= tata_end + tss_start");
    t.ln2 = std::log(2.0);
    std::string fname = full + opt.get("/UtrModel/infile");
    {
        std::ifstream probe(fname.c_str());
        if (!probe) throw ConfigError("UtrModel::readProbabilities: Couldn't open file " + fname);
    }
    PblReader r(fname);
    r.need("[UTRLENGTH]");
    r.comment(); const int D = r.readInt();
    r.comment(); (void)r.readDouble();
    r.comment(); (void)r.readInt();
    double num[8], huge[8];
    r.comment(); for (int i = 0; i < 8; i++) num[i] = r.readInt();
    r.comment(); for (int i = 0; i < 8; i++) huge[i] = r.readInt();
    r.comment();
    const int ML = t.utr_max_exon_len, M3S = t.utr_max3single, M3T = t.utr_max3term;
    if (D > ML || D > M3S || D > M3T) throw ConfigError("UtrModel: exonLenD is larger than a max_exon_len.");
    const int maxLen[8] = {ML, ML, ML, ML, M3S, ML, ML, M3T};
    std::vector<double> dist[8], lnd[8];
    for (int i = 0; i < 8; i++) { dist[i].assign(maxLen[i] + 1, 0.0); lnd[i].assign(maxLen[i] + 1, NEG_INF); }
    for (int i = 0; i <= D; i++) {
        (void)r.readInt();
        for (int c = 0; c < 8; c++) dist[c][i] = r.readDouble() / 1000;
    }
    for (int c = 0; c < 8; c++) fillTail(dist[c], lnd[c], D, maxLen[c], huge[c], num[c]);
    // (the tail of the reference's LLDouble product never leaves the double range for shipped parameters; lnd[] continues in
    //  log space if it did, the tail sums below then treat such entries as 0)
    std::vector<double> *dst[8] = {&len5_single, &len5_initial, &len5_internal, &len5_terminal, &len3_single, &len3_initial, &len3_internal, &len3_terminal};
    for (int c = 0; c < 8; c++) {
        *dst[c] = lnd[c];
        // the device evaluates every predecessor end of a window; the reference skips ends through its EOPList
        // (src/statemodel.cc:473-520), which is the same set only if the support of the distribution is an interval
        int first = -1, last = -1;
        for (int i = 0; i <= maxLen[c]; i++) if (lnd[c][i] > NEG_INF) { if (first < 0) first = i; last = i; }
        for (int i = first; first >= 0 && i <= last; i++)
            if (!(lnd[c][i] > NEG_INF)) throw UnsupportedError("UTR length distribution with a gap in its support (EOPList order dependence); not supported");
    }
    auto tails = [&](const std::vector<double> &d, std::vector<double> &out) { // src/utrmodel.cc:341-360
        const int n = (int)d.size() - 1;
        double total = 0.0, cum = 0.0;
        for (int i = 0; i <= n; i++) total += d[i];
        out.assign(n + 1, NEG_INF);
        for (int i = n; i >= 0; i--) { cum += d[i]; out[i] = lnp(cum / total); }
    };
    tails(dist[0], tail5_single);
    tails(dist[4], tail3_single);
    r.need("[AATAAA]");
    r.comment(); const int asz = r.readInt();
    if (asz != ipow4(t.aataaa_boxlen)) throw ConfigError("Could not read in polyA signal.\n");
    aataaa.assign(asz, NEG_INF);
    for (;;) {
        r.comment();
        if (r.peek() < 0 || r.peek() == '[') break;
        const int pn = r.readPattern(t.aataaa_boxlen);
        const double p = r.readDouble();
        aataaa[pn] = lnp(p * probPolya);
    }
    t.ln_tts_rand = lnp((1.0 - probPolya) * (1.0 / ipow4(t.aataaa_boxlen)));
    bool sized = false;
    for (int c = 0; c < C; c++) {
        char tag[16];
        snprintf(tag, sizeof tag, "[%d]", c + 1);
        r.need(tag);
        auto readEmi = [&](const char *sec, std::vector<double> &lin) {
            r.need(sec);
            r.comment(); const int size = r.readInt();
            r.comment(); const int kk = r.readInt();
            r.comment(); (void)r.readDouble();
            if (!sized) { // (the first section of the first class sets the order)
                k = kk; NP = ipow4(k + 1); sized = true;
                if (k < 0 || k > t.k) throw UnsupportedError("UTR Markov order above the exon/intron order; not supported");
                utr5init_emi.assign((size_t)C * NP, NEG_INF); utr5_emi.assign((size_t)C * NP, NEG_INF); utr3_emi.assign((size_t)C * NP, NEG_INF);
            }
            if (kk != k || size != NP) throw ConfigError("UtrModel: emission order mismatch");
            lin.assign(NP, 0.0);
            for (int i = 0; i < size; i++) {
                r.comment();
                const int pn = r.readPattern(k + 1);
                lin[pn] = r.readDouble();
            }
        };
        std::vector<double> e5i, e5, e3;
        readEmi("[EMISSION-5INITIAL]", e5i);
        readEmi("[EMISSION-5]", e5);
        readEmi("[EMISSION-3]", e3);
        r.need("[EMISSION-TSSUPWIN]");
        r.comment(); const int usz = r.readInt();
        r.comment(); const int uk = r.readInt();
        r.comment(); (void)r.readDouble();
        if (c == 0) { t.tssup_k = uk; tssup_emi.assign((size_t)C * ipow4(uk + 1), NEG_INF); }
        else if (uk != t.tssup_k) throw ConfigError("UtrModel: tssup_k differs between GC classes");
        if (usz != ipow4(uk + 1)) throw ConfigError("UtrModel: EMISSION-TSSUPWIN size mismatch");
        for (int i = 0; i < usz; i++) {
            r.comment();
            const int pn = r.readPattern(uk + 1);
            tssup_emi[(size_t)c * usz + pn] = lnp(r.readDouble());
        }
        auto readM = [&](const char *sec, std::vector<double> &out, int32_t &mn, int32_t &mk) {
            r.need(sec);
            Motif m = readMotif(r);
            if (c == 0) { mn = m.n; mk = m.k; out.assign((size_t)C * m.p.size(), NEG_INF); }
            else if (m.n != mn || m.k != mk) throw ConfigError(std::string("UtrModel: motif shape of ") + sec + " differs between GC classes");
            for (size_t i = 0; i < m.p.size(); i++) out[(size_t)c * m.p.size() + i] = lnp(m.p[i]);
        };
        readM("[TSSMOTIF]", tss_motif, t.tss_n, t.tss_k);
        readM("[TSSMOTIFTATA]", tsstata_motif, t.tsstata_n, t.tsstata_k);
        readM("[TATAMOTIF]", tata_motif, t.tata_n, t.tata_k);
        readM("[TTSMOTIF]", tts_motif, t.tts_n, t.tts_k);
        // "change the content models so they are much closer to the intronmodel" (src/utrmodel.cc:681-688).  Entry i meets entry i:
        // with an order below the intron model's that is the intron pattern 'a..a' + pattern i -- as the reference does it
        for (int i = 0; i < NP; i++) {
            const double in = inEmi[(size_t)c * NPin + i];
            utr5init_emi[(size_t)c * NP + i] = lnp(e5i[i] * w5 + in * (1.0 - w5));
            utr5_emi[(size_t)c * NP + i] = lnp(e5[i] * w5 + in * (1.0 - w5));
            utr3_emi[(size_t)c * NP + i] = lnp(e3[i] * w3 + in * (1.0 - w3));
        }
    }
    t.utr_k = k;
}

// Exact arithmetic (include/augx.h: AUGX_Q_BITS): every ln term of the model is rounded ONCE, here, to a multiple of
// 2^-AUGX_Q_BITS.  All quantities of the decode are sums of such terms with magnitude below 2^(52 - AUGX_Q_BITS), so every
// fp64 addition on the decode path is exact: the result does not depend on the association of the sums, and adding a
// constant to all Viterbi values of a column commutes with everything that follows -- which is what lets a piece be decoded
// in segments that start from an unknown offset (segment-parallel trellis).  The rounding error is <= 2^-32 per
// factor, i.e. ~1e-13 relative on a Viterbi score (the reference's own LLDouble products round at 1.1e-16 per factor).
void Model::quantiseTables() {
    const double sc = std::ldexp(1.0, AUGX_Q_BITS), inv = std::ldexp(1.0, -AUGX_Q_BITS);
    auto q = [&](double &x) { if (std::isfinite(x)) x = std::nearbyint(x * sc) * inv; };
    for (std::vector<double> *v : {&ln_trans, &ig_emi, &ig_short, &in_emi, &ex_emi, &ex_init, &ex_et, &ex_pls, &tis_motif, &ass_motif,
                                   &tis_bin_ln, &ass_pat, &dss_pat, &len_intron, &len_single, &len_initial, &len_internal, &len_terminal,
                                   &utr5init_emi, &utr5_emi, &utr3_emi, &tssup_emi, &tss_motif, &tsstata_motif, &tata_motif, &tts_motif,
                                   &aataaa, &len5_single, &len5_initial, &len5_internal, &len5_terminal, &len3_single, &len3_initial,
                                   &len3_internal, &len3_terminal, &tail5_single, &tail3_single})
        for (double &x : *v) q(x);
    for (int i = 0; i < AUGX_MAX_STATES; i++) { q(t.ln_init[i]); q(t.ln_term[i]); }
    for (int i = 0; i < 64; i++) q(t.ln_startcodon[i]);
    q(t.ln_stop_ochre); q(t.ln_stop_amber); q(t.ln_stop_opal); q(t.ln_quarter); q(t.ln_n_coding); q(t.ln4);
    q(t.ass_pat_invalid); q(t.ln_soft_bonus); q(t.ln_tts_rand); q(t.ln2);
}

} // namespace augx
