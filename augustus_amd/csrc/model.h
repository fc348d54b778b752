// model.h -- host-side model: option store + flat ln-tables (see include/augx.h: augx_tables).
// Mirrors what the reference holds after Properties::init + Constant::init + NAMGene() +
// StateModel::readAllParameters() (reference src/augustus.cc:111-176), restricted to the ab-initio path.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/augx.h"

namespace augx {

struct ConfigError : std::runtime_error {
    explicit ConfigError(const std::string &m) : std::runtime_error(m) {}
};
struct UnsupportedError : std::runtime_error {
    explicit UnsupportedError(const std::string &m) : std::runtime_error(m) {}
};

// key/value option store with the reference's precedence (species cfg < command line < states cfg),
// reference src/properties.cc:66-450, line format src/properties.cc:539-555
class Options {
public:
    void readFile(const std::string &path, const std::string &configPath);
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    const std::string &get(const std::string &k) const;
    std::string get(const std::string &k, const std::string &dflt) const { return has(k) ? kv.at(k) : dflt; }
    int getInt(const std::string &k) const;
    int getInt(const std::string &k, int dflt) const { return has(k) ? getInt(k) : dflt; }
    double getDouble(const std::string &k) const;
    double getDouble(const std::string &k, double dflt) const { return has(k) ? getDouble(k) : dflt; }
    bool getBool(const std::string &k) const;
    bool getBool(const std::string &k, bool dflt) const { return has(k) ? getBool(k) : dflt; }
    void set(const std::string &k, const std::string &v) { kv[k] = v; }
    std::map<std::string, std::string> kv;
};

struct Model {
    Options opt;
    std::string configPath, species, speciesDir;
    std::string transFileUsed;       // for the "# human version. Using ..." header line
    bool speciesSpecificTrans = false;
    std::string transTable;          // 64 amino-acid letters by codon index aaa, aac, ... ttt ('*': stop) of --translation_table
    std::string stderrNotes;         // what the reference writes to its error stream while it reads the parameters
    std::string codeWarnings;        // the lines the reference prints first when the table is not the standard one
    augx_tables t{};
    // owning storage behind the pointers of t
    std::vector<double> ln_trans, ig_emi, ig_short, in_emi, ex_emi, ex_init, ex_et, ex_pls, tis_motif, ass_motif,
        tis_bin_bounds, tis_bin_ln, ass_pat, dss_pat, len_intron, len_single, len_initial, len_internal, len_terminal;
    // --UTR=on (reference UtrModel::readAllParameters, src/utrmodel.cc:540-696)
    std::vector<double> utr5init_emi, utr5_emi, utr3_emi, tssup_emi, tss_motif, tsstata_motif, tata_motif, tts_motif, aataaa,
        len5_single, len5_initial, len5_internal, len5_terminal, len3_single, len3_initial, len3_internal, len3_terminal,
        tail5_single, tail3_single;
    void loadUtr(const std::string &full, const std::vector<double> &inEmiLinear);
    void load(const std::string &configPath, const std::string &species,
              const std::vector<std::pair<std::string, std::string>> &cmdline);
    void bindPointers();
    void quantiseTables();
};

int stateTypeFromName(const std::string &name);       // reference stateTypeIdentifiers, src/types.cc:157-171
const char *stateTypeName(int type);
int kindOfType(int type);
int winOfType(int type);                               // stateReadingFrames, src/types.cc:174-189
inline int ipow4(int e) { return 1 << (2 * e); }

} // namespace augx
