// capi_model.cc -- C ABI, model half (host only): augx_model_load / tables / option / destroy.
#include <cstring>
#include <string>
#include <vector>
#include "capi_internal.h"

namespace augx {
thread_local std::string g_lastError;
void setLastError(const std::string &m) { g_lastError = m; }
} // namespace augx

using namespace augx;

extern "C" {

const char *augx_last_error(void) { return g_lastError.c_str(); }
const char *augx_version(void) { return "augx 0.1 (MI355X/gfx950 GHMM Viterbi decode; parity target AUGUSTUS 3.5.0)"; }

int augx_model_load(const char *config_path, const char *species, int n_opts, const char *const *opt_names,
                    const char *const *opt_values, augx_model **out) {
    if (!config_path || !species || !out) { setLastError("augx_model_load: NULL argument"); return AUGX_E_ARG; }
    *out = nullptr;
    augx_model *m = new augx_model();
    try {
        std::vector<std::pair<std::string, std::string>> cmd;
        for (int i = 0; i < n_opts; i++) cmd.emplace_back(opt_names[i], opt_values[i]);
        m->m.load(config_path, species, cmd);
    } catch (UnsupportedError &e) {
        setLastError(e.what());
        delete m;
        return AUGX_E_UNSUPPORTED;
    } catch (std::exception &e) {
        setLastError(e.what());
        delete m;
        return AUGX_E_CONFIG;
    }
    *out = m;
    return AUGX_OK;
}

const augx_tables *augx_model_tables(const augx_model *m) { return m ? &m->m.t : nullptr; }

const char *augx_model_option(const augx_model *m, const char *name) {
    if (!m || !name) return nullptr;
    auto it = m->m.opt.kv.find(name);
    return it == m->m.opt.kv.end() ? nullptr : it->second.c_str();
}

void augx_model_destroy(augx_model *m) { delete m; }
}
