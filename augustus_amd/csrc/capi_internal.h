// capi_internal.h -- opaque handle definitions shared by the C-ABI translation units
#pragma once
#include <string>
#include "model.h"

struct augx_model {
    augx::Model m;
};

namespace augx {
void setLastError(const std::string &m);
}
