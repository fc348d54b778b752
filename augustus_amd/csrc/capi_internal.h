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

// sampling in two halves (device/decoder.hip): what the sampler reads of a piece is fetched and indexed ahead of the sampling
struct augx_sample_prep;
extern "C" {
int augx_batch_sample_prepare(augx_decoder *d, augx_batch *b, int piece, augx_sample_prep **out);
int augx_sample_prep_run(augx_sample_prep *h, int n_samples, augx_rand *R, augx_path *out);
void augx_sample_prep_destroy(augx_sample_prep *h);
}
