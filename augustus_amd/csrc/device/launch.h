// launch.h -- host entry points of the heavy kernel families.  Each family is a translation unit of its own, compiled once per
// block size (Makefile: -DAUGX_TU_BLK=8|4|2), so that `make -j` builds them side by side and a change to one kernel rebuilds one
// object; decoder.hip (the prep kernels, scans and the device half of the C ABI) calls through these.
#pragma once
#include <hip/hip_runtime.h>
#include "dp.h"

namespace augx { namespace dev {
// kTrellis<BLK, MODE, TIES> (k_trellis.hip; kernels.h: trellisPiece)
void launchTrellis(int blk, int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);
// kCand<BLK, MULTI, DENSE> (k_cand.hip; kernels.h: candWorkgroup)
void launchCand(int blk, bool multi, bool dense, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W);
// kForward<BLK> (k_forward.hip; kernels.h: forwardPiece)
void launchForward(int blk, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);
// kDense<BLK, MODE, TIES>, kUtrDesc<BLK> (k_dense.hip; dense.h: densePiece, utrDescGroup)
void launchDense(int blk, int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);
void launchUtrDesc(int blk, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W);

#define AUGX_DECL_TU(BLK_)                                                                                                     \
    void launchTrellis_##BLK_(int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);     \
    void launchCand_##BLK_(bool multi, bool dense, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W);     \
    void launchForward_##BLK_(unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);                          \
    void launchDense_##BLK_(int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B);       \
    void launchUtrDesc_##BLK_(unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W);
AUGX_DECL_TU(8) AUGX_DECL_TU(4) AUGX_DECL_TU(2)
#undef AUGX_DECL_TU
#define AUGX_TU_CAT2(a, b) a##b
#define AUGX_TU_CAT(a, b) AUGX_TU_CAT2(a, b)
#define AUGX_TU_NAME(f) AUGX_TU_CAT(f, AUGX_TU_BLK)
}} // namespace
