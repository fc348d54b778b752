// k_cand.hip -- candidates of the variable-length states for one block size (AUGX_TU_BLK): one wavefront per tile of 64 bases
// (describe + count, reserve, evaluate).  Body: kernels.h: candWorkgroup.
// (MULTI: some piece of the batch has more than one GC class -- the plane of every class-dependent array is then chosen per
//  end base; batches without such a piece run the variant with the plane folded away.  DENSE: records for the dense kernels.)
#include "kernels.h"
#include "launch.h"

using namespace augx;
using namespace augx::dev;

template <int BLK, bool MULTI, bool DENSE> __global__ void __launch_bounds__(NT, 4) kCand(const DevTables *__restrict__ T, const BatchView B) {
    __shared__ CandLds lds; // (the batch view by value: its pointers are then known to be global, not generic)
    candWorkgroup<BLK, MULTI, DENSE>(*T, B, lds, blockIdx.x);
}

namespace augx { namespace dev {
void AUGX_TU_NAME(launchCand_)(bool multi, bool dense, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W) {
    constexpr int BLK = AUGX_TU_BLK;
    if (dense) { if (multi) hipLaunchKernelGGL((kCand<BLK, true, true>), dim3(grid), dim3(NT), 0, st, T, W);
                 else hipLaunchKernelGGL((kCand<BLK, false, true>), dim3(grid), dim3(NT), 0, st, T, W); }
    else if (multi) hipLaunchKernelGGL((kCand<BLK, true, false>), dim3(grid), dim3(NT), 0, st, T, W);
    else hipLaunchKernelGGL((kCand<BLK, false, false>), dim3(grid), dim3(NT), 0, st, T, W);
}
}} // namespace
