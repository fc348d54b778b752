// k_dense.hip -- the dense Viterbi / forward kernel (71 states with UTR, 48 with two intergenic states) and the UTR descriptor kernel
// for one block size (AUGX_TU_BLK): one workgroup per piece.  Bodies: dense.h: densePiece, utrDescGroup.
#include "kernels.h"
#include "dense.h"
#include "launch.h"

using namespace augx;
using namespace augx::dev;

template <int BLK> __global__ void __launch_bounds__(NT) kUtrDesc(const DevTables *__restrict__ T, const BatchView B) {
    __shared__ UDescLds lds;
    utrDescGroup<BLK>(*T, B, lds, blockIdx.x);
}
// MODE 0: Viterbi (max, bit-exact), MODE 1: forward (ln-sum); TIES: the chain runs flag near ties (a build of its own)
template <int BLK, int MODE, bool TIES> __global__ void __launch_bounds__(NT) kDense(const DevTables *__restrict__ T, const BatchView *__restrict__ B) {
    __shared__ DenseLds lds;
    densePiece<BLK, MODE, TIES>(*T, *B, lds, blockIdx.x);
}

namespace augx { namespace dev {
void AUGX_TU_NAME(launchDense_)(int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) {
    constexpr int BLK = AUGX_TU_BLK;
    if (mode == 1) hipLaunchKernelGGL((kDense<BLK, 1, false>), dim3(grid), dim3(NT), 0, st, T, B);
    else if (ties) hipLaunchKernelGGL((kDense<BLK, 0, true>), dim3(grid), dim3(NT), 0, st, T, B);
    else hipLaunchKernelGGL((kDense<BLK, 0, false>), dim3(grid), dim3(NT), 0, st, T, B);
}
void AUGX_TU_NAME(launchUtrDesc_)(unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W) {
    hipLaunchKernelGGL((kUtrDesc<AUGX_TU_BLK>), dim3(grid), dim3(NT), 0, st, T, W);
}
}} // namespace
