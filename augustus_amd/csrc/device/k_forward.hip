// k_forward.hip -- forward algorithm (posterior sampling only) for one block size (AUGX_TU_BLK): one workgroup per piece, after the
// Viterbi decode.  Body: kernels.h: forwardPiece.
#include "kernels.h"
#include "launch.h"

using namespace augx;
using namespace augx::dev;

template <int BLK> __global__ void __launch_bounds__(NT) kForward(const DevTables *__restrict__ T, const BatchView *__restrict__ B) {
    __shared__ FwdLds lds;
    forwardPiece<BLK>(*T, *B, lds, blockIdx.x);
}

namespace augx { namespace dev {
void AUGX_TU_NAME(launchForward_)(unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) {
    hipLaunchKernelGGL(kForward<AUGX_TU_BLK>, dim3(grid), dim3(NT), 0, st, T, B);
}
}} // namespace
