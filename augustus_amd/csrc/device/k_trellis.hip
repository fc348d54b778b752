// k_trellis.hip -- the trellis kernels of one block size (AUGX_TU_BLK): pass 1 (MODE 0), fix-ups (1), continuation (2, 3), each
// also as the build whose chain wavefront flags near ties.  Body: kernels.h: trellisPiece.
#include "kernels.h"
#include "launch.h"

using namespace augx;
using namespace augx::dev;

// MODE 0: pass 1, one workgroup per segment (= per piece when no piece is cut); 1: the fix-ups; 2: continuation of pieces whose
// fix-up gave up, one workgroup per piece (kernels.h: trellisPiece).
// The default build (no near-tie flags) is ROLE-SPECIALISED: every wavefront of the workgroup branches once, on its (scalar)
// index, into the instantiation of trellisPiece that carries its own role's constants only -- 203 instead of 256 VGPRs and no
// scratch for <8, 0> (profiles/EXPERIMENTS.md, round 4/5).
// INVARIANT the switch rests on: the workgroup barriers of trellisPiece (BLOCK_SYNC / __syncthreads -> s_barrier) then sit in
// wave-divergent control flow, outside HIP's barrier contract; on s_barrier hardware it is sound exactly while all eight
// instantiations execute the SAME NUMBER of barriers -- every barrier of trellisPiece is reached by every wavefront, none stands
// under a condition that depends on the wavefront's index (role work between two barriers is guarded per role, the barriers are
// not).  The sequential emulator cannot see a violation; tests/test_gpu_parity.py::test_gpu_role_specialised_equals_common_body
// compares this build with the common-body (TIES) build cell for cell on the device.
template <int BLK, int MODE, bool TIES> __global__ void __launch_bounds__(NT) kTrellis(const DevTables *__restrict__ T, const BatchView *__restrict__ B) {
    __shared__ TrellisLds lds;
    if constexpr (!TIES) {
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        switch (w) {
            case 0: trellisPiece<BLK, MODE, TIES, 0>(*T, *B, lds, blockIdx.x); break;
            case 1: trellisPiece<BLK, MODE, TIES, 1>(*T, *B, lds, blockIdx.x); break;
            case 2: trellisPiece<BLK, MODE, TIES, 2>(*T, *B, lds, blockIdx.x); break;
            case 3: trellisPiece<BLK, MODE, TIES, 3>(*T, *B, lds, blockIdx.x); break;
            case 4: trellisPiece<BLK, MODE, TIES, 4>(*T, *B, lds, blockIdx.x); break;
            case 5: trellisPiece<BLK, MODE, TIES, 5>(*T, *B, lds, blockIdx.x); break;
            case 6: trellisPiece<BLK, MODE, TIES, 6>(*T, *B, lds, blockIdx.x); break;
            default: trellisPiece<BLK, MODE, TIES, 7>(*T, *B, lds, blockIdx.x); break;
        }
    } else
        trellisPiece<BLK, MODE, TIES>(*T, *B, lds, blockIdx.x);
}

namespace augx { namespace dev {
void AUGX_TU_NAME(launchTrellis_)(int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) {
    constexpr int BLK = AUGX_TU_BLK;
#define L(MODE_) do { if (ties) hipLaunchKernelGGL((kTrellis<BLK, MODE_, true>), dim3(grid), dim3(NT), 0, st, T, B); \
                      else hipLaunchKernelGGL((kTrellis<BLK, MODE_, false>), dim3(grid), dim3(NT), 0, st, T, B); } while (0)
    switch (mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; default: L(3); break; }
#undef L
}
}} // namespace
