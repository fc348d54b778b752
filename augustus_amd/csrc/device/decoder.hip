// decoder.hip -- gfx950 kernels (thin __global__ wrappers around kernels.h), chunked prefix scans, and the
// device half of the C ABI (augx_decoder_* / augx_batch_* / augx_decode_batch).
//
// Launch shapes (MI355X: 256 CUs, wave64):
//   prep kernels : one thread per slot, 256-thread blocks, grid = N/256  (HBM-streaming, fully coalesced rows)
//   scans        : grid (chunks x fields), 256 threads x 4 slots, rows of CHUNK=1024 contiguous uint64
//   candidates   : one wavefront per tile of 64 bases, 8 tiles per workgroup, 2 workgroups per CU (77 KB LDS, 64 VGPRs)   [k_cand.hip]
//   trellis      : one workgroup of 8 role-specialised wavefronts per SEGMENT of a piece = one CU per segment (156 KB LDS),
//                  position-sequential inside it; fix-ups and continuations in further launches                           [k_trellis.hip]
//   dense        : one workgroup per piece (models with UTR states / two intergenic states)                                [k_dense.hip]
//   backtrace    : one wavefront per piece
// The heavy kernel families are translation units of their own (launch.h); this file keeps the prep kernels and the host objects.
// There is no CPU fallback anywhere in this file: without a HIP device augx_decoder_create fails.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <algorithm>
#include <map>
#include <array>
#include <cmath>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <system_error>
#include <thread>
#include <atomic>
#include <chrono>
#include <vector>
#include "kernels.h"
#include "dense.h"
#include "layout.h"
#include "launch.h"
#include "assmemo.h"
#include "../capi_internal.h"

using namespace augx;
using namespace augx::dev;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            setLastError(std::string("HIP error: ") + hipGetErrorString(_e) + " in " #expr + " (" __FILE__ ":" + \
                         std::to_string(__LINE__) + ")");                                               \
            return AUGX_E_HIP;                                                                          \
        }                                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// the bases of the 256 slots of a workgroup and of their surroundings, staged in LDS once: the per-slot code reads dozens of
// single bases around its own (motif windows, splice-site and codon tests) -- from LDS instead of one byte load each
constexpr int SLOT_HALO = 64;
struct SlotCodes {
    uint8_t c[256 + 2 * SLOT_HALO];
    int lo;
    __device__ void load(const BatchView &B) {
        const int64_t g0 = (int64_t)blockIdx.x * 256;
        const int p = B.chunkPiece[g0 / CHUNK]; // (256 | CHUNK: the slots of a workgroup belong to one piece)
        const int64_t o = B.off[p];
        const int n = B.len[p], q0 = (int)(g0 - o - 1) - SLOT_HALO;
        for (int i = threadIdx.x; i < 256 + 2 * SLOT_HALO; i += 256) {
            const int q = q0 + i;
            c[i] = (q >= 0 && q < n) ? B.code[o + 1 + q] : 4;
        }
        if (threadIdx.x == 0) lo = q0;
        __syncthreads();
    }
};
__global__ void __launch_bounds__(256) kEncode(BatchView B) {
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < B.N) k1Encode(B, g);
}
// (min, max) of the window classes of every 256 slots (no atomics: tens of thousands of wavefronts of one piece hammering the
// same two words cost more than the rest of the kernel); kClassFinal folds them per piece
__global__ void __launch_bounds__(256) kWindowClass(const DevTables *__restrict__ T, BatchView B, int32_t *blkMinMax) {
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int c = g < B.N ? k1WindowClass(*T, B, g) : -1;
    // all slots of a block belong to one piece (pieces are CHUNK-aligned, 256 | CHUNK): reduce in the wave first
    int mn = c >= 0 ? c : (1 << 30), mx = c;
    for (int o = 32; o >= 1; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    __shared__ int w[4][2];
    if ((threadIdx.x & 63) == 0) { w[threadIdx.x >> 6][0] = mn; w[threadIdx.x >> 6][1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blkMinMax[2 * (int64_t)blockIdx.x] = min(min(w[0][0], w[1][0]), min(w[2][0], w[3][0]));
        blkMinMax[2 * (int64_t)blockIdx.x + 1] = max(max(w[0][1], w[1][1]), max(w[2][1], w[3][1]));
    }
}
__global__ void __launch_bounds__(64) kClassFinal(BatchView B, const int32_t *blkMinMax) { // one wavefront per piece
    const int p = blockIdx.x;
    int mn = 1 << 30, mx = -1;
    for (int64_t b = B.off[p] / 256 + threadIdx.x; b < B.off[p + 1] / 256; b += 64) {
        mn = min(mn, blkMinMax[2 * b]);
        mx = max(mx, blkMinMax[2 * b + 1]);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (threadIdx.x == 0) {
        B.cls[p] = mn == mx ? mn : -1;
        B.nPlanes[p] = 1;
        B.planeCls[p * MAXPL] = B.cls[p];
    }
}
// smoothed GC-content stairs of the pieces whose windows disagree, on the device (the same algorithm as layout.h: stairsPlanes,
// reference ContentStairs::computeStairs, src/motif.cc:543-616): runs of equal window class, a run shorter than 1000 bases
// between two runs of one class is dissolved, classes are numbered by first appearance (planes), every base gets its plane.
// One workgroup per piece; a piece with more runs than STAIR_RUNS is left to the host (info[1] is set).
constexpr int STAIR_RUNS = 4096;
__global__ void __launch_bounds__(256) kStairs(const DevTables *__restrict__ T, BatchView B, int32_t *info /* [0] max planes, [1] pieces left to the host */) {
    const int p = blockIdx.x, t = threadIdx.x;
    if (B.cls[p] >= 0) return; // all windows agree: one class, plane 0 (gcPlane is zero)
    __shared__ int st[STAIR_RUNS];
    __shared__ uint8_t cl[STAIR_RUNS], pl[STAIR_RUNS];
    __shared__ int cnt[256];
    __shared__ int R, nPlS;
    const int n = B.len[p];
    int win = T->gc_win;
    if (win > n || win < 1) win = n;
    const int half = win / 2, last = n - win;
    const uint8_t *wc = B.gcRaw + B.off[p] + 1;
    // window starts 1..last where the class changes, counted per thread (contiguous chunks), then written in order
    const int per = (last + 256) / 256, a = 1 + t * per, b = a + per < last + 1 ? a + per : last + 1;
    int c = 0;
    for (int s0 = a; s0 < b; s0++) c += wc[s0] != wc[s0 - 1];
    cnt[t] = c;
    __syncthreads();
    if (t == 0) {
        int acc = 1; // (run 0 starts at window 0)
        for (int i = 0; i < 256; i++) { const int v = cnt[i]; cnt[i] = acc; acc += v; }
        R = acc;
    }
    __syncthreads();
    if (R > STAIR_RUNS) { if (t == 0) atomicAdd(&info[1], 1); return; }
    if (t == 0) { st[0] = 0; cl[0] = wc[0]; }
    int k = cnt[t];
    for (int s0 = a; s0 < b; s0++)
        if (wc[s0] != wc[s0 - 1]) { st[k] = s0 + half; cl[k] = wc[s0]; k++; } // window s is centred on base s + half
    __syncthreads();
    if (t == 0) {
        for (int q = 2; q < R; q++)
            if (st[q] - st[q - 1] < 1000 && st[q - 1] > 0 && cl[q - 2] == cl[q]) cl[q - 1] = cl[q];
        int map[AUGX_MAX_CLASSES], nPl = 0;
        for (int i = 0; i < AUGX_MAX_CLASSES; i++) map[i] = -1;
        for (int q = 0; q < R; q++) {
            const int cc = cl[q] < AUGX_MAX_CLASSES ? cl[q] : 0;
            if (map[cc] < 0) { B.planeCls[p * MAXPL + nPl] = cc; map[cc] = nPl++; } // (MAXPL = the largest class count a model may have)
            pl[q] = (uint8_t)map[cc];
        }
        B.cls[p] = B.planeCls[p * MAXPL];
        B.nPlanes[p] = nPl;
        nPlS = nPl;
        atomicMax(&info[0], nPl);
    }
    __syncthreads();
    if (nPlS <= 1) return;
    // every base: the plane of its run (thread = contiguous chunk of bases; the run is found by bisection, then walked)
    const int perB = (n + 255) / 256, b0 = t * perB, b1 = b0 + perB < n ? b0 + perB : n;
    if (b0 >= n) return;
    int lo = 0, hi = R - 1;
    while (lo < hi) { const int m = (lo + hi + 1) / 2; if (st[m] <= b0) lo = m; else hi = m - 1; }
    uint8_t *gp2 = B.gcPlane + B.off[p] + 1;
    for (int q = b0; q < b1; q++) {
        while (lo + 1 < R && st[lo + 1] <= q) lo++;
        gp2[q] = pl[lo];
    }
}
__global__ void kListCount(BatchView B) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B.nPieces) k1ListCount(B, p);
}
__global__ void __launch_bounds__(256) kSignals(const DevTables *__restrict__ T, BatchView B) {
    __shared__ SlotCodes C;
    C.load(B);
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < B.N) k1Signals(*T, B, g, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
}
__global__ void __launch_bounds__(256) kSiteSignals(const DevTables *__restrict__ T, BatchView B) {
    k1SiteSignals(*T, B, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}
__global__ void __launch_bounds__(256) kSiteConsts(const DevTables *__restrict__ T, BatchView B) { // grid.y = plane
    __shared__ SlotCodes C;
    C.load(B);
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < B.N) k1SiteConsts(*T, B, g, blockIdx.y, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
}

// ---- fused term + scan kernels: one workgroup of SCAN_T threads per scan block of SCAN_T slots, one thread per slot.  The terms
// of a slot (11 site counts + 6 stop positions, or 20 fixed-point content terms) are computed in registers in BOTH passes --
// totals of the block, then the scan proper -- so that the prefix arrays are written to HBM once and never read by the scans
// (round 1: separate term kernels wrote them, kScanTotals read them, kScanApply read and rewrote them: four passes over
// 296 B per base).  The bases the terms look at are staged in LDS (SlotCodes).  Integer sums and maxima are exact: any scan
// shape gives the same bits as the sequential loop of the emulator.  (SCAN_T = 256: several workgroups per compute unit hide
// each other's barriers; with one workgroup of 1024 threads per chunk the scans ran 40 % longer.)
constexpr int SCAN_T = 256, SCAN_W = SCAN_T / 64;
static_assert(CHUNK % SCAN_T == 0, "scan blocks do not straddle pieces");
template <int NF> struct ScanLds { uint64_t w[NF][SCAN_W]; };
// inclusive scan over the 64 lanes of a wavefront in the data-parallel-primitive form of the VALU moves (no trip through the LDS
// crossbar as __shfl_up makes: the scans issued 240 ds_bpermute per wavefront and stalled on them 45 % of their time,
// profiles/r06_sq.txt): row_shr:1/2/4/8 inside the rows of 16 lanes, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2
// and 3 (gfx9 DPP controls 0x111.., 0x142, 0x143).  A lane without a source reads 0: the identity of the sums and of the maxima
// of unsigned values
template <int CTRL, int ROW_MASK> __device__ inline uint64_t dppFrom(uint64_t x) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
template <bool SUM> __device__ inline uint64_t waveScanIncl(uint64_t x) {
    auto op = [](uint64_t a, uint64_t b) -> uint64_t { return SUM ? a + b : (a > b ? a : b); };
    x = op(x, dppFrom<0x111, 0xf>(x));
    x = op(x, dppFrom<0x112, 0xf>(x));
    x = op(x, dppFrom<0x114, 0xf>(x));
    x = op(x, dppFrom<0x118, 0xf>(x));
    x = op(x, dppFrom<0x142, 0xa>(x));
    x = op(x, dppFrom<0x143, 0xc>(x));
    return x;
}
// the first nSum fields are summed, the rest take the maximum
template <int NF> __device__ inline void blockTotals(const uint64_t (&v)[NF], int nSum, ScanLds<NF> &L, uint64_t *tot /* [NF] of this block */) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const uint64_t x = f < nSum ? waveScanIncl<true>(v[f]) : waveScanIncl<false>(v[f]);
        if (lane == 63) L.w[f][wave] = x;
    }
    __syncthreads();
    if (t < NF) {
        uint64_t x = L.w[t][0];
        for (int i = 1; i < SCAN_W; i++) x = t < nSum ? x + L.w[t][i] : (x > L.w[t][i] ? x : L.w[t][i]);
        tot[t] = x;
    }
}
// inclusive scan of every field across the slots of the block, on top of the block's exclusive offset pre[f]
template <int NF> __device__ inline void blockScan(uint64_t (&v)[NF], int nSum, ScanLds<NF> &L, const uint64_t *pre) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const uint64_t x = f < nSum ? waveScanIncl<true>(v[f]) : waveScanIncl<false>(v[f]);
        v[f] = x;
        if (lane == 63) L.w[f][wave] = x;
    }
    __syncthreads();
    if (t < NF) { // exclusive scan of the wave totals of field t, seeded with the block's offset
        uint64_t acc = pre[t];
        for (int i = 0; i < SCAN_W; i++) {
            const uint64_t x = L.w[t][i];
            L.w[t][i] = acc;
            acc = t < nSum ? acc + x : (acc > x ? acc : x);
        }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const uint64_t a = L.w[f][wave];
        v[f] = f < nSum ? v[f] + a : (v[f] > a ? v[f] : a);
    }
}
constexpr int NSF = NCNT + 6; // site scans: 11 counts (sums) then 6 stop positions (maxima)
__global__ void __launch_bounds__(SCAN_T) kSiteScanTotals(const DevTables *__restrict__ T, BatchView B, uint64_t *tot /* [N / SCAN_T][NSF] */) {
    __shared__ ScanLds<NSF> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NSF];
    k1SiteTermsCalc(*T, B, g, v, v + NCNT, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
    blockTotals<NSF>(v, NCNT, L, tot + (int64_t)blockIdx.x * NSF);
}
__global__ void __launch_bounds__(SCAN_T) kSiteScanApply(const DevTables *__restrict__ T, BatchView B, const uint64_t *tot) {
    __shared__ ScanLds<NSF> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NSF];
    k1SiteTermsCalc(*T, B, g, v, v + NCNT, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
    blockScan<NSF>(v, NCNT, L, tot + (int64_t)blockIdx.x * NSF);
#pragma unroll
    for (int f = 0; f < NCNT; f++) B.cnt[fidx(g, f, NCNT)] = (uint32_t)v[f];
#pragma unroll
    for (int f = 0; f < 6; f++) B.nsm[fidx(g, f, 6)] = (uint32_t)v[NCNT + f];
}
__global__ void __launch_bounds__(SCAN_T) kFxScanTotals(const DevTables *__restrict__ T, BatchView B, uint64_t *tot /* [nPl][N / SCAN_T][NFX] */) { // grid.y = plane
    __shared__ ScanLds<NFX> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NFX];
    if (!k1FxTermsCalc(*T, B, g, blockIdx.y, v, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO)) return; // (uniform over the block: it belongs to one piece)
    blockTotals<NFX>(v, NFX, L, tot + ((int64_t)blockIdx.y * (B.N / SCAN_T) + blockIdx.x) * NFX);
}
__global__ void __launch_bounds__(SCAN_T) kFxScanApply(const DevTables *__restrict__ T, BatchView B, const uint64_t *tot) { // grid.y = plane
    __shared__ ScanLds<NFX> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NFX];
    if (!k1FxTermsCalc(*T, B, g, blockIdx.y, v, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO)) return;
    blockScan<NFX>(v, NFX, L, tot + ((int64_t)blockIdx.y * (B.N / SCAN_T) + blockIdx.x) * NFX);
    uint64_t *fx = B.fx + (int64_t)blockIdx.y * B.N * NFX;
#pragma unroll
    for (int f = 0; f < NFX; f++) fx[fidx(g, f, NFX)] = v[f];
}
// exclusive scan of the block totals inside each piece: one wavefront per (piece, plane, field), 64 blocks per step (the first
// nSum fields are sums, the rest maxima)
__global__ void __launch_bounds__(64) kChunkOffsets(uint64_t *tot, BatchView B, int nf, int nSum) {
    const int p = blockIdx.x, f = blockIdx.z, lane = threadIdx.x;
    tot += (int64_t)blockIdx.y * (B.N / SCAN_T) * nf;
    const bool sum = f < nSum;
    uint64_t acc = 0;
    const int64_t c1 = B.off[p + 1] / SCAN_T;
    for (int64_t c0 = B.off[p] / SCAN_T; c0 < c1; c0 += 64) {
        const int64_t ch = c0 + lane;
        const uint64_t v = ch < c1 ? tot[ch * nf + f] : 0;
        uint64_t x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = (uint64_t)__shfl_up((unsigned long long)x, o, 64);
            if (lane >= o) x = sum ? x + y : (x > y ? x : y);
        }
        // exclusive value of the lane = what came before the 64 blocks combined with the inclusive value of the lane before
        const uint64_t prev = (uint64_t)__shfl_up((unsigned long long)x, 1, 64);
        const uint64_t ex = lane == 0 ? acc : (sum ? acc + prev : (acc > prev ? acc : prev));
        if (ch < c1) tot[ch * nf + f] = ex;
        const uint64_t last = (uint64_t)__shfl((unsigned long long)x, 63, 64);
        acc = sum ? acc + last : (acc > last ? acc : last);
    }
}

// ---- untranslated regions (dense.h): content prefix sums + begin-site counts in one fused scan, then the signal records, the
// end gates of the UTR exon states and the entries of the site lists
constexpr int NUF = NUFX + NUCNT;
__global__ void __launch_bounds__(SCAN_T) kUtrScanTotals(const DevTables *__restrict__ T, BatchView B, uint64_t *tot /* [N / SCAN_T][NUF] */) {
    __shared__ ScanLds<NUF> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NUF];
    k1UtrTermsCalc(*T, B, g, v, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
    blockTotals<NUF>(v, NUF, L, tot + (int64_t)blockIdx.x * NUF);
}
__global__ void __launch_bounds__(SCAN_T) kUtrScanApply(const DevTables *__restrict__ T, BatchView B, const uint64_t *tot) {
    __shared__ ScanLds<NUF> L;
    __shared__ SlotCodes C;
    C.load(B);
    const int64_t g = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    uint64_t v[NUF];
    k1UtrTermsCalc(*T, B, g, v, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
    blockScan<NUF>(v, NUF, L, tot + (int64_t)blockIdx.x * NUF);
#pragma unroll
    for (int f = 0; f < NUFX; f++) B.ufx[fidx(g, f, NUFX)] = v[f];
#pragma unroll
    for (int f = 0; f < NUCNT; f++) B.ucnt[fidx(g, f, NUCNT)] = (uint32_t)v[NUFX + f];
}
__global__ void kUtrListCount(BatchView B) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B.nPieces) k1UtrListCount(B, p);
}
__global__ void __launch_bounds__(256) kUtrSignals(const DevTables *__restrict__ T, BatchView B) {
    __shared__ SlotCodes C;
    C.load(B);
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < B.N) k1UtrSignals(*T, B, g, C.c, C.lo, C.lo + 256 + 2 * SLOT_HALO);
}
// ---- the two call-history caches of the reference that UTR states read, on pieces with several GC classes (dense.h, assmemo.h)
struct MemoReq { int n; int s[8]; };
__global__ void __launch_bounds__(256) kTssReplay(const DevTables *__restrict__ T, BatchView B, const double *mat, int32_t *changed) {
    const int p = blockIdx.y, li = blockIdx.x * 256 + threadIdx.x;
    if (B.nPlanes[p] <= 1 || B.cls[p] < 0) return;
    if (k1TssReplay(*T, B, p, li, mat + (B.off[p] + 1) * T->S)) atomicAdd(changed, 1);
}
// what the replay of the aSSProb memo reads: per acceptor site (entry listOffs[p] + 8 p + li; 8 more entries per piece for the sites
// past its end) q and the aliveness bits, per slot the end-gate bits of the asking UTR exon states
__global__ void __launch_bounds__(256) kMemoSites(const DevTables *__restrict__ T, BatchView B, const double *mat, MemoReq rq, int32_t *siteQ, uint8_t *alive) {
    const int p = blockIdx.y, li = blockIdx.x * 256 + threadIdx.x;
    if (B.nPlanes[p] <= 1 || B.cls[p] < 0) return;
    const int nList = (int)B.cnt[fidx(B.off[p] + B.len[p], CNT_LA, NCNT)];
    if (li >= nList + 8) return;
    uint8_t a = 0;
    const int q = li < nList + T->Ae ? memoAssSite(*T, B, p, li, mat + (B.off[p] + 1) * T->S, rq.s, rq.n, a) : -1;
    const int64_t i = B.listOffs[p] + 8 * (int64_t)p + li;
    siteQ[i] = q; alive[i] = a;
}
__global__ void __launch_bounds__(256) kMemoGates(const DevTables *__restrict__ T, BatchView B, MemoReq rq, uint8_t *gate) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= B.N) return;
    const int p = B.chunkPiece[g / CHUNK];
    const int j = (int)(g - B.off[p] - 1);
    gate[g] = (j >= 0 && j < B.len[p] && B.nPlanes[p] > 1 && B.cls[p] >= 0) ? memoGateBits(*T, B, p, j, rq.s, rq.n) : 0;
}
__global__ void __launch_bounds__(256) kAssPatch(const DevTables *__restrict__ T, BatchView B, const AssPatch *A, const int32_t *piece, int n, const AssSwIn *sw, LaSw *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) k1AssPatch(*T, B, piece[i], A[i], sw, out);
}
// (kUtrDesc, kDense: k_dense.hip; kCand: k_cand.hip; kTrellis: k_trellis.hip; kForward: k_forward.hip -- launch.h)
__global__ void __launch_bounds__(64) kDenseBacktrace(const DevTables *T, BatchView B) { denseBacktracePiece(*T, B, blockIdx.x); }

__global__ void __launch_bounds__(256) kTileCross(BatchView B) {
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gt < B.N / WAVE) tileCrossOne(B, gt);
}
__global__ void __launch_bounds__(64) kSegFinalize(BatchView B) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p < B.nPieces) segFinalizePiece(B, p);
}
__global__ void __launch_bounds__(64) kBacktrace(const DevTables *T, BatchView B) { backtracePiece(*T, B, blockIdx.x); }
// forward algorithm (posterior sampling only): one workgroup per piece, after the Viterbi decode (kernels.h: forwardPiece)
// (snipmemo.h) candidate terms rebuilt on the host from the reference's snippet cache go back into the candidate records
// (snipmemo.h) everything the replay of one piece's windows reads, packed for ONE copy to the host: the candidate records of each
// window's blocks in block order, the intron content prefix slots of both strands and every plane, the rows of the matrix that
// tells which cells are alive.  One workgroup per window.
struct GatherWin { int64_t o, gb0, poolOff, fxOff, fOff; int32_t b0, b1, g0, nSlots, nPl, r0, nRows, pad; };
__global__ void __launch_bounds__(256) kGatherWindows(BatchView V, const GatherWin *W, Item *outItems, uint64_t *outFx, const double *mat, double *outF, int S) {
    __shared__ uint32_t sc[256];
    __shared__ uint64_t running;
    const GatherWin w = W[blockIdx.x];
    const int t = threadIdx.x;
    if (t == 0) running = 0;
    __syncthreads();
    for (int base = w.b0; base <= w.b1; base += 256) { // the blocks, 256 at a time: exclusive scan of their record counts
        const int q = base + t;
        const uint32_t cnt = q <= w.b1 ? V.blkCnt[(w.gb0 + q) * 2 + 1] : 0;
        sc[t] = cnt;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t v = t >= off ? sc[t - off] : 0;
            __syncthreads();
            sc[t] += v;
            __syncthreads();
        }
        const uint64_t dst = (uint64_t)w.poolOff + running + (sc[t] - cnt);
        if (cnt) {
            const Item *src = V.items + V.blkOff[(w.gb0 + q) * 2 + 1];
            for (uint32_t i = 0; i < cnt; i++) outItems[dst + i] = src[i];
        }
        __syncthreads();
        if (t == 255) running += sc[255];
        __syncthreads();
    }
    const int64_t nFx = (int64_t)w.nPl * 2 * w.nSlots;
    for (int64_t i = t; i < nFx; i += 256) {
        const int pl = (int)(i / (2 * w.nSlots)), rev = (int)((i / w.nSlots) % 2), g = w.g0 + (int)(i % w.nSlots);
        outFx[w.fxOff + i] = V.fx[(int64_t)pl * V.N * NFX + fidx(w.o + g, rev ? FX_INR : FX_INF, NFX)];
    }
    if (mat && outF)
        for (int64_t i = t; i < (int64_t)w.nRows * S; i += 256) outF[w.fOff + i] = mat[(w.o + 1 + w.r0) * S + i];
}
__global__ void __launch_bounds__(256) kPatchItems(BatchView B, const uint64_t *idx, const double *te, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) B.items[idx[i]].te = te[i];
}

// ---- launch.h: the family of a block size lives in its own translation unit ----
namespace augx { namespace dev {
#define AUGX_BY_BLK(f, ...) do { if (blk == 8) f##8(__VA_ARGS__); else if (blk == 4) f##4(__VA_ARGS__); else f##2(__VA_ARGS__); } while (0)
void launchTrellis(int blk, int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) { AUGX_BY_BLK(launchTrellis_, mode, ties, grid, st, T, B); }
void launchCand(int blk, bool multi, bool dense, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W) { AUGX_BY_BLK(launchCand_, multi, dense, grid, st, T, W); }
void launchForward(int blk, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) { AUGX_BY_BLK(launchForward_, grid, st, T, B); }
void launchDense(int blk, int mode, bool ties, unsigned grid, hipStream_t st, const DevTables *T, const BatchView *B) { AUGX_BY_BLK(launchDense_, mode, ties, grid, st, T, B); }
void launchUtrDesc(int blk, unsigned grid, hipStream_t st, const DevTables *T, const BatchView &W) { AUGX_BY_BLK(launchUtrDesc_, grid, st, T, W); }
#undef AUGX_BY_BLK
}} // namespace

// ---------------------------------------------------------------------------------------------------
// host objects
// ---------------------------------------------------------------------------------------------------
struct augx_decoder {
    const augx_model *model = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    DevTables hostT;          // scalars + DEVICE table pointers
    DevTables *dT = nullptr;
    std::vector<void *> tableBufs;
    bool debugCells = false;
    int blk = 8;              // block size of the candidate / trellis kernels for this model (layout.h: chooseBlockSize)
    int nCU = 256;            // compute units of the device = trellis workgroups in flight (one per CU: 155 KB of LDS each)
    int share = 1;            // decoders working on this device at the same time (augx_decoder_set_share)
    // device buffers of destroyed batches, kept for the next batch (the cut finder decodes a small batch per round, a genome
    // many large ones: dozens of hipMalloc / hipFree per batch otherwise); given back to the runtime when an allocation fails
    std::multimap<size_t, void *> pool;
    std::unordered_map<void *, size_t> live;
    std::atomic<size_t> pooledBytes{0};
    std::mutex poolMu;        // (another decoder of the same device may empty this pool when its own allocation fails)
    std::vector<hipStream_t> copyStreams; // (snippetCacheReplay: one per piece replayed at a time)
    bool dense = false;        // the model is decoded by the dense kernels (dense.h)
    bool countNearTies = false;    // the back-trace counts the near ties on the chosen paths (AUGX_NEAR_TIES=1, augx_decoder_count_near_ties)
    int64_t nearTies = 0, nearTiePieces = 0; // ... summed over the batches whose paths were fetched
    double mallocSeconds = 0, mallocBytes = 0; // hipMalloc calls of this decoder so far (AUGX_TIMING; under poolMu: the replay helpers allocate from many threads)
    bool exactMulti = true;    // replay the reference's snippet cache on multi-class pieces for the Viterbi run as well (augx_decoder_set_exact)
};

namespace {
int snippetCacheReplay(augx_decoder *d, augx_batch *b, int64_t &nPatched, bool fromLists, const double *mat); // (below, with the forward algorithm)
int utrCachesReplay(augx_decoder *d, augx_batch *b, const double *mat, bool &rebuilt); // (below: tssProbsPlus and the aSSProb memo, UTR states on multi-class pieces)
// every decoder of the process (several may share a device: augx_decoder_set_share, the bench's resident batches): when an
// allocation fails, the buffers the OTHER decoders of the device keep for re-use are given back to the runtime as well
std::mutex g_regMu;
std::vector<augx_decoder *> g_decoders;
// the decoders of ONE DEVICE together keep at most this much for re-use (288 GB of HBM per device).  Round 6: a batch of 127 Mbp of
// multi-class DNA holds 170 GB, and what a destroyed batch could not leave in the pool went back to the driver -- which wipes freed
// memory lazily, so that the NEXT batch's hipMalloc of the same 40 GB waited 1.5-4 s for it (the 1 Gbp genome run: 33 s of hipMalloc
// in 26 batches).  An allocation that fails empties the pools and tries again (devMalloc), so the cap only has to leave the runtime
// some room.
constexpr size_t POOL_CAP_BYTES = (size_t)250 << 30;
// Buffers are handed out in size classes (a sixteenth of the next power of two: at most 1/8 larger than asked for): consecutive
// batches of a genome differ by a fraction of a per cent in their base counts, and a pooled buffer 0.3 % too small is no use
// (before: every other batch of the genome run found nothing in the pool that fitted and allocated all its arrays afresh)
size_t sizeClass(size_t bytes) {
    if (bytes < ((size_t)1 << 20)) return (bytes + 4095) & ~(size_t)4095;
    size_t p2 = (size_t)1 << 20;
    while (p2 < bytes) p2 <<= 1;
    const size_t step = p2 >> 4;
    return (bytes + step - 1) / step * step;
}
void poolReleaseLocked(augx_decoder *d) {
    for (auto &kv : d->pool) (void)hipFree(kv.second);
    d->pool.clear();
    d->pooledBytes = 0;
}
void poolRelease(augx_decoder *d) {
    std::lock_guard<std::mutex> lk(d->poolMu);
    poolReleaseLocked(d);
}
hipError_t devMalloc(augx_decoder *d, void **out, size_t bytes) {
    if (bytes == 0) bytes = 1;
    bytes = sizeClass(bytes);
    {
        std::lock_guard<std::mutex> lk(d->poolMu);
        auto it = d->pool.lower_bound(bytes);
        if (it != d->pool.end() && it->first <= bytes + bytes / 4 + 4096) { // a pooled buffer that is not wastefully large
            *out = it->second;
            d->live[*out] = it->first;
            d->pooledBytes -= it->first;
            d->pool.erase(it);
            return hipSuccess;
        }
    }
    const auto tm0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, bytes);
    {   // (developer aid: printed under AUGX_TIMING)
        std::lock_guard<std::mutex> lk(d->poolMu);
        d->mallocSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count();
        d->mallocBytes += (double)bytes;
    }
    if (e != hipSuccess) { // this decoder's pool first, then those of the other decoders on the device
        (void)hipGetLastError();
        poolRelease(d);
        e = hipMalloc(out, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> rk(g_regMu);
            for (augx_decoder *o : g_decoders)
                if (o != d && o->device == d->device) poolRelease(o);
            e = hipMalloc(out, bytes);
        }
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(d->poolMu); d->live[*out] = bytes; }
    return e;
}
void devFree(augx_decoder *d, void *p) {
    if (!p) return;
    size_t onDevice = 0; // what the decoders of this device hold in their pools (lock order: registry, then pool -- as devMalloc)
    {
        std::lock_guard<std::mutex> rk(g_regMu);
        for (augx_decoder *o : g_decoders)
            if (o->device == d->device) onDevice += o->pooledBytes.load();
    }
    std::lock_guard<std::mutex> lk(d->poolMu);
    auto it = d->live.find(p);
    if (it == d->live.end()) { (void)hipFree(p); return; }
    if (onDevice + it->second > POOL_CAP_BYTES) { (void)hipFree(p); d->live.erase(it); return; } // (the pools of a device are bounded together)
    d->pool.emplace(it->second, p);
    d->pooledBytes += it->second;
    d->live.erase(it);
}
} // namespace


constexpr int NARR = 26; // arrays managed by ensureArrays
struct augx_batch {
    augx_decoder *dec = nullptr;
    BatchLayout L;
    BatchView V;               // device pointers
    BatchView *dV = nullptr;   // device copy of V (kernels with high register pressure take it by pointer)
    std::vector<void *> bufs;
    int32_t *blkMinMax = nullptr; // [N/256][2] window-class range of every 256 slots
    int32_t *stairInfo = nullptr; // [2] kStairs: most planes of a piece, pieces left to the host
    int nPlAlloc = 0;          // planes the class-dependent arrays are allocated for (0: not yet)
    int64_t listCapAlloc = 0;  // entries per plane the candidate-list arrays are allocated for
    bool listsReady = false;   // the list offsets of this batch's pieces have been computed (its first decode)
    int64_t *dListOffs = nullptr;
    void *planeBufs[NARR] = {};  // (ensureArrays)
    bool utrScanned = false;   // (dense) the UTR prefix scan of the current decode has run already (first decode: before the lists are sized)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // start, prep done, trellis done, backtrace done
    hipEvent_t evFwd = nullptr; // forward matrix complete (the sampler waits for this, not for what the stream got after it)
    uint64_t nItems = 0, nPairs = 0;
    void *itemBuf = nullptr; // candidate buffer, sized per decode (kept while large enough)
    void *udBuf = nullptr;   // (dense, UTR) descriptor buffer, sized like the candidate buffer
    uint64_t nDescs = 0;
    bool decoded = false;
    bool itemsVerified = false; // the candidate buffer in use has held all candidates of this batch once
    SegPlan plan;            // segments of the trellis (layout.h: planSegments)
    int chunkTotPlanes = 1;  // planes the scan totals are allocated for
    std::vector<int64_t> hListOffs; // host copy of the list offsets (first entry of every piece in the list arrays)
    bool memoReplayed = false; // the site values of this decode have been rebuilt from the reference's tssProbsPlus / aSSProb caches already
    void *laSwBuf = nullptr;   // (dense, UTR) acceptor sites whose value changes during the sweep (BatchView::laSw)
    size_t nLaSw = 0;          // its entries
    std::vector<std::shared_ptr<augx::dev::AssMemoReplay>> memoOf; // [piece] the aSSProb memo as the sweep left it (the sampler goes on from a copy), or null
};

namespace {

template <class T> int devAlloc(augx_batch *b, T **ptr, int64_t count) {
    void *p = nullptr;
    const hipError_t e = devMalloc(b->dec, &p, (size_t)(count > 0 ? count : 1) * sizeof(T));
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        setLastError("augx: out of device memory while allocating a batch; decode fewer bases per batch");
        return AUGX_E_NOMEM;
    }
    HIP_TRY(e);
    b->bufs.push_back(p);
    *ptr = (T *)p;
    return 0;
}

// Arrays whose size is only known at decode time: the class-dependent arrays ([nPl][...], see BatchView: one plane when the
// batch is created, more when a piece turns out to have several GC classes) and the candidate-list arrays (sized from the
// counted sites).  Re-allocated, larger, when a decode needs more planes or entries than the batch has.
int ensureArrays(augx_batch *b, int nPl, int64_t listCap) {
    if (nPl <= b->nPlAlloc && listCap <= b->listCapAlloc) return 0;
    if (nPl < b->nPlAlloc) nPl = b->nPlAlloc;
    if (listCap < b->listCapAlloc) listCap = b->listCapAlloc;
    BatchView &V = b->V;
    const int64_t N = b->L.N, LC = listCap > 0 ? listCap : 1;
    struct Slot { void **field; size_t elem; int64_t count; int planes; };
    const Slot slots[NARR] = {
        {(void **)&V.fx, sizeof(uint64_t), N * NFX, nPl},          {(void **)&V.plsR, sizeof(double), N * 3, nPl},
        {(void **)&V.ldEnt, sizeof(IntronStart), LC, nPl},         {(void **)&V.rdEnt, sizeof(IntronStart), LC, nPl},
        {(void **)&V.laPls, sizeof(double), LC * 3, nPl},          {(void **)&V.laFx, sizeof(uint64_t), LC * 3, nPl},
        {(void **)&V.lrEt, sizeof(double), LC * 3, nPl},           {(void **)&V.lrFx, sizeof(uint64_t), LC * 3, nPl},
        {(void **)&V.atgD, sizeof(double), LC * 3, nPl},           {(void **)&V.atgFx, sizeof(uint64_t), LC, nPl},
        {(void **)&V.rsFx, sizeof(uint64_t), LC * 3, nPl},
        {(void **)&V.laPos, sizeof(int32_t), LC, 1},               {(void **)&V.laVal, sizeof(double), LC * 3, 1},
        {(void **)&V.lrPos, sizeof(int32_t), LC, 1},               {(void **)&V.lrVal, sizeof(double), LC * 3, 1},
        {(void **)&V.ldVal, sizeof(double), LC * 3, 1},            {(void **)&V.rdVal, sizeof(double), LC * 3, 1},
        {(void **)&V.atgPos, sizeof(int32_t), LC, 1},              {(void **)&V.rsPos, sizeof(int32_t), LC, 1},
        {(void **)&V.rsBegin, sizeof(double), LC, 1},
        // (dense kernels: the site lists of the UTR exon states; the value lists above are not used there)
        {(void **)&V.tfSite, sizeof(USite), b->dec->dense ? LC : 1, 1}, {(void **)&V.laSite, sizeof(USite), b->dec->dense ? LC : 1, 1},
        {(void **)&V.fsSite, sizeof(USite), b->dec->dense ? LC : 1, 1}, {(void **)&V.lrSite, sizeof(USite), b->dec->dense ? LC : 1, 1},
        {(void **)&V.tmSite, sizeof(USite), b->dec->dense ? LC : 1, 1}, {(void **)&V.rtSite, sizeof(USite), b->dec->dense ? LC : 1, 1}};
    for (int i = 0; i < NARR; i++) {
        const bool isN = i < 2; // (fx, plsR: sized by the slots, they only grow with the planes)
        if (isN && nPl == b->nPlAlloc && b->planeBufs[i]) continue;
        if (b->planeBufs[i]) { devFree(b->dec, b->planeBufs[i]); b->planeBufs[i] = nullptr; *slots[i].field = nullptr; }
        void *p = nullptr;
        if (devMalloc(b->dec, &p, (size_t)slots[i].planes * (size_t)slots[i].count * slots[i].elem) != hipSuccess) {
            (void)hipGetLastError();
            b->nPlAlloc = 0; b->listCapAlloc = 0;
            for (int k = 0; k < NARR; k++)
                if (b->planeBufs[k]) { devFree(b->dec, b->planeBufs[k]); b->planeBufs[k] = nullptr; }
            setLastError("augx: out of device memory for the candidate-list / per-GC-class arrays (" + std::to_string(nPl) + " classes in one piece); decode fewer bases per batch");
            return AUGX_E_NOMEM;
        }
        b->planeBufs[i] = p;
        *slots[i].field = p;
    }
    b->nPlAlloc = nPl;
    b->listCapAlloc = listCap;
    return 0;
}

} // namespace

extern "C" {

int augx_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ndev;
}

int augx_decoder_create(const augx_model *m, int device, augx_decoder **out) {
    if (!m || !out) { setLastError("augx_decoder_create: NULL argument"); return AUGX_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        setLastError("augx_decoder_create: no HIP device available (this library has no CPU decode path)");
        return AUGX_E_NODEVICE;
    }
    if (device < 0 || device >= ndev) { setLastError("augx_decoder_create: bad device index"); return AUGX_E_ARG; }
    const augx_tables &t = m->m.t;
    int blk = 8;
    try {
        blk = chooseBlockSize(t);
    } catch (std::exception &ex) {
        setLastError(ex.what());
        return AUGX_E_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(device));
    augx_decoder *d = new augx_decoder();
    d->model = m;
    d->device = device;
    d->blk = blk;
    d->dense = modelIsDense(t);
    d->countNearTies = getenv("AUGX_NEAR_TIES") && atoi(getenv("AUGX_NEAR_TIES")) != 0; // (NOT implied by AUGX_TIMING: timing runs time the product's default kernels)
    const char *dbg = getenv("AUGX_DEBUG_CELLS");
    d->debugCells = dbg && atoi(dbg) != 0;
    if (const char *ex = getenv("AUGX_EXACT_MULTICLASS")) d->exactMulti = atoi(ex) != 0;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) d->nCU = cu; else (void)hipGetLastError(); }
    const int rc = [&]() -> int { // (any failure below: the half-built decoder is destroyed, nothing leaks)
        HIP_TRY(hipStreamCreate(&d->stream));
        try { fillDevTablesScalars(t, d->hostT); } catch (std::exception &ex) { setLastError(ex.what()); return AUGX_E_UNSUPPORTED; }
        for (auto &sp : tableSpans(t, d->hostT)) {
            void *p = nullptr;
            size_t bytes = (size_t)(sp.count > 0 ? sp.count : 1) * sizeof(double);
            HIP_TRY(hipMalloc(&p, bytes));
            d->tableBufs.push_back(p);
            if (sp.count > 0) HIP_TRY(hipMemcpy(p, sp.src, (size_t)sp.count * sizeof(double), hipMemcpyHostToDevice));
            *sp.dst = (TabPtr)p;
        }
        HIP_TRY(hipMalloc((void **)&d->dT, sizeof(DevTables)));
        HIP_TRY(hipMemcpy(d->dT, &d->hostT, sizeof(DevTables), hipMemcpyHostToDevice));
        return AUGX_OK;
    }();
    if (rc) { augx_decoder_destroy(d); return rc; }
    { std::lock_guard<std::mutex> rk(g_regMu); g_decoders.push_back(d); }
    *out = d;
    return AUGX_OK;
}

int augx_decoder_set_share(augx_decoder *d, int n) {
    if (!d || n < 1) { setLastError("augx_decoder_set_share: bad argument"); return AUGX_E_ARG; }
    d->share = n;
    return AUGX_OK;
}

int augx_decoder_count_near_ties(augx_decoder *d, int on) { if (!d) return AUGX_E_ARG; d->countNearTies = on != 0; return AUGX_OK; }
int64_t augx_decoder_near_ties(const augx_decoder *d, int64_t *pieces) { if (pieces) *pieces = d ? d->nearTiePieces : 0; return d ? d->nearTies : 0; }
int augx_decoder_exact(const augx_decoder *d) { return d && d->exactMulti ? 1 : 0; }
int augx_decoder_set_exact(augx_decoder *d, int on) {
    if (!d) return AUGX_E_ARG;
    d->exactMulti = on != 0;
    return AUGX_OK;
}

int64_t augx_decoder_batch_capacity(augx_decoder *d) {
    if (!d) return 0;
    size_t freeB = 0, totalB = 0;
    if (hipSetDevice(d->device) != hipSuccess || hipMemGetInfo(&freeB, &totalB) != hipSuccess) return 16L * 1000 * 1000;
    // ~0.6 KB of per-base arrays + ~0.3 KB of candidates + the candidate lists (sized from the counted sites: ~20 B per base,
    // up to 150 B on site-dense sequence), with head room; a model with several GC classes may need the class-dependent
    // arrays (~0.2 KB per base) once more per extra class met inside one piece: room for two extra
    freeB += d->pooledBytes; // (buffers of earlier batches kept by this decoder are free for the next one)
    {   // several decoders on one device (the bench's resident batches, AUGX_DEVICES=0,0,...: one host thread each, all allocating at
        // the same time): a decoder plans for its share of the device, not for what happens to be free while the others have not
        // allocated yet (round 6: eight decoders on one MI355X each planned 128 Mbp and the 1 Gbp run died out of memory)
        size_t nOn = 0;
        std::lock_guard<std::mutex> rk(g_regMu);
        for (augx_decoder *o : g_decoders) nOn += o->device == d->device;
        if (nOn > 1) freeB = std::min(freeB, (size_t)((double)totalB * 0.9 / (double)nOn));
    }
    // (dense kernels: the ln V matrix itself, 8 S bytes per base, and once more for the forward matrix when sampling)
    int64_t cap = (int64_t)(freeB / (d->dense ? 1500 + 18 * (size_t)d->model->m.t.S : d->model->m.t.n_classes > 1 ? 2000 : 1500));
    if (cap > 128L * 1000 * 1000) cap = 128L * 1000 * 1000;
    if (cap < 1000 * 1000) cap = 1000 * 1000;
    return cap;
}

// with the forward matrix of posterior sampling on top (8 S bytes per base): round 6 -- until then augx_decode_sampled halved the
// capacity, 64 pieces of 1 Mbp per batch where 100 fit, and a batch more costs a forward pass more (a second per Mbp of its longest piece)
int64_t augx_decoder_sampled_capacity(augx_decoder *d) {
    const int64_t cap = augx_decoder_batch_capacity(d);
    if (!d || d->dense) return cap; // (the estimate of the dense kernels counts both matrices already)
    const int64_t base = d->model->m.t.n_classes > 1 ? 2000 : 1500;
    return cap * base / (base + 8 * (int64_t)d->model->m.t.S);
}

void augx_release_host_pools();
void augx_decoder_destroy(augx_decoder *d) {
    if (!d) return;
    bool last;
    { std::lock_guard<std::mutex> rk(g_regMu); g_decoders.erase(std::remove(g_decoders.begin(), g_decoders.end(), d), g_decoders.end()); last = g_decoders.empty(); }
    if (last) augx_release_host_pools();
    (void)hipSetDevice(d->device);
    poolRelease(d);
    for (void *p : d->tableBufs) (void)hipFree(p);
    if (d->dT) (void)hipFree(d->dT);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    for (hipStream_t cs : d->copyStreams) if (cs) (void)hipStreamDestroy(cs);
    delete d;
}

void augx_batch_destroy(augx_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->dec->device);
    for (void *p : b->bufs) devFree(b->dec, p);
    for (void *p : b->planeBufs)
        if (p) devFree(b->dec, p);
    if (b->itemBuf) devFree(b->dec, b->itemBuf);
    if (b->udBuf) devFree(b->dec, b->udBuf);
    if (b->laSwBuf) devFree(b->dec, b->laSwBuf);
    for (auto &e : b->ev)
        if (e) (void)hipEventDestroy(e);
    if (b->evFwd) (void)hipEventDestroy(b->evFwd);
    delete b;
}

// ---- the TSS window that begins at base 0 of a piece (include/augx.h: augx_tss0, augx_tss0_override; dense.h: k1UtrSignals)
static std::mutex g_tss0Mu;
static std::map<const char *, std::array<double, 2>> g_tss0Of; // by the piece's sequence pointer
int augx_tss0_override(const char *seq, const double *v) {
    if (!seq) { setLastError("augx_tss0_override: bad argument"); return AUGX_E_ARG; }
    std::lock_guard<std::mutex> lk(g_tss0Mu);
    if (v) g_tss0Of[seq] = {v[0], v[1]}; else g_tss0Of.erase(seq);
    return AUGX_OK;
}
int augx_tss0(const augx_model *m, const char *seq, int64_t len, double *out) {
    if (!m || !seq || len < 1 || !out) { setLastError("augx_tss0: bad argument"); return AUGX_E_ARG; }
    const augx_tables &t = m->m.t;
    out[0] = out[1] = AUGX_NINF;
    if (!t.utr) return AUGX_OK;
    try {
        DevTables T;
        fillDevTablesScalars(t, T);
        for (auto &sp : tableSpans(t, T)) *sp.dst = (TabPtr)sp.src; // (host tables)
        const int up = t.tss_upwin, te = t.tss_end;
        const int64_t need = std::min<int64_t>(len, (int64_t)up + te + 64);
        std::vector<uint8_t> code((size_t)need);
        auto enc = [](char c) -> uint8_t { switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return 4; } };
        for (int64_t i = 0; i < need; i++) code[(size_t)i] = enc(seq[i]);
        // the class of the first GC window: the first win/2 bases have it (ContentStairs::computeStairs, src/motif.cc:561-565)
        int win = t.gc_win;
        if (win > len || win < 1) win = (int)len;
        double cnt[4] = {0, 0, 0, 0};
        for (int i = 0; i < win; i++) { const uint8_t c = enc(seq[i]); if (c < 4) cnt[c] += 1.0; }
        Piece P;
        P.t = &T; P.n = (int)len; P.c = T.C == 1 ? 0 : nearestClass(T, cnt); P.o = 0; P.code = code.data(); P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
        P.lcode = nullptr; P.lLo = 0; P.lHi = 0;
        // (the window [0, up + te) and the few bases a reverse pattern reads past it lie inside `need`; the bound `right >= n` is the piece's)
        out[0] = tssProbCalc(P, 0, true);
        out[1] = tssProbCalc(P, 0, false);
    } catch (std::exception &ex) { setLastError(ex.what()); return AUGX_E_UNSUPPORTED; }
    return AUGX_OK;
}

int augx_batch_create(augx_decoder *d, const augx_piece *pieces, int n, augx_batch **out) {
    if (!d || !pieces || n < 1 || !out) { setLastError("augx_batch_create: bad argument"); return AUGX_E_ARG; }
    *out = nullptr;
    HIP_TRY(hipSetDevice(d->device));
    augx_batch *b = new augx_batch();
    b->dec = d;
    try {
        b->L.build(pieces, n);
    } catch (std::exception &ex) {
        setLastError(ex.what());
        delete b;
        return AUGX_E_ARG;
    }
    const BatchLayout &L = b->L;
    BatchSizes Z(L);
    BatchView &V = b->V;
    memset(&V, 0, sizeof V);
    V.nPieces = n; V.N = L.N; V.nChunks = L.nChunks;
    int rc = 0;
#define DA(field, T, count) do { T *_p = nullptr; rc = devAlloc(b, &_p, (count)); if (rc) { augx_batch_destroy(b); return rc; } field = _p; } while (0)
    int64_t *dOff; int32_t *dLen, *dIk, *dTk, *dCp; char *dRaw;
    DA(dOff, int64_t, n + 1); DA(dLen, int32_t, n); DA(dIk, int32_t, n); DA(dTk, int32_t, n); DA(dCp, int32_t, L.nChunks);
    DA(dRaw, char, Z.N);
    V.off = dOff; V.len = dLen; V.initKind = dIk; V.termKind = dTk; V.chunkPiece = dCp; V.raw = dRaw;
    DA(V.cls, int32_t, n); DA(V.clsMinMax, int32_t, 2 * n);
    DA(b->blkMinMax, int32_t, (Z.N / 256 + 1) * 2);
    DA(b->stairInfo, int32_t, 2);
    DA(V.nPlanes, int32_t, n); DA(V.planeCls, int32_t, (int64_t)n * MAXPL);
    DA(V.gcRaw, uint8_t, Z.N); DA(V.gcPlane, uint8_t, Z.N);
    V.nPl = 1; V.listCap = 0; // (the list arrays are sized by the first decode, from the counted sites)
    DA(V.code, uint8_t, Z.N);
    DA(V.cnt, uint32_t, Z.N * NCNT);
    DA(V.nsm, uint32_t, Z.N * 6);
    DA(V.sig, double, Z.N * NSIG);
    DA(V.gate, uint64_t, Z.N);
    DA(V.site, int32_t, Z.N * NSITE);
    DA(V.chunkTot, uint64_t, Z.N / SCAN_T * NFX);
    if (!d->dense) {
        DA(V.bp, uint16_t, Z.N * SP);
        DA(V.bpChain, uint8_t, Z.N * 8);
    } else {
        DA(V.bpD, uint8_t, Z.N * d->hostT.S);
        DA(V.ufx, uint64_t, Z.N * NUFX); DA(V.ucnt, uint32_t, Z.N * NUCNT); DA(V.usig, double, Z.N * NUSIG);
    }
    if (d->debugCells || d->dense) DA(V.cells, double, Z.N * d->hostT.S);
    if (d->countNearTies) DA(V.nearTie, int32_t, n);
    if (getenv("AUGX_PROF")) { DA(V.prof, uint64_t, (int64_t)n * 56 + 64); if (hipMemset(V.prof, 0, ((size_t)n * 56 + 64) * 8) != hipSuccess) { augx_batch_destroy(b); setLastError("augx_batch_create: hipMemset failed"); return AUGX_E_HIP; } }
    if (!d->dense) { DA(V.vig, double, Z.N); DA(V.longV, double, Z.N * 6); }
    DA(V.listCnt, int32_t, n); DA(b->dListOffs, int64_t, n + 1);
    V.listOffs = b->dListOffs;
    if ((rc = ensureArrays(b, 1, 0))) { augx_batch_destroy(b); return rc; }
    V.blk = d->blk;
    V.nBlk = Z.N / V.blk;
    DA(V.blkCnt, uint32_t, V.nBlk * 2); DA(V.blkSplit, uint32_t, V.nBlk * 3); DA(V.blkOff, uint64_t, V.nBlk * 2);
    DA(V.candAlloc, CandAlloc, 1);
    if (d->dense && d->hostT.utr) { DA(V.udOff, uint64_t, V.nBlk); DA(V.udCnt, uint32_t, V.nBlk); }
    DA(V.lnv, double, n); DA(V.status, int32_t, n); DA(V.finalState, int32_t, n); DA(V.pathCount, int32_t, n);
    {   // pieces whose TSS window at base 0 is answered from an earlier sequence (augx_tss0_override)
        std::vector<double> v;
        {
            std::lock_guard<std::mutex> lk(g_tss0Mu);
            if (!g_tss0Of.empty() && d->dense && d->hostT.utr)
                for (int p = 0; p < n; p++) {
                    auto it = g_tss0Of.find(pieces[p].seq);
                    if (it == g_tss0Of.end()) continue;
                    if (v.empty()) v.assign((size_t)n * 2, std::nan(""));
                    v[(size_t)p * 2] = it->second[0]; v[(size_t)p * 2 + 1] = it->second[1];
                }
        }
        if (!v.empty()) {
            double *dv = nullptr;
            DA(dv, double, (int64_t)n * 2);
            if (hipMemcpy(dv, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice) != hipSuccess) { augx_batch_destroy(b); setLastError("augx_batch_create: upload failed"); return AUGX_E_HIP; }
            V.tss0 = dv;
        }
    }
    DA(V.pathRec, int32_t, Z.pathCap * 3);
    // segments of the trellis: enough workgroups for every compute unit, none shorter than what a fix-up needs
    b->plan = planSegments(L, d->model->m.t, d->nCU / d->share > 0 ? d->nCU / d->share : 1, d->dense ? -1 : 0); // (dense kernels: one workgroup per piece)
    SegDesc *dSegs; int32_t *dSeg0;
    const int nSegs = (int)b->plan.segs.size();
    DA(dSegs, SegDesc, nSegs); DA(dSeg0, int32_t, n + 1);
    V.nSegs = nSegs; V.segs = dSegs; V.pieceSeg0 = dSeg0; V.segCheckTiles = b->plan.checkTiles;
    if (const char *e = getenv("AUGX_SEG_CHECK_TILES")) V.segCheckTiles = atoi(e); // (tests of the give-up path: an unreachable check length)
    DA(V.segStop, int32_t, nSegs); DA(V.segStatus, int32_t, nSegs); DA(V.segD, double, nSegs); DA(V.brkPos, int32_t, nSegs); DA(V.brkOff, double, nSegs);
    DA(V.segStop2, int32_t, nSegs); DA(V.segD2, double, nSegs); DA(V.pieceCovered, int32_t, n);
    if (b->plan.cut()) {
        DA(V.ckRing, double, (int64_t)nSegs * 2 * WAVE * SP); DA(V.ckCol, double, Z.N / WAVE * SP);
        DA(V.tileMinEop, int32_t, Z.N / WAVE); DA(V.tileCross, int32_t, Z.N / WAVE);
    }
#undef DA
    rc = [&]() -> int { // (any failure below: the batch is destroyed with everything it owns)
    const double tAlloc = getenv("AUGX_TIMING") ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
    HIP_TRY(hipMemcpy(dOff, L.off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dLen, L.len.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dIk, L.initKind.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dTk, L.termKind.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dCp, L.chunkPiece.data(), sizeof(int32_t) * L.nChunks, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dSegs, b->plan.segs.data(), sizeof(SegDesc) * nSegs, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dSeg0, b->plan.pieceSeg0.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dRaw, 'n', (size_t)Z.N));
    HIP_TRY(hipMemset(V.gcPlane, 0, (size_t)Z.N));
    for (int p = 0; p < n; p++)
        HIP_TRY(hipMemcpy(dRaw + L.off[p] + 1, pieces[p].seq, (size_t)L.len[p], hipMemcpyHostToDevice));
    { void *pv = nullptr; HIP_TRY(devMalloc(d, &pv, sizeof(BatchView))); b->bufs.push_back(pv); b->dV = (BatchView *)pv; }
    HIP_TRY(hipMemcpy(b->dV, &V, sizeof(BatchView), hipMemcpyHostToDevice));
    for (auto &e : b->ev) HIP_TRY(hipEventCreate(&e));
    if (tAlloc > 0) fprintf(stderr, "augx timing:       batch created: uploads + events %.3f s after the allocations (hipMalloc calls of this decoder so far: %.3f s for %.1f GB)\n",
                            std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tAlloc, d->mallocSeconds, d->mallocBytes / 1e9);
    return AUGX_OK;
    }();
    if (rc) { augx_batch_destroy(b); return rc; }
    *out = b;
    return AUGX_OK;
}

int augx_batch_decode(augx_decoder *d, augx_batch *b) {
    if (!d || !b) { setLastError("augx_batch_decode: NULL argument"); return AUGX_E_ARG; }
    HIP_TRY(hipSetDevice(d->device));
    const BatchView &V = b->V;
    const int n = V.nPieces;
    const unsigned gridN = (unsigned)((V.N + 255) / 256);
    hipStream_t st = d->stream;
    HIP_TRY(hipEventRecord(b->ev[0], st));
    b->memoReplayed = false; // (the prep kernels write the site values afresh)
    b->memoOf.clear();
    hipLaunchKernelGGL(kEncode, dim3(gridN), dim3(256), 0, st, V);
    int rc;
    // (developer aid, AUGX_TIMING: where the host spends the first decode of a batch -- classes, list sizes, array allocation)
    const bool timingFirst = !b->decoded && getenv("AUGX_TIMING") != nullptr;
    auto nowS = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tf[6] = {0, 0, 0, 0, 0, 0};
    if (timingFirst) tf[0] = nowS();
    // site counts and stop positions: terms and prefix scans fused (the prefix arrays are written once, never read back here)
    const unsigned nScan = (unsigned)(V.N / SCAN_T);
    hipLaunchKernelGGL(kSiteScanTotals, dim3(nScan), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
    hipLaunchKernelGGL(kChunkOffsets, dim3(n, 1, NSF), dim3(64), 0, st, V.chunkTot, V, NSF, NCNT);
    hipLaunchKernelGGL(kSiteScanApply, dim3(nScan), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
    if (!b->listsReady) hipLaunchKernelGGL(kListCount, dim3((n + 63) / 64), dim3(64), 0, st, V);
    b->utrScanned = false;
    auto utrScan = [&]() { // (dense) the UTR content prefix sums follow the class of each base: after the content stairs
        hipLaunchKernelGGL(kUtrScanTotals, dim3(nScan), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
        hipLaunchKernelGGL(kChunkOffsets, dim3(n, 1, NUF), dim3(64), 0, st, V.chunkTot, V, NUF, NUF);
        hipLaunchKernelGGL(kUtrScanApply, dim3(nScan), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
        b->utrScanned = true;
    };
    // GC classes, planes and list sizes are properties of the batch's sequences: settled by its first decode (with one host
    // round trip); a batch decoded again re-uses them and never waits for the host
    if (!b->decoded) {
    hipLaunchKernelGGL(kWindowClass, dim3(gridN), dim3(256), 0, st, d->dT, V, b->blkMinMax);
    hipLaunchKernelGGL(kClassFinal, dim3(n), dim3(64), 0, st, V, b->blkMinMax);
    HIP_TRY(hipGetLastError());
    {   // pieces whose 1-bp-shifted GC windows do not all agree: the smoothed content stairs are settled on the host from
        // the window classes (1 byte per base; reference ContentStairs::computeStairs, src/motif.cc:543-616), and the
        // classes of the piece become planes of the class-dependent arrays
        BatchView &W = b->V;
        HIP_TRY(hipMemsetAsync(b->stairInfo, 0, 2 * sizeof(int32_t), st));
        hipLaunchKernelGGL(kStairs, dim3(n), dim3(256), 0, st, d->dT, V, b->stairInfo);
        int32_t info[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(info, b->stairInfo, sizeof info, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int nPl = info[0] > 1 ? info[0] : 1;
        if (timingFirst) tf[1] = nowS();
        if (info[1] > 0) { // a piece with more class changes than the kernel holds (thousands): its stairs on the host
            std::vector<int32_t> cls(n), nPlanes(n), planeCls((size_t)n * MAXPL);
            HIP_TRY(hipMemcpy(cls.data(), V.cls, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(nPlanes.data(), V.nPlanes, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(planeCls.data(), V.planeCls, sizeof(int32_t) * n * MAXPL, hipMemcpyDeviceToHost));
            std::vector<uint8_t> wc, plane;
            for (int p = 0; p < n; p++) {
                if (cls[p] >= 0) continue;
                const int len = b->L.len[p];
                wc.resize((size_t)len);
                HIP_TRY(hipMemcpy(wc.data(), V.gcRaw + b->L.off[p] + 1, (size_t)len, hipMemcpyDeviceToHost));
                const int np = stairsPlanes(wc.data(), len, d->model->m.t.gc_win, plane, &planeCls[(size_t)p * MAXPL]);
                if (np < 0) continue; // (cannot happen: MAXPL is the largest class count a model may have)
                cls[p] = planeCls[(size_t)p * MAXPL];
                nPlanes[p] = np;
                if (np > nPl) nPl = np;
                if (np > 1) HIP_TRY(hipMemcpy(V.gcPlane + b->L.off[p] + 1, plane.data(), (size_t)len, hipMemcpyHostToDevice));
            }
            HIP_TRY(hipMemcpy(V.cls, cls.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(V.nPlanes, nPlanes.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(V.planeCls, planeCls.data(), sizeof(int32_t) * n * MAXPL, hipMemcpyHostToDevice));
        }
        bool changed = W.nPl != nPl;
        if (d->dense && !b->listsReady) { // the UTR site lists share the capacity of the candidate lists: count their sites first
            utrScan();
            hipLaunchKernelGGL(kUtrListCount, dim3((n + 63) / 64), dim3(64), 0, st, V);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
        }
        if (!b->listsReady) { // candidate lists: first entry of every piece from the counted sites
            std::vector<int32_t> lc(n);
            std::vector<int64_t> offs;
            HIP_TRY(hipMemcpy(lc.data(), V.listCnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            W.listCap = listOffsets(lc.data(), n, offs);
            b->hListOffs = offs;
            HIP_TRY(hipMemcpy(b->dListOffs, offs.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
            b->listsReady = true;
            changed = true;
        }
        const void *fxBefore = W.fx;
        if (timingFirst) tf[2] = nowS();
        if ((rc = ensureArrays(b, nPl, W.listCap))) return rc;
        if (timingFirst) tf[3] = nowS();
        if (changed || fxBefore != W.fx) {
            W.nPl = nPl;
            HIP_TRY(hipMemcpy(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice));
        }
    }
    }
    if (V.nPl > b->chunkTotPlanes) { // (one set of chunk totals per plane of the class-dependent arrays)
        uint64_t *nt = nullptr;
        if (devMalloc(d, (void **)&nt, sizeof(uint64_t) * (size_t)V.nPl * (V.N / SCAN_T) * NFX) != hipSuccess) {
            (void)hipGetLastError();
            setLastError("augx_batch_decode: out of device memory for the scan totals");
            return AUGX_E_NOMEM;
        }
        b->bufs.push_back(nt);
        b->V.chunkTot = nt;
        b->chunkTotPlanes = V.nPl;
    }
    if (d->dense && !b->utrScanned) utrScan();
    hipLaunchKernelGGL(kFxScanTotals, dim3(nScan, V.nPl), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
    hipLaunchKernelGGL(kChunkOffsets, dim3(n, V.nPl, NFX), dim3(64), 0, st, V.chunkTot, V, NFX, NFX);
    hipLaunchKernelGGL(kFxScanApply, dim3(nScan, V.nPl), dim3(SCAN_T), 0, st, d->dT, V, V.chunkTot);
    hipLaunchKernelGGL(kSignals, dim3(gridN), dim3(256), 0, st, d->dT, V);
    hipLaunchKernelGGL(kSiteSignals, dim3((unsigned)((V.listCap + 255) / 256), 4), dim3(256), 0, st, d->dT, V);
    hipLaunchKernelGGL(kSiteConsts, dim3(gridN, V.nPl), dim3(256), 0, st, d->dT, V);
    if (d->dense) hipLaunchKernelGGL(kUtrSignals, dim3(gridN), dim3(256), 0, st, d->dT, V);
    HIP_TRY(hipGetLastError());
    {   // candidates of the variable-length states.  The kernel reserves buffer space tile by tile; if the buffers turn
        // out too small (first decode of a batch, unusual sequence), it reports the size needed and is run again.
        BatchView &W = b->V;
        const unsigned nWg = (unsigned)(W.N / (WAVE * NWAVES)); // one wavefront per tile of 64 bases, NWAVES tiles per workgroup
        if (b->itemBuf && b->nItems > 0 && (uint64_t)W.itemCap > b->nItems + b->nItems / 16 + 65536) {
            // a batch decoded again: its candidate count is known, give back what the first estimate took too much
            HIP_TRY(hipStreamSynchronize(st));
            devFree(d, b->itemBuf);
            b->itemBuf = nullptr;
            b->itemsVerified = false;
        }
        if (!b->itemBuf) { // first estimate: uniform-random DNA has 1.2 pairs and 15 candidates per base
            W.itemCap = b->nItems > 0 ? (int64_t)(b->nItems + b->nItems / 16 + 65536) : W.N * 18 + 65536;
            if (devMalloc(d, &b->itemBuf, (size_t)W.itemCap * sizeof(Item)) != hipSuccess) {
                (void)hipGetLastError();
                b->itemBuf = nullptr;
                setLastError("augx_batch_decode: out of device memory for the candidate buffer; decode fewer bases per batch");
                return AUGX_E_NOMEM;
            }
            W.items = (Item *)b->itemBuf;
            HIP_TRY(hipMemcpyAsync(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice, st));
        }
        const bool utrDesc = d->dense && d->hostT.utr;
        if (utrDesc && !b->udBuf) { // descriptors of the UTR exon cells: uniform-random DNA has 0.35 per base with the human parameters
            W.udCap = b->nDescs > 0 ? (int64_t)(b->nDescs + 64) : W.N / 2 + 65536;
            if (const char *e = getenv("AUGX_UD_CAP")) { if (b->nDescs == 0) W.udCap = atol(e) > 0 ? atol(e) : 1; } // (tests: a first estimate that is too small)
            if (devMalloc(d, &b->udBuf, (size_t)W.udCap * sizeof(UDesc)) != hipSuccess) {
                (void)hipGetLastError();
                b->udBuf = nullptr;
                setLastError("augx_batch_decode: out of device memory for the UTR descriptors; decode fewer bases per batch");
                return AUGX_E_NOMEM;
            }
            W.ud = (UDesc *)b->udBuf;
            HIP_TRY(hipMemcpyAsync(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice, st));
        }
        // (steady: the batch has been decoded with this very buffer before -- the same sequences give the same candidates, no
        //  need to read the count back)
        const bool steady = b->decoded && b->itemBuf && b->nItems > 0 && (uint64_t)W.itemCap >= b->nItems && b->itemsVerified;
        for (int attempt = 0;; attempt++) {
            HIP_TRY(hipMemsetAsync(W.candAlloc, 0, sizeof(CandAlloc), st));
            const bool multi = W.nPl > 1;
            if (utrDesc) {
                const unsigned nGrp = (unsigned)(W.N / (NT / 16));
                launchUtrDesc(d->blk, nGrp, st, d->dT, W);
            }
            launchCand(d->blk, multi, d->dense, nWg, st, d->dT, W);
            HIP_TRY(hipGetLastError());
            if (steady) break;
            CandAlloc tot;
            HIP_TRY(hipMemcpyAsync(&tot, W.candAlloc, sizeof tot, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            b->nPairs = tot.pairs; b->nItems = tot.items; b->nDescs = tot.descs;
            const bool itemsFit = (int64_t)tot.items <= W.itemCap, descsFit = !utrDesc || (int64_t)tot.descs <= W.udCap;
            if (itemsFit && descsFit) { b->itemsVerified = true; break; }
            if (attempt > 0) { setLastError("augx_batch_decode: candidate buffers overflowed twice"); return AUGX_E_HIP; }
            if (!itemsFit) {
                devFree(d, b->itemBuf);
                b->itemBuf = nullptr;
                W.itemCap = (int64_t)tot.items + 64;
                if (devMalloc(d, &b->itemBuf, (size_t)W.itemCap * sizeof(Item)) != hipSuccess) {
                    (void)hipGetLastError();
                    b->itemBuf = nullptr;
                    setLastError("augx_batch_decode: out of device memory for the candidate buffer (" + std::to_string(tot.items) + " candidates); decode fewer bases per batch");
                    return AUGX_E_NOMEM;
                }
                W.items = (Item *)b->itemBuf;
            }
            if (!descsFit) {
                devFree(d, b->udBuf);
                b->udBuf = nullptr;
                W.udCap = (int64_t)tot.descs + 64;
                if (devMalloc(d, &b->udBuf, (size_t)W.udCap * sizeof(UDesc)) != hipSuccess) {
                    (void)hipGetLastError();
                    b->udBuf = nullptr;
                    setLastError("augx_batch_decode: out of device memory for the UTR descriptors; decode fewer bases per batch");
                    return AUGX_E_NOMEM;
                }
                W.ud = (UDesc *)b->udBuf;
            }
            HIP_TRY(hipMemcpyAsync(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice, st));
        }
    }
    if (timingFirst) {
        tf[4] = nowS();
        fprintf(stderr, "augx timing:       first decode of the batch, host side: class kernels + stairs %.3f s, stairs on the host + list sizes %.3f s, per-class / list arrays (%d planes) %.3f s, "
                        "prep launches + candidate buffer + candidates counted %.3f s (hipMalloc so far: %.3f s for %.1f GB)\n", tf[1] - tf[0], tf[2] - tf[1], b->V.nPl, tf[3] - tf[2], tf[4] - tf[3],
                d->mallocSeconds, d->mallocBytes / 1e9);
    }
    if (b->plan.cut()) hipLaunchKernelGGL(kTileCross, dim3((unsigned)((V.N / WAVE + 255) / 256)), dim3(256), 0, st, V); // how far back the future reads (fix-ups)
    HIP_TRY(hipEventRecord(b->ev[1], st));
    auto runTrellis = [&]() -> int { // the trellis passes and the back-trace (run again when candidate terms were rebuilt, see below)
    if (d->dense) { // the dense kernels: one workgroup per piece, the matrix in HBM (dense.h)
        launchDense(d->blk, 0, V.nearTie, (unsigned)n, st, d->dT, b->dV); // (near ties counted: the build whose chain runs flag them)
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(b->ev[2], st));
        hipLaunchKernelGGL(kDenseBacktrace, dim3(n), dim3(64), 0, st, d->dT, V);
        HIP_TRY(hipGetLastError());
        return AUGX_OK;
    }
    HIP_TRY(hipMemsetAsync(V.segStatus, 0, sizeof(int32_t) * V.nSegs, st));
#define AUGX_LAUNCH_TRELLIS(MODE_, grid_) launchTrellis(d->blk, MODE_, V.nearTie, (unsigned)(grid_), st, d->dT, b->dV) // (near ties counted: the build whose chain wavefront flags them)
    AUGX_LAUNCH_TRELLIS(0, V.nSegs);                 // pass 1: every segment at once (one workgroup per piece when no piece is cut)
    if (b->plan.cut()) {
        HIP_TRY(hipMemsetAsync(V.segStop2, 0xFF, sizeof(int32_t) * V.nSegs, st));
        HIP_TRY(hipMemsetAsync(V.pieceCovered, 0xFF, sizeof(int32_t) * n, st));
        // pass 2: the fix-ups (the first segment of a piece has none: its workgroup returns).  (Both passes in one launch of
        // persistent workgroups over a queue was measured: no faster in queue order -- a fix-up cannot start before pass 1 of
        // its neighbours -- and the kernel holding both bodies spills more, 88 instead of 54 VGPRs.)
        AUGX_LAUNCH_TRELLIS(1, V.nSegs);
        // pass 3: a fix-up that gave up continues, still comparing, one per piece and launch (workgroups with nothing to do
        // return at once); whatever is left after that goes sequentially to the end of its piece
        for (int round = 0; round < SEG_CONT_ROUNDS; round++) AUGX_LAUNCH_TRELLIS(2, n);
        AUGX_LAUNCH_TRELLIS(3, n);
    }
#undef AUGX_LAUNCH_TRELLIS
    hipLaunchKernelGGL(kSegFinalize, dim3((n + 63) / 64), dim3(64), 0, st, V);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->ev[2], st));
    hipLaunchKernelGGL(kBacktrace, dim3(n), dim3(64), 0, st, d->dT, V);
    HIP_TRY(hipGetLastError());
    return AUGX_OK;
    };
    int rcT = runTrellis();
    if (rcT) return rcT;
    b->decoded = true;
    // Pieces with several GC classes, exact mode (augx_decoder_set_exact; always on when sampling): the reference's snippet cache
    // around the class steps is replayed from which donor-site values are alive (snipmemo.h), the candidate terms concerned are
    // rebuilt and the trellis runs once more -- every Viterbi variable is then the reference's to 1e-9 there, too.  On by default:
    // a randomised soak found a record whose optimal path depends on it; AUGX_EXACT_MULTICLASS=0 trades that for
    // the second trellis run.
    if (d->exactMulti && b->nPlAlloc > 1) {
        int64_t nPatched = 0;
        int rc2;
        const bool timing = getenv("AUGX_TIMING") != nullptr; // (developer aid)
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        if (timing) (void)hipStreamSynchronize(st);
        const double t0 = timing ? now() : 0.0;
        bool sitesRebuilt = false;
        try {
            rc2 = d->dense ? snippetCacheReplay(d, b, nPatched, false, b->V.cells) : snippetCacheReplay(d, b, nPatched, true, nullptr);
            if (!rc2 && d->dense && d->hostT.utr) rc2 = utrCachesReplay(d, b, b->V.cells, sitesRebuilt);
        } catch (const std::exception &e) { setLastError(std::string("augx_batch_decode: replay of the reference's caches: ") + e.what()); return AUGX_E_NOMEM; }
        if (rc2) return rc2;
        const double t1 = timing ? now() : 0.0;
        if ((nPatched > 0 || sitesRebuilt) && (rc2 = runTrellis())) return rc2;
        if (timing) {
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "augx timing:     pieces with several GC classes: snippet-cache replay %.3f s (%lld candidate terms rebuilt), second trellis run %.3f s\n", t1 - t0, (long long)nPatched, now() - t1);
        }
    }
    HIP_TRY(hipEventRecord(b->ev[3], st));
    return AUGX_OK;
}

int augx_batch_sync(augx_decoder *d) {
    if (!d) return AUGX_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return AUGX_OK;
}

int augx_batch_kernel_ms(augx_decoder *d, augx_batch *b, float *prep_ms, float *trellis_ms, float *back_ms) {
    if (!d || !b || !b->decoded) return AUGX_E_ARG;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipEventSynchronize(b->ev[3]));
    float a = 0, c = 0, e = 0;
    HIP_TRY(hipEventElapsedTime(&a, b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&c, b->ev[1], b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&e, b->ev[2], b->ev[3]));
    if (prep_ms) *prep_ms = a;
    if (trellis_ms) *trellis_ms = c;
    if (back_ms) *back_ms = e;
    if (b->V.prof && d->dense) { // developer aid (AUGX_PROF=1): cycles per stage of the dense kernel, averaged over pieces
        std::vector<uint64_t> h((size_t)b->L.nPieces * 56 + 64);
        HIP_TRY(hipMemcpy(h.data(), b->V.prof, h.size() * 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "kDense Mcycles/piece:");
        for (int i = 0; i < 11; i++) {
            double sum = 0;
            for (int p = 0; p < b->L.nPieces; p++) sum += (double)h[(size_t)p * 56 + i];
            fprintf(stderr, " [%d]=%.2f", i, sum / b->L.nPieces / 1e6);
        }
        fprintf(stderr, "  (wavefront 1: 0 loop, 1 stage 1: fixed-lag states + staging, 2 ... wait for the UTR units, 3 stage 2: records + early chains' inputs, 4 forward sums, "
                        "5 stage 3: cells + early chain runs, 6 last-base redo, 7 stage 4: late chains, 8 RTERMINAL records, 9 their cells, 10 fence)\n");
    } else if (b->V.prof) { // developer aid (AUGX_PROF=1): cycle counters of the four trellis wavefronts, averaged over pieces
        std::vector<uint64_t> h((size_t)b->L.nPieces * 56 + 64);
        HIP_TRY(hipMemcpy(h.data(), b->V.prof, h.size() * 8, hipMemcpyDeviceToHost));
        static const char *role[5] = {"work0", "work1", "work2", "chain", "far"};
        for (int w = 0; w < 5; w++) {
            fprintf(stderr, "trellis %-6s Mcycles/piece:", role[w]);
            for (int i = 0; i < 8; i++) {
                double sum = 0;
                for (int p = 0; p < b->L.nPieces; p++) sum += (double)h[((size_t)p * 5 + w) * 8 + i];
                fprintf(stderr, " [%d]=%.2f", i, sum / b->L.nPieces / 1e6);
            }
            fprintf(stderr, "  (0 tile wait, 1 flag wait, 2 role work, 3 chain, 4 item load, 5 scan, 6 publish, 7 item setup)\n");
        }
        {
            const uint64_t *cp = &h[(size_t)b->L.nPieces * 56];
            const double nw = (double)(cp[4] ? cp[4] : 1);
            fprintf(stderr, "kCand cycles per wavefront (= tile; avg over %.0f): %.0f  (describe + count %.0f, reserve %.0f)\n", nw, cp[0] / nw, cp[1] / nw, cp[2] / nw);
        }
        {   // time stamps of block 1000 of piece 0, relative to the start of its fixed-lag step
            const uint64_t *ts = &h[(size_t)b->L.nPieces * 40];
            fprintf(stderr, "block 1000 of piece 0 (cycles): F %lld..%lld  I1 %lld..%lld  I2 %lld..%lld  late %lld..%lld  chain %lld..%lld\n",
                    0LL, (long long)(ts[1] - ts[0]), (long long)(ts[2] - ts[0]), (long long)(ts[3] - ts[0]), (long long)(ts[4] - ts[0]),
                    (long long)(ts[5] - ts[0]), (long long)(ts[6] - ts[0]), (long long)(ts[7] - ts[0]), (long long)(ts[8] - ts[0]), (long long)(ts[9] - ts[0]));
            fprintf(stderr, "end of tile 124 (cycles after worker 0 left its last block): igenic tail %lld..%lld  worker-0 tail done %lld  far tail done %lld  barrier passed %lld  (block 1000: late starts %lld)\n",
                    (long long)(ts[11] - ts[10]), (long long)(ts[12] - ts[10]), (long long)(ts[13] - ts[10]), (long long)(ts[14] - ts[10]), (long long)(ts[15] - ts[10]), (long long)(ts[6] - ts[10]));
        }
    }
    return AUGX_OK;
}

int augx_batch_paths(augx_decoder *d, augx_batch *b, augx_path *out) {
    if (!d || !b || !out || !b->decoded) { setLastError("augx_batch_paths: bad argument"); return AUGX_E_ARG; }
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    const BatchView &V = b->V;
    const int n = V.nPieces;
    std::vector<double> lnv(n);
    std::vector<int32_t> status(n), count(n);
    HIP_TRY(hipMemcpy(lnv.data(), V.lnv, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status.data(), V.status, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(count.data(), V.pathCount, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (V.nearTie) { // near ties on the chosen paths of this batch (dp.h: AUGX_NEAR_TIE), summed on the decoder
        std::vector<int32_t> nt(n);
        HIP_TRY(hipMemcpy(nt.data(), V.nearTie, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        for (int p = 0; p < n; p++)
            if (status[p] == 0 && nt[p] > 0) { d->nearTies += nt[p]; d->nearTiePieces++; }
    }
    const augx_tables &t = d->model->m.t;
    for (int p = 0; p < n; p++) {
        out[p].states = nullptr;
        out[p].n_states = 0;
        out[p].status = status[p];
        out[p].ln_viterbi = lnv[p];
        if (status[p] != 0) continue;
        int cnt = count[p];
        std::vector<int32_t> rec((size_t)cnt * 3 + 3);
        int64_t po = b->L.off[p] / 8 + 64 * (int64_t)p;
        if (cnt > 0) HIP_TRY(hipMemcpy(rec.data(), V.pathRec + po * 3, sizeof(int32_t) * 3 * (size_t)cnt, hipMemcpyDeviceToHost));
        out[p].states = (augx_state *)malloc(sizeof(augx_state) * (size_t)(cnt > 0 ? cnt : 1));
        if (!out[p].states) {
            for (int q = 0; q < p; q++) augx_path_free(&out[q]);
            setLastError("augx_batch_paths: out of host memory");
            return AUGX_E_NOMEM;
        }
        out[p].n_states = cnt;
        for (int i = 0; i < cnt; i++) {
            const int32_t *r = &rec[(size_t)(cnt - 1 - i) * 3];
            out[p].states[i].begin = r[0];
            out[p].states[i].end = r[1];
            out[p].states[i].state = (int16_t)r[2];
            out[p].states[i].type = (int16_t)t.state_type[r[2]];
        }
    }
    return AUGX_OK;
}

int augx_batch_cells(augx_decoder *d, augx_batch *b, int piece, double *out) {
    if (!d || !b || !out || piece < 0 || piece >= b->V.nPieces) return AUGX_E_ARG;
    if (!b->V.cells) { setLastError("augx_batch_cells: decoder was not created with AUGX_DEBUG_CELLS=1"); return AUGX_E_ARG; }
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    const int S = d->hostT.S;
    HIP_TRY(hipMemcpy(out, b->V.cells + (b->L.off[piece] + 1) * S, sizeof(double) * (size_t)b->L.len[piece] * S, hipMemcpyDeviceToHost));
    const int s0 = b->plan.pieceSeg0[piece], K = b->plan.pieceSeg0[piece + 1] - s0;
    if (K > 1) { // a piece decoded in segments stores its values region by region up to a constant (dp.h: frameOff)
        std::vector<int32_t> brkPos(K);
        std::vector<double> brkOff(K);
        HIP_TRY(hipMemcpy(brkPos.data(), b->V.brkPos + s0, sizeof(int32_t) * K, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(brkOff.data(), b->V.brkOff + s0, sizeof(double) * K, hipMemcpyDeviceToHost));
        int r = 0;
        for (int q = 0; q < b->L.len[piece]; q++) {
            while (r + 1 < K && q > brkPos[r]) r++;
            if (brkOff[r] != 0.0)
                for (int s2 = 0; s2 < S; s2++) out[(size_t)q * S + s2] += brkOff[r];
        }
    }
    return AUGX_OK;
}

static int augx_batch_forward_launch(augx_decoder *d, augx_batch *b) {
    if (!d || !b || !b->decoded) { setLastError("augx_batch_forward: the batch has not been decoded"); return AUGX_E_ARG; }
    HIP_TRY(hipSetDevice(d->device));
    BatchView &W = b->V;
    if (!W.fwd) {
        void *p = nullptr, *q = nullptr;
        if (devMalloc(d, &p, sizeof(double) * (size_t)W.N * d->hostT.S) != hipSuccess || devMalloc(d, &q, sizeof(double) * (size_t)W.nPieces) != hipSuccess) {
            (void)hipGetLastError();
            if (p) devFree(d, p);
            setLastError("augx_batch_forward: out of device memory for the forward matrix; decode fewer bases per batch");
            return AUGX_E_NOMEM;
        }
        b->bufs.push_back(p); b->bufs.push_back(q);
        W.fwd = (double *)p; W.lnFwd = (double *)q;
        HIP_TRY(hipMemcpyAsync(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice, d->stream));
    }
    if (d->dense) launchDense(d->blk, 1, false, (unsigned)W.nPieces, d->stream, d->dT, b->dV);
    else launchForward(d->blk, (unsigned)W.nPieces, d->stream, d->dT, b->dV);
    HIP_TRY(hipGetLastError());
    return AUGX_OK;
}

} // extern "C"

namespace {
// The copies of a piece (0.7 KB per base, into pageable memory: 11-13 GB/s however many run) are let through three pieces at a
// time, the piece the sampling thread will ask for first before the others: the pieces of a batch are prepared side by side on
// many threads, and with all of them copying at once the first piece arrived when all had (measured, 32 x 1 Mbp: 1.7 s
// against 0.5 s).
std::mutex g_copyMu;
std::condition_variable g_copyCv;
int g_copyFree = getenv("AUGX_COPY_SLOTS") ? atoi(getenv("AUGX_COPY_SLOTS")) : 3; // (the variable: developer aid)
std::multiset<int> g_copyWaiting;
struct CopySlot {
    explicit CopySlot(int piece) {
        std::unique_lock<std::mutex> lk(g_copyMu);
        auto me = g_copyWaiting.insert(piece);
        g_copyCv.wait(lk, [&] { return g_copyFree > 0 && *g_copyWaiting.begin() >= piece; });
        g_copyWaiting.erase(me);
        g_copyFree--;
        lk.unlock();
        g_copyCv.notify_all(); // (the next in line may have been waiting for this one to go first)
    }
    ~CopySlot() { { std::lock_guard<std::mutex> lk(g_copyMu); g_copyFree++; } g_copyCv.notify_all(); }
};
}

#include "snipmemo.h"
#include "assmemo.h"

namespace {
// Pieces with several GC classes: within 2 d bases after a class step the reference's short-intron interiors are products of
// cached chunks scored under different classes (snipmemo.h).  Which chunks depends on which predecessor cells are alive, and
// that the first forward run has just told: the terms of the candidates concerned are rebuilt on the host and written back into
// the candidate records.  Returns the number of rebuilt terms (> 0: the forward kernel has to run once more).
// (mat: the dense matrix that tells which cells are alive -- ln F after a forward run, ln V after a Viterbi run of the dense kernels;
//  fromLists: after a Viterbi run of the 47-state kernels, which keep no matrix, the values left at the donor sites instead)
int snippetCacheReplay(augx_decoder *d, augx_batch *b, int64_t &nPatched, bool fromLists, const double *mat) {
    nPatched = 0;
    const BatchView &V = b->V;
    const int n = V.nPieces, S = d->hostT.S;
    std::vector<int32_t> nPl((size_t)n);
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipMemcpy(nPl.data(), V.nPlanes, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<uint64_t> pIdx;
    std::vector<double> pTe;
    // what the replay of one piece reads, fetched from HBM; the replays themselves run side by side on host threads, 48 pieces
    // at a time (each holds its piece's candidate records: 0.23 GB per Mbp)
    struct PieceData {
        SnippetReplay R;
        std::vector<double> F, F0, ldV, rdV, col0;
        std::vector<uint8_t> plane;
        std::vector<int32_t> planeCls;
        std::vector<uint64_t> blkOff;
        std::vector<uint32_t> blkCnt;
        hipStream_t st = nullptr; // the copies of one window are queued on a stream of the piece's own and waited for once (a window is
                                  // dozens of small copies: one wait each, on the one default stream of the process, was most of the replay's time)
        // every window of the piece packed on the device (kGatherWindows) and fetched with one copy per array
        std::vector<GatherWin> wins;
        std::vector<Item> poolAll;
        std::vector<uint64_t> fxAll;
        std::vector<double> FAll;
        size_t nextWin = 0;
    };
    constexpr int GROUP_MAX = 128;
    static const int GROUP = [] { const char *e = getenv("AUGX_REPLAY_THREADS"); const int v = e ? atoi(e) : 48; return v < 1 ? 1 : v > GROUP_MAX ? GROUP_MAX : v; }(); // pieces replayed side by side
    {   // (made once per decoder: creating a stream takes milliseconds; augx_batch_sample_prepare shares the table)
        std::lock_guard<std::mutex> lk(g_copyMu);
        if ((int)d->copyStreams.size() < GROUP) d->copyStreams.resize((size_t)GROUP, nullptr);
        for (int i = 0; i < GROUP; i++)
            if (!d->copyStreams[i] && hipStreamCreateWithFlags(&d->copyStreams[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); d->copyStreams[i] = nullptr; }
    }
    hipStream_t *copySt = d->copyStreams.data();
    auto fetch = [&](int p, PieceData &D) -> int {
        SnippetReplay &R = D.R;
        const hipStream_t fst = D.st; // (the piece's own stream: queued, waited for once)
        auto cp = [fst](void *dst, const void *src, size_t bytes) { return fst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, fst) : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); };
        const int len = b->L.len[p];
        const int64_t o = b->L.off[p];
        R.t = &d->model->m.t; R.n = len; R.S = S; R.blk = V.blk; R.d = d->model->m.t.d; R.nPlanes = nPl[p]; R.dense = d->dense;
        if (!fromLists) { // (the rows of F a window reads are fetched with the window; the initial column now)
            D.F0.resize((size_t)S);
            HIP_TRY(cp(D.F0.data(), mat + (o + 1) * S, sizeof(double) * (size_t)S));
        } else { // (after a Viterbi run: what the trellis left at the donor sites of the short introns, and the initial column)
            int64_t lo2[2] = {0, 0};
            HIP_TRY(cp(lo2, V.listOffs + p, sizeof(int64_t) * 2));
            if (D.st) HIP_TRY(hipStreamSynchronize(D.st));
            const size_t cnt = (size_t)(lo2[1] - lo2[0]) * 3;
            D.ldV.resize(cnt + 1); D.rdV.resize(cnt + 1);
            if (cnt) {
                HIP_TRY(cp(D.ldV.data(), V.ldVal + lo2[0] * 3, sizeof(double) * cnt));
                HIP_TRY(cp(D.rdV.data(), V.rdVal + lo2[0] * 3, sizeof(double) * cnt));
            }
            D.col0.resize((size_t)S);
            for (int s2 = 0; s2 < S; s2++) D.col0[s2] = b->L.initKind[p] == 0 ? d->hostT.ln_init[s2] : (s2 == d->hostT.synch ? 0.0 : -INFINITY);
        }
        D.plane.resize((size_t)len);
        HIP_TRY(cp(D.plane.data(), V.gcPlane + o + 1, (size_t)len));
        D.planeCls.resize(MAXPL);
        HIP_TRY(cp(D.planeCls.data(), V.planeCls + (int64_t)p * MAXPL, sizeof(int32_t) * MAXPL));
        const int nBlocks = (len + V.blk - 1) / V.blk;
        const int64_t gb0 = o / V.blk;
        D.blkOff.resize((size_t)nBlocks * 2);
        D.blkCnt.resize((size_t)nBlocks * 2);
        HIP_TRY(cp(D.blkOff.data(), V.blkOff + gb0 * 2, sizeof(uint64_t) * D.blkOff.size()));
        HIP_TRY(cp(D.blkCnt.data(), V.blkCnt + gb0 * 2, sizeof(uint32_t) * D.blkCnt.size()));
        if (D.st) HIP_TRY(hipStreamSynchronize(D.st));
        R.F = nullptr; R.F0 = fromLists ? nullptr : D.F0.data(); R.ldVal = D.ldV.data(); R.rdVal = D.rdV.data(); R.col0 = D.col0.data();
        R.plane = D.plane.data(); R.planeCls = D.planeCls.data(); R.blkOff = D.blkOff.data(); R.blkCnt = D.blkCnt.data();
        R.items = nullptr; R.item0 = 0;
        R.blkPool.assign((size_t)nBlocks, -1);
        R.fxF.assign((size_t)nPl[p], {}); R.fxR.assign((size_t)nPl[p], {});
        const int nplP = nPl[p], blkSz = V.blk;
        PieceData *DP = &D;
        // what the windows read -- the candidate records of their blocks, the rows of the matrix, the slots of the intron content
        // prefix from d bases before each on -- is packed on the device and comes over in one copy per array (a window by itself was
        // dozens of small copies: 28 000 of them on a 100 Mbp genome, 0.4 s in the runtime)
        R.prefetch = [=](const std::vector<std::pair<int, int>> &tt) -> int {
            const hipStream_t cst = DP->st;
            SnippetReplay &R2 = DP->R;
            DP->wins.clear(); DP->nextWin = 0;
            int64_t pool = 0, fx = 0, fr = 0;
            for (auto w : tt) {
                int t0 = w.first < 0 ? 0 : w.first, t1 = w.second > len - 1 ? len - 1 : w.second;
                GatherWin g;
                memset(&g, 0, sizeof g);
                g.o = o; g.gb0 = gb0; g.b0 = t0 / blkSz; g.b1 = t1 / blkSz;
                g.r0 = t0 - R2.d - 2 > 0 ? t0 - R2.d - 2 : 0;
                g.g0 = g.r0; g.nSlots = t1 + 1 - g.r0 + 1; g.nPl = nplP;
                g.nRows = fromLists ? 0 : t1 - g.r0 + 1;
                g.poolOff = pool; g.fxOff = fx; g.fOff = fr;
                for (int q = g.b0; q <= g.b1; q++) pool += DP->blkCnt[(size_t)q * 2 + 1];
                fx += (int64_t)g.nPl * 2 * g.nSlots;
                fr += (int64_t)g.nRows * S;
                DP->wins.push_back(g);
            }
            if (DP->wins.empty()) return AUGX_OK;
            DP->poolAll.resize((size_t)pool + 1); DP->fxAll.resize((size_t)fx + 1); DP->FAll.resize((size_t)fr + 1);
            void *dW = nullptr, *dI = nullptr, *dX = nullptr, *dF = nullptr;
            auto freeAll = [&]() { if (dW) devFree(d, dW); if (dI) devFree(d, dI); if (dX) devFree(d, dX); if (dF) devFree(d, dF); };
            if (devMalloc(d, &dW, sizeof(GatherWin) * DP->wins.size()) != hipSuccess || devMalloc(d, &dI, sizeof(Item) * ((size_t)pool + 1)) != hipSuccess ||
                devMalloc(d, &dX, sizeof(uint64_t) * ((size_t)fx + 1)) != hipSuccess || (fr > 0 && devMalloc(d, &dF, sizeof(double) * ((size_t)fr + 1)) != hipSuccess)) {
                (void)hipGetLastError();
                freeAll();
                return AUGX_E_NOMEM;
            }
            hipError_t e = hipMemcpyAsync(dW, DP->wins.data(), sizeof(GatherWin) * DP->wins.size(), hipMemcpyHostToDevice, cst);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(kGatherWindows, dim3((unsigned)DP->wins.size()), dim3(256), 0, cst, V, (const GatherWin *)dW, (Item *)dI, (uint64_t *)dX, fromLists ? nullptr : mat, (double *)dF, S);
                e = hipGetLastError();
            }
            if (e == hipSuccess && pool > 0) e = hipMemcpyAsync(DP->poolAll.data(), dI, sizeof(Item) * (size_t)pool, hipMemcpyDeviceToHost, cst);
            if (e == hipSuccess && fx > 0) e = hipMemcpyAsync(DP->fxAll.data(), dX, sizeof(uint64_t) * (size_t)fx, hipMemcpyDeviceToHost, cst);
            if (e == hipSuccess && fr > 0) e = hipMemcpyAsync(DP->FAll.data(), dF, sizeof(double) * (size_t)fr, hipMemcpyDeviceToHost, cst);
            const hipError_t e2 = cst ? hipStreamSynchronize(cst) : hipDeviceSynchronize();
            freeAll();
            if (e != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); return AUGX_E_HIP; }
            return AUGX_OK;
        };
        R.fetch = [=](int ft0, int ft1) -> int { // the next window's part of what prefetch brought
            SnippetReplay &R2 = DP->R;
            if (DP->nextWin >= DP->wins.size()) return AUGX_E_ARG;
            const GatherWin &g = DP->wins[DP->nextWin++];
            {   // (the windows are handed out in the order prefetch was told them: this must be the one asked for)
                const int t0 = ft0 < 0 ? 0 : ft0, t1 = ft1 > len - 1 ? len - 1 : ft1;
                if (g.b0 != t0 / blkSz || g.b1 != t1 / blkSz || g.r0 != (t0 - R2.d - 2 > 0 ? t0 - R2.d - 2 : 0)) return AUGX_E_ARG;
            }
            if (DP->nextWin >= 2) { // (only the blocks of the window before hold anything: a 2 Mbp piece has 250 000 blocks, a window 300)
                const GatherWin &pw = DP->wins[DP->nextWin - 2];
                for (int q = pw.b0; q <= pw.b1; q++) R2.blkPool[(size_t)q] = -1;
            }
            size_t total = 0;
            for (int q = g.b0; q <= g.b1; q++) {
                if (DP->blkCnt[(size_t)q * 2 + 1]) R2.blkPool[(size_t)q] = (int64_t)total;
                total += DP->blkCnt[(size_t)q * 2 + 1];
            }
            R2.pool.assign(DP->poolAll.begin() + g.poolOff, DP->poolAll.begin() + g.poolOff + (int64_t)total);
            R2.pool.resize(total + 1);
            if (!fromLists) { R2.F = DP->FAll.data() + g.fOff; R2.fRow0 = g.r0; }
            R2.fx0 = g.r0;
            for (int pl = 0; pl < nplP; pl++)
                for (int rev = 0; rev < 2; rev++) {
                    std::vector<uint64_t> &dst = rev ? R2.fxR[pl] : R2.fxF[pl];
                    const uint64_t *src = DP->fxAll.data() + g.fxOff + ((int64_t)pl * 2 + rev) * g.nSlots;
                    dst.assign(src, src + g.nSlots);
                }
            return AUGX_OK;
        };
        return AUGX_OK;
    };
    std::vector<int> todo;
    for (int p = 0; p < n; p++)
        if (nPl[p] > 1) todo.push_back(p);
    const bool timing = getenv("AUGX_TIMING") != nullptr; // (developer aid)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tFetch = 0, tRun = 0;
    {   // GROUP host threads, each with a copy stream of its own, take the pieces one after the other (round 6: fixed groups of 48
        // left the last few pieces of a batch of 52 to run alone -- the replay of a 2 Mbp piece is 0.1 s on one thread)
        const double tf1 = timing ? now() : 0.0;
        std::atomic<size_t> nextPiece{0};
        std::atomic<int> rcRun{0};
        std::mutex outMu;
        auto worker = [&](int w) {
            (void)hipSetDevice(d->device);
            for (size_t k; (k = nextPiece.fetch_add(1)) < todo.size();) {
                std::unique_ptr<PieceData> D(new PieceData());
                D->st = copySt[w];
                int r0 = 0;
                try { r0 = fetch(todo[k], *D); if (!r0) r0 = D->R.run(); } catch (const std::exception &) { r0 = AUGX_E_NOMEM; }
                if (r0) { int z = 0; rcRun.compare_exchange_strong(z, r0); continue; }
                std::lock_guard<std::mutex> lk(outMu);
                for (const MemoPatch &mp : D->R.patches) { pIdx.push_back(mp.item); pTe.push_back(mp.te); }
            }
        };
        // (with a matrix to read -- forward algorithm, dense kernels -- a piece whose class steps crowd brings rows x S doubles for most
        //  of its length: 1.1 GB for 2 Mbp at S = 71; fewer of those side by side)
        const int nW = (int)std::min<size_t>(fromLists ? GROUP : 12, todo.size());
        std::vector<std::thread> th;
        for (int w = 1; w < nW; w++) {
            try { th.emplace_back(worker, w); } catch (const std::system_error &) { break; } // (no more threads to be had: fewer of them)
        }
        if (nW > 0) worker(0);
        for (auto &t2 : th) t2.join();
        if (timing) tRun += now() - tf1;
        if (rcRun.load()) { setLastError("augx: fetching the data of a snippet-cache window failed"); return rcRun.load(); }
    }
    nPatched = (int64_t)pIdx.size();
    if (timing) fprintf(stderr, "augx timing:       replay of %zu pieces: set up in %.3f s, tables fetched and windows replayed (%d pieces side by side) in %.3f s\n", todo.size(), tFetch, GROUP, tRun);
    if (pIdx.empty()) return AUGX_OK;
    void *dIdx = nullptr, *dTe = nullptr;
    if (devMalloc(d, &dIdx, sizeof(uint64_t) * pIdx.size()) != hipSuccess || devMalloc(d, &dTe, sizeof(double) * pTe.size()) != hipSuccess) {
        (void)hipGetLastError();
        if (dIdx) devFree(d, dIdx);
        setLastError("augx_batch_forward: out of device memory");
        return AUGX_E_NOMEM;
    }
    HIP_TRY(hipMemcpyAsync(dIdx, pIdx.data(), sizeof(uint64_t) * pIdx.size(), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(dTe, pTe.data(), sizeof(double) * pTe.size(), hipMemcpyHostToDevice, d->stream));
    hipLaunchKernelGGL(kPatchItems, dim3((unsigned)((pIdx.size() + 255) / 256)), dim3(256), 0, d->stream, V, (const uint64_t *)dIdx, (const double *)dTe, (int)pIdx.size());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream)); // (the host vectors and the two buffers go away)
    devFree(d, dIdx); devFree(d, dTe);
    return AUGX_OK;
}
// UTR states on pieces with several GC classes: the forward TSS windows and the acceptor sites whose value the reference computed
// under another class than the prep kernels took (tssProbsPlus: first asker; the aSSProb memo: assmemo.h) are rebuilt from the
// aliveness the first run left in `mat`.  rebuilt: some were -- the UTR descriptors (they hold values made from the site records)
// have been made again, and the dense kernel has to run once more.
int utrCachesReplay(augx_decoder *d, augx_batch *b, const double *mat, bool &rebuilt) {
    rebuilt = false;
    if (b->memoReplayed || getenv("AUGX_NO_ASSMEMO")) return AUGX_OK;
    b->memoReplayed = true;
    BatchView &W = b->V;
    const int n = W.nPieces;
    hipStream_t st = d->stream;
    const bool timing = getenv("AUGX_TIMING") != nullptr; // (developer aid)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = timing ? now() : 0.0;
    std::vector<int32_t> nPl((size_t)n);
    HIP_TRY(hipMemcpyAsync(nPl.data(), W.nPlanes, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int> todo;
    int64_t maxList = 0;
    for (int p = 0; p < n; p++)
        if (nPl[p] > 1) { todo.push_back(p); maxList = std::max(maxList, b->hListOffs[(size_t)p + 1] - b->hListOffs[(size_t)p]); }
    if (todo.empty()) return AUGX_OK;
    AssMemoReplay proto;
    proto.T = &d->hostT;
    proto.requesters();
    MemoReq rq;
    rq.n = proto.nReq;
    for (int r = 0; r < 8; r++) rq.s[r] = r < proto.nReq ? proto.reqS[r] : 0;
    const int64_t nSiteSlots = W.listCap + 8 * (int64_t)n + 8;
    int32_t *dQ = nullptr, *dChanged = nullptr;
    uint8_t *dAlive = nullptr, *dGate = nullptr;
    std::vector<void *> tmp;
    auto freeTmp = [&]() { for (void *q : tmp) devFree(d, q); tmp.clear(); };
    auto grab = [&](void **ptr, size_t bytes) -> bool { if (devMalloc(d, ptr, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return false; } tmp.push_back(*ptr); return true; };
    if (!grab((void **)&dQ, sizeof(int32_t) * (size_t)nSiteSlots) || !grab((void **)&dAlive, (size_t)nSiteSlots) || !grab((void **)&dGate, (size_t)W.N) || !grab((void **)&dChanged, sizeof(int32_t))) {
        freeTmp();
        setLastError("augx: out of device memory for the replay of the aSSProb memo");
        return AUGX_E_NOMEM;
    }
    HIP_TRY(hipMemsetAsync(dChanged, 0, sizeof(int32_t), st));
    const dim3 gridSites((unsigned)((maxList + 8 + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(kTssReplay, gridSites, dim3(256), 0, st, d->dT, W, mat, dChanged);
    hipLaunchKernelGGL(kMemoSites, gridSites, dim3(256), 0, st, d->dT, W, mat, rq, dQ, dAlive);
    hipLaunchKernelGGL(kMemoGates, dim3((unsigned)((W.N + 255) / 256)), dim3(256), 0, st, d->dT, W, rq, dGate);
    HIP_TRY(hipGetLastError());
    // (only what the multi-class pieces own comes over)
    struct PieceIn { std::vector<int32_t> q; std::vector<uint8_t> alive; std::shared_ptr<AssMemoReplay> Rp; std::vector<AssPatch> pt; std::vector<AssSwIn> sw; int extras = 0; };
    b->memoOf.assign((size_t)n, nullptr);
    std::vector<std::unique_ptr<PieceIn>> in;
    int32_t nTss = 0;
    HIP_TRY(hipMemcpyAsync(&nTss, dChanged, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    for (int p : todo) {
        in.emplace_back(new PieceIn());
        PieceIn &I = *in.back();
        const int len = b->L.len[p];
        const int64_t o = b->L.off[p], s0 = b->hListOffs[(size_t)p] + 8 * (int64_t)p, ns = b->hListOffs[(size_t)p + 1] - b->hListOffs[(size_t)p] + 8;
        I.Rp = std::make_shared<AssMemoReplay>();
        b->memoOf[(size_t)p] = I.Rp;
        I.q.resize((size_t)ns); I.alive.resize((size_t)ns); I.Rp->gateOwn.resize((size_t)len); I.Rp->planeOwn.resize((size_t)len);
        HIP_TRY(hipMemcpyAsync(I.q.data(), dQ + s0, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(I.alive.data(), dAlive + s0, (size_t)ns, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(I.Rp->gateOwn.data(), dGate + o + 1, (size_t)len, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(I.Rp->planeOwn.data(), W.gcPlane + o + 1, (size_t)len, hipMemcpyDeviceToHost, st));
    }
    std::vector<int32_t> nLa((size_t)n, 0);
    {   // length of every piece's LA list: the count at its last slot
        std::vector<uint32_t> tmpc((size_t)todo.size());
        for (size_t k = 0; k < todo.size(); k++)
            HIP_TRY(hipMemcpyAsync(&tmpc[k], W.cnt + fidx(b->L.off[todo[k]] + b->L.len[todo[k]], CNT_LA, NCNT), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t k = 0; k < todo.size(); k++) nLa[(size_t)todo[k]] = (int32_t)tmpc[k];
    }
    const double t1 = timing ? now() : 0.0;
    // the requests of every piece walked on host threads
    auto job = [&](size_t k) {
        PieceIn &I = *in[k];
        const int p = todo[k], nList = nLa[(size_t)p];
        AssMemoReplay &R = *I.Rp;
        R.T = &d->hostT; R.n = b->L.len[p]; R.plane = R.planeOwn.data(); R.gate = R.gateOwn.data();
        R.requesters();
        for (int li = 0; li < nList + 8 && li < (int)I.q.size(); li++)
            if (I.q[(size_t)li] >= 0) { R.siteQ.push_back(I.q[(size_t)li]); R.siteLi.push_back(li); R.siteAlive.push_back(I.alive[(size_t)li]); }
        R.run();
        I.extras = R.patches(nList, I.pt, I.sw);
    };
    {
        const size_t nThreads = std::min<size_t>(todo.size(), std::max(1u, std::min(32u, std::thread::hardware_concurrency())));
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        auto worker = [&]() { for (size_t k; (k = next.fetch_add(1)) < todo.size();) job(k); };
        try { for (size_t i = 1; i < nThreads; i++) th.emplace_back(worker); } catch (const std::system_error &) {}
        worker();
        for (auto &t2 : th) t2.join();
    }
    const double t2 = timing ? now() : 0.0;
    std::vector<AssPatch> pt;
    std::vector<int32_t> ptPiece;
    std::vector<AssSwIn> sw;
    long long calls = 0, flushes = 0, extras = 0;
    for (size_t k = 0; k < todo.size(); k++) {
        PieceIn &I = *in[k];
        for (AssPatch A : I.pt) { A.swOff += (int32_t)sw.size(); pt.push_back(A); ptPiece.push_back(todo[k]); }
        sw.insert(sw.end(), I.sw.begin(), I.sw.end());
        calls += I.Rp->calls; flushes += I.Rp->flushes; extras += I.extras;
    }
    if (!pt.empty()) {
        AssPatch *dPt = nullptr; int32_t *dPp = nullptr; AssSwIn *dSw = nullptr;
        if (b->laSwBuf) { devFree(d, b->laSwBuf); b->laSwBuf = nullptr; W.laSw = nullptr; b->nLaSw = 0; }
        if (!grab((void **)&dPt, sizeof(AssPatch) * pt.size()) || !grab((void **)&dPp, sizeof(int32_t) * pt.size()) || !grab((void **)&dSw, sizeof(AssSwIn) * (sw.size() + 1)) ||
            devMalloc(d, &b->laSwBuf, sizeof(LaSw) * (sw.size() + 1)) != hipSuccess) {
            (void)hipGetLastError();
            b->laSwBuf = nullptr;
            freeTmp();
            setLastError("augx: out of device memory for the replay of the aSSProb memo");
            return AUGX_E_NOMEM;
        }
        HIP_TRY(hipMemcpyAsync(dPt, pt.data(), sizeof(AssPatch) * pt.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dPp, ptPiece.data(), sizeof(int32_t) * pt.size(), hipMemcpyHostToDevice, st));
        if (!sw.empty()) HIP_TRY(hipMemcpyAsync(dSw, sw.data(), sizeof(AssSwIn) * sw.size(), hipMemcpyHostToDevice, st));
        W.laSw = (const LaSw *)b->laSwBuf;
        b->nLaSw = sw.size();
        HIP_TRY(hipMemcpyAsync(b->dV, &W, sizeof(BatchView), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(kAssPatch, dim3((unsigned)((pt.size() + 255) / 256)), dim3(256), 0, st, d->dT, W, dPt, dPp, (int)pt.size(), dSw, (LaSw *)b->laSwBuf);
        HIP_TRY(hipGetLastError());
    }
    rebuilt = nTss > 0 || !pt.empty();
    if (rebuilt) { // the descriptors hold the first candidates' terms, made from the site records: once more (same count, same buffer)
        HIP_TRY(hipMemsetAsync(&W.candAlloc->descs, 0, sizeof(W.candAlloc->descs), st));
        launchUtrDesc(d->blk, (unsigned)(W.N / (NT / 16)), st, d->dT, W);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st)); // (the host vectors and the temporary buffers go away)
    freeTmp();
    if (timing)
        fprintf(stderr, "augx timing:       UTR states, %zu pieces with several GC classes: %d TSS windows and %zu acceptor sites (%zu changes of value during the sweep; %lld past the end of a piece left) rebuilt from "
                        "the reference's caches; inputs gathered in %.3f s, %lld calls of the aSSProb memo walked (emptied %lld times) in %.3f s, values rebuilt in %.3f s\n",
                todo.size(), (int)nTss, pt.size(), sw.size(), extras, t1 - t0, calls, flushes, t2 - t1, now() - t2);
    return AUGX_OK;
}
} // namespace

extern "C" {

int augx_batch_forward(augx_decoder *d, augx_batch *b) {
    int rc = augx_batch_forward_launch(d, b);
    if (rc) return rc;
    if (b->nPlAlloc > 1 && !getenv("AUGX_NO_MEMO")) { // (a batch with a multi-class piece)
        int64_t nPatched = 0;
        bool sitesRebuilt = false;
        try {
            rc = snippetCacheReplay(d, b, nPatched, false, b->V.fwd);
            if (!rc && d->dense && d->hostT.utr) rc = utrCachesReplay(d, b, b->V.fwd, sitesRebuilt); // (nothing to do when the Viterbi run has replayed them already)
        } catch (const std::exception &e) { setLastError(std::string("augx_batch_forward: replay of the reference's caches: ") + e.what()); return AUGX_E_NOMEM; }
        if (rc) return rc;
        if ((nPatched > 0 || sitesRebuilt) && (rc = augx_batch_forward_launch(d, b))) return rc;
    }
    if (!b->evFwd) HIP_TRY(hipEventCreateWithFlags(&b->evFwd, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->evFwd, d->stream));
    return AUGX_OK;
}

int augx_batch_forward_cells(augx_decoder *d, augx_batch *b, int piece, double *out, double *ln_p) {
    if (!d || !b || piece < 0 || piece >= b->V.nPieces || !b->V.fwd) { setLastError("augx_batch_forward_cells: run augx_batch_forward first"); return AUGX_E_ARG; }
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    const int S = d->hostT.S;
    if (out) HIP_TRY(hipMemcpy(out, b->V.fwd + (b->L.off[piece] + 1) * S, sizeof(double) * (size_t)b->L.len[piece] * S, hipMemcpyDeviceToHost));
    if (ln_p) HIP_TRY(hipMemcpy(ln_p, b->V.lnFwd + piece, sizeof(double), hipMemcpyDeviceToHost));
    if (b->V.prof) { // developer aid (AUGX_PROF=1): where thread 0 of the forward kernel spent its time
        uint64_t h[12];
        HIP_TRY(hipMemcpy(h, b->V.prof + (int64_t)piece * 56, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "forward piece %d, ticks of thread 0:", piece);
        for (int k = 0; k < 11; k++) fprintf(stderr, " [%d]=%.2fM", k, h[k] / 1e6);
        fprintf(stderr, "  (0 loop, 1 A, 2 B others, 3 B run, 4 C max+sum, 5 C cells, 6 D others, 7 D run, 8 E max+sum, 9 E cells, 10 fence)\n");
    }
    return AUGX_OK;
}

// ---------------------------------------------------------------------------------------------------
// posterior sampling of state paths (host; reference NAMGene::getSampledPath, src/namgene.cc:367-426)
// ---------------------------------------------------------------------------------------------------
} // extern "C"

#include "sampler.h"

struct augx_sample_prep { SamplePiece P; };
namespace {
std::mutex g_fPoolMu;
std::vector<std::pair<size_t, double *>> g_fPool; // host buffers of forward matrices, kept for the next piece: at most 36 and FPOOL_CAP_BYTES, released with the last decoder
size_t g_fPoolBytes = 0;
constexpr size_t FPOOL_CAP_BYTES = (size_t)12 << 30;
}
// The other host buffers of a prepared piece (candidate records 230 B, signal terms 64 B per base, ...) go round with the
// augx_sample_prep object itself: its vectors keep their memory for the next piece (at most 36 objects wait here).
namespace {
std::vector<augx_sample_prep *> g_prepPool;
}
// a forward matrix on the host: 2 MB pages where the kernel grants them (a piece's matrix is hundreds of MB; in 4 KB pages its
// first touch and its release were each a tenth of a second of page-table work per GB)
static double *hugeAlloc(size_t n) {
    void *q = nullptr;
    const size_t bytes = ((n * sizeof(double) + ((size_t)2 << 20) - 1) >> 21) << 21;
    if (posix_memalign(&q, (size_t)2 << 20, bytes) != 0 || !q) throw std::bad_alloc();
    (void)madvise(q, bytes, MADV_HUGEPAGE);
    return (double *)q;
}
void augx_release_host_pools() { // (augx_decoder_destroy, when the last decoder goes)
    std::lock_guard<std::mutex> lk(g_fPoolMu);
    for (auto &e : g_fPool) free(e.second);
    g_fPool.clear();
    g_fPoolBytes = 0;
    for (augx_sample_prep *h : g_prepPool) delete h;
    g_prepPool.clear();
}

extern "C" {

augx_rand *augx_rand_create(unsigned seed) { return new augx_rand(seed); }
int augx_rand_next(augx_rand *r) { return r ? r->next() : -1; }
void augx_rand_skip(augx_rand *r, int64_t n) { if (r && n > 0) r->skip(n); }
void augx_rand_destroy(augx_rand *r) { delete r; }

// everything the host sampler reads of one piece, fetched from HBM, and the sampler's look-up tables: independent of the draws,
// so a run prepares the pieces ahead of the sampling, on other threads (sharded.cc)
int augx_batch_sample_prepare(augx_decoder *d, augx_batch *b, int piece, augx_sample_prep **out) {
    if (!d || !b || !out || piece < 0 || piece >= b->V.nPieces || !b->V.fwd) {
        setLastError("augx_batch_sample: bad argument (run augx_batch_forward first)");
        return AUGX_E_ARG;
    }
    *out = nullptr;
    HIP_TRY(hipSetDevice(d->device));
    // Everything but the forward matrix is there before the forward kernel runs (the candidate records are final when
    // augx_batch_forward returns: it waits for the replay of the snippet cache): it is copied while the kernel runs, on a stream
    // that does not wait for the decoder's, and the matrix when the kernel has finished.
    hipStream_t cst = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_copyMu);
        if (d->copyStreams.empty()) d->copyStreams.assign(48, nullptr);
        hipStream_t &slotSt = d->copyStreams[(size_t)piece % d->copyStreams.size()];
        if (!slotSt && hipStreamCreateWithFlags(&slotSt, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); slotSt = nullptr; }
        cst = slotSt;
    }
    if (!cst) { if (b->evFwd) HIP_TRY(hipEventSynchronize(b->evFwd)); else HIP_TRY(hipStreamSynchronize(d->stream)); } // (no stream to be had: after the kernel, on the default stream)
    // (the copy stream has no order of its own against the decoder's: what it reads is written by the kernels of the decode,
    //  whose end is ev[3]; the terms the replay rebuilds are in place when augx_batch_forward returns, it waits for them)
    else if (b->decoded && b->ev[3]) HIP_TRY(hipStreamWaitEvent(cst, b->ev[3], 0));
    auto cp = [cst](void *dst, const void *src, size_t bytes) { return cst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cst) : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); };
    auto flush = [cst]() { return cst ? hipStreamSynchronize(cst) : hipSuccess; };
    const BatchView &V = b->V;
    const augx_tables &t = d->model->m.t;
    try {
    std::unique_ptr<augx_sample_prep> H;
    {
        std::lock_guard<std::mutex> lk(g_fPoolMu);
        if (!g_prepPool.empty()) { H.reset(g_prepPool.back()); g_prepPool.pop_back(); }
    }
    if (!H) H.reset(new augx_sample_prep());
    SamplePiece &P = H->P;
    // (declared after H, so it runs BEFORE H is freed on every early return: copies still in flight on the non-blocking stream must
    //  not land in host buffers that have been given back; on the way out of a successful call the stream is idle already)
    struct CopyDrain { hipStream_t s; ~CopyDrain() { if (s) (void)hipStreamSynchronize(s); } } copyDrain{cst};
    {   // (an object that comes round: everything a piece sets only under a condition goes back to its default)
        P.plane.clear(); P.planeCls.clear(); P.uh.reset(); P.dense = false; P.hT = nullptr; P.hB = nullptr; P.hp = 0;
        P.memo = nullptr; P.memoOwner.reset(); P.vitPath.clear(); P.memoVitDiffs = 0;
        P.igS = -1; P.termKind = 0; P.anyNuc = true; P.prepared = false; P.item0 = 0;
        P.buildSeconds = 0; P.nBuilt = P.nStops = P.nVar = 0; P.tkChain = P.tkVar = P.tkTail = 0;
    }
    P.t = &t; P.S = t.S; P.n = b->L.len[piece]; P.blk = V.blk;
    const int n = P.n, S = P.S;
    const int64_t o = b->L.off[piece];
    int32_t cls = 0, nPl = 1;
    std::unique_ptr<CopySlot> slot(new CopySlot(piece));
    HIP_TRY(cp(&cls, V.cls + piece, 4));
    HIP_TRY(cp(&nPl, V.nPlanes + piece, 4));
    HIP_TRY(flush());
    P.cls0 = cls; P.nPlanes = nPl;
    {   // the host copy of the piece's forward matrix (hundreds of MB): buffers go round between the pieces of a run instead of
        // being mapped, faulted in and unmapped once per piece
        const size_t need = (size_t)n * S;
        std::pair<size_t, double *> got{0, nullptr};
        {
            std::lock_guard<std::mutex> lk(g_fPoolMu);
            for (size_t i = 0; i < g_fPool.size(); i++)
                if (g_fPool[i].first >= need && g_fPool[i].first <= need + need / 2) { got = g_fPool[i]; g_fPoolBytes -= got.first * sizeof(double); g_fPool.erase(g_fPool.begin() + (long)i); break; }
        }
        if (!got.second) got = {need, hugeAlloc(need)};
        const size_t cap = got.first;
        P.Fown = std::shared_ptr<double>(got.second, [cap](double *q) {
            std::lock_guard<std::mutex> lk(g_fPoolMu);
            if (g_fPool.size() < 36 && g_fPoolBytes + cap * sizeof(double) <= FPOOL_CAP_BYTES) { g_fPool.push_back({cap, q}); g_fPoolBytes += cap * sizeof(double); } else free(q);
        });
    }
    P.F = P.Fown.get();
    P.sig.resize((size_t)n * NSIG);
    HIP_TRY(cp(P.sig.data(), V.sig + (o + 1) * NSIG, sizeof(double) * P.sig.size()));
    if (nPl > 1) {
        P.plane.resize((size_t)n);
        HIP_TRY(cp(P.plane.data(), V.gcPlane + o + 1, (size_t)n));
        P.planeCls.resize(MAXPL);
        HIP_TRY(cp(P.planeCls.data(), V.planeCls + (int64_t)piece * MAXPL, sizeof(int32_t) * MAXPL));
    }
    const int nBlocks = (n + V.blk - 1) / V.blk;
    const int64_t gb0 = o / V.blk;
    P.blkOff.resize((size_t)nBlocks * 2); P.blkCnt.resize((size_t)nBlocks * 2);
    HIP_TRY(cp(P.blkOff.data(), V.blkOff + gb0 * 2, sizeof(uint64_t) * P.blkOff.size()));
    HIP_TRY(cp(P.blkCnt.data(), V.blkCnt + gb0 * 2, sizeof(uint32_t) * P.blkCnt.size()));
    HIP_TRY(flush());
    // (K2a hands out the candidate ranges of the blocks through an atomic counter: a piece's candidates lie wherever its work
    //  groups were served, so the span between its lowest and highest range is fetched)
    uint64_t lo = ~0ull, hi = 0;
    for (int q = 0; q < nBlocks; q++) {
        const uint64_t a = P.blkOff[(size_t)q * 2 + 1], c = P.blkCnt[(size_t)q * 2 + 1];
        if (c == 0) continue;
        lo = a < lo ? a : lo;
        hi = a + c > hi ? a + c : hi;
    }
    if (hi <= lo) lo = hi = 0;
    if (hi > b->nItems && b->nItems) { setLastError("augx_batch_sample: candidate ranges outside the candidate buffer"); return AUGX_E_HIP; }
    uint64_t own = 0;
    for (int q = 0; q < nBlocks; q++) own += P.blkCnt[(size_t)q * 2 + 1];
    if (hi - lo <= 4 * own + 65536) { // the piece's ranges lie close together: one copy of the span
        P.item0 = lo;
        P.items.resize((size_t)(hi - lo) + 1);
        if (hi > lo) HIP_TRY(cp(P.items.data(), V.items + lo, sizeof(Item) * (size_t)(hi - lo)));
    } else { // scattered over the batch's buffer (many pieces decoded side by side): run by run of contiguous blocks, packed
        P.item0 = 0;
        P.items.resize((size_t)own + 1);
        uint64_t w = 0;
        for (int q = 0; q < nBlocks;) {
            const uint64_t a0 = P.blkOff[(size_t)q * 2 + 1];
            uint64_t len = P.blkCnt[(size_t)q * 2 + 1];
            int q2 = q + 1;
            while (q2 < nBlocks && (P.blkCnt[(size_t)q2 * 2 + 1] == 0 || P.blkOff[(size_t)q2 * 2 + 1] == a0 + len)) { len += P.blkCnt[(size_t)q2 * 2 + 1]; q2++; }
            if (len) HIP_TRY(cp(P.items.data() + w, V.items + a0, sizeof(Item) * (size_t)len));
            uint64_t at = w;
            for (int k = q; k < q2; k++) { P.blkOff[(size_t)k * 2 + 1] = at; at += P.blkCnt[(size_t)k * 2 + 1]; }
            w += len;
            q = q2;
        }
    }
    P.termKind = b->L.termKind[piece];
    if (d->dense) { // the host view of the piece the UTR exon candidates are evaluated from (sampler.h: SamplePiece::UtrHost)
        P.dense = true;
        P.uh.reset(new SamplePiece::UtrHost());
        SamplePiece::UtrHost &U = *P.uh;
        fillDevTablesScalars(t, U.T);
        for (auto &sp : tableSpans(t, U.T)) *sp.dst = (TabPtr)sp.src; // (host tables)
        const int64_t slots = b->L.off[piece + 1] - o, ch0 = o / CHUNK, nch = slots / CHUNK;
        memset(&U.B, 0, sizeof U.B);
        U.off = {0, slots}; U.len = {n}; U.initKind = {b->L.initKind[piece]}; U.termKind = {b->L.termKind[piece]};
        U.chunkPiece.assign((size_t)nch, 0);
        U.cls = {cls}; U.nPlanes = {nPl};
        U.planeCls.assign(MAXPL, 0);
        HIP_TRY(cp(U.planeCls.data(), V.planeCls + (int64_t)piece * MAXPL, sizeof(int32_t) * MAXPL));
        U.code.resize((size_t)slots); U.gcPlane.resize((size_t)slots);
        HIP_TRY(cp(U.code.data(), V.code + o, (size_t)slots));
        HIP_TRY(cp(U.gcPlane.data(), V.gcPlane + o, (size_t)slots));
        U.cnt.resize((size_t)slots * NCNT); U.ucnt.resize((size_t)slots * NUCNT); U.ufx.resize((size_t)slots * NUFX);
        HIP_TRY(cp(U.cnt.data(), V.cnt + ch0 * NCNT * CHUNK, sizeof(uint32_t) * U.cnt.size())); // (chunk-major: a piece's chunks are contiguous)
        HIP_TRY(cp(U.ucnt.data(), V.ucnt + ch0 * NUCNT * CHUNK, sizeof(uint32_t) * U.ucnt.size()));
        HIP_TRY(cp(U.ufx.data(), V.ufx + ch0 * NUFX * CHUNK, sizeof(uint64_t) * U.ufx.size()));
        U.usig.resize((size_t)slots * NUSIG); U.sigAll.resize((size_t)slots * NSIG);
        HIP_TRY(cp(U.usig.data(), V.usig + o * NUSIG, sizeof(double) * U.usig.size()));
        HIP_TRY(cp(U.sigAll.data(), V.sig + o * NSIG, sizeof(double) * U.sigAll.size()));
        std::vector<int64_t> lo2(2);
        HIP_TRY(cp(lo2.data(), b->dListOffs + piece, sizeof(int64_t) * 2));
        HIP_TRY(flush());
        const int64_t nEnt = lo2[1] - lo2[0];
        U.listOffs = {0, nEnt};
        const USite *src[6] = {V.tfSite, V.laSite, V.fsSite, V.lrSite, V.tmSite, V.rtSite};
        for (int k = 0; k < 6; k++) {
            U.sites[k].resize((size_t)(nEnt > 0 ? nEnt : 1));
            if (nEnt > 0) HIP_TRY(cp(U.sites[k].data(), src[k] + lo2[0], sizeof(USite) * (size_t)nEnt));
        }
        if (b->laSwBuf && b->nLaSw) { // (acceptor sites whose value changes during the sweep: the records index the batch's table)
            U.laSw.resize(b->nLaSw);
            HIP_TRY(cp(U.laSw.data(), b->laSwBuf, sizeof(LaSw) * b->nLaSw));
            HIP_TRY(flush());
        }
        BatchView &HB = U.B;
        HB.laSw = U.laSw.empty() ? nullptr : U.laSw.data();
        HB.nPieces = 1; HB.N = slots; HB.nChunks = (int)nch;
        HB.off = U.off.data(); HB.len = U.len.data(); HB.initKind = U.initKind.data(); HB.termKind = U.termKind.data(); HB.chunkPiece = U.chunkPiece.data();
        HB.cls = U.cls.data(); HB.nPlanes = U.nPlanes.data(); HB.planeCls = U.planeCls.data(); HB.nPl = 1;
        HB.listOffs = U.listOffs.data(); HB.listCap = nEnt;
        HB.code = U.code.data(); HB.gcPlane = U.gcPlane.data(); HB.cnt = U.cnt.data(); HB.ucnt = U.ucnt.data(); HB.ufx = U.ufx.data();
        HB.usig = U.usig.data(); HB.sig = U.sigAll.data();
        HB.tfSite = U.sites[0].data(); HB.laSite = U.sites[1].data(); HB.fsSite = U.sites[2].data(); HB.lrSite = U.sites[3].data();
        HB.tmSite = U.sites[4].data(); HB.rtSite = U.sites[5].data();
        HB.blk = V.blk;
        P.hT = &U.T; P.hB = &U.B; P.hp = 0;
        if ((size_t)piece < b->memoOf.size() && b->memoOf[(size_t)piece] && nPl > 1) {
            // the aSSProb memo goes on from where the sweep left it: through the back-tracking of the Viterbi path, then the sampled
            // paths (sampler.h: memoStep) -- on a copy, so that the piece can be sampled again
            auto mc = std::make_shared<AssMemoReplay>(*b->memoOf[(size_t)piece]);
            mc->T = &U.T; mc->plane = mc->planeOwn.data(); mc->gate = mc->gateOwn.data();
            int32_t cnt = 0;
            HIP_TRY(cp(&cnt, V.pathCount + piece, sizeof(int32_t)));
            HIP_TRY(flush());
            std::vector<int32_t> rec((size_t)(cnt > 0 ? cnt : 0) * 3);
            if (cnt > 0) { HIP_TRY(cp(rec.data(), V.pathRec + (b->L.off[piece] / 8 + 64 * (int64_t)piece) * 3, sizeof(int32_t) * rec.size())); HIP_TRY(flush()); }
            for (int i = cnt - 1; i >= 0; i--) P.vitPath.push_back({rec[(size_t)i * 3], rec[(size_t)i * 3 + 1], (int16_t)rec[(size_t)i * 3 + 2], (int16_t)t.state_type[rec[(size_t)i * 3 + 2]]});
            P.memo = mc.get();
            P.memoOwner = mc;
        }
    }
    {
        std::vector<uint8_t> code((size_t)n);
        HIP_TRY(cp(code.data(), V.code + o + 1, (size_t)n));
        HIP_TRY(flush());
        P.anyNuc = false;
        for (int q = 0; q < n && !P.anyNuc; q++) P.anyNuc = code[q] < 4;
    }
    slot.reset(); // (the wait for the kernel holds no slot)
    if (b->evFwd) HIP_TRY(hipEventSynchronize(b->evFwd)); else HIP_TRY(hipStreamSynchronize(d->stream));
    slot.reset(new CopySlot(piece));
    HIP_TRY(cp(P.Fown.get(), V.fwd + (o + 1) * S, sizeof(double) * (size_t)n * S));
    HIP_TRY(flush());
    slot.reset();
    prepareStops(P);
    *out = H.release();
    return AUGX_OK;
    } catch (const std::exception &e) { setLastError(std::string("augx_batch_sample: ") + e.what() + " (out of host memory?)"); return AUGX_E_NOMEM; }
}

void augx_sample_prep_destroy(augx_sample_prep *h) {
    if (!h) return;
    h->P.Fown.reset(); // (the forward matrix goes back to its own pool)
    h->P.F = nullptr;
    h->P.uh.reset();
    h->P.memo = nullptr; h->P.memoOwner.reset();
    {
        std::lock_guard<std::mutex> lk(g_fPoolMu);
        if (g_prepPool.size() < 36) { g_prepPool.push_back(h); return; }
    }
    delete h;
}

int augx_sample_prep_run(augx_sample_prep *h, int n_samples, augx_rand *R, augx_path *out) {
    if (!h || !R || !out || n_samples < 0) { setLastError("augx_sample_prep_run: bad argument"); return AUGX_E_ARG; }
    for (int i = 0; i < n_samples; i++) { out[i].states = nullptr; out[i].n_states = 0; out[i].status = 0; out[i].ln_viterbi = 0; }
    std::vector<std::vector<augx_state>> paths;
    std::vector<int> status;
    const double gen0 = R->refillSeconds;
    try { samplePaths(h->P, n_samples, *R, paths, status); }
    catch (const std::exception &e) { setLastError(std::string("augx_batch_sample: ") + e.what() + " (out of host memory?)"); return AUGX_E_NOMEM; }
    if (getenv("AUGX_TIMING_SAMPLER")) // (developer aid)
        fprintf(stderr, "augx timing:     sampler, piece of %d bases: generator %.4f s, %ld option lists built in %.4f s, %ld stops passed, %ld draws at other states; Mticks: chain runs %.1f, other states %.1f, paths put together %.1f%s\n", h->P.n,
                R->refillSeconds - gen0, h->P.nBuilt, h->P.buildSeconds, h->P.nStops, h->P.nVar, h->P.tkChain / 1e6, h->P.tkVar / 1e6, h->P.tkTail / 1e6,
                h->P.memo ? (std::string("; aSSProb memo carried on: ") + std::to_string(h->P.memo->calls) + " calls so far, emptied " + std::to_string(h->P.memo->flushes) + " times; candidates of the Viterbi path's UTR exon steps valued under another class by the back-tracking: " + std::to_string(h->P.memoVitDiffs)).c_str() : "");
    for (int it = 0; it < n_samples; it++) {
        out[it].status = status[it];
        if (status[it] != AUGX_OK) continue;
        const std::vector<augx_state> &m2 = paths[it];
        out[it].states = (augx_state *)malloc(sizeof(augx_state) * (m2.size() ? m2.size() : 1));
        if (!out[it].states) { setLastError("augx_batch_sample: out of host memory"); return AUGX_E_NOMEM; }
        memcpy(out[it].states, m2.data(), sizeof(augx_state) * m2.size());
        out[it].n_states = (int32_t)m2.size();
    }
    return AUGX_OK;
}

int augx_batch_sample(augx_decoder *d, augx_batch *b, int piece, int n_samples, augx_rand *R, augx_path *out) {
    if (!R || !out || n_samples < 0) { setLastError("augx_batch_sample: bad argument"); return AUGX_E_ARG; }
    augx_sample_prep *h = nullptr;
    int rc = augx_batch_sample_prepare(d, b, piece, &h);
    if (!rc) rc = augx_sample_prep_run(h, n_samples, R, out);
    augx_sample_prep_destroy(h);
    return rc;
}

int augx_decode_batch(augx_decoder *d, const augx_piece *pieces, int n, augx_path *out) {
    // AUGX_TIMING=1 (developer aid): the phases of the batch on stderr
    const bool timing = getenv("AUGX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = timing ? now() : 0.0;
    augx_batch *b = nullptr;
    int rc = augx_batch_create(d, pieces, n, &b);
    if (rc) return rc;
    const double t1 = timing ? now() : 0.0;
    rc = augx_batch_decode(d, b);
    if (!rc && timing) (void)hipStreamSynchronize(d->stream);
    const double t2 = timing ? now() : 0.0;
    if (!rc) rc = augx_batch_paths(d, b, out);
    const double t3 = timing ? now() : 0.0;
    float kPrep = 0, kTrellis = 0, kBack = 0;
    int planes = 0;
    long long nItems = 0;
    if (timing && !rc) { (void)augx_batch_kernel_ms(d, b, &kPrep, &kTrellis, &kBack); planes = b->V.nPl; nItems = (long long)b->nItems; }
    augx_batch_destroy(b);
    if (timing) {
        int64_t bases = 0;
        for (int i = 0; i < n; i++) bases += pieces[i].len;
        size_t freeB = 0, totalB = 0;
        (void)hipMemGetInfo(&freeB, &totalB);
        fprintf(stderr, "augx timing:   batch on device %d: %d pieces, %lld bases: create + upload %.3f s, decode %.3f s (on the device: prep %.0f ms, trellis passes%s %.0f ms, back-trace %.0f ms; %d plane(s), %lld candidates), paths %.3f s, destroy %.3f s (device memory free %.1f of %.1f GB)\n",
                d->device, n, (long long)bases, t1 - t0, t2 - t1, kPrep, planes > 1 && d->exactMulti ? " + replay + second run" : "", kTrellis, kBack, planes, nItems, t3 - t2, now() - t3, freeB / 1e9, totalB / 1e9);
    }
    return rc;
}

void augx_path_free(augx_path *p) {
    if (p && p->states) { free(p->states); p->states = nullptr; p->n_states = 0; }
}
}
