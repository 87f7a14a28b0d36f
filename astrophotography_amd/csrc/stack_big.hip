// stack_big.hip - stacks of 129 .. 512 frames (APGPU_MAX_STACK): exact sigma-clipped statistics / median of a column that
// no longer fits the registers.  Same semantics and outputs as the register kernels (stack_kernels.h):
//   astropy.stats.sigma_clipped_stats(cube, axis=0)   astropy/stats/sigma_clipping.py:298-383, 924-937
//   ccdproc.combine(median / mad_std, one pass)       reference call site scripts/ap_combine_darks.py:394-420
//   np.nanmedian(cube, axis=0)
// ccdproc and astropy take any N (ap_combine_darks.py:411-420 combines whatever the directory holds).
//
// One wavefront per workgroup, one pixel per lane; the lane's column lives in LDS as [slot][lane] (bank = lane: every
// access is conflict-free, nothing is shared between lanes), 64 KB for 256 slots, 128 KB for 512 (one or two
// workgroups per CU).  The column is sorted by register-sized pieces:
//   1. every 128-frame chunk is loaded, calibrated and sorted in registers by the ordinary 128-slot path
//      (load_sorted_column: same fused calibration, guards and compile-time Batcher network) and parked in LDS;
//   2. sorted runs are merged pairwise with the bitonic scheme: one streaming pass of compare-exchanges
//      (col[i], col[len-1-i]) leaves the lower and the upper half as bitonic sequences; halves longer than 128 get one
//      streaming half-cleaner pass per level; every 128-element bitonic piece is then sorted in registers by the
//      7-layer bitonic merge network (448 compare-exchanges, static indices).  LDS traffic per 256-merge: 2 x 256
//      element reads + writes instead of 4 per compare-exchange.
//   3. the clipping passes run on the LDS column with run-time cursors (reduce_and_store_rich, the rich kernels' code).
// HBM traffic is the algorithmic minimum (every frame row read once, coalesced, 256 B per wavefront and frame).
#include "stack_kernels.h"

namespace apgpu_stack {

using namespace apgpu;

// 7-layer bitonic merge network of 128 wires (sorts any bitonic sequence ascending), generated at compile time.
struct BitonicNet128 {
    CE ce[448];
};

constexpr BitonicNet128 make_bitonic128()
{
    BitonicNet128 net{};
    int c = 0;
    for (int k = 64; k >= 1; k /= 2)
        for (int i = 0; i < 128; i++)
            if ((i & k) == 0) {
                net.ce[c].a = (unsigned char)i;
                net.ce[c].b = (unsigned char)(i + k);
                c++;
            }
    return net;
}

template <int BASE, int... I>
__device__ __forceinline__ void bitonic_chunk(float (&v)[128], std::integer_sequence<int, I...>)
{
    constexpr BitonicNet128 net = make_bitonic128();
    (cmpx(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int BASE>
__device__ __forceinline__ void bitonic_from(float (&v)[128])
{
    if constexpr (BASE < 448) {
        bitonic_chunk<BASE>(v, std::make_integer_sequence<int, 64>{});
        bitonic_from<BASE + 64>(v);
    }
}

constexpr int kBigLanes = 64;

// (col[lo + i], col[lo + j(i)]) <- (min, max) for i in [0, half): j = len - 1 - i (MIRROR: first layer of the merge of two
// sorted runs) or j = i + half (half-cleaner of a bitonic sequence).  8 pairs in flight per trip.
template <bool MIRROR>
__device__ __forceinline__ void lds_exchange_pass(float *col, int lo, int len)
{
    const int half = len / 2;
    for (int i0 = 0; i0 < half; i0 += 8) {
        float x[8], y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = lo + i0 + k;
            const int j = MIRROR ? lo + len - 1 - (i0 + k) : i + half;
            x[k] = col[i * kBigLanes];
            y[k] = col[j * kBigLanes];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = lo + i0 + k;
            const int j = MIRROR ? lo + len - 1 - (i0 + k) : i + half;
            cmpx(x[k], y[k]);
            col[i * kBigLanes] = x[k];
            col[j * kBigLanes] = y[k];
        }
    }
}

// Sorts the bitonic sequence col[lo .. lo + len) ascending (len = 128 * 2^k).
__device__ __forceinline__ void lds_bitonic_sort(float *col, int lo, int len)
{
    for (int span = len; span > 128; span /= 2)
        for (int s0 = lo; s0 < lo + len; s0 += span) lds_exchange_pass<false>(col, s0, span);
#pragma unroll 1
    for (int s0 = lo; s0 < lo + len; s0 += 128) {
        float v[128];
#pragma unroll
        for (int i = 0; i < 128; i++) v[i] = col[(s0 + i) * kBigLanes];
        bitonic_from<0>(v);
#pragma unroll
        for (int i = 0; i < 128; i++) col[(s0 + i) * kBigLanes] = v[i];
    }
}

// redo_count == nullptr: workgroup b reduces the 64 pixels [64 b, 64 b + 64).  Otherwise: the PIXELS the chunked fast
// kernel (stack_chunks.hip) was not sure about - *redo_count entries of redo_list, pixel indices - are shared out over the grid,
// 64 listed pixels per workgroup and trip (a gather: every lane loads its own pixel's column).  ws != nullptr: count and list
// live in the caller's workspace (stack_kernels.h, "workspace"): the last workgroup to finish clears the counter it used and
// books the statistics.
template <int NP, typename RawT, bool CALIB, bool MEDIAN>
__global__ __launch_bounds__(kBigLanes, 1) void stack_big_kernel(const StackParams prm, int32_t *redo_count, const int32_t *redo_list, int32_t *ws)
{
    extern __shared__ float col_all[];                      // [NP][64]
    __shared__ FrameScalars<128> fs;
    const int lane = threadIdx.x;
    float *const col = col_all + lane;
    const bool redo = redo_count != nullptr;
    const int64_t nlisted = redo ? __builtin_amdgcn_readfirstlane(ws_load(redo_count)) : 0;
    const int64_t nitems = redo ? (nlisted + kBigLanes - 1) / kBigLanes : (int64_t)gridDim.x;
#pragma unroll 1
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
    // (list mode: base 0 and the lane's own pixel as its offset; pixel indices fit an int - chunks_eligible)
    const int64_t base = redo ? 0 : item * kBigLanes;
    const bool listed = redo && item * kBigLanes + lane < nlisted;
    const int plane = redo ? (listed ? redo_list[item * kBigLanes + lane] : 0) : lane;
    const int64_t p = base + plane;
    const bool live = redo ? listed : p < prm.P;            // dead lanes of the last workgroup still stage and vote
    __syncthreads();                                        // (redo loop) the previous item is done with LDS
    int n = 0;
#pragma unroll 1
    for (int c0 = 0; c0 < NP; c0 += 128) {
        StackParams q = prm;                                // the chunk as a stack of its own
        q.frames = static_cast<const RawT *>(prm.frames) + (int64_t)c0 * prm.stride;
        q.N = prm.N - c0 < 128 ? prm.N - c0 : 128;
        if (prm.exp_ratio) q.exp_ratio = prm.exp_ratio + c0;
        if (prm.pedestal) q.pedestal = prm.pedestal + c0;
        float v[128];
        int nc = 0;
        if (q.N > 0) {
            __syncthreads();                                // the previous chunk is done with the staged scalars
            stage_frame_scalars<128>(q, fs);
            if (live) nc = load_sorted_column<128, RawT, CALIB, !MEDIAN, false, 0>(q, fs, base, plane, v);   // a chunk may hold 1..128 frames
        }
        if (!(q.N > 0 && live)) {
#pragma unroll
            for (int i = 0; i < 128; i++) v[i] = __builtin_inff();
        }
#pragma unroll
        for (int i = 0; i < 128; i++) col[(c0 + i) * kBigLanes] = v[i];
        n += nc;
    }
    // merge the sorted 128-runs: 2 -> 256, (2 x 256) -> 512
    for (int len = 256; len <= NP; len *= 2)
        for (int lo = 0; lo < NP; lo += len) {
            lds_exchange_pass<true>(col, lo, len);
            lds_bitonic_sort(col, lo, len / 2);
            lds_bitonic_sort(col, lo + len / 2, len / 2);
        }
    if (live) {
        if constexpr (MEDIAN) {
            const float m1 = col_read<NP, kBigLanes>(col, (n - 1) >> 1);
            const float m2 = col_read<NP, kBigLanes>(col, n >> 1);
            const double med = ((double)m1 + (double)m2) / 2.0;
            if (prm.median) prm.median[p] = n > 0 ? (float)med : __builtin_nanf("");
            if (prm.count) prm.count[p] = n;
        } else {
            float dummy[1] = {0.f};
            reduce_and_store_rich<NP, kBigLanes, false, 1>(prm, dummy, n, p, col);
        }
    }
    }
    if (ws && lane == 0) {
        // (one wavefront per workgroup: no barrier needed) every workgroup has read the count before it arrives here
        const int arrived = __hip_atomic_fetch_add(ws + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (int)gridDim.x - 1) {
            unsigned long long *const stats = reinterpret_cast<unsigned long long *>(ws + kWsStats);
            atomicAdd(stats + 0, 1ull);
            atomicAdd(stats + 1, (unsigned long long)prm.P);
            atomicAdd(stats + 2, (unsigned long long)nlisted);
            ws[0] = 0;
            ws[2] = 0;
        }
    }
}

template <int NP, typename RawT, bool CALIB, bool MEDIAN>
static int launch_big_one(const StackParams &prm, hipStream_t st, char *describe, int32_t *redo, const int32_t *list, int32_t *ws)
{
    if (describe) {
        snprintf(describe, 256, "stack_big_kernel<%d, %s, %s, %s>", NP, sizeof(RawT) == 2 ? "unsigned short" : "float",
                 CALIB ? "true" : "false", MEDIAN ? "true" : "false");
        return APGPU_OK;
    }
    int64_t grid = (prm.P + kBigLanes - 1) / kBigLanes;
    if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
    if (redo && grid > 2048) grid = 2048;                    // the redo list is short: a fixed grid walks it
    const size_t lds = (size_t)NP * kBigLanes * sizeof(float);
    auto kern = stack_big_kernel<NP, RawT, CALIB, MEDIAN>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(APGPU_ELAUNCH, "stack (big): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBigLanes), lds, st, prm, redo, list, ws);
    return check_launch("stack kernel (129..512 frames)");
}

template <typename RawT, bool CALIB>
static int launch_big_np(const StackParams &prm, bool median_only, hipStream_t st, char *describe, int32_t *redo, const int32_t *list, int32_t *ws)
{
    if (prm.N <= 256)
        return median_only ? launch_big_one<256, RawT, CALIB, true>(prm, st, describe, redo, list, ws) : launch_big_one<256, RawT, CALIB, false>(prm, st, describe, redo, list, ws);
    return median_only ? launch_big_one<512, RawT, CALIB, true>(prm, st, describe, redo, list, ws) : launch_big_one<512, RawT, CALIB, false>(prm, st, describe, redo, list, ws);
}

// The exact LDS-resident kernel, on every pixel (redo == nullptr) or on the wavefronts of a redo list.
int launch_big_exact(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe, int32_t *redo,
                     const int32_t *list, int32_t *ws)
{
    if (u16) return calib ? launch_big_np<uint16_t, true>(prm, median_only, st, describe, redo, list, ws) : launch_big_np<uint16_t, false>(prm, median_only, st, describe, redo, list, ws);
    return calib ? launch_big_np<float, true>(prm, median_only, st, describe, redo, list, ws) : launch_big_np<float, false>(prm, median_only, st, describe, redo, list, ws);
}

bool chunks_eligible(const StackParams &prm, bool median_only);                                        // stack_chunks.hip
int launch_chunks(const StackParams &prm, bool u16, bool calib, hipStream_t st, char *describe);
bool rank_chunks_eligible(const StackParams &prm, bool calib, bool median_only);                       // stack_chunks.hip (round 6)
int launch_rank_chunks(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe);

// 129 .. 512 frames: the chunked float32 fast path where it applies (clipped mean; round 6: its median / std planes, the plain
// median - with fused calibration too - and the median / mad_std configuration on raw frames), else the exact kernel.
int launch_big(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe)
{
    if (chunks_eligible(prm, median_only)) return launch_chunks(prm, u16, calib, st, describe);
    if (rank_chunks_eligible(prm, calib, median_only)) return launch_rank_chunks(prm, u16, calib, median_only, st, describe);
    StackParams q = prm;
    q.redo = nullptr;
    return launch_big_exact(q, u16, calib, median_only, st, describe, nullptr, nullptr, nullptr);
}

}  // namespace apgpu_stack
