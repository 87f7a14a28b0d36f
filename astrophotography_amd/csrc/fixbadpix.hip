// fixbadpix.hip - A5 ApFixBadPixels.fix_bad_pixels (core/ApFixBadPixels.py:292-445) on gfx950.
//
// The reference walks the bad pixels in a Python loop (np.mgrid + ~0.3 ms per pixel).  Here the image
// is streamed once (out = data), the bad pixels of each tile are compacted into an LDS list, and each
// one gathers its (2*deltapix+1)^2 window from the ORIGINAL image and mask (ApFixBadPixels.py:388-392),
// so a repaired neighbour never feeds another repair, exactly as in the reference.
//   good >= min_valid (4, ApFixBadPixels.py:45,397) -> out = np.median(good)   (float32: even count ->
//   float32(a + b) / 2 through np.mean), else unchanged.  np.median returns NaN if a good value is NaN.
// HBM traffic: 4P + P read, 4P written; the gathers hit L2 (neighbouring rows were just streamed).
#include "common.h"
#include "stack_sort.h"

namespace {
using namespace apgpu;

constexpr int kMaxDelta = 3;
constexpr int kMaxWin = (2 * kMaxDelta + 1) * (2 * kMaxDelta + 1);

// Median of the good neighbours of pixel p, or the pixel's own value if fewer than min_valid exist - the general
// form: any deltapix (the reference has no limit, core/ApFixBadPixels.py:292), float32 or float64 images (a float64
// calibration feeds float64 data, golden G11).  No window array: the two middle order statistics are found by rank
// counting - a good value x is the k-th smallest iff #(values < x) <= k < #(values <= x) - i.e. (window size)^2 reads
// from L2 per bad pixel; bad pixels are few (~0.02 %), and the fast paths below serve float32 with deltapix <= 3.
template <typename T>
__device__ T repair_pixel(const T *__restrict__ data, const uint8_t *__restrict__ mask, int H, int W, int delta,
                          int min_valid, int64_t p, T own, unsigned &nfix)
{
    const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
    const int rmin = max(0, r - delta), rmax = min(H, r + delta + 1);
    const int cmin = max(0, c - delta), cmax = min(W, c + delta + 1);
    int ng = 0;
    bool has_nan = false;
    for (int rr = rmin; rr < rmax; rr++)
        for (int cc = cmin; cc < cmax; cc++) {
            const int64_t q = (int64_t)rr * W + cc;
            if (mask[q] == 0) {
                const T x = data[q];
                has_nan = has_nan || (x != x);
                ng++;
            }
        }
    if (ng < min_valid) return own;
    nfix++;
    if (has_nan) return (T)__builtin_nan("");       // np.median propagates NaN
    const int k1 = (ng - 1) >> 1, k2 = ng >> 1;
    T m1 = (T)0, m2 = (T)0;
    for (int rr = rmin; rr < rmax; rr++)
        for (int cc = cmin; cc < cmax; cc++) {
            const int64_t q = (int64_t)rr * W + cc;
            if (mask[q] != 0) continue;
            const T x = data[q];
            int lt = 0, le = 0;
            for (int r2 = rmin; r2 < rmax; r2++)
                for (int c2 = cmin; c2 < cmax; c2++) {
                    const int64_t q2 = (int64_t)r2 * W + c2;
                    if (mask[q2] != 0) continue;
                    const T y = data[q2];
                    lt += (y < x) ? 1 : 0;
                    le += (y <= x) ? 1 : 0;
                }
            if (lt <= k1 && k1 < le) m1 = x;
            if (lt <= k2 && k2 < le) m2 = x;
        }
    if (ng & 1) return m2;
    if constexpr (sizeof(T) == 8) return (m1 + m2) / 2.0;   // np.mean of the two middle float64 values
    else {
        const float t = m1 + m2;                    // float32: np.mean = float32 sum, then / 2
        return (float)((double)t / 2.0);
    }
}

// Register-resident repair for a compile-time window: the (2D+1)^2 neighbourhood is gathered into a
// power-of-two column (bad / out-of-image slots hold +inf and sort to the top), sorted with the same
// compile-time Batcher network the stack kernels use, and the middle of the ng good values is picked
// with a multiplexer tree.  No scratch memory, no data-dependent loops.
template <int D>
__device__ __forceinline__ float repair_window(const float *__restrict__ data, const uint8_t *__restrict__ mask, int H, int W,
                                               int min_valid, int64_t p, float own, unsigned &nfix)
{
    constexpr int kSide = 2 * D + 1;
    constexpr int kWin = kSide * kSide;
    constexpr int NP = kWin <= 16 ? 16 : (kWin <= 32 ? 32 : 64);
    const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
    const float inf = __builtin_inff();
    float w[NP];
    int ng = 0;
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        if (i < kWin) {
            const int rr = r + i / kSide - D, cc = c + i % kSide - D;
            const bool inside = rr >= 0 && rr < H && cc >= 0 && cc < W;
            const int64_t q = inside ? (int64_t)rr * W + cc : p;
            const bool good = inside && mask[q] == 0;
            const float x = data[q];
            const bool is_nan = x != x;
            has_nan = has_nan || (good && is_nan);
            w[i] = (good && !is_nan) ? x : inf;
            ng += good ? 1 : 0;
        } else {
            w[i] = inf;
        }
    }
    if (ng < min_valid) return own;
    nfix++;
    apgpu_stack::sort_column<NP>(w);
    const float m1 = apgpu_stack::pick_at<NP>(w, (ng - 1) >> 1);
    const float m2 = apgpu_stack::pick_at<NP>(w, ng >> 1);
    const float t = m1 + m2;
    const float med = (ng & 1) ? m2 : (float)((double)t / 2.0);
    return has_nan ? __builtin_nanf("") : med;      // np.median propagates NaN
}

template <typename T, int D>
__device__ __forceinline__ T repair_any(const T *__restrict__ data, const uint8_t *__restrict__ mask, int H, int W,
                                        int delta, int min_valid, int64_t p, T own, unsigned &nfix)
{
    if constexpr (D > 0 && sizeof(T) == 4) return repair_window<D>(data, mask, H, W, min_valid, p, own, nfix);
    else return repair_pixel<T>(data, mask, H, W, delta, min_valid, p, own, nfix);
}

// A block streams tiles of kTile pixels (16-byte loads/stores, 4-byte mask loads) and appends the
// offsets of the tile's bad pixels to an LDS list; after the barrier the list is repaired one bad pixel
// per lane, so the gathers run on densely packed lanes instead of diverging inside the streaming loop.
// Repairs read the ORIGINAL image and overwrite the streamed copy (same block, ordered by the barrier).
constexpr int kPxPerLane = 16;
constexpr int kTile = 256 * kPxPerLane;

// VEC: float32 image with 16-byte aligned data/out and a 4-byte aligned mask (float4 / uchar4 accesses); otherwise
// (float64 images, or a frame cut out of a slab whose pixel count is not a multiple of 4) the same tile is streamed with
// coalesced scalar accesses.
template <typename T, int D, bool VEC>
__global__ __launch_bounds__(256) void fix_badpix_kernel(const T *__restrict__ data, const uint8_t *__restrict__ mask,
                                                        int H, int W, int delta, int min_valid, T *__restrict__ out,
                                                        unsigned long long *__restrict__ stats)
{
    __shared__ int s_n;
    __shared__ int s_list[kTile];
    const int64_t P = (int64_t)H * W;
    const int64_t groups = P / 4;
    const int64_t ntiles = (P + kTile - 1) / kTile;
    unsigned nbad = 0, nfix = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int64_t base = tile * kTile;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < kPxPerLane / 4; q++) {
                const int64_t g = base / 4 + q * 256 + threadIdx.x;
                if (g < groups) {
                    const float4 v = reinterpret_cast<const float4 *>(data)[g];
                    const uchar4 m = reinterpret_cast<const uchar4 *>(mask)[g];
                    reinterpret_cast<float4 *>(out)[g] = v;
                    if (m.x | m.y | m.z | m.w) {
                        const int off = (int)(g * 4 - base);
                        if (m.x) s_list[atomicAdd(&s_n, 1)] = off;
                        if (m.y) s_list[atomicAdd(&s_n, 1)] = off + 1;
                        if (m.z) s_list[atomicAdd(&s_n, 1)] = off + 2;
                        if (m.w) s_list[atomicAdd(&s_n, 1)] = off + 3;
                    }
                }
            }
            // the up-to-3 pixels past the last whole group belong to the last tile
            if (tile == ntiles - 1 && threadIdx.x < (int)(P - groups * 4)) {
                const int64_t pp = groups * 4 + threadIdx.x;
                out[pp] = data[pp];
                if (mask[pp] != 0) s_list[atomicAdd(&s_n, 1)] = (int)(pp - base);
            }
        } else {
#pragma unroll 4
            for (int q = 0; q < kPxPerLane; q++) {
                const int64_t pp = base + q * 256 + threadIdx.x;
                if (pp < P) {
                    out[pp] = data[pp];
                    if (mask[pp] != 0) s_list[atomicAdd(&s_n, 1)] = (int)(pp - base);
                }
            }
        }
        __syncthreads();
        const int n = s_n;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int64_t pp = base + s_list[i];
            nbad++;
            const T own = data[pp];
            const T val = repair_any<T, D>(data, mask, H, W, delta, min_valid, pp, own, nfix);
            out[pp] = val;
        }
        __syncthreads();
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        nbad += __shfl_down(nbad, d);
        nfix += __shfl_down(nfix, d);
    }
    if ((threadIdx.x % kWave) == 0 && nbad) {
        atomicAdd(&stats[0], (unsigned long long)nbad);
        atomicAdd(&stats[1], (unsigned long long)nfix);
    }
}

__global__ void fix_badpix_finish_kernel(unsigned long long *stats)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) stats[2] = stats[0] - stats[1];
}

}  // namespace

template <typename T>
static int fix_badpix_impl(const T *data, const uint8_t *mask, int64_t height, int64_t width, int32_t deltapix,
                           int32_t min_valid, T *out, int64_t *stats_out, void *stream)
{
    if (!data || !mask || !out || !stats_out) return fail(APGPU_EINVAL, "fix_badpix: NULL pointer argument");
    if (out == data) return fail(APGPU_EINVAL, "fix_badpix: out may not alias data");
    if (height <= 0 || width <= 0 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "fix_badpix: bad shape");
    if (deltapix < 0) return fail(APGPU_EINVAL, "fix_badpix: deltapix %d < 0", deltapix);
    if (min_valid < 1) return fail(APGPU_EINVAL, "fix_badpix: min_valid must be >= 1");
    if ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & (sizeof(T) - 1))
        return fail(APGPU_EINVAL, "fix_badpix: data/out must be aligned to their element size");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(stats_out, 0, 3 * sizeof(int64_t), st) != hipSuccess) return fail(APGPU_ELAUNCH, "fix_badpix: memset failed");
    const int64_t P = height * width;
    int64_t grid = (P + kTile - 1) / kTile;
    if (grid > kNumCU * 8) grid = kNumCU * 8;
    unsigned long long *st_dev = reinterpret_cast<unsigned long long *>(stats_out);
#define APGPU_FIX_LAUNCH(TT, D, V)                                                                                         \
    hipLaunchKernelGGL((fix_badpix_kernel<TT, D, V>), dim3((unsigned)grid), dim3(256), 0, st, data, mask, (int)height, (int)width, \
                       deltapix, min_valid, out, st_dev)
    if constexpr (sizeof(T) == 4) {
        const bool vec = !(((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & 15) || (reinterpret_cast<uintptr_t>(mask) & 3));
        if (vec) {
            switch (deltapix) {
            case 1: APGPU_FIX_LAUNCH(float, 1, true); break;
            case 2: APGPU_FIX_LAUNCH(float, 2, true); break;
            case 3: APGPU_FIX_LAUNCH(float, 3, true); break;
            default: APGPU_FIX_LAUNCH(float, 0, true); break;
            }
        } else {
            switch (deltapix) {
            case 1: APGPU_FIX_LAUNCH(float, 1, false); break;
            case 2: APGPU_FIX_LAUNCH(float, 2, false); break;
            case 3: APGPU_FIX_LAUNCH(float, 3, false); break;
            default: APGPU_FIX_LAUNCH(float, 0, false); break;
            }
        }
    } else {
        APGPU_FIX_LAUNCH(double, 0, false);
    }
#undef APGPU_FIX_LAUNCH
    if (int rc = check_launch("fix_badpix")) return rc;
    hipLaunchKernelGGL(fix_badpix_finish_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long *>(stats_out));
    return check_launch("fix_badpix_finish");
}

extern "C" int apgpu_fix_badpix_f32(const float *data, const uint8_t *mask, int64_t height, int64_t width, int32_t deltapix,
                                    int32_t min_valid, float *out, int64_t *stats_out, void *stream)
{
    return fix_badpix_impl<float>(data, mask, height, width, deltapix, min_valid, out, stats_out, stream);
}

extern "C" int apgpu_fix_badpix_f64(const double *data, const uint8_t *mask, int64_t height, int64_t width, int32_t deltapix,
                                    int32_t min_valid, double *out, int64_t *stats_out, void *stream)
{
    return fix_badpix_impl<double>(data, mask, height, width, deltapix, min_valid, out, stats_out, stream);
}
