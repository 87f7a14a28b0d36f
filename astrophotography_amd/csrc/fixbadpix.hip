// fixbadpix.hip - A5 ApFixBadPixels.fix_bad_pixels (core/ApFixBadPixels.py:292-445) on gfx950.
//
// The reference walks the bad pixels in a Python loop (np.mgrid + ~0.3 ms per pixel).  Here the image
// is streamed once (out = data) and the rare lanes that sit on a bad pixel gather their
// (2*deltapix+1)^2 window from the ORIGINAL image and mask (ApFixBadPixels.py:388-392), so a repaired
// neighbour never feeds another repair, exactly as in the reference.
//   good >= min_valid (4, ApFixBadPixels.py:45,397) -> out = np.median(good)   (float32: even count ->
//   float32(a + b) / 2 through np.mean), else unchanged.  np.median returns NaN if a good value is NaN.
// HBM traffic: 4P + P read, 4P written; the gathers hit L2 (neighbouring rows were just streamed).
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kMaxDelta = 3;
constexpr int kMaxWin = (2 * kMaxDelta + 1) * (2 * kMaxDelta + 1);

__global__ __launch_bounds__(256) void fix_badpix_kernel(const float *__restrict__ data, const uint8_t *__restrict__ mask,
                                                        int H, int W, int delta, int min_valid, float *__restrict__ out,
                                                        unsigned long long *__restrict__ stats)
{
    const int64_t P = (int64_t)H * W;
    unsigned nbad = 0, nfix = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        float val = data[p];
        if (mask[p] != 0) {
            nbad++;
            const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
            const int rmin = max(0, r - delta), rmax = min(H, r + delta + 1);
            const int cmin = max(0, c - delta), cmax = min(W, c + delta + 1);
            float good[kMaxWin];
            int ng = 0;
            bool has_nan = false;
            for (int rr = rmin; rr < rmax; rr++)
                for (int cc = cmin; cc < cmax; cc++) {
                    const int64_t q = (int64_t)rr * W + cc;
                    if (mask[q] == 0) {
                        const float x = data[q];
                        has_nan = has_nan || (x != x);
                        // insertion into the sorted prefix
                        int k = ng++;
                        while (k > 0 && good[k - 1] > x) { good[k] = good[k - 1]; k--; }
                        good[k] = x;
                    }
                }
            if (ng >= min_valid) {
                nfix++;
                if (has_nan) val = __builtin_nanf("");
                else if (ng & 1) val = good[ng / 2];
                else {
                    const float t = good[ng / 2 - 1] + good[ng / 2];
                    val = (float)((double)t / 2.0);
                }
            }
        }
        out[p] = val;
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

extern "C" int apgpu_fix_badpix_f32(const float *data, const uint8_t *mask, int64_t height, int64_t width, int32_t deltapix,
                                    int32_t min_valid, float *out, int64_t *stats_out, void *stream)
{
    if (!data || !mask || !out || !stats_out) return fail(APGPU_EINVAL, "fix_badpix: NULL pointer argument");
    if (out == data) return fail(APGPU_EINVAL, "fix_badpix: out may not alias data");
    if (height <= 0 || width <= 0 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "fix_badpix: bad shape");
    if (deltapix < 0 || deltapix > kMaxDelta) return fail(APGPU_EUNSUPPORTED, "fix_badpix: deltapix %d outside 0..%d", deltapix, kMaxDelta);
    if (min_valid < 1) return fail(APGPU_EINVAL, "fix_badpix: min_valid must be >= 1");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(stats_out, 0, 3 * sizeof(int64_t), st) != hipSuccess) return fail(APGPU_ELAUNCH, "fix_badpix: memset failed");
    const int64_t P = height * width;
    int64_t grid = (P + 255) / 256;
    if (grid > kNumCU * 8) grid = kNumCU * 8;
    hipLaunchKernelGGL(fix_badpix_kernel, dim3((unsigned)grid), dim3(256), 0, st, data, mask, (int)height, (int)width, deltapix,
                       min_valid, out, reinterpret_cast<unsigned long long *>(stats_out));
    if (int rc = check_launch("fix_badpix")) return rc;
    hipLaunchKernelGGL(fix_badpix_finish_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long *>(stats_out));
    return check_launch("fix_badpix_finish");
}
