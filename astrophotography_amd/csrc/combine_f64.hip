// combine_f64.hip - the ccdproc.combine configuration of scripts/ap_combine_darks.py:394-420 on FLOAT64 frames.
//
// ccdproc's Combiner works on a float64 masked cube whatever the files hold; frames that ARE float64 (BITPIX -64: masters
// written by ApMasterCal / ccdproc, float64 calibrated frames) therefore must not pass through the float32 stack kernels.
// This kernel is the float64 statement of oracle/apref.c's apref_combine_ccdproc(), operation for operation, so that the
// result is bit-identical to it (and, through golden group G12, to numpy.ma + astropy):
//   base = median of the finite values, dev = 1.482602218505602 * median(|x - base|)       (np.ma.median, astropy mad_std)
//   keep = finite and not (x - base < -low * dev or x - base > high * dev)                   (strict, one pass; form LEGACY)
//   keep = finite and not (x < base - dev * low or x > base + dev * high), and a column holding a non-finite value is not
//          clipped at all                                             (form ASTROPY: astropy.stats.sigma_clip, ccdproc >= 2.2)
//   mean = (sum of the kept values in FRAME order) / m,  std = sqrt(sum((x - mean)^2) / m)   (float64)
// It is a correctness path, not a fast one (float64 frames are the exception): one pixel per lane, the two sorted columns
// (values, absolute deviations) live in a caller-provided workspace double[2][N][P] (coalesced: slot-major) and are built
// by insertion.
#include "common.h"

namespace {
using namespace apgpu;

__device__ __forceinline__ void insert_sorted(double *col, int64_t P, int n, double v)
{
    int j = n;
    while (j > 0) {
        const double u = col[(int64_t)(j - 1) * P];
        if (!(u > v)) break;
        col[(int64_t)j * P] = u;
        j--;
    }
    col[(int64_t)j * P] = v;
}

__device__ __forceinline__ double median_sorted(const double *col, int64_t P, int n)
{
    return (n & 1) ? col[(int64_t)(n / 2) * P] : (col[(int64_t)(n / 2 - 1) * P] + col[(int64_t)(n / 2) * P]) / 2.0;
}

__global__ __launch_bounds__(256) void combine_ccdproc_f64_kernel(const double *__restrict__ frames, int N, int64_t P, int64_t stride,
                                                                 double low, double high, int form, double *__restrict__ mean_out,
                                                                 int32_t *__restrict__ count_out, double *__restrict__ std_out,
                                                                 double *__restrict__ ws)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double *A = ws + p, *B = ws + (int64_t)N * P + p;
    const double nan = __builtin_nan("");
    int n = 0;
    for (int f = 0; f < N; f++) {
        const double v = frames[(int64_t)f * stride + p];
        if (fabs(v) < __builtin_inf()) {
            insert_sorted(A, P, n, v);
            n++;
        }
    }
    if (n == 0) {
        if (mean_out) mean_out[p] = nan;
        if (count_out) count_out[p] = 0;
        if (std_out) std_out[p] = nan;
        return;
    }
    const double base = median_sorted(A, P, n);
    for (int i = 0; i < n; i++) insert_sorted(B, P, i, fabs(A[(int64_t)i * P] - base));
    const double sd = median_sorted(B, P, n) * 1.482602218505602;
    const bool astropy = form == APGPU_CCDPROC_ASTROPY;
    const double lo = -low * sd, hi = high * sd;                        // LEGACY: thresholds of the differences
    const double lob = base - sd * low, hib = base + sd * high;         // ASTROPY: bounds of the values (sigma_clipping.py:295-296)
    const bool unclipped = astropy && n < N;
    auto rejected = [&](double v) {
        if (astropy) return !unclipped && (v < lob || v > hib);
        const double d = v - base;
        return d < lo || d > hi;
    };
    double sum = 0.0;
    int m = 0;
    for (int f = 0; f < N; f++) {
        const double v = frames[(int64_t)f * stride + p];
        if (!(fabs(v) < __builtin_inf())) continue;
        if (rejected(v)) continue;
        sum += v;
        m++;
    }
    const double mean = m > 0 ? sum / (double)m : nan;
    if (mean_out) mean_out[p] = mean;
    if (count_out) count_out[p] = m;
    if (std_out) {
        double q = 0.0;
        for (int f = 0; f < N; f++) {
            const double v = frames[(int64_t)f * stride + p];
            if (!(fabs(v) < __builtin_inf())) continue;
            if (rejected(v)) continue;
            const double e = v - mean;
            q += e * e;
        }
        std_out[p] = m > 0 ? sqrt(q / (double)m) : nan;
    }
}

}  // namespace

extern "C" size_t apgpu_combine_ccdproc_f64_ws_bytes(int32_t n_frames, int64_t n_pixels)
{
    if (n_frames < 1 || n_pixels < 1) return 0;
    return (size_t)2 * (size_t)n_frames * (size_t)n_pixels * sizeof(double);
}

extern "C" int apgpu_combine_ccdproc_f64(const double *frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, double low,
                                         double high, int32_t form, double *mean, int32_t *count, double *std, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    if (!frames || n_frames < 1 || n_pixels < 1) return fail(APGPU_EINVAL, "combine_ccdproc_f64: bad arguments");
    if (!(low >= 0.0) || !(high >= 0.0)) return fail(APGPU_EINVAL, "combine_ccdproc_f64: thresholds must be >= 0");
    if (form != APGPU_CCDPROC_ASTROPY && form != APGPU_CCDPROC_LEGACY) return fail(APGPU_EINVAL, "combine_ccdproc_f64: bad form %d", form);
    if (!mean && !count && !std) return fail(APGPU_EINVAL, "combine_ccdproc_f64: no output requested");
    if (frame_stride == 0) frame_stride = n_pixels;
    if (frame_stride < n_pixels) return fail(APGPU_EINVAL, "combine_ccdproc_f64: frame_stride < n_pixels");
    if (!workspace || workspace_bytes < apgpu_combine_ccdproc_f64_ws_bytes(n_frames, n_pixels))
        return fail(APGPU_EWORKSPACE, "combine_ccdproc_f64: workspace of %zu bytes needed", apgpu_combine_ccdproc_f64_ws_bytes(n_frames, n_pixels));
    const int64_t grid = (n_pixels + 255) / 256;
    if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "combine_ccdproc_f64: too many pixels");
    hipLaunchKernelGGL(combine_ccdproc_f64_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), frames, (int)n_frames, n_pixels,
                       frame_stride, low, high, (int)form, mean, count, std, static_cast<double *>(workspace));
    return check_launch("combine_ccdproc_f64");
}
