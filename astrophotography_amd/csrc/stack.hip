// stack.hip - argument checking and dispatch for the stack reductions (kernels: stack_kernels.h,
// instantiated per raw dtype / fused-calibration flag / slot-count group in stack_inst_*.hip so that they
// build in parallel).
#include "stack_kernels.h"

#include <cstdlib>

namespace apgpu_stack {
int launch_big(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe);   // stack_big.hip
bool chunks_eligible(const StackParams &prm, bool median_only);                                                  // stack_chunks.hip
}

namespace {
using namespace apgpu;
using namespace apgpu_stack;

__global__ __launch_bounds__(256) void moments_finalize_kernel(const float *__restrict__ mom, float *mean,
                                                              float *std, int64_t P)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float s = mom[p], n = mom[P + p];                   // planes: sum, count (, sum of squares: not used here)
    const float m = s / n;                                    // n == 0 -> NaN
    if (mean) mean[p] = n > 0.f ? m : __builtin_nanf("");
}

__global__ __launch_bounds__(256) void moments_finalize_f64_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                                                                  const int32_t *__restrict__ count, float *mean, float *std,
                                                                  double *mean64, double *std64, int64_t P)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int cnt = count[p];
    const double n = (double)cnt;
    const double nan = __builtin_nan("");
    const double m = cnt > 0 ? sum[p] / n : nan;
    if (mean) mean[p] = (float)m;
    if (mean64) mean64[p] = m;
    if (sumsq && (std || std64)) {
        double var = sumsq[p] / n - m * m;
        var = var > 0.0 ? var : 0.0;
        const double sd = cnt > 0 ? sqrt(var) : nan;
        if (std) std[p] = (float)sd;
        if (std64) std64[p] = sd;
    }
}

// the packed float64 layout (3): planes sum, count, sumsq as three pointers (the caller passes buffer, buffer + P, buffer + 2 P)
__global__ __launch_bounds__(256) void moments_finalize_f64p_kernel(const double *__restrict__ sum, const double *__restrict__ count,
                                                                   const double *__restrict__ sumsq, float *mean, float *std,
                                                                   double *mean64, double *std64, int64_t P)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double n = count[p];
    const double nan = __builtin_nan("");
    const double m = n > 0.0 ? sum[p] / n : nan;
    if (mean) mean[p] = (float)m;
    if (mean64) mean64[p] = m;
    if (sumsq && (std || std64)) {
        double var = sumsq[p] / n - m * m;
        var = var > 0.0 ? var : 0.0;
        const double sd = n > 0.0 ? sqrt(var) : nan;
        if (std) std[p] = (float)sd;
        if (std64) std64[p] = sd;
    }
}

int stack_dispatch(const apgpu_stack_args *args, bool median_only, void *stream, char *describe = nullptr)
{
    if (!args) return fail(APGPU_EINVAL, "stack: args is NULL");
    if (!args->frames) return fail(APGPU_EINVAL, "stack: frames is NULL");
    if (args->n_pixels <= 0) return fail(APGPU_EINVAL, "stack: n_pixels = %lld", (long long)args->n_pixels);
    if (args->n_frames < 1) return fail(APGPU_EINVAL, "stack: n_frames = %d", args->n_frames);
    if (args->n_frames > APGPU_MAX_STACK)
        return fail(APGPU_EUNSUPPORTED, "stack: n_frames = %d exceeds APGPU_MAX_STACK = %d (shard the frames over GPUs or "
                    "combine partial moments)", args->n_frames, APGPU_MAX_STACK);
    if (args->dtype != APGPU_F32 && args->dtype != APGPU_U16) return fail(APGPU_EINVAL, "stack: bad dtype %d", args->dtype);
    if (args->frame_stride != 0 && args->frame_stride < args->n_pixels)
        return fail(APGPU_EINVAL, "stack: frame_stride %lld < n_pixels %lld", (long long)args->frame_stride, (long long)args->n_pixels);
    const bool calib = args->bias != nullptr;
    if (calib && (!args->dark || !args->exp_ratio))
        return fail(APGPU_EINVAL, "stack: fused calibration needs bias, dark and exp_ratio");
    if (!median_only) {
        if (args->center != APGPU_CENTER_MEDIAN && args->center != APGPU_CENTER_MEAN)
            return fail(APGPU_EINVAL, "stack: bad center %d", args->center);
        if (args->dev != APGPU_DEV_STD && args->dev != APGPU_DEV_MAD_STD)
            return fail(APGPU_EINVAL, "stack: bad dev %d", args->dev);
        if (args->maxiters == 0) return fail(APGPU_EINVAL, "stack: maxiters must be >= 1 or < 0");
        if (!(args->sigma_lower >= 0.0) || !(args->sigma_upper >= 0.0))
            return fail(APGPU_EINVAL, "stack: sigma must be >= 0");
        if (!args->mean && !args->median && !args->std && !args->count && !args->moments && !args->mean_f64 && !args->std_f64)
            return fail(APGPU_EINVAL, "stack: no output requested");
        if (args->moments_f64 < 0 || args->moments_f64 > 4) return fail(APGPU_EINVAL, "stack: bad moments_f64 %d", args->moments_f64);
        if (args->moments && args->moments_f64 && (reinterpret_cast<uintptr_t>(args->moments) & 7))
            return fail(APGPU_EINVAL, "stack: float64 moments must be 8-byte aligned");
        if (args->flags & ~(APGPU_STACK_EXACT_MOMENTS | APGPU_STACK_MOMENTS_MEAN | APGPU_STACK_SINGLE_KERNEL | APGPU_STACK_NONFINITE_UNCLIPPED))
            return fail(APGPU_EINVAL, "stack: unknown flags 0x%x", args->flags);
        if ((args->flags & APGPU_STACK_NONFINITE_UNCLIPPED) && args->dev != APGPU_DEV_MAD_STD)
            return fail(APGPU_EINVAL, "stack: APGPU_STACK_NONFINITE_UNCLIPPED belongs to the median / mad_std configuration (dev = APGPU_DEV_MAD_STD)");
        if (args->workspace) {
            if (reinterpret_cast<uintptr_t>(args->workspace) & 15) return fail(APGPU_EINVAL, "stack: the workspace must be 16-byte aligned");
            if (args->workspace_bytes < apgpu_stack_ws_bytes(args->n_pixels, nullptr))
                return fail(APGPU_EWORKSPACE, "stack: workspace of %zu bytes, %lld pixels need %zu (apgpu_stack_ws_bytes)",
                            args->workspace_bytes, (long long)args->n_pixels, apgpu_stack_ws_bytes(args->n_pixels, nullptr));
        }
    } else if (!args->median) {
        return fail(APGPU_EINVAL, "stack_median: median output is NULL");
    }
    StackParams prm{};
    prm.frames = args->frames;
    prm.bias = args->bias;
    prm.dark = args->dark;
    prm.nflat = args->nflat;
    prm.exp_ratio = args->exp_ratio;
    prm.pedestal = args->pedestal;
    prm.pixmask = args->pixmask;
    prm.mean = args->mean;
    prm.median = args->median;
    prm.std = args->std;
    prm.moments = args->moments;
    prm.count = args->count;
    prm.P = args->n_pixels;
    prm.stride = args->frame_stride > 0 ? args->frame_stride : args->n_pixels;
    prm.sl2 = args->sigma_lower * args->sigma_lower;
    prm.su2 = args->sigma_upper * args->sigma_upper;
    prm.N = args->n_frames;
    prm.still_biased = args->dark_still_biased;
    prm.center = args->center;
    prm.dev = median_only ? 0 : args->dev;
    prm.maxiters = args->maxiters;
    prm.moments64 = median_only ? 0 : args->moments_f64;
    prm.mean64 = median_only ? nullptr : args->mean_f64;
    prm.std64 = median_only ? nullptr : args->std_f64;
    prm.redo = median_only ? nullptr : static_cast<int32_t *>(args->workspace);
    prm.single_kernel = (args->flags & APGPU_STACK_SINGLE_KERNEL) ? 1 : 0;
    prm.unclipped_nonfinite = (!median_only && (args->flags & APGPU_STACK_NONFINITE_UNCLIPPED)) ? 1 : 0;
    prm.fast32 = (median_only || (args->flags & APGPU_STACK_EXACT_MOMENTS)) ? 0 : ((args->flags & APGPU_STACK_MOMENTS_MEAN) ? 2 : 1);
#ifdef APGPU_DEVELOPMENT                                     // measurement knobs, never in a release build
    if (getenv("APGPU_DEBUG_STRIDE0")) prm.stride = 0;      // all frames alias frame 0: compute-only timing
    if (const char *e = getenv("APGPU_DEBUG_MAXITERS")) prm.maxiters = atoi(e);
    if (getenv("APGPU_DEBUG_EXACT")) prm.fast32 = 0;
#endif
    hipStream_t st = as_stream(stream);
    // beyond 128 frames the column does not fit the registers: chunked fast path / LDS-resident exact kernel (stack_big.hip)
    if (prm.N > 128)
        return launch_big(prm, args->dtype == APGPU_U16, calib, median_only, st, describe);
    if (args->dtype == APGPU_F32)
        return calib ? launch_np<float, true>(prm, median_only, st, describe) : launch_np<float, false>(prm, median_only, st, describe);
    return calib ? launch_np<uint16_t, true>(prm, median_only, st, describe) : launch_np<uint16_t, false>(prm, median_only, st, describe);
}

}  // namespace

static_assert(APGPU_STACK_WS_STATS_OFFSET == kWsStats * sizeof(int32_t), "include/apgpu.h names the statistics' place in the workspace");

extern "C" size_t apgpu_stack_ws_bytes(int64_t n_pixels, size_t *zero_bytes)
{
    if (n_pixels <= 0) {
        if (zero_bytes) *zero_bytes = 0;
        return 0;
    }
    if (zero_bytes) *zero_bytes = (size_t)ws_list_off(n_pixels) * sizeof(int32_t);
    return (size_t)ws_total_words(n_pixels) * sizeof(int32_t);
}

extern "C" int apgpu_stack_sigclip(const apgpu_stack_args *args, void *stream)
{
    return stack_dispatch(args, false, stream);
}

extern "C" int apgpu_stack_median(const apgpu_stack_args *args, void *stream)
{
    return stack_dispatch(args, true, stream);
}

extern "C" int apgpu_stack_kernel_name(const apgpu_stack_args *args, int median_only, char *name_host, size_t name_bytes)
{
    if (!name_host || name_bytes == 0) return fail(APGPU_EINVAL, "stack_kernel_name: no buffer");
    char buf[256] = {0};
    const int rc = stack_dispatch(args, median_only != 0, nullptr, buf);
    if (rc != APGPU_OK) return rc;
    snprintf(name_host, name_bytes, "%s", buf);
    return APGPU_OK;
}

extern "C" int apgpu_moments_finalize(const float *moments, float *mean, float *std, int64_t n_pixels, void *stream)
{
    if (!moments || n_pixels <= 0) return fail(APGPU_EINVAL, "moments_finalize: bad arguments");
    if (std) return fail(APGPU_EUNSUPPORTED, "moments_finalize: no standard deviation from float32 moments (sumsq/n - mean^2 "
                         "cancels); use the float64 moment layout and apgpu_moments_finalize_f64");
    const int block = 256;
    const int64_t grid = (n_pixels + block - 1) / block;
    hipLaunchKernelGGL(moments_finalize_kernel, dim3((unsigned)grid), dim3(block), 0, as_stream(stream), moments, mean,
                       std, n_pixels);
    return check_launch("moments_finalize");
}

extern "C" int apgpu_moments_finalize_f64p(const double *sum, const double *count, const double *sumsq, float *mean, float *std,
                                           double *mean_f64, double *std_f64, int64_t n_pixels, void *stream)
{
    if (!sum || !count || n_pixels <= 0) return fail(APGPU_EINVAL, "moments_finalize_f64p: bad arguments");
    if ((std || std_f64) && !sumsq) return fail(APGPU_EINVAL, "moments_finalize_f64p: std wanted but sumsq is NULL");
    const int block = 256;
    const int64_t grid = (n_pixels + block - 1) / block;
    hipLaunchKernelGGL(moments_finalize_f64p_kernel, dim3((unsigned)grid), dim3(block), 0, as_stream(stream), sum, count, sumsq,
                       mean, std, mean_f64, std_f64, n_pixels);
    return check_launch("moments_finalize_f64p");
}

extern "C" int apgpu_moments_finalize_f64(const double *sum, const double *sumsq, const int32_t *count, float *mean, float *std,
                                          double *mean_f64, double *std_f64, int64_t n_pixels, void *stream)
{
    if (!sum || !count || n_pixels <= 0) return fail(APGPU_EINVAL, "moments_finalize_f64: bad arguments");
    if ((std || std_f64) && !sumsq) return fail(APGPU_EINVAL, "moments_finalize_f64: std wanted but sumsq is NULL");
    const int block = 256;
    const int64_t grid = (n_pixels + block - 1) / block;
    hipLaunchKernelGGL(moments_finalize_f64_kernel, dim3((unsigned)grid), dim3(block), 0, as_stream(stream), sum, sumsq, count,
                       mean, std, mean_f64, std_f64, n_pixels);
    return check_launch("moments_finalize_f64");
}
