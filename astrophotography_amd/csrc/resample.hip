// resample.hip - F3: affine Lanczos-3 resample of registered frames on gfx950 (the step the reference
// delegates to SWarp: scripts/resample_all.sh:123-131 RESAMPLING_TYPE LANCZOS3, :298 FSCALE = 1/EXPTIME,
// :330-342 the swarp call).  There is no in-tree arithmetic to match; the definition is this build's own and
// is restated on the CPU in oracle/apref.c (apref_resample_affine_f32) - the two agree bit for bit.
//
//   F[k] = llrint(A[k] * 2^32); xin = F0 x + F1 y + F2, yin = F3 x + F4 y + F5   (32.32 fixed point, output -> input: exactly
//   reproducible, and a 64-bit add per row where the float64 evaluation of round 2 cost ~25 four-cycle instructions per pixel)
//   ix = xin >> 32, phase px = top log2(n_phases) bits of the fraction, rounded; taps ix-2 .. ix+3 weighted by lut[px][0..5]
//   (a host-built table of normalised Lanczos-3 weights), rows combined by lut[py]; even / odd fmaf chains in a
//   fixed order (= packed float32 arithmetic); result * fscale[f].  Any tap outside the frame, masked or non-finite -> NaN, weight 0.
//
// (Round 3 also built a variant in which a lane produces four CONSECUTIVE rows and keeps its 6 x 6 window in registers between
// them - one new window row per pixel, 10 instead of 18 LDS reads - behind a wave vote on "same columns, next row": the
// register shuffling around the vote cost more VALU work (151 against 133 instructions per pixel) than the LDS reads it
// saved: 6.2 ms against 5.2 ms per 64 x 4096^2, not kept.  The kernel is VALU-bound: 133 instructions per pixel.)
//
// A workgroup produces a 64 x 16 output tile.  For registration-sized transforms (small rotation / shift /
// scale near 1) the tile's input footprint is ~70 x 22 pixels: it is staged in LDS once (coalesced runs,
// invalid pixels stored as NaN) and the 36 taps of every output pixel are LDS reads, so HBM traffic is one
// read of the input (+ ~40 % halo, mostly L2 hits) and one write of the output: 8 B per pixel.  A footprint
// that does not fit (strong shear / large scale) takes the direct-gather path, same arithmetic.
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kTileW = APGPU_RESAMPLE_TILE_W, kTileH = APGPU_RESAMPLE_TILE_H;
constexpr int kLdsFloats = 4096;               // 16 KB footprint buffer: eight workgroups (all 32 wave slots) per CU
typedef float v2f __attribute__((ext_vector_type(2)));

struct FrameView {
    const float *src;
    const uint8_t *mask;
    int h_in, w_in;
};

// src[row][col], or NaN if the pixel is outside the frame, masked or not finite
__device__ __forceinline__ float fetch_global(const FrameView &fv, int row, int col)
{
    if (row < 0 || row >= fv.h_in || col < 0 || col >= fv.w_in) return __builtin_nanf("");
    const int64_t q = (int64_t)row * fv.w_in + col;
    const float v = fv.src[q];
    const bool bad = !(fabsf(v) < __builtin_inff()) || (fv.mask && fv.mask[q] != 0);
    return bad ? __builtin_nanf("") : v;
}

// One output pixel from its 6 x 6 window; `row(j)` returns the 6 samples of window row j as three pairs.
// Evaluation order (restated in the oracle): per row the even and the odd taps are two fmaf chains, the rows
// are combined by two fmaf chains over j, and the two halves are added last - which is exactly a sequence of
// packed float32 operations on (even, odd) pairs: 4 instructions per row for 6 taps.
struct Weights {
    v2f wx01, wx23, wx45;       // x taps as (even, odd) pairs
    v2f wy01, wy23, wy45;
};

__device__ __forceinline__ Weights load_weights(const float *__restrict__ lut, int px, int py)
{
    const v2f *wxp = reinterpret_cast<const v2f *>(lut + 6 * px);
    const v2f *wyp = reinterpret_cast<const v2f *>(lut + 6 * py);
    Weights w;
    w.wx01 = wxp[0]; w.wx23 = wxp[1]; w.wx45 = wxp[2];
    w.wy01 = wyp[0]; w.wy23 = wyp[1]; w.wy45 = wyp[2];
    return w;
}

template <typename RowFn>
__device__ __forceinline__ float window_sum(const Weights &w, RowFn row)
{
    const float wy[6] = {w.wy01.x, w.wy01.y, w.wy23.x, w.wy23.y, w.wy45.x, w.wy45.y};
    v2f V = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; j++) {
        v2f s01, s23, s45;
        row(j, s01, s23, s45);
        v2f acc = w.wx01 * s01;
        acc = __builtin_elementwise_fma(w.wx23, s23, acc);
        acc = __builtin_elementwise_fma(w.wx45, s45, acc);
        const v2f wyj = {wy[j], wy[j]};
        V = (j == 0) ? wyj * acc : __builtin_elementwise_fma(wyj, acc, V);
    }
    return V.x + V.y;
}

// What thread 0 works out once per workgroup (everything here is tile-uniform): round 2 had every lane redo the float64
// corner arithmetic - ~60 four-cycle instructions per lane for four output pixels each.
struct TileInfo {
    long long F[6];             // the transform in 32.32 fixed point
    int bx0, by0, fw, fh;       // input footprint staged in LDS
    int staged, sane, interior; // interior: the footprint lies inside the frame and the tile inside the output rows
    float fs;                   // flux scale
};

// The pixels of one lane: column x, rows yb0, yb0 + 4, ...  INTERIOR: every window of the tile lies inside the frame (decided
// once per workgroup), so the per-pixel frame tests and the selects they feed disappear.
template <bool INTERIOR>
__device__ __forceinline__ void resample_pixels(const TileInfo &ti, const FrameView &fv, const float *tile, const float *__restrict__ lut,
                                                int sh, int x, int yb0, int h_out, int64_t row_stride, float *op, uint8_t *wp)
{
    const long long F0 = ti.F[0], F1 = ti.F[1], F2 = ti.F[2], F3 = ti.F[3], F4 = ti.F[4], F5 = ti.F[5];
    const int bx0 = ti.bx0, by0 = ti.by0, fw = ti.fw;
    const bool staged = ti.staged != 0, sane = ti.sane != 0;
    const float fs = ti.fs;
    // 64-bit two's-complement sums: exact, because the true coordinates fit (sane), whatever the partial products do
    unsigned long long X = (unsigned long long)F0 * (unsigned long long)(long long)x + (unsigned long long)F1 * (unsigned long long)(long long)yb0 + (unsigned long long)F2;
    unsigned long long Y = (unsigned long long)F3 * (unsigned long long)(long long)x + (unsigned long long)F4 * (unsigned long long)(long long)yb0 + (unsigned long long)F5;
    const unsigned long long dX = (unsigned long long)F1 * 4ull, dY = (unsigned long long)F4 * 4ull;
    // One pixel per trip (not unrolled): residency hides latency better than batching (round 1 measurement).
#pragma unroll 1
    for (int k = 0; k < kTileH / 4; k++) {
        const int y = yb0 + 4 * k;
        const long long xin = (long long)X, yin = (long long)Y;
        X += dX;
        Y += dY;
        // a sane tile keeps the coordinates within +-1e9: the integer part IS the high dword (no 64-bit compares or selects)
        const int jx = (int)(xin >> 32), jy = (int)(yin >> 32);
        const unsigned frx = (unsigned)(unsigned long long)xin, fry = (unsigned)(unsigned long long)yin;
        const int px = (int)((frx >> sh) + ((frx >> (sh - 1)) & 1u));
        const int py = (int)((fry >> sh) + ((fry >> (sh - 1)) & 1u));
        // 2 <= ix <= w_in - 4 (the 6 x 6 window inside the frame)
        const bool inside = INTERIOR || (sane && (y < h_out) && (unsigned)(jx - 2) < (unsigned)(fv.w_in - 5) && (unsigned)(jy - 2) < (unsigned)(fv.h_in - 5));
        const int ix = (INTERIOR || inside) ? jx : 0, iy = (INTERIOR || inside) ? jy : 0;
        const Weights wts = load_weights(lut, (INTERIOR || inside) ? px : 0, (INTERIOR || inside) ? py : 0);
        if (!INTERIOR && y >= h_out) break;
        float v;
        if (INTERIOR || staged) {
            // pixels outside the frame read (and discard) the tile origin
            const int off = (INTERIOR || inside) ? (iy - 2 - by0) * fw + (ix - 2 - bx0) : 0;
            const int stride = (INTERIOR || inside) ? fw : 0;
            const float *t = tile + off;
            // all 18 ds_read2_b32 of the window are issued before the first product
            v2f smp[6][3];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const float *r = t + j * stride;
                smp[j][0] = v2f{r[0], r[1]};
                smp[j][1] = v2f{r[2], r[3]};
                smp[j][2] = v2f{r[4], r[5]};
            }
            v = window_sum(wts, [&](int j, v2f &s01, v2f &s23, v2f &s45) {
                s01 = smp[j][0];
                s23 = smp[j][1];
                s45 = smp[j][2];
            });
        } else {
            v = window_sum(wts, [&](int j, v2f &s01, v2f &s23, v2f &s45) {
                s01 = v2f{fetch_global(fv, iy - 2 + j, ix - 2), fetch_global(fv, iy - 2 + j, ix - 1)};
                s23 = v2f{fetch_global(fv, iy - 2 + j, ix), fetch_global(fv, iy - 2 + j, ix + 1)};
                s45 = v2f{fetch_global(fv, iy - 2 + j, ix + 2), fetch_global(fv, iy - 2 + j, ix + 3)};
            });
        }
        // invalid taps arrive as NaN and poison v; the weight plane is "out is not NaN" in the oracle too
        const float res = ((INTERIOR || inside) && v == v) ? v * fs : __builtin_nanf("");
        *op = res;
        op += row_stride;
        if (wp) {
            *wp = (res == res) ? 1 : 0;
            wp += row_stride;
        }
    }
}

template <bool HAS_MASK>
__global__ __launch_bounds__(256) void resample_affine_kernel(const float *__restrict__ frames, const uint8_t *__restrict__ mask,
                                                             const double *__restrict__ affines, int per_tile, int conserve_flux,
                                                             const float *__restrict__ fscale, const float *__restrict__ lut,
                                                             int log2_phases, float *__restrict__ out, uint8_t *__restrict__ wout,
                                                             int h_in, int w_in, int h_out, int w_out)
{
    __shared__ float tile[kLdsFloats];
    __shared__ TileInfo ti;
    const int64_t f = blockIdx.z;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    FrameView fv;
    fv.src = frames + f * (int64_t)h_in * w_in;
    fv.mask = HAS_MASK ? mask : nullptr;
    fv.h_in = h_in;
    fv.w_in = w_in;
    if (threadIdx.x == 0) {
        // one transform per frame, or one per output tile (= per workgroup)
        const double *A = affines + 6 * (per_tile ? (f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x : f);
        const double a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5];
        float fs = fscale ? fscale[f] : 1.0f;
        if (conserve_flux) fs = (float)((double)fs * fabs(fma(a0, a4, -(a1 * a3))));   // output pixel area in input pixels
        ti.fs = fs;
        // input footprint of the tile: an affine map takes its extremes at the tile corners
        const double xa = (double)x0, xb = (double)(x0 + kTileW - 1 < w_out - 1 ? x0 + kTileW - 1 : w_out - 1);
        const double ya = (double)y0, yb = (double)(y0 + kTileH - 1 < h_out - 1 ? y0 + kTileH - 1 : h_out - 1);
        double mnx = __builtin_inf(), mxx = -__builtin_inf(), mny = __builtin_inf(), mxy = -__builtin_inf();
        const double cx[4] = {xa, xb, xa, xb}, cy[4] = {ya, ya, yb, yb};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double xi = fma(a0, cx[k], fma(a1, cy[k], a2));
            const double yi = fma(a3, cx[k], fma(a4, cy[k], a5));
            mnx = fmin(mnx, xi); mxx = fmax(mxx, xi);
            mny = fmin(mny, yi); mxy = fmax(mxy, yi);
        }
        // the tile is defined if its corner coordinates stay within +-1e9 pixels and the coefficients below 2^30 (the
        // fixed-point evaluation is then exact: the true sums fit 64 bits); false for NaN coefficients
        const double amax = fmax(fmax(fmax(fabs(a0), fabs(a1)), fmax(fabs(a2), fabs(a3))), fmax(fabs(a4), fabs(a5)));
        const bool sane = (mnx > -1e9) && (mxx < 1e9) && (mny > -1e9) && (mxy < 1e9) && (amax < 1073741824.0) &&
                          (a0 == a0) && (a1 == a1) && (a2 == a2) && (a3 == a3) && (a4 == a4) && (a5 == a5);
#pragma unroll
        for (int k = 0; k < 6; k++) ti.F[k] = sane ? __double2ll_rn(A[k] * 4294967296.0) : 0;
        int bx0 = 0, by0 = 0, w = 0, h = 0;
        bool staged = false;
        if (sane) {
            // (one pixel of slack on every side: the corners are evaluated in float64, the pixels in fixed point, and the two
            // can fall on different sides of an integer)
            bx0 = (int)floor(mnx) - 3;
            by0 = (int)floor(mny) - 3;
            w = (int)floor(mxx) + 4 - bx0 + 1;
            h = (int)floor(mxy) + 4 - by0 + 1;
            staged = w > 0 && h > 0 && w <= kLdsFloats && h <= kLdsFloats && w * h <= kLdsFloats;
        }
        ti.bx0 = bx0; ti.by0 = by0; ti.fw = w; ti.fh = h;
        ti.staged = staged;
        ti.sane = sane;
        ti.interior = staged && bx0 >= 0 && by0 >= 0 && bx0 + w <= w_in && by0 + h <= h_in && y0 + kTileH <= h_out;
    }
    __syncthreads();
    const bool staged = ti.staged != 0, interior = ti.interior != 0;
    if (staged) {
        const int bx0 = ti.bx0, by0 = ti.by0, fw = ti.fw, h = ti.fh;
        // branch-free fill: a wave takes every 4th footprint row (row address math is scalar), 3 rows and
        // up to 2 x 64 columns per trip with clamped - always valid - addresses, so that all the loads of
        // a trip are in flight together; validity is applied afterwards
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
        // HAS_MASK is a template flag: as a run-time test every mask load became a branch followed by a
        // full wait, which serialised the whole batch of loads
        constexpr int RU = 3;
        for (int r0 = wave; r0 < h; r0 += 4 * RU) {
            for (int c0 = 0; c0 < fw; c0 += 2 * kWave) {
                float val[RU][2];
                uint8_t mk[RU][2];
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const int row = by0 + r0 + 4 * u;
                    const int rc = row < 0 ? 0 : (row >= h_in ? h_in - 1 : row);
                    const float *rp = fv.src + (int64_t)rc * w_in;
                    const uint8_t *mp = HAS_MASK ? mask + (int64_t)rc * w_in : nullptr;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int col = bx0 + c0 + q * kWave + lane;
                        const int cc = col < 0 ? 0 : (col >= w_in ? w_in - 1 : col);
                        val[u][q] = rp[cc];
                        if constexpr (HAS_MASK) mk[u][q] = mp[cc];
                        else mk[u][q] = 0;
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const int r = r0 + 4 * u;
                    const int row = by0 + r;
                    const bool row_ok = row >= 0 && row < h_in;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int c = c0 + q * kWave + lane;
                        const int col = bx0 + c;
                        const bool good = row_ok && col >= 0 && col < w_in && (fabsf(val[u][q]) < __builtin_inff()) && mk[u][q] == 0;
                        if (r < h && c < fw) tile[r * fw + c] = good ? val[u][q] : __builtin_nanf("");
                    }
                }
            }
        }
    }
    __syncthreads();

    // lane -> output column x0 + lx and the rows y0 + ly, + 4, + 8, + 12
    const int lx = threadIdx.x % kTileW, ly = threadIdx.x / kTileW;
    const int x = x0 + lx;
    if (x >= w_out) return;
    const int yb0 = y0 + ly;
    const int sh = 32 - log2_phases;
    const int64_t o0 = (f * h_out + yb0) * (int64_t)w_out + x;
    float *op = out + o0;
    uint8_t *wp = wout ? wout + o0 : nullptr;
    const int64_t row_stride = 4 * (int64_t)w_out;
    if (interior) resample_pixels<true>(ti, fv, tile, lut, sh, x, yb0, h_out, row_stride, op, wp);
    else resample_pixels<false>(ti, fv, tile, lut, sh, x, yb0, h_out, row_stride, op, wp);
}

}  // namespace

extern "C" int apgpu_resample_affine_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                         const double *affines, int32_t affines_per_tile, int32_t conserve_flux,
                                         const float *fscale, const float *lut, int32_t n_phases, float *out,
                                         uint8_t *weight_out, int64_t h_out, int64_t w_out, void *stream)
{
    if (!frames || !affines || !lut || !out) return fail(APGPU_EINVAL, "resample_affine: NULL pointer argument");
    if (n_frames <= 0 || n_frames > 65535) return fail(APGPU_EINVAL, "resample_affine: n_frames = %d (1..65535)", n_frames);
    if (h_in < 6 || w_in < 6 || h_out <= 0 || w_out <= 0) return fail(APGPU_EINVAL, "resample_affine: bad shape");
    if (h_in > 0x3fffffff || w_in > 0x3fffffff || h_out > 0x3fffffff || w_out > 0x3fffffff)
        return fail(APGPU_EUNSUPPORTED, "resample_affine: image sides are limited to 2^30 pixels");
    if (n_phases < 2 || n_phases > (1 << 20) || (n_phases & (n_phases - 1)))
        return fail(APGPU_EINVAL, "resample_affine: n_phases = %d (a power of two, 2 .. 2^20: the phase is the top bits of a 32-bit fraction)", n_phases);
    int log2_phases = 0;
    while ((1 << log2_phases) < n_phases) log2_phases++;
    if (reinterpret_cast<uintptr_t>(lut) & 7) return fail(APGPU_EINVAL, "resample_affine: lut must be 8-byte aligned");
    const int64_t gx = (w_out + kTileW - 1) / kTileW, gy = (h_out + kTileH - 1) / kTileH;
    if (gx > 0x7fffffffLL || gy > 65535) return fail(APGPU_EUNSUPPORTED, "resample_affine: output too large");
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)n_frames);
    if (mask)
        hipLaunchKernelGGL(resample_affine_kernel<true>, grid, dim3(256), 0, st, frames, mask, affines, affines_per_tile, conserve_flux, fscale, lut, log2_phases, out,
                           weight_out, (int)h_in, (int)w_in, (int)h_out, (int)w_out);
    else
        hipLaunchKernelGGL(resample_affine_kernel<false>, grid, dim3(256), 0, st, frames, mask, affines, affines_per_tile, conserve_flux, fscale, lut, log2_phases, out,
                           weight_out, (int)h_in, (int)w_in, (int)h_out, (int)w_out);
    return check_launch("resample_affine");
}

// ---- OVERSAMPLING n and COMBINE_TYPE WEIGHTED (include/apgpu.h, F3 continued) --------------------------------------
namespace {

__global__ __launch_bounds__(256) void block_mean_kernel(const float *__restrict__ fine, int64_t h, int64_t w, int n, float *__restrict__ out)
{
    const int64_t P = h * w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t wf = w * n;
    const double inv = 1.0 / (double)(n * n);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const int64_t i = p / w, j = p - i * w;
        const float *src = fine + i * n * wf + j * n;
        double acc = 0.0;                                       // a NaN sub-sample makes the sum, and the pixel, NaN
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) acc += (double)src[a * wf + b];
        out[p] = (float)(acc * inv);
    }
}

__global__ __launch_bounds__(256) void weighted_mean_kernel(const float *__restrict__ slab, int n_frames, int64_t P,
                                                           const float *__restrict__ weights, float *__restrict__ mean_out,
                                                           float *__restrict__ wsum_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        double num = 0.0, den = 0.0;
        for (int i = 0; i < n_frames; i++) {
            const float x = slab[(int64_t)i * P + p];
            if (fabsf(x) < __builtin_inff()) {                  // false for NaN and +-inf
                const double wi = (double)weights[i];
                num += wi * (double)x;
                den += wi;
            }
        }
        if (mean_out) mean_out[p] = den > 0.0 ? (float)(num / den) : __builtin_nanf("");
        if (wsum_out) wsum_out[p] = (float)den;
    }
}

}  // namespace

extern "C" int apgpu_block_mean_f32(const float *fine, int64_t h_out, int64_t w_out, int32_t oversampling, float *out, void *stream)
{
    if (!fine || !out) return fail(APGPU_EINVAL, "block_mean: NULL pointer argument");
    if (h_out <= 0 || w_out <= 0 || oversampling < 1 || oversampling > 16)
        return fail(APGPU_EINVAL, "block_mean: bad shape / oversampling %d (1..16)", oversampling);
    int64_t g = (h_out * w_out + 255) / 256;
    if (g > kNumCU * 16) g = kNumCU * 16;
    hipLaunchKernelGGL(block_mean_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), fine, h_out, w_out, oversampling, out);
    return check_launch("block_mean");
}

extern "C" int apgpu_weighted_mean_f32(const float *slab, int32_t n_frames, int64_t n_pixels, const float *weights, float *mean_out,
                                       float *wsum_out, void *stream)
{
    if (!slab || !weights || (!mean_out && !wsum_out)) return fail(APGPU_EINVAL, "weighted_mean: NULL pointer argument");
    if (n_frames < 1 || n_pixels <= 0) return fail(APGPU_EINVAL, "weighted_mean: bad shape");
    int64_t g = (n_pixels + 255) / 256;
    if (g > kNumCU * 16) g = kNumCU * 16;
    hipLaunchKernelGGL(weighted_mean_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), slab, n_frames, n_pixels, weights,
                       mean_out, wsum_out);
    return check_launch("weighted_mean");
}
