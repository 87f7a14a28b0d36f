// resample.hip - F3: affine Lanczos-3 resample of registered frames on gfx950 (the step the reference
// delegates to SWarp: scripts/resample_all.sh:123-131 RESAMPLING_TYPE LANCZOS3, :298 FSCALE = 1/EXPTIME,
// :330-342 the swarp call).  There is no in-tree arithmetic to match; the definition is this build's own and
// is restated on the CPU in oracle/apref.c (apref_resample_affine_f32) - the two agree bit for bit.
//
//   xin = fma(A0, x, fma(A1, y, A2)), yin = fma(A3, x, fma(A4, y, A5))           (float64, output -> input)
//   ix = floor(xin), phase px = (int)((xin - ix) * n_phases + 0.5); taps ix-2 .. ix+3 weighted by lut[px][0..5]
//   (a host-built table of normalised Lanczos-3 weights), rows combined by lut[py]; fmaf chains in a fixed
//   order; result * fscale[f].  Any tap outside the frame, masked or non-finite -> NaN, weight 0.
//
// A workgroup produces a 64 x 16 output tile.  For registration-sized transforms (small rotation / shift /
// scale near 1) the tile's input footprint is ~70 x 22 pixels: it is staged in LDS once (coalesced rows,
// invalid pixels stored as NaN) and the 36 taps of every output pixel are LDS reads, so HBM traffic is one
// read of the input (+ ~40 % halo, mostly L2 hits) and one write of the output: 8 B per pixel.  A footprint
// that does not fit (strong shear / large scale) takes the direct-gather path, same arithmetic.
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kTileW = 64, kTileH = 16;
constexpr int kLdsFloats = 12288;              // 48 KB: three workgroups per CU

__device__ __forceinline__ float fetch_global(const float *__restrict__ src, const uint8_t *__restrict__ mask, int64_t h_in,
                                              int64_t w_in, int64_t row, int64_t col)
{
    if (row < 0 || row >= h_in || col < 0 || col >= w_in) return __builtin_nanf("");
    const int64_t q = row * w_in + col;
    const float v = src[q];
    const bool bad = !(fabsf(v) < __builtin_inff()) || (mask && mask[q] != 0);
    return bad ? __builtin_nanf("") : v;
}

template <bool LDS>
__global__ __launch_bounds__(256) void resample_affine_kernel(const float *__restrict__ frames, const uint8_t *__restrict__ mask,
                                                             const double *__restrict__ affines,
                                                             const float *__restrict__ fscale, const float *__restrict__ lut,
                                                             int n_phases, float *__restrict__ out, uint8_t *__restrict__ wout,
                                                             int64_t h_in, int64_t w_in, int64_t h_out, int64_t w_out)
{
    __shared__ float tile[LDS ? kLdsFloats : 1];
    const int64_t f = blockIdx.z;
    const int64_t x0 = (int64_t)blockIdx.x * kTileW, y0 = (int64_t)blockIdx.y * kTileH;
    const double *A = affines + 6 * f;
    const double a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5];
    const float fs = fscale ? fscale[f] : 1.0f;
    const float *src = frames + f * h_in * w_in;

    // input footprint of the tile: an affine map takes its extremes at the tile corners
    int64_t bx0 = 0, by0 = 0;
    int fw = 0;
    bool staged = false;
    if constexpr (LDS) {
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
        const bool sane = (mnx > -1e15) && (mxx < 1e15) && (mny > -1e15) && (mxy < 1e15);   // false for NaN too
        if (sane) {
            bx0 = (int64_t)floor(mnx) - 2;
            by0 = (int64_t)floor(mny) - 2;
            const int64_t w = (int64_t)floor(mxx) + 3 - bx0 + 1, h = (int64_t)floor(mxy) + 3 - by0 + 1;
            if (w > 0 && h > 0 && w <= kLdsFloats && h <= kLdsFloats && w * h <= kLdsFloats) {
                staged = true;
                fw = (int)w;
                const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
                for (int r = wave; r < (int)h; r += 256 / kWave)
                    for (int c = lane; c < fw; c += kWave) tile[r * fw + c] = fetch_global(src, mask, h_in, w_in, by0 + r, bx0 + c);
            }
        }
        __syncthreads();
    }

    const int lx = threadIdx.x % kTileW, ly = threadIdx.x / kTileW;
    const int64_t x = x0 + lx;
    if (x >= w_out) return;
#pragma unroll
    for (int k = 0; k < kTileH / 4; k++) {
        const int64_t y = y0 + ly + 4 * k;
        if (y >= h_out) break;
        const double xin = fma(a0, (double)x, fma(a1, (double)y, a2));
        const double yin = fma(a3, (double)x, fma(a4, (double)y, a5));
        float res = __builtin_nanf("");
        if (xin >= 2.0 && yin >= 2.0 && xin < (double)(w_in - 3) && yin < (double)(h_in - 3)) {
            const double fx0 = floor(xin), fy0 = floor(yin);
            const int64_t ix = (int64_t)fx0, iy = (int64_t)fy0;
            const int px = (int)((xin - fx0) * (double)n_phases + 0.5);
            const int py = (int)((yin - fy0) * (double)n_phases + 0.5);
            const float2 *wxp = reinterpret_cast<const float2 *>(lut + 6 * px);
            const float2 *wyp = reinterpret_cast<const float2 *>(lut + 6 * py);
            const float2 wx01 = wxp[0], wx23 = wxp[1], wx45 = wxp[2];
            const float2 wy01 = wyp[0], wy23 = wyp[1], wy45 = wyp[2];
            const float wx[6] = {wx01.x, wx01.y, wx23.x, wx23.y, wx45.x, wx45.y};
            const float wy[6] = {wy01.x, wy01.y, wy23.x, wy23.y, wy45.x, wy45.y};
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                float s[6];
                if (LDS && staged) {
                    const float *t = tile + (int)(iy - 2 + j - by0) * fw + (int)(ix - 2 - bx0);
#pragma unroll
                    for (int i = 0; i < 6; i++) s[i] = t[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 6; i++) s[i] = fetch_global(src, mask, h_in, w_in, iy - 2 + j, ix - 2 + i);
                }
                float r = wx[0] * s[0];
#pragma unroll
                for (int i = 1; i < 6; i++) r = fmaf(wx[i], s[i], r);
                v = (j == 0) ? wy[0] * r : fmaf(wy[j], r, v);
            }
            // invalid taps arrive as NaN and poison v; the weight plane is "out is not NaN" in the oracle too
            res = (v == v) ? v * fs : __builtin_nanf("");
        }
        const int64_t o = (f * h_out + y) * w_out + x;
        out[o] = res;
        if (wout) wout[o] = (res == res) ? 1 : 0;
    }
}

}  // namespace

extern "C" int apgpu_resample_affine_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                         const double *affines, const float *fscale, const float *lut, int32_t n_phases,
                                         float *out, uint8_t *weight_out, int64_t h_out, int64_t w_out, void *stream)
{
    if (!frames || !affines || !lut || !out) return fail(APGPU_EINVAL, "resample_affine: NULL pointer argument");
    if (n_frames <= 0 || n_frames > 65535) return fail(APGPU_EINVAL, "resample_affine: n_frames = %d (1..65535)", n_frames);
    if (h_in < 6 || w_in < 6 || h_out <= 0 || w_out <= 0) return fail(APGPU_EINVAL, "resample_affine: bad shape");
    if (n_phases < 1 || n_phases > (1 << 20)) return fail(APGPU_EINVAL, "resample_affine: n_phases = %d", n_phases);
    if (reinterpret_cast<uintptr_t>(lut) & 7) return fail(APGPU_EINVAL, "resample_affine: lut must be 8-byte aligned");
    const int64_t gx = (w_out + kTileW - 1) / kTileW, gy = (h_out + kTileH - 1) / kTileH;
    if (gx > 0x7fffffffLL || gy > 65535) return fail(APGPU_EUNSUPPORTED, "resample_affine: output too large");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(resample_affine_kernel<true>, dim3((unsigned)gx, (unsigned)gy, (unsigned)n_frames), dim3(256), 0, st, frames,
                       mask, affines, fscale, lut, n_phases, out, weight_out, h_in, w_in, h_out, w_out);
    return check_launch("resample_affine");
}
