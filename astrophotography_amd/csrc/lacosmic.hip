// lacosmic.hip - F4 (second half): L.A.Cosmic cosmic-ray rejection, the step ApFixCosmicRays hands to
// ccdproc.cosmicray_lacosmic -> astroscrappy.detect_cosmics (core/ApFixCosmicRays.py:267-295: sigclip 4.5, sigfrac 0.3,
// objlim 5, readnoise 12 e-, niter 6, fsmode 'convolve' with a 3.5-pixel Gaussian, sepmed, cleantype 'meanmask').
// astroscrappy / ccdproc are absent from the build container: PARITY UNPINNED.  The kernels implement van Dokkum's (2001)
// algorithm in the structure of astroscrappy's detect_cosmics as restated in oracle/lacosmic_ref.py (float32 planes):
//   per iteration:  s  = rebin(clip0(laplace(subsample2(clean))))            Laplacian of the 2x subsampled image
//                   m5 = sepmed7(clean); noise = sqrt(max(m5, 1e-5) + rn^2); s /= 2 noise; sp = s - sepmed7(s)
//                   f  = conv(clean, psf 7x7); f = max((f - sepmed9(f)) / noise, 0.01)   fine-structure image
//                   cr = good & (sp > sigclip) & (sp / f > objlim); grown twice by 3x3 with sp > sigclip, sp > sigfrac*sigclip
//                   clean[cr] = mean of the non-CR, unmasked pixels of the 5x5 neighbourhood (background level if none)
//   "sepmedK" = median of K along rows, then of K along columns, border pixels copied (astroscrappy's separable filters).
// Everything is one pass over the image per step (HBM / L2 streaming); the 1-D medians sort 5 / 7 / 9 values in registers
// with the compile-time networks of the stack kernels.
#include "common.h"
#include "stack_sort.h"

namespace {
using namespace apgpu;

constexpr int kBlock = 256;

inline unsigned grid1d(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kNumCU * 16) g = kNumCU * 16;
    return (unsigned)g;
}

#define APGPU_FOR_PIXELS(p, P) \
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, _stride = (int64_t)gridDim.x * blockDim.x; p < (P); p += _stride)

// What the second (column) pass of a separable median does with the median m of pixel p - the elementwise kernels that
// used to follow it, fused in (same operations, same order):
//   POST_STORE   out[p] = m
//   POST_MINUS   out[p] = aux[p] - m                                   sp = s - sepmed7(s)
//   POST_NOISE   n = sqrt(max(m, 1e-5) + rn2); out[p] = n; aux2[p] = aux2[p] / (2 n)     noise model and significance map
//   POST_FINE    out[p] = max((aux[p] - m) / aux2[p], 0.01)            fine-structure image (aux = f, aux2 = noise)
enum { POST_STORE = 0, POST_MINUS = 1, POST_NOISE = 2, POST_FINE = 3 };

// 1-D median of K (odd) along rows (ALONG_X) or columns; pixels closer than K/2 to the border are copied.
template <int K, bool ALONG_X, int POST = POST_STORE>
__global__ __launch_bounds__(kBlock) void median1d_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W,
                                                         const float *__restrict__ aux = nullptr, float *__restrict__ aux2 = nullptr,
                                                         float rn2 = 0.f)
{
    constexpr int NP = K <= 8 ? 8 : 16;
    constexpr int HALF = K / 2;
    const int64_t P = (int64_t)H * W;
    APGPU_FOR_PIXELS(p, P) {
        const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
        const int pos = ALONG_X ? c : r, len = ALONG_X ? W : H;
        float m;
        if (pos < HALF || pos >= len - HALF) {
            m = in[p];
        } else {
            float v[NP];
#pragma unroll
            for (int k = 0; k < NP; k++) {
                if (k < K) v[k] = in[ALONG_X ? p + (k - HALF) : p + (int64_t)(k - HALF) * W];
                else v[k] = __builtin_inff();
            }
            apgpu_stack::sort_column<NP>(v);
            m = v[HALF];
        }
        if constexpr (POST == POST_STORE) {
            out[p] = m;
        } else if constexpr (POST == POST_MINUS) {
            out[p] = aux[p] - m;
        } else if constexpr (POST == POST_NOISE) {
            const float n = sqrtf(fmaxf(m, 0.00001f) + rn2);
            out[p] = n;
            aux2[p] = aux2[p] / (2.0f * n);
        } else {
            out[p] = fmaxf((aux[p] - m) / aux2[p], 0.01f);
        }
    }
}

// Laplacian of the 2x2-subsampled image, negative values clipped, block-averaged back (kernel 0 -1 0 / -1 4 -1 / 0 -1 0;
// a neighbour outside the image is dropped).  For pixel v with neighbours u, d, l, r the four sub-pixels give
// 2v - u - l, 2v - u - r, 2v - d - l, 2v - d - r (a missing neighbour leaves its own term out: e.g. 3v - l ... see oracle).
__global__ __launch_bounds__(kBlock) void laplace_kernel(const float *__restrict__ a, float *__restrict__ s, int H, int W)
{
    const int64_t P = (int64_t)H * W;
    APGPU_FOR_PIXELS(p, P) {
        const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
        const float v = a[p];
        const bool hu = r > 0, hd = r < H - 1, hl = c > 0, hr = c < W - 1;
        const float u = hu ? a[p - W] : 0.f, d = hd ? a[p + W] : 0.f, l = hl ? a[p - 1] : 0.f, rr = hr ? a[p + 1] : 0.f;
        // sub-pixel (top-left): neighbours up (pixel above, or outside), left (pixel to the left, or outside), right and down are v itself
        const float tl = 4.f * v - u - l - v - v;
        const float tr = 4.f * v - u - rr - v - v;
        const float bl = 4.f * v - d - l - v - v;
        const float br = 4.f * v - d - rr - v - v;
        s[p] = (fmaxf(tl, 0.f) + fmaxf(tr, 0.f) + fmaxf(bl, 0.f) + fmaxf(br, 0.f)) * 0.25f;
    }
}

// f = sum_k psf[k] * clean[shifted]  (7 x 7, zero outside the image), accumulated in row-major kernel order.
// A workgroup stages a (64 + 6) x (16 + 6) input tile in LDS (zeros outside the image: adding k * 0 leaves the running sum
// as it is, so the tile version equals "skip the tap" bit for bit) and a lane forms 4 neighbouring outputs from 7 x 10
// LDS values, each output its own 49-step multiply-then-add chain in the reference order.
constexpr int kConvTW = 64, kConvTH = 16, kConvLW = kConvTW + 6 + 2;       // LDS row: 70 values + 2 pad
__global__ __launch_bounds__(kBlock) void convolve7_kernel(const float *__restrict__ a, const float *__restrict__ psf, float *__restrict__ f,
                                                          int H, int W)
{
    __shared__ float tile[(kConvTH + 6) * kConvLW];
    const int tilesx = (W + kConvTW - 1) / kConvTW, tilesy = (H + kConvTH - 1) / kConvTH;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    for (int t = blockIdx.x; t < tilesx * tilesy; t += gridDim.x) {
        const int x0 = (t % tilesx) * kConvTW, y0 = (t / tilesx) * kConvTH;
        __syncthreads();
        for (int e = threadIdx.x; e < (kConvTH + 6) * (kConvTW + 6); e += kBlock) {
            const int ly = e / (kConvTW + 6), lx = e - ly * (kConvTW + 6);
            const int gy = y0 + ly - 3, gx = x0 + lx - 3;
            tile[ly * kConvLW + lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? a[(int64_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 7; dy++) {
            float row[10];
#pragma unroll
            for (int q = 0; q < 10; q++) row[q] = tile[(ty + dy) * kConvLW + 4 * tx + q];
#pragma unroll
            for (int dx = 0; dx < 7; dx++) {
                const float k = psf[dy * 7 + dx];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = acc[j] + k * row[j + dx];
            }
        }
        const int gy = y0 + ty;
        if (gy < H) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gx = x0 + 4 * tx + j;
                if (gx < W) f[(int64_t)gy * W + gx] = acc[j];
            }
        }
    }
}

// out = dilate(in) [& !mask] [& sp > thr].  SHAPE 3: 3x3 square; SHAPE 5: 5x5 without its corners.  Zero outside the image.
// Four pixels per lane (one 16-byte read of sp, one 4-byte store); the cheap conditions come first - a grown cosmic ray also
// has to stand above its threshold, which almost no pixel does - and only the pixels that pass look at their neighbours.
template <int SHAPE>
__device__ __forceinline__ bool dilate_any(const uint8_t *__restrict__ in, int64_t p, int H, int W)
{
    constexpr int R = SHAPE / 2;
    const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
    bool any = false;
#pragma unroll
    for (int dy = -R; dy <= R; dy++)
#pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            if (SHAPE == 5 && (dy == -2 || dy == 2) && (dx == -2 || dx == 2)) continue;
            const int rr = r + dy, cc = c + dx;
            if (rr >= 0 && rr < H && cc >= 0 && cc < W) any = any || in[(int64_t)rr * W + cc] != 0;
        }
    return any;
}

template <int SHAPE>
__global__ __launch_bounds__(kBlock) void dilate_kernel(const uint8_t *__restrict__ in, const float *__restrict__ sp, const uint8_t *__restrict__ mask,
                                                       float thr, uint8_t *__restrict__ out, int H, int W)
{
    const int64_t P = (int64_t)H * W;
    const int64_t P4 = P / 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask)) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
    const bool rows_aligned = (W % 4) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0;      // every row starts on a word
    if (vec) {
        APGPU_FOR_PIXELS(q, P4) {
            const int64_t p = 4 * q;
            bool cand[4] = {true, true, true, true};
            if (sp) {
                const float4 v = reinterpret_cast<const float4 *>(sp)[q];
                cand[0] = v.x > thr; cand[1] = v.y > thr; cand[2] = v.z > thr; cand[3] = v.w > thr;
            }
            if (mask) {
                const unsigned m = reinterpret_cast<const unsigned *>(mask)[q];
#pragma unroll
                for (int j = 0; j < 4; j++) cand[j] = cand[j] && ((m >> (8 * j)) & 0xffu) == 0;
            }
            // without a threshold to test first (the saturation mask): the map is sparse, so look at the neighbourhood of the
            // four pixels as 32-bit words - 3 words x (2R+1) rows - and only walk it pixel by pixel when one is non-zero
            if (!sp && rows_aligned) {
                constexpr int R = SHAPE / 2;
                const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);          // c is a multiple of 4
                unsigned seen = 0;
#pragma unroll
                for (int dy = -R; dy <= R; dy++) {
                    const int rr = r + dy;
                    if (rr < 0 || rr >= H) continue;
                    const unsigned *row = reinterpret_cast<const unsigned *>(in + (int64_t)rr * W);
                    seen |= row[c / 4];
                    if (c >= 4) seen |= row[c / 4 - 1];
                    if (c + 4 < W) seen |= row[c / 4 + 1];
                }
                if (seen == 0) {
                    reinterpret_cast<unsigned *>(out)[q] = 0;
                    continue;
                }
            }
            unsigned o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (cand[j] && dilate_any<SHAPE>(in, p + j, H, W)) o |= 1u << (8 * j);
            reinterpret_cast<unsigned *>(out)[q] = o;
        }
    }
    const int64_t done = vec ? 4 * P4 : 0;
    APGPU_FOR_PIXELS(t, P - done) {
        const int64_t p = done + t;
        const bool cand = !(mask && mask[p]) && !(sp && !(sp[p] > thr));
        out[p] = (cand && dilate_any<SHAPE>(in, p, H, W)) ? 1 : 0;
    }
}

// Candidate selection + first growth step in one pass: out = dilate3(cr) & !mask & (sp > sigclip) with
// cr[q] = !mask[q] & (sp[q] > sigclip) & (sp[q] / f[q] > objlim) evaluated on the fly for the nine neighbours of the few
// pixels that pass the cheap conditions themselves (the candidate map is never written).
__device__ __forceinline__ bool selected(const float *__restrict__ sp, const float *__restrict__ f, const uint8_t *__restrict__ mask,
                                         float sigclip, float objlim, int64_t q)
{
    const float x = sp[q];
    return !(mask && mask[q]) && x > sigclip && (x / f[q]) > objlim;
}

__global__ __launch_bounds__(kBlock) void select_grow_kernel(const float *__restrict__ sp, const float *__restrict__ f,
                                                            const uint8_t *__restrict__ mask, float sigclip, float objlim,
                                                            uint8_t *__restrict__ out, int H, int W)
{
    const int64_t P = (int64_t)H * W;
    auto grown = [&](int64_t p) -> bool {
        const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
        bool any = false;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int rr = r + dy, cc = c + dx;
                if (rr >= 0 && rr < H && cc >= 0 && cc < W) any = any || selected(sp, f, mask, sigclip, objlim, (int64_t)rr * W + cc);
            }
        return any;
    };
    const int64_t P4 = P / 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask)) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(sp) & 15) == 0;
    if (vec) {
        APGPU_FOR_PIXELS(q, P4) {
            const float4 v = reinterpret_cast<const float4 *>(sp)[q];
            bool cand[4] = {v.x > sigclip, v.y > sigclip, v.z > sigclip, v.w > sigclip};
            if (mask) {
                const unsigned m = reinterpret_cast<const unsigned *>(mask)[q];
#pragma unroll
                for (int j = 0; j < 4; j++) cand[j] = cand[j] && ((m >> (8 * j)) & 0xffu) == 0;
            }
            unsigned o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (cand[j] && grown(4 * q + j)) o |= 1u << (8 * j);
            reinterpret_cast<unsigned *>(out)[q] = o;
        }
    }
    const int64_t done = vec ? 4 * P4 : 0;
    APGPU_FOR_PIXELS(t, P - done) {
        const int64_t p = done + t;
        const bool cand = !(mask && mask[p]) && sp[p] > sigclip;
        out[p] = (cand && grown(p)) ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void merge_count_kernel(const uint8_t *__restrict__ cr, uint8_t *__restrict__ crmask,
                                                            unsigned long long *__restrict__ ncr, int64_t P)
{
    unsigned n = 0;
    APGPU_FOR_PIXELS(p, P) {
        if (cr[p]) {
            n++;
            crmask[p] = 1;
        }
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) n += __shfl_down(n, d);
    if ((threadIdx.x % kWave) == 0 && n) atomicAdd(ncr, (unsigned long long)n);
}

// cleantype 'meanmask': every CR pixel at least 2 pixels from the border becomes the mean of the pixels of its 5x5
// neighbourhood that are neither CR nor masked (the background level if there is none).  Only CR pixels are written and
// only non-CR pixels are read: in place.
__device__ __forceinline__ void clean_pixel(float *__restrict__ a, const uint8_t *__restrict__ crmask, const uint8_t *__restrict__ mask,
                                            float background, int64_t p, int H, int W)
{
    const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
    if (r < 2 || r >= H - 2 || c < 2 || c >= W - 2) return;
    float sum = 0.f;
    int n = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; dy++)
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            const int64_t q = p + (int64_t)dy * W + dx;
            if (!crmask[q] && !(mask && mask[q])) {
                sum = sum + a[q];
                n++;
            }
        }
    a[p] = n > 0 ? sum / (float)n : background;
}

__global__ __launch_bounds__(kBlock) void clean_meanmask_kernel(float *__restrict__ a, const uint8_t *__restrict__ crmask,
                                                               const uint8_t *__restrict__ mask, float background, int H, int W)
{
    const int64_t P = (int64_t)H * W;
    // the cosmic-ray map is almost empty: it is scanned four pixels per 32-bit word
    const int64_t P4 = (reinterpret_cast<uintptr_t>(crmask) & 3) == 0 ? P / 4 : 0;
    APGPU_FOR_PIXELS(q, P4) {
        const unsigned m = reinterpret_cast<const unsigned *>(crmask)[q];
        if (m == 0) continue;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((m >> (8 * j)) & 0xffu) clean_pixel(a, crmask, mask, background, 4 * q + j, H, W);
    }
    APGPU_FOR_PIXELS(t, P - 4 * P4) {
        const int64_t p = 4 * P4 + t;
        if (crmask[p]) clean_pixel(a, crmask, mask, background, p, H, W);
    }
}

__global__ __launch_bounds__(kBlock) void saturated_kernel(const float *__restrict__ a, const float *__restrict__ m5, float satlevel,
                                                          uint8_t *__restrict__ sat, int64_t P)
{
    APGPU_FOR_PIXELS(p, P) sat[p] = (a[p] >= satlevel && m5[p] > satlevel / 10.0f) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void or_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, int64_t P)
{
    APGPU_FOR_PIXELS(p, P) out[p] = (a[p] || (b && b[p])) ? 1 : 0;
}

template <int K, int POST = POST_STORE>
void sepmed(const float *in, float *out, float *tmp, int H, int W, hipStream_t st, const float *aux = nullptr, float *aux2 = nullptr,
            float rn2 = 0.f)
{
    const unsigned g = grid1d((int64_t)H * W);
    hipLaunchKernelGGL((median1d_kernel<K, true>), dim3(g), dim3(kBlock), 0, st, in, tmp, H, W, (const float *)nullptr, (float *)nullptr, 0.f);
    hipLaunchKernelGGL((median1d_kernel<K, false, POST>), dim3(g), dim3(kBlock), 0, st, tmp, out, H, W, aux, aux2, rn2);
}

struct Work {
    float *s, *noise, *f, *t1, *t2;
    uint8_t *u1, *u2;
};

Work carve(void *ws, int64_t P)
{
    Work w;
    float *fp = static_cast<float *>(ws);
    const int64_t Pa = (P + 3) & ~(int64_t)3;
    w.s = fp; w.noise = fp + Pa; w.f = fp + 2 * Pa; w.t1 = fp + 3 * Pa; w.t2 = fp + 4 * Pa;
    w.u1 = reinterpret_cast<uint8_t *>(fp + 5 * Pa);
    w.u2 = w.u1 + Pa;
    return w;
}

}  // namespace

extern "C" size_t apgpu_lacosmic_ws_bytes(int64_t height, int64_t width)
{
    if (height <= 0 || width <= 0) return 0;
    const int64_t Pa = (height * width + 3) & ~(int64_t)3;
    return (size_t)Pa * (5 * sizeof(float) + 2) + 64;
}

extern "C" int apgpu_sepmedfilt_f32(const float *data, int64_t height, int64_t width, int32_t size, float *out, void *ws, size_t ws_bytes,
                                    void *stream)
{
    if (!data || !out || !ws) return fail(APGPU_EINVAL, "sepmedfilt: NULL pointer argument");
    if (height <= 0 || width <= 0 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "sepmedfilt: bad shape");
    if (ws_bytes < (size_t)(height * width) * sizeof(float)) return fail(APGPU_EWORKSPACE, "sepmedfilt: workspace too small");
    hipStream_t st = as_stream(stream);
    float *tmp = static_cast<float *>(ws);
    switch (size) {
    case 5: sepmed<5>(data, out, tmp, (int)height, (int)width, st); break;
    case 7: sepmed<7>(data, out, tmp, (int)height, (int)width, st); break;
    case 9: sepmed<9>(data, out, tmp, (int)height, (int)width, st); break;
    default: return fail(APGPU_EUNSUPPORTED, "sepmedfilt: size %d (5, 7 or 9)", size);
    }
    return check_launch("sepmedfilt");
}

extern "C" int apgpu_lacosmic_satmask(const float *data, const uint8_t *inmask, int64_t height, int64_t width, float satlevel,
                                      uint8_t *mask_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!data || !mask_out || !ws) return fail(APGPU_EINVAL, "lacosmic_satmask: NULL pointer argument");
    if (height <= 0 || width <= 0 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "lacosmic_satmask: bad shape");
    if (ws_bytes < apgpu_lacosmic_ws_bytes(height, width)) return fail(APGPU_EWORKSPACE, "lacosmic_satmask: workspace too small");
    hipStream_t st = as_stream(stream);
    const int H = (int)height, W = (int)width;
    const int64_t P = height * width;
    Work w = carve(ws, P);
    const unsigned g = grid1d(P);
    sepmed<7>(data, w.t1, w.t2, H, W, st);                                                  // large-scale structure
    hipLaunchKernelGGL(saturated_kernel, dim3(g), dim3(kBlock), 0, st, data, w.t1, satlevel, w.u1, P);
    hipLaunchKernelGGL(dilate_kernel<5>, dim3(g), dim3(kBlock), 0, st, w.u1, (const float *)nullptr, (const uint8_t *)nullptr, 0.f, w.u2, H, W);
    hipLaunchKernelGGL(dilate_kernel<5>, dim3(g), dim3(kBlock), 0, st, w.u2, (const float *)nullptr, (const uint8_t *)nullptr, 0.f, w.u1, H, W);
    if (inmask) {
        hipLaunchKernelGGL(dilate_kernel<3>, dim3(g), dim3(kBlock), 0, st, inmask, (const float *)nullptr, (const uint8_t *)nullptr, 0.f, w.u2, H, W);
        hipLaunchKernelGGL(or_kernel, dim3(g), dim3(kBlock), 0, st, w.u1, w.u2, mask_out, P);
    } else {
        hipLaunchKernelGGL(or_kernel, dim3(g), dim3(kBlock), 0, st, w.u1, (const uint8_t *)nullptr, mask_out, P);
    }
    return check_launch("lacosmic_satmask");
}

extern "C" int apgpu_lacosmic_iterate(float *clean, const uint8_t *mask, uint8_t *crmask, int64_t height, int64_t width, float sigclip,
                                      float sigfrac, float objlim, float readnoise, const float *psfk, float background_level,
                                      int64_t *ncr_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!clean || !crmask || !ncr_out || !ws) return fail(APGPU_EINVAL, "lacosmic_iterate: NULL pointer argument");
    if (height < 5 || width < 5 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "lacosmic_iterate: bad shape");
    if (ws_bytes < apgpu_lacosmic_ws_bytes(height, width)) return fail(APGPU_EWORKSPACE, "lacosmic_iterate: workspace too small");
    hipStream_t st = as_stream(stream);
    const int H = (int)height, W = (int)width;
    const int64_t P = height * width;
    Work w = carve(ws, P);
    const unsigned g = grid1d(P);
    if (hipMemsetAsync(ncr_out, 0, sizeof(int64_t), st) != hipSuccess) return fail(APGPU_ELAUNCH, "lacosmic_iterate: memset failed");
    hipLaunchKernelGGL(laplace_kernel, dim3(g), dim3(kBlock), 0, st, clean, w.s, H, W);
    // noise = sqrt(max(sepmed7(clean), 1e-5) + rn^2), s /= 2 noise - in the median's column pass
    sepmed<7, POST_NOISE>(clean, w.noise, w.t2, H, W, st, nullptr, w.s, readnoise * readnoise);
    sepmed<7, POST_MINUS>(w.s, w.t1, w.t2, H, W, st, w.s);                                  // sp = s - sepmed7(s)  -> t1
    const float *sp = w.t1;
    if (psfk) {
        const int64_t tiles = (int64_t)((W + kConvTW - 1) / kConvTW) * ((H + kConvTH - 1) / kConvTH);
        hipLaunchKernelGGL(convolve7_kernel, dim3((unsigned)(tiles < kNumCU * 64 ? tiles : kNumCU * 64)), dim3(kBlock), 0, st, clean, psfk,
                           w.f, H, W);
    } else {
        sepmed<5>(clean, w.f, w.t2, H, W, st);                                              // fsmode 'median'
    }
    sepmed<9, POST_FINE>(w.f, w.s, w.t2, H, W, st, w.f, w.noise);                           // fine structure -> s (free by now)
    const float *fine = w.s;
    hipLaunchKernelGGL(select_grow_kernel, dim3(g), dim3(kBlock), 0, st, sp, fine, mask, sigclip, objlim, w.u2, H, W);
    hipLaunchKernelGGL(dilate_kernel<3>, dim3(g), dim3(kBlock), 0, st, w.u2, sp, mask, sigfrac * sigclip, w.u1, H, W);
    hipLaunchKernelGGL(merge_count_kernel, dim3(g), dim3(kBlock), 0, st, w.u1, crmask, reinterpret_cast<unsigned long long *>(ncr_out), P);
    hipLaunchKernelGGL(clean_meanmask_kernel, dim3(g), dim3(kBlock), 0, st, clean, crmask, mask, background_level, H, W);
    return check_launch("lacosmic_iterate");
}
