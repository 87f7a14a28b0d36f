// background.hip - F4 (first half): the sky-background mesh of ApMeasureBackground (core/ApMeasureBackground.py:142-175,
// 382-415) on gfx950.  The reference delegates everything to photutils (absent from the build container: parity
// unpinned; the definitions below restate photutils' published algorithms and are what oracle/background_ref.py computes):
//   source mask     detect_threshold(nsigma=2, SigmaClip(3, maxiters=10)) -> detect_sources(npixels=5, 8-connectivity)
//                   -> make_source_mask(size=13)                                   (:154-157)
//   mesh            Background2D(box_size, mask, exclude_percentile, SigmaClip(sigma), MedianBackground)   (:404-410)
//   full image      BkgZoomInterpolator = scipy.ndimage.zoom(mesh, box_size, order=3, mode='reflect', grid_mode=True)
// Kernels here do the per-pixel work: 8-connected component labelling with a minimum-area filter (lock-free union-find),
// square binary dilation, per-box sigma-clipped median / std (one workgroup per box, exact radix-select medians) and the
// cubic B-spline evaluation of the mesh over every pixel.  The mesh-sized steps between them (ny x nx numbers: filling
// excluded boxes, the 3 x 3 median filter, the spline prefilter) are host logic in core/ApMeasureBackground.py.
#include "common.h"

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

// ------------------------------------------------------------------------------------------------
// Connected components (8-connectivity) by lock-free union-find on the label array: label[p] = p for
// foreground pixels, every pixel is united with its W / NW / N / NE foreground neighbours (the other four
// directions are covered from the neighbour's side), then every pixel looks up its root.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int *L, int i)
{
    int r = L[i];
    while (r != i) {
        i = r;
        r = L[i];
    }
    return i;
}

__device__ __forceinline__ void uf_union(int *L, int a, int b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[b], a);               // hang the larger root under the smaller one
        if (old == b) return;
        b = old;                                            // somebody moved b meanwhile: retry from there
    }
}

__global__ __launch_bounds__(kBlock) void ccl_init_kernel(const uint8_t *__restrict__ fg, int *__restrict__ L, int *__restrict__ size, int64_t P)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        L[p] = fg[p] ? (int)p : -1;
        size[p] = 0;
    }
}

__global__ __launch_bounds__(kBlock) void ccl_union_kernel(const uint8_t *__restrict__ fg, int *__restrict__ L, int H, int W)
{
    const int64_t P = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        if (!fg[p]) continue;
        const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
        if (c > 0 && fg[p - 1]) uf_union(L, (int)p, (int)p - 1);
        if (r > 0) {
            if (fg[p - W]) uf_union(L, (int)p, (int)(p - W));
            if (c > 0 && fg[p - W - 1]) uf_union(L, (int)p, (int)(p - W - 1));
            if (c < W - 1 && fg[p - W + 1]) uf_union(L, (int)p, (int)(p - W + 1));
        }
    }
}

__global__ __launch_bounds__(kBlock) void ccl_count_kernel(int *__restrict__ L, int *__restrict__ size, int64_t P)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        if (L[p] < 0) continue;
        const int root = uf_find(L, (int)p);
        L[p] = root;                                        // flatten (roots keep L[root] == root)
        atomicAdd(&size[root], 1);
    }
}

__global__ __launch_bounds__(kBlock) void ccl_filter_kernel(const int *__restrict__ L, const int *__restrict__ size, int min_pixels,
                                                           uint8_t *__restrict__ out, unsigned long long *__restrict__ nsrc, int64_t P)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const int l = L[p];
        const bool keep = l >= 0 && size[l >= 0 ? uf_find(L, l) : 0] >= min_pixels;
        out[p] = keep ? 1 : 0;
        if (keep && l == (int)p && nsrc) atomicAdd(nsrc, 1ull);        // one count per surviving component (its root pixel)
    }
}

// ------------------------------------------------------------------------------------------------
// Binary dilation with a size x size square footprint (scipy.ndimage.binary_dilation, border_value 0), separable:
// a pixel is set if any pixel within +-size/2 along the row (pass 1) / column (pass 2) is set.
// ------------------------------------------------------------------------------------------------
template <bool ROWS>
__global__ __launch_bounds__(kBlock) void dilate_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int H, int W, int half)
{
    const int64_t P = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const int r = (int)(p / W), c = (int)(p - (int64_t)r * W);
        uint8_t any = 0;
        if (ROWS) {
            const int c0 = max(0, c - half), c1 = min(W - 1, c + half);
            for (int cc = c0; cc <= c1; cc++) any |= in[(int64_t)r * W + cc];
        } else {
            const int r0 = max(0, r - half), r1 = min(H - 1, r + half);
            for (int rr = r0; rr <= r1; rr++) any |= in[(int64_t)rr * W + c];
        }
        out[p] = any ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-box sigma-clipped statistics: one workgroup per mesh box.  astropy's SigmaClip(sigma, maxiters, median / std)
// along the box: every pass computes the median (exact, 4-pass 8-bit radix select on order-preserving keys) and the
// standard deviation of the current survivors and keeps lo <= x <= hi; clipping only ever shrinks the interval, so the
// survivors of pass k are exactly the unmasked finite values inside the running [lo, hi] - no compaction: a box of up to
// 32768 pixels is staged in LDS once (NaN = masked) and every pass walks the LDS copy; larger boxes re-read the image.  Pixels outside the image (edge_method 'pad'), masked or non-finite
// pixels count as masked.  Output per box (float64): median, std of the final survivors, number of survivors,
// number of masked pixels before clipping.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f32_key(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_f32(unsigned k)
{
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct BoxView {
    const float *data;
    const uint8_t *mask;
    int H, W, r0, c0, bh, bw;
};

// value of box element e (row-major inside the box) or NaN if masked / outside / non-finite
__device__ __forceinline__ float box_value(const BoxView &b, int e)
{
    const int rr = b.r0 + e / b.bw, cc = b.c0 + e % b.bw;
    if (rr >= b.H || cc >= b.W) return __builtin_nanf("");
    const int64_t p = (int64_t)rr * b.W + cc;
    if (b.mask && b.mask[p]) return __builtin_nanf("");
    const float x = b.data[p];
    return (fabsf(x) < __builtin_inff()) ? x : __builtin_nanf("");
}

template <int NT>
__device__ double block_sum(double v, double *scratch)
{
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if ((threadIdx.x % kWave) == 0) scratch[threadIdx.x / kWave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < NT / kWave; w++) t += scratch[w];       // every thread forms the same ordered sum
    return t;
}

constexpr int kBoxDigit = 11;               // radix-select digit: 11 + 11 + 10 bits of the order-preserving float32 key
constexpr int kBoxBins = 1 << kBoxDigit;
constexpr int kBoxLdsPixels = 32768;        // boxes up to this many pixels are staged in LDS (128 KB of the 160 KB)
constexpr unsigned kNoDigit = 0xffffffffu;

struct DigitPick {
    int digit;          // bin that holds the searched rank
    unsigned cum;       // values in the bins below it
    int dlow;           // highest occupied bin below it (-1: none)
};

// Wavefront 0 finds the bin of rank k in hist[0 .. kBoxBins) (32 bins per lane + a wavefront scan) and publishes it.
__device__ __forceinline__ DigitPick pick_digit(const unsigned *hist, unsigned k, DigitPick *shared_pick)
{
    if (threadIdx.x < kWave) {
        const int lane = threadIdx.x;
        constexpr int per = kBoxBins / kWave;
        unsigned tot = 0;
#pragma unroll 4
        for (int j = 0; j < per; j++) tot += hist[lane * per + j];
        unsigned inc = tot;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        const unsigned exc = inc - tot;
        int dlow = -1;
        if (k >= exc && k < inc) {
            unsigned run = exc;
#pragma unroll 1
            for (int j = 0; j < per; j++) {
                const unsigned c = hist[lane * per + j];
                if (k < run + c) { shared_pick->digit = lane * per + j; shared_pick->cum = run; break; }
                if (c) dlow = lane * per + j;
                run += c;
            }
        } else if (k >= inc) {
#pragma unroll 1
            for (int j = per - 1; j >= 0; j--)
                if (hist[lane * per + j]) { dlow = lane * per + j; break; }
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            const int o = __shfl_down(dlow, d);
            dlow = o > dlow ? o : dlow;
        }
        if (lane == 0) shared_pick->dlow = dlow;
    }
    __syncthreads();
    const DigitPick r = *shared_pick;
    __syncthreads();
    return r;
}

// NT threads per box: 256 for small boxes, 1024 for the 16 x 16 meshes of full frames (one workgroup per CU, 16 wavefronts).
// RES: the box (<= kBoxLdsPixels pixels) is staged in LDS once, masked / outside / non-finite pixels as NaN; otherwise every
// pass walks the image rows of the box (coalesced runs, L2 / Infinity Cache resident after the first pass).
// Passes: the first iteration makes four - (a) count + sum, (b) sum of squared deviations + first select level, (c) second
// level, (d) third level + the largest survivor below the found prefix (the lower middle element of an even count comes from
// the last histogram or from (d)'s maximum, not from a second select); every later iteration makes two - count, moments about
// the previous median, first level and a speculated second level in one, then (d).
template <int NT, bool RES>
__global__ __launch_bounds__(NT) void box_stats_kernel(const float *__restrict__ data, const uint8_t *__restrict__ mask, int H, int W,
                                                      int bh, int bw, int nx, double sigma, int maxiters, double *__restrict__ out)
{
    extern __shared__ float vals[];
    __shared__ unsigned hist[kBoxBins], hist2[kBoxBins];
    __shared__ double scratch[NT / kWave];
    __shared__ unsigned s_below[NT / kWave];
    __shared__ DigitPick s_pick;
    BoxView b;
    b.data = data; b.mask = mask; b.H = H; b.W = W; b.bh = bh; b.bw = bw;
    const int box = blockIdx.x;
    b.r0 = (box / nx) * bh;
    b.c0 = (box % nx) * bw;
    const int npix = bh * bw;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if constexpr (RES) {
        for (int r = wave; r < bh; r += NT / kWave)
            for (int c = lane; c < bw; c += kWave) vals[r * bw + c] = box_value(b, r * bw + c);
        __syncthreads();
    }
    // f(x) for every candidate value (RES: NaN marks the excluded ones and fails every comparison)
    auto for_each = [&](auto f) {
        if constexpr (RES) {
            for (int e = threadIdx.x; e < npix; e += NT) f(vals[e]);
        } else {
            // two rows x four 64-column chunks per trip: 16 independent loads in flight per lane (one load at a time leaves
            // the pass bound by the Infinity Cache latency)
            constexpr int RU = 2, CU = 4;
            const int cend = min(bw, W - b.c0);
            const int rend = min(bh, H - b.r0);
            for (int r = wave * RU; r < rend; r += (NT / kWave) * RU) {
                for (int c0 = 0; c0 < cend; c0 += CU * kWave) {
                    float x[RU][CU];
                    uint8_t m[RU][CU];
#pragma unroll
                    for (int u = 0; u < RU; u++) {
                        const int64_t rowp = (int64_t)(b.r0 + r + u) * W + b.c0;
#pragma unroll
                        for (int q = 0; q < CU; q++) {
                            const int c = c0 + q * kWave + lane;
                            const bool in = (r + u) < rend && c < cend;
                            m[u][q] = in ? (mask ? mask[rowp + c] : (uint8_t)0) : (uint8_t)1;
                            x[u][q] = in ? data[rowp + c] : 0.0f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RU; u++)
#pragma unroll
                        for (int q = 0; q < CU; q++)
                            if (!m[u][q]) f(x[u][q]);
                }
            }
        }
    };
    auto zero_hist = [&]() {
        for (int t = threadIdx.x; t < kBoxBins; t += NT) hist[t] = 0;
        __syncthreads();
    };
    // Working set of the clipping passes: values inside the running intersection [lo_run, hi_run] of all bounds so far
    // (astropy packs its buffer the same way).  The FINAL survivors are the values inside the LAST computed bounds
    // [lo_last, hi_last] applied to all data (astropy sigma_clipping.py:356-358: a value clipped by an earlier, tighter
    // pass can come back).  Bounds are float64 in astropy and applied to float32 data: lo <= (double)x <= hi is the
    // same set as ceil32(lo) <= x <= floor32(hi).  Non-finite values fail lo <= x <= hi from the first pass on
    // (+-inf only while the bounds are still infinite: excluded explicitly).
    float lo_run = -3.4028234663852886e38f, hi_run = 3.4028234663852886e38f, lo_last = lo_run, hi_last = hi_run;
    double med = __builtin_nan(""), sd = __builtin_nan("");
    int n = 0, n_prev = -1, n_unmasked = -1;
    unsigned guess0 = 0;                                    // first-level digit of the previous iteration's median
    for (int pass = 0;; pass++) {
        const bool final_pass = pass >= maxiters || n_prev == -2;
        const float lo = final_pass ? lo_last : lo_run, hi = final_pass ? hi_last : hi_run;
        unsigned prefix, kk;
        if (pass == 0 || RES) {
            // first iteration (and every iteration of an LDS-resident box, which is bound by its LDS atomics and gains nothing
            // from fewer reads): count + sum, then spread + first select level, then the second level (3 passes)
            double cnt = 0.0, sum = 0.0;
            for_each([&](float x) {
                if (x >= lo && x <= hi) {
                    cnt += 1.0;
                    sum += (double)x;
                }
            });
            n = (int)block_sum<NT>(cnt, scratch);
            const double total = block_sum<NT>(sum, scratch);
            if (n_unmasked < 0) n_unmasked = n;
            if (n == 0) {
                med = sd = __builtin_nan("");
                break;
            }
            if (!final_pass && n == n_prev) {               // the last bounds removed nothing: converged, evaluate the survivors
                n_prev = -2;
                continue;
            }
            const double mean = total / (double)n;
            const unsigned k = (unsigned)(n >> 1);          // upper middle element (the median itself for odd n)
            // spread + leading 11 key bits.  The sky values of a box share their leading digits: run-length coded per
            // thread before they reach the LDS histogram.
            zero_hist();
            double ss = 0.0;
            unsigned cur_d = kNoDigit, cur_n = 0;
            for_each([&](float x) {
                if (x >= lo && x <= hi) {
                    const double d = mean - (double)x;
                    ss += d * d;
                    const unsigned dg = f32_key(x) >> (32 - kBoxDigit);
                    if (dg != cur_d) {
                        if (cur_d != kNoDigit) atomicAdd(&hist[cur_d], cur_n);
                        cur_d = dg;
                        cur_n = 0;
                    }
                    cur_n++;
                }
            });
            if (cur_d != kNoDigit) atomicAdd(&hist[cur_d], cur_n);
            sd = sqrt(block_sum<NT>(ss, scratch) / (double)n);   // (its barriers also close the histogram)
            DigitPick pk = pick_digit(hist, k, &s_pick);
            prefix = (unsigned)pk.digit;
            kk = k - pk.cum;
            guess0 = prefix;
            zero_hist();
            for_each([&](float x) {
                if (x >= lo && x <= hi) {
                    const unsigned key = f32_key(x);
                    if ((key >> 21) == prefix) atomicAdd(&hist[(key >> 10) & (kBoxBins - 1)], 1u);
                }
            });
            __syncthreads();
            pk = pick_digit(hist, kk, &s_pick);
            prefix = (prefix << kBoxDigit) | (unsigned)pk.digit;
            kk -= pk.cum;
        } else {
            // later iterations: ONE pass for count, moments about the previous median (S1 = sum(x - p), S2 = sum((x - p)^2):
            // mean = p + S1 / n, n var = S2 - S1^2 / n - p sits inside the survivors, so nothing cancels), the first select
            // level, and the second level speculated on the first level's digit of the previous iteration (the median hardly
            // moves); a wrong guess costs the separate second-level pass.
            const double p = med;
            zero_hist();
            const unsigned g0 = guess0;
            for (int t = threadIdx.x; t < kBoxBins; t += NT) hist2[t] = 0;
            __syncthreads();
            double cnt = 0.0, s1 = 0.0, s2 = 0.0;
            unsigned cur_d = kNoDigit, cur_n = 0;
            for_each([&](float x) {
                if (x >= lo && x <= hi) {
                    cnt += 1.0;
                    const double d = (double)x - p;
                    s1 += d;
                    s2 = fma(d, d, s2);
                    const unsigned key = f32_key(x);
                    const unsigned dg = key >> (32 - kBoxDigit);
                    if (dg != cur_d) {
                        if (cur_d != kNoDigit) atomicAdd(&hist[cur_d], cur_n);
                        cur_d = dg;
                        cur_n = 0;
                    }
                    cur_n++;
                    if (dg == g0) atomicAdd(&hist2[(key >> 10) & (kBoxBins - 1)], 1u);
                }
            });
            if (cur_d != kNoDigit) atomicAdd(&hist[cur_d], cur_n);
            n = (int)block_sum<NT>(cnt, scratch);
            const double S1 = block_sum<NT>(s1, scratch), S2 = block_sum<NT>(s2, scratch);
            if (n == 0) {
                med = sd = __builtin_nan("");
                break;
            }
            if (!final_pass && n == n_prev) {               // the last bounds removed nothing: converged, evaluate the survivors
                n_prev = -2;
                continue;
            }
            const double nv = S2 - S1 * S1 / (double)n;
            sd = sqrt((nv > 0.0 ? nv : 0.0) / (double)n);
            const unsigned k = (unsigned)(n >> 1);
            DigitPick pk = pick_digit(hist, k, &s_pick);
            prefix = (unsigned)pk.digit;
            kk = k - pk.cum;
            if (prefix == g0) {
                pk = pick_digit(hist2, kk, &s_pick);
            } else {
                guess0 = prefix;
                zero_hist();
                for_each([&](float x) {
                    if (x >= lo && x <= hi) {
                        const unsigned key = f32_key(x);
                        if ((key >> 21) == prefix) atomicAdd(&hist[(key >> 10) & (kBoxBins - 1)], 1u);
                    }
                });
                __syncthreads();
                pk = pick_digit(hist, kk, &s_pick);
            }
            prefix = (prefix << kBoxDigit) | (unsigned)pk.digit;
            kk -= pk.cum;
        }
        // (d) last 10 bits + the largest survivor key below the 22-bit prefix
        zero_hist();
        unsigned below = 0;
        for_each([&](float x) {
            if (x >= lo && x <= hi) {
                const unsigned key = f32_key(x);
                const unsigned top = key >> 10;
                if (top == prefix) atomicAdd(&hist[key & 1023u], 1u);
                else if (top < prefix) below = key > below ? key : below;
            }
        });
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            const unsigned o = __shfl_down(below, d);
            below = o > below ? o : below;
        }
        if (lane == 0) s_below[wave] = below;
        __syncthreads();
        const DigitPick pk = pick_digit(hist, kk, &s_pick);
        const unsigned key2 = (prefix << 10) | (unsigned)pk.digit;
        unsigned key1 = key2;
        if ((n & 1) == 0 && kk - pk.cum == 0) {             // rank k is the first of its key: the lower neighbour is another key
            if (pk.dlow >= 0) key1 = (prefix << 10) | (unsigned)pk.dlow;
            else {
                unsigned mx = 0;
                for (int w = 0; w < NT / kWave; w++) mx = s_below[w] > mx ? s_below[w] : mx;
                key1 = mx;
            }
        }
        med = ((double)key_f32(key1) + (double)key_f32(key2)) / 2.0;
        if (final_pass) break;
        const double lo64 = med - sigma * sd, hi64 = med + sigma * sd;
        float lof = (float)lo64, hif = (float)hi64;
        if ((double)lof < lo64) lof = nextafterf(lof, __builtin_inff());
        if ((double)hif > hi64) hif = nextafterf(hif, -__builtin_inff());
        lo_last = lof;
        hi_last = hif;
        lo_run = fmaxf(lo_run, lof);
        hi_run = fminf(hi_run, hif);
        n_prev = n;
    }
    if (threadIdx.x == 0) {
        out[4 * box + 0] = med;
        out[4 * box + 1] = sd;
        out[4 * box + 2] = (double)n;
        out[4 * box + 3] = (double)(npix - (n_unmasked < 0 ? 0 : n_unmasked));
    }
}

// ------------------------------------------------------------------------------------------------
// scipy.ndimage.zoom(mesh, (zy, zx), order=3, mode='reflect', grid_mode=True) evaluated at every output pixel from the
// prefiltered cubic B-spline coefficients coef[ny][nx] (float64; the prefilter of the ny x nx mesh is host logic):
// input coordinate y = (i + 0.5) * ny / Hz - 0.5 (Hz = the zoomed height ny * zy), folded into [-0.5, ny - 0.5] by
// half-sample reflection; taps floor(y) - 1 .. + 2, tap indices folded the same way.  Output clipped to [vmin, vmax]
// (BkgZoomInterpolator(clip=True)) and cropped to H x W.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

__device__ __forceinline__ void bspline3(double t, double (&w)[4])
{
    const double t2 = t * t, t3 = t2 * t, u = 1.0 - t;
    w[0] = u * u * u / 6.0;
    w[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
    w[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
    w[3] = t3 / 6.0;
}

__global__ __launch_bounds__(kBlock) void spline_zoom_kernel(const double *__restrict__ coef, int ny, int nx, int zy, int zx, int H, int W,
                                                            double vmin, double vmax, double *__restrict__ out)
{
    const int64_t P = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const int i = (int)(p / W), j = (int)(p - (int64_t)i * W);
        const double y = ((double)i + 0.5) / (double)zy - 0.5, x = ((double)j + 0.5) / (double)zx - 0.5;
        const double fy = floor(y), fx = floor(x);
        double wy[4], wx[4];
        bspline3(y - fy, wy);
        bspline3(x - fx, wx);
        const int iy = (int)fy - 1, ix = (int)fx - 1;
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double *row = coef + (int64_t)reflect_idx(iy + a, ny) * nx;
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) r += wx[q] * row[reflect_idx(ix + q, nx)];
            acc += wy[a] * r;
        }
        out[p] = fmin(fmax(acc, vmin), vmax);
    }
}

}  // namespace

extern "C" size_t apgpu_source_mask_ws_bytes(int64_t height, int64_t width)
{
    if (height <= 0 || width <= 0) return 0;
    return (size_t)(height * width) * (2 * sizeof(int32_t) + 2) + 64;
}

extern "C" int apgpu_source_mask_u8(const uint8_t *above, int64_t height, int64_t width, int32_t min_pixels, int32_t dilate_size,
                                    uint8_t *mask_out, int64_t *nsources_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!above || !mask_out || !ws) return fail(APGPU_EINVAL, "source_mask: NULL pointer argument");
    if (height <= 0 || width <= 0 || height * width > 0x7fffffffLL) return fail(APGPU_EINVAL, "source_mask: bad shape");
    if (min_pixels < 1) return fail(APGPU_EINVAL, "source_mask: min_pixels must be >= 1");
    if (dilate_size < 1 || (dilate_size & 1) == 0) return fail(APGPU_EINVAL, "source_mask: dilate_size must be odd and >= 1");
    if (ws_bytes < apgpu_source_mask_ws_bytes(height, width)) return fail(APGPU_EWORKSPACE, "source_mask: workspace too small");
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail(APGPU_EINVAL, "source_mask: workspace must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int64_t P = height * width;
    int *L = static_cast<int *>(ws);
    int *size = L + P;
    uint8_t *tmp1 = reinterpret_cast<uint8_t *>(size + P);
    uint8_t *tmp2 = tmp1 + P;
    const unsigned g = grid1d(P);
    if (nsources_out && hipMemsetAsync(nsources_out, 0, sizeof(int64_t), st) != hipSuccess) return fail(APGPU_ELAUNCH, "source_mask: memset failed");
    hipLaunchKernelGGL(ccl_init_kernel, dim3(g), dim3(kBlock), 0, st, above, L, size, P);
    hipLaunchKernelGGL(ccl_union_kernel, dim3(g), dim3(kBlock), 0, st, above, L, (int)height, (int)width);
    hipLaunchKernelGGL(ccl_count_kernel, dim3(g), dim3(kBlock), 0, st, L, size, P);
    hipLaunchKernelGGL(ccl_filter_kernel, dim3(g), dim3(kBlock), 0, st, L, size, min_pixels, tmp1,
                       reinterpret_cast<unsigned long long *>(nsources_out), P);
    if (int rc = check_launch("source_mask (components)")) return rc;
    const int half = dilate_size / 2;
    hipLaunchKernelGGL(dilate_kernel<true>, dim3(g), dim3(kBlock), 0, st, tmp1, tmp2, (int)height, (int)width, half);
    hipLaunchKernelGGL(dilate_kernel<false>, dim3(g), dim3(kBlock), 0, st, tmp2, mask_out, (int)height, (int)width, half);
    return check_launch("source_mask (dilation)");
}

extern "C" int apgpu_box_clipped_stats_f32(const float *data, const uint8_t *mask, int64_t height, int64_t width, int32_t box_height,
                                           int32_t box_width, double sigma, int32_t maxiters, double *stats_out, void *stream)
{
    if (!data || !stats_out) return fail(APGPU_EINVAL, "box_clipped_stats: NULL pointer argument");
    if (height <= 0 || width <= 0 || height > 0x7fffffff || width > 0x7fffffff) return fail(APGPU_EINVAL, "box_clipped_stats: bad shape");
    if (box_height < 1 || box_width < 1 || (int64_t)box_height * box_width > (1 << 24))
        return fail(APGPU_EINVAL, "box_clipped_stats: bad box size %d x %d", box_height, box_width);
    if (!(sigma >= 0.0) || maxiters < 0) return fail(APGPU_EINVAL, "box_clipped_stats: bad clip parameters");
    const int ny = (int)((height + box_height - 1) / box_height), nx = (int)((width + box_width - 1) / box_width);
    const int64_t npix = (int64_t)box_height * box_width;
    const dim3 grid((unsigned)(ny * nx));
    hipStream_t st = as_stream(stream);
    const int h = (int)height, w = (int)width;
    if (npix <= 8192) {
        hipLaunchKernelGGL((box_stats_kernel<256, true>), grid, dim3(256), (size_t)npix * sizeof(float), st, data, mask, h, w, box_height,
                           box_width, nx, sigma, maxiters, stats_out);
    } else if (npix <= kBoxLdsPixels) {
        static bool raised = false;                         // above 64 KB of dynamic LDS needs the attribute once per process
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(box_stats_kernel<1024, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kBoxLdsPixels * (int)sizeof(float)) != hipSuccess)
                return fail(APGPU_ELAUNCH, "box_clipped_stats: cannot raise the dynamic LDS limit");
            raised = true;
        }
        hipLaunchKernelGGL((box_stats_kernel<1024, true>), grid, dim3(1024), (size_t)npix * sizeof(float), st, data, mask, h, w, box_height,
                           box_width, nx, sigma, maxiters, stats_out);
    } else {
        hipLaunchKernelGGL((box_stats_kernel<1024, false>), grid, dim3(1024), 0, st, data, mask, h, w, box_height, box_width, nx, sigma,
                           maxiters, stats_out);
    }
    return check_launch("box_clipped_stats");
}

extern "C" int apgpu_spline_zoom_f64(const double *coef, int32_t ny, int32_t nx, int32_t zoom_y, int32_t zoom_x, int64_t height,
                                     int64_t width, double vmin, double vmax, double *out, void *stream)
{
    if (!coef || !out) return fail(APGPU_EINVAL, "spline_zoom: NULL pointer argument");
    if (ny < 1 || nx < 1 || zoom_y < 1 || zoom_x < 1) return fail(APGPU_EINVAL, "spline_zoom: bad mesh / zoom");
    if (height <= 0 || width <= 0 || height > (int64_t)ny * zoom_y || width > (int64_t)nx * zoom_x)
        return fail(APGPU_EINVAL, "spline_zoom: output %lld x %lld exceeds the zoomed mesh", (long long)height, (long long)width);
    hipLaunchKernelGGL(spline_zoom_kernel, dim3(grid1d(height * width)), dim3(kBlock), 0, as_stream(stream), coef, ny, nx, zoom_y, zoom_x,
                       (int)height, (int)width, vmin, vmax, out);
    return check_launch("spline_zoom");
}
