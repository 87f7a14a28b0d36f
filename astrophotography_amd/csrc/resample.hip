// resample.hip - F3: affine Lanczos-3 resample of registered frames on gfx950 (the step the reference
// delegates to SWarp: scripts/resample_all.sh:123-131 RESAMPLING_TYPE LANCZOS3, :298 FSCALE = 1/EXPTIME,
// :112 / :339 OVERSAMPLING 4, :330-342 the swarp call).  There is no in-tree arithmetic to match; the definition is this
// build's own and is restated on the CPU in oracle/apref.c (apref_resample_affine_f32, apref_resample_oversampled_f32) -
// the two agree bit for bit.
//
//   F[k] = llrint(A[k] * 2^32); xin = F0 x + F1 y + F2, yin = F3 x + F4 y + F5   (32.32 fixed point, output -> input: exactly
//   reproducible, a 64-bit add per row)
//   ix = xin >> 32, phase px = top log2(n_phases) bits of the fraction, rounded; taps ix-2 .. ix+3 weighted by lut[px][0..5]
//   (a host-built table of normalised Lanczos-3 weights), rows combined by lut[py]; even / odd fmaf chains in a
//   fixed order (= packed float32 arithmetic); result * fscale[f].  Any tap outside the frame, masked or non-finite -> NaN, weight 0.
//   OVERSAMPLING n: the transform is that of the n-times finer grid, every output pixel the float64 mean of its n x n
//   sub-samples in row-major order (= apgpu_block_mean_f32 of the fine resample, without the fine image ever existing).
//
// Two launches (four with a bad-pixel mask, see "mask by scatter" below).  resample_tiles_kernel works out, once per
// workgroup tile - 64 x 16 output pixels, or 64 x 32 where a frame has ONE transform -, what every lane of that tile's workgroup
// needs: the fixed-point coefficients, the flux scale, the input footprint (exact: the integer corner coordinates) and which
// path the tile takes - 64 bytes per tile that the main kernel reads with ONE scalar load (rounds 1-2 did the float64 corner
// arithmetic in every lane, round 3 first in thread 0 behind an LDS broadcast and a barrier).
// The main kernel stages the footprint in LDS and evaluates the taps from there:
//   FAST tiles (the window of every pixel inside the frame, footprint at most 80 x (tile rows + 10): registration-sized
//   rotations up to ~2.5 degrees, frames below 2^30 pixels): fixed LDS pitch of 80 floats and TWO copies of the footprint, the second shifted
//   by one float, so that every lane reads its six taps of a row as three ALIGNED ds_read_b64 from the copy that matches the
//   parity of its first column - 256 B/clk where ds_read2_b32 gets 128 (MI355X guide, LDS table) - with all 18 reads of a
//   window off one address register (row j at the immediate offset 320 j).  The second copy starts 32 banks after the first
//   (offset = 32 mod 64 dwords): an even-start lane and its odd-start neighbour read the same dword offsets of different
//   copies, and land on disjoint banks.  Footprint rows are fetched through a bounds-checked buffer resource (no clamps),
//   table rows and output stores go through buffer instructions with 32-bit offsets.
//   Other tiles (frame border, strong shear / minification, giant frames): the general path - footprint at its own pitch up
//   to 16 KB with validity applied at the fill, or a direct gather from global memory - same arithmetic.
// HBM traffic is one read of the input (+ halo, mostly L2 hits) and one write of the output: 8 B per pixel.
#include "resample_core.h"

namespace {
// (Round 5, measured and dropped: PERSISTENT workgroups - a grid of what the chip holds at once, 5 x 256 workgroups, each walking
// tiles w, w + G, .. in the same XCD-aware order, the next tile's record touched ahead so that its load hits the scalar cache.
// 4.27 ms against 3.83 ms for one workgroup per tile on the same box (16 x 8192^2, +-0.2 degrees; 4.57 ms before the workgroups
// per XCD were made an odd number: with 160 a workgroup met the same few tile columns again and again - 160 k mod 128 - and the
// ones that drew the frame's edge columns, which take the general path, finished long after the rest).  The dispatcher's 0.6 ms
// for 524,288 workgroups is not on the critical path - it runs ahead of the workgroups - and its dynamic assignment balances
// the slow edge tiles, which a static walk cannot; the loop also cost 20 - 40 spilled SGPRs.  profiles/r05_c5/ab_resample.txt.)
#ifndef APGPU_RESAMPLE_WEIGHTS_BESIDE_FILL
#define APGPU_RESAMPLE_WEIGHTS_BESIDE_FILL 1
#endif
#ifndef APGPU_RESAMPLE_AHEAD_NONSTEADY
#define APGPU_RESAMPLE_AHEAD_NONSTEADY 0                     // pixels whose weight rows the non-steady path fetches ahead (0: in the trip that uses them)
#endif
template <bool HAS_MASK, bool OVERSAMPLED, int TH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void resample_affine_kernel(const float *__restrict__ frames, const uint8_t *__restrict__ mask,
                                                             const TileRec *__restrict__ recs, int ntiles, int gx, int gy,
                                                             const float *__restrict__ lut,
                                                             int log2_phases, int os, float *__restrict__ out, uint8_t *__restrict__ wout,
                                                             int h_in, int w_in, int h_out, int w_out, const int *__restrict__ mask_ctl, int mask_cap)
{
    constexpr bool kRolling = !OVERSAMPLED && APGPU_RESAMPLE_ROLLING;
    using G = FastGeom<TH>;
    __shared__ __attribute__((aligned(16))) float tile[G::kLdsFloats];
    // Tile order: grid = (tiles per frame rounded up to 8, frames).  Workgroups are dispatched round-robin over the 8 XCDs
    // (x fastest, and gridDim.x is a multiple of 8: workgroup x runs on XCD x % 8); a frame's tiles are cut into 8 contiguous
    // ranges, one per XCD, and consecutive workgroups of an XCD take consecutive tiles of its range, so that the halo a tile
    // shares with its neighbours is a hit in that XCD's L2 rather than a second fetch (one workgroup per tile in launch order
    // left 39 % hits and cost 7 %; with this order FETCH_SIZE is 1.03 x the input).
    const int per_frame = gx * gy;
#ifdef APGPU_VARIANT_RESAMPLE_LINEAR
    const int rem = blockIdx.x;
#else
    const int chunk = (per_frame + 7) >> 3;
    const int rem = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
#endif
    if (rem >= per_frame) return;
    const int64_t f = blockIdx.y;
    const int t = (int)f * per_frame + rem;
    const int tyi = rem / gx, txi = rem - tyi * gx;
    const int x0 = txi * kTileW, y0 = tyi * TH;
    FrameView fv;
    fv.src = frames + f * (int64_t)h_in * w_in;
    fv.mask = nullptr;
    fv.h_in = h_in;
    fv.w_in = w_in;
    const TileRec *rp = recs + t;                                 // uniform address: scalar loads
    TileCtx tc;
#pragma unroll
    for (int k = 0; k < 6; k++) tc.F[k] = rp->F[k];
    tc.bx0 = rp->bx0;
    tc.by0 = rp->by0;
    const unsigned dims = rp->dims;
    tc.fw = (int)(dims & 0xfffu);
    tc.fh = (int)((dims >> 12) & 0xfffu);
    const unsigned flags = dims >> 24;
    tc.staged = (flags & kStaged) != 0;
    tc.sane_top = (flags & kSaneTop) != 0;
    tc.sane_bot = (flags & kSaneBot) != 0;
    tc.fs = rp->fs;
    const bool fast = (flags & kFast) != 0, interior = (flags & kInterior) != 0;
    const int tid = threadIdx.x;

    // the mask is applied here (footprint fill / gather) when the tile says so or the bad-pixel list overflowed; otherwise
    // mask_scatter_kernel poisons the affected output pixels afterwards (wave-uniform choice between whole code paths: as a
    // test around each mask load it serialised the loads)
    bool inline_mask = false;
    if constexpr (HAS_MASK) inline_mask = (flags & kInlineMask) != 0 || mask_ctl[0] > mask_cap;
    if (inline_mask) fv.mask = mask;
    // lane -> output column x0 + lx and the rows y0 + ly, + 4, + 8, + 12 (rolling fast path: the rows y0 + R ly .. + R - 1)
    const int lx = tid % kTileW, ly = tid / kTileW;
    const int sh = 32 - log2_phases;
    // steady: over a lane's TH / 4 consecutive rows the y phase moves by less than one table row (F4 within 1 / (rows x phases) of
    // an integer: rotations up to ~0.9 degrees at unit scale, scale errors up to 1.4e-4) - the lane keeps two y rows; otherwise
    // both table rows are fetched per pixel, in the trip that uses them (fetching them ahead as well would put this path at 94
    // VGPRs and the whole kernel at five wavefronts per SIMD instead of six; it bought 2 % when it was measured)
    Rolling<TH, APGPU_RESAMPLE_AHEAD> ro;
    bool steady = false;
    if constexpr (kRolling) {
        const unsigned fr4 = (unsigned)tc.F[4];
        const unsigned dist = fr4 < 0x80000000u ? fr4 : 0u - fr4;
#if APGPU_RESAMPLE_KEEP_WY
        steady = (unsigned long long)dist * (TH / 4 - 1) < (1ull << sh);
#endif
    }
    // The first pixels' table rows are fetched BESIDE the footprint (round 6): issued between the footprint's loads and its LDS
    // stores.  Until then they were issued after the stores - i.e. after a vmcnt(0) - and a workgroup paid the two memory round
    // trips one after the other in front of its barrier (the ISA said so; the source order had promised the overlap since round 5).
    auto begin_weights = [&]() {
        if constexpr (kRolling) {
            const v4i lrsrc0 = make_rsrc(lut, (unsigned)((1 << log2_phases) + 1) * 24u);
            if (steady) rolling_begin<TH, true, APGPU_RESAMPLE_AHEAD, APGPU_RESAMPLE_AHEAD>(ro, tc, lrsrc0, sh, x0, y0, lx, ly);
            else rolling_begin<TH, false, APGPU_RESAMPLE_AHEAD_NONSTEADY, APGPU_RESAMPLE_AHEAD>(ro, tc, lrsrc0, sh, x0, y0, lx, ly);
        }
    };
    if (fast) {
        if (inline_mask) {
            FastFill<true, G::kTrips> ff;
            fast_fill_issue<true, G::kTrips>(ff, fv.src, mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
#if APGPU_RESAMPLE_WEIGHTS_BESIDE_FILL
            begin_weights();
#endif
            fast_fill_store<true, G::kTrips, G::kOffB>(ff, tc.fh, tile, tid);
        } else {
            FastFill<false, G::kTrips> ff;
            fast_fill_issue<false, G::kTrips>(ff, fv.src, mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
#if APGPU_RESAMPLE_WEIGHTS_BESIDE_FILL
            begin_weights();
#endif
            fast_fill_store<false, G::kTrips, G::kOffB>(ff, tc.fh, tile, tid);
        }
#if !APGPU_RESAMPLE_WEIGHTS_BESIDE_FILL
        begin_weights();
#endif
    } else if (tc.staged) {
        if (inline_mask) general_fill<true>(tc, fv, tile, tid);
        else general_fill<false>(tc, fv, tile, tid);
    }
    __syncthreads();

    const int x = x0 + lx;
    if (x >= w_out) return;
    const int yb0 = y0 + ly;
    if (fast) {
        // stores through a buffer resource based at the tile's first pixel: 32-bit offsets (w_out < 2^26, checked by the launcher)
        const int64_t t0 = (f * h_out + y0) * (int64_t)w_out + x0;
        const v4i orsrc = make_rsrc(out + t0, 0xffffffffu);
        const v4i wrsrc = make_rsrc(wout + t0, 0xffffffffu);            // (not used when wout is NULL)
        const v4i lrsrc = make_rsrc(lut, (unsigned)((1 << log2_phases) + 1) * 24u);
        if constexpr (kRolling) {
            if (steady) pixels_fast_rolling<TH, true, APGPU_RESAMPLE_AHEAD, APGPU_RESAMPLE_AHEAD>(ro, tc, tile, lrsrc, sh, lx, ly, orsrc, wrsrc, wout != nullptr, w_out);
            else pixels_fast_rolling<TH, false, APGPU_RESAMPLE_AHEAD_NONSTEADY, APGPU_RESAMPLE_AHEAD>(ro, tc, tile, lrsrc, sh, lx, ly, orsrc, wrsrc, wout != nullptr, w_out);
        } else {
            const int ooff = (ly * w_out + lx) * 4, ostep = 16 * w_out;
            pixels_fast<OVERSAMPLED, TH, 1>(tc, tile, lrsrc, sh, os, x0, y0, lx, ly, orsrc, wrsrc, wout != nullptr, ooff, ostep);
        }
        return;
    }
    const int64_t o0 = (f * h_out + yb0) * (int64_t)w_out + x;
    float *op = out + o0;
    uint8_t *wp = wout ? wout + o0 : nullptr;
    const int64_t row_stride = 4 * (int64_t)w_out;
    if (interior) pixels_general<true, OVERSAMPLED, TH>(tc, fv, tile, lut, sh, os, x, yb0, h_out, row_stride, op, wp);
    else pixels_general<false, OVERSAMPLED, TH>(tc, fv, tile, lut, sh, os, x, yb0, h_out, row_stride, op, wp);
}

int launch_resample(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask, const double *affines,
                    int32_t affines_per_tile, int32_t conserve_flux, const float *fscale, const float *lut, int32_t n_phases,
                    int32_t os, float *out, uint8_t *weight_out, int64_t h_out, int64_t w_out, void *stream, const char *who)
{
    if (!frames || !affines || !lut || !out) return fail(APGPU_EINVAL, "%s: NULL pointer argument", who);
    if (n_frames <= 0 || n_frames > 65535) return fail(APGPU_EINVAL, "%s: n_frames = %d (1..65535)", who, n_frames);
    if (h_in < 6 || w_in < 6 || h_out <= 0 || w_out <= 0) return fail(APGPU_EINVAL, "%s: bad shape", who);
    if (os < 1 || os > 16) return fail(APGPU_EINVAL, "%s: oversampling %d (1..16)", who, os);
    // (the fine grid's coordinates stay below 2^30 as well)
    if (h_in > 0x3fffffff || w_in > 0x3fffffff || h_out * os > 0x3fffffff || w_out * os > 0x3fffffff)
        return fail(APGPU_EUNSUPPORTED, "%s: image sides are limited to 2^30 (fine) pixels", who);
    if (n_phases < 2 || n_phases > (1 << 20) || (n_phases & (n_phases - 1)))
        return fail(APGPU_EINVAL, "%s: n_phases = %d (a power of two, 2 .. 2^20: the phase is the top bits of a 32-bit fraction)", who, n_phases);
    int log2_phases = 0;
    while ((1 << log2_phases) < n_phases) log2_phases++;
    if (reinterpret_cast<uintptr_t>(lut) & 7) return fail(APGPU_EINVAL, "%s: lut must be 8-byte aligned", who);
    // 32 output rows per workgroup where a frame has one transform, else the 16 of the API tile
#ifdef APGPU_VARIANT_RESAMPLE_TH16
    const int th = kTileH;
#else
    const int th = (!affines_per_tile && h_out > kTileH) ? 2 * kTileH : kTileH;
#endif
    const int64_t gx = (w_out + kTileW - 1) / kTileW, gy = (h_out + th - 1) / th;
    hipStream_t st = as_stream(stream);
    // the per-tile records: 64 bytes per workgroup tile, stream-ordered scratch
    const int64_t ntiles = (int64_t)n_frames * gx * gy;
    if (ntiles > 0x7ffffff0LL) return fail(APGPU_EUNSUPPORTED, "%s: too many tiles (%lld)", who, (long long)ntiles);
    TileRec *recs = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&recs), (size_t)ntiles * sizeof(TileRec), st);
    if (e != hipSuccess) return fail(APGPU_ELAUNCH, "%s: cannot allocate %lld tile records: %s", who, (long long)ntiles, hipGetErrorString(e));
    // the fast path addresses a frame and a tile's output rows with 32-bit byte offsets
    const int fast_ok = (h_in * w_in < (1LL << 30)) && (w_out < (1LL << 25));
    // bad-pixel mask: list + counter for the scatter form (see mask_list_kernel); pixel indices fit an int then
#ifdef APGPU_VARIANT_RESAMPLE_MASK_INLINE
    const int mask_scatter = 0;
#else
    const int mask_scatter = mask && !affines_per_tile && h_in * w_in < (1LL << 31);
#endif
    int *mctl = nullptr;
    int64_t mcap64 = h_in * w_in / 64;
    if (mcap64 < 256) mcap64 = 256;
    const int kMaskListCap = (int)(mcap64 > kMaskListCapMax ? kMaskListCapMax : mcap64);
    if (mask) {
        e = hipMallocAsync(reinterpret_cast<void **>(&mctl), (size_t)(2 + (mask_scatter ? kMaskListCap : 0)) * sizeof(int), st);
        if (e == hipSuccess) e = hipMemsetAsync(mctl, 0, 2 * sizeof(int), st);
        if (e != hipSuccess) {
            (void)hipFreeAsync(recs, st);
            return fail(APGPU_ELAUNCH, "%s: cannot allocate the mask list: %s", who, hipGetErrorString(e));
        }
        if (mask_scatter) {
            int64_t g = (h_in * w_in / 16 / 16 + 255) / 256;    // ~16 loads of 16 bytes per lane: few reservations, enough wavefronts
            if (g < 1) g = 1;
            if (g > kNumCU * 32) g = kNumCU * 32;
            hipLaunchKernelGGL(mask_list_kernel, dim3((unsigned)g), dim3(256), 0, st, mask, h_in * w_in, kMaskListCap, mctl, mctl + 2);
        }
    }
    const int64_t tb = (ntiles + 255) / 256;
    hipLaunchKernelGGL(resample_tiles_kernel, dim3((unsigned)tb), dim3(256), 0, st, affines, affines_per_tile, conserve_flux, fscale, (int)os, th,
                       (int)gx, (int)gy, ntiles, (int)h_in, (int)w_in, (int)h_out, (int)w_out, fast_ok, mask_scatter, recs);
    int rc = check_launch(who);
    if (rc == APGPU_OK) {
        const dim3 grid((unsigned)(((gx * gy + 7) / 8) * 8), (unsigned)n_frames);
#define APGPU_RESAMPLE_LAUNCH(M, O)                                                                                                            \
    do {                                                                                                                                       \
        if (th == kTileH)                                                                                                                      \
            hipLaunchKernelGGL((resample_affine_kernel<M, O, kTileH>), grid, dim3(256), 0, st, frames, mask, recs, (int)ntiles, (int)gx, (int)gy, \
                               lut, log2_phases, (int)os, out, weight_out, (int)h_in, (int)w_in, (int)h_out, (int)w_out, mctl, kMaskListCap);         \
        else                                                                                                                                   \
            hipLaunchKernelGGL((resample_affine_kernel<M, O, 2 * kTileH>), grid, dim3(256), 0, st, frames, mask, recs, (int)ntiles, (int)gx,    \
                               (int)gy, lut, log2_phases, (int)os, out, weight_out, (int)h_in, (int)w_in, (int)h_out, (int)w_out, mctl, kMaskListCap); \
    } while (0)
        if (mask) {
            if (os > 1) APGPU_RESAMPLE_LAUNCH(true, true);
            else APGPU_RESAMPLE_LAUNCH(true, false);
        } else {
            if (os > 1) APGPU_RESAMPLE_LAUNCH(false, true);
            else APGPU_RESAMPLE_LAUNCH(false, false);
        }
#undef APGPU_RESAMPLE_LAUNCH
        rc = check_launch(who);
        if (rc == APGPU_OK && mask_scatter) {
            hipLaunchKernelGGL(mask_scatter_kernel, dim3(64, (unsigned)n_frames), dim3(256), 0, st, mctl, mctl + 2, kMaskListCap, affines, (int)os,
                               (int)w_in, out, weight_out, (int)h_out, (int)w_out);
            rc = check_launch(who);
        }
    }
    (void)hipFreeAsync(recs, st);
    if (mctl) (void)hipFreeAsync(mctl, st);
    return rc;
}

}  // namespace

extern "C" int apgpu_resample_affine_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                         const double *affines, int32_t affines_per_tile, int32_t conserve_flux,
                                         const float *fscale, const float *lut, int32_t n_phases, float *out,
                                         uint8_t *weight_out, int64_t h_out, int64_t w_out, void *stream)
{
    return launch_resample(frames, n_frames, h_in, w_in, mask, affines, affines_per_tile, conserve_flux, fscale, lut, n_phases, 1, out,
                           weight_out, h_out, w_out, stream, "resample_affine");
}

extern "C" int apgpu_resample_oversampled_f32(const float *frames, int32_t n_frames, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                              const double *fine_affines, int32_t affines_per_tile, int32_t conserve_flux,
                                              const float *fscale, const float *lut, int32_t n_phases, int32_t oversampling, float *out,
                                              uint8_t *weight_out, int64_t h_out, int64_t w_out, void *stream)
{
    return launch_resample(frames, n_frames, h_in, w_in, mask, fine_affines, affines_per_tile, conserve_flux, fscale, lut, n_phases,
                           oversampling, out, weight_out, h_out, w_out, stream, "resample_oversampled");
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
