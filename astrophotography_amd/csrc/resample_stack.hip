// resample_stack.hip - F3 + A7 in ONE launch (round 6): the registered frames of a co-add are resampled onto the common grid
// and sigma-clipped along N without the resampled slab ever existing in memory.  Two kernels: resample_clip_kernel_v2 (the one
// that runs: 512 threads, tile records and weight table in LDS, footprints fetched two frames ahead) and resample_clip_kernel (the
// first form, kept for weight tables of more than 1024 phases, which do not fit LDS).  Measured: correct and SLOWER than the two
// launches it replaces (6.2 against 5.1 ms for C5's share; DESIGN 4.4d has the ablations) - an opt-in that saves the slab's memory.
//
// The reference's flow is one SWarp call per stack (scripts/resample_all.sh:330-342: RESAMPLING_TYPE LANCZOS3, COMBINE_TYPE ...),
// this build's was two: apgpu_resample_affine_f32 writes [N][H][W] float32, apgpu_stack_sigclip reads it back - for C5's
// per-GPU share (16 x 8192^2) 8.6 of the 13.2 GB the step moves, plus a redo pass over the 6 % of the pixels whose column holds a
// NaN (frame borders, the footprints of bad pixels).  Here a workgroup owns one 64 x 16 OUTPUT tile: for each frame in turn it
// stages the tile's input footprint in LDS and evaluates its pixels' 6 x 6 windows exactly as resample_affine_kernel does
// (resample_core.h: same tile records, same fills, same window arithmetic - the values are bit for bit those of
// apgpu_resample_affine_f32), keeps the four values per lane and frame in registers - four columns of N values per lane - and
// then reduces each column like the lean stack kernels: non-finite values and bad-pixel hits become +inf sentinels ("frame
// absent"), pruned sort, clip_fast32 with per-lane sentinel counts, outputs; a wavefront with an unsure lane takes the exact
// float64 clip on the spot (there is no slab a redo pass could gather from).  HBM traffic: 4 N P (+ halo, L2 hits) read,
// 4 P written: 4.6 instead of 13.2 GB for C5's share.
//
// Bad-pixel mask: as in the two-step form the frames are resampled WITHOUT the mask; mask_bits_kernel walks list x frames and
// sets bit f of the output pixel's word in a [H][W] uint32 plane for every output pixel whose window in frame f holds a listed
// bad pixel (the scatter of resample.hip, into bits instead of NaNs), and the reduction reads one word per pixel.  A list that
// overflows, per-tile transforms and strongly magnifying transforms apply the mask at the footprint fill (tile by tile, decided
// on the device like there).
//
// Results: survivors identical to apgpu_stack_sigclip(apgpu_resample_affine_f32(..)) - same values, same clip - and the mean within
// the fast path's float32 rounding of it (tests/test_gpu_resample_stack.py: counts equal, mean <= 1 ulp of the oracle's
// float64 evaluation, bit-equal to the two-step form where both took the same path).
#include "stack_kernels.h"
#include "resample_core.h"

#include <hip/hip_runtime.h>

namespace {
using namespace apgpu_stack;

typedef float v16f __attribute__((ext_vector_type(16)));

struct FusedArgs {
    const TileRec *recs;        // [n_frames][gy][gx] (resample_tiles_kernel, th = 16)
    const float *lut;
    const uint32_t *bits;       // [h_out][w_out] bad-pixel hits per frame (bit f), or NULL
    const uint8_t *mask;        // the input mask (for tiles that apply it at the fill), or NULL
    const int *mask_ctl;        // the bad-pixel list's counter (NULL without a mask)
    int mask_cap;
    int gx, gy, log2_phases;
    int h_in, w_in, h_out, w_out;
    int n_frames;
};

// Output pixels whose window in frame f holds a listed bad pixel: bit f of the pixel's word (see mask_scatter_kernel, whose walk
// this is - candidates from the inverted transform in float64, each tested with the kernel's own fixed-point coordinates).
__global__ __launch_bounds__(256) void mask_bits_kernel(const int *__restrict__ ctl, const int *__restrict__ list, int cap,
                                                       const double *__restrict__ affines, int w_in, uint32_t *__restrict__ bits,
                                                       int h_out, int w_out)
{
    const int count = ctl[0];
    if (count > cap) return;                                   // the fused kernel applies the mask at the fill
    const int f = blockIdx.y;
    const double *A = affines + 6 * (int64_t)f;
    const double a[6] = {A[0], A[1], A[2], A[3], A[4], A[5]};
    if (!mask_scatter_ok(a)) return;                           // (ditto, for this frame: its tiles carry kInlineMask)
    long long F[6];
#pragma unroll
    for (int k = 0; k < 6; k++) F[k] = __double2ll_rn(a[k] * 4294967296.0);
    const double idet = 1.0 / fma(a[0], a[4], -(a[1] * a[3]));
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
        const int p = list[e];
        const int by = p / w_in, bx = p - by * w_in;
        double u0 = __builtin_inf(), u1 = -__builtin_inf(), v0 = __builtin_inf(), v1 = -__builtin_inf();
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double xi = (double)(bx + ((c & 1) ? 4 : -3)) - a[2], yi = (double)(by + ((c & 2) ? 4 : -3)) - a[5];
            const double u = (a[4] * xi - a[1] * yi) * idet, v = (a[0] * yi - a[3] * xi) * idet;
            u0 = fmin(u0, u); u1 = fmax(u1, u);
            v0 = fmin(v0, v); v1 = fmax(v1, v);
        }
        long long ua = (long long)floor(u0) - 2, ub = (long long)ceil(u1) + 2, va = (long long)floor(v0) - 2, vb = (long long)ceil(v1) + 2;
        ua = ua < 0 ? 0 : ua; va = va < 0 ? 0 : va;
        ub = ub > w_out - 1 ? w_out - 1 : ub; vb = vb > h_out - 1 ? h_out - 1 : vb;
        for (long long v = va; v <= vb; v++) {
            for (long long u = ua; u <= ub; u++) {
                const unsigned long long X = (unsigned long long)F[0] * (unsigned long long)u + (unsigned long long)F[1] * (unsigned long long)v + (unsigned long long)F[2];
                const unsigned long long Y = (unsigned long long)F[3] * (unsigned long long)u + (unsigned long long)F[4] * (unsigned long long)v + (unsigned long long)F[5];
                const long long jx = (long long)X >> 32, jy = (long long)Y >> 32;
                if (jx >= bx - 3 && jx <= bx + 2 && jy >= by - 3 && jy <= by + 2) atomicOr(bits + v * (int64_t)w_out + u, 1u << f);
            }
        }
    }
}

// The tile records are read through the CONSTANT address space: the record pointer is an ordinary kernel argument inside a struct
// (no __restrict__), the kernel stores to global memory, so the compiler would read a record with VECTOR loads - and wait for them
// with vmcnt(0), i.e. for every footprint load in flight as well (the counter is in order): the first build of the kernels below
// ran the frames of a tile strictly one after the other for this reason alone (8.6 ms per 16 x 8192^2 against 5.1 for the two-step
// form; profiles/r06/bench_fused_v1.txt).  Nothing writes the records during the kernel: scalar loads.
typedef const TileRec __attribute__((address_space(4))) CTileRec;
__device__ __forceinline__ CTileRec *crec(const TileRec *p) { return (CTileRec *)(uintptr_t)p; }

// The tile of frame f: record -> context (tile-uniform: scalar loads).
__device__ __forceinline__ unsigned load_tile(const TileRec *rpg, TileCtx &tc)
{
    CTileRec *rp = crec(rpg);
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
    return flags;
}

// NP: column slots (a multiple of 4, n_frames <= NP; the slots beyond n_frames are sentinels).
template <int NP>
__global__ __launch_bounds__(256, 3) void resample_clip_kernel(const StackParams prm, const FusedArgs fa)
{
    static_assert(NP % 4 == 0 && NP >= 4 && NP <= 16, "column slots");
    constexpr int TH = kTileH;
    using G = FastGeom<TH>;
    __shared__ __attribute__((aligned(16))) float tile[G::kLdsFloats];
    // tile order as in resample_affine_kernel: a frame's tiles in 8 contiguous ranges, one per XCD
    const int per_frame = fa.gx * fa.gy;
    const int chunk = (per_frame + 7) >> 3;
    const int rem = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (rem >= per_frame) return;
    const int tyi = rem / fa.gx, txi = rem - tyi * fa.gx;
    const int x0 = txi * kTileW, y0 = tyi * TH;
    const int tid = threadIdx.x;
    const int lx = tid % kTileW, ly = tid / kTileW;            // the lane's pixels: column x0 + lx, rows y0 + ly + 4 k
    const int x = x0 + lx;
    const int h_in = fa.h_in, w_in = fa.w_in, h_out = fa.h_out, w_out = fa.w_out;
    const int sh = 32 - fa.log2_phases;
    const int N = fa.n_frames;
    const v4i lrsrc = make_rsrc(fa.lut, (unsigned)((1 << fa.log2_phases) + 1) * 24u);
    const bool list_overflow = fa.mask != nullptr && fa.mask_ctl[0] > fa.mask_cap;
    v16f col[4];
#pragma unroll
    for (int k = 0; k < 4; k++) col[k] = v16f(__builtin_inff());

#pragma unroll 1
    for (int f = 0; f < N; f++) {
        TileCtx tc;
        const unsigned flags = load_tile(fa.recs + ((int64_t)f * per_frame + rem), tc);
        const bool fast = (flags & kFast) != 0, interior = (flags & kInterior) != 0;
        FrameView fv;
        fv.src = static_cast<const float *>(prm.frames) + (int64_t)f * prm.stride;
        fv.mask = nullptr;
        fv.h_in = h_in;
        fv.w_in = w_in;
        const bool inline_mask = fa.mask != nullptr && ((flags & kInlineMask) != 0 || list_overflow);
        if (inline_mask) fv.mask = fa.mask;
        if (f > 0) __syncthreads();                           // the previous frame's windows have been read
        if (fast) {
            if (inline_mask) {
                FastFill<true, G::kTrips> ff;
                fast_fill_issue<true, G::kTrips>(ff, fv.src, fa.mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
                fast_fill_store<true, G::kTrips, G::kOffB>(ff, tc.fh, tile, tid);
            } else {
                FastFill<false, G::kTrips> ff;
                fast_fill_issue<false, G::kTrips>(ff, fv.src, fa.mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
                fast_fill_store<false, G::kTrips, G::kOffB>(ff, tc.fh, tile, tid);
            }
        } else if (tc.staged) {
            if (inline_mask) general_fill<true>(tc, fv, tile, tid);
            else general_fill<false>(tc, fv, tile, tid);
        }
        __syncthreads();
        float r[4] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        if (x < w_out) {
            const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
            if (fast) {
                // (pixels_fast of resample.hip: coordinates relative to the footprint origin + 2)
                const unsigned long long us = (unsigned long long)(long long)x0, vs = (unsigned long long)(long long)y0;
                const unsigned long long Xs = F0 * us + F1 * vs + F2 - ((unsigned long long)(unsigned)(tc.bx0 + 2) << 32);
                const unsigned long long Ys = F3 * us + F4 * vs + F5 - ((unsigned long long)(unsigned)(tc.by0 + 2) << 32);
                unsigned long long X = Xs + F0 * (unsigned long long)(unsigned)lx + F1 * (unsigned long long)(unsigned)ly;
                unsigned long long Y = Ys + F3 * (unsigned long long)(unsigned)lx + F4 * (unsigned long long)(unsigned)ly;
                const unsigned long long dX = F1 * 4ull, dY = F4 * 4ull;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const FastPrep cur = prep_fast<G::kOffB>(X, Y, sh, lrsrc);
                    X += dX;
                    Y += dY;
                    const float v = eval_fast(cur, tile);
                    r[k] = (v == v) ? v * tc.fs : __builtin_nanf("");
                }
            } else {
                const unsigned long long u0 = (unsigned long long)(long long)x, v0 = (unsigned long long)(long long)(y0 + ly);
                unsigned long long X = F0 * u0 + F1 * v0 + F2;
                unsigned long long Y = F3 * u0 + F4 * v0 + F5;
                const unsigned long long dX = F1 * 4ull, dY = F4 * 4ull;
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    const int y = y0 + ly + 4 * k;
                    if (y < h_out) {
                        const float s = interior ? sample_general<true>(tc, fv, tile, fa.lut, X, Y, sh, tc.sane_top)
                                                 : sample_general<false>(tc, fv, tile, fa.lut, X, Y, sh, tc.sane_top);
                        r[0] = k == 0 ? s : r[0];
                        r[1] = k == 1 ? s : r[1];
                        r[2] = k == 2 ? s : r[2];
                        r[3] = k == 3 ? s : r[3];
                    }
                    X += dX;
                    Y += dY;
                }
            }
        }
        // (f is wave-uniform: register-indexed moves, no scratch)
        col[0][f] = r[0];
        col[1][f] = r[1];
        col[2][f] = r[2];
        col[3][f] = r[3];
    }

    // the reduction: the lane's four columns, one after the other
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int y = y0 + ly + 4 * k;
        const bool inside = x < w_out && y < h_out;
        const int64_t p = inside ? (int64_t)y * w_out + x : 0;
        const v16f cur = k == 0 ? col[0] : (k == 1 ? col[1] : (k == 2 ? col[2] : col[3]));
        const uint32_t hit = (fa.bits && inside) ? fa.bits[p] : 0u;
        float c[NP];
        int n = 0;
#pragma unroll
        for (int f = 0; f < NP; f++) {
            const float xv = cur[f];
            const bool ok = (fabsf(xv) < __builtin_inff()) && ((hit >> f) & 1u) == 0u;      // slots >= n_frames hold +inf
            c[f] = ok ? xv : __builtin_inff();
            n += ok ? 1 : 0;
        }
        bool good = false;
        if constexpr (NP >= 16) {
            constexpr int T = 6;
            if (fast32_wanted(prm)) {
                good = inside;
                sort_column<NP, T>(c);
                int nonfin = 0;                                // the sentinels sit at the top: a full tail = too many
#pragma unroll
                for (int j = 1; j <= T; j++) nonfin += (c[NP - j] == __builtin_inff()) ? 1 : 0;
                good = good && nonfin < T;
                if (wave_any(good)) finish_fast_column<NP, T, false>(c, good, p, 0, nonfin);
                if (inside && !good) {                         // an unsure lane: the exact clip, here
                    const StackParams q = read_params(late_params());
                    reduce_and_store<NP, NP>(q, c, n, p, true, false);
                }
                continue;
            }
        }
        if (inside) {
            sort_column<NP>(c);
            const StackParams q = read_params(late_params());
            reduce_and_store<NP, NP>(q, c, n, p, false, false);
        }
    }
}

// ---- the same, built for latency (v2) -------------------------------------------------------------------------------------
// v1 above runs one frame of a tile after the other with three dependent memory round trips per frame (tile record -> footprint
// loads -> barrier -> weight-table gathers) and three workgroups per CU to overlap them: 7.6 ms per 16 x 8192^2 where the two-step
// form takes 5.1 (profiles/r06/bench_fused_v1.txt).  Here
//   * a workgroup has 512 threads - two pixels per lane: 32 instead of 64 registers of columns - and stays under 128 VGPRs:
//     four wavefronts per SIMD;
//   * the Lanczos weight table (up to 1025 rows of 24 bytes) is copied into LDS once per workgroup: a pixel's two table rows are
//     six ds_read_b64 instead of four gathers from memory (in resample_affine_kernel the gathers were what the kernel waited for;
//     staging the table lost there because its LDS cost occupancy - here the registers bound the occupancy first);
//   * the footprints of the next TWO frames are on their way while a frame is evaluated (two register sets, two LDS buffers,
//     ONE barrier per frame): per CU four footprint fills are in flight at any time.
// Tiles with a frame off the fast path (frame borders, large transforms, masks applied at the fill) take the frame-by-frame
// loop of v1 with this kernel's geometry.  Tables of more than 1024 phases: v1.
struct FusedGeom {
    static constexpr int kThreads = 512;
    static constexpr int kRowsPerTrip = 6;                                  // 480 of the 512 lanes: six footprint rows of 80 columns
    static constexpr int kTrips = 5;                                        // 30 rows >= 16 + 10
    static constexpr int kCopy = kFastPitch * kRowsPerTrip * kTrips;        // 2400 floats per copy
    static constexpr int kOffB = ((kCopy + 1 + 31) / 64) * 64 + 32;         // copy B: 32 banks after copy A (resample_core.h, FastGeom)
    static constexpr int kBuf = kOffB + kCopy;                              // 4864 floats = 19 KB per buffer
    static constexpr int kTableRows = 1025, kTableFloats = kTableRows * 6;  // 24.6 KB
    static_assert(kOffB % 64 == 32 && kBuf >= kGenericFloats && kTileH + 10 <= kRowsPerTrip * kTrips, "LDS layout");
};

template <bool HAS_MASK>
struct FusedFill {
    float val[FusedGeom::kTrips];
    char mk[FusedGeom::kTrips];
};

// EVERY lane issues EVERY trip's load, unconditionally: the compiler's wait-count pass can then count the loads in flight (behind
// "if (row < fh && tid < 480)" each load sat in its own exec-masked block, their number was unknown, and every wait for one
// footprint became vmcnt(0) - a wait for the footprints fetched ahead as well).  Rows beyond the footprint's height and the lanes
// 480 .. 511 (which address the first columns of the next trip's first row) read inside the frame's buffer or get the bounds
// check's 0; fused_fill_store drops them.
template <bool HAS_MASK>
__device__ __forceinline__ void fused_fill_issue(FusedFill<HAS_MASK> &ff, const float *src, const uint8_t *mask, int bx0, int by0, int fh,
                                                 int h_in, int w_in, int tid)
{
    using G = FusedGeom;
    const v4i irsrc = make_rsrc(src, (unsigned)(h_in * w_in) * 4u);
    const v4i mrsrc = make_rsrc(mask, (unsigned)(h_in * w_in));
    const int r = tid / kFastPitch, c = tid - r * kFastPitch;
    const int e0 = (by0 + r) * w_in + bx0 + c;
    const int estep = G::kRowsPerTrip * w_in;
#pragma unroll
    for (int k = 0; k < G::kTrips; k++) {
        ff.val[k] = apgpu_buffer_load_f32(irsrc, (e0 + k * estep) * 4, 0, 0);
        if constexpr (HAS_MASK) ff.mk[k] = apgpu_buffer_load_i8(mrsrc, e0 + k * estep, 0, 0);
        else ff.mk[k] = 0;
    }
}

template <bool HAS_MASK>
__device__ __forceinline__ void fused_fill_store(const FusedFill<HAS_MASK> &ff, int fh, float *tile, int tid)
{
    using G = FusedGeom;
#pragma unroll
    for (int k = 0; k < G::kTrips; k++) {
        if (G::kRowsPerTrip * k < fh && tid < G::kRowsPerTrip * kFastPitch) {
            const bool good = (fabsf(ff.val[k]) < __builtin_inff()) && ff.mk[k] == 0;
            const float xv = good ? ff.val[k] : __builtin_nanf("");
            tile[tid + G::kRowsPerTrip * kFastPitch * k] = xv;                  // copy A
            tile[G::kOffB - 1 + tid + G::kRowsPerTrip * kFastPitch * k] = xv;   // copy B: element e - 1
        }
    }
}

// Two VERTICALLY ADJACENT pixels of a fast tile at once (rows y and y + 1 of one column).  For a registration-sized transform
// the lower pixel's window is the same six columns one input row further down - then the two windows share five of their six
// rows: the seven rows are read ONCE (21 ds_read_b64 instead of 36) and every row feeds both pixels' sums, each in its own order
// (row j is row j of the upper window and row j - 1 of the lower one: per pixel exactly eval_fast's sequence of operations).  The
// y weights of the lower pixel are those of the upper one unless its phase moved on (read again behind a wave vote); a wavefront
// in which some lane's windows do not line up (the column or the copy changed: every 1 / |sin(rotation)| rows) evaluates the two
// pixels one after the other.  LDS reads per pixel: 15 instead of 24 - with the weight table in LDS the fused kernel's evaluation
// was bound by LDS bandwidth (profiles/r06/ablate_fused.txt).
__device__ __forceinline__ void eval_pair_lds(unsigned long long Xr, unsigned long long Yr, unsigned long long F1, unsigned long long F4, int sh,
                                              const float *tile, const float *tab, float &v0, float &v1)
{
    typedef __attribute__((address_space(3))) const v2f *lds_v2f;
    int js0, jr0, px0, py0, js1, jr1, px1, py1;
    phases(Xr, Yr, sh, js0, jr0, px0, py0);
    phases(Xr + F1, Yr + F4, sh, js1, jr1, px1, py1);
    const unsigned s0 = (unsigned)js0, r0 = (unsigned)jr0, s1 = (unsigned)js1, r1 = (unsigned)jr1;
    const int idx0 = (int)mad_u24(s0 & 1u, (unsigned)(FusedGeom::kOffB - 1), mad_u24(r0, (unsigned)kFastPitch, s0));
    const int idx1 = (int)mad_u24(s1 & 1u, (unsigned)(FusedGeom::kOffB - 1), mad_u24(r1, (unsigned)kFastPitch, s1));
    lds_v2f wa = (lds_v2f)(tab + 6 * px0), wb = (lds_v2f)(tab + 6 * px1), wc = (lds_v2f)(tab + 6 * py0);
    const v2f ax01 = wa[0], ax23 = wa[1], ax45 = wa[2];
    const v2f bx01 = wb[0], bx23 = wb[1], bx45 = wb[2];
    v2f ay01 = wc[0], ay23 = wc[1], ay45 = wc[2];
    v2f by01 = ay01, by23 = ay23, by45 = ay45;
    if (__builtin_amdgcn_ballot_w64(py1 != py0) != 0) {
        lds_v2f wd = (lds_v2f)(tab + 6 * py1);
        by01 = wd[0]; by23 = wd[1]; by45 = wd[2];
    }
    if (__builtin_amdgcn_ballot_w64(idx1 != idx0 + kFastPitch) == 0) {
        lds_pair_p t = (lds_pair_p)(tile + idx0);
        const float ay[6] = {ay01.x, ay01.y, ay23.x, ay23.y, ay45.x, ay45.y};
        const float by[6] = {by01.x, by01.y, by23.x, by23.y, by45.x, by45.y};
        v2f V0 = {0.f, 0.f}, V1 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const v2f q0 = t[j * (kFastPitch / 2) + 0], q1 = t[j * (kFastPitch / 2) + 1], q2 = t[j * (kFastPitch / 2) + 2];
            if (j < 6) {
                v2f acc = ax01 * q0;
                acc = __builtin_elementwise_fma(ax23, q1, acc);
                acc = __builtin_elementwise_fma(ax45, q2, acc);
                const v2f w = {ay[j], ay[j]};
                V0 = (j == 0) ? w * acc : __builtin_elementwise_fma(w, acc, V0);
            }
            if (j > 0) {
                v2f acc = bx01 * q0;
                acc = __builtin_elementwise_fma(bx23, q1, acc);
                acc = __builtin_elementwise_fma(bx45, q2, acc);
                const v2f w = {by[j - 1], by[j - 1]};
                V1 = (j == 1) ? w * acc : __builtin_elementwise_fma(w, acc, V1);
            }
            if (j == 2 || j == 4) __builtin_amdgcn_sched_barrier(0);
        }
        v0 = V0.x + V0.y;
        v1 = V1.x + V1.y;
    } else {
        FastPrep p0, p1;
        p0.w.wx01 = ax01; p0.w.wx23 = ax23; p0.w.wx45 = ax45; p0.w.wy01 = ay01; p0.w.wy23 = ay23; p0.w.wy45 = ay45; p0.idx = idx0;
        p1.w.wx01 = bx01; p1.w.wx23 = bx23; p1.w.wx45 = bx45; p1.w.wy01 = by01; p1.w.wy23 = by23; p1.w.wy45 = by45; p1.idx = idx1;
        v0 = eval_fast(p0, tile);
        v1 = eval_fast(p1, tile);
    }
}

#ifndef APGPU_FUSED_ABLATE
#define APGPU_FUSED_ABLATE 0                                 // development (tools/variant_lib.sh): 1 no window evaluation, 2 no footprint fills,
#endif                                                       // 4 no table copy, 8 no clip (timing only: results are garbage)
template <int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void resample_clip_kernel_v2(const StackParams prm, const FusedArgs fa)
{
    static_assert(NP % 4 == 0 && NP >= 4 && NP <= 16, "column slots");
    using G = FusedGeom;
    constexpr int TH = kTileH;
    __shared__ __attribute__((aligned(16))) float bufs[2][G::kBuf];
    __shared__ __attribute__((aligned(16))) float tab[G::kTableFloats];
    const int per_frame = fa.gx * fa.gy;
    const int chunk = (per_frame + 7) >> 3;
    const int rem = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (rem >= per_frame) return;
    const int tyi = rem / fa.gx, txi = rem - tyi * fa.gx;
    const int x0 = txi * kTileW, y0 = tyi * TH;
    const int tid = threadIdx.x;
    const int lx = tid % kTileW, ly = 2 * (tid / kTileW);      // the lane's pixels: column x0 + lx, rows y0 + ly and y0 + ly + 1
    const int x = x0 + lx;
    const int h_in = fa.h_in, w_in = fa.w_in, h_out = fa.h_out, w_out = fa.w_out;
    const int sh = 32 - fa.log2_phases;
    const int N = fa.n_frames;
    const bool list_overflow = fa.mask != nullptr && fa.mask_ctl[0] > fa.mask_cap;
    const TileRec *const rec0 = fa.recs + rem;
    // the tile's N records -> LDS with ONE coalesced load (16 lanes per 64-byte record): read frame by frame with scalar loads, each
    // first touch of a record was a memory round trip in front of a barrier - 2.1 ms of the first v2's 7.5 (ablate_fused.txt)
    __shared__ __attribute__((aligned(16))) unsigned rl[16 * 16];
    __shared__ int nlist;
    if (tid < N * 16) rl[tid] = reinterpret_cast<const unsigned *>(rec0 + (int64_t)(tid >> 4) * per_frame)[tid & 15];
    if (tid == 0) nlist = 0;
    __syncthreads();
    auto rdw = [&](int f, int k) { return (unsigned)__builtin_amdgcn_readfirstlane((int)rl[f * 16 + k]); };     // dword k of frame f's record
    auto rec_ctx = [&](int f, TileCtx &tc) {
#pragma unroll
        for (int k = 0; k < 6; k++) tc.F[k] = (long long)(((unsigned long long)rdw(f, 2 * k + 1) << 32) | rdw(f, 2 * k));
        tc.bx0 = (int)rdw(f, 12);
        tc.by0 = (int)rdw(f, 13);
        const unsigned dims = rdw(f, 14);
        tc.fw = (int)(dims & 0xfffu);
        tc.fh = (int)((dims >> 12) & 0xfffu);
        const unsigned flags = dims >> 24;
        tc.staged = (flags & kStaged) != 0;
        tc.sane_top = (flags & kSaneTop) != 0;
        tc.sane_bot = (flags & kSaneBot) != 0;
        tc.fs = __uint_as_float(rdw(f, 15));
        return flags;
    };
    // every frame's tile on the fast path, no mask at the fill?  (lane f of every wavefront looks at frame f)
    bool all_fast;
    {
        const int f = tid & 63;
        bool bad = false;
        if (f < N) {
            const unsigned flags = rl[f * 16 + 14] >> 24;
            bad = (flags & kFast) == 0 || (fa.mask != nullptr && ((flags & kInlineMask) != 0 || list_overflow));
        }
        all_fast = __builtin_amdgcn_ballot_w64(bad) == 0;
    }
    v16f col[2];
    col[0] = v16f(__builtin_inff());
    col[1] = v16f(__builtin_inff());
    const float *const frames = static_cast<const float *>(prm.frames);
    const int64_t fstride = prm.stride;

    if (all_fast) {
        // Two register sets (fA, fB) and two LDS buffers; every step issues the loads of the frame two ahead UNCONDITIONALLY (the
        // last steps fetch the last frame again: a cache hit) and stores the set fetched one step ago - straight-line code with the
        // same number of loads in flight at the loop's entry and at its back edge, so that the compiler's wait counts are exact
        // (vmcnt(5) in front of a store: the five loads just issued stay in flight).  With the fetch behind "if (g + 2 < N)" the
        // join of the two paths made every wait a vmcnt(0).
        FusedFill<false> fA, fB;
        auto issue = [&](FusedFill<false> &ff, int g) {
            const int gc = g < N ? g : N - 1;
            fused_fill_issue<false>(ff, frames + (int64_t)gc * fstride, nullptr, (int)rdw(gc, 12), (int)rdw(gc, 13), 0, h_in, w_in, tid);
        };
        auto store = [&](const FusedFill<false> &ff, int g, float *buf) {
            const int gc = g < N ? g : N - 1;
            fused_fill_store<false>(ff, (int)((rdw(gc, 14) >> 12) & 0xfffu), buf, tid);
        };
        issue(fA, 0);
        if (!(APGPU_FUSED_ABLATE & 4)) {
            // the weight table -> LDS ((2^log2_phases + 1) rows of 6 floats; 8-byte aligned)
            const int nfl2 = (((1 << fa.log2_phases) + 1) * 6) / 2;
            const v2f *src = reinterpret_cast<const v2f *>(fa.lut);
            v2f *dst = reinterpret_cast<v2f *>(tab);
            for (int i = tid; i < nfl2; i += G::kThreads) dst[i] = src[i];
        }
        store(fA, 0, bufs[0]);
        __syncthreads();
        issue(fB, 1);
#pragma unroll 1
        for (int f = 0; f < N; f += 2) {
#pragma unroll
            for (int h = 0; h < 2; h++) {                      // h = 0: frame g in bufs[0], set B in flight, set A re-issued; h = 1: the mirror image
                const int g = f + h;
                FusedFill<false> &fetch = h == 0 ? fA : fB;    // receives frame g + 2
                FusedFill<false> &ready = h == 0 ? fB : fA;    // holds frame g + 1 (issued one step ago)
                if (!(APGPU_FUSED_ABLATE & 2)) issue(fetch, g + 2);
                TileCtx tc;
                rec_ctx(g, tc);
                float r0v = __builtin_nanf(""), r1v = __builtin_nanf("");
                if (x < w_out && !(APGPU_FUSED_ABLATE & 1)) {
                    const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
                    const unsigned long long us = (unsigned long long)(long long)x0, vs = (unsigned long long)(long long)y0;
                    const unsigned long long Xs = F0 * us + F1 * vs + F2 - ((unsigned long long)(unsigned)(tc.bx0 + 2) << 32);
                    const unsigned long long Ys = F3 * us + F4 * vs + F5 - ((unsigned long long)(unsigned)(tc.by0 + 2) << 32);
                    const unsigned long long X = Xs + F0 * (unsigned long long)(unsigned)lx + F1 * (unsigned long long)(unsigned)ly;
                    const unsigned long long Y = Ys + F3 * (unsigned long long)(unsigned)lx + F4 * (unsigned long long)(unsigned)ly;
                    float v0, v1;
                    eval_pair_lds(X, Y, F1, F4, sh, bufs[h], tab, v0, v1);
                    r0v = (v0 == v0) ? v0 * tc.fs : __builtin_nanf("");
                    r1v = (v1 == v1) ? v1 * tc.fs : __builtin_nanf("");
                }
                col[0][g] = r0v;
                col[1][g] = r1v;
                if (!(APGPU_FUSED_ABLATE & 2)) store(ready, g + 1, bufs[h ^ 1]);
                __syncthreads();
                if (g + 1 >= N) break;
            }
        }
    } else {
        // frame by frame (v1's loop): any path per frame, the mask at the fill where the tile asks for it
        float *const tile = bufs[0];
        const v4i lrsrc = make_rsrc(fa.lut, (unsigned)((1 << fa.log2_phases) + 1) * 24u);
#pragma unroll 1
        for (int f = 0; f < N; f++) {
            TileCtx tc;
            const unsigned flags = rec_ctx(f, tc);
            const bool fast = (flags & kFast) != 0, interior = (flags & kInterior) != 0;
            FrameView fv;
            fv.src = frames + (int64_t)f * fstride;
            fv.mask = nullptr;
            fv.h_in = h_in;
            fv.w_in = w_in;
            const bool inline_mask = fa.mask != nullptr && ((flags & kInlineMask) != 0 || list_overflow);
            if (inline_mask) fv.mask = fa.mask;
            if (f > 0) __syncthreads();
            if (fast) {
                if (inline_mask) {
                    FusedFill<true> ff;
                    fused_fill_issue<true>(ff, fv.src, fa.mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
                    fused_fill_store<true>(ff, tc.fh, tile, tid);
                } else {
                    FusedFill<false> ff;
                    fused_fill_issue<false>(ff, fv.src, fa.mask, tc.bx0, tc.by0, tc.fh, h_in, w_in, tid);
                    fused_fill_store<false>(ff, tc.fh, tile, tid);
                }
            } else if (tc.staged) {
                if (inline_mask) general_fill<true>(tc, fv, tile, tid);
                else general_fill<false>(tc, fv, tile, tid);
            }
            __syncthreads();
            float r[2] = {__builtin_nanf(""), __builtin_nanf("")};
            if (x < w_out) {
                const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
                if (fast) {
                    const unsigned long long us = (unsigned long long)(long long)x0, vs = (unsigned long long)(long long)y0;
                    const unsigned long long Xs = F0 * us + F1 * vs + F2 - ((unsigned long long)(unsigned)(tc.bx0 + 2) << 32);
                    const unsigned long long Ys = F3 * us + F4 * vs + F5 - ((unsigned long long)(unsigned)(tc.by0 + 2) << 32);
                    unsigned long long X = Xs + F0 * (unsigned long long)(unsigned)lx + F1 * (unsigned long long)(unsigned)ly;
                    unsigned long long Y = Ys + F3 * (unsigned long long)(unsigned)lx + F4 * (unsigned long long)(unsigned)ly;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const FastPrep cur = prep_fast<G::kOffB>(X, Y, sh, lrsrc);
                        X += F1;
                        Y += F4;
                        const float v = eval_fast(cur, tile);
                        r[k] = (v == v) ? v * tc.fs : __builtin_nanf("");
                    }
                } else {
                    const unsigned long long u0 = (unsigned long long)(long long)x, v0 = (unsigned long long)(long long)(y0 + ly);
                    unsigned long long X = F0 * u0 + F1 * v0 + F2;
                    unsigned long long Y = F3 * u0 + F4 * v0 + F5;
#pragma unroll 1
                    for (int k = 0; k < 2; k++) {
                        if (y0 + ly + k < h_out) {
                            const float sv = interior ? sample_general<true>(tc, fv, tile, fa.lut, X, Y, sh, tc.sane_top)
                                                      : sample_general<false>(tc, fv, tile, fa.lut, X, Y, sh, tc.sane_top);
                            r[0] = k == 0 ? sv : r[0];
                            r[1] = k == 1 ? sv : r[1];
                        }
                        X += F1;
                        Y += F4;
                    }
                }
            }
            col[0][f] = r[0];
            col[1][f] = r[1];
        }
    }

    // The reduction: the lane's two columns on the float32 fast path, one after the other.  A pixel the fast path cannot finish
    // (an unsure comparison, too many sentinels) needs the exact float64 clip - and here nearly EVERY wavefront holds one (C5: 6 % of
    // the pixels - the footprints of bad pixels, the frames' borders - i.e. 98 % of the 64-pixel wavefronts): running the exact clip
    // wavefront by wavefront, as the complete stack kernels do, cost 2.4 ms of the first v2's 7.5.  So such pixels are LISTED - pixel
    // index + the 16 values, in the LDS the footprints no longer need - and the workgroup reduces its list afterwards, one pixel per
    // lane, densely: about one wavefront's worth per workgroup.  (A list that overflows: the pixel is reduced on the spot.)
    if (APGPU_FUSED_ABLATE & 8) {
        if (x < w_out && y0 + ly + 1 < h_out && prm.mean) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int f = 0; f < NP; f++) { a0 += col[0][f]; a1 += col[1][f]; }
            prm.mean[(int64_t)(y0 + ly) * w_out + x] = a0;
            prm.mean[(int64_t)(y0 + ly + 1) * w_out + x] = a1;
        }
        return;
    }
    constexpr int kEntry = 18;                                 // pixel index (2 dwords) + 16 values
    constexpr int kListCap = (2 * G::kBuf) / kEntry;
    unsigned *const list = reinterpret_cast<unsigned *>(&bufs[0][0]);
    __syncthreads();                                           // the last frame's windows have been read: the footprint buffers are free
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        const int y = y0 + ly + k;
        const bool inside = x < w_out && y < h_out;
        const int64_t p = inside ? (int64_t)y * w_out + x : 0;
        const v16f cur = k == 0 ? col[0] : col[1];
        const uint32_t hit = (fa.bits && inside) ? fa.bits[p] : 0u;
        float c[NP];
        int n = 0;
#pragma unroll
        for (int f = 0; f < NP; f++) {
            const float xv = cur[f];
            const bool ok = (fabsf(xv) < __builtin_inff()) && ((hit >> f) & 1u) == 0u;
            c[f] = ok ? xv : __builtin_inff();
            n += ok ? 1 : 0;
        }
        bool good = false;
        if constexpr (NP >= 16) {
            constexpr int T = 6;
            if (fast32_wanted(prm)) {
                good = inside;
                sort_column<NP, T>(c);
                int nonfin = 0;
#pragma unroll
                for (int j = 1; j <= T; j++) nonfin += (c[NP - j] == __builtin_inff()) ? 1 : 0;
                good = good && nonfin < T;
                if (wave_any(good)) finish_fast_column<NP, T, false>(c, good, p, 0, nonfin);
            }
        }
        if (inside && !good) {
            const int slot = atomicAdd(&nlist, 1);
            if (slot < kListCap) {
                unsigned *e = list + slot * kEntry;
                e[0] = (unsigned)p;
                e[1] = (unsigned)((unsigned long long)p >> 32);
#pragma unroll
                for (int f = 0; f < NP; f++) e[2 + f] = __float_as_uint(c[f]);
            } else {
                sort_column<NP>(c);                            // (complete: after the pruned network, or unsorted)
                const StackParams q = read_params(late_params());
                reduce_and_store<NP, NP>(q, c, n, p, false, false);
            }
        }
    }
    __syncthreads();
    const int nl = nlist < kListCap ? nlist : kListCap;
#pragma unroll 1
    for (int i = tid; i < nl; i += G::kThreads) {
        const unsigned *e = list + i * kEntry;
        const int64_t p = (int64_t)(((unsigned long long)e[1] << 32) | e[0]);
        float c[NP];
        int n = 0;
#pragma unroll
        for (int f = 0; f < NP; f++) {
            c[f] = __uint_as_float(e[2 + f]);
            n += (c[f] < __builtin_inff()) ? 1 : 0;            // the sentinels are +inf; every other value is finite
        }
        sort_column<NP>(c);
        const StackParams q = read_params(late_params());
        reduce_and_store<NP, NP>(q, c, n, p, false, false);
    }
}

template <int NP>
void launch_fused(const StackParams &prm, const FusedArgs &fa, hipStream_t st)
{
    const int per_frame = fa.gx * fa.gy;
    const unsigned grid = (unsigned)(((per_frame + 7) / 8) * 8);
    static const bool v1_only = getenv("APGPU_FUSED_V1") != nullptr;           // development: the first form, for A/B timing
    if (fa.log2_phases <= 10 && !v1_only) hipLaunchKernelGGL((resample_clip_kernel_v2<NP>), dim3(grid), dim3(FusedGeom::kThreads), 0, st, prm, fa);
    else hipLaunchKernelGGL((resample_clip_kernel<NP>), dim3(grid), dim3(256), 0, st, prm, fa);
}

struct FusedWs {
    size_t recs_off, ctl_off, list_off, bits_off, total;
    int list_cap;
};

FusedWs fused_ws_layout(int32_t n_frames, int64_t h_in, int64_t w_in, int64_t h_out, int64_t w_out, bool has_mask)
{
    FusedWs w{};
    const int64_t gx = (w_out + kTileW - 1) / kTileW, gy = (h_out + kTileH - 1) / kTileH;
    size_t off = 0;
    w.recs_off = off;
    off += (size_t)n_frames * (size_t)gx * (size_t)gy * sizeof(TileRec);
    int64_t mcap64 = h_in * w_in / 64;
    if (mcap64 < 256) mcap64 = 256;
    w.list_cap = (int)(mcap64 > kMaskListCapMax ? kMaskListCapMax : mcap64);
    if (has_mask) {
        w.ctl_off = off;
        off += 64;
        w.list_off = off;
        off += (((size_t)w.list_cap * sizeof(int)) + 63) / 64 * 64;
        w.bits_off = off;
        off += (((size_t)h_out * (size_t)w_out * sizeof(uint32_t)) + 63) / 64 * 64;
    }
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t apgpu_resample_stack_ws_bytes(int32_t n_frames, int64_t h_in, int64_t w_in, int64_t h_out, int64_t w_out, int32_t has_mask)
{
    if (n_frames < 1 || h_in < 1 || w_in < 1 || h_out < 1 || w_out < 1) return 0;
    return fused_ws_layout(n_frames, h_in, w_in, h_out, w_out, has_mask != 0).total;
}

extern "C" int apgpu_resample_stack_sigclip(const apgpu_stack_args *args, int64_t h_in, int64_t w_in, const uint8_t *mask,
                                            const double *affines, int32_t affines_per_tile, int32_t conserve_flux, const float *fscale,
                                            const float *lut, int32_t n_phases, int64_t h_out, int64_t w_out, void *workspace,
                                            size_t workspace_bytes, void *stream)
{
    const char *who = "resample_stack_sigclip";
    if (!args || !args->frames || !affines || !lut) return fail(APGPU_EINVAL, "%s: NULL pointer argument", who);
    if (args->dtype != APGPU_F32) return fail(APGPU_EUNSUPPORTED, "%s: float32 frames", who);
    const int N = args->n_frames;
    if (N < 1 || N > 16) return fail(APGPU_EUNSUPPORTED, "%s: n_frames = %d (1 .. 16 per call; resample and stack larger sets in two steps)", who, N);
    if (h_in < 6 || w_in < 6 || h_out <= 0 || w_out <= 0) return fail(APGPU_EINVAL, "%s: bad shape", who);
    if (args->n_pixels != h_out * w_out) return fail(APGPU_EINVAL, "%s: n_pixels %lld != h_out * w_out", who, (long long)args->n_pixels);
    if (h_in > 0x3fffffff || w_in > 0x3fffffff || h_out > 0x3fffffff || w_out > 0x3fffffff)
        return fail(APGPU_EUNSUPPORTED, "%s: image sides are limited to 2^30 pixels", who);
    if (n_phases < 2 || n_phases > (1 << 20) || (n_phases & (n_phases - 1)))
        return fail(APGPU_EINVAL, "%s: n_phases = %d (a power of two, 2 .. 2^20)", who, n_phases);
    if (reinterpret_cast<uintptr_t>(lut) & 7) return fail(APGPU_EINVAL, "%s: lut must be 8-byte aligned", who);
    if (args->bias || args->dark || args->nflat || args->pedestal || args->pixmask)
        return fail(APGPU_EUNSUPPORTED, "%s: no fused calibration / pixel mask (calibrate first; the bad-pixel mask is `mask`)", who);
    if (args->median || args->std || args->mean_f64 || args->std_f64)
        return fail(APGPU_EUNSUPPORTED, "%s: outputs are mean, count and moments", who);
    if (!args->mean && !args->count && !args->moments) return fail(APGPU_EINVAL, "%s: no output requested", who);
    if (args->center != APGPU_CENTER_MEDIAN && args->center != APGPU_CENTER_MEAN) return fail(APGPU_EINVAL, "%s: bad center %d", who, args->center);
    if (args->dev != APGPU_DEV_STD) return fail(APGPU_EUNSUPPORTED, "%s: dev must be APGPU_DEV_STD", who);
    if (args->maxiters == 0) return fail(APGPU_EINVAL, "%s: maxiters must be >= 1 or < 0", who);
    if (!(args->sigma_lower >= 0.0) || !(args->sigma_upper >= 0.0)) return fail(APGPU_EINVAL, "%s: sigma must be >= 0", who);
    if (args->moments_f64 < 0 || args->moments_f64 > 4) return fail(APGPU_EINVAL, "%s: bad moments_f64 %d", who, args->moments_f64);
    if (args->flags & ~(APGPU_STACK_EXACT_MOMENTS | APGPU_STACK_MOMENTS_MEAN)) return fail(APGPU_EINVAL, "%s: unknown flags 0x%x", who, args->flags);
    const int64_t fstride = args->frame_stride > 0 ? args->frame_stride : h_in * w_in;
    if (fstride < h_in * w_in) return fail(APGPU_EINVAL, "%s: frame_stride < h_in * w_in", who);
    const FusedWs wl = fused_ws_layout(N, h_in, w_in, h_out, w_out, mask != nullptr);
    if (!workspace || workspace_bytes < wl.total || (reinterpret_cast<uintptr_t>(workspace) & 63))
        return fail(APGPU_EWORKSPACE, "%s: a 64-byte aligned workspace of %zu bytes is needed (apgpu_resample_stack_ws_bytes)", who, wl.total);
    int log2_phases = 0;
    while ((1 << log2_phases) < n_phases) log2_phases++;
    hipStream_t st = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    TileRec *recs = reinterpret_cast<TileRec *>(ws + wl.recs_off);
    const int64_t gx = (w_out + kTileW - 1) / kTileW, gy = (h_out + kTileH - 1) / kTileH;
    const int64_t ntiles = (int64_t)N * gx * gy;
    if (ntiles > 0x7ffffff0LL || gx * gy > 0x7ffffff0LL) return fail(APGPU_EUNSUPPORTED, "%s: too many tiles (%lld)", who, (long long)ntiles);
    const int fast_ok = (h_in * w_in < (1LL << 30)) && (w_out < (1LL << 25));
    const int mask_scatter = mask && !affines_per_tile && h_in * w_in < (1LL << 31) && N <= 32;
    int *mctl = nullptr;
    uint32_t *bits = nullptr;
    if (mask) {
        mctl = reinterpret_cast<int *>(ws + wl.ctl_off);
        hipError_t e = hipMemsetAsync(mctl, 0, 64, st);
        if (e == hipSuccess && mask_scatter) {
            bits = reinterpret_cast<uint32_t *>(ws + wl.bits_off);
            e = hipMemsetAsync(bits, 0, (size_t)h_out * (size_t)w_out * sizeof(uint32_t), st);
        }
        if (e != hipSuccess) return fail(APGPU_ELAUNCH, "%s: memset failed: %s", who, hipGetErrorString(e));
        if (mask_scatter) {
            int64_t g = (h_in * w_in / 16 / 16 + 255) / 256;
            if (g < 1) g = 1;
            if (g > kNumCU * 32) g = kNumCU * 32;
            hipLaunchKernelGGL(mask_list_kernel, dim3((unsigned)g), dim3(256), 0, st, mask, h_in * w_in, wl.list_cap, mctl,
                               reinterpret_cast<int *>(ws + wl.list_off));
        }
    }
    const int64_t tb = (ntiles + 255) / 256;
    hipLaunchKernelGGL(resample_tiles_kernel, dim3((unsigned)tb), dim3(256), 0, st, affines, affines_per_tile, conserve_flux, fscale, 1, (int)kTileH,
                       (int)gx, (int)gy, ntiles, (int)h_in, (int)w_in, (int)h_out, (int)w_out, fast_ok, mask_scatter, recs);
    int rc = check_launch(who);
    if (rc != APGPU_OK) return rc;
    if (mask_scatter) {
        hipLaunchKernelGGL(mask_bits_kernel, dim3(64, (unsigned)N), dim3(256), 0, st, mctl, reinterpret_cast<int *>(ws + wl.list_off), wl.list_cap,
                           affines, (int)w_in, bits, (int)h_out, (int)w_out);
        rc = check_launch(who);
        if (rc != APGPU_OK) return rc;
    }
    StackParams prm{};
    prm.frames = args->frames;
    prm.mean = args->mean;
    prm.count = args->count;
    prm.moments = args->moments;
    prm.P = args->n_pixels;
    prm.stride = fstride;
    prm.sl2 = args->sigma_lower * args->sigma_lower;
    prm.su2 = args->sigma_upper * args->sigma_upper;
    prm.N = N;
    prm.center = args->center;
    prm.dev = args->dev;
    prm.maxiters = args->maxiters;
    prm.moments64 = args->moments_f64;
    prm.fast32 = (args->flags & APGPU_STACK_EXACT_MOMENTS) ? 0 : ((args->flags & APGPU_STACK_MOMENTS_MEAN) ? 2 : 1);
    FusedArgs fa{};
    fa.recs = recs;
    fa.lut = lut;
    fa.bits = bits;
    fa.mask = mask;
    fa.mask_ctl = mctl;
    fa.mask_cap = wl.list_cap;
    fa.gx = (int)gx;
    fa.gy = (int)gy;
    fa.log2_phases = log2_phases;
    fa.h_in = (int)h_in;
    fa.w_in = (int)w_in;
    fa.h_out = (int)h_out;
    fa.w_out = (int)w_out;
    fa.n_frames = N;
    if (N <= 4) launch_fused<4>(prm, fa, st);
    else if (N <= 8) launch_fused<8>(prm, fa, st);
    else if (N <= 12) launch_fused<12>(prm, fa, st);
    else launch_fused<16>(prm, fa, st);
    return check_launch(who);
}
