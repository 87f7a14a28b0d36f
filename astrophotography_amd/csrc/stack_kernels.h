// stack_kernels.h - per-pixel reduction of a frame slab [N][P] along N on gfx950 (MI355X).
//
// Replaces, for the N-frame stack, the arithmetic of
//   astropy.stats.sigma_clipped_stats(cube, axis=0)        (astropy/stats/sigma_clipping.py:298-383,
//                                                           924-937; C loop src/compute_bounds.c)
//   ccdproc.combine(sigma_clip=True, median/mad_std)       (reference call site
//                                                           scripts/ap_combine_darks.py:394-420)
// optionally fused with ApCalibrate.calibrate's per-value arithmetic (core/ApCalibrate.py:439-464)
// so that a raw slab is read from HBM exactly once.
//
// Data layout: frames[f][p], p fastest.  One lane owns one pixel: for every frame f a wavefront
// reads 64 consecutive pixels (a fully coalesced 256-byte segment for f32), so the N loads of a
// lane are N independent row streams.  The lane's column of N values stays in VGPRs for the whole
// kernel (N <= 128): it is sorted once with a Batcher odd-even merge network (static register
// indices only), after which every clipping iteration only trims the two ends of the sorted column:
//   survivors are always a contiguous range [a, b) of the sorted column,
//   S = sum(x - c), Q = sum((x - c)^2) over the range are kept in float64 relative to a pivot c
//   (the first median) and updated for the few trimmed elements only,
//   keep-tests are done without division or square root:
//        x >= cen - s_lo*std   <=>   not( w < 0 and w^2 > s_lo^2 * (n*Q - S^2) ),  w = n*(x - cen)
// The bound is algebraically the one astropy computes (cen +/- sigma*sqrt(sum((mean-x)^2)/n));
// it differs from astropy's float64 evaluation by a few ulp(float64), far below the float32 spacing of
// the data, so keep/reject decisions - and therefore the survivors - are identical.
//
// HBM traffic per output pixel (algorithmic): N*sizeof(raw) + 12 (bias, dark, nflat) read,
// 4 per requested output plane written.  No LDS, no cross-lane traffic: the kernel is a pure
// stream-in / reduce-in-registers / stream-out design, bounded by HBM when the VALU work hides.
#pragma once
#include "common.h"

#include <utility>

namespace apgpu_stack {

using namespace apgpu;

// zero-cost section markers in the generated assembly (tools/isa_sections.py counts instructions per section)
#ifdef APGPU_PROFILE_SECTIONS
#define APGPU_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; APGPU_SECTION " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define APGPU_MARK(name) asm volatile("; APGPU_SECTION " name)
#endif

// -------------------------------------------------------------------------------------------------
// Batcher odd-even merge sorting network, generated at compile time.  For NP that is not a power of two the
// network of the next power of two is pruned to its first NP wires: the missing inputs are +inf, which never
// move from the top wires, so every compare-exchange that touches one of them is a no-op and can be dropped.
// -------------------------------------------------------------------------------------------------
struct CE {
    unsigned char a, b;
};

template <int NP>
struct Net {
    CE ce[NP * 12 + 1]; // NP=128 needs 1471 < 1536
    int n;
};

constexpr int next_pow2(int n)
{
    int p = 1;
    while (p < n) p *= 2;
    return p;
}

template <int NP>
constexpr Net<NP> make_net()
{
    constexpr int P2 = next_pow2(NP);
    Net<NP> net{};
    int c = 0;
    for (int p = 1; p < P2; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P2 - 1 - k; j += 2 * k) {
                int lim = (k - 1 < P2 - j - k - 1) ? k - 1 : P2 - j - k - 1;
                for (int i = 0; i <= lim; i++)
                    if ((i + j) / (p * 2) == (i + j + k) / (p * 2) && i + j + k < NP) {
                        net.ce[c].a = (unsigned char)(i + j);
                        net.ce[c].b = (unsigned char)(i + j + k);
                        c++;
                    }
            }
    net.n = c;
    return net;
}

// Compare-exchange.  Written as the two machine instructions: through fminf/fmaxf the compiler has to
// quiet possible signalling NaNs first (IEEE mode) and adds a v_max_f32 x, x, x canonicalisation per
// network input (~120 instructions per column); the columns are NaN-free by construction here.
__device__ __forceinline__ void cmpx(float &x, float &y)
{
    float lo, hi;
    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    x = lo;
    y = hi;
}

template <int NP, int BASE, int... I>
__device__ __forceinline__ void net_chunk(float (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = make_net<NP>();
    (cmpx(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int NP, int BASE>
__device__ __forceinline__ void net_from(float (&v)[NP])
{
    constexpr int total = make_net<NP>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        net_chunk<NP, BASE>(v, std::make_integer_sequence<int, len>{});
        net_from<NP, BASE + len>(v);
    }
}

template <int NP>
__device__ __forceinline__ void sort_column(float (&v)[NP])
{
    if constexpr (NP > 1) net_from<NP, 0>(v);
}

// v[LO + rel] for a per-lane rel in [0, LEN): binary multiplexer tree (LEN-1 v_cndmask), static
// register indices only (a runtime-indexed register array would be demoted to scratch memory).
template <int LO, int LEN, int NP>
__device__ __forceinline__ float pick_rel(const float (&v)[NP], int rel)
{
    if constexpr (LEN == 1) {
        return v[LO];
    } else if constexpr ((LEN & (LEN - 1)) == 0) {
        constexpr int H = LEN / 2;
        float lo = pick_rel<LO, H, NP>(v, rel);
        float hi = pick_rel<LO + H, H, NP>(v, rel);
        return (rel & H) ? hi : lo;
    } else {
        constexpr int H = next_pow2(LEN) / 2;               // 48 = 32 + 16, 96 = 64 + 32, 24 = 16 + 8, 12 = 8 + 4
        float lo = pick_rel<LO, H, NP>(v, rel);
        float hi = pick_rel<LO + H, LEN - H, NP>(v, rel - H);
        return (rel >= H) ? hi : lo;
    }
}

template <int NP>
__device__ __forceinline__ float pick_at(const float (&v)[NP], int idx)
{
    idx = idx < 0 ? 0 : (idx > NP - 1 ? NP - 1 : idx);
    return pick_rel<0, NP, NP>(v, idx);
}

// The two middle elements v[i1], v[i2] (i2 = i1 or i1 + 1) of the survivor range.  The clip trims a
// few values off either end, so the middle stays within a few slots of NP/2: if every lane of the
// wave is inside the 8-slot window around NP/2 the multiplexer needs 2 x 7 selects instead of
// 2 x (NP-1); otherwise the whole wave takes the full tree.
// 8-slot window [4K, 4K + 8) chosen at run time by a wave-uniform K (static register indices per case).
template <int K, int NP>
__device__ __forceinline__ void pick_window(const float (&v)[NP], int k, int i1, int i2, float &m1, float &m2)
{
    if (k == K) {
        m1 = pick_rel<4 * K, 8, NP>(v, i1 - 4 * K);
        m2 = pick_rel<4 * K, 8, NP>(v, i2 - 4 * K);
    } else if constexpr (4 * (K + 1) + 8 <= NP) {
        pick_window<K + 1, NP>(v, k, i1, i2, m1, m2);
    }
}

template <int NP>
__device__ __forceinline__ void pick_middle(const float (&v)[NP], int i1, int i2, float &m1, float &m2)
{
    if constexpr (NP <= 8) {
        m1 = pick_at<NP>(v, i1);
        m2 = pick_at<NP>(v, i2);
    } else {
        constexpr int WLO = NP / 2 - 4;
        const bool inside = (i1 >= WLO) && (i2 < WLO + 8);
        if (__all(inside)) {
            m1 = pick_rel<WLO, 8, NP>(v, i1 - WLO);
            m2 = pick_rel<WLO, 8, NP>(v, i2 - WLO);
        } else {
            // stacks with fewer frames than slots (or many rejected values) have their middle elsewhere:
            // take the 8-slot window around the first lane's middle if it holds every lane's
            int k = (__builtin_amdgcn_readfirstlane(i1) - 2) >> 2;
            k = k < 0 ? 0 : (k > NP / 4 - 2 ? NP / 4 - 2 : k);
            const bool inside_k = (i1 >= 4 * k) && (i2 < 4 * k + 8);
            if (__all(inside_k)) {
                pick_window<0, NP>(v, k, i1, i2, m1, m2);
            } else {
                m1 = pick_at<NP>(v, i1);
                m2 = pick_at<NP>(v, i2);
            }
        }
    }
}

// Per-lane state of the clipping loop.  Survivors are v[a .. b) of the sorted column.
struct ClipState {
    double S, Q;            // sum(x - c), sum((x - c)^2) over the survivors
    double c;               // pivot
    double cen, nn;         // centre and count the last bounds were computed with
    double wscale;          // scale of the bound test: n (std mode, T = sigma^2 n^2 var) or 1 (mad_std mode)
    double Tlo, Thi;        // sigma^2 * (n*Q - S^2): squared, n^2-scaled half-widths of the bounds
    int a, b;
};

__device__ __forceinline__ bool below(const ClipState &st, double xd)
{
    const double w = st.wscale * (xd - st.cen);
    return (w < 0.0) && (w * w > st.Tlo);
}

__device__ __forceinline__ bool above(const ClipState &st, double xd)
{
    const double w = st.wscale * (xd - st.cen);
    return (w > 0.0) && (w * w > st.Thi);
}

// Moves a wave-uniform value into VGPRs.  The kernel arguments arrive in 40 SGPRs; whatever stays live
// across the clipping loop is spilled to VGPR lanes and re-read (v_readlane) on every iteration, so the
// handful of values needed inside / after the loop are parked in VGPRs once instead.
template <typename T>
__device__ __forceinline__ T park_in_vgpr(T x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// float -> double of a column element, opaque to the optimiser: without the barrier LLVM hoists and
// CSEs the 64 conversions out of the clipping loop and keeps 64 doubles (128 VGPRs) live.
__device__ __forceinline__ double widen(float x)
{
    asm volatile("" : "+v"(x));
    return (double)x;
}

// Trim rejected values from the low end: element I, then (only if some lane still has its cut
// above I) element I+1, ...  Static recursion keeps every register index a compile-time constant.
template <int I, int NP>
__device__ __forceinline__ void trim_low(const float (&v)[NP], ClipState &st, bool active)
{
    if constexpr (I < NP) {
        const double xd = widen(v[I]);
        const bool rej = active && (I >= st.a) && (I < st.b) && below(st, xd);
        if (rej) {
            const double d = xd - st.c;
            st.S -= d;
            st.Q = fma(-d, d, st.Q);
            st.a = I + 1;
        }
        if (__any(active && (st.a > I))) trim_low<I + 1, NP>(v, st, active);
    }
}

template <int I, int NP>
__device__ __forceinline__ void trim_high(const float (&v)[NP], ClipState &st, bool active)
{
    if constexpr (I >= 0) {
        if (__any(active && (I < st.b))) {                  // padding slots above every lane's range: just step down
            const double xd = widen(v[I]);
            const bool rej = active && (I >= st.a) && (I < st.b) && above(st, xd);
            if (rej) {
                const double d = xd - st.c;
                st.S -= d;
                st.Q = fma(-d, d, st.Q);
                st.b = I;
            }
        }
        if (__any(active && (st.b <= I))) trim_high<I - 1, NP>(v, st, active);
    }
}

template <int I, int NP>
__device__ __forceinline__ void readmit_low(const float (&v)[NP], ClipState &st, int &a_new)
{
    if constexpr (I < NP) {
        if (__any(I < st.a)) {
            const double xd = widen(v[I]);
            const bool keep = (I < st.a) && !below(st, xd) && !above(st, xd);
            if (keep) {
                const double d = xd - st.c;
                st.S += d;
                st.Q = fma(d, d, st.Q);
                a_new = a_new < I ? a_new : I;
            }
            readmit_low<I + 1, NP>(v, st, a_new);
        }
    }
}

template <int I, int NP>
__device__ __forceinline__ void readmit_high(const float (&v)[NP], ClipState &st, int n, int &b_new)
{
    if constexpr (I >= 0) {
        if (__any(I >= st.b)) {
            const double xd = widen(v[I]);
            const bool keep = (I >= st.b) && (I < n) && !below(st, xd) && !above(st, xd);
            if (keep) {
                const double d = xd - st.c;
                st.S += d;
                st.Q = fma(d, d, st.Q);
                b_new = b_new > I + 1 ? b_new : I + 1;
            }
            readmit_high<I - 1, NP>(v, st, n, b_new);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// "Rich" kernels (median / std output planes, mad_std deviation): after the sort the column is parked in
// LDS - row i holds element i of every lane's column - so that a lane reads ITS column with a run-time
// index (one ds_read_b32, bank = lane: conflict-free) where the lean kernel needs an (NP-1)-select
// multiplexer tree, and every later phase is a compact run-time loop over LDS instead of NP levels of
// statically indexed code.  64 KB per workgroup (256 lanes x 64 rows, or 128 lanes x 128 rows).
// -------------------------------------------------------------------------------------------------
template <int NP>
constexpr int rich_block() { return NP > 64 ? 128 : 256; }

template <int NP, bool RICH>
struct ColumnLds {
    __device__ __forceinline__ float *lane_ptr(int) { return nullptr; }
};
template <int NP>
struct ColumnLds<NP, true> {
    float x[NP][rich_block<NP>()];
    __device__ __forceinline__ float *lane_ptr(int lane) { return &x[0][lane]; }
};

template <int NP>
__device__ __forceinline__ float col_read(const float *col, int i)
{
    i = i < 0 ? 0 : (i > NP - 1 ? NP - 1 : i);
    return col[i * rich_block<NP>()];
}

// astropy.stats.mad_std of the survivors x[a .. b): 1.482602218505602 * median(|x - med|)
// (astropy/stats/funcs.py:844-850, 917-920; the C loop's mad_buffer).  No second sort: the column is
// sorted, so the j+1 deviations nearest to med belong to a contiguous window [L, L+j], and the j-th order
// statistic of the deviations is  min over L of max(|x_L - med|, |x_(L+j) - med|)  (|x - med| is convex
// along the sorted column, so a window's largest deviation sits at one of its ends).  Windows leaving
// [a, b) get an infinite deviation.  Every value is the exact float64 |x - med| the reference sorts.
template <int NP>
__device__ __forceinline__ double mad_std_window(const float *col, bool active, int a, int b, double med)
{
    const int n = b - a;
    const int k1 = n > 0 ? (n - 1) >> 1 : 0;
    const bool even = (n & 1) == 0;
    const int bk = b - k1;                                  // L + k1 < b  <=>  L < bk
    const float inf = __builtin_inff();
    double m1 = __builtin_inf(), m2 = __builtin_inf();
    double dl_prev = __builtin_inf();
    constexpr int CH = NP >= 4 ? 4 : NP;                    // windows per trip: 2*CH LDS reads in flight
    for (int L0 = 0; L0 < NP; L0 += CH) {
        if (!__any(active && L0 + CH > a && L0 < bk)) {     // no lane has a window starting in this chunk
            dl_prev = __builtin_inf();
            continue;
        }
        float xl[CH], xr[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            xl[j] = col_read<NP>(col, L0 + j);
            xr[j] = col_read<NP>(col, L0 + j + k1);
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int L = L0 + j;
            const double dl = fabs((double)((L >= a) ? xl[j] : inf) - med);
            const double dr = fabs((double)((L < bk) ? xr[j] : inf) - med);
            m1 = fmin(m1, fmax(dl, dr));                    // window [L, L + k1]
            m2 = fmin(m2, fmax(dl_prev, dr));               // window [L - 1, L + k1]
            dl_prev = dl;
        }
    }
    const double x1 = m1, x2 = even ? m2 : m1;
    return (0.5 * (x1 + x2)) * 1.482602218505602;
}

struct StackParams {
    const void *frames;
    const float *bias, *dark, *nflat, *exp_ratio, *pedestal;
    const uint8_t *pixmask;
    float *mean, *median, *std, *moments;
    int32_t *count;
    int64_t P;
    int64_t stride;         // elements between frames
    double sl2, su2;        // sigma_lower^2, sigma_upper^2
    int N;
    int still_biased;
    int center;             // 0 median, 1 mean
    int dev;                // 0 std, 1 mad_std (EXTRA kernels only)
    int maxiters;           // < 0: until convergence
    int persistent;         // use the persistent, load/compute-overlapped kernel where available
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(uint16_t x) { return (float)x; }

// x / nf for a per-pixel divisor with y = RN(1 / nf) precomputed: two Newton steps on the quotient with
// exact FMA residuals (Markstein): q0 = RN(x*y) is within 1.5 ulp, q1 is faithful, q2 = RN(x / nf)
// provided nothing over/underflows - the caller guards the ranges and falls back to IEEE division.
// 5 instructions instead of the 12 of the IEEE sequence (v_div_scale x2, v_rcp, 6 FMA, v_div_fmas,
// v_div_fixup), 64 times per pixel.
__device__ __forceinline__ float div_by_recip(float x, float nf, float y)
{
    const float q0 = x * y;
    const float r0 = __builtin_fmaf(-nf, q0, x);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-nf, q1, x);
    return __builtin_fmaf(r1, y, q1);
}

// Per-frame scalars (exposure ratio, pedestal) staged in LDS once per workgroup: as SGPR values the
// 2*NP scalars exceed the 102-SGPR budget and get spilled to VGPR lanes; from LDS they arrive as
// broadcast ds_read_b128 (4 frames per instruction) just before use.
template <int NP>
struct FrameScalars {
    float e[NP];
    float ped[NP];
    float pad[NP];          // -inf for a real frame, +inf for a padding slot (f >= N): v = max(v, pad) pads a column
};

template <int NP>
__device__ __forceinline__ void stage_frame_scalars(const StackParams &prm, FrameScalars<NP> &fs)
{
    for (int t = threadIdx.x; t < NP; t += blockDim.x) {
        const int ff = t < prm.N ? t : prm.N - 1;
        fs.e[t] = prm.exp_ratio ? prm.exp_ratio[ff] : 0.f;
        fs.ped[t] = prm.pedestal ? prm.pedestal[ff] : 0.f;
        fs.pad[t] = t < prm.N ? -__builtin_inff() : __builtin_inff();
    }
    __syncthreads();
}

template <int NP, typename RawT, bool FULL>
__device__ __forceinline__ void load_raw(const StackParams &prm, int64_t base, int lane, RawT (&raw)[NP])
{
    // Wave-uniform frame pointer (SGPR pair) + per-lane offset: one coalesced row segment per frame.
    const RawT *fb = static_cast<const RawT *>(prm.frames) + base;
    // opaque per call: the fast path and its (rare) exact fallback each load the column; sharing the NP
    // clamped address steps between the two calls would keep 2*NP SGPRs live across the calibration
    int nframes = prm.N;
    if constexpr (!FULL) asm volatile("" : "+s"(nframes));
#pragma unroll
    for (int f = 0; f < NP; f++) {
        raw[f] = fb[lane];
        if (FULL || f + 1 < nframes) fb += prm.stride;  // padded slots re-read the last frame (cache hit)
        // fence: otherwise the scheduler materialises all NP frame addresses (2 SGPRs each) at once
        if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

// Fast calibration of a full column (N == NP): reciprocal division, no per-value fix-ups.  Returns
// true if the lane's results are exact AND all finite; otherwise the wave redoes the column exactly.
// Frames are processed in pairs with 2-wide vector arithmetic so that the backend emits the packed
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 forms: the kernel is bound by VALU issue slots (one
// wave64 instruction per 4 cycles per SIMD) and a packed instruction retires two values per slot.
// Each lane of a packed operation is an ordinary IEEE float32 operation, so results do not change.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NP, typename RawT, bool HAS_PED>
__device__ __forceinline__ bool calibrate_fast(const FrameScalars<NP> &fs, const RawT (&raw)[NP], float b, float D, float nf,
                                               bool dodiv, float (&v)[NP])
{
    // lanes that do not divide (no flat / nflat == 0) run the same code with a divisor of exactly 1:
    // q0 = x, r0 = 0, ... -> x, bit for bit; no per-value select
    const float nfe = dodiv ? nf : 1.0f;
    const float y = __fdiv_rn(1.0f, nfe);
    const float anf = fabsf(nfe);
    const bool nf_ok = (anf >= 0x1p-40f && anf <= 0x1p40f);
    float mx = 0.f, mn = __builtin_inff();
    if constexpr (NP >= 2) {
        v2f acc = {0.f, 0.f};
        const v2f b2 = {b, b}, D2 = {D, D}, nf2 = {-nfe, -nfe}, y2 = {y, y}, zero2 = {0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NP; f += 2) {
            v2f x = {to_f32(raw[f]), to_f32(raw[f + 1])};
            if constexpr (HAS_PED) {
                const v2f ped = {fs.ped[f], fs.ped[f + 1]};
                x = x + ped;                                 // ApCalibrate.py:318-326; a zero pedestal adds +0.0,
            }                                                // which changes nothing but the sign of a -0.0 input
            const v2f e2 = {fs.e[f], fs.e[f + 1]};
            x = x - b2;                                      // :439
            const v2f ds = e2 * D2;                          // :450
            x = x - ds;                                      // :451
            const v2f q0 = x * y2;                           // :462-464 via reciprocal + 2 FMA corrections
            const v2f r0 = __builtin_elementwise_fma(nf2, q0, x);
            const v2f q1 = __builtin_elementwise_fma(r0, y2, q0);
            const v2f r1 = __builtin_elementwise_fma(nf2, q1, x);
            const v2f q = __builtin_elementwise_fma(r1, y2, q1);
            v[f] = q.x;
            v[f + 1] = q.y;
            acc = __builtin_elementwise_fma(q, zero2, acc);  // NaN iff some value is not finite
            mx = fmaxf(fmaxf(mx, fabsf(q.x)), fabsf(q.y));
            mn = fminf(fminf(mn, fabsf(q.x)), fabsf(q.y));
        }
        const bool range_ok = !dodiv || (mx < 0x1p50f && mn > 0x1p-50f);
        return nf_ok && range_ok && (acc.x == 0.f) && (acc.y == 0.f);
    } else {
        float x = to_f32(raw[0]);
        if constexpr (HAS_PED) x = x + fs.ped[0];
        x = x - b;
        const float ds = fs.e[0] * D;
        x = x - ds;
        const float q = div_by_recip(x, nfe, y);
        v[0] = q;
        const float acc = __builtin_fmaf(q, 0.0f, 0.0f);
        const bool range_ok = !dodiv || (fabsf(q) < 0x1p50f && fabsf(q) > 0x1p-50f);
        return nf_ok && range_ok && (acc == 0.f);
    }
}

// Loads the lane's column, applies the fused calibration, maps non-finite values (sigma clip) or
// NaNs (plain median) to the +inf sentinel and returns the number of valid values.
// FULL = the stack has exactly NP frames: no padding logic at all (no clamped frame indices, no
// wave-wide (f < N) masks - NP of those cost 2 SGPRs each and end up spilled to VGPR lanes).
template <int NP, typename RawT, bool CALIB, bool FINITE_ONLY, bool FULL>
__device__ __forceinline__ int load_column(const StackParams &prm, const FrameScalars<NP> &fs, int64_t base, int lane,
                                           float (&v)[NP])
{
    const int N = prm.N;
    const int64_t p = base + lane;
    RawT raw[NP];
    load_raw<NP, RawT, FULL>(prm, base, lane, raw);
    float b = 0.f, D = 0.f, nf = 1.f;
    bool dodiv = false;
    if constexpr (CALIB) {
        b = prm.bias[p];
        const float d = prm.dark[p];
        D = prm.still_biased ? d - b : d;                    // ApCalibrate.py:440-445
        if (prm.nflat) {
            nf = prm.nflat[p];
            dodiv = (nf != 0.f);                             // ApCalibrate.py:462 (NaN != 0 is True)
        }
    }
    const bool skip = prm.pixmask && prm.pixmask[p];
    if constexpr (CALIB) {
        bool good;
        if (prm.pedestal) good = calibrate_fast<NP, RawT, true>(fs, raw, b, D, nf, dodiv, v);
        else good = calibrate_fast<NP, RawT, false>(fs, raw, b, D, nf, dodiv, v);
        if (__all(good && !skip)) {
            if constexpr (FULL) return NP;
            // padding slots hold a calibrated copy of the last frame: lift them to the +inf sentinel with
            // one v_max against the staged pad vector (no NP wave-wide (f < N) masks)
#pragma unroll
            for (int f = 0; f < NP; f++) asm("v_max_f32 %0, %1, %2" : "=v"(v[f]) : "v"(v[f]), "v"(fs.pad[f]));
            return N;
        }
        // rare: a non-finite value, a masked pixel or an out-of-range operand somewhere in the wave:
        // redo the column exactly (IEEE division), one frame at a time - no second raw[] column in flight
    }
    int n = 0;
    const RawT *fp = static_cast<const RawT *>(prm.frames) + p;
    int nleft = N;
    asm volatile("" : "+s"(nleft));
#pragma unroll
    for (int f = 0; f < NP; f++) {
        float x;
        if constexpr (CALIB) {
            x = to_f32(*fp);
            if (FULL || f + 1 < nleft) fp += prm.stride;
        } else {
            x = to_f32(raw[f]);
        }
        if constexpr (CALIB) {
            const float e = fs.e[f];
            const float ped = fs.ped[f];
            if (ped != 0.f) x = x + ped;                     // ApCalibrate.py:318-326
            x = x - b;                                       // :439
            const float ds = e * D;                          // :450
            x = x - ds;                                      // :451
            if (dodiv) x = __fdiv_rn(x, nf);                 // :463
        }
        bool ok;
        if constexpr (FINITE_ONLY) ok = fabsf(x) < __builtin_inff();
        else ok = (x == x);
        ok = ok && (FULL || f < N) && !skip;
        n += ok ? 1 : 0;
        v[f] = ok ? x : __builtin_inff();
    }
    return n;
}

// Lean reduction (mean / count / moments outputs, std deviation): the benchmarked path.  Everything after
// the column load is in registers: sort, moments, clipping iterations, outputs.
template <int NP, bool PRESORTED = false>
__device__ __forceinline__ void reduce_and_store(const StackParams &prm, float (&v)[NP], const int n, const int64_t p)
{
    // everything the loop and the epilogue need from the kernel arguments, parked before the sort
    float *const out_mean = park_in_vgpr(prm.mean);
    int32_t *const out_count = park_in_vgpr(prm.count);
    float *const out_moments = park_in_vgpr(prm.moments);
    const int64_t Pn = park_in_vgpr(prm.P);
    const double sl2 = park_in_vgpr(prm.sl2), su2 = park_in_vgpr(prm.su2);
    const int maxiters = park_in_vgpr(prm.maxiters);
    const bool use_median = park_in_vgpr((int)prm.center) == APGPU_CENTER_MEDIAN;
    APGPU_MARK("sort");
    if constexpr (!PRESORTED) sort_column<NP>(v);           // PRESORTED: ascending, sentinels last (uint16 pair kernel)
    APGPU_MARK("moments");

    // pivot: the lower median of the finite values
    float cf, cf2;
    pick_middle<NP>(v, (n - 1) >> 1, (n - 1) >> 1, cf, cf2);
    cf = n > 0 ? cf : 0.f;
    const double c = (double)cf;
    // S = sum(x - c), Q = sum((x - c)^2): four independent float64 chains (ILP), fixed association
    double Sa[4] = {0.0, 0.0, 0.0, 0.0}, Qa[4] = {0.0, 0.0, 0.0, 0.0};
    if (__all(n == NP)) {               // the usual case: no padding, no rejected value in the whole wave
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const double d = (double)v[i] - c;
            Sa[i & 3] += d;
            Qa[i & 3] = fma(d, d, Qa[i & 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const float x = (i < n) ? v[i] : cf;
            const double d = widen(x) - c;      // opaque: keeps this rare path from sharing (and hoisting)
            Sa[i & 3] += d;                     // the 64 conversions of the common path above
            Qa[i & 3] = fma(d, d, Qa[i & 3]);
        }
    }
    const double S0 = (Sa[0] + Sa[1]) + (Sa[2] + Sa[3]);
    const double Q0 = (Qa[0] + Qa[1]) + (Qa[2] + Qa[3]);

    APGPU_MARK("clip_loop");
    ClipState st;
    st.S = S0;
    st.Q = Q0;
    st.c = c;
    st.a = 0;
    st.b = n;
    // parameters of the last bounds computed for this lane
    st.cen = c;
    st.nn = (double)n;
    st.wscale = (double)n;
    st.Tlo = 0.0;
    st.Thi = 0.0;
    bool active = n > 0;
    int it = 0;

    while (__any(active)) {
        const int a0 = st.a, b0 = st.b;
        float m1 = 0.f, m2 = 0.f;
        if (use_median) pick_middle<NP>(v, (st.a + st.b - 1) >> 1, (st.a + st.b) >> 1, m1, m2);
        const double med = 0.5 * ((double)m1 + (double)m2);  // wirth_median (even: mean of the two)
        if (active) {
            st.nn = (double)(st.b - st.a);
            st.cen = use_median ? med : c + st.S / st.nn;
            st.wscale = st.nn;
            double V = fma(st.nn, st.Q, -(st.S * st.S));     // n^2 * variance
            V = V > 0.0 ? V : 0.0;
            st.Tlo = sl2 * V;
            st.Thi = su2 * V;
        }
        trim_low<0, NP>(v, st, active);
        trim_high<NP - 1, NP>(v, st, active);
        it++;
        const bool changed = (st.a != a0) || (st.b != b0);
        active = active && changed && (maxiters < 0 || it < maxiters);
    }

    APGPU_MARK("readmit_output");
    // astropy applies the FINAL bounds to all values (sigma_clipping.py:356-358): values trimmed by
    // an earlier, tighter pass that lie inside the final bounds are re-admitted.
    if (__any(st.a > 0)) {
        int a_new = st.a;
        readmit_low<0, NP>(v, st, a_new);
        st.a = a_new;
    }
    if (__any(st.b < n)) {
        int b_new = st.b;
        readmit_high<NP - 1, NP>(v, st, n, b_new);
        st.b = b_new;
    }
    const int a = st.a, b = st.b;
    const double S = st.S, Q = st.Q;

    const int cnt = b - a;
    const double nf = (double)cnt;
    const double nan = __builtin_nan("");
    const double ms = S / nf;                                 // mean - c
    if (out_mean) out_mean[p] = cnt > 0 ? (float)(c + ms) : (float)nan;
    if (out_count) out_count[p] = cnt;
    if (out_moments) {
        const double sum = cnt > 0 ? fma(nf, c, S) : 0.0;
        const double sq = cnt > 0 ? Q + 2.0 * c * S + nf * c * c : 0.0;
        out_moments[p] = (float)sum;                      // plane order: sum, count, sum of squares - the first
        out_moments[Pn + p] = (float)cnt;                  // two are all a mean needs, so an N-shard exchange that
        out_moments[2 * Pn + p] = (float)sq;               // does not want std all-reduces a contiguous [2][P] prefix
    }
}

// Rich reduction: the lean algorithm with (a) mad_std as an alternative deviation, (b) the median and
// std output planes, (c) the sorted column in LDS (see above).  Arithmetic on S / Q is performed in the
// same order as in the lean kernel, so both produce identical mean / count / moments.
template <int NP>
__device__ __forceinline__ void reduce_and_store_rich(const StackParams &prm, float (&v)[NP], const int n, const int64_t p,
                                                      float *const col)
{
    constexpr int B = rich_block<NP>();
    const bool use_median = prm.center == APGPU_CENTER_MEDIAN;
    const bool use_mad = prm.dev == APGPU_DEV_MAD_STD;
    const double sl2 = prm.sl2, su2 = prm.su2;
    const int maxiters = prm.maxiters;
    sort_column<NP>(v);
#pragma unroll
    for (int i = 0; i < NP; i++) col[i * B] = v[i];

    // pivot: the lower median of the finite values; S = sum(x - c), Q = sum((x - c)^2) as in the lean kernel
    const float cf = n > 0 ? col_read<NP>(col, (n - 1) >> 1) : 0.f;
    const double c = (double)cf;
    double Sa[4] = {0.0, 0.0, 0.0, 0.0}, Qa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const float x = (i < n) ? v[i] : cf;
        const double d = (double)x - c;
        Sa[i & 3] += d;
        Qa[i & 3] = fma(d, d, Qa[i & 3]);
    }
    ClipState st;
    st.S = (Sa[0] + Sa[1]) + (Sa[2] + Sa[3]);
    st.Q = (Qa[0] + Qa[1]) + (Qa[2] + Qa[3]);
    st.c = c;
    st.a = 0;
    st.b = n;
    st.cen = c;
    st.nn = (double)n;
    st.wscale = (double)n;
    st.Tlo = 0.0;
    st.Thi = 0.0;
    bool active = n > 0;
    int it = 0;

    while (__any(active)) {
        const int a0 = st.a, b0 = st.b;
        const float m1 = col_read<NP>(col, (st.a + st.b - 1) >> 1);
        const float m2 = col_read<NP>(col, (st.a + st.b) >> 1);
        const double med = 0.5 * ((double)m1 + (double)m2);  // wirth_median (even: mean of the two)
        double mad = 0.0;
        if (use_mad) mad = mad_std_window<NP>(col, active, st.a, st.b, med);
        if (active) {
            st.nn = (double)(st.b - st.a);
            st.cen = use_median ? med : c + st.S / st.nn;
            if (use_mad) {
                st.wscale = 1.0;
                st.Tlo = sl2 * (mad * mad);
                st.Thi = su2 * (mad * mad);
            } else {
                st.wscale = st.nn;
                double V = fma(st.nn, st.Q, -(st.S * st.S)); // n^2 * variance
                V = V > 0.0 ? V : 0.0;
                st.Tlo = sl2 * V;
                st.Thi = su2 * V;
            }
        }
        // trim from the low end, then from the high end: every lane walks its own cursor
        for (;;) {
            const double xd = (double)col_read<NP>(col, st.a);
            const bool rej = active && (st.a < st.b) && below(st, xd);
            if (rej) {
                const double d = xd - st.c;
                st.S -= d;
                st.Q = fma(-d, d, st.Q);
                st.a++;
            }
            if (!__any(rej)) break;
        }
        for (;;) {
            const double xd = (double)col_read<NP>(col, st.b - 1);
            const bool rej = active && (st.a < st.b) && above(st, xd);
            if (rej) {
                const double d = xd - st.c;
                st.S -= d;
                st.Q = fma(-d, d, st.Q);
                st.b--;
            }
            if (!__any(rej)) break;
        }
        it++;
        const bool changed = (st.a != a0) || (st.b != b0);
        active = active && changed && (maxiters < 0 || it < maxiters);
    }

    // astropy applies the FINAL bounds to all values (sigma_clipping.py:356-358): values trimmed by an
    // earlier, tighter pass that lie inside the final bounds are re-admitted (ascending, then descending,
    // like the lean kernel's chains).
    if (__any(st.a > 0)) {
        int a_new = st.a;
        for (int i = 0; __any(i < st.a); i++) {
            const double xd = (double)col_read<NP>(col, i);
            const bool keep = (i < st.a) && !below(st, xd) && !above(st, xd);
            if (keep) {
                const double d = xd - st.c;
                st.S += d;
                st.Q = fma(d, d, st.Q);
                a_new = a_new < i ? a_new : i;
            }
        }
        st.a = a_new;
    }
    if (__any(st.b < n)) {
        int b_new = st.b;
        for (int i = NP - 1; __any(i >= st.b); i--) {
            const double xd = (double)col_read<NP>(col, i);
            const bool keep = (i >= st.b) && (i < n) && !below(st, xd) && !above(st, xd);
            if (keep) {
                const double d = xd - st.c;
                st.S += d;
                st.Q = fma(d, d, st.Q);
                b_new = b_new > i + 1 ? b_new : i + 1;
            }
        }
        st.b = b_new;
    }
    const int a = st.a, b = st.b;
    const double S = st.S, Q = st.Q;
    const int cnt = b - a;
    const double nf = (double)cnt;
    const double nan = __builtin_nan("");
    const double ms = S / nf;                                 // mean - c
    if (prm.mean) prm.mean[p] = cnt > 0 ? (float)(c + ms) : (float)nan;
    if (prm.count) prm.count[p] = cnt;
    if (prm.std) {
        // np.nanstd of the survivors: two passes like numpy (a column of identical survivors must give
        // exactly 0, which the running S/Q - updated by subtraction - cannot guarantee).
        constexpr int CH = NP >= 8 ? 8 : NP;                  // LDS reads in flight per trip
        double s1 = 0.0;
        for (int i0 = 0; i0 < NP; i0 += CH) {
            if (!__any(i0 + CH > a && i0 < b)) continue;
            float x[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) x[j] = col_read<NP>(col, i0 + j);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const bool in = (i0 + j >= a && i0 + j < b);
                s1 += (double)(in ? x[j] : cf) - c;          // a rejected slot contributes exactly 0
            }
        }
        const double m1 = s1 / nf;
        double q1 = 0.0;
        for (int i0 = 0; i0 < NP; i0 += CH) {
            if (!__any(i0 + CH > a && i0 < b)) continue;
            float x[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) x[j] = col_read<NP>(col, i0 + j);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const bool in = (i0 + j >= a && i0 + j < b);
                const double dd = ((double)x[j] - c) - m1;
                const double d = in ? dd : 0.0;
                q1 = fma(d, d, q1);
            }
        }
        prm.std[p] = cnt > 0 ? (float)sqrt(q1 > 0.0 ? q1 / nf : 0.0) : (float)nan;
    }
    if (prm.median) {
        const float m1 = col_read<NP>(col, (a + b - 1) >> 1);
        const float m2 = col_read<NP>(col, (a + b) >> 1);
        prm.median[p] = cnt > 0 ? (float)(((double)m1 + (double)m2) / 2.0) : (float)nan;
    }
    if (prm.moments) {
        const double sum = cnt > 0 ? fma(nf, c, S) : 0.0;
        const double sq = cnt > 0 ? Q + 2.0 * c * S + nf * c * c : 0.0;
        prm.moments[p] = (float)sum;
        prm.moments[prm.P + p] = (float)cnt;
        prm.moments[2 * prm.P + p] = (float)sq;
    }
}

template <int NP, typename RawT, bool CALIB, bool EXTRA, bool FULL>
__global__ __launch_bounds__(EXTRA ? rich_block<NP>() : 256, NP <= 64 ? 2 : 1) void stack_sigclip_kernel(const StackParams prm)
{
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int lane = threadIdx.x;
    const int64_t p = base + lane;
    __shared__ FrameScalars<NP> fs;
    __shared__ ColumnLds<NP, EXTRA> cols;        // lane-private columns: no barrier around their use
    if constexpr (CALIB || !FULL) stage_frame_scalars<NP>(prm, fs);
    if (p >= prm.P) return;

    float v[NP];
    APGPU_MARK("load_calibrate");
    const int n = load_column<NP, RawT, CALIB, true, FULL>(prm, fs, base, lane, v);
    if constexpr (EXTRA) reduce_and_store_rich<NP>(prm, v, n, p, cols.lane_ptr(lane));
    else reduce_and_store<NP>(prm, v, n, p);
}

// Persistent form of the lean kernel for full stacks (N == NP) with fused calibration: every
// workgroup walks over tiles of 256 pixels; as soon as the calibrated column of tile t is in v[] the
// loads of tile t + gridDim.x are issued into the (now dead) raw registers, so the HBM-bound load
// phase of the next tile runs underneath the VALU-bound sort / clip of the current one.  Without this
// the two phases of co-resident workgroups stay in lockstep (identical work) and add up.
// Register budget: raw[NP] (in flight) + v[NP] + ~50 temporaries -> 2 waves per SIMD, which is enough
// for a kernel bound by VALU issue (one wave64 instruction per 4 cycles per SIMD).
template <int NP, typename RawT>
__global__ __launch_bounds__(256, 2) void stack_sigclip_persistent_kernel(const StackParams prm)
{
    __shared__ FrameScalars<NP> fs;
    stage_frame_scalars<NP>(prm, fs);
    const int lane = threadIdx.x;
    const int64_t ntiles = (prm.P + 255) / 256;
    int64_t tile = blockIdx.x;
    RawT raw[NP];
    float b = 0.f, d = 0.f, nf = 1.f;
    const bool has_flat = prm.nflat != nullptr;
    auto issue = [&](int64_t t) {
        const int64_t base = t * 256;
        int64_t pc = base + lane;
        pc = pc < prm.P ? pc : prm.P - 1;                   // lanes past the end re-read the last pixel
        b = prm.bias[pc];
        d = prm.dark[pc];
        if (has_flat) nf = prm.nflat[pc];
        const RawT *fp = static_cast<const RawT *>(prm.frames) + pc;
#pragma unroll
        for (int f = 0; f < NP; f++) {
            raw[f] = *fp;
            fp += prm.stride;
            if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (tile < ntiles) issue(tile);
    while (tile < ntiles) {
        const int64_t base = tile * 256;
        const int64_t p = base + lane;
        const bool valid = p < prm.P;
        const int64_t pc = valid ? p : prm.P - 1;
        float v[NP];
        APGPU_MARK("p_calibrate");
        const float D = prm.still_biased ? d - b : d;       // ApCalibrate.py:440-445
        const bool dodiv = has_flat && (nf != 0.f);         // ApCalibrate.py:462 (NaN != 0 is True)
        const bool skip = prm.pixmask && prm.pixmask[pc];
        const float cb = b, cnf = nf;
        bool good;
        if (prm.pedestal) good = calibrate_fast<NP, RawT, true>(fs, raw, cb, D, cnf, dodiv, v);
        else good = calibrate_fast<NP, RawT, false>(fs, raw, cb, D, cnf, dodiv, v);
        // raw[] is dead now: start the next tile's loads, they complete under the reduction below
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) issue(next);
        int n = NP;
        if (!__all(good && !skip)) {
            // rare: a non-finite value, a masked pixel or an out-of-range operand somewhere in the wave:
            // redo the column exactly (IEEE division), one frame at a time
            n = 0;
            const RawT *fp = static_cast<const RawT *>(prm.frames) + pc;
#pragma unroll
            for (int f = 0; f < NP; f++) {
                float x = to_f32(*fp);
                fp += prm.stride;
                const float ped = fs.ped[f];
                if (ped != 0.f) x = x + ped;
                x = x - cb;
                const float ds = fs.e[f] * D;
                x = x - ds;
                if (dodiv) x = __fdiv_rn(x, cnf);
                const bool ok = (fabsf(x) < __builtin_inff()) && !skip;
                n += ok ? 1 : 0;
                v[f] = ok ? x : __builtin_inff();
            }
        }
        if (valid) reduce_and_store<NP>(prm, v, n, p);
        tile = next;
    }
}

// np.nanmedian(axis=0): NaNs dropped, +/-inf are ordinary values.
template <int NP, typename RawT, bool CALIB, bool FULL>
__global__ __launch_bounds__(256) void stack_median_kernel(const StackParams prm)
{
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int lane = threadIdx.x;
    const int64_t p = base + lane;
    __shared__ FrameScalars<NP> fs;
    if constexpr (CALIB || !FULL) stage_frame_scalars<NP>(prm, fs);
    if (p >= prm.P) return;
    float v[NP];
    const int n = load_column<NP, RawT, CALIB, false, FULL>(prm, fs, base, lane, v);
    sort_column<NP>(v);
    const float m1 = pick_at<NP>(v, (n - 1) >> 1);
    const float m2 = pick_at<NP>(v, n >> 1);
    const double med = ((double)m1 + (double)m2) / 2.0;
    if (prm.median) prm.median[p] = n > 0 ? (float)med : __builtin_nanf("");
    if (prm.count) prm.count[p] = n;
}

// -------------------------------------------------------------------------------------------------
// uint16 median stacks (config 4): order statistics commute with a monotone map.  When every frame has
// the same exposure ratio and no pedestal (the usual case: one exposure time per set), a pixel's
// calibration  raw -> ((raw - b) - e*D) / nf  is the same monotone function for all N frames (each float32
// operation is monotone; nf < 0 merely reverses the order), so the two middle calibrated values are the
// calibrations of the two middle RAW values: the raw uint16 columns are sorted - two pixels per lane with
// v_pk_min_u16 / v_pk_max_u16, i.e. half the compare-exchange instructions per pixel, and one 4-byte load
// per lane per frame - and only two values per pixel are calibrated (exact IEEE path).  uint16 data cannot
// be NaN, and with a uniform map the calibrated column is all-NaN or NaN-free, so count is N or 0.
// The kernel verifies the precondition itself (staged scalars, one __syncthreads_or); if it does not hold
// the workgroup processes its 512 pixels as two ordinary 256-pixel tiles.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cmpx_pk16(uint32_t &x, uint32_t &y)
{
    uint32_t lo, hi;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    x = lo;
    y = hi;
}

template <int NP, int BASE, int... I>
__device__ __forceinline__ void net_chunk_pk16(uint32_t (&v)[NP], std::integer_sequence<int, I...>)
{
    constexpr Net<NP> net = make_net<NP>();
    (cmpx_pk16(v[net.ce[BASE + I].a], v[net.ce[BASE + I].b]), ...);
}

template <int NP, int BASE>
__device__ __forceinline__ void net_from_pk16(uint32_t (&v)[NP])
{
    constexpr int total = make_net<NP>().n;
    constexpr int CH = 64;
    if constexpr (BASE < total) {
        constexpr int len = (total - BASE < CH) ? total - BASE : CH;
        net_chunk_pk16<NP, BASE>(v, std::make_integer_sequence<int, len>{});
        net_from_pk16<NP, BASE + len>(v);
    }
}

// v[LO + rel] for rel in [0, LEN): select tree with static register indices (a branchy binary search over the
// registers gets turned into a run-time indexed array by the compiler, i.e. the column is demoted to scratch).
template <int LO, int LEN, int NP>
__device__ __forceinline__ uint32_t pick_rel_u32(const uint32_t (&v)[NP], int rel)
{
    if constexpr (LEN == 1) {
        return v[LO];
    } else if constexpr ((LEN & (LEN - 1)) == 0) {
        constexpr int H = LEN / 2;
        const uint32_t lo = pick_rel_u32<LO, H, NP>(v, rel);
        const uint32_t hi = pick_rel_u32<LO + H, H, NP>(v, rel);
        return (rel & H) ? hi : lo;
    } else {
        constexpr int H = next_pow2(LEN) / 2;
        const uint32_t lo = pick_rel_u32<LO, H, NP>(v, rel);
        const uint32_t hi = pick_rel_u32<LO + H, LEN - H, NP>(v, rel - H);
        return (rel >= H) ? hi : lo;
    }
}

template <bool CALIB>
__device__ __forceinline__ float calibrate_exact_u16(unsigned raw, float b, float D, float e, float nf, bool dodiv)
{
    float x = (float)raw;
    if constexpr (CALIB) {
        x = x - b;                                          // ApCalibrate.py:439
        const float ds = e * D;                             // :450
        x = x - ds;                                         // :451
        if (dodiv) x = __fdiv_rn(x, nf);                    // :463
    }
    return x;
}

template <int NP, bool CALIB, bool FULL>
__global__ __launch_bounds__(256) void stack_median_u16_kernel(const StackParams prm)
{
    __shared__ FrameScalars<NP> fs;
    const int lane = threadIdx.x;
    bool monotone = true;
    if constexpr (CALIB) {
        stage_frame_scalars<NP>(prm, fs);
        const bool differs = lane < NP && (!(fs.e[lane] == fs.e[0]) || fs.ped[lane] != 0.f);
        monotone = !__syncthreads_or(differs);
    } else if constexpr (!FULL) {
        stage_frame_scalars<NP>(prm, fs);
    }
    const int N = prm.N;
    if (monotone) {
        const int64_t p2 = ((int64_t)blockIdx.x * 256 + lane) * 2;     // this lane's pixel pair (P is even here)
        if (p2 >= prm.P) return;
        uint32_t w[NP];
        {
            const uint32_t *fp = reinterpret_cast<const uint32_t *>(static_cast<const uint16_t *>(prm.frames) + (int64_t)blockIdx.x * 512);
            const int64_t step = prm.stride / 2;
            int nframes = N;
            if constexpr (!FULL) asm volatile("" : "+s"(nframes));     // see load_raw
#pragma unroll
            for (int f = 0; f < NP; f++) {
                w[f] = fp[lane];
                if (FULL || f + 1 < nframes) fp += step;    // padded slots re-read the last frame
                if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!FULL) {
#pragma unroll
                for (int f = 0; f < NP; f++) w[f] |= (uint32_t)((N - 1 - f) >> 31);  // f >= N: all ones, sorts to the top
            }
        }
        if constexpr (NP > 1) net_from_pk16<NP, 0>(w);
        const int i1 = FULL ? (NP - 1) >> 1 : (N - 1) >> 1, i2 = FULL ? NP >> 1 : N >> 1;
        const uint32_t m1 = FULL ? w[(NP - 1) >> 1] : pick_rel_u32<0, NP, NP>(w, i1);
        const uint32_t m2 = FULL ? w[NP >> 1] : pick_rel_u32<0, NP, NP>(w, i2);
        float bb[2] = {0.f, 0.f}, dd[2] = {0.f, 0.f}, nn[2] = {1.f, 1.f};
        bool dodiv[2] = {false, false};
        const float e = CALIB ? fs.e[0] : 0.f;
        if constexpr (CALIB) {
            const float2 b2 = *reinterpret_cast<const float2 *>(prm.bias + p2);
            const float2 d2 = *reinterpret_cast<const float2 *>(prm.dark + p2);
            bb[0] = b2.x; bb[1] = b2.y;
            dd[0] = prm.still_biased ? d2.x - b2.x : d2.x;  // ApCalibrate.py:440-445
            dd[1] = prm.still_biased ? d2.y - b2.y : d2.y;
            if (prm.nflat) {
                const float2 n2 = *reinterpret_cast<const float2 *>(prm.nflat + p2);
                nn[0] = n2.x; nn[1] = n2.y;
                dodiv[0] = n2.x != 0.f;                     // ApCalibrate.py:462 (NaN != 0 is True)
                dodiv[1] = n2.y != 0.f;
            }
        }
        float med[2];
        int cnt[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned r1 = (m1 >> (16 * h)) & 0xffffu, r2 = (m2 >> (16 * h)) & 0xffffu;
            const float c1 = calibrate_exact_u16<CALIB>(r1, bb[h], dd[h], e, nn[h], dodiv[h]);
            const float c2 = calibrate_exact_u16<CALIB>(r2, bb[h], dd[h], e, nn[h], dodiv[h]);
            const bool skip = prm.pixmask && prm.pixmask[p2 + h];
            const bool any = (c1 == c1) && (c2 == c2) && !skip;     // a uniform map gives all-NaN or NaN-free columns
            cnt[h] = any ? N : 0;
            med[h] = any ? (float)(((double)c1 + (double)c2) / 2.0) : __builtin_nanf("");
        }
        if (prm.median) *reinterpret_cast<float2 *>(prm.median + p2) = make_float2(med[0], med[1]);
        if (prm.count) *reinterpret_cast<int2 *>(prm.count + p2) = make_int2(cnt[0], cnt[1]);
    } else {
        // per-frame exposure ratios / pedestals: the ordinary path, two 256-pixel tiles per workgroup
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int64_t base = ((int64_t)blockIdx.x * 2 + half) * 256;
            const int64_t p = base + lane;
            if (p >= prm.P) break;
            float v[NP];
            StackParams q = prm;
            asm volatile("" : "+s"(q.N));                  // keeps the NP (f < N) masks of this rare path inside the loop
            const int n = load_column<NP, uint16_t, CALIB, false, FULL>(q, fs, base, lane, v);
            sort_column<NP>(v);
            const float m1 = pick_at<NP>(v, (n - 1) >> 1);
            const float m2 = pick_at<NP>(v, n >> 1);
            const double med = ((double)m1 + (double)m2) / 2.0;
            if (prm.median) prm.median[p] = n > 0 ? (float)med : __builtin_nanf("");
            if (prm.count) prm.count[p] = n;
        }
    }
}

// One pixel of the uint16 pair kernel: its sorted raw column arrives packed two values per register.
template <int NP, bool CALIB, bool FULL>
__device__ __forceinline__ void reduce_sorted_raw_column(const StackParams &prm, const FrameScalars<NP> &fs,
                                                         const uint32_t (&cur)[NP >= 2 ? NP / 2 : 1], float b, float D, float nf,
                                                         bool dv, int64_t p)
{
    constexpr int HP = NP >= 2 ? NP / 2 : 1;
    const int N = prm.N;
    const bool skip = prm.pixmask && prm.pixmask[p];
    float v[NP];
    int n = N;
    bool fast = !skip;
    float rawf[NP];
    if constexpr (NP >= 2) {
#pragma unroll
        for (int k = 0; k < HP; k++) {
            rawf[2 * k] = (float)(cur[k] & 0xffffu);
            rawf[2 * k + 1] = (float)(cur[k] >> 16);
        }
    } else {
        rawf[0] = (float)cur[0];
    }
    if constexpr (CALIB) {
        // non-decreasing map: finite masters and a positive (or unused) flat
        const bool increasing = (fabsf(b) < __builtin_inff()) && (fabsf(D) < __builtin_inff()) && (!dv || (nf > 0.f && nf < __builtin_inff()));
        const bool good = calibrate_fast<NP, float, false>(fs, rawf, b, D, nf, dv, v);
        fast = fast && good && increasing;
    } else {
#pragma unroll
        for (int f = 0; f < NP; f++) v[f] = rawf[f];
    }
    if (__all(fast)) {
        if constexpr (!FULL) {
#pragma unroll
            for (int f = 0; f < NP; f++) asm("v_max_f32 %0, %1, %2" : "=v"(v[f]) : "v"(v[f]), "v"(fs.pad[f]));
        }
    } else {
        // rare: exact IEEE calibration of every value (frame order is irrelevant: the per-frame scalars are
        // uniform), non-finite results and padding become sentinels, and the column is sorted the ordinary way.
        // The raw values are re-read from memory: the sorted copy has left the registers.
        n = 0;
        const float e = CALIB ? fs.e[0] : 0.f;
        const uint16_t *fp = static_cast<const uint16_t *>(prm.frames) + p;
#pragma unroll
        for (int f = 0; f < NP; f++) {
            const float x = calibrate_exact_u16<CALIB>(*fp, b, D, e, nf, dv);
            if (FULL || f + 1 < N) fp += prm.stride;
            const bool ok = (fabsf(x) < __builtin_inff()) && (FULL || f < N) && !skip;
            n += ok ? 1 : 0;
            v[f] = ok ? x : __builtin_inff();
        }
        sort_column<NP>(v);
    }
    reduce_and_store<NP, true>(prm, v, n, p);
}

// -------------------------------------------------------------------------------------------------
// uint16 clipped stacks, two pixels per lane.  Same observation as for the median kernel: with one exposure
// ratio for all frames, no pedestal and a positive flat, calibration is one non-decreasing function per pixel,
// so sorting the RAW uint16 column sorts the calibrated column.  The raw columns of two neighbouring pixels
// are sorted together with packed 16-bit compare-exchanges (543 v_pk_min_u16 + 543 v_pk_max_u16 for BOTH
// pixels - half the sort cost per pixel, and one 4-byte load per lane per frame); each pixel's sorted raw
// column is then calibrated (same packed fast path and guards as the lean kernel) and reduced WITHOUT a
// second sort.  Lanes that do not meet the precondition (flat <= 0 / non-finite masters / guard hit) take the
// exact per-value path followed by the ordinary sort; workgroups whose per-frame scalars are not uniform run
// the ordinary kernel body on their 512 pixels.  Results are bit-identical to the one-pixel-per-lane kernel:
// the survivors are summed in sorted order in both.
// -------------------------------------------------------------------------------------------------
template <int NP, bool CALIB, bool FULL>
__global__ __launch_bounds__(256, NP <= 64 ? 3 : 1) void stack_sigclip_u16_pairs_kernel(const StackParams prm)
{
    __shared__ FrameScalars<NP> fs;
    const int lane = threadIdx.x;
    bool monotone = true;
    if constexpr (CALIB) {
        stage_frame_scalars<NP>(prm, fs);
        const bool differs = lane < NP && (!(fs.e[lane] == fs.e[0]) || fs.ped[lane] != 0.f);
        monotone = !__syncthreads_or(differs);
    } else if constexpr (!FULL) {
        stage_frame_scalars<NP>(prm, fs);
    }
    const int N = prm.N;
    if (!monotone) {
        // per-frame exposure ratios / pedestals: the ordinary path, two 256-pixel tiles per workgroup
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int64_t base = ((int64_t)blockIdx.x * 2 + half) * 256;
            const int64_t p = base + lane;
            if (p >= prm.P) break;
            float v[NP];
            StackParams q = prm;
            asm volatile("" : "+s"(q.N));                  // keeps the NP (f < N) masks of this rare path inside the loop
            const int n = load_column<NP, uint16_t, CALIB, true, FULL>(q, fs, base, lane, v);
            reduce_and_store<NP>(prm, v, n, p);
        }
        return;
    }
    const int64_t p2 = ((int64_t)blockIdx.x * 256 + lane) * 2;         // this lane's pixel pair (P is even here)
    if (p2 >= prm.P) return;
    uint32_t w[NP];
    {
        const uint32_t *fp = reinterpret_cast<const uint32_t *>(static_cast<const uint16_t *>(prm.frames) + (int64_t)blockIdx.x * 512);
        const int64_t step = prm.stride / 2;
        int nframes = N;
        if constexpr (!FULL) asm volatile("" : "+s"(nframes));         // see load_raw
#pragma unroll
        for (int f = 0; f < NP; f++) {
            w[f] = fp[lane];
            if (FULL || f + 1 < nframes) fp += step;        // padded slots re-read the last frame
            if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!FULL) {
#pragma unroll
            for (int f = 0; f < NP; f++) w[f] |= (uint32_t)((N - 1 - f) >> 31);  // f >= N: all ones, sorts to the top
        }
    }
    if constexpr (NP > 1) net_from_pk16<NP, 0>(w);

    // Re-pack the two sorted columns: cur[k] = (raw[2k], raw[2k+1]) of the first pixel stays in registers, the
    // second pixel's column is parked in LDS (NP/2 dwords per lane, bank = lane) while the first is reduced, so
    // only one column is register-resident at a time.
    constexpr int HP = NP >= 2 ? NP / 2 : 1;
    __shared__ uint32_t parked[HP][256];
    uint32_t cur[HP];
    if constexpr (NP >= 2) {
#pragma unroll
        for (int k = 0; k < HP; k++) {
            cur[k] = (w[2 * k] & 0xffffu) | (w[2 * k + 1] << 16);
            parked[k][lane] = (w[2 * k] >> 16) | (w[2 * k + 1] & 0xffff0000u);
        }
    } else {
        cur[0] = w[0] & 0xffffu;
        parked[0][lane] = w[0] >> 16;
    }

    float bb[2] = {0.f, 0.f}, dd[2] = {0.f, 0.f}, nn[2] = {1.f, 1.f};
    bool dodiv[2] = {false, false};
    if constexpr (CALIB) {
        const float2 b2 = *reinterpret_cast<const float2 *>(prm.bias + p2);
        const float2 d2 = *reinterpret_cast<const float2 *>(prm.dark + p2);
        bb[0] = b2.x; bb[1] = b2.y;
        dd[0] = prm.still_biased ? d2.x - b2.x : d2.x;      // ApCalibrate.py:440-445
        dd[1] = prm.still_biased ? d2.y - b2.y : d2.y;
        if (prm.nflat) {
            const float2 n2 = *reinterpret_cast<const float2 *>(prm.nflat + p2);
            nn[0] = n2.x; nn[1] = n2.y;
            dodiv[0] = n2.x != 0.f;                         // ApCalibrate.py:462 (NaN != 0 is True)
            dodiv[1] = n2.y != 0.f;
        }
    }
    reduce_sorted_raw_column<NP, CALIB, FULL>(prm, fs, cur, bb[0], dd[0], nn[0], dodiv[0], p2);
    // the parked column takes over the registers (through an opaque pointer: otherwise the compiler forwards the
    // stored values to these loads, i.e. keeps the column in NP/2 registers across the whole first reduction)
    int slot = lane;
    asm volatile("" : "+v"(slot) : : "memory");             // opaque index: no store-to-load forwarding in registers
#pragma unroll
    for (int k = 0; k < HP; k++) cur[k] = parked[k][slot];
    reduce_sorted_raw_column<NP, CALIB, FULL>(prm, fs, cur, bb[1], dd[1], nn[1], dodiv[1], p2 + 1);
}

template <int NP, typename RawT, bool CALIB>
int launch_one(const StackParams &prm, bool median_only, hipStream_t st)
{
    if constexpr (sizeof(RawT) == 2) {
        // uint16 median: pixel pairs per lane need 4-byte aligned frame rows and 8-byte aligned planes
        const bool pairs = median_only && (prm.P % 2 == 0) && (prm.stride % 2 == 0) &&
                           ((reinterpret_cast<uintptr_t>(prm.frames) & 3) == 0) &&
                           ((reinterpret_cast<uintptr_t>(prm.bias) | reinterpret_cast<uintptr_t>(prm.dark) |
                             reinterpret_cast<uintptr_t>(prm.nflat) | reinterpret_cast<uintptr_t>(prm.median) |
                             reinterpret_cast<uintptr_t>(prm.count)) & 7) == 0;
        const bool rich_out = !median_only && (prm.median || prm.std || prm.dev == APGPU_DEV_MAD_STD);
        const bool pairs_clip = !median_only && !rich_out && !prm.persistent && (prm.P % 2 == 0) && (prm.stride % 2 == 0) &&
                                ((reinterpret_cast<uintptr_t>(prm.frames) & 3) == 0) &&
                                ((reinterpret_cast<uintptr_t>(prm.bias) | reinterpret_cast<uintptr_t>(prm.dark) |
                                  reinterpret_cast<uintptr_t>(prm.nflat)) & 7) == 0;
        if (pairs_clip) {
            const int64_t grid = (prm.P + 511) / 512;
            if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
            if (prm.N == NP) hipLaunchKernelGGL((stack_sigclip_u16_pairs_kernel<NP, CALIB, true>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            else hipLaunchKernelGGL((stack_sigclip_u16_pairs_kernel<NP, CALIB, false>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            return check_launch("stack kernel (uint16 pairs)");
        }
        if (pairs) {
            const int64_t grid = (prm.P + 511) / 512;
            if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
            if (prm.N == NP) hipLaunchKernelGGL((stack_median_u16_kernel<NP, CALIB, true>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            else hipLaunchKernelGGL((stack_median_u16_kernel<NP, CALIB, false>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            return check_launch("stack median kernel (uint16 pairs)");
        }
    }

    const bool rich = !median_only && (prm.median || prm.std || prm.dev == APGPU_DEV_MAD_STD);
    const bool full = prm.N == NP;
    const int block = rich ? rich_block<NP>() : 256;
    const int64_t grid = (prm.P + block - 1) / block;
    if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
    const dim3 g((unsigned)grid), b(block);
    if (median_only) {
        if (full) hipLaunchKernelGGL((stack_median_kernel<NP, RawT, CALIB, true>), g, b, 0, st, prm);
        else hipLaunchKernelGGL((stack_median_kernel<NP, RawT, CALIB, false>), g, b, 0, st, prm);
    } else if (rich) {
        if (full) hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, true>), g, b, 0, st, prm);
        else hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, false>), g, b, 0, st, prm);
    } else if (full) {
        if constexpr (CALIB && NP >= 2 && NP <= 64) {       // two columns in flight: register budget of <= 64 slots
            if (prm.persistent) {
                const int64_t ntiles = (prm.P + 255) / 256;
                const int64_t gp = ntiles < 2 * kNumCU ? ntiles : 2 * kNumCU;
                hipLaunchKernelGGL((stack_sigclip_persistent_kernel<NP, RawT>), dim3((unsigned)gp), b, 0, st, prm);
                return check_launch("stack kernel (persistent)");
            }
        }
        hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, true>), g, b, 0, st, prm);
    } else {
        hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, false>), g, b, 0, st, prm);
    }
    return check_launch("stack kernel");
}

// launch_one<NP, RawT, CALIB> is explicitly instantiated in the stack_inst_*.hip translation units (one group of
// slot counts each, so that the build parallelises); everybody else only sees these declarations.
#ifndef APGPU_STACK_INSTANTIATE
#define APGPU_DECLARE_LAUNCH(NP)                                                                          \
    extern template int launch_one<NP, float, true>(const StackParams &, bool, hipStream_t);              \
    extern template int launch_one<NP, float, false>(const StackParams &, bool, hipStream_t);             \
    extern template int launch_one<NP, uint16_t, true>(const StackParams &, bool, hipStream_t);           \
    extern template int launch_one<NP, uint16_t, false>(const StackParams &, bool, hipStream_t);
APGPU_DECLARE_LAUNCH(1) APGPU_DECLARE_LAUNCH(4) APGPU_DECLARE_LAUNCH(8) APGPU_DECLARE_LAUNCH(12) APGPU_DECLARE_LAUNCH(16)
APGPU_DECLARE_LAUNCH(24) APGPU_DECLARE_LAUNCH(32) APGPU_DECLARE_LAUNCH(48) APGPU_DECLARE_LAUNCH(64) APGPU_DECLARE_LAUNCH(96)
APGPU_DECLARE_LAUNCH(128)
#undef APGPU_DECLARE_LAUNCH
#endif

template <typename RawT, bool CALIB>
int launch_np(const StackParams &prm, bool median_only, hipStream_t st)
{
    const int N = prm.N;
    // slot counts: powers of two and their 3/4 points (pruned networks), so padding wastes at most a third
    if (N <= 1) return launch_one<1, RawT, CALIB>(prm, median_only, st);
    if (N <= 4) return launch_one<4, RawT, CALIB>(prm, median_only, st);
    if (N <= 8) return launch_one<8, RawT, CALIB>(prm, median_only, st);
    if (N <= 12) return launch_one<12, RawT, CALIB>(prm, median_only, st);
    if (N <= 16) return launch_one<16, RawT, CALIB>(prm, median_only, st);
    if (N <= 24) return launch_one<24, RawT, CALIB>(prm, median_only, st);
    if (N <= 32) return launch_one<32, RawT, CALIB>(prm, median_only, st);
    if (N <= 48) return launch_one<48, RawT, CALIB>(prm, median_only, st);
    if (N <= 64) return launch_one<64, RawT, CALIB>(prm, median_only, st);
    if (N <= 96) return launch_one<96, RawT, CALIB>(prm, median_only, st);
    return launch_one<128, RawT, CALIB>(prm, median_only, st);
}


}  // namespace apgpu_stack
