// stack_reduce.h - per-pixel sigma-clipped reduction of a sorted column: clip state and trim chains, the lean (register) and rich (LDS)
// reductions (astropy sigma_clipping.py:298-383, 924-937; mad_std funcs.py:844-850).
#pragma once
#include "stack_calibrate.h"

namespace apgpu_stack {

using namespace apgpu;

// Per-lane state of the clipping loop.  Survivors are v[a .. b) of the sorted column.
struct ClipState {
    double S, Q;            // sum(x - c), sum((x - c)^2) over the survivors
    double c;               // pivot
    double cen, woff, nn;   // the last bounds: w = wscale (x - cen) - woff is the scaled distance from their centre; count
    double wscale;          // scale of the bound test: n (std mode, T = sigma^2 n^2 var) or 1 (mad_std mode)
    double Tlo, Thi;        // sigma^2 * (n*Q - S^2): squared, n^2-scaled half-widths of the bounds
    int a, b;
};

// Centre = median: cen = med, woff = 0.  Centre = mean = c + S / n: cen = c, woff = wscale S / n, i.e. n (x - mean) is formed
// as n (x - c) - S without the division - exact for integer-valued frames, so that a value that EQUALS a bound (few discrete
// levels: levels L, L+d, L+2d with counts 3, 5, 5 give mean - 1.5 std = L) is kept, as x >= lower_bound keeps it in astropy
// whenever its own rounding lands on the tie.
__device__ __forceinline__ bool below(const ClipState &st, double xd)
{
    const double w = fma(st.wscale, xd - st.cen, -st.woff);
    return (w < 0.0) && (w * w > st.Tlo);
}

__device__ __forceinline__ bool above(const ClipState &st, double xd)
{
    const double w = fma(st.wscale, xd - st.cen, -st.woff);
    return (w > 0.0) && (w * w > st.Thi);
}

// The kernel's argument block read LATE (round 4): the kernarg pointer, made opaque where it is used, so that the scalar loads
// happen THERE - the output pointers and clip parameters used to be read at kernel entry and then either sat in SGPRs across
// the calibration and the sort (spilled to VGPR lanes) or were parked in 15 VGPRs; read after the sort they cost a handful
// of s_load and no register that lives through the column phase.  Valid in kernels whose first argument is the StackParams.
typedef const StackParams __attribute__((address_space(4))) LateParams;
__device__ __forceinline__ LateParams *late_params()
{
    LateParams *kp = (LateParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
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
    double d;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x));     // the conversion itself is the barrier: no extra v_mov
    return d;
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
        if (wave_any(active && (st.a > I))) trim_low<I + 1, NP>(v, st, active);
    }
}

// ns (wave-uniform, = N) / MINN: slots >= ns are padding for every lane; levels I >= MINN test that with a scalar compare
// and step down without touching the VALU (one call site per level: a second one would double the inlined chain).
template <int I, int NP, int MINN = NP>
__device__ __forceinline__ void trim_high(const float (&v)[NP], ClipState &st, bool active, int ns = NP)
{
    if constexpr (I >= 0) {
        bool down = true;
        if (I < MINN || I < ns) {
            if (wave_any(active && (I < st.b))) {           // slots above every lane's range: just step down
                const double xd = widen(v[I]);
                const bool rej = active && (I >= st.a) && (I < st.b) && above(st, xd);
                if (rej) {
                    const double d = xd - st.c;
                    st.S -= d;
                    st.Q = fma(-d, d, st.Q);
                    st.b = I;
                }
            }
            down = wave_any(active && (st.b <= I));
        }
        if (down) trim_high<I - 1, NP, MINN>(v, st, active, ns);
    }
}

template <int I, int NP>
__device__ __forceinline__ void readmit_low(const float (&v)[NP], ClipState &st, int &a_new)
{
    if constexpr (I < NP) {
        if (wave_any(I < st.a)) {
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

template <int I, int NP, int MINN = NP>
__device__ __forceinline__ void readmit_high(const float (&v)[NP], ClipState &st, int n, int &b_new, int ns = NP)
{
    if constexpr (I >= 0) {
        bool down = true;
        if (I < MINN || I < ns) {
            down = wave_any(I >= st.b);
            if (down) {
                const double xd = widen(v[I]);
                const bool keep = (I >= st.b) && (I < n) && !below(st, xd) && !above(st, xd);
                if (keep) {
                    const double d = xd - st.c;
                    st.S += d;
                    st.Q = fma(d, d, st.Q);
                    b_new = b_new > I + 1 ? b_new : I + 1;
                }
            }
        }
        if (down) readmit_high<I - 1, NP, MINN>(v, st, n, b_new, ns);
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

// Element i of the lane's column; B = lanes per LDS row (the workgroup size of the kernel that parked the column).
template <int NP, int B = rich_block<NP>()>
__device__ __forceinline__ float col_read(const float *col, int i)
{
    i = i < 0 ? 0 : (i > NP - 1 ? NP - 1 : i);
    return col[i * B];
}

// astropy.stats.mad_std of the survivors x[a .. b): 1.482602218505602 * median(|x - med|)
// (astropy/stats/funcs.py:844-850, 917-920; the C loop's mad_buffer).  No second sort: the column is
// sorted, so the j+1 deviations nearest to med belong to a contiguous window [L, L+j], and the j-th order
// statistic of the deviations is  min over L of max(|x_L - med|, |x_(L+j) - med|)  (|x - med| is convex
// along the sorted column, so a window's largest deviation sits at one of its ends).  Every value is the
// exact float64 |x - med| the reference sorts.
// Round 4: the minimum is FOUND, not scanned for.  With j = k1 = (n - 1) / 2 the windows start at L in [a, b - k1): their
// left ends lie at or below the median pair, their right ends at or above it, so dl(L) = |x_L - med| never increases and
// dr(L) = |x_(L+k1) - med| never decreases along L (x is sorted; float64 subtraction and |.| are monotone) and
// f(L) = max(dl(L), dr(L)) has its minimum where they cross: a binary search for the first L with dr >= dl (it exists: the
// last window ends on the column's maximum), then f at that L and the one before.  The even-count partner
// g(L) = max(dl(L - 1), dr(L)) (the window one longer) crosses at the same L or the next.  12 + 6 LDS reads per pass for
// 64 slots where the scan made 128, and the float64 arithmetic of 64 windows is gone (the ccdproc configuration of
// ApMasterCal - one pass of median / mad_std - spent more than half of its time here).
template <int NP, int B = rich_block<NP>()>
__device__ __forceinline__ double mad_std_window(const float *col, bool active, int a, int b, double med)
{
    const int n = b - a;
    const int k1 = n > 0 ? (n - 1) >> 1 : 0;
    const bool even = (n & 1) == 0;
    const int bk = b - k1;                                  // windows start at L in [a, bk)
    const double inf = __builtin_inf();
    int lo = a, hi = bk - 1;                                // the crossing lies in [lo, hi]
    for (;;) {
        const bool open = active && lo < hi;
        if (!wave_any(open)) break;
        const int mid = (lo + hi) >> 1;
        const double dl = fabs((double)col_read<NP, B>(col, mid) - med);
        const double dr = fabs((double)col_read<NP, B>(col, mid + k1) - med);
        if (open) {
            if (dr >= dl) hi = mid;
            else lo = mid + 1;
        }
    }
    // dl at lo - 2 .. lo, dr at lo - 1 .. lo + 1; windows that leave [a, b) get an infinite deviation
    double dl[3], dr[3];
    float xl[3], xr[3];
#pragma unroll
    for (int j = 0; j < 3; j++) xl[j] = col_read<NP, B>(col, lo - 2 + j);
#pragma unroll
    for (int j = 0; j < 3; j++) xr[j] = col_read<NP, B>(col, lo - 1 + j + k1);
#pragma unroll
    for (int j = 0; j < 3; j++) dl[j] = (lo - 2 + j >= a) ? fabs((double)xl[j] - med) : inf;
#pragma unroll
    for (int j = 0; j < 3; j++) dr[j] = (lo - 1 + j < bk) ? fabs((double)xr[j] - med) : inf;
    const double m1 = fmin(fmax(dl[1], dr[0]), fmax(dl[2], dr[1]));                                  // f(lo - 1), f(lo)
    const double m2 = fmin(fmin(fmax(dl[0], dr[0]), fmax(dl[1], dr[1])), fmax(dl[2], dr[2]));       // g(lo - 1), g(lo), g(lo + 1)
    const double x1 = m1, x2 = even ? m2 : m1;
    return (0.5 * (x1 + x2)) * 1.482602218505602;
}

// N-shard partial moments of the survivors: sum(x) = n c + S, sum(x^2) = Q + 2 c S + n c^2 (exact identities, float64).
//   layout 0  float32 [3][P]: sum, count, sum of squares - the first two are all a mean needs, so an exchange that does
//             not want std all-reduces a contiguous [2][P] prefix (8 bytes per pixel);
//   layout 1  double sum[P], double sumsq[P], int32 count[P] - the combine of SURVEY 8(e) "f64 sum + i32 count"
//             (layout 2: add to what the buffer holds - the chunks of a stack beyond APGPU_MAX_STACK);
//   layout 3  packed float64 [3][P]: sum, count, sum of squares - the count as a float64 (exact to 2^53), so that ONE
//             all-reduce of the contiguous [2][P] prefix (16 bytes per pixel; [3][P] = 24 with a std) carries everything
//             (layout 4: add to what the buffer holds).
__device__ __forceinline__ void store_moments(void *out, int f64_layout, int64_t Pn, int64_t p, int cnt, double c, double S, double Q)
{
    const double nf = (double)cnt;
    const double sum = cnt > 0 ? fma(nf, c, S) : 0.0;
    const double sq = cnt > 0 ? Q + 2.0 * c * S + nf * c * c : 0.0;
    if (f64_layout >= 3) {
        double *d = static_cast<double *>(out);
        const bool acc = f64_layout == 4;
        d[p] = acc ? d[p] + sum : sum;
        d[Pn + p] = acc ? d[Pn + p] + nf : nf;
        d[2 * Pn + p] = acc ? d[2 * Pn + p] + sq : sq;
    } else if (f64_layout) {
        double *d = static_cast<double *>(out);
        int32_t *k = reinterpret_cast<int32_t *>(d + 2 * Pn);
        const bool acc = f64_layout == 2;                   // accumulate onto the moments of earlier chunks of the stack
        d[p] = acc ? d[p] + sum : sum;
        d[Pn + p] = acc ? d[Pn + p] + sq : sq;
        k[p] = acc ? k[p] + cnt : cnt;
    } else {
        float *f = static_cast<float *>(out);
        f[p] = (float)sum;
        f[Pn + p] = (float)cnt;
        f[2 * Pn + p] = (float)sq;
    }
}

// -------------------------------------------------------------------------------------------------
// float32 fast path of the lean reduction (round 3).  The exact path below keeps S, Q and every bound test in float64
// (4-cycle instructions on gfx950); here the moments and the tests are float32 (2-cycle class: v_sub/v_add/v_mul/v_fmac_f32)
// with an ERROR MARGIN: a comparison whose outcome the float32 rounding errors could change marks the lane "unsure", and a
// wave with an unsure lane redoes the column on the exact path - so the survivor sets are those of the exact path, always.
//
// Layout of the sums: the clip only trims the ends of the sorted column, so the column is cut into a core [T, NP-T) that
// is summed once (Sc, Qc) and two tails of T = 4 elements whose partial sums are tabulated from the inside out
// (SL[k] = sum of d_k .. d_(T-1), SH[k] = sum of the first k upper-tail values).  S and Q of the current range are then
// Sc + SL[a] + SH[b - (NP-T)]: sums of what is IN the range only - no subtraction of an outlier from a total that it
// dominates, which in float32 would leave no digits (a 5000 ADU cosmic ray in a 30 ADU column is 4 decimal digits of Q).
// A lane that wants to trim into the core leaves the fast path.
//
// Error budget (u = 2^-24, all quantities relative unless noted; d_i = fl(x_i - c) with c the lower median):
//   Q  = sum d_i^2 : every term and partial sum is positive, 14 + 2 + 4 roundings deep, inputs 2u      -> |dQ| <= 20u Q
//   S  = sum d_i   : 19u sum|d_i| <= 19u sqrt(n Q)   (absolute; Cauchy-Schwarz)
//   V  = n Q - S^2 : |dV| <= 21u nQ + 2 sqrt(nQ) 19u sqrt(nQ) + u V <= 60u nQ + u V; the guard V >= nQ / 4 makes it <= 250u V
//   T  = 4 sigma^2 V (float32 sigma^2: u, two products: 2u)                                            -> <= 253u
//   w  = n ((x - m1) + (x - m2)) = 2 n (x - median): differences u each, same-sign sum u, product u     -> <= 3u, w^2 <= 7u
//   reject <=> w^2 > T exactly; decided in float32 as w^2 > T (1 + rho) [sure reject] / w^2 <= T (1 - rho) [sure keep] with
//   rho = 2^-15 = 512u > (1 + 7u) / (1 - 253u) - 1 = 261u (+ u for each threshold product): twice the worst case.
//   The sign tests of the exact path are implied here: the lowest survivor of a sorted column is <= its median.
// Range guards: |d| of the two column ends in (2^-40, 2^40) or all d = 0 (no underflow of d^2, no overflow of n^2 w^2).
// The output mean c + S / n inherits S's float32 rounding (a few 1e-3 ulp(float32) of the mean for rms(d) << |c|); the
// guard rms(d) <= |c| / 4 keeps columns whose mean is small against their spread (sky-subtracted data) on the exact path.
// Requirements: full stack (n = NP for the whole wave, no sentinel), median centre, std deviation, NP >= 16.
// -------------------------------------------------------------------------------------------------
#ifndef APGPU_LATE_PARAMS
#define APGPU_LATE_PARAMS 1
#endif
#ifndef APGPU_FAST32_RHO
#define APGPU_FAST32_RHO 0x1p-15f
#endif

struct Fast32 {
    float m1, m2;                       // the middle pair of the current range (its median is their mean)
    float nf;                           // b - a
    float tl_hi, tl_lo, th_hi, th_lo;   // 4 sigma^2 V (1 +- rho) for the low / high side
    float Slo, Qlo, Shi, Qhi;           // tail sums inside the current range
    int a, b;
    bool unsure;
};

__device__ __forceinline__ float fast32_t(const Fast32 &f, float x)
{
    const float w = f.nf * ((x - f.m1) + (x - f.m2));
    return w * w;
}

template <int I, int NP, int T = kFastTail>
__device__ __forceinline__ void trim_low_fast(const float (&v)[NP], Fast32 &f, const float (&SL)[T + 1], const float (&QL)[T + 1])
{
    const float t = fast32_t(f, v[I]);
    const bool at = f.a == I;
    if constexpr (I < T) {
        const bool rej = at && (t > f.tl_hi);
        const bool maybe = at && (t > f.tl_lo);
        f.unsure = f.unsure || (maybe != rej);
        if (rej) {
            f.a = I + 1;
            f.Slo = SL[I + 1];
            f.Qlo = QL[I + 1];
        }
        if (wave_any(f.a > I)) trim_low_fast<I + 1, NP, T>(v, f, SL, QL);   // some lane's cursor is (now or from an earlier pass, or by padding) past I
    } else {
        f.unsure = f.unsure || (at && (t > f.tl_lo));       // would trim into the core
    }
}

// th_hi / th_lo: the upper thresholds (the lower ones again when sigma_lower == sigma_upper - clip_fast32)
template <int K, int NP, int T = kFastTail>                  // K = number of upper-tail elements still in the range
__device__ __forceinline__ void trim_high_fast(const float (&v)[NP], Fast32 &f, const float (&SH)[T + 1], const float (&QH)[T + 1],
                                               const float th_hi, const float th_lo)
{
    constexpr int I = NP - T + K - 1;                       // the element under test: the highest one in the range
    const float t = fast32_t(f, v[I]);
    const bool at = f.b == I + 1;
    if constexpr (K > 0) {
        const bool rej = at && (t > th_hi);
        const bool maybe = at && (t > th_lo);
        f.unsure = f.unsure || (maybe != rej);
        if (rej) {
            f.b = I;
            f.Shi = SH[K - 1];
            f.Qhi = QH[K - 1];
        }
        if (wave_any(f.b <= I)) trim_high_fast<K - 1, NP, T>(v, f, SH, QH, th_hi, th_lo);
    } else {
        f.unsure = f.unsure || (at && (t > th_lo));
    }
}

// arr[idx] for a WAVE-UNIFORM idx in [0, N): scalar compare-and-branch chain
template <int LO, int N>
__device__ __forceinline__ float uniform_elem(const float (&arr)[N], int idx)
{
    if constexpr (LO >= N - 1) return arr[N - 1];
    else {
        if (idx <= LO) return arr[LO];
        return uniform_elem<LO + 1, N>(arr, idx);
    }
}

// Returns true (wave-uniform) when every lane of the wave completed on the fast path; a, b, cf, S, Q are then final.
// T: tail length.  plo / phi (wave-uniform; padded stacks with split pads, stack_calibrate.h): v[0 .. plo) are -inf and
// v[NP - phi .. NP) +inf sentinels, phi - plo in {0, 1}, both at most T - 4: the clip simply starts with them trimmed.
// MODE 1 (stack_fast_kernel): the result is PER LANE - whether this lane completed; lanes that did not (or whose column is
// garbage: the kernel's bad lanes ride along) are the caller's to redo.
// MODE 2: per lane, and phi is a PER-LANE count (<= T - 1) of +inf sentinels at the top of the column - non-finite values of
// the stack (np.nan from a resample or an earlier calibration), sorted to the top like padding: the lane's clip starts with
// them trimmed, its middle pair is picked per lane; the pivot of the sums stays the column's static middle (any pivot within
// the data gives the same S and Q up to the float32 roundings the margins cover).
// PLO / PHI >= 0 (static pads: the kernels instantiated per pad count, slot counts up to 64): plo / phi (the wave-uniform
// part of phi in MODE 2) are these compile-time values - the trim chains start BEHIND the pads instead of walking over them
// in every pass (13 instructions per pad and pass: the "holes" between the slot counts of small stacks).
template <int NP, int T = kFastTail, int MODE = 0, int PLO = -1, int PHI = -1>
__device__ __forceinline__ bool clip_fast32(const float (&v)[NP], float sl2f, float su2f, int maxiters, int &a_out, int &b_out,
                                            float &cf_out, float &S_out, float &Q_out, int plo = 0, int phi = 0)
{
    if constexpr (PLO >= 0) plo = PLO;
    if constexpr (PHI >= 0 && MODE != 2) phi = PHI;
    static_assert(NP >= 2 * T + 4 && NP % 4 == 0, "fast path needs a core");
    const bool padded = T > kFastTail || MODE == 2;         // (compile time: full stacks keep every index static)
    const float cf = v[(NP - 1) >> 1];
    // core sums: four chains, fixed association.  S adds the deviations in mirror pairs (i, NP-1-i) of the sorted column:
    // a pair nearly cancels, so the partial sums - and with them the float32 rounding errors, which scale with the
    // magnitude of what is being added - stay at the column's asymmetry instead of its spread (same number of additions).
    // PACKED (round 6): the same chains as 2-wide vectors -> v_pk_add_f32 / v_pk_fma_f32, -83 instructions per wavefront at 64
    // slots.  Round 4 rejected it on the complete kernel (the sorted values leave the network in single registers, the compiler
    // added ~50 v_mov and 5 VGPRs: 170, one over that kernel's budget of three wavefronts per SIMD).  On the FAST kernel (MODE 1 / 2:
    // 110 VGPRs, budget 128) the compiler places the pairs without a single move and the register count does not change: 0.892 ->
    // 0.880 ms on the benchmark, five interleaved runs each on one box (profiles/r06/ab_packed.txt) - although packed float32
    // instructions are of the 4-cycle class and the scalar forms they replace of the 2-cycle class (tools/issue_cost.hip): what
    // is saved are issue slots.  Same four chains per sum, same number of terms per chain: the error budget above is unchanged.
#ifndef APGPU_PACKED_MOMENTS_MAX_NP
#define APGPU_PACKED_MOMENTS_MAX_NP 120   /* (128 slots: at its register budget already - 168 VGPRs, 5 spilled) */
#endif
    constexpr bool kPacked = MODE != 0 && NP <= APGPU_PACKED_MOMENTS_MAX_NP && T % 2 == 0 && (NP / 2 - T) % 2 == 0;
    float Sc, Qc;
    if constexpr (kPacked) {
        const v2f cf2 = {cf, cf};
        v2f Sp[2] = {{0.f, 0.f}, {0.f, 0.f}}, Qp[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int i = T; i < NP / 2; i += 2) {
            const v2f lo = {v[i], v[i + 1]}, hi = {v[NP - 1 - i], v[NP - 2 - i]};       // hi: the mirror images, in mirror order
            const v2f d1 = lo - cf2, d2 = hi - cf2;
            const int c = (i >> 1) & 1;
            Sp[c] += d1 + d2;
            Qp[c] = __builtin_elementwise_fma(d1, d1, Qp[c]);
            Qp[c ^ 1] = __builtin_elementwise_fma(d2, d2, Qp[c ^ 1]);
        }
        Sc = (Sp[0].x + Sp[0].y) + (Sp[1].x + Sp[1].y);
        Qc = (Qp[0].x + Qp[0].y) + (Qp[1].x + Qp[1].y);
    } else {
        float Sa[4] = {0.f, 0.f, 0.f, 0.f}, Qa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = T; i < NP / 2; i++) {
            const float d1 = v[i] - cf, d2 = v[NP - 1 - i] - cf;
            Sa[i & 3] += d1 + d2;
            Qa[i & 3] = __builtin_fmaf(d1, d1, Qa[i & 3]);
            Qa[(i + 2) & 3] = __builtin_fmaf(d2, d2, Qa[(i + 2) & 3]);
        }
        Sc = (Sa[0] + Sa[1]) + (Sa[2] + Sa[3]);
        Qc = (Qa[0] + Qa[1]) + (Qa[2] + Qa[3]);
    }
    // tails, summed from the inside out
    float SL[T + 1], QL[T + 1], SH[T + 1], QH[T + 1];
    SL[T] = 0.f; QL[T] = 0.f; SH[0] = 0.f; QH[0] = 0.f;
#pragma unroll
    for (int k = T - 1; k >= 0; k--) {
        const float d = v[k] - cf;
        SL[k] = SL[k + 1] + d;
        QL[k] = __builtin_fmaf(d, d, QL[k + 1]);
    }
#pragma unroll
    for (int k = 1; k <= T; k++) {
        const float d = v[NP - T + k - 1] - cf;
        SH[k] = SH[k - 1] + d;
        QH[k] = __builtin_fmaf(d, d, QH[k - 1]);
    }
    Fast32 f;
    float vlo = v[0], vhi = v[NP - 1];
    if (!padded) {
        f.a = 0;
        f.b = NP;
        f.Slo = SL[0]; f.Qlo = QL[0]; f.Shi = SH[T]; f.Qhi = QH[T];
        f.m2 = v[NP >> 1];
    } else {
        f.a = plo;
        f.b = NP - phi;
        f.Slo = uniform_elem<0, T + 1>(SL, plo);
        f.Qlo = uniform_elem<0, T + 1>(QL, plo);
        if constexpr (MODE == 2) {
            f.Shi = pick_rel<0, T + 1, T + 1>(SH, T - phi);
            f.Qhi = pick_rel<0, T + 1, T + 1>(QH, T - phi);
            vhi = pick_rel<NP - T, T, NP>(v, T - 1 - phi);
        } else {
            f.Shi = uniform_elem<0, T + 1>(SH, T - phi);
            f.Qhi = uniform_elem<0, T + 1>(QH, T - phi);
            f.m2 = (phi > plo) ? cf : v[NP >> 1];           // an odd number of values: the middle one twice
            vhi = uniform_pick<NP - T, NP>(v, NP - 1 - phi);
        }
        vlo = uniform_pick<0, NP>(v, plo);                  // (scalar compare chains: a few steps from the given start)
    }
    f.m1 = cf;
    if constexpr (MODE == 2) {
        constexpr int LO1 = (NP - T - 1) >> 1, LO2 = (NP - T) >> 1;
        f.m1 = pick_rel<LO1, T + 1, NP>(v, ((f.a + f.b - 1) >> 1) - LO1);
        f.m2 = pick_rel<LO2, T + 1, NP>(v, ((f.a + f.b) >> 1) - LO2);
    }
    // range guard on the extreme deviations (sorted column: they sit at the ends)
    const float dmax = fmaxf(cf - vlo, vhi - cf);
    f.unsure = !(dmax == 0.f || (dmax > 0x1p-40f && dmax < 0x1p40f));
    const float rho = APGPU_FAST32_RHO;
    const float sl4 = 4.f * sl2f, su4 = 4.f * su2f;
    const int max_passes = __builtin_amdgcn_readfirstlane(maxiters);
    float S, Q;
    int it = 0;
    for (;;) {
        const int a0 = f.a, b0 = f.b;
        f.nf = (float)(f.b - f.a);
        S = (Sc + f.Slo) + f.Shi;
        Q = (Qc + f.Qlo) + f.Qhi;
        const float nQ = f.nf * Q;
        const float V = __builtin_fmaf(-S, S, nQ);
        f.unsure = f.unsure || !(V >= 0.25f * nQ);
        const float tl = sl4 * V;
        f.tl_hi = __builtin_fmaf(tl, rho, tl);
        f.tl_lo = __builtin_fmaf(tl, -rho, tl);
        trim_low_fast<(PLO > 0 ? PLO : 0), NP, T>(v, f, SL, QL);
        // (one sigma for both sides would make these the lower thresholds again - 3 instructions per pass less behind a
        // wave-uniform branch, but the second copy of the chain costs 3 VGPRs: 170, over the three-wavefront budget)
        const float th = su4 * V;
        f.th_hi = __builtin_fmaf(th, rho, th);
        f.th_lo = __builtin_fmaf(th, -rho, th);
        trim_high_fast<(PHI > 0 ? T - PHI : T), NP, T>(v, f, SH, QH, f.th_hi, f.th_lo);
        it++;
        const bool changed = (f.a != a0) || (f.b != b0);
        if (!(wave_any(changed) && (max_passes < 0 || it < max_passes))) break;
        // the middle pair of the new range: a <= T, b >= NP - T keep it inside a 5-slot window around NP / 2
        constexpr int LO1 = (NP - T - 1) >> 1, LO2 = (NP - T) >> 1;
        f.m1 = pick_rel<LO1, T + 1, NP>(v, ((f.a + f.b - 1) >> 1) - LO1);
        f.m2 = pick_rel<LO2, T + 1, NP>(v, ((f.a + f.b) >> 1) - LO2);
    }
    // astropy applies the FINAL bounds to every value: a value trimmed by an earlier pass comes back if it lies inside them.
    // Here: the innermost trimmed value of either side must be surely outside, otherwise the exact path decides.
    if (wave_any(f.a > plo)) {
        const float t = fast32_t(f, pick_rel<0, T, NP>(v, f.a > 0 ? f.a - 1 : 0));       // (T need not be a power of two)
        f.unsure = f.unsure || (f.a > plo && !(t > f.tl_hi));
    }
    if (wave_any(f.b < NP - phi)) {
        const float t = fast32_t(f, pick_rel<NP - T, T, NP>(v, f.b < NP ? f.b - (NP - T) : 0));
        f.unsure = f.unsure || (f.b < NP - phi && !(t > f.th_hi));
    }
    // the final sums (the last pass may have trimmed) and the mean-accuracy guard rms(d) <= |c| / 4 (round 6: was |c| / 2 - columns
    // whose spread is comparable to their level, e.g. frames co-added with very different flux scales, came out 2 ulp from the
    // float64 mean: |dS| / n <= 19u rms(d); found by the `fused` fuzz family, tools/fuzz_long.py.  |c| / 8 was tried first and is
    // too tight: the benchmark's own frames - sky 500 ADU, noise 50-70 ADU towards the vignetted corners, rms(d) / |c| up to 0.15 -
    // sent 1.2 % of C2's and 19 % of C5's pixels to the redo pass, 0.89 -> 1.13 and 5.1 -> 6.0 ms)
    S = (Sc + f.Slo) + f.Shi;
    Q = (Qc + f.Qlo) + f.Qhi;
    f.unsure = f.unsure || !(16.f * Q <= (float)(f.b - f.a) * (cf * cf));
    a_out = f.a;
    b_out = f.b;
    cf_out = cf;
    S_out = S;
    Q_out = Q;
    if constexpr (MODE != 0) return !f.unsure;
    return !wave_any(f.unsure);
}

// Exact clip of a sorted column (float64 moments and tests): the survivors v[a .. b), the pivot c and S, Q about it.
template <int NP, int MINN>
__device__ __forceinline__ void clip_exact(const float (&v)[NP], const int n, const int ns, const double sl2, const double su2,
                                           const int maxiters, const bool use_median, int &a_out, int &b_out, float &cf_out,
                                           double &S_out, double &Q_out)
{
    APGPU_MARK("moments");
    // the two middle values of the finite range: the lower one is the pivot, and together they are the first pass's median
    float m1, m2;
    pick_middle<NP>(v, (n - 1) >> 1, n >> 1, m1, m2);
    float cf = n > 0 ? m1 : 0.f;
    double c = (double)cf;
    // S = sum(x - c), Q = sum((x - c)^2): four independent float64 chains (ILP), fixed association
    double Sa[4] = {0.0, 0.0, 0.0, 0.0}, Qa[4] = {0.0, 0.0, 0.0, 0.0};
    if (wave_all(n == ns)) {               // the usual case: no rejected value in the whole wave (padding: scalar skips)
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (i >= MINN && i >= ns) continue;
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
    st.woff = 0.0;
    st.nn = (double)n;
    st.wscale = (double)n;
    st.Tlo = 0.0;
    st.Thi = 0.0;
    bool active = n > 0;
    int it = 0;
    // S and Q are updated by subtraction: their absolute error stays at the scale of the values they were last summed
    // from (~ 2^-50 n Q).  When the spread of the survivors has collapsed below 2^-22 of that scale - a huge outlier gone,
    // or the clip converging on a handful of near-identical values - the variance n Q - S^2 is no longer good to the ~1e-9
    // the keep / reject decisions need, and the moments are summed afresh about a pivot inside the survivors.
    double refresh_below = ldexp((double)n * Q0, -22);

    // The LAST pass may itself have trimmed a value that dominated the sums (maxiters reached while huge outliers were still
    // being peeled off: five outliers of 1e16 .. 1e29 times the spread with maxiters = 5 - round 3's adversarial test): what
    // the subtraction leaves of S is then rounding noise of the removed value.  So once no lane is active any more the loop
    // makes one more visit (fin) that only repeats the refresh test, on every lane's final range.
    for (;;) {
        const bool fin = !wave_any(active);
        const int a0 = st.a, b0 = st.b;
        const double nn = (double)(st.b - st.a);
        double V = fma(nn, st.Q, -(st.S * st.S));            // n^2 * variance
        if (active) st.nn = nn;
#ifdef APGPU_VARIANT_NO_REFRESH
        const bool refresh = false;
#else
        const bool refresh = (active || fin) && (st.b > st.a) && (V < refresh_below);
#endif
        if (wave_any(refresh)) {                             // rare: see above
            float p1, p2;
            pick_middle<NP>(v, (st.a + st.b - 1) >> 1, (st.a + st.b) >> 1, p1, p2);
            const double cn = refresh ? (double)p1 : st.c;
            double Sn = 0.0, Qn = 0.0;
            // one wave-uniform test per slot even for full stacks: the scalar branches keep this pass as NP short blocks
            // (as one straight-line block the register allocator keeps every converted value alive: 250 VGPRs)
            int nslots = ns;
            asm volatile("" : "+s"(nslots));
#pragma unroll
            for (int i = 0; i < NP; i++) {
                if (i >= nslots) continue;
                const double d0 = widen(v[i]) - cn;
                const double d = (refresh && i >= st.a && i < st.b) ? d0 : 0.0;
                Sn += d;
                Qn = fma(d, d, Qn);
            }
            if (refresh) {
                st.S = Sn;
                st.Q = Qn;
                st.c = cn;
                c = cn;
                cf = p1;
                V = fma(nn, Qn, -(Sn * Sn));
                refresh_below = ldexp(nn * Qn, -22);
            }
        }
        if (fin) break;
        if (active) {
            const double med = 0.5 * ((double)m1 + (double)m2);  // wirth_median (even: mean of the two)
            st.cen = use_median ? med : c;
            st.woff = (use_median || st.nn == 0.0) ? 0.0 : st.S;      // n (x - mean) = n (x - c) - S
            st.wscale = st.nn;
            V = V > 0.0 ? V : 0.0;
            st.Tlo = sl2 * V;
            st.Thi = su2 * V;
        }
        trim_low<0, NP>(v, st, active);
        trim_high<NP - 1, NP, MINN>(v, st, active, ns);
        it++;
        const bool changed = (st.a != a0) || (st.b != b0);
        active = active && changed && (maxiters < 0 || it < maxiters);
        // the middle pair for the next pass (the first pass uses the pair picked for the pivot)
        if (use_median && wave_any(active)) pick_middle<NP>(v, (st.a + st.b - 1) >> 1, (st.a + st.b) >> 1, m1, m2);
    }

    APGPU_MARK("readmit_output");
    // astropy applies the FINAL bounds to all values (sigma_clipping.py:356-358): values trimmed by
    // an earlier, tighter pass that lie inside the final bounds are re-admitted.  The column is sorted, so nothing
    // can come back on a side whose innermost trimmed value (v[a-1] / v[b]) is still outside; for the usual few
    // trimmed values that one element is picked with a 4-way select and the walk over the trimmed slots is skipped.
    if (wave_any(st.a > 0)) {
        bool walk = st.a > 0;
        if constexpr (NP >= 8) if (wave_all(st.a <= 4)) {
            const double xd = widen(pick_rel<0, 4, NP>(v, (st.a - 1) & 3));
            walk = st.a > 0 && (st.a >= st.b || !below(st, xd));
        }
        if (wave_any(walk)) {
            int a_new = st.a;
            readmit_low<0, NP>(v, st, a_new);
            st.a = a_new;
        }
    }
    if (wave_any(st.b < n)) {
        bool walk = st.b < n;
        if constexpr (NP >= 8) {
            if (wave_all(n == NP && st.b >= NP - 4)) {
                const double xd = widen(pick_rel<NP - 4, 4, NP>(v, st.b & 3));     // NP is a multiple of 4: (b - (NP - 4)) & 3
                walk = st.b < n && (st.a >= st.b || !above(st, xd));
            }
        }
        if (wave_any(walk)) {
            int b_new = st.b;
            readmit_high<NP - 1, NP, MINN>(v, st, n, b_new, ns);
            st.b = b_new;
        }
    }
    a_out = st.a;
    b_out = st.b;
    cf_out = cf;
    S_out = st.S;
    Q_out = st.Q;
}

// v[i] = v[i + P] for a wave-uniform P in [0, 3], +inf sentinels moved in at the top: undoes split pads (rare path).
template <int NP>
__device__ __forceinline__ void shift_down(float (&v)[NP], int P)
{
#pragma unroll
    for (int bit = 1; bit <= 2; bit *= 2) {
        if (P & bit) {
#pragma unroll
            for (int i = 0; i < NP; i++) v[i] = (i + bit < NP) ? v[i + bit] : __builtin_inff();
        }
    }
}

// Lean reduction (mean / count / moments outputs, std deviation): the benchmarked path.  Everything after
// the column load is in registers: sort, moments, clipping iterations, outputs.
// pruned (wave-uniform): the column came out of the pruned network (load_sorted_column) - the caller has established
// fast32_wanted(prm) and n == NP for the whole wave; before the exact path may read it the sort is completed.
// allow_fast (wave-uniform): false = the exact clip whatever the arguments say (the list pass of the redo kernel; the column
// was then loaded without split pads - load_sorted_column got the same flag).
template <int NP, int MINN = NP, bool PLUS = false>
__device__ __forceinline__ void reduce_and_store(const StackParams &prm, float (&v)[NP], const int n, const int64_t p,
                                                 const bool pruned = false, const bool allow_fast = true)
{
    const int ns = MINN < NP ? prm.N : NP;                  // wave-uniform number of real frames (slots >= ns: padding)
    // everything the loop and the epilogue need from the kernel arguments: read from the argument block HERE, after the sort
#if APGPU_LATE_PARAMS
    LateParams *const kp = late_params();
    void *const out_moments = kp->moments;
    const int mom64 = kp->moments64;
    const double sl2 = kp->sl2, su2 = kp->su2;
    const int maxiters = kp->maxiters;
    const bool use_median = kp->center == APGPU_CENTER_MEDIAN;
    const int fast32 = kp->fast32;
#else
    float *const out_mean = park_in_vgpr(prm.mean);
    int32_t *const out_count = park_in_vgpr(prm.count);
    void *const out_moments = park_in_vgpr(prm.moments);
    const int mom64 = park_in_vgpr(prm.moments64);
    const int64_t Pn = park_in_vgpr(prm.P);
    const double sl2 = park_in_vgpr(prm.sl2), su2 = park_in_vgpr(prm.su2);
    const int maxiters = park_in_vgpr(prm.maxiters);
    const bool use_median = park_in_vgpr((int)prm.center) == APGPU_CENTER_MEDIAN;
    const int fast32 = park_in_vgpr(prm.fast32);
#endif
    APGPU_MARK("fast32");                                    // v: sorted ascending, sentinels last (load_sorted_column)

    int a, b;
    float cf, Sf = 0.f;
    double S, Q;
    bool done = false;
    int ns_eff = ns;                                        // slots the output planes' loops visit
#ifndef APGPU_VARIANT_NO_FAST32
    if constexpr (fast32_possible_padded(NP, MINN)) {
        // padded stack with split pads (the kernel loaded the column with SPLIT_PADS, same decision from the same arguments):
        // the fast path starts with the pads trimmed; for the exact path the column is moved down onto the -inf pads first
        const int plo = allow_fast ? pad_low<NP>(prm) : 0;
        if (allow_fast && fast32_wanted(prm)) {
            if (pruned) {                                    // (load_sorted_column: every lane of the wave holds all N values)
                float Qf;
                done = clip_fast32<NP, fast_tail_padded(NP)>(v, (float)sl2, (float)su2, maxiters, a, b, cf, Sf, Qf, plo, NP - ns - plo);
                S = (double)Sf;
                Q = (double)Qf;
                if (!done) sort_column<NP>(v);
            }
            if (!done) shift_down<NP>(v, plo);
            else ns_eff = NP;                                // the survivors v[a .. b) sit between the pads: scan every slot
        }
    }
    if constexpr (fast32_possible(NP, MINN)) {
        // float32 fast path (see clip_fast32): full columns, median centre; float64-layout moments carry a sum of squares
        // that callers turn into a std, so they stay on the exact path (the float32 layout is refused a std anyway)
        if (pruned || (allow_fast && use_median && fast32 && (out_moments == nullptr || mom64 == 0 || fast32 == 2) && wave_all(n == NP))) {
            float Qf;
            done = clip_fast32<NP>(v, (float)sl2, (float)su2, maxiters, a, b, cf, Sf, Qf);
            S = (double)Sf;
            Q = (double)Qf;
            if (!done && pruned) sort_column<NP>(v);
        }
    }
#endif
    if (!done) clip_exact<NP, MINN>(v, n, ns, sl2, su2, maxiters, use_median, a, b, cf, S, Q);
    APGPU_MARK("output");
#if APGPU_LATE_PARAMS
    LateParams *const ko = late_params();                    // (a second read: nothing of the first one stays live across the clip)
    float *const out_mean = ko->mean;
    int32_t *const out_count = ko->count;
    const int64_t Pn = ko->P;
#endif
    const double c = (double)cf;

    const int cnt = b - a;
    const double nf = (double)cnt;
    const double nan = __builtin_nan("");
    if (done) {
        // fast path: S is a float32 sum, so the mean is formed in float32 as well - S / n by reciprocal + one residual step
        // (n is a small integer: within an ulp of the quotient, i.e. ~1e-8 of the mean), then ONE rounding in c + S / n.
        // (The float64 division of the exact branch costs ~30 four-cycle instructions per pixel: 2 % of the kernel.)
        const float nf32 = (float)cnt;
        const float y = __builtin_amdgcn_rcpf(nf32);
        const float q0 = Sf * y;
        const float ms32 = __builtin_fmaf(__builtin_fmaf(-nf32, q0, Sf), y, q0);
        if (out_mean) out_mean[p] = cf + ms32;                // cnt >= NP - 2 * kFastTail > 0 here
    } else {
        const double ms = S / nf;                             // mean - c
        if (out_mean) out_mean[p] = cnt > 0 ? (float)(c + ms) : (float)nan;
    }
    if (out_count) out_count[p] = cnt;
    if (out_moments) store_moments(out_moments, mom64, Pn, p, cnt, c, S, Q);
    if constexpr (PLUS) {
        // median and std planes of the final survivors v[a .. b), same definitions as the rich kernel (nanmedian: mean of the
        // two middle values; nanstd: two passes - a column of identical survivors gives exactly 0, which the running S / Q,
        // updated by subtraction, cannot guarantee)
        if (prm.median) {
            float m1, m2;
            pick_middle<NP>(v, (a + b - 1) >> 1, (a + b) >> 1, m1, m2);
            prm.median[p] = cnt > 0 ? (float)(((double)m1 + (double)m2) / 2.0) : (float)nan;
        }
        if (prm.std) {
            // one wave-uniform test per slot even for full stacks: the scalar branches keep the two passes as 64 short blocks
            // (as one straight-line block the register allocator keeps every converted value alive: 256 VGPRs + scratch)
            int nslots = ns_eff;
            asm volatile("" : "+s"(nslots));
            double s1 = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                if (i >= nslots) continue;
                const bool in = (i >= a) && (i < b);
                s1 += widen(in ? v[i] : cf) - c;             // a rejected slot contributes exactly 0
            }
            const double mm = s1 / nf;
            double q1 = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                if (i >= nslots) continue;
                const bool in = (i >= a) && (i < b);
                const double dd = (widen(v[i]) - c) - mm;
                const double d = in ? dd : 0.0;
                q1 = fma(d, d, q1);
            }
            prm.std[p] = cnt > 0 ? (float)sqrt(q1 > 0.0 ? q1 / nf : 0.0) : (float)nan;
        }
    }
}

// Rich reduction: the lean algorithm with (a) mad_std as an alternative deviation, (b) the median and
// std output planes, (c) the sorted column in LDS (see above).  Arithmetic on S / Q is performed in the
// same order as in the lean kernel, so both produce identical mean / count / moments.
// FROM_REGS = false: the sorted column is already in LDS (the big-stack kernel, stack_big.h: N > 128 does not fit the
// registers); v is then a dummy and the moments pass reads LDS as well.
template <int NP, int B = rich_block<NP>(), bool FROM_REGS = true, int NV = NP>
__device__ __forceinline__ void reduce_and_store_rich(const StackParams &prm, float (&v)[NV], const int n, const int64_t p,
                                                      float *const col)
{
    const bool use_median = prm.center == APGPU_CENTER_MEDIAN;
    const bool use_mad = prm.dev == APGPU_DEV_MAD_STD;
    const double sl2 = prm.sl2, su2 = prm.su2;
    const int maxiters = prm.maxiters;
    if constexpr (FROM_REGS) {
#pragma unroll
        for (int i = 0; i < NP; i++) col[i * B] = v[i];      // v: sorted ascending, sentinels last (load_sorted_column)
    }

    // pivot: the lower median of the finite values; S = sum(x - c), Q = sum((x - c)^2) as in the lean kernel
    float cf = n > 0 ? col_read<NP, B>(col, (n - 1) >> 1) : 0.f;
    double c = (double)cf;
    double Sa[4] = {0.0, 0.0, 0.0, 0.0}, Qa[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (FROM_REGS) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const float x = (i < n) ? v[i] : cf;
            const double d = (double)x - c;
            Sa[i & 3] += d;
            Qa[i & 3] = fma(d, d, Qa[i & 3]);
        }
    } else {
        for (int i0 = 0; i0 < NP; i0 += 8) {                  // same association as above: element i feeds chain i & 3
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = col[(i0 + j) * B];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double d = (double)((i0 + j < n) ? x[j] : cf) - c;
                Sa[j & 3] += d;
                Qa[j & 3] = fma(d, d, Qa[j & 3]);
            }
        }
    }
    ClipState st;
    st.S = (Sa[0] + Sa[1]) + (Sa[2] + Sa[3]);
    st.Q = (Qa[0] + Qa[1]) + (Qa[2] + Qa[3]);
    st.c = c;
    st.a = 0;
    st.b = n;
    st.cen = c;
    st.woff = 0.0;
    st.nn = (double)n;
    st.wscale = (double)n;
    st.Tlo = 0.0;
    st.Thi = 0.0;
    // APGPU_STACK_NONFINITE_UNCLIPPED (ccdproc >= 2.2 through astropy.stats.sigma_clip with callables): NaN bounds for a column
    // that holds a non-finite value - nothing is rejected, the outputs are those of its finite values
    bool active = n > 0 && !(prm.unclipped_nonfinite != 0 && n < prm.N);
    int it = 0;
    double refresh_below = ldexp((double)n * st.Q, -22);     // see the lean kernel: when to sum the moments afresh

    for (;;) {
        const bool fin = !wave_any(active);                  // one last visit for the refresh test alone (see clip_exact)
        const int a0 = st.a, b0 = st.b;
        const float m1 = col_read<NP, B>(col, (st.a + st.b - 1) >> 1);
        const float m2 = col_read<NP, B>(col, (st.a + st.b) >> 1);
        const double med = 0.5 * ((double)m1 + (double)m2);  // wirth_median (even: mean of the two)
        double mad = 0.0;
        if (use_mad && !fin) mad = mad_std_window<NP, B>(col, active, st.a, st.b, med);
        const double nn = (double)(st.b - st.a);
        double V = fma(nn, st.Q, -(st.S * st.S));            // n^2 * variance
        if (active) st.nn = nn;
        const bool refresh = (active || fin) && (st.b > st.a) && (V < refresh_below);
        if (wave_any(refresh)) {
            const double cn = refresh ? (double)m1 : st.c;
            double Sn = 0.0, Qn = 0.0;
            constexpr int CH = NP >= 8 ? 8 : NP;
            for (int i0 = 0; i0 < NP; i0 += CH) {
                if (!wave_any(refresh && i0 + CH > st.a && i0 < st.b)) continue;
                float x[CH];
#pragma unroll
                for (int j = 0; j < CH; j++) x[j] = col_read<NP, B>(col, i0 + j);
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    const double d0 = (double)x[j] - cn;
                    const double d = (refresh && i0 + j >= st.a && i0 + j < st.b) ? d0 : 0.0;
                    Sn += d;
                    Qn = fma(d, d, Qn);
                }
            }
            if (refresh) {
                st.S = Sn;
                st.Q = Qn;
                st.c = cn;
                c = cn;
                cf = m1;
                V = fma(nn, Qn, -(Sn * Sn));
                refresh_below = ldexp(nn * Qn, -22);
            }
        }
        if (fin) break;
        if (active) {
            st.cen = use_median ? med : c;
            if (use_mad) {
                st.wscale = 1.0;
                st.woff = (use_median || st.nn == 0.0) ? 0.0 : st.S / st.nn;
                st.Tlo = sl2 * (mad * mad);
                st.Thi = su2 * (mad * mad);
            } else {
                st.wscale = st.nn;
                st.woff = (use_median || st.nn == 0.0) ? 0.0 : st.S;  // n (x - mean) = n (x - c) - S
                V = V > 0.0 ? V : 0.0;
                st.Tlo = sl2 * V;
                st.Thi = su2 * V;
            }
        }
        // trim from the low end, then from the high end: every lane walks its own cursor
        for (;;) {
            const double xd = (double)col_read<NP, B>(col, st.a);
            const bool rej = active && (st.a < st.b) && below(st, xd);
            if (rej) {
                const double d = xd - st.c;
                st.S -= d;
                st.Q = fma(-d, d, st.Q);
                st.a++;
            }
            if (!wave_any(rej)) break;
        }
        for (;;) {
            const double xd = (double)col_read<NP, B>(col, st.b - 1);
            const bool rej = active && (st.a < st.b) && above(st, xd);
            if (rej) {
                const double d = xd - st.c;
                st.S -= d;
                st.Q = fma(-d, d, st.Q);
                st.b--;
            }
            if (!wave_any(rej)) break;
        }
        it++;
        const bool changed = (st.a != a0) || (st.b != b0);
        active = active && changed && (maxiters < 0 || it < maxiters);
    }

    // astropy applies the FINAL bounds to all values (sigma_clipping.py:356-358): values trimmed by an
    // earlier, tighter pass that lie inside the final bounds are re-admitted (ascending, then descending,
    // like the lean kernel's chains).
    if (wave_any(st.a > 0)) {
        int a_new = st.a;
        for (int i = 0; wave_any(i < st.a); i++) {
            const double xd = (double)col_read<NP, B>(col, i);
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
    if (wave_any(st.b < n)) {
        int b_new = st.b;
        for (int i = (prm.N < NP ? prm.N : NP) - 1; wave_any(i >= st.b); i--) {   // slots >= N are padding for every lane
            const double xd = (double)col_read<NP, B>(col, i);
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
    if (prm.mean64) prm.mean64[p] = cnt > 0 ? c + ms : nan;
    if (prm.count) prm.count[p] = cnt;
    if (prm.std || prm.std64) {
        // np.nanstd of the survivors: two passes like numpy (a column of identical survivors must give
        // exactly 0, which the running S/Q - updated by subtraction - cannot guarantee).
        constexpr int CH = NP >= 8 ? 8 : NP;                  // LDS reads in flight per trip
        double s1 = 0.0;
        for (int i0 = 0; i0 < NP; i0 += CH) {
            if (!wave_any(i0 + CH > a && i0 < b)) continue;
            float x[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) x[j] = col_read<NP, B>(col, i0 + j);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const bool in = (i0 + j >= a && i0 + j < b);
                s1 += (double)(in ? x[j] : cf) - c;          // a rejected slot contributes exactly 0
            }
        }
        const double m1 = s1 / nf;
        double q1 = 0.0;
        for (int i0 = 0; i0 < NP; i0 += CH) {
            if (!wave_any(i0 + CH > a && i0 < b)) continue;
            float x[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) x[j] = col_read<NP, B>(col, i0 + j);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const bool in = (i0 + j >= a && i0 + j < b);
                const double dd = ((double)x[j] - c) - m1;
                const double d = in ? dd : 0.0;
                q1 = fma(d, d, q1);
            }
        }
        const double sd = cnt > 0 ? sqrt(q1 > 0.0 ? q1 / nf : 0.0) : nan;
        if (prm.std) prm.std[p] = (float)sd;
        if (prm.std64) prm.std64[p] = sd;
    }
    if (prm.median) {
        const float m1 = col_read<NP, B>(col, (a + b - 1) >> 1);
        const float m2 = col_read<NP, B>(col, (a + b) >> 1);
        prm.median[p] = cnt > 0 ? (float)(((double)m1 + (double)m2) / 2.0) : (float)nan;
    }
    if (prm.moments) store_moments(prm.moments, prm.moments64, prm.P, p, cnt, c, S, Q);
}

}  // namespace apgpu_stack
