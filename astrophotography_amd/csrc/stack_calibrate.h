// stack_calibrate.h - kernel parameters, per-frame scalars, column loads and the fused calibration (core/ApCalibrate.py:439-464) of the stack kernels.
#pragma once
#include "stack_sort.h"

namespace apgpu_stack {

using namespace apgpu;

struct StackParams {
    const void *frames;
    const float *bias, *dark, *nflat, *exp_ratio, *pedestal;
    const uint8_t *pixmask;
    float *mean, *median, *std;
    void *moments;          // float[3][P] (sum, count, sumsq) or, with moments64, double[2][P] (sum, sumsq) + int32[P] (count)
    int32_t *count;
    int64_t P;
    int64_t stride;         // elements between frames
    double sl2, su2;        // sigma_lower^2, sigma_upper^2
    int N;
    int still_biased;
    int center;             // 0 median, 1 mean
    int dev;                // 0 std, 1 mad_std (EXTRA kernels only)
    int maxiters;           // < 0: until convergence
    int moments64;          // layout of `moments` (see above)
    double *mean64, *std64; // float64 output planes (rich kernels only): ccdproc.combine writes float64 (ap_combine_darks.py:437)
    int single_kernel;      // host side only: never the fast kernel + redo pass pair (APGPU_STACK_SINGLE_KERNEL)
    int fast32;             // 0: float64 clip only; 1: float32 fast path for mean / count / float32 moments; 2: also float64-layout moments
    int flag_mode;          // rich kernels only (EXTRA): 1 = behind stack_mad_fast_kernel - a wavefront reduces its 64-pixel block only if
                            // the block's flag in `redo` is set, and clears it (stack_mad.hip)
    int32_t *redo;          // the workspace of the two-kernel scheme (stack_kernels.h, "workspace layout"): segment counters,
                            // tile flags and per-segment lists of the PIXELS the fast kernel could not finish.  NULL in a
                            // plain launch of the complete kernels (stack_sigclip_kernel: non-NULL = redo pass)
    int unclipped_nonfinite; // rich kernels only: a column holding a non-finite value is not clipped (APGPU_STACK_NONFINITE_UNCLIPPED)
};

// The slot counts the dispatcher uses (launch_np) and, for each, the largest N that still selects the previous one: a
// stack that runs with NP slots has more than prev_slots(NP) frames, so only the slots from there on can be padding.
// Padded stacks of 112 / 120 / 128 slots (round 4: 112 and 120 too - with the skipping scheme they take 258 VGPRs, one wavefront
// per SIMD) keep the older scheme (every slot loaded and calibrated, padding lifted to +inf with
// one v_max per slot): the per-slot scalar tests cost them ~50 VGPRs, i.e. the second wavefront per SIMD.
constexpr int prev_slots(int np);
constexpr int padded_minn(int np, bool full) { return (full || np >= 112) ? np : prev_slots(np); }

constexpr int prev_slots(int np)
{
    constexpr int counts[] = {1, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128};
    int prev = 0;
    for (int c : counts) {
        if (c >= np) break;
        prev = c;
    }
    return prev;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(uint16_t x) { return (float)x; }

// x / nf for a per-pixel divisor with y = RN(1 / nf) and yl = fma(-nf, y, 1) * y (the reciprocal's rounding error: 1 / nf =
// y + yl to 2^-48) precomputed.  q0 = RN(x y + RN(x yl)) is a FAITHFUL quotient (the true one to 2^-47 before its rounding);
// r = x - nf q0 is exact in an FMA; q = RN(q0 + r y) is then the correctly rounded x / nf (Markstein's theorem: correctly
// rounded reciprocal + faithful quotient + exact residual), provided nothing over/underflows - the caller guards the ranges
// and falls back to IEEE division.  4 instructions (rounds 1-3: RN(x y) - only within 1.5 ulp - and TWO residual steps, 5)
// instead of the 12 of the IEEE sequence (v_div_scale x2, v_rcp, 6 FMA, v_div_fmas, v_div_fixup), 64 times per pixel;
// tools/div_check.hip: 1.3e11 random, special-divisor and next-to-a-rounding-boundary operand pairs, no mismatch with __fdiv_rn.
__device__ __forceinline__ float recip_low(float nf, float y) { return __builtin_fmaf(-nf, y, 1.0f) * y; }

__device__ __forceinline__ float div_by_recip(float x, float nf, float y, float yl)
{
    const float q0 = __builtin_fmaf(x, y, x * yl);
    const float r0 = __builtin_fmaf(-nf, q0, x);
    return __builtin_fmaf(r0, y, q0);
}

// Stacks without pedestals read the exposure ratios with scalar loads (calibrate_fast, E_DIRECT; padding slots through a
// clamped index) and stage nothing in LDS.
#ifndef APGPU_DIRECT_RATIOS
#define APGPU_DIRECT_RATIOS 1
#endif
#ifndef APGPU_DIRECT_RATIOS_PADDED
#define APGPU_DIRECT_RATIOS_PADDED 1
#endif
constexpr bool direct_ratios(bool) { return APGPU_DIRECT_RATIOS != 0; }

// Whether a kernel of the given kind has to stage the per-frame scalars in LDS (wave-uniform, from the arguments):
// pedestals, and the padded 128-slot kernels' pad vector (the lift scheme of load_column, MINN >= NP).
template <bool CALIB, bool FULL, int NP = 0>
__device__ __forceinline__ bool needs_staging(const StackParams &prm)
{
    if constexpr (!CALIB) return false;
    else if constexpr (!FULL && (NP >= 112 || !APGPU_DIRECT_RATIOS_PADDED)) return true;
    else return !direct_ratios(FULL) || prm.pedestal != nullptr;
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

// The same in two steps (round 4): stage_fetch issues the global loads of this thread's slot, stage_commit writes LDS and
// synchronises - so that a kernel can put its column loads in between and does not pay two memory round trips in a row
// (the first wavefront's loads of the ratios, the barrier, and only then every wavefront's frame loads).  NP <= blockDim.x.
template <int NP>
__device__ __forceinline__ void stage_fetch(const StackParams &prm, float &e, float &ped)
{
    const int t = threadIdx.x;
    e = 0.f;
    ped = 0.f;
    if (t < NP) {
        const int ff = t < prm.N ? t : prm.N - 1;
        if (prm.exp_ratio) e = prm.exp_ratio[ff];
        if (prm.pedestal) ped = prm.pedestal[ff];
    }
}

template <int NP>
__device__ __forceinline__ void stage_commit(const StackParams &prm, FrameScalars<NP> &fs, float e, float ped)
{
    const int t = threadIdx.x;
    if (t < NP) {
        fs.e[t] = e;
        fs.ped[t] = ped;
        fs.pad[t] = t < prm.N ? -__builtin_inff() : __builtin_inff();
    }
    __syncthreads();
}

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

// F0 / CNT: the slots F0 .. F0+CNT-1 of the column (large slot counts are loaded and calibrated in two halves so that
// only half of the raw values sit in registers next to v[]); F0 > 0 is a real frame for every N that selects this NP.
// MINN: slots below MINN are real frames for every N this instantiation serves (prev_slots; 0 = no such knowledge).
// NSTATIC > 0: the frame count is this compile-time value (kernels instantiated per pad count): the tests below fold.
#ifndef APGPU_PIN_RAGGED_F32
#define APGPU_PIN_RAGGED_F32 0      // a translation unit may set it: the same pin for float32 frames (fused calibration sinks likewise)
#endif
template <int NP, typename RawT, bool FULL, int F0 = 0, int CNT = NP, int MINN = 0, int NSTATIC = 0>
__device__ __forceinline__ void load_raw(const StackParams &prm, int64_t base, int lane, RawT (&raw)[CNT])
{
    // Wave-uniform frame pointer (SGPR pair) + per-lane offset: one coalesced row segment per frame.
    const RawT *fb = static_cast<const RawT *>(prm.frames) + base + (int64_t)F0 * prm.stride;
    // opaque per call: the fast path and its (rare) exact fallback each load the column; sharing the NP
    // clamped address steps between the two calls would keep 2*NP SGPRs live across the calibration
    int nframes = NSTATIC > 0 ? NSTATIC : prm.N;
    if constexpr (!FULL && NSTATIC == 0) asm volatile("" : "+s"(nframes));
#pragma unroll
    for (int f = 0; f < CNT; f++) {
        constexpr bool SKIP = !FULL && MINN < NP;           // padding slots are not loaded (wave-uniform test); without
        if (!SKIP || F0 + f < MINN || F0 + f < nframes) raw[f] = fb[lane];      // SKIP they re-read the last frame (cache hit)
        else raw[f] = RawT(0);
        if (FULL || (SKIP && F0 + f + 1 < MINN) || F0 + f + 1 < nframes) fb += prm.stride;
        // fence: otherwise the scheduler materialises all NP frame addresses (2 SGPRs each) at once
        if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    // uint16 frames, ragged stack: the compiler sinks the caller's uint16 -> float32 conversion into each slot's branch, next to
    // the load, and then has to wait for that load there - s_waitcnt vmcnt(0) once per conditional slot, one exposed memory latency
    // each (round 6, read off the ISA of the 129..512-frame kernels, whose 128-slot chunks have 128 such slots: the exact kernel took
    // 73 ms for 256 uint16 frames where float32 frames took 27).  Pinning the raw values HERE keeps the conversions behind the loads.
    if constexpr ((sizeof(RawT) == 2 || APGPU_PIN_RAGGED_F32) && !FULL && MINN < NP && NSTATIC == 0) {
#pragma unroll
        for (int f = 0; f < CNT; f++) asm volatile("" : "+v"(raw[f]));
    }
}

// Fast calibration of a full column (N == NP): reciprocal division, no per-value fix-ups.  Returns
// true if the lane's results are exact AND all finite; otherwise the wave redoes the column exactly.
// Frames are processed in pairs with 2-wide vector arithmetic so that the backend emits the packed
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 forms: the kernel is bound by VALU issue slots (one
// wave64 instruction per 4 cycles per SIMD) and a packed instruction retires two values per slot.
// Each lane of a packed operation is an ordinary IEEE float32 operation, so results do not change.
typedef float v2f __attribute__((ext_vector_type(2)));

// RANGE_GUARD = false: the quotient-range test is left to the caller, who reads it off the ends of the SORTED column
// (range_ok_sorted) instead of tracking max |q| and min |q| through the loop (one v_max3 + one v_min3 per frame pair).
// UNI_E (round 4): every frame has the exposure ratio fs.e[0] (the caller's wave vote over the staged scalars,
// exposures_uniform) - the dark term e * D is formed once per pixel instead of once per frame pair (one v_pk_mul_f32 less
// per pair; round 2 tried it with a workgroup vote and a run-time select and found nothing - as a template parameter
// behind one ballot per wave it removes 31 of the 281 calibration instructions of the 64-frame kernel).
// nframes / MINN (padded stacks): groups of 8 slots that are entirely padding (>= nframes, wave-uniform) are skipped;
// every padding slot ends as the +inf sentinel.  FULL stacks pass nframes = NP, MINN = NP: nothing of this remains.
// E_DIRECT (round 4; full stacks without pedestals): the exposure ratios are read straight from the caller's array `eg` - a
// wave-uniform address with a constant offset, i.e. scalar loads into SGPR pairs that v_pk_mul_f32 takes as they are - instead
// of from the LDS copy: no staging pass, no barrier before the first frame load, no ds_read in the calibration.
template <int NP, typename RawT, bool HAS_PED, int F0 = 0, int CNT = NP, bool RANGE_GUARD = true, int MINN = NP, bool UNI_E = false,
          bool E_DIRECT = false>
__device__ __forceinline__ bool calibrate_fast(const FrameScalars<NP> &fs, const RawT (&raw)[CNT], float b, float D, float nf,
                                               bool dodiv, float (&v)[NP], int nframes = NP, int plo = 0, const float *eg = nullptr)
{
    // lanes that do not divide (no flat / nflat == 0) run the same code with a divisor of exactly 1:
    // q0 = x, r0 = 0, ... -> x, bit for bit; no per-value select
    const float nfe = dodiv ? nf : 1.0f;
    const float y = __fdiv_rn(1.0f, nfe);
    const float yl = recip_low(nfe, y);
    const float anf = fabsf(nfe);
    const bool nf_ok = (anf >= 0x1p-40f && anf <= 0x1p40f);
    float mx = 0.f, mn = __builtin_inff();
    if constexpr (NP >= 2) {
        v2f acc = {0.f, 0.f};
        const v2f b2 = {b, b}, D2 = {D, D}, nf2 = {-nfe, -nfe}, y2 = {y, y}, yl2 = {yl, yl}, zero2 = {0.f, 0.f};
        float e0 = 0.f;
        if constexpr (UNI_E) {
            if constexpr (E_DIRECT) e0 = ((const float __attribute__((address_space(4))) *)(uintptr_t)eg)[0];
            else e0 = fs.e[0];
        }
        const float ds0 = e0 * D;                            // :450 for every frame when UNI_E
        const v2f ds_uni = {ds0, ds0};
#pragma unroll
        for (int g = 0; g < CNT; g += 2) {
            const int f = F0 + g;
            if ((f & ~7) >= MINN && (f & ~7) >= nframes) continue;      // a whole 8-slot group of padding (scalar test)
            v2f x = {to_f32(raw[g]), to_f32(raw[g + 1])};
            if constexpr (HAS_PED) {
                const v2f ped = {fs.ped[f], fs.ped[f + 1]};
                x = x + ped;                                 // ApCalibrate.py:318-326; a zero pedestal adds +0.0,
            }                                                // which changes nothing but the sign of a -0.0 input
            x = x - b2;                                      // :439
            v2f ds = ds_uni;
            if constexpr (UNI_E) {
            } else if constexpr (E_DIRECT) {
                // constant address space: the array is not written while the kernel runs, and only such (or provably
                // unclobbered) uniform loads are selected as scalar loads - behind the staging branch's barrier these are not
                typedef const float __attribute__((address_space(4))) cfloat;
                const cfloat *ec = (const cfloat *)(uintptr_t)eg;
                // (padding slots - f >= MINN can be one - read the last frame's ratio: the array has nframes entries)
                const int f0 = (f < MINN || f < nframes) ? f : nframes - 1;
                const int f1 = (f + 1 < MINN || f + 1 < nframes) ? f + 1 : nframes - 1;
                const v2f e2 = {ec[f0], ec[f1]};
                ds = e2 * D2;                                // :450
            } else {
                const v2f e2 = {fs.e[f], fs.e[f + 1]};
                ds = e2 * D2;                                // :450
            }
            x = x - ds;                                      // :451
            const v2f q0 = __builtin_elementwise_fma(x, y2, x * yl2);    // :462-464: faithful quotient (div_by_recip)
            const v2f r0 = __builtin_elementwise_fma(nf2, q0, x);
            const v2f q = __builtin_elementwise_fma(r0, y2, q0);
            v[f] = q.x;
            v[f + 1] = q.y;
            acc = __builtin_elementwise_fma(q, zero2, acc);  // NaN iff some value is not finite
            if constexpr (RANGE_GUARD) {
                mx = fmaxf(fmaxf(mx, fabsf(q.x)), fabsf(q.y));
                mn = fminf(fminf(mn, fabsf(q.x)), fabsf(q.y));
            }
        }
        if constexpr (MINN < NP) {
#pragma unroll
            for (int g = 0; g < CNT; g++)
                if (F0 + g >= MINN && F0 + g >= nframes)                                   // scalar tests per slot, one v_mov per pad;
                    v[F0 + g] = (F0 + g < nframes + plo) ? -__builtin_inff() : __builtin_inff();   // the first plo pads sort to the bottom (split pads)
        }
        const bool range_ok = !RANGE_GUARD || !dodiv || (mx < 0x1p50f && mn > 0x1p-50f);
        return nf_ok && range_ok && (acc.x == 0.f) && (acc.y == 0.f);
    } else {
        float x = to_f32(raw[0]);
        if constexpr (HAS_PED) x = x + fs.ped[0];
        x = x - b;
        const float ds = (E_DIRECT ? eg[0] : fs.e[0]) * D;
        x = x - ds;
        const float q = div_by_recip(x, nfe, y, yl);
        v[0] = q;
        const float acc = __builtin_fmaf(q, 0.0f, 0.0f);
        const bool range_ok = !dodiv || (fabsf(q) < 0x1p50f && fabsf(q) > 0x1p-50f);
        return nf_ok && range_ok && (acc == 0.f);
    }
}

// The quotient-range condition of the reciprocal division (all |q| in (2^-50, 2^50), see calibrate_fast), read off a column
// that is already sorted ascending and finite: max |q| sits at an end; min |q| too unless the signs are mixed - then (and
// only then, wave-wide) the magnitudes are scanned.
// v[idx] for a WAVE-UNIFORM idx in [LO, NP): scalar compare-and-branch chain, one v_mov executed.
template <int LO, int NP>
__device__ __forceinline__ float uniform_pick(const float (&v)[NP], int idx)
{
    if constexpr (LO >= NP - 1) return v[NP - 1];
    else {
        if (idx <= LO) return v[LO];
        return uniform_pick<LO + 1, NP>(v, idx);
    }
}

// nvalid (wave-uniform): the column's finite values occupy v[0 .. nvalid), +inf sentinels follow; MINN <= nvalid.
// plo (wave-uniform, split pads): the column starts with plo -inf sentinels; the finite values occupy v[plo .. plo + nvalid).
template <int NP, int MINN = NP>
__device__ __forceinline__ bool range_ok_sorted(const float (&v)[NP], bool dodiv, int nvalid = NP, int plo = 0)
{
    float lo = v[0];
    float hi;
    if constexpr (MINN >= NP) hi = v[NP - 1];
    else {
        if (plo > 0) lo = uniform_pick<0, NP>(v, plo);
        hi = uniform_pick<(MINN > 0 ? MINN - 1 : 0), NP>(v, plo + __builtin_amdgcn_readfirstlane(nvalid) - 1);
    }
    const bool big_ok = fmaxf(fabsf(lo), fabsf(hi)) < 0x1p50f;
    const bool one_sign = lo > 0x1p-50f || hi < -0x1p-50f;    // then min |q| = |lo| or |hi| and it is above 2^-50
    bool small_ok = one_sign;
    if (wave_any(dodiv && !one_sign)) {
        float mn = __builtin_inff();                        // (+inf sentinels of padding slots do not disturb a minimum)
        if constexpr (NP >= 2) {
#pragma unroll
            for (int i = 0; i < NP; i += 2) mn = fminf(fminf(mn, fabsf(v[i])), fabsf(v[i + 1]));
        } else {
            mn = fabsf(v[0]);
        }
        small_ok = mn > 0x1p-50f;
    }
    return !dodiv || (big_ok && small_ok);
}

// A column's loads issued EARLY (round 4): the per-pixel masters first - the calibration's first instruction needs them, and
// the memory counter retires in order: issued after the 64 frame loads (rounds 1-3) they held the whole calibration back until
// the last frame had arrived - then the frames.  Issued before the frame-scalar barrier (stage_commit) by the kernels that can.
template <int NP, typename RawT>
struct EarlyLoads {
    RawT raw[NP];
    float b, d, nf;
    bool skip;
};

template <int NP, typename RawT, bool CALIB, bool FULL, int MINN, int NSTATIC = 0>
__device__ __forceinline__ void issue_early_loads(const StackParams &prm, int64_t base, int lane, EarlyLoads<NP, RawT> &L)
{
    const int64_t p = base + lane;
    L.b = 0.f; L.d = 0.f; L.nf = 1.f;
    if constexpr (CALIB) {
        L.b = prm.bias[p];
        L.d = prm.dark[p];
        if (prm.nflat) L.nf = prm.nflat[p];
    }
    L.skip = prm.pixmask && prm.pixmask[p];
    __builtin_amdgcn_sched_barrier(0);
    load_raw<NP, RawT, FULL, 0, NP, MINN, NSTATIC>(prm, base, lane, L.raw);
}

// Wave vote over the staged per-frame scalars: every frame has the exposure ratio of frame 0 (a NaN ratio fails the test and
// takes the per-frame path).  One or two LDS reads, one compare and one ballot per wave.
template <int NP>
__device__ __forceinline__ bool exposures_uniform(const FrameScalars<NP> &fs)
{
    bool same = true;
    for (int t = threadIdx.x & 63; t < NP; t += 64) same = same && (fs.e[t] == fs.e[0]);
    return wave_all(same);
}

// The same vote on the caller's array itself (kernels that read the ratios with scalar loads and stage nothing): scalar loads
// and scalar compares only - no VALU instruction.
template <int NP>
__device__ __forceinline__ bool ratios_uniform(const float *eg, int nframes)
{
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat *ec = (const cfloat *)(uintptr_t)eg;
    const int e0 = __builtin_amdgcn_readfirstlane(__float_as_int(ec[0]));
    bool same = true;
#pragma unroll
    for (int f = 1; f < NP; f++) same = same && (f >= nframes || __builtin_amdgcn_readfirstlane(__float_as_int(ec[f < nframes ? f : 0])) == e0);
    return same;
}

// Per-lane context of a column load: what the exact fallback and the deferred range check need.
struct ColumnCtx {
    float b, D, nf;
    bool dodiv, skip;
    bool range_pending;     // wave-uniform: the fast path ran without range guards; verify with range_ok_sorted after the sort
};

// The exact column: IEEE division, one frame at a time (re-read from memory: no second raw[] column in flight), non-finite
// values (sigma clip) or NaNs (plain median), padding slots and masked pixels become +inf sentinels.  Returns the number of
// valid values.  RAWREG: take the raw values from `raw` (already in registers, !CALIB) instead of re-reading them.
template <int NP, typename RawT, bool CALIB, bool FINITE_ONLY, bool FULL, int NRAW>
__device__ __forceinline__ int load_column_exact(const StackParams &prm, const FrameScalars<NP> &fs, int64_t p, const ColumnCtx &cx,
                                                 const RawT (&raw)[NRAW], float (&v)[NP], int plo = 0)
{
    const int N = prm.N;
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
            x = to_f32(raw[f < NRAW ? f : 0]);
        }
        if constexpr (CALIB) {
            // (from the caller's arrays, not the LDS copy: kernels that read the ratios directly never stage it; rare path)
            typedef const float __attribute__((address_space(4))) cfloat;       // (scalar loads: see calibrate_fast)
            const int ff = (FULL || f < N) ? f : N - 1;
            const float e = prm.exp_ratio ? ((const cfloat *)(uintptr_t)prm.exp_ratio)[ff] : 0.f;
            const float ped = prm.pedestal ? ((const cfloat *)(uintptr_t)prm.pedestal)[ff] : 0.f;
            if (ped != 0.f) x = x + ped;                     // ApCalibrate.py:318-326
            x = x - cx.b;                                    // :439
            const float ds = e * cx.D;                       // :450
            x = x - ds;                                      // :451
            if (cx.dodiv) x = __fdiv_rn(x, cx.nf);           // :463
        }
        bool ok;
        if constexpr (FINITE_ONLY) ok = fabsf(x) < __builtin_inff();
        else ok = (x == x);
        ok = ok && (FULL || f < N) && !cx.skip;
        n += ok ? 1 : 0;
        v[f] = ok ? x : __builtin_inff();
        if (!FULL && f >= N && f < N + plo) v[f] = -__builtin_inff();      // split pads (wave-uniform test)
    }
    return n;
}

// Loads the lane's column, applies the fused calibration, maps non-finite values (sigma clip) or
// NaNs (plain median) to the +inf sentinel and returns the number of valid values.
// FULL = the stack has exactly NP frames: no padding logic at all (no clamped frame indices, no
// wave-wide (f < N) masks - NP of those cost 2 SGPRs each and end up spilled to VGPR lanes), and the range guards of the
// reciprocal division are deferred to the sorted column (cx.range_pending, see load_sorted_column).
template <int NP, typename RawT, bool CALIB, bool FINITE_ONLY, bool FULL, int MINN = padded_minn(NP, FULL), bool FORCE_HALVES = false, bool PRE = false>
__device__ __forceinline__ int load_column(const StackParams &prm, const FrameScalars<NP> &fs, int64_t base, int lane,
                                           float (&v)[NP], ColumnCtx &cx, int plo = 0, const EarlyLoads<NP, RawT> &pre = EarlyLoads<NP, RawT>())
{
    const int N = prm.N;
    const int64_t p = base + lane;
#ifndef APGPU_HALVES_MIN
#define APGPU_HALVES_MIN 104
#endif
// (measured, round 4: with the hoisted-dark body next to the per-frame one the 64-slot kernel needs 177 VGPRs - two
// wavefronts per SIMD, 1.11 ms - or, capped at 168 by its launch bounds, 4 spilled VGPRs = a scratch allocation per wave:
// 1.635 instead of 1.662 instructions per wave but 1.5-4.7 % SLOWER on two boxes.  Off for the one-pixel-per-lane kernels;
// the uint16 pair kernels, whose precondition is one exposure ratio, use UNI_E unconditionally.)
#ifndef APGPU_HOIST_DARK
#define APGPU_HOIST_DARK 0
#endif
    constexpr bool HALVES = CALIB && (NP >= APGPU_HALVES_MIN || FORCE_HALVES);  // 104 .. 128 slots: two half columns (register budget: 2 waves/SIMD)
    // the range guards are read off the sorted column (load_sorted_column) - except for the largest slot counts, where
    // keeping the lane's masters alive across the sort would push the kernel over 256 VGPRs (one wavefront per SIMD)
    constexpr bool GUARD = NP >= 104;
    // exposure ratios by scalar loads - except in the padded kernels of the lift scheme (MINN >= NP), whose padding slots are
    // calibrated like frames: they keep the staged copy with its clamped frame index
    constexpr bool EDIR = direct_ratios(FULL) && (FULL || (APGPU_DIRECT_RATIOS_PADDED && MINN < NP));
    // PRE: the loads were issued by the caller (`pre`); otherwise here
    static_assert(!(PRE && HALVES), "half-column kernels load their own halves");
    EarlyLoads<NP, RawT> here;
    if constexpr (PRE) {
    } else if constexpr (!HALVES) {
        issue_early_loads<NP, RawT, CALIB, FULL, MINN>(prm, base, lane, here);
    } else {
        here.b = 0.f; here.d = 0.f; here.nf = 1.f;
        if constexpr (CALIB) {
            here.b = prm.bias[p];
            here.d = prm.dark[p];
            if (prm.nflat) here.nf = prm.nflat[p];
        }
        here.skip = prm.pixmask && prm.pixmask[p];
    }
    const EarlyLoads<NP, RawT> &L = PRE ? pre : here;
    const RawT (&raw)[NP] = L.raw;
    cx.b = 0.f; cx.D = 0.f; cx.nf = 1.f;
    cx.dodiv = false;
    cx.range_pending = false;
    if constexpr (CALIB) {
        cx.b = L.b;
        cx.D = prm.still_biased ? L.d - cx.b : L.d;          // ApCalibrate.py:440-445
        if (prm.nflat) {
            cx.nf = L.nf;
            cx.dodiv = (cx.nf != 0.f);                       // ApCalibrate.py:462 (NaN != 0 is True)
        }
    }
    cx.skip = L.skip;
    if constexpr (CALIB) {
        const float b = cx.b, D = cx.D, nf = cx.nf;
        const bool dodiv = cx.dodiv;
        bool good;
        if constexpr (HALVES) {
            constexpr int HN = NP / 2;
            RawT half[HN];
            load_raw<NP, RawT, FULL, 0, HN, MINN>(prm, base, lane, half);
            good = prm.pedestal ? calibrate_fast<NP, RawT, true, 0, HN, GUARD, MINN>(fs, half, b, D, nf, dodiv, v, N, plo)
                                : calibrate_fast<NP, RawT, false, 0, HN, GUARD, MINN, false, EDIR>(fs, half, b, D, nf, dodiv, v, N, plo, prm.exp_ratio);
            load_raw<NP, RawT, FULL, HN, HN, MINN>(prm, base, lane, half);
            const bool good2 = prm.pedestal ? calibrate_fast<NP, RawT, true, HN, HN, GUARD, MINN>(fs, half, b, D, nf, dodiv, v, N, plo)
                                            : calibrate_fast<NP, RawT, false, HN, HN, GUARD, MINN, false, EDIR>(fs, half, b, D, nf, dodiv, v, N, plo, prm.exp_ratio);
            good = good && good2;
        } else {
            if (prm.pedestal) good = calibrate_fast<NP, RawT, true, 0, NP, GUARD, MINN>(fs, raw, b, D, nf, dodiv, v, N, plo);
            else if (APGPU_HOIST_DARK && FULL && EDIR && ratios_uniform<NP>(prm.exp_ratio, NP)) good = calibrate_fast<NP, RawT, false, 0, NP, GUARD, MINN, true, true>(fs, raw, b, D, nf, dodiv, v, N, plo, prm.exp_ratio);
            else good = calibrate_fast<NP, RawT, false, 0, NP, GUARD, MINN, false, EDIR>(fs, raw, b, D, nf, dodiv, v, N, plo, prm.exp_ratio);
        }
        if (wave_all(good && !cx.skip)) {
            cx.range_pending = !GUARD;
            if constexpr (!FULL && MINN >= NP) {
                // padding slots hold a calibrated copy of the last frame: lift them to the +inf sentinel with
                // one v_max against the staged pad vector (with MINN < NP calibrate_fast wrote the sentinels itself)
#pragma unroll
                for (int f = 0; f < NP; f++) asm("v_max_f32 %0, %1, %2" : "=v"(v[f]) : "v"(v[f]), "v"(fs.pad[f]));
            }
            return FULL ? NP : N;
        }
        // rare: a non-finite value, a masked pixel or an out-of-range operand somewhere in the wave:
        // redo the column exactly
    }
    return load_column_exact<NP, RawT, CALIB, FINITE_ONLY, FULL>(prm, fs, p, cx, raw, v, plo);
}

// Slot counts / padding schemes for which the float32 fast path exists: full columns with a core between two tails.
constexpr int kFastTail = 4;
// (up to 96 slots: with the fast path next to the exact one the 104 .. 128-slot kernels need more than 256 VGPRs, i.e. one
// wavefront per SIMD - 128 frames 2.7 -> 3.9 ms; such stacks take the chunked kernel, stack_chunks.hip, when they qualify)
constexpr bool fast32_possible(int np, int minn) { return np >= 16 && np <= 96 && np % 4 == 0 && minn >= np; }
// Padded stacks (N between two slot counts, at most 7 padding slots): the same fast path with tails of 8 and SPLIT PADS - the
// first (NP - N) / 2 padding slots are -inf, the others +inf, so that after the sort the real values sit in the middle of the
// column: the median keeps its static position, the mirror pairs of the core stay mirror pairs, and the pads are simply the
// first values "already trimmed" from either tail (clip_fast32 starts with a = pads below, b = NP - pads above).
constexpr int kFastTailPadded = 8;
// (round 4: with slot counts at every multiple of 4 up to 64 a padded stack there has at most 3 pads - one below, two above
// the real values - so tails of 6 leave four trimmable values per side: a smaller pruned network and 8 table registers less)
constexpr int fast_tail_padded(int np) { return np <= 64 ? 6 : kFastTailPadded; }
constexpr bool fast32_possible_padded(int np, int minn) { return np >= 16 && np <= 96 && np % 4 == 0 && minn < np && np - minn <= 8; }

// Slot counts of stack_fast_kernel (stack_kernels.h): the fast path ALONE fits the registers up to 128 slots - 104 .. 128 with the
// raw column loaded in two halves - where stack_sigclip_kernel, which holds the exact path next to it, stops at 96.  Padded
// stacks always skip their padding slots there (the lift scheme of padded_minn is the complete kernels' register budget).
constexpr bool fast_kernel_slots(int np) { return np >= 16 && np <= 128 && np % 4 == 0; }
constexpr int fast_kernel_minn(int np, bool full) { return full ? np : prev_slots(np); }
constexpr int fast_kernel_tail(int np, bool full) { return full ? kFastTail : fast_tail_padded(np); }

// Whether the lean reduction will try its float32 fast path (stack_reduce.h, clip_fast32) - wave-uniform, from the arguments.
__device__ __forceinline__ bool fast32_wanted(const StackParams &prm)
{
#ifdef APGPU_VARIANT_NO_FAST32
    return false;
#endif
    return prm.fast32 != 0 && prm.center == APGPU_CENTER_MEDIAN && (prm.moments == nullptr || prm.moments64 == 0 || prm.fast32 == 2);
}

// Padding slots of a padded stack that go to the BOTTOM of the column (wave-uniform, from the arguments alone).
template <int NP>
__device__ __forceinline__ int pad_low(const StackParams &prm)
{
    return fast32_wanted(prm) ? (NP - prm.N) >> 1 : 0;
}

// Column loaded AND sorted ascending (sentinels last).  When the fast calibration deferred its range guards, they are
// evaluated on the sorted column; a failing lane sends the wave through the exact path and a second sort (rare).
// PRUNE_T > 0 and *pruned on entry (wave-uniform: the caller wants the fast path): a column without sentinels in the whole
// wave is sorted with the pruned network (its ends and middle window only, make_pruned_net) and *pruned stays true;
// otherwise the sort is complete and *pruned is cleared.
template <int NP, typename RawT, bool CALIB, bool FINITE_ONLY, bool FULL, int MINN = padded_minn(NP, FULL), int PRUNE_T = 0,
          bool FORCE_HALVES = false, bool SPLIT_PADS = false, bool PRE = false>
__device__ __forceinline__ int load_sorted_column(const StackParams &prm, const FrameScalars<NP> &fs, int64_t base,
                                                  int lane, float (&v)[NP], bool *pruned = nullptr, const EarlyLoads<NP, RawT> &pre = EarlyLoads<NP, RawT>(),
                                                  const bool allow_fast = true)
{
    ColumnCtx cx;
    // SPLIT_PADS (padded stacks headed for the float32 fast path, fast32_possible_padded): the first plo padding slots become
    // -inf, the rest +inf; the caller (reduce_and_store) knows - from the same arguments - and undoes it for the exact path.
    // allow_fast (wave-uniform; false in the list pass of the redo kernel, which is exact only): no split, every pad +inf.
    int plo = 0;
    if constexpr (SPLIT_PADS) plo = allow_fast ? pad_low<NP>(prm) : 0;
    int n = load_column<NP, RawT, CALIB, FINITE_ONLY, FULL, MINN, FORCE_HALVES, PRE>(prm, fs, base, lane, v, cx, plo, pre);
    bool prune = false;
    if constexpr (PRUNE_T > 0) prune = *pruned && wave_all(n == (FULL ? NP : prm.N));
    if constexpr (PRUNE_T > 0) {
        if (prune) sort_column<NP, PRUNE_T>(v);
        else sort_column<NP>(v);
    } else {
        sort_column<NP>(v);
    }
    if constexpr (CALIB) {
        // (the range test reads the two ends of the column - sorted by the pruned network too - and, for columns of mixed
        // sign, scans all magnitudes, in any order)
        if (cx.range_pending && !wave_all(range_ok_sorted<NP, MINN>(v, cx.dodiv, n, plo))) {
            RawT none[1] = {};
            n = load_column_exact<NP, RawT, CALIB, FINITE_ONLY, FULL>(prm, fs, base + lane, cx, none, v, plo);
            sort_column<NP>(v);
            prune = false;
        }
    }
    if constexpr (PRUNE_T > 0) *pruned = prune;
    return n;
}

}  // namespace apgpu_stack
