// stack_mad.hip - the ccdproc.combine configuration (A6) on a float32 fast path (round 5).
//
// scripts/ap_combine_darks.py:394-420 - the reference's only in-tree stack - calls ccdproc.combine(method='average',
// sigma_clip=True, sigma_clip_low_thresh=5, sigma_clip_high_thresh=5, sigma_clip_func=np.ma.median,
// sigma_clip_dev_func=mad_std) on the raw frames: ONE pass - base = median over N, dev = mad_std = 1.482602218505602 *
// median(|x - base|) (astropy/stats/funcs.py:844-850), mask where x - base < -low dev or > high dev (strict), float64 mean /
// std / count of the rest (restated in oracle/apref.c:apref_combine_ccdproc; parity unpinned).  The complete
// ("rich") kernel does this with the sorted column parked in LDS - 64 KB per 256 pixels, two wavefronts per SIMD - and a
// binary search for the MAD: 1.16 ms for the mean alone, 1.65 ms with the float64 planes, 1.46 ms on uint16 frames (64 x
// 4096^2).  This kernel keeps the column in registers, like stack_fast_kernel in front of the lean kernels:
//   * load (uint16 -> float32 is exact), finite test, the complete 64-slot sorting network;
//   * e_i = (x_i - m1) + (x_i - m2) = 2 (x_i - base) with m1 <= m2 the two middle values: float32, relative error <= 2u
//     (a value outside [m1, m2] has both differences of one sign; the middle values themselves give 0 + one rounded difference);
//   * the MAD without a second sort: |e_i| over the sorted column is V-shaped - a bitonic sequence - so ONE half-cleaner
//     layer (|e_i| against |e_(i + N/2)|) leaves the N / 2 smallest in the lower outputs: the two middle order statistics are
//     the maximum of the lower and the minimum of the upper outputs, E = their sum = 4 MAD (relative error <= 3u);
//   * the bounds |e_i| > c E, c = thresh x 1.4826 / 2, tested on the 8 lowest / 8 highest values with a margin rho = 2^-20
//     (7u of accumulated rounding + the constant's); a value inside the margin, a tail used up, a non-finite value, a spread
//     outside 2^-40 .. 2^40 make the pixel UNSURE;
//   * float64 sums of the survivors about the pivot m1 (middle values unconditionally, the tails by select - a rejected
//     outlier is never added, so nothing is subtracted from a total it dominates), mean = m1 + S / n, std = sqrt((Q - S^2 / n) / n).
// A wavefront with an unsure lane stores nothing and sets its 64-pixel block's flag in the caller's workspace (the block flags
// of the two-kernel scheme, stack_kernels.h); the rich kernel follows in FLAG MODE - a wavefront whose flag is clear leaves at
// once, the others reduce their block exactly and clear the flag.  Results are the rich kernel's wherever the two could
// differ; elsewhere the same survivors by construction and float64 sums of the same values.
// The guard: a stack whose every block holds an unsure pixel would pay both kernels (for blocks with a non-finite value this one
// gives up right after its loads: 1.3 x the rich kernel for float32 frames full of NaN; for blocks that use a tail up only at its
// end: 1.6 x).  So the rich kernel's first workgroup leaves a mode word behind - the blocks given up among every 16th tile against an
// eighth of them - and in mode 1 the next call's fast kernel tries only those sampled tiles and hands the others over at once:
// the steady state on bad data is the rich kernel + a sixteenth of this one, the first call after the data turned bad pays in full.
// APGPU_STACK_SINGLE_KERNEL (or no workspace) keeps the call on the rich kernel alone.
#include "stack_mad.h"

#include <hip/hip_runtime.h>
#include <utility>

namespace apgpu_stack {

using namespace apgpu;

namespace {

constexpr int kMadTail = 8;

// NP = the number of frames itself (one instantiation per count, 3 .. 128: no padding slots, every index a compile-time fact).
// Odd counts: the median is the middle element (m1 = m2), its own deviation is the smallest, and the MAD is the lower middle of
// the other NP - 1 - the same half-cleaner over the column without its vertex, E = twice the maximum of the lower outputs.
template <int NP, typename RawT>
// (uint16 frames: the 64-frame kernel fits the four-wavefront budget without spills; float32 frames need 146 registers.  65 .. 128
// frames: two wavefronts per SIMD - 216 registers at 96 frames; at 128 the float32 kernel spills 28, the uint16 one 2 - where the
// rich kernel, its column in LDS, runs ONE)
__global__ __launch_bounds__(256, NP > 64 ? 2 : (sizeof(RawT) == 2 ? 4 : 3)) void stack_mad_fast_kernel(const MadParams q)
{
    constexpr int H = NP / 2, UP = (NP + 1) / 2, T = kMadTail < H ? kMadTail : H;
    const int lane = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 256 + lane;
    const bool inside = p < q.P;
    const int64_t pc = inside ? p : q.P - 1;
    // the guard: in mode 1 (the previous call on this workspace gave up more than an eighth of its sampled blocks) only every
    // 16th tile is tried - enough to see the data turn good again - and the others go to the rich kernel at once
    typedef const int __attribute__((address_space(4))) cint;
    const bool sampled = (blockIdx.x % kMadSample) == 0;
    if (!sampled && ((cint *)(uintptr_t)q.ws)[kWsCall + kWsMadMode] != 0) {
        if ((lane & 63) == 0 && inside) q.ws[kWsFlags + 4 * (int64_t)blockIdx.x + (lane >> 6)] = 1;
        return;
    }
    float v[NP];
    const RawT *src = static_cast<const RawT *>(q.frames) + pc;
#pragma unroll
    for (int f = 0; f < NP; f++) v[f] = to_f32(src[(int64_t)f * q.stride]);
    bool unsure = !inside;                                    // (a partial last block goes to the rich kernel whole)
    if constexpr (sizeof(RawT) == 4) {
        float acc = 0.f;                                      // NaN iff some value is not finite (x * 0 is NaN for NaN and inf)
#pragma unroll
        for (int f = 0; f < NP; f++) acc = __builtin_fmaf(v[f], 0.f, acc);
        unsure = unsure || !(acc == 0.f);
#pragma unroll
        for (int f = 0; f < NP; f++) v[f] = (v[f] == v[f]) ? v[f] : __builtin_inff();      // the network is for NaN-free columns
    }
    const int wv = lane >> 6;
    auto give_up = [&]() {                                    // the block goes to the rich kernel whole
        if ((lane & 63) == 0 && inside) {                     // (a wavefront wholly behind the image has no block: no flag - an early
                                                              // return for it at the top cost the float32 kernel 5-10 %)
            q.ws[kWsFlags + 4 * (int64_t)blockIdx.x + wv] = 1;
            atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 3, 1ull);
            if (sampled) atomicAdd(q.ws + kWsCall + kWsMadCount, 1);
        }
    };
    if (blockIdx.x == 0 && lane == 0) {                       // the call's share of the workspace's cumulative statistics
        atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 0, 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 1, (unsigned long long)q.P);
        if (((cint *)(uintptr_t)q.ws)[kWsCall + kWsMadMode] != 0) {      // mode 1: the tiles handed over without a try
            const int64_t ntiles = (q.P + 255) / 256;
            atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 3,
                      (unsigned long long)(4 * (ntiles - (ntiles + kMadSample - 1) / kMadSample)));
        }
    }
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {           // a non-finite value (or the image's partial last block): before the sort,
        give_up();                                            // so that frames full of NaN cost this kernel its loads only
        return;
    }
    sort_column<NP>(v);
    const float m1 = v[(NP - 1) / 2], m2 = v[NP / 2];
    auto dev2 = [&](float x) { return (x - m1) + (x - m2); };                               // 2 (x - base)
    // the two middle order statistics of |e|: one half-cleaner layer of the bitonic (V-shaped) sequence
    float maxlo = 0.f, minhi = __builtin_inff();
#pragma unroll
    for (int i = 0; i < H; i++) {
        const float a = __builtin_fabsf(dev2(v[i])), b = __builtin_fabsf(dev2(v[i + UP]));
        maxlo = __builtin_fmaxf(maxlo, __builtin_fminf(a, b));
        minhi = __builtin_fminf(minhi, __builtin_fmaxf(a, b));
    }
    const float E = (NP & 1) ? 2.f * maxlo : maxlo + minhi;                                  // 4 MAD
    const float dmax = __builtin_fmaxf(-dev2(v[0]), dev2(v[NP - 1]));
    unsure = unsure || !(dmax == 0.f || (dmax > 0x1p-40f && dmax < 0x1p40f));
    const float rho = 0x1p-20f;
    const float tl = q.cl * E, th = q.cu * E;
    const float tl_hi = __builtin_fmaf(tl, rho, tl), tl_lo = __builtin_fmaf(tl, -rho, tl);
    const float th_hi = __builtin_fmaf(th, rho, th), th_lo = __builtin_fmaf(th, -rho, th);
    int na = 0, nb = 0;                                       // values rejected from the low / high end (prefix / suffix of the sorted column)
#pragma unroll
    for (int i = 0; i < T; i++) {
        const float a = -dev2(v[i]);
        const bool rej = a > tl_hi, keep = a <= tl_lo;
        unsure = unsure || !(rej || keep);
        na += rej ? 1 : 0;
        const float b = dev2(v[NP - 1 - i]);
        const bool rejb = b > th_hi, keepb = b <= th_lo;
        unsure = unsure || !(rejb || keepb);
        nb += rejb ? 1 : 0;
    }
    if constexpr (T < H) unsure = unsure || na == T || nb == T;   // the tail is used up: the next value is not tested here (T = H: every value is)
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
        give_up();
        return;
    }
    const int n = NP - na - nb;
    const double nn = (double)n;
    const double c = (double)m1;
    const bool want_std = q.std64 != nullptr;
    double Ss = 0.0, Qs = 0.0;                                // sums of (x - m1), (x - m1)^2 over the survivors
    bool summed = false;
    if constexpr (sizeof(RawT) == 2) {
        // uint16 frames: x - m1 is an integer below 2^16 in magnitude - exact in float32, and so are its partial sums (< 2^22);
        // the squares are summed as integers (v_mad_i32_i24), exact while every survivor lies within 4096 of m1 (64 x 2^24 < 2^31):
        // the survivors lie within half the bound of the median, so the bound says whether that holds (one wave vote)
        const float reach = 0.5f * __builtin_fmaxf(tl_hi, th_hi) + (m2 - m1);
        if (__builtin_amdgcn_ballot_w64(!(reach < 4000.f)) == 0) {
            float Sf[4] = {0.f, 0.f, 0.f, 0.f};
            int Qi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < NP; i++) {
                float d = v[i] - m1;
                if (i < T) d = (i >= na) ? d : 0.f;
                if (i >= NP - T) d = (NP - 1 - i >= nb) ? d : 0.f;
                Sf[i & 3] += d;
                if (want_std) {
                    const int di = (int)d;
                    Qi[i & 3] = __mul24(di, di) + Qi[i & 3];
                }
            }
            Ss = (double)((Sf[0] + Sf[1]) + (Sf[2] + Sf[3]));
            Qs = (double)((Qi[0] + Qi[1]) + (Qi[2] + Qi[3]));
            summed = true;
        }
    }
    if (!summed) {
        // float64 sums about m1 (middle values unconditionally, the tails by select)
        double S[4] = {0.0, 0.0, 0.0, 0.0}, Q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NP; i++) {
            double d = (double)v[i] - c;
            if (i < T) d = (i >= na) ? d : 0.0;
            if (i >= NP - T) d = (NP - 1 - i >= nb) ? d : 0.0;
            S[i & 3] += d;
            if (want_std) Q[i & 3] = fma(d, d, Q[i & 3]);
        }
        Ss = (S[0] + S[1]) + (S[2] + S[3]);
        Qs = (Q[0] + Q[1]) + (Q[2] + Q[3]);
    }
    const double mean = c + Ss / nn;
    if (q.mean) q.mean[p] = (float)mean;
    if (q.count) q.count[p] = n;
    if (q.mean64) q.mean64[p] = mean;
    if (want_std) {
        const double var = (Qs - Ss * Ss / nn) / nn;
        q.std64[p] = sqrt(var > 0.0 ? var : 0.0);
    }
}

template <int NP, typename RawT>
int launch_mad_np(const MadParams &q, hipStream_t st)
{
    const int64_t grid = (q.P + 255) / 256;
    hipLaunchKernelGGL((stack_mad_fast_kernel<NP, RawT>), dim3((unsigned)grid), dim3(256), 0, st, q);
    return check_launch("stack kernel (median / mad_std fast path)");
}

// Two translation units: this file as it stands holds the kernels of 3 .. 64 frames; stack_mad_wide.hip includes it with
// APGPU_MAD_WIDE for 65 .. 128.  In ONE code object the 126 wider kernels cost the 64-frame float32 kernel 8 % (0.95 -> 1.04 ms,
// same box, identical instructions: only its place in a 4.7 MB text section differs).
#ifdef APGPU_MAD_WIDE
constexpr int kMadMin = 65, kMadMax = 128;
#else
constexpr int kMadMin = 3, kMadMax = 64;
#endif

template <typename RawT, int... I>
int launch_mad_seq(const MadParams &q, int np, hipStream_t st, std::integer_sequence<int, I...>)
{
    int rc = kNoRedoList;
    (void)((np == kMadMin + I ? (rc = launch_mad_np<kMadMin + I, RawT>(q, st), true) : false) || ...);
    return rc;
}

template <typename RawT>
int launch_mad_t(const MadParams &q, int np, hipStream_t st)
{
    return launch_mad_seq<RawT>(q, np, st, std::make_integer_sequence<int, kMadMax - kMadMin + 1>{});
}

}  // namespace

#ifdef APGPU_MAD_WIDE
int launch_mad_wide(const MadParams &q, int np, bool u16, hipStream_t st)
{
    return u16 ? launch_mad_t<uint16_t>(q, np, st) : launch_mad_t<float>(q, np, st);
}
#else
// Whether a call is the configuration this kernel implements: unfused stack of 3 .. 128 frames, one pass of median / mad_std,
// outputs among mean / count / float64 mean / float64 std, the caller's workspace for the block flags.
bool mad_fast_eligible(const StackParams &prm, bool calib)
{
    if (calib || prm.N < 3 || prm.N > 128) return false;
    if (prm.dev != APGPU_DEV_MAD_STD || prm.center != APGPU_CENTER_MEDIAN || prm.maxiters != 1) return false;
    if (prm.pixmask || prm.pedestal || prm.median || prm.std || prm.moments) return false;
    if (!prm.redo || prm.single_kernel || prm.fast32 == 0) return false;
    if (!(prm.sl2 > 0.0 && prm.su2 > 0.0 && prm.sl2 < 1e12 && prm.su2 < 1e12)) return false;
    if ((prm.P + 255) / 256 > 0x7fffffffLL) return false;
    return true;
}

// Launches the fast kernel (kNoRedoList: not this configuration - nothing launched).  The caller follows with the rich kernel
// in flag mode.
int launch_mad_fast(const StackParams &prm, bool u16, hipStream_t st)
{
    MadParams q;
    q.frames = prm.frames;
    q.stride = prm.stride;
    q.P = prm.P;
    q.mean = prm.mean;
    q.count = prm.count;
    q.mean64 = prm.mean64;
    q.std64 = prm.std64;
    q.ws = prm.redo;
    q.cl = (float)(sqrt(prm.sl2) * 1.482602218505602 * 0.5);
    q.cu = (float)(sqrt(prm.su2) * 1.482602218505602 * 0.5);
    // uint16 frames, two pixels per lane on the packed sorting network (stack_mad_pairs.hip): pixel pairs must be whole words
    if (u16 && (prm.P % 2) == 0 && (prm.stride % 2) == 0 && (reinterpret_cast<uintptr_t>(prm.frames) & 3) == 0 &&
        (!prm.mean64 || (reinterpret_cast<uintptr_t>(prm.mean64) & 15) == 0) && (!prm.std64 || (reinterpret_cast<uintptr_t>(prm.std64) & 15) == 0) &&
        (!prm.mean || (reinterpret_cast<uintptr_t>(prm.mean) & 7) == 0) && (!prm.count || (reinterpret_cast<uintptr_t>(prm.count) & 7) == 0))
        return launch_mad_pairs(q, prm.N, st);
    if (prm.N > 64) return launch_mad_wide(q, prm.N, u16, st);
    return u16 ? launch_mad_t<uint16_t>(q, prm.N, st) : launch_mad_t<float>(q, prm.N, st);
}
#endif

}  // namespace apgpu_stack
