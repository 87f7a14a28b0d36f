// stack_mad_pairs.hip - the ccdproc.combine configuration (A6) on raw uint16 frames, TWO pixels per lane (round 5).
//
// The one-pixel-per-lane kernel of stack_mad.hip spends two thirds of its instructions sorting the column.  Raw frames are uint16
// (scripts/ap_combine_darks.py:411 reads the FITS files as they are), and a pair of uint16 values sorts as one register:
// v_pk_min_u16 / v_pk_max_u16 run the compare-exchange network on both pixels of a 4-byte load at once (the scheme of
// stack_median_u16_kernel) - 543 packed compare-exchanges for 64 frames where the float32 network needs ~930
// instructions per pixel.  The MAD's half-cleaner layer runs on packed uint16 keys as well (pair_bounds), so that of each pixel's
// column only the 16 tail values are unpacked for the bound tests (floats of exact integers, the 2^-20 margin as in stack_mad.hip)
// and the sums of the survivors are integer sums.  A lane's two pixels are neighbours: lanes
// 0 .. 31 of a wavefront hold one 64-pixel block, lanes 32 .. 63 the next; an unsure pixel flags ITS block (the rich kernel redoes it
// in flag mode), the other half of the wavefront stores its results.  Outputs are written as pairs (16-byte stores of the float64
// planes).  Same decisions and the same sums as the one-pixel kernel, which serves what this one cannot take: odd pixel counts or
// strides, unaligned planes, float32 frames.
#include "stack_mad.h"

#include <hip/hip_runtime.h>
#include <utility>

namespace apgpu_stack {

using namespace apgpu;

namespace {

constexpr int kPairTail = 8;

__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

struct PairBounds {
    int m1, reach;              // the lower middle value; how far from it a survivor can lie (for the exact integer sums)
    int na, nb;                 // values rejected from the low / high end
    bool unsure;
};

// The clip of one pixel of the pair (half = 0: low halves of the words, 1: high halves).  Integers throughout: with m1 <= m2
// the middle values and g = m2 - m1, |e_i| = |2 x_i - m1 - m2| is 2 key_i - g for key_i = m2 - x_i below the middle and
// key_i = x_i - m1 above it - keys are uint16 numbers, so the half-cleaner layer of the MAD (stack_mad.hip) ran on the PACKED words
// for both pixels at once (maxlo, minhi) and only its two results are unpacked here; E = 4 MAD is an exact integer.
template <int NP>
__device__ __forceinline__ PairBounds pair_bounds(const uint32_t (&w)[NP], int half, uint32_t maxlo, uint32_t minhi, float cl, float cu)
{
    constexpr int H = NP / 2, T = kPairTail < H ? kPairTail : H;
    const int sh = 16 * half;
    auto val = [&](uint32_t x) { return (int)((x >> sh) & 0xffffu); };
    const int m1 = val(w[(NP - 1) / 2]), m2 = val(w[NP / 2]), g = m2 - m1;
    const int Ei = (NP & 1) ? 2 * (2 * val(maxlo) - g) : 2 * (val(maxlo) + val(minhi)) - 2 * g;
    const float E = (float)Ei;                                                               // 4 MAD, exact
    const float rho = 0x1p-20f;
    const float tl = cl * E, th = cu * E;
    const float tl_hi = __builtin_fmaf(tl, rho, tl), tl_lo = __builtin_fmaf(tl, -rho, tl);
    const float th_hi = __builtin_fmaf(th, rho, th), th_lo = __builtin_fmaf(th, -rho, th);
    int na = 0, nb = 0;
    bool unsure = false;
#pragma unroll
    for (int i = 0; i < T; i++) {
        const int xa = val(w[i]), xb = val(w[NP - 1 - i]);
        const float a = (float)((m1 - xa) + (m2 - xa));                                      // -e_i >= 0, exact
        const bool rej = a > tl_hi, keep = a <= tl_lo;
        unsure = unsure || !(rej || keep);
        na += rej ? 1 : 0;
        const float b = (float)((xb - m1) + (xb - m2));
        const bool rejb = b > th_hi, keepb = b <= th_lo;
        unsure = unsure || !(rejb || keepb);
        nb += rejb ? 1 : 0;
    }
    if constexpr (T < H) unsure = unsure || na == T || nb == T;
    PairBounds r;
    r.m1 = m1;
    r.reach = (int)(0.5f * __builtin_fmaxf(tl_hi, th_hi)) + g + 1;
    r.na = na;
    r.nb = nb;
    r.unsure = unsure;
    return r;
}

// mean and standard deviation of the survivors of one pixel of the pair; exact_ints: every survivor within 4096 of m1 (wave-uniform:
// then the squares' sum stays below 2^31), else float64 sums
template <int NP>
__device__ __forceinline__ void pair_sums(const uint32_t (&w)[NP], int half, const PairBounds &b, bool want_std, bool exact_ints,
                                          double &mean, double &sd, int &n)
{
    constexpr int H = NP / 2, T = kPairTail < H ? kPairTail : H;
    const int sh = 16 * half;
    const int na = b.na, nb = b.nb, m1 = b.m1;
    n = NP - na - nb;
    const double nn = (double)n;
    double Ss, Qs;
    int Si[4] = {0, 0, 0, 0};                                // sums of x - m1: |x - m1| < 2^16, 64 of them - exact
    if (exact_ints) {
        int Qi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NP; i++) {
            int d = (int)((w[i] >> sh) & 0xffffu) - m1;
            if (i < T) d = (i >= na) ? d : 0;
            if (i >= NP - T) d = (NP - 1 - i >= nb) ? d : 0;
            Si[i & 3] += d;
            if (want_std) Qi[i & 3] = __mul24(d, d) + Qi[i & 3];
        }
        Qs = (double)((Qi[0] + Qi[1]) + (Qi[2] + Qi[3]));
    } else {
        double Q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NP; i++) {
            int d = (int)((w[i] >> sh) & 0xffffu) - m1;
            if (i < T) d = (i >= na) ? d : 0;
            if (i >= NP - T) d = (NP - 1 - i >= nb) ? d : 0;
            Si[i & 3] += d;
            if (want_std) {
                const double dd = (double)d;
                Q[i & 3] = fma(dd, dd, Q[i & 3]);
            }
        }
        Qs = (Q[0] + Q[1]) + (Q[2] + Q[3]);
    }
    Ss = (double)((Si[0] + Si[1]) + (Si[2] + Si[3]));
    mean = (double)m1 + Ss / nn;
    const double var = (Qs - Ss * Ss / nn) / nn;
    sd = sqrt(var > 0.0 ? var : 0.0);
}

template <int NP>
__global__ __launch_bounds__(256, NP > 64 ? 2 : 3) void stack_mad_pairs_kernel(const MadParams q)
{
    constexpr int H = NP / 2, UP = (NP + 1) / 2;
    const int lane = threadIdx.x;
    const int64_t p2 = ((int64_t)blockIdx.x * 256 + lane) * 2;      // this lane's pixel pair (P is even)
    const bool inside = p2 < q.P;
    // the 64-pixel blocks of the workgroup's 512 pixels: lanes 32 h .. 32 h + 31 hold block 8 blockIdx.x + h
    const int64_t block = 8 * (int64_t)blockIdx.x + (lane >> 5);
    typedef const int __attribute__((address_space(4))) cint;
    const bool sampled = (blockIdx.x % kMadSample) == 0;                                      // (512-pixel tiles here: the same share of the image)
    if (!sampled && ((cint *)(uintptr_t)q.ws)[kWsCall + kWsMadMode] != 0) {
        if ((lane & 31) == 0 && inside) q.ws[kWsFlags + block] = 1;
        return;
    }
    if (blockIdx.x == 0 && lane == 0) {                       // the call's share of the workspace's cumulative statistics
        atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 0, 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 1, (unsigned long long)q.P);
        if (((cint *)(uintptr_t)q.ws)[kWsCall + kWsMadMode] != 0) {
            const int64_t ntiles = (q.P + 511) / 512;
            atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 3,
                      (unsigned long long)(8 * (ntiles - (ntiles + kMadSample - 1) / kMadSample)));
        }
    }
    uint32_t w[NP];
    {
        const int64_t pc = inside ? p2 : q.P - 2;
        const uint32_t *fp = reinterpret_cast<const uint32_t *>(static_cast<const uint16_t *>(q.frames) + pc);
        const int64_t step = q.stride / 2;
#pragma unroll
        for (int f = 0; f < NP; f++) w[f] = fp[(int64_t)f * step];
    }
    net_from_pk16<NP, 0>(w);
    // the half-cleaner layer of the MAD on the packed keys, both pixels at once (see pair_bounds)
    const uint32_t M1 = w[(NP - 1) / 2], M2 = w[NP / 2];
    uint32_t maxlo = 0u, minhi = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < H; i++) {
        const uint32_t klo = pk_sub_u16(M2, w[i]), khi = pk_sub_u16(w[i + UP], M1);
        maxlo = pk_max_u16(maxlo, pk_min_u16(klo, khi));
        minhi = pk_min_u16(minhi, pk_max_u16(klo, khi));
    }
    const PairBounds ba = pair_bounds<NP>(w, 0, maxlo, minhi, q.cl, q.cu);
    const PairBounds bb = pair_bounds<NP>(w, 1, maxlo, minhi, q.cl, q.cu);
    const bool unsure = !inside || ba.unsure || bb.unsure;
    const uint64_t um = __builtin_amdgcn_ballot_w64(unsure);
    const bool mine = ((lane & 32) ? (um >> 32) : (um & 0xffffffffull)) != 0;                 // my half of the wavefront = my block
    if (mine) {
        if ((lane & 31) == 0 && inside) {                     // (a half-wavefront wholly behind the image has no block)
            q.ws[kWsFlags + block] = 1;
            atomicAdd(reinterpret_cast<unsigned long long *>(q.ws + kWsStats) + 3, 1ull);
            if (sampled) atomicAdd(q.ws + kWsCall + kWsMadCount, 1);
        }
        return;
    }
    const bool want_std = q.std64 != nullptr;
    const bool ints = __builtin_amdgcn_ballot_w64(!(ba.reach < 4000 && bb.reach < 4000)) == 0;
    double mean_b, sd_b, mean_a, sd_a;
    int n_b, n_a;
    pair_sums<NP>(w, 0, ba, want_std, ints, mean_a, sd_a, n_a);
    pair_sums<NP>(w, 1, bb, want_std, ints, mean_b, sd_b, n_b);
    if (q.mean) *reinterpret_cast<float2 *>(q.mean + p2) = make_float2((float)mean_a, (float)mean_b);
    if (q.count) *reinterpret_cast<int2 *>(q.count + p2) = make_int2(n_a, n_b);
    if (q.mean64) *reinterpret_cast<double2 *>(q.mean64 + p2) = make_double2(mean_a, mean_b);
    if (want_std) *reinterpret_cast<double2 *>(q.std64 + p2) = make_double2(sd_a, sd_b);
}

template <int NP>
int launch_pairs_np(const MadParams &q, hipStream_t st)
{
    const int64_t grid = (q.P / 2 + 255) / 256;
    hipLaunchKernelGGL((stack_mad_pairs_kernel<NP>), dim3((unsigned)grid), dim3(256), 0, st, q);
    return check_launch("stack kernel (median / mad_std fast path, uint16 pairs)");
}

// (two translation units, like stack_mad.hip / stack_mad_wide.hip: stack_mad_pairs_wide.hip includes this file with APGPU_MAD_WIDE
// for 65 .. 128 frames - w[128] and little else: two wavefronts per SIMD)
#ifdef APGPU_MAD_WIDE
constexpr int kPairMin = 65, kPairMax = 128;
#else
constexpr int kPairMin = 3, kPairMax = 64;
#endif

template <int... I>
int launch_pairs_seq(const MadParams &q, int np, hipStream_t st, std::integer_sequence<int, I...>)
{
    int rc = kNoRedoList;
    (void)((np == kPairMin + I ? (rc = launch_pairs_np<kPairMin + I>(q, st), true) : false) || ...);
    return rc;
}

}  // namespace

#ifdef APGPU_MAD_WIDE
int launch_mad_pairs_wide(const MadParams &q, int np, hipStream_t st)
{
    return launch_pairs_seq(q, np, st, std::make_integer_sequence<int, kPairMax - kPairMin + 1>{});
}
#else
int launch_mad_pairs(const MadParams &q, int np, hipStream_t st)
{
    if (np > kPairMax) return launch_mad_pairs_wide(q, np, st);
    return launch_pairs_seq(q, np, st, std::make_integer_sequence<int, kPairMax - kPairMin + 1>{});
}
#endif

}  // namespace apgpu_stack
