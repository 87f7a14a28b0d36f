// stack_chunks.hip - 129 .. 256 frames (round 3) and 257 .. 512 frames (round 5: window_step, pairs of chunks) on the float32
// fast path: the column never exists as a whole.  Round 6: the median / std planes of the clipped mean (epilogue +
// stack_std_pass_kernel), and - second half of the file - the plain median and the median / mad_std configuration on order
// statistics alone (stack_rank_chunks_kernel, stack_mad_sums_kernel).
//
// Same semantics and outputs as the lean register kernels (stack_kernels.h / stack_reduce.h): the sigma-clipped mean /
// count / partial moments of astropy.stats.sigma_clipped_stats(cube, axis=0) (sigma_clipping.py:298-383, 924-937) with the
// median as centre and the std as deviation, optionally fused with ApCalibrate's arithmetic (core/ApCalibrate.py:439-464).
// scripts/ap_combine_darks.py:411-420 (ccdproc.combine) takes any N; BASELINE's C3 needs 256 frames on one GPU.
//
// What the clip of a sorted column needs (clip_fast32): the few lowest and highest values in order, the few order
// statistics around the middle, and SUMS over everything in between.  None of that needs the whole column sorted, or
// resident.  One lane owns one pixel and walks K = ceil(N / 64) chunks of c_k <= 64 frames; every chunk is loaded,
// calibrated and sorted in registers by the ordinary 64-slot path (load_sorted_column: same fused calibration, guards and
// network), and only three things survive it:
//   * its 8 lowest / 8 highest values, merged into the running global tails GL / GH (bitonic 8 + 8 -> 8; the values
//     pushed out of a tail are ADDED to the running sums - sums only ever grow by what belongs to them, clip_fast32's
//     rule against subtracting an outlier from a total it dominates);
//   * a window of 32 values around ITS middle.  A chunk is a stride-K sample of the column (interleaved chunks, round 4: robust to drift in acquisition order), so the column's middle order
//     statistics have local rank c_k / 2 +- 4 (one sigma) in it: the 4-sigma window holds them.  After the last chunk the
//     K windows are merged (two levels of Batcher's odd-even merge, pruned to the 16 middle outputs) and the element of
//     global rank r is merged element r - (number of values below the windows) - PROVIDED it lies in the zone
//     [max_k W_k[0], min_k W_k[31]] in which every value below a window is known to be smaller and every value above
//     one larger.  A median outside that zone (probability ~1e-4 per wave) sends the wave to the exact kernel;
//   * S = sum(x - c0), Q = sum((x - c0)^2) of its other values, c0 = the first chunk's median (float32, per-chunk
//     partial sums added to running totals: the error budget of clip_fast32 holds with the same rho, see below).
// The clip then runs exactly as clip_fast32 does, with tails of 8 (7 usable: the value after the tail is not known) and
// the median picked from the merged window.  Lanes that are not sure - a comparison inside the error margin, a tail used
// up, non-finite values, a masked pixel, a median outside the zone - put their pixel on a redo list, and
// stack_big_kernel (the exact LDS-resident kernel, stack_big.hip) recomputes exactly those pixels afterwards.
// Survivor sets are therefore those of the exact path; the mean carries the float32 rounding of its sum like the
// 64-frame kernel's.
//
// Error budget: per chunk 4 chains of <= 12 terms + 2 (+ <= 16 pushed-out values in 4 chains + 2), K <= 4 chunk totals
// + 3, tails 8 + 2: every partial sum at most ~24 roundings deep -> |dQ| <= 26u Q, |dS| <= 25u sqrt(n Q), |dV| <=
// (27 + 50 + 1)u nQ <= 312u V under the guard V >= nQ / 4, + 10u for the test itself: rho = 2^-15 = 512u covers it.
// 257 .. 512 frames (8 chunks, tails of 16): per chunk 4 chains of <= 8 terms + 2 and <= 32 pushed-out values in 4 chains
// + 2, 8 chunk totals + 7, tails 16 + 2, the final three-term sum + 2: at most ~30 roundings -> |dQ| <= 32u Q, |dS| <=
// 31u sqrt(n Q), |dV| <= (33 + 62 + 1)u nQ <= 384u V, + 10u for the test: still inside rho = 512u.
//
// Registers / LDS: 64 (chunk) + 32 (half of its raw values) + the last two windows + 16 (tails) + sums in registers; the
// windows of the first K - 2 chunks are parked in LDS (64 KB per 256-pixel workgroup for K = 4): two workgroups = eight
// wavefronts per CU.
#include "stack_kernels.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

namespace apgpu_stack {

using namespace apgpu;

constexpr int kChunkSlots = 64;
constexpr int kChunkTail = 8;
#ifndef APGPU_CHUNKS_PAIR_TAIL
#define APGPU_CHUNKS_PAIR_TAIL 16
#endif
constexpr int kChunkTailPairs = APGPU_CHUNKS_PAIR_TAIL;     // 257 .. 512 frames
constexpr int kChunkWin = 32;
constexpr int kChunkMinFrames = 33;                          // a chunk holds its whole window: c_k >= 33
#ifndef APGPU_CHUNKS_HALVES
#define APGPU_CHUNKS_HALVES 1
#endif
#ifndef APGPU_CHUNKS_CLAMPED_LOADS
#define APGPU_CHUNKS_CLAMPED_LOADS 1
#endif
#ifndef APGPU_CHUNKS_MINBLOCKS
#define APGPU_CHUNKS_MINBLOCKS 2
#endif

// out[j] = v[BASE + s + j], j < LEN, for a WAVE-UNIFORM s in [0, MAXS]: logarithmic shifter over a working copy, every
// register index static, one scalar branch per bit of s.
template <int LEN, int MAXS, int BASE, int NV>
__device__ __forceinline__ void uniform_slice(const float (&v)[NV], int s, float (&out)[LEN])
{
    constexpr int WN = LEN + MAXS;
    static_assert(BASE + WN <= NV, "slice leaves the array");
    float w[WN];
#pragma unroll
    for (int i = 0; i < WN; i++) w[i] = v[BASE + i];
#pragma unroll
    for (int bit = 16; bit >= 1; bit >>= 1) {
        if (bit <= MAXS && (s & bit)) {
#pragma unroll
            for (int i = 0; i + bit < WN; i++) w[i] = w[i + bit];
        }
    }
#pragma unroll
    for (int j = 0; j < LEN; j++) out[j] = w[j];
}

// Sorts a bitonic sequence of T = 8 or 16 values ascending (log2 T layers of T / 2 compare-exchanges).
template <int T>
__device__ __forceinline__ void bitonic_sort(float (&t)[T])
{
#pragma unroll
    for (int d = T / 2; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < T; i++)
            if ((i & d) == 0) cmpx(t[i], t[i + d]);
    }
}

// The frames of a chunk, F0 .. F0 + CNT - 1 of its slots.  A ragged chunk (FULLCH = false: c < 64 frames) loads EVERY slot, the slots
// beyond the chunk re-reading its last frame (a cache hit; the callers overwrite those slots with sentinels): straight-line code.
// load_raw (stack_calibrate.h) skips the padding slots of a ragged stack with one wave-uniform branch per slot, and the compiler
// then waits for every load inside its branch - s_waitcnt vmcnt(0) 31 times per chunk in the uint16 kernels, one memory latency
// each (round 6, read off the ISA: 300 uint16 frames took 6.8 ms for the median where 384 took 4.1).
template <int NP, typename RawT, bool FULLCH, int F0, int CNT, int MINN>
__device__ __forceinline__ void load_chunk_raw(const StackParams &prm, int64_t base, int lane, RawT (&raw)[CNT])
{
#if APGPU_CHUNKS_CLAMPED_LOADS
    if constexpr (FULLCH || sizeof(RawT) == 4) {            // (float32 chunks: load_raw's branches carry no waits; measured equal or 2-4 % better)
        load_raw<NP, RawT, FULLCH, F0, CNT, MINN>(prm, base, lane, raw);
    } else {
        const RawT *fb = static_cast<const RawT *>(prm.frames) + base + (int64_t)F0 * prm.stride;
        int nframes = prm.N;
        asm volatile("" : "+s"(nframes));
#pragma unroll
        for (int f = 0; f < CNT; f++) {
            raw[f] = fb[lane];
            if (F0 + f + 1 < MINN || F0 + f + 1 < nframes) fb += prm.stride;      // (a scalar select, no branch)
            if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
    load_raw<NP, RawT, FULLCH, F0, CNT, MINN>(prm, base, lane, raw);
#endif
}

struct ChunkSums {
    float S[4], Q[4];
};

__device__ __forceinline__ void add_value(ChunkSums &cs, int k, float x, float c0)
{
    const float d = x - c0;
    cs.S[k & 3] += d;
    cs.Q[k & 3] = __builtin_fmaf(d, d, cs.Q[k & 3]);
}

// The chunk's column, calibrated on the fast path and sorted (sentinels of a ragged chunk last).  No exact fallback in this
// kernel: a lane whose fast calibration does not vouch for its values (non-finite value, operand or quotient outside the
// guarded ranges - calibrate_fast, range_ok_sorted) returns false and its wave is redone by the exact kernel.  (With the
// exact reload path and its second sort inlined four times the kernel needed 256 VGPRs + 134 spilled ones, and the spill
// traffic - as many memory operations as the frame loads - halved its throughput.)
template <typename RawT, bool CALIB, bool FULLCH>
__device__ __forceinline__ bool load_chunk(const StackParams &q, const FrameScalars<kChunkSlots> &fs, int64_t base, int lane,
                                           float (&v)[kChunkSlots])
{
    constexpr int NP = kChunkSlots, MINN = FULLCH ? NP : kChunkMinFrames - 1;
    const int N = q.N;
    const int64_t p = base + lane;
    bool good = true;
    if constexpr (CALIB) {
        const float b = q.bias[p];
        const float d = q.dark[p];
        const float D = q.still_biased ? d - b : d;          // ApCalibrate.py:440-445
        float nf = 1.f;
        bool dodiv = false;
        if (q.nflat) {
            nf = q.nflat[p];
            dodiv = (nf != 0.f);                             // ApCalibrate.py:462 (NaN != 0 is True)
        }
#if APGPU_CHUNKS_HALVES
        constexpr int HN = NP / 2;
        RawT half[HN];
        load_chunk_raw<NP, RawT, FULLCH, 0, HN, MINN>(q, base, lane, half);
        good = q.pedestal ? calibrate_fast<NP, RawT, true, 0, HN, false, MINN>(fs, half, b, D, nf, dodiv, v, N)
                          : calibrate_fast<NP, RawT, false, 0, HN, false, MINN>(fs, half, b, D, nf, dodiv, v, N);
        load_chunk_raw<NP, RawT, FULLCH, HN, HN, MINN>(q, base, lane, half);
        const bool good2 = q.pedestal ? calibrate_fast<NP, RawT, true, HN, HN, false, MINN>(fs, half, b, D, nf, dodiv, v, N)
                                      : calibrate_fast<NP, RawT, false, HN, HN, false, MINN>(fs, half, b, D, nf, dodiv, v, N);
        good = good && good2;
#else
        RawT raw[NP];
        load_chunk_raw<NP, RawT, FULLCH, 0, NP, MINN>(q, base, lane, raw);
        good = q.pedestal ? calibrate_fast<NP, RawT, true, 0, NP, false, MINN>(fs, raw, b, D, nf, dodiv, v, N)
                          : calibrate_fast<NP, RawT, false, 0, NP, false, MINN>(fs, raw, b, D, nf, dodiv, v, N);
#endif
        sort_column<NP>(v);
        good = good && range_ok_sorted<NP, MINN>(v, dodiv, N);
    } else {
        RawT raw[NP];
        load_chunk_raw<NP, RawT, FULLCH, 0, NP, MINN>(q, base, lane, raw);
        int nframes = N;
        asm volatile("" : "+s"(nframes));
        float acc = 0.f;                                    // NaN iff some value is not finite (x * 0 is NaN for NaN and inf)
#pragma unroll
        for (int f = 0; f < NP; f++) {
            if (f >= MINN && f >= nframes) {
                v[f] = __builtin_inff();                    // padding slot of a ragged chunk (wave-uniform test)
            } else {
                v[f] = to_f32(raw[f]);
                acc = __builtin_fmaf(v[f], 0.f, acc);
            }
        }
        good = acc == 0.f;
        asm volatile("" : "+v"(acc));                       // (evaluated here, not sunk to where the flag is read)
        sort_column<NP>(v);
    }
    return good;
}

// One chunk: frames [f0, f0 + c) of the stack -> tails merged, window kept, sums accumulated.  Returns false for a lane
// whose chunk holds a non-finite value (or whose pixel is masked): its wave goes to the exact kernel.
// Round 4 (advisor): the chunks are INTERLEAVED - chunk k holds frames k, k + NCH, k + 2 NCH, .. - so that every chunk is a
// stride-NCH sample of the whole sequence: a stack that drifts in acquisition order (sky background over a night, dark
// current with temperature) no longer separates the chunk medians, which would empty the zone the windows vouch for and
// send every wavefront to the redo list (correct, but the chunk pass AND the exact pass were paid).
template <typename RawT, bool CALIB, bool FULLCH, bool FIRST, int NCH, int T>
__device__ __forceinline__ bool chunk_step(const StackParams &prm, const FrameScalars<kChunkSlots> &fs, int kchunk, int c, int64_t base,
                                           int lane, float (&GL)[T], float (&GH)[T], float (&win)[kChunkWin],
                                           float &c0, float &Stot, float &Qtot, float &Lmax, float &Umin)
{
    constexpr int NP = kChunkSlots;
    __builtin_amdgcn_sched_barrier(0);                      // chunks do not overlap: the scheduler otherwise stretches live ranges across them
    StackParams q = prm;                                    // the chunk as a stack of its own
    q.frames = static_cast<const RawT *>(prm.frames) + (int64_t)kchunk * prm.stride;
    q.stride = prm.stride * NCH;
    q.N = c;                                                // (the per-frame scalars come from the staged copy fs, gathered with the same stride)
    float v[NP];
    const bool ok = load_chunk<RawT, CALIB, FULLCH>(q, fs, base, lane, v);     // v[0 .. c) are the chunk, sorted; false: redo
    float lo[T], hi[T];
#pragma unroll
    for (int j = 0; j < T; j++) lo[j] = v[j];
    if constexpr (FULLCH) {
#pragma unroll
        for (int j = 0; j < T; j++) hi[j] = v[NP - T + j];
#pragma unroll
        for (int j = 0; j < kChunkWin; j++) win[j] = v[(NP - kChunkWin) / 2 + j];
    } else {
        uniform_slice<T, NP - kChunkMinFrames, kChunkMinFrames - T, NP>(v, c - kChunkMinFrames, hi);          // v[c - 8 .. c)
        uniform_slice<kChunkWin, (NP - kChunkWin) / 2, 0, NP>(v, c / 2 - kChunkWin / 2, win);                  // v[c/2 - 16 .. c/2 + 16)
    }
    if constexpr (FIRST) c0 = win[kChunkWin / 2];       // the first chunk's (upper) median: the pivot of every sum
    Lmax = FIRST ? win[0] : fmaxf(Lmax, win[0]);
    Umin = FIRST ? win[kChunkWin - 1] : fminf(Umin, win[kChunkWin - 1]);
    // the chunk's own sums: everything between its tails, in mirror pairs (clip_fast32)
    ChunkSums cs = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (FULLCH) {
#pragma unroll
        for (int i = T; i < NP / 2; i++) {
            const float d1 = v[i] - c0, d2 = v[NP - 1 - i] - c0;
            cs.S[i & 3] += d1 + d2;
            cs.Q[i & 3] = __builtin_fmaf(d1, d1, cs.Q[i & 3]);
            cs.Q[(i + 2) & 3] = __builtin_fmaf(d2, d2, cs.Q[(i + 2) & 3]);
        }
    } else {
        int cend = c - T;                                   // wave-uniform: slots [8, c - 8)
        asm volatile("" : "+s"(cend));
#pragma unroll
        for (int i = T; i < NP - T; i++) {
            if (i >= kChunkMinFrames - T && i >= cend) continue;
            add_value(cs, i, v[i], c0);
        }
    }
    // tails: T + T -> T (bitonic), the values pushed out join the sums
    if constexpr (FIRST) {
#pragma unroll
        for (int j = 0; j < T; j++) { GL[j] = lo[j]; GH[j] = hi[j]; }
    } else {
        float out[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            float a = GL[j], b = lo[T - 1 - j];
            cmpx(a, b);                                     // a = min stays in the tail, b = max leaves it
            GL[j] = a;
            out[j] = b;
        }
        bitonic_sort<T>(GL);
#pragma unroll
        for (int j = 0; j < T; j++) add_value(cs, j, out[j], c0);
#pragma unroll
        for (int j = 0; j < T; j++) {
            float a = GH[j], b = hi[T - 1 - j];
            cmpx(a, b);                                     // b = max stays
            GH[j] = b;
            out[j] = a;
        }
        bitonic_sort<T>(GH);
#pragma unroll
        for (int j = 0; j < T; j++) add_value(cs, j + 2, out[j], c0);
    }
    const float Sk = (cs.S[0] + cs.S[1]) + (cs.S[2] + cs.S[3]);
    const float Qk = (cs.Q[0] + cs.Q[1]) + (cs.Q[2] + cs.Q[3]);
    Stot = FIRST ? Sk : Stot + Sk;
    Qtot = FIRST ? Qk : Qtot + Qk;
    // pin the sums HERE: their only use is the clip at the end, inside the region the `ok` flags guard, and LLVM sinks the
    // whole summation down there - keeping every chunk's values alive (132 spilled registers, measured)
    asm volatile("" : "+v"(Stot), "+v"(Qtot));
    __builtin_amdgcn_sched_barrier(0);
    return ok;
}

struct FastT {
    float m1, m2, nf;
    float tl_hi, tl_lo, th_hi, th_lo;
    float Slo, Qlo, Shi, Qhi;
    int ta, tb;                                             // values trimmed from the low / high end
    bool unsure;
};

__device__ __forceinline__ float fast_t(const FastT &f, float x)
{
    const float w = f.nf * ((x - f.m1) + (x - f.m2));
    return w * w;
}

template <int I, int T>
__device__ __forceinline__ void trim_low_t(const float (&GL)[T], FastT &f, const float (&SL)[T + 1], const float (&QL)[T + 1])
{
    if constexpr (I < T) {
        const float t = fast_t(f, GL[I]);
        const bool at = f.ta == I;
        const bool rej = at && (t > f.tl_hi);
        const bool maybe = at && (t > f.tl_lo);
        f.unsure = f.unsure || (maybe != rej);
        if (rej) {
            f.ta = I + 1;
            f.Slo = SL[I + 1];
            f.Qlo = QL[I + 1];
        }
        if (wave_any(f.ta > I)) trim_low_t<I + 1, T>(GL, f, SL, QL);
    } else {
        f.unsure = f.unsure || (f.ta == T);        // the tail is used up: the next value is not known here
    }
}

template <int I, int T>                                      // GH ascending: GH[T - 1] is the column's maximum; I counts from the top
__device__ __forceinline__ void trim_high_t(const float (&GH)[T], FastT &f, const float (&SH)[T + 1], const float (&QH)[T + 1])
{
    if constexpr (I < T) {
        const float t = fast_t(f, GH[T - 1 - I]);
        const bool at = f.tb == I;
        const bool rej = at && (t > f.th_hi);
        const bool maybe = at && (t > f.th_lo);
        f.unsure = f.unsure || (maybe != rej);
        if (rej) {
            f.tb = I + 1;
            f.Shi = SH[T - 1 - I];
            f.Qhi = QH[T - 1 - I];
        }
        if (wave_any(f.tb > I)) trim_high_t<I + 1, T>(GH, f, SH, QH);
    } else {
        f.unsure = f.unsure || (f.tb == T);
    }
}

// One WINDOW of the final merge.  A single chunk (129 .. 256 frames: every window), or a PAIR of chunks (257 .. 512 frames,
// round 5): chunks 2 SIDX and 2 SIDX + 1 are reduced one after the other - tails and sums exactly as before - and their two
// windows are merged (Batcher's 32 + 32 merge pruned to the 32 middle outputs) into ONE window of 32: the 16 lowest and 16
// highest of the 64 are dropped (they are already part of the sums, which cover everything between the tails).  The pair is a
// stride-KS sample of 66 .. 128 frames in which the column's middle order statistics have local rank c / 2 +- 5 (one sigma;
// hypergeometric), i.e. position 32 +- 5 in the merged list of 64: the kept 16 on either side are 3 sigma, and the pixels
// whose median falls outside (~1 % of them at 512 frames, measured) fail the zone test below and are redone exactly.  The zone
// the windows vouch for shrinks accordingly: a value x is at merged index (rank - below) iff everything that was left out is
// known to lie on its proper side of x - the chunk values below / above the chunk windows (Lmax / Umin as before) AND the
// dropped 16 + 16, which are <= P[0] / >= P[31]: Lmax = max(.., P[0]), Umin = min(.., P[31]), below += 16 per pair.
template <typename RawT, bool CALIB, bool FULLCH, int SIDX, int KS, int NPAIR, int T>
__device__ __forceinline__ bool window_step(const StackParams &prm, const FrameScalars<kChunkSlots> *fs, int cbase, int cextra, int64_t base,
                                            int lane, float (&GL)[T], float (&GH)[T], float (&win)[kChunkWin], float &c0, float &Stot,
                                            float &Qtot, float &Lmax, float &Umin, int &below)
{
    constexpr int W = kChunkWin, K = KS + NPAIR;            // the first NPAIR windows are pairs of chunks, the others single chunks
    if constexpr (SIDX >= NPAIR) {
        constexpr int kc = SIDX + NPAIR;
        const int c = cbase + (kc < cextra ? 1 : 0);
        below += c / 2 - W / 2;
        return chunk_step<RawT, CALIB, FULLCH, kc == 0, K, T>(prm, fs[kc], kc, c, base, lane, GL, GH, win, c0, Stot, Qtot, Lmax, Umin);
    } else {
        constexpr int ka = 2 * SIDX, kb = 2 * SIDX + 1;
        const int ca = cbase + (ka < cextra ? 1 : 0), cb = cbase + (kb < cextra ? 1 : 0);
        float Y[2 * W];
        bool ok;
        {
            float wa[W];
            ok = chunk_step<RawT, CALIB, FULLCH, SIDX == 0, K, T>(prm, fs[ka], ka, ca, base, lane, GL, GH, wa, c0, Stot, Qtot, Lmax, Umin);
#pragma unroll
            for (int j = 0; j < W; j++) Y[j] = wa[j];
        }
        {
            float wb[W];
            ok = chunk_step<RawT, CALIB, FULLCH, false, K, T>(prm, fs[kb], kb, cb, base, lane, GL, GH, wb, c0, Stot, Qtot, Lmax, Umin) && ok;
#pragma unroll
            for (int j = 0; j < W; j++) Y[W + j] = wb[j];
        }
        window_net_from<2 * W, W, W / 2, W / 2 + W>(Y);
#pragma unroll
        for (int j = 0; j < W; j++) win[j] = Y[W / 2 + j];
        Lmax = fmaxf(Lmax, win[0]);
        Umin = fminf(Umin, win[W - 1]);
        below += (ca / 2 - W / 2) + (cb / 2 - W / 2) + W / 2;
        __builtin_amdgcn_sched_barrier(0);
        return ok;
    }
}

// KS windows reach the final merge (3 or 4; the template also takes 2).  The first NPAIR of them are made of two chunks each, K = KS +
// NPAIR chunks in all (round 5: NPAIR = KS, 6 or 8 chunks; round 6: also 5 = one pair + three single chunks for 257 .. 320 frames and
// 7 = three pairs + one for 385 .. 448 - a chunk costs what it costs whether it holds 43 frames or 64, so fewer, fuller chunks),
// chunks in all (6: 257 .. 384 frames, 8: 385 .. 512), and the tails hold 16 values (a 512-frame column loses twice as many
// values to the same clip as a 256-frame one).
template <int KS, int NPAIR, typename RawT, bool CALIB, bool FULLCH>
__global__ __launch_bounds__(256, APGPU_CHUNKS_MINBLOCKS) void stack_chunks_kernel(const StackParams prm, int32_t *redo_count, int32_t *redo_list,
                                                                                       float *rich_tmp)
{
    constexpr int K = KS + NPAIR;
    constexpr int T = NPAIR > 0 ? kChunkTailPairs : kChunkTail, W = kChunkWin;
    __shared__ FrameScalars<kChunkSlots> fs[K];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int64_t p = base + lane;
    const int N = prm.N;
    // chunk sizes: N / K, the first N % K chunks one more (all in [33, 64])
    const int cbase = N / K, cextra = N % K;
    for (int k = 0; k < K; k++) {
        const int c = cbase + (k < cextra ? 1 : 0);
        for (int t = threadIdx.x; t < kChunkSlots; t += blockDim.x) {
            const int ff = k + (t < c ? t : c - 1) * K;     // interleaved chunks: frames k, k + K, ..
            fs[k].e[t] = prm.exp_ratio ? prm.exp_ratio[ff] : 0.f;
            fs[k].ped[t] = prm.pedestal ? prm.pedestal[ff] : 0.f;
            fs[k].pad[t] = t < c ? -__builtin_inff() : __builtin_inff();
        }
    }
    __syncthreads();
    if (p >= prm.P) return;

    // the first KS - 2 windows wait in LDS ([slot][lane]: bank = lane), the last two stay in registers: with all
    // four in registers next to a 64-slot chunk the kernel spills (256 VGPRs, 142 spilled); 64 KB of LDS per workgroup still
    // leaves two workgroups (8 wavefronts) per CU
    extern __shared__ float parked[];                       // [(KS - 2) * W][256]
    float R0[W], R1[W];
    float GL[T], GH[T];
    float c0 = 0.f, Stot = 0.f, Qtot = 0.f, Lmax = 0.f, Umin = 0.f;
    bool ok = !(prm.pixmask && prm.pixmask[p]);
    int below = 0;                                          // values below the windows (wave-uniform)
    if constexpr (KS >= 3) {
        float win[W];
        ok = window_step<RawT, CALIB, FULLCH, 0, KS, NPAIR, T>(prm, fs, cbase, cextra, base, lane, GL, GH, win, c0, Stot, Qtot, Lmax, Umin, below) && ok;
#pragma unroll
        for (int j = 0; j < W; j++) parked[j * 256 + lane] = win[j];
    }
    if constexpr (KS == 4) {
        float win[W];
        ok = window_step<RawT, CALIB, FULLCH, 1, KS, NPAIR, T>(prm, fs, cbase, cextra, base, lane, GL, GH, win, c0, Stot, Qtot, Lmax, Umin, below) && ok;
#pragma unroll
        for (int j = 0; j < W; j++) parked[(W + j) * 256 + lane] = win[j];
    }
    ok = window_step<RawT, CALIB, FULLCH, KS - 2, KS, NPAIR, T>(prm, fs, cbase, cextra, base, lane, GL, GH, R0, c0, Stot, Qtot, Lmax, Umin, below) && ok;
    ok = window_step<RawT, CALIB, FULLCH, KS - 1, KS, NPAIR, T>(prm, fs, cbase, cextra, base, lane, GL, GH, R1, c0, Stot, Qtot, Lmax, Umin, below) && ok;
    float X[4 * W];                                         // the KS windows side by side, +inf beyond them
    {
        int slot = lane;
        asm volatile("" : "+v"(slot) : : "memory");         // opaque index: no store-to-load forwarding through registers
#pragma unroll
        for (int j = 0; j < (KS - 2) * W; j++) X[j] = parked[j * 256 + slot];
#pragma unroll
        for (int j = 0; j < W; j++) {
            X[(KS - 2) * W + j] = R0[j];
            X[(KS - 1) * W + j] = R1[j];
        }
#pragma unroll
        for (int j = KS * W; j < 4 * W; j++) X[j] = __builtin_inff();
    }
    APGPU_MARK("chunks_merge");
    // merge the windows: [0,32)+[32,64) and [64,96)+[96,128), then the two halves - only the 16 middle outputs are read
    constexpr int MLO = 16 * KS - 8;
#ifndef APGPU_EXP_NOMERGE
    window_net_from<4 * W, W, MLO, MLO + 16>(X);
#endif
    float M[16];
#pragma unroll
    for (int j = 0; j < 16; j++) M[j] = X[MLO + j];

    APGPU_MARK("chunks_clip");
    const float sl2f = (float)prm.sl2, su2f = (float)prm.su2;
    const int maxiters = prm.maxiters;
    // tail tables, from the inside out (clip_fast32): SL[k] = sum of d(GL[k .. 8)), SH[k] = sum of d(GH[0 .. k))
    float SL[T + 1], QL[T + 1], SH[T + 1], QH[T + 1];
    SL[T] = 0.f; QL[T] = 0.f; SH[0] = 0.f; QH[0] = 0.f;
#pragma unroll
    for (int k = T - 1; k >= 0; k--) {
        const float d = GL[k] - c0;
        SL[k] = SL[k + 1] + d;
        QL[k] = __builtin_fmaf(d, d, QL[k + 1]);
    }
#pragma unroll
    for (int k = 1; k <= T; k++) {
        const float d = GH[k - 1] - c0;
        SH[k] = SH[k - 1] + d;
        QH[k] = __builtin_fmaf(d, d, QH[k - 1]);
    }
    FastT f;
    f.ta = 0;
    f.tb = 0;
    f.Slo = SL[0]; f.Qlo = QL[0]; f.Shi = SH[T]; f.Qhi = QH[T];
    const float dmax = fmaxf(c0 - GL[0], GH[T - 1] - c0);
    f.unsure = !ok || !(dmax == 0.f || (dmax > 0x1p-40f && dmax < 0x1p40f));
    const float rho = APGPU_FAST32_RHO;
    const float sl4 = 4.f * sl2f, su4 = 4.f * su2f;
    const int mbase = below + MLO;                          // merged-window index of global rank r: r - below; M index: r - mbase
    float S, Q;
    int it = 0;
    for (;;) {
        // the middle pair of the current range [ta, N - tb): ranks (ta + N - tb - 1) >> 1 and (ta + N - tb) >> 1
        const int i1 = ((f.ta + N - f.tb - 1) >> 1) - mbase, i2 = ((f.ta + N - f.tb) >> 1) - mbase;
        f.unsure = f.unsure || i1 < 0 || i2 > 15;
        f.m1 = pick_rel<0, 16, 16>(M, i1 & 15);
        f.m2 = pick_rel<0, 16, 16>(M, i2 & 15);
        f.unsure = f.unsure || !(f.m1 >= Lmax && f.m2 <= Umin);          // outside the zone the windows vouch for
        const int ta0 = f.ta, tb0 = f.tb;
        f.nf = (float)(N - f.ta - f.tb);
        S = (Stot + f.Slo) + f.Shi;
        Q = (Qtot + f.Qlo) + f.Qhi;
        const float nQ = f.nf * Q;
        const float V = __builtin_fmaf(-S, S, nQ);
        f.unsure = f.unsure || !(V >= 0.25f * nQ);
        const float tl = sl4 * V, th = su4 * V;
        f.tl_hi = __builtin_fmaf(tl, rho, tl);
        f.tl_lo = __builtin_fmaf(tl, -rho, tl);
        f.th_hi = __builtin_fmaf(th, rho, th);
        f.th_lo = __builtin_fmaf(th, -rho, th);
#ifndef APGPU_EXP_NOTRIM
        trim_low_t<0, T>(GL, f, SL, QL);
        trim_high_t<0, T>(GH, f, SH, QH);
#endif
        it++;
        const bool changed = (f.ta != ta0) || (f.tb != tb0);
        if (!(wave_any(changed) && (maxiters < 0 || it < maxiters))) break;
    }
    // re-admission (astropy applies the final bounds to all values): the innermost trimmed value of either side must be
    // surely outside them
    if (wave_any(f.ta > 0)) {
        const float t = fast_t(f, pick_rel<0, T, T>(GL, (f.ta - 1) & (T - 1)));
        f.unsure = f.unsure || (f.ta > 0 && !(t > f.tl_hi));
    }
    if (wave_any(f.tb > 0)) {
        const float t = fast_t(f, pick_rel<0, T, T>(GH, (T - f.tb) & (T - 1)));
        f.unsure = f.unsure || (f.tb > 0 && !(t > f.th_hi));
    }
    S = (Stot + f.Slo) + f.Shi;
    Q = (Qtot + f.Qlo) + f.Qhi;
    const int cnt = N - f.ta - f.tb;
    f.unsure = f.unsure || !(16.f * Q <= (float)cnt * (c0 * c0));          // mean-accuracy guard, see clip_fast32
    // Round 6 - the median and std planes of the FINAL survivors (np.nanmedian / np.nanstd of what the clip kept) for 129 .. 512
    // frames, which used to send the whole stack to the LDS-resident exact kernel (25 ms for 256 frames, 120 for 512):
    //   median: the middle pair of the final range [ta, N - tb) from the merged window, under the same two conditions as the
    //           clip's own medians (inside the 16 merged values, inside the zone the windows vouch for);
    //   std:    numpy's two-pass definition needs the survivors again - they are the values in [GL[ta], GH[T - 1 - tb]] (the
    //           column's ta smallest and tb largest values are gone; equal values share their fate, so the cut never runs
    //           through a tie) - so this kernel leaves that range and the mean in rich_tmp, and stack_std_pass_kernel streams
    //           the frames once more (float64 sum of (x - mean)^2 over the values inside the range).
    float med1 = 0.f, med2 = 0.f;
    if (prm.median) {
        const int j1 = ((f.ta + N - f.tb - 1) >> 1) - mbase, j2 = ((f.ta + N - f.tb) >> 1) - mbase;
        f.unsure = f.unsure || j1 < 0 || j2 > 15;
        med1 = pick_rel<0, 16, 16>(M, j1 & 15);
        med2 = pick_rel<0, 16, 16>(M, j2 & 15);
        f.unsure = f.unsure || !(med1 >= Lmax && med2 <= Umin);
    }
    {
        // the lanes that are not sure are redone by the exact kernel (stack_big_kernel over the redo list, one listed PIXEL per
        // lane - round 4; rounds 2-3 listed whole wavefronts: 2.3 % of them at 256 frames for a handful of lanes each); they
        // write nothing here.  One atomic per wavefront that has any.
        const uint64_t m = __builtin_amdgcn_ballot_w64(f.unsure);
        if (m != 0) {
            int first = 0;
            if (__builtin_amdgcn_readfirstlane(lane) == lane) first = atomicAdd(redo_count, (int)__builtin_popcountll(m));
            first = __builtin_amdgcn_readfirstlane(first);
            const int mine = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (f.unsure) {
                redo_list[first + mine] = (int32_t)p;
                if (rich_tmp) rich_tmp[p] = __builtin_nanf("");          // the exact kernel writes this pixel's std: the second pass skips it
                return;
            }
        }
    }
    const float nf32 = (float)cnt;
    const float y = __builtin_amdgcn_rcpf(nf32);
    const float q0 = S * y;
    const float ms32 = __builtin_fmaf(__builtin_fmaf(-nf32, q0, S), y, q0);
    const float meanf = c0 + ms32;
    if (prm.mean) prm.mean[p] = meanf;
    if (prm.count) prm.count[p] = cnt;
    if (prm.moments) store_moments(prm.moments, prm.moments64, prm.P, p, cnt, (double)c0, (double)S, (double)Q);
    if (prm.median) prm.median[p] = (float)(((double)med1 + (double)med2) / 2.0);
    if (rich_tmp) {
        // (the trim chains stop before a tail is used up - ta, tb <= T - 1 - so both picks are real values)
        rich_tmp[p] = pick_rel<0, T, T>(GL, f.ta & (T - 1));
        rich_tmp[prm.P + p] = pick_rel<0, T, T>(GH, (T - 1 - f.tb) & (T - 1));
        rich_tmp[2 * prm.P + p] = meanf;
    }
}

// One value of a streamed column (the second / third passes below), frame f of N.  uint16 frames: an unconditional load with the frame
// index clamped - in a conditional one the compiler sinks the conversion next to the load and waits there (load_raw's note in
// stack_calibrate.h; measured on these passes: uint16 A6 256 frames 7.5 against 9.2 ms).  float32 frames: the conditional load - the
// clamped form cost the std pass 11 % (3.9 against 3.5 ms for 256 frames).  The caller ignores the value of a frame f >= N.
template <typename RawT>
__device__ __forceinline__ float stream_value(const RawT *fp, int f, int N, int64_t stride)
{
    if constexpr (sizeof(RawT) == 2) return to_f32(fp[(int64_t)(f < N ? f : N - 1) * stride]);
    else return f < N ? to_f32(fp[(int64_t)f * stride]) : 0.f;
}

// The std plane of the survivors for 129 .. 512 frames, second pass (see the epilogue of stack_chunks_kernel): one pixel per lane,
// the frames streamed once more with the EXACT calibration (IEEE division, ApCalibrate.py:439-464 operation for operation - the chunk
// kernel's fast calibration gives these values or lists the pixel), float64 sum of (x - mean)^2 over the values inside the pixel's
// survivor range, std = sqrt(sum / n) - numpy's two-pass nanstd with the clipped mean this call returns.
template <typename RawT, bool CALIB>
__global__ __launch_bounds__(256) void stack_std_pass_kernel(const StackParams prm, const float *__restrict__ rich_tmp)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= prm.P) return;
    const float vmin = rich_tmp[p];
    if (!(vmin == vmin)) return;                             // listed: the exact kernel wrote its std
    const float vmax = rich_tmp[prm.P + p];
    const double mean = (double)rich_tmp[2 * prm.P + p];
    float b = 0.f, D = 0.f, nf = 1.f;
    bool dodiv = false;
    if constexpr (CALIB) {
        b = prm.bias[p];
        D = prm.still_biased ? prm.dark[p] - b : prm.dark[p];          // ApCalibrate.py:440-445
        if (prm.nflat) {
            nf = prm.nflat[p];
            dodiv = (nf != 0.f);                                        // :462
        }
    }
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat *eg = (const cfloat *)(uintptr_t)prm.exp_ratio, *pg = (const cfloat *)(uintptr_t)prm.pedestal;
    const RawT *fp = static_cast<const RawT *>(prm.frames) + p;
    const int N = prm.N;
    double q = 0.0, s = 0.0;
    int n = 0;
    for (int f0 = 0; f0 < N; f0 += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = stream_value<RawT>(fp, f0 + j, N, prm.stride);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float v = (f0 + j < N) ? x[j] : __builtin_nanf("");
            if constexpr (CALIB) {
                const int ff = f0 + j < N ? f0 + j : N - 1;
                const float e = eg ? eg[ff] : 0.f, ped = pg ? pg[ff] : 0.f;
                if (ped != 0.f) v = v + ped;                 // :318-326
                v = v - b;                                   // :439
                const float ds = e * D;                      // :450
                v = v - ds;                                  // :451
                if (dodiv) v = __fdiv_rn(v, nf);             // :463
            }
            const bool in = (v >= vmin) && (v <= vmax);      // false for NaN; +-inf lie outside every finite range
            const double d = in ? (double)v - mean : 0.0;
            s += d;
            q = fma(d, d, q);
            n += in ? 1 : 0;
        }
    }
    // about the pivot `mean` (the float32 mean this call returns, within an ulp of the true one): mean64 = pivot + S / n,
    // var = (Q - S^2 / n) / n - the float64 planes of ApStack / the N-shard exchange, and the std plane without the pivot's rounding
    const double nn = (double)n;
    const double var = n > 0 ? (q - s * s / nn) / nn : 0.0;
    const double sd = n > 0 ? sqrt(var > 0.0 ? var : 0.0) : (double)__builtin_nanf("");
    if (prm.std) prm.std[p] = (float)sd;
    if (prm.mean64) prm.mean64[p] = n > 0 ? mean + s / nn : (double)__builtin_nanf("");
    if (prm.std64) prm.std64[p] = sd;
}

// ---- dispatch -----------------------------------------------------------------------------------------------------
// stack_big.hip; redo_count != nullptr: only the listed pixels; ws != nullptr: the list lives in the caller's workspace - clear its counter afterwards
int launch_big_exact(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe, int32_t *redo_count = nullptr,
                     const int32_t *redo_list = nullptr, int32_t *ws = nullptr);

// Whether a stack of 129 .. 512 frames can take the chunked fast path: lean outputs, median centre / std deviation, the
// float32 path not switched off, float64-layout moments only when they are for a mean.
bool chunks_eligible(const StackParams &prm, bool median_only)
{
    // (97 .. 128 frames as two chunks were measured too: 2.8 - 3.1 ms for 128 frames against 2.7 ms for the 128-slot
    // register kernel - a chunk costs what the whole 64-frame kernel costs - so the register kernels keep that range)
    if (median_only || prm.N <= 128 || prm.N > 8 * kChunkSlots) return false;
    if (prm.P >= 0x7fffffffLL) return false;                 // the redo list holds pixel indices as int32
    // (the median / std planes and the float64 mean / std planes: round 6, stack_std_pass_kernel)
    if (prm.center != APGPU_CENTER_MEDIAN || prm.dev != APGPU_DEV_STD) return false;
    if (prm.fast32 == 0) return false;
    if (prm.moments && !(prm.moments64 == 0 || prm.moments64 == 3 || prm.moments64 == 4) ) return false;
    if (prm.moments && prm.moments64 != 0 && prm.fast32 != 2) return false;
    return true;
}

template <int KS, int NPAIR, typename RawT, bool CALIB>
static int launch_chunks_k(const StackParams &prm, bool u16, hipStream_t st, char *describe)
{
    const bool fullch = prm.N == (KS + NPAIR) * kChunkSlots;
    if (describe) {
        snprintf(describe, 256, "stack_chunks_kernel<%d, %d, %s, %s, %s>", KS, NPAIR, sizeof(RawT) == 2 ? "unsigned short" : "float",
                 CALIB ? "true" : "false", fullch ? "true" : "false");
        return APGPU_OK;
    }
    const int64_t grid = (prm.P + 255) / 256;
    if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
    // redo list: a count + up to one entry per pixel.  In the caller's workspace (apgpu_stack_args.workspace: the count is word 0
    // of its first counter line - zero between calls -, the entries its list area, which holds more than P words) or, without
    // one, a stream-ordered temporary; if that cannot be had either, the exact kernel reduces the whole stack (it needs no list).
    int32_t *ws = prm.redo, *cnt = nullptr, *list = nullptr, *own = nullptr;
    StackParams q = prm;
    q.redo = nullptr;
    if (ws) {
        cnt = ws;
        list = ws + ws_list_off(prm.P);
    } else {
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&own), (size_t)(prm.P + 1) * sizeof(int32_t), st);
        if (e == hipSuccess) {
            e = hipMemsetAsync(own, 0, sizeof(int32_t), st);
            if (e != hipSuccess) (void)hipFreeAsync(own, st);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return launch_big_exact(q, u16, CALIB, false, st, nullptr);
        }
        cnt = own;
        list = own + 1;
    }
    const size_t lds = (size_t)(KS - 2) * kChunkWin * 256 * sizeof(float);
    if (lds > 48 * 1024) {
        const void *kern = fullch ? reinterpret_cast<const void *>(stack_chunks_kernel<KS, NPAIR, RawT, CALIB, true>)
                                  : reinterpret_cast<const void *>(stack_chunks_kernel<KS, NPAIR, RawT, CALIB, false>);
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {                              // no room for the parked windows: the exact kernel does the whole stack
            (void)hipGetLastError();
            if (own) (void)hipFreeAsync(own, st);
            return launch_big_exact(q, u16, CALIB, false, st, nullptr);
        }
    }
    // the std plane needs the survivors' range and mean of every pixel between the two passes: 12 bytes per pixel, a stream-ordered
    // temporary (like the resample's tile records); if it cannot be had the exact kernel reduces the whole stack
    float *rich_tmp = nullptr;
    if (prm.std || prm.mean64 || prm.std64) {
        const hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&rich_tmp), (size_t)prm.P * 3 * sizeof(float), st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (own) (void)hipFreeAsync(own, st);
            return launch_big_exact(q, u16, CALIB, false, st, nullptr);
        }
    }
    if (fullch) hipLaunchKernelGGL((stack_chunks_kernel<KS, NPAIR, RawT, CALIB, true>), dim3((unsigned)grid), dim3(256), lds, st, q, cnt, list, rich_tmp);
    else hipLaunchKernelGGL((stack_chunks_kernel<KS, NPAIR, RawT, CALIB, false>), dim3((unsigned)grid), dim3(256), lds, st, q, cnt, list, rich_tmp);
    int rc = check_launch("stack kernel (chunked, 129..512 frames)");
    if (rc == APGPU_OK) rc = launch_big_exact(q, u16, CALIB, false, st, nullptr, cnt, list, ws);
    if (rc == APGPU_OK && rich_tmp) {
        hipLaunchKernelGGL((stack_std_pass_kernel<RawT, CALIB>), dim3((unsigned)grid), dim3(256), 0, st, q, rich_tmp);
        rc = check_launch("stack kernel (chunked: std plane, second pass)");
    }
    if (rich_tmp) {
        const hipError_t ef = hipFreeAsync(rich_tmp, st);
        if (rc == APGPU_OK && ef != hipSuccess) rc = fail(APGPU_ELAUNCH, "stack (chunks): free: %s", hipGetErrorString(ef));
    }
    if (own) {
        const hipError_t ef = hipFreeAsync(own, st);
        if (rc == APGPU_OK && ef != hipSuccess) return fail(APGPU_ELAUNCH, "stack (chunks): free: %s", hipGetErrorString(ef));
    }
    return rc;
}

template <int KS, int NPAIR>
static int launch_chunks_t(const StackParams &prm, bool u16, bool calib, hipStream_t st, char *describe)
{
    if (u16) return calib ? launch_chunks_k<KS, NPAIR, uint16_t, true>(prm, u16, st, describe) : launch_chunks_k<KS, NPAIR, uint16_t, false>(prm, u16, st, describe);
    return calib ? launch_chunks_k<KS, NPAIR, float, true>(prm, u16, st, describe) : launch_chunks_k<KS, NPAIR, float, false>(prm, u16, st, describe);
}

// How a stack of 129 .. 512 frames is cut: K = ceil(N / 64) chunks of N / K (+ 1) frames, KS = 3 or 4 windows of which the first K - KS
// are pairs.  K = 3, 4: one window per chunk; 5: (4, 1); 6: (3, 3); 7: (4, 3); 8: (4, 4).
#ifndef APGPU_CHUNKS_ODD_COUNTS
#define APGPU_CHUNKS_ODD_COUNTS 1                            // 0: round 5's cut (6 chunks up to 384 frames, 8 beyond)
#endif
static inline int chunk_count(int N)
{
    int K = (N + kChunkSlots - 1) / kChunkSlots;
#if !APGPU_CHUNKS_ODD_COUNTS
    if (K == 5 || K == 7) K++;
#endif
    return K;
}

int launch_chunks(const StackParams &prm, bool u16, bool calib, hipStream_t st, char *describe)
{
    switch (chunk_count(prm.N)) {
    case 3: return launch_chunks_t<3, 0>(prm, u16, calib, st, describe);
    case 4: return launch_chunks_t<4, 0>(prm, u16, calib, st, describe);
    case 5: return launch_chunks_t<4, 1>(prm, u16, calib, st, describe);
    case 6: return launch_chunks_t<3, 3>(prm, u16, calib, st, describe);
    case 7: return launch_chunks_t<4, 3>(prm, u16, calib, st, describe);
    default: return launch_chunks_t<4, 4>(prm, u16, calib, st, describe);
    }
}


// ---- round 6: order statistics without the clip - the plain median and the median / mad_std configuration, 129 .. 512 frames ----
//
// scripts/ap_combine_darks.py:394-420 (ccdproc.combine with np.ma.median / mad_std, ONE pass at 5 deviations; restated in
// oracle/apref.c:apref_combine_ccdproc_form) takes whatever the directory holds; beyond 128 frames every such call - and every plain
// median (np.median along N, apgpu_stack_median) - ran on the LDS-resident exact kernel, one wavefront per SIMD: 33 ms for 256 x
// 4096^2, 147 ms for 512.  What these need of a column are ORDER STATISTICS, which the windows of the chunked scheme give:
//   pass 1 (stack_rank_chunks_kernel<.., MODE 0>): the two middle values m1 <= m2 of the column - every chunk sorted in registers,
//           its 32-value middle window kept, the windows merged, the elements of global rank (N - 1) / 2 and N / 2 read off under the
//           same zone test as the clipped mean's medians.  MODE 2 writes (m1 + m2) / 2 as the median plane and stops there.
//   pass 2 (MODE 1): the same machinery on a_i = |(x_i - m1) + (x_i - m2)| = 2 |x_i - base| (float32, relative error <= 2u -
//           stack_mad.hip's e_i) gives the two middle deviations: E = a_(r1) + a_(r2) = 4 MAD (<= 3u).
//   pass 3 (stack_mad_sums_kernel): the frames streamed once more; a value is rejected when -e > cl E (low side) or e > cu E
//           (high side), c = thresh x 1.4826 / 2, decided with stack_mad.hip's margin rho = 2^-20 - a comparison inside the margin
//           makes the pixel unsure -; float64 sums of (x - m1), (x - m1)^2 over the survivors: mean = m1 + S / n, std =
//           sqrt((Q - S^2 / n) / n).
// Unsure pixels - a non-finite value (the astropy form leaves such a column unclipped: the exact kernel knows), a median outside
// the zone, a comparison inside the margin, a spread outside 2^-40 .. 2^40 - carry NaN through the per-pixel temporary, are listed
// by the last pass and reduced by stack_big_kernel, like the clipped mean's.  Three reads of the frames instead of one.
template <typename RawT, bool FULLCH, bool DEV, bool CALIB>
__device__ __forceinline__ bool rank_chunk_window(const StackParams &prm, const FrameScalars<kChunkSlots> &fs, int kchunk, int nch, int c, int64_t base,
                                                  int lane, float m1, float m2, float (&win)[kChunkWin], float &Lmax, float &Umin, bool first)
{
    constexpr int NP = kChunkSlots, MINN = FULLCH ? NP : kChunkMinFrames - 1;
    __builtin_amdgcn_sched_barrier(0);
    StackParams q = prm;
    q.frames = static_cast<const RawT *>(prm.frames) + (int64_t)kchunk * prm.stride;          // interleaved chunks, as above
    q.stride = prm.stride * nch;
    q.N = c;
    float v[NP];
    bool good;
    if constexpr (CALIB) {
        // the median of CALIBRATED frames (config 4 beyond 128 frames): the chunk kernel's loader - fast calibration, sort, range guards;
        // false: the fast path does not vouch for this lane's values, the exact kernel takes the pixel
        static_assert(!DEV, "the median / mad_std passes take raw frames");
        good = load_chunk<RawT, true, FULLCH>(q, fs, base, lane, v);
    } else {
        RawT raw[NP];
        load_chunk_raw<NP, RawT, FULLCH, 0, NP, MINN>(q, base, lane, raw);
        int nframes = c;
        asm volatile("" : "+s"(nframes));
        float acc = 0.f;                                    // NaN iff some value is not finite
#pragma unroll
        for (int f = 0; f < NP; f++) {
            if (f >= MINN && f >= nframes) {
                v[f] = __builtin_inff();                    // padding slot of a ragged chunk (wave-uniform test): sorts last
            } else {
                const float x = to_f32(raw[f]);
                acc = __builtin_fmaf(x, 0.f, acc);
                v[f] = DEV ? __builtin_fabsf((x - m1) + (x - m2)) : x;
            }
        }
        good = acc == 0.f;
        asm volatile("" : "+v"(acc));
        sort_column<NP>(v);
    }
    if constexpr (FULLCH) {
#pragma unroll
        for (int j = 0; j < kChunkWin; j++) win[j] = v[(NP - kChunkWin) / 2 + j];
    } else {
        uniform_slice<kChunkWin, (NP - kChunkWin) / 2, 0, NP>(v, c / 2 - kChunkWin / 2, win);
    }
    Lmax = first ? win[0] : fmaxf(Lmax, win[0]);
    Umin = first ? win[kChunkWin - 1] : fminf(Umin, win[kChunkWin - 1]);
    __builtin_amdgcn_sched_barrier(0);
    return good;
}

template <typename RawT, bool FULLCH, bool DEV, bool CALIB, int SIDX, int KS, int NPAIR>
__device__ __forceinline__ bool rank_window_step(const StackParams &prm, const FrameScalars<kChunkSlots> *fs, int cbase, int cextra, int64_t base, int lane,
                                                 float m1, float m2, float (&win)[kChunkWin], float &Lmax, float &Umin, int &below)
{
    constexpr int W = kChunkWin, K = KS + NPAIR;
    if constexpr (SIDX >= NPAIR) {
        constexpr int kc = SIDX + NPAIR;
        const int c = cbase + (kc < cextra ? 1 : 0);
        below += c / 2 - W / 2;
        return rank_chunk_window<RawT, FULLCH, DEV, CALIB>(prm, fs[CALIB ? kc : 0], kc, K, c, base, lane, m1, m2, win, Lmax, Umin, kc == 0);
    } else {
        constexpr int ka = 2 * SIDX, kb = 2 * SIDX + 1;
        const int ca = cbase + (ka < cextra ? 1 : 0), cb = cbase + (kb < cextra ? 1 : 0);
        float Y[2 * W];
        bool ok;
        {
            float wa[W];
            ok = rank_chunk_window<RawT, FULLCH, DEV, CALIB>(prm, fs[CALIB ? ka : 0], ka, K, ca, base, lane, m1, m2, wa, Lmax, Umin, SIDX == 0);
#pragma unroll
            for (int j = 0; j < W; j++) Y[j] = wa[j];
        }
        {
            float wb[W];
            ok = rank_chunk_window<RawT, FULLCH, DEV, CALIB>(prm, fs[CALIB ? kb : 0], kb, K, cb, base, lane, m1, m2, wb, Lmax, Umin, false) && ok;
#pragma unroll
            for (int j = 0; j < W; j++) Y[W + j] = wb[j];
        }
        window_net_from<2 * W, W, W / 2, W / 2 + W>(Y);      // the pair's 64 window values: the 32 middle ones (window_step above)
#pragma unroll
        for (int j = 0; j < W; j++) win[j] = Y[W / 2 + j];
        Lmax = fmaxf(Lmax, win[0]);
        Umin = fminf(Umin, win[W - 1]);
        below += (ca / 2 - W / 2) + (cb / 2 - W / 2) + W / 2;
        __builtin_amdgcn_sched_barrier(0);
        return ok;
    }
}

// MODE 0: tmp[p], tmp[P + p] = the column's two middle values (NaN, -: not sure).  MODE 1: tmp[2 P + p] = the sum of the two middle
// values of |(x - m1) + (x - m2)| (not sure: tmp[p] = NaN).  MODE 2: the median plane (+ count = N); a pixel that is not sure is listed.
// MODE 3: MODE 2 with fused calibration (ApCalibrate.py:439-464 on the fast path, the chunk kernel's loader).
template <int KS, int NPAIR, typename RawT, bool FULLCH, int MODE>
__global__ __launch_bounds__(256, 2) void stack_rank_chunks_kernel(const StackParams prm, float *tmp, int32_t *redo_count, int32_t *redo_list)
{
    constexpr int K = KS + NPAIR, W = kChunkWin;
    constexpr bool DEV = MODE == 1, CALIB = MODE == 3;
    __shared__ FrameScalars<kChunkSlots> fs[CALIB ? K : 1];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int64_t p = base + lane;
    const int N = prm.N;
    const int cbase = N / K, cextra = N % K;
    if constexpr (CALIB) {                                  // the per-frame scalars of every chunk, as stack_chunks_kernel stages them
        for (int k = 0; k < K; k++) {
            const int c = cbase + (k < cextra ? 1 : 0);
            for (int t = threadIdx.x; t < kChunkSlots; t += blockDim.x) {
                const int ff = k + (t < c ? t : c - 1) * K;
                fs[k].e[t] = prm.exp_ratio ? prm.exp_ratio[ff] : 0.f;
                fs[k].ped[t] = prm.pedestal ? prm.pedestal[ff] : 0.f;
                fs[k].pad[t] = t < c ? -__builtin_inff() : __builtin_inff();
            }
        }
        __syncthreads();
    }
    if (p >= prm.P) return;
    extern __shared__ float parked[];                       // [(KS - 2) * W][256]
    float m1 = 0.f, m2 = 0.f;
    bool ok = true;
    if constexpr (MODE >= 2) ok = !(prm.pixmask && prm.pixmask[p]);     // a masked pixel: the exact kernel's answer
    if constexpr (DEV) {
        m1 = tmp[p];
        m2 = tmp[prm.P + p];
        ok = m1 == m1;
    }
    float R0[W], R1[W];
    float Lmax = 0.f, Umin = 0.f;
    int below = 0;
    if constexpr (KS >= 3) {
        float win[W];
        ok = rank_window_step<RawT, FULLCH, DEV, CALIB, 0, KS, NPAIR>(prm, fs, cbase, cextra, base, lane, m1, m2, win, Lmax, Umin, below) && ok;
#pragma unroll
        for (int j = 0; j < W; j++) parked[j * 256 + lane] = win[j];
    }
    if constexpr (KS == 4) {
        float win[W];
        ok = rank_window_step<RawT, FULLCH, DEV, CALIB, 1, KS, NPAIR>(prm, fs, cbase, cextra, base, lane, m1, m2, win, Lmax, Umin, below) && ok;
#pragma unroll
        for (int j = 0; j < W; j++) parked[(W + j) * 256 + lane] = win[j];
    }
    ok = rank_window_step<RawT, FULLCH, DEV, CALIB, KS - 2, KS, NPAIR>(prm, fs, cbase, cextra, base, lane, m1, m2, R0, Lmax, Umin, below) && ok;
    ok = rank_window_step<RawT, FULLCH, DEV, CALIB, KS - 1, KS, NPAIR>(prm, fs, cbase, cextra, base, lane, m1, m2, R1, Lmax, Umin, below) && ok;
    float X[4 * W];
    {
        int slot = lane;
        asm volatile("" : "+v"(slot) : : "memory");
#pragma unroll
        for (int j = 0; j < (KS - 2) * W; j++) X[j] = parked[j * 256 + slot];
#pragma unroll
        for (int j = 0; j < W; j++) {
            X[(KS - 2) * W + j] = R0[j];
            X[(KS - 1) * W + j] = R1[j];
        }
#pragma unroll
        for (int j = KS * W; j < 4 * W; j++) X[j] = __builtin_inff();
    }
    constexpr int MLO = 16 * KS - 8;
    window_net_from<4 * W, W, MLO, MLO + 16>(X);
    float M[16];
#pragma unroll
    for (int j = 0; j < 16; j++) M[j] = X[MLO + j];
    const int mbase = below + MLO;
    const int i1 = ((N - 1) >> 1) - mbase, i2 = (N >> 1) - mbase;
    bool unsure = !ok || i1 < 0 || i2 > 15;
    const float a = pick_rel<0, 16, 16>(M, i1 & 15), b = pick_rel<0, 16, 16>(M, i2 & 15);
    unsure = unsure || !(a >= Lmax && b <= Umin);           // outside the zone the windows vouch for
    if constexpr (MODE == 0) {
        tmp[p] = unsure ? __builtin_nanf("") : a;
        tmp[prm.P + p] = b;
    } else if constexpr (MODE == 1) {
        if (unsure) tmp[p] = __builtin_nanf("");
        else tmp[2 * prm.P + p] = a + b;
    } else {                                                // MODE 2 / 3: the median plane
        const uint64_t m = __builtin_amdgcn_ballot_w64(unsure);
        if (m != 0) {
            int first = 0;
            if (__builtin_amdgcn_readfirstlane(lane) == lane) first = atomicAdd(redo_count, (int)__builtin_popcountll(m));
            first = __builtin_amdgcn_readfirstlane(first);
            const int mine = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (unsure) {
                redo_list[first + mine] = (int32_t)p;
                return;
            }
        }
        if (prm.median) prm.median[p] = (float)(((double)a + (double)b) / 2.0);
        if (prm.count) prm.count[p] = N;
    }
}

// Pass 3 of the median / mad_std configuration: which values the bounds keep, and their float64 sums (stack_mad_fast_kernel's
// arithmetic on a streamed column).
template <typename RawT>
__global__ __launch_bounds__(256) void stack_mad_sums_kernel(const StackParams prm, const float *__restrict__ tmp, float cl, float cu,
                                                             int32_t *redo_count, int32_t *redo_list)
{
    const int lane = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 256 + lane;
    if (p >= prm.P) return;
    const float m1 = tmp[p], m2 = tmp[prm.P + p];
    bool unsure = !(m1 == m1);
    const float E = unsure ? 0.f : tmp[2 * prm.P + p];      // 4 MAD
    const float rho = 0x1p-20f;
    const float tl = cl * E, th = cu * E;
    const float tl_hi = __builtin_fmaf(tl, rho, tl), tl_lo = __builtin_fmaf(tl, -rho, tl);
    const float th_hi = __builtin_fmaf(th, rho, th), th_lo = __builtin_fmaf(th, -rho, th);
    const RawT *fp = static_cast<const RawT *>(prm.frames) + p;
    const int N = prm.N;
    const double c = (double)m1;
    double S = 0.0, Q = 0.0;
    float dmax = 0.f;
    int n = 0;
    for (int f0 = 0; f0 < N; f0 += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = stream_value<RawT>(fp, f0 + j, N, prm.stride);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool valid = f0 + j < N;
            const float e = (x[j] - m1) + (x[j] - m2);     // 2 (x - base)
            const bool rej_lo = -e > tl_hi, keep_lo = -e <= tl_lo;
            const bool rej_hi = e > th_hi, keep_hi = e <= th_lo;
            unsure = unsure || !(rej_lo || keep_lo) || !(rej_hi || keep_hi);   // (a non-finite x never gets here sure: pass 1 gave the pixel up)
            const bool in = valid && !rej_lo && !rej_hi;
            const double d = in ? (double)x[j] - c : 0.0;
            S += d;
            Q = fma(d, d, Q);
            n += in ? 1 : 0;
            dmax = __builtin_fmaxf(dmax, __builtin_fabsf(e));
        }
    }
    unsure = unsure || !(dmax == 0.f || (dmax > 0x1p-40f && dmax < 0x1p40f));
    const uint64_t m = __builtin_amdgcn_ballot_w64(unsure);
    if (m != 0) {
        int first = 0;
        if (__builtin_amdgcn_readfirstlane(lane) == lane) first = atomicAdd(redo_count, (int)__builtin_popcountll(m));
        first = __builtin_amdgcn_readfirstlane(first);
        const int mine = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (unsure) {
            redo_list[first + mine] = (int32_t)p;
            return;
        }
    }
    const double nn = (double)n;
    const double mean = c + S / nn;
    if (prm.mean) prm.mean[p] = (float)mean;
    if (prm.count) prm.count[p] = n;
    if (prm.mean64) prm.mean64[p] = mean;
    if (prm.std64) {
        const double var = (Q - S * S / nn) / nn;
        prm.std64[p] = sqrt(var > 0.0 ? var : 0.0);
    }
}

// Whether a stack of 129 .. 512 frames is the plain median or the one-pass median / mad_std configuration on raw frames
// (mad_fast_eligible's conditions without the workspace: the list may be a stream-ordered temporary here).
bool rank_chunks_eligible(const StackParams &prm, bool calib, bool median_only)
{
    if (prm.N <= 128 || prm.N > 8 * kChunkSlots || prm.P >= 0x7fffffffLL) return false;
    if (getenv("APGPU_RANK_CHUNKS_OFF")) return false;      // development: time the exact kernel on the same call
    if (median_only) return calib || !prm.pedestal;         // fused calibration and a pixel mask: the median only (MODE 3 / the list)
    if (calib || prm.pixmask || prm.pedestal) return false;
    if (prm.dev != APGPU_DEV_MAD_STD || prm.center != APGPU_CENTER_MEDIAN || prm.maxiters != 1) return false;
    if (prm.median || prm.std || prm.moments || prm.single_kernel || prm.fast32 == 0) return false;
    if (!(prm.sl2 > 0.0 && prm.su2 > 0.0 && prm.sl2 < 1e12 && prm.su2 < 1e12)) return false;
    return true;
}

template <int KS, int NPAIR, typename RawT, int MODE>
static int launch_rank_pass(const StackParams &q, bool fullch, float *tmp, int32_t *cnt, int32_t *list, hipStream_t st)
{
    const int64_t grid = (q.P + 255) / 256;
    const size_t lds = (size_t)(KS - 2) * kChunkWin * 256 * sizeof(float);
    const void *kern = fullch ? reinterpret_cast<const void *>(stack_rank_chunks_kernel<KS, NPAIR, RawT, true, MODE>)
                              : reinterpret_cast<const void *>(stack_rank_chunks_kernel<KS, NPAIR, RawT, false, MODE>);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return kNoRedoList;                             // no room for the parked windows: the caller takes the exact kernel
        }
    }
    if (fullch) hipLaunchKernelGGL((stack_rank_chunks_kernel<KS, NPAIR, RawT, true, MODE>), dim3((unsigned)grid), dim3(256), lds, st, q, tmp, cnt, list);
    else hipLaunchKernelGGL((stack_rank_chunks_kernel<KS, NPAIR, RawT, false, MODE>), dim3((unsigned)grid), dim3(256), lds, st, q, tmp, cnt, list);
    return check_launch("stack kernel (chunked order statistics, 129..512 frames)");
}

template <int KS, int NPAIR, typename RawT>
static int launch_rank_k(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe)
{
    const bool fullch = prm.N == (KS + NPAIR) * kChunkSlots;
    if (describe) {
        snprintf(describe, 256, "stack_rank_chunks_kernel<%d, %d, %s, %s, %d>", KS, NPAIR, sizeof(RawT) == 2 ? "unsigned short" : "float",
                 fullch ? "true" : "false", median_only ? (calib ? 3 : 2) : 0);
        return APGPU_OK;
    }
    if ((prm.P + 255) / 256 > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
    int32_t *ws = prm.redo, *cnt = nullptr, *list = nullptr, *own = nullptr;
    StackParams q = prm;
    q.redo = nullptr;
    auto exact = [&]() { return launch_big_exact(q, u16, calib, median_only, st, nullptr); };
    if (ws) {
        cnt = ws;
        list = ws + ws_list_off(prm.P);
    } else {
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&own), (size_t)(prm.P + 1) * sizeof(int32_t), st);
        if (e == hipSuccess) {
            e = hipMemsetAsync(own, 0, sizeof(int32_t), st);
            if (e != hipSuccess) (void)hipFreeAsync(own, st);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return exact();
        }
        cnt = own;
        list = own + 1;
    }
    float *tmp = nullptr;
    if (!median_only) {
        const hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&tmp), (size_t)prm.P * 3 * sizeof(float), st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (own) (void)hipFreeAsync(own, st);
            return exact();
        }
    }
    int rc;
    bool launched = false;                                   // something may have been listed: the exact kernel must follow
    if (median_only) {
        rc = calib ? launch_rank_pass<KS, NPAIR, RawT, 3>(q, fullch, nullptr, cnt, list, st) : launch_rank_pass<KS, NPAIR, RawT, 2>(q, fullch, nullptr, cnt, list, st);
        launched = rc == APGPU_OK;
    } else {
        rc = launch_rank_pass<KS, NPAIR, RawT, 0>(q, fullch, tmp, cnt, list, st);
        if (rc == APGPU_OK) rc = launch_rank_pass<KS, NPAIR, RawT, 1>(q, fullch, tmp, cnt, list, st);
        if (rc == APGPU_OK) {
            const float cl = (float)(sqrt(prm.sl2) * 1.482602218505602 * 0.5), cu = (float)(sqrt(prm.su2) * 1.482602218505602 * 0.5);
            hipLaunchKernelGGL((stack_mad_sums_kernel<RawT>), dim3((unsigned)((prm.P + 255) / 256)), dim3(256), 0, st, q, tmp, cl, cu, cnt, list);
            rc = check_launch("stack kernel (chunked median / mad_std: sums)");
            launched = rc == APGPU_OK;
        }
    }
    if (rc == kNoRedoList) rc = exact();                     // (nothing was launched, nothing listed)
    else if (launched) rc = launch_big_exact(q, u16, calib, median_only, st, nullptr, cnt, list, ws);
    if (tmp) {
        const hipError_t ef = hipFreeAsync(tmp, st);
        if (rc == APGPU_OK && ef != hipSuccess) rc = fail(APGPU_ELAUNCH, "stack (chunks): free: %s", hipGetErrorString(ef));
    }
    if (own) {
        const hipError_t ef = hipFreeAsync(own, st);
        if (rc == APGPU_OK && ef != hipSuccess) rc = fail(APGPU_ELAUNCH, "stack (chunks): free: %s", hipGetErrorString(ef));
    }
    return rc;
}

template <int KS, int NPAIR>
static int launch_rank_t(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe)
{
    return u16 ? launch_rank_k<KS, NPAIR, uint16_t>(prm, u16, calib, median_only, st, describe) : launch_rank_k<KS, NPAIR, float>(prm, u16, calib, median_only, st, describe);
}

int launch_rank_chunks(const StackParams &prm, bool u16, bool calib, bool median_only, hipStream_t st, char *describe)
{
    switch (chunk_count(prm.N)) {
    case 3: return launch_rank_t<3, 0>(prm, u16, calib, median_only, st, describe);
    case 4: return launch_rank_t<4, 0>(prm, u16, calib, median_only, st, describe);
    case 5: return launch_rank_t<4, 1>(prm, u16, calib, median_only, st, describe);
    case 6: return launch_rank_t<3, 3>(prm, u16, calib, median_only, st, describe);
    case 7: return launch_rank_t<4, 3>(prm, u16, calib, median_only, st, describe);
    default: return launch_rank_t<4, 4>(prm, u16, calib, median_only, st, describe);
    }
}

}  // namespace apgpu_stack
