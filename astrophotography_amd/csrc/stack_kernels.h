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
#include "stack_reduce.h"

namespace apgpu_stack {

using namespace apgpu;

// EXTRA: the rich kernels (sorted column parked in LDS: mad_std, float64 planes).  PLUS (lean only): median and std planes
// straight from the register-resident column - the same kernel as the benchmarked one with a longer epilogue.
// (Round 4, measured and dropped: workgroups of 64 lanes, and persistent workgroups walking the tiles with stride gridDim.x
// so that a finished wavefront's slot is refilled by its own next tile - either pushes the 64-slot kernel over 168 VGPRs,
// i.e. to two wavefronts per SIMD: 1.09 / 1.24 ms against 1.02 on the same box.)
// (round 5: the complete lean kernels - plain launch and redo pass in one loop - are held at the three-wavefront budget of 168
// VGPRs: left alone the loop costs them 174-183 (loop-invariant values kept in registers across the body), i.e. two wavefronts
// per SIMD and 10-14 % of their speed; with machine LICM off for these translation units (_build.py) the cap costs the full
// 64-slot kernel nothing and the padded / PLUS ones 2-4 spilled registers)
#ifndef APGPU_LEAN_MIN_BLOCKS
#define APGPU_LEAN_MIN_BLOCKS 3
#endif
#ifndef APGPU_PLUS_MIN_BLOCKS
#define APGPU_PLUS_MIN_BLOCKS 3
#endif
// (lean kernels of 72 .. 128 slots: two wavefronts per SIMD = at most 256 VGPRs; left alone the padded 112- and 120-slot
// kernels take 258 - ONE wavefront per SIMD, 4.0 ms where the full 112-slot kernel takes 2.2)
#ifndef APGPU_WIDE_MIN_BLOCKS
#define APGPU_WIDE_MIN_BLOCKS 2
#endif
#ifndef APGPU_PADDED_MIN_BLOCKS
#define APGPU_PADDED_MIN_BLOCKS 3
#endif

// -------------------------------------------------------------------------------------------------
// Workspace of the two-kernel scheme (apgpu_stack_args.workspace, include/apgpu.h; int32 words; StackParams.redo points at it):
//   [0, 4096)                 256 segment lines of 16 words (64 bytes apart, so that their atomics do not share a cache line):
//                             [0] pixels listed, [2] arrivals of the redo pass, [3] "blocks of tiles with tile % 256 == this
//                             segment were given up", [4] / [5] the guard's window (alert mode only): pixels listed and
//                             (window number << 16 | 64-pixel blocks finished) by the window's wavefronts - fast_block_bails
//   [4096, 4112)              int64 statistics, cumulative over the calls that used this workspace (never reset by the library):
//                             calls, pixels, pixels listed, 64-pixel blocks given up (APGPU_STACK_WS_STATS_OFFSET = 4 * 4096)
//   [4112, 4128)              the call line: [0] segments the redo pass has finished, [1] pixels listed by this call, [2] blocks
//                             given up by this call, [3] the MODE the next call's fast kernel starts in (0 = alert: every
//                             wavefront checks its segment's counter before it loads anything; 1 = quiet: the previous call on
//                             this workspace sent under an eighth of its pixels to the redo pass, no check - fast_block_bails)
//   [4128, list_off)          one int32 flag per 64-pixel block (four per 256-pixel tile): the fast kernel's wavefront gave the
//                             block up, the redo pass does it whole
//   [list_off, ...)           256 segments of redo_seg_capacity(P) pixel indices (segment = workgroup % 256)
// Everything below list_off is ZERO between calls, the statistics and the mode word excepted: the caller zeroes it once, the redo
// pass leaves it zero again (its last workgroup per segment clears what the call used).  The layout depends on P: a workspace
// serves one image size at a time.
// -------------------------------------------------------------------------------------------------
constexpr int kRedoSegs = 256;
constexpr int kWsLine = 16;
constexpr int kWsStats = kRedoSegs * kWsLine;
constexpr int kWsCall = kWsStats + 16;
// words of the workspace's call line that belong to the median / mad_std pair (stack_mad.hip): the blocks given up among the sampled
// tiles of this call (zero between calls), and the mode the next call starts in (0: the fast kernel runs on every tile; 1: the last
// call gave up more than an eighth of its sampled blocks - the fast kernel runs on every 16th tile only and hands the rest over)
constexpr int kWsMadCount = 8, kWsMadMode = 10, kMadSample = 16;
constexpr int kWsFlags = kWsCall + 16;
constexpr int kModeAlert = 0, kModeQuiet = 1;
// entries a segment can receive: the pixels of its workgroups (every kRedoSegs-th of the P / 256 tiles; one tile more for
// the pair kernels, whose workgroups cover 512 pixels)
__host__ __device__ inline int64_t redo_seg_capacity(int64_t P) { return (((P + 255) / 256 + kRedoSegs - 1) / kRedoSegs + 1) * 256; }
__host__ __device__ inline int64_t ws_flag_words(int64_t P) { return ((4 * ((P + 255) / 256 + 2) + 15) / 16) * 16; }
__host__ __device__ inline int64_t ws_list_off(int64_t P) { return kWsFlags + ws_flag_words(P); }
__host__ __device__ inline int64_t ws_total_words(int64_t P) { return ws_list_off(P) + (int64_t)kRedoSegs * redo_seg_capacity(P); }

// The kernel's argument block as a value, every field by its own scalar load from the kernarg segment (late_params: opaque,
// so the loads happen where this is called).
__device__ __forceinline__ StackParams read_params(LateParams *kp)
{
    StackParams q;
    q.frames = kp->frames; q.bias = kp->bias; q.dark = kp->dark; q.nflat = kp->nflat; q.exp_ratio = kp->exp_ratio;
    q.pedestal = kp->pedestal; q.pixmask = kp->pixmask; q.mean = kp->mean; q.median = kp->median; q.std = kp->std;
    q.moments = kp->moments; q.count = kp->count; q.P = kp->P; q.stride = kp->stride; q.sl2 = kp->sl2; q.su2 = kp->su2;
    q.N = kp->N; q.still_biased = kp->still_biased; q.center = kp->center; q.dev = kp->dev; q.maxiters = kp->maxiters;
    q.moments64 = kp->moments64; q.mean64 = kp->mean64; q.std64 = kp->std64; q.single_kernel = kp->single_kernel;
    q.fast32 = kp->fast32; q.redo = kp->redo;
    return q;
}

#ifndef APGPU_QUIET_DEN
#define APGPU_QUIET_DEN 8
#endif
#ifndef APGPU_REDO_RELAXED
#define APGPU_REDO_RELAXED 1
#endif

// A coherent read of a counter other workgroups bump with atomics (agent scope: past the non-coherent vector / scalar caches).
__device__ __forceinline__ int ws_load(const int32_t *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The complete kernels.  EXTRA: the rich kernel (sorted column parked in LDS: mad_std, float64 planes), one tile per workgroup.
// Otherwise the lean kernel (float32 fast path with the exact path behind it, wavefront by wavefront), in one of two roles:
//   prm.redo == nullptr   the whole stack, workgroup b reduces the tile [256 b, 256 b + 256);
//   prm.redo != nullptr   the REDO PASS behind stack_fast_kernel / stack_fast_u16_pairs_kernel (round 5; round 4 had a separate
//                         stack_redo_kernel family - a third of the library's code): a fixed grid of kRedoSegs * k workgroups;
//                         workgroup b serves segment b % kRedoSegs with the k - 1 others of that segment:
//                         (1) the segment's LISTED PIXELS, one per lane (a gather: base 0, the pixel index as the lane offset),
//                             exact clip only - a listed pixel has failed the float32 path already;
//                         (2) if the segment's word [3] is set, the FLAGGED 64-PIXEL BLOCKS of its tiles (tile % 256 ==
//                             segment; wavefront w of the workgroup looks at block w of each tile), whole and coalesced, fast
//                             path allowed - what the fast kernel's wavefronts gave up without trying (fast_block_bails:
//                             more than an eighth of the segment's pixels were failing, or the frames of a uint16 pair stack
//                             do not share one exposure ratio).  This is the guard against the two-kernel scheme's worst
//                             case: a stack whose pixels mostly fail costs one pass of THIS kernel plus a few rounds of the
//                             fast one, not both in full.  A wavefront reads the flags of its next 64 tiles with ONE load
//                             and walks the set bits: one flag per load left a memory round trip in front of every block.
//                         (3) the last workgroup of a segment to finish clears the segment's counters and flags and adds to
//                             the statistics; the last segment to finish sets the mode of the next call's fast kernel.
// One body serves all three item kinds (one copy of the code): a small wave-uniform state machine picks the next item.
template <int NP, typename RawT, bool CALIB, bool EXTRA, bool FULL, bool PLUS = false>
// (padded lean kernels: capped at the three-wavefront register budget - 172 VGPRs uncapped -, the spills land in the
// exact-fallback blocks; full ones need 166 by themselves)
__global__ __launch_bounds__(EXTRA ? rich_block<NP>() : 256, NP <= 64 ? (EXTRA ? 2 : (PLUS ? APGPU_PLUS_MIN_BLOCKS : (FULL ? APGPU_LEAN_MIN_BLOCKS : APGPU_PADDED_MIN_BLOCKS))) : ((EXTRA || PLUS) ? 1 : APGPU_WIDE_MIN_BLOCKS)) void stack_sigclip_kernel(const StackParams prm)
{
    const int lane = threadIdx.x;
    __shared__ FrameScalars<NP> fs;
    __shared__ ColumnLds<NP, EXTRA> cols;        // lane-private columns: no barrier around their use
    // Full calibrated stacks without pedestals read the exposure ratios with scalar loads and stage nothing: no barrier before
    // the first frame load.  (Round 4 also tried issuing the loads BEFORE the barrier of the staged kernels: the first
    // wavefront's wait for its ratios is an in-order vmcnt(0) that then covers its 67 column loads as well - worse.)
    constexpr int MINN = padded_minn(NP, FULL);
    if (needs_staging<CALIB, FULL, NP>(prm)) stage_frame_scalars<NP>(prm, fs);
    if constexpr (EXTRA) {
        const int64_t base = (int64_t)blockIdx.x * blockDim.x;
        const int64_t p = base + lane;
        {
            if (prm.flag_mode) {                                 // behind stack_mad_fast_kernel: only the blocks it could not finish
                if (blockIdx.x == 0 && lane == 0) {
                    // the guard of that pair (stack_mad.hip): the fast kernel counted the blocks it gave up among every 16th tile;
                    // more than an eighth of them -> the next call's fast kernel only samples (mode 1), else it runs in full
                    const int64_t ntiles = (prm.P + 255) / 256;
                    const int64_t sampled = 4 * ((ntiles + kMadSample - 1) / kMadSample);
                    const int given = prm.redo[kWsCall + kWsMadCount];
                    prm.redo[kWsCall + kWsMadMode] = ((int64_t)given * 8 > sampled) ? 1 : 0;
                    prm.redo[kWsCall + kWsMadCount] = 0;
                }
                int32_t *const fl = prm.redo + kWsFlags + (base >> 6) + (lane >> 6);      // one flag per 64 pixels, whatever the workgroup's size
                if (*fl == 0) return;
                if ((lane & 63) == 0) *fl = 0;
            }
        }
        if (p >= prm.P) return;
        float v[NP];
        APGPU_MARK("load_calibrate_sort");
        const int n = load_sorted_column<NP, RawT, CALIB, true, FULL>(prm, fs, base, lane, v);
        reduce_and_store_rich<NP>(prm, v, n, p, cols.lane_ptr(lane));
    } else {
        enum { kPlain = 0, kList = 1, kTiles = 2, kDone = 3 };
        const bool redo_pass = prm.redo != nullptr;
        const int seg = blockIdx.x % kRedoSegs, rank = blockIdx.x / kRedoSegs, wv = __builtin_amdgcn_readfirstlane(lane >> 6);
        int nitems = 0, flagged = 0, phase = kPlain, cursor = 0, nblocks_done = 0;
        uint64_t pending = 0;                                // kTiles: the flagged ones among the 64 tiles before `cursor`
        if (redo_pass) {
            nitems = __builtin_amdgcn_readfirstlane(ws_load(prm.redo + seg * kWsLine));
            flagged = __builtin_amdgcn_readfirstlane(ws_load(prm.redo + seg * kWsLine + 3));      // blocks of tiles with tile % 256 == seg were given up
            phase = kList;
            cursor = rank * 256;
        }
#pragma unroll 1
        for (;;) {
            int64_t base = 0;
            int off = 0;
            bool valid = false, fastok = true;
            if (phase == kPlain) {
                base = (int64_t)blockIdx.x * 256;
                off = lane;
                valid = base + off < prm.P;
                phase = kDone;
            } else if (phase == kList) {
                if (cursor >= nitems) {
                    phase = flagged ? kTiles : kDone;
                    cursor = rank;
                    continue;
                }
                LateParams *const kp = late_params();
                const int32_t *const list = kp->redo + ws_list_off(kp->P) + (int64_t)seg * redo_seg_capacity(kp->P);
                valid = cursor + lane < nitems;
                off = valid ? list[cursor + lane] : 0;
                fastok = false;
                cursor += (int)(gridDim.x / kRedoSegs) * 256;
            } else if (phase == kTiles) {
                // this workgroup's tiles of the segment: tile(c) = seg + c * kRedoSegs, c = rank, rank + per_seg, ...
                LateParams *const kp = late_params();
                const int per_seg = (int)(gridDim.x / kRedoSegs);
                const int64_t ntiles = (kp->P + 255) / 256;
                if (pending == 0) {
                    if (seg + (int64_t)cursor * kRedoSegs >= ntiles) {
                        phase = kDone;
                        continue;
                    }
                    // lane i: the flag of this wavefront's block in the i-th of the next 64 tiles
                    const int64_t t = seg + ((int64_t)cursor + (int64_t)(lane & 63) * per_seg) * kRedoSegs;
                    const int f = t < ntiles ? kp->redo[kWsFlags + 4 * t + wv] : 0;
                    pending = __builtin_amdgcn_ballot_w64(f != 0);
                    cursor += 64 * per_seg;
                    continue;
                }
                const int i = __builtin_ctzll(pending);
                pending &= pending - 1;
                const int64_t tile = seg + ((int64_t)cursor - (int64_t)(64 - i) * per_seg) * kRedoSegs;
                nblocks_done++;
                base = tile * 256 + wv * 64;
                off = lane & 63;
                valid = base + off < kp->P;
            } else {
                break;
            }
            if (valid) {
                const int64_t p = base + off;
                float v[NP];
                APGPU_MARK("load_calibrate_sort");
                // the arguments are read from the kernarg segment HERE, once per item: as values that live across the loop they
                // cost ~40 SGPRs that the body then spills to VGPR lanes (175 instead of 166 VGPRs: two wavefronts per SIMD)
                const StackParams q = read_params(late_params());
                if constexpr (fast32_possible(NP, MINN)) {
                    // full stacks headed for the float32 fast path only sort what it reads (pruned network); `pruned` tells the
                    // reduction to complete the sort should it have to fall back to the exact path
                    bool pruned = fastok && fast32_wanted(q);
                    const int n = load_sorted_column<NP, RawT, CALIB, true, FULL, MINN, kFastTail>(q, fs, base, off, v, &pruned);
                    reduce_and_store<NP, MINN, PLUS>(q, v, n, p, pruned, fastok);
                } else if constexpr (fast32_possible_padded(NP, MINN)) {
                    // padded stacks: split pads (-inf below, +inf above the real values) and tails of 8 - see fast32_possible_padded
                    bool pruned = fastok && fast32_wanted(q);
                    const int n = load_sorted_column<NP, RawT, CALIB, true, FULL, MINN, fast_tail_padded(NP), false, true>(q, fs, base, off, v, &pruned, EarlyLoads<NP, RawT>(), fastok);
                    reduce_and_store<NP, MINN, PLUS>(q, v, n, p, pruned, fastok);
                } else {
                    const int n = load_sorted_column<NP, RawT, CALIB, true, FULL>(q, fs, base, off, v);
                    reduce_and_store<NP, MINN, PLUS>(q, v, n, p, false, fastok);
                }
            }
        }
        if (redo_pass) {
            // The segment's last workgroup to get here clears what the call used and books the call's statistics; the last
            // segment sets the next call's mode.  Every workgroup of the segment has read the counters by then (at its start);
            // the block flags are read during the walk, so a workgroup that walked tiles waits for all its wavefronts before it
            // reports in.  (The common case - nothing flagged - is one atomic by one lane and no barrier: with a barrier pair here
            // the 4096 mostly idle workgroups of the benchmark's redo pass took 80 us where its one busy wavefront per segment
            // needs 20.)
            LateParams *const kp = late_params();
            int32_t *const ws = kp->redo;
            unsigned long long *const stats = reinterpret_cast<unsigned long long *>(ws + kWsStats);
            if (nblocks_done && (lane & 63) == 0) {
                atomicAdd(stats + 3, (unsigned long long)nblocks_done);
                int got = atomicAdd(ws + kWsCall + 2, nblocks_done);
                asm volatile("" : "+v"(got));                    // performed before the barrier below
            }
            if (flagged) __syncthreads();
            if (lane == 0) {
                const int per_seg = (int)(gridDim.x / kRedoSegs);
#if APGPU_REDO_RELAXED
                // Relaxed atomics, ordered by their RETURN VALUES where order matters (an atomic whose result has arrived has been
                // performed at the memory side): a release / acquire pair here is an L2 write-back + invalidate per workgroup on this
                // chip - 768 of them per call for counters that only other atomics and the next kernel ever read.  What has to hold:
                // a workgroup has READ its segment's counters and flags before it reports in (it has used their values), every
                // wavefront's block totals are performed before the workgroup reports in (the barrier above waits for them), and a
                // segment's totals are performed before the segment counts itself as done (`sync` below).
                const int arrived = atomicAdd(ws + seg * kWsLine + 2, 1);
                int sync = 1;
#else
                const int arrived = __hip_atomic_fetch_add(ws + seg * kWsLine + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
                if (arrived == per_seg - 1) {
                    if (nitems) {
                        atomicAdd(stats + 2, (unsigned long long)nitems);
#if APGPU_REDO_RELAXED
                        int got = atomicAdd(ws + kWsCall + 1, nitems);
                        asm volatile("" : "+v"(sync), "+v"(got));   // `sync` exists only after the total has been performed
#else
                        atomicAdd(ws + kWsCall + 1, nitems);
#endif
                    }
                    if (flagged) {
                        const int64_t ntiles = (kp->P + 255) / 256;
                        for (int64_t t = seg; t < ntiles + 1; t += kRedoSegs) *reinterpret_cast<int4 *>(ws + kWsFlags + 4 * t) = make_int4(0, 0, 0, 0);
                    }
                    ws[seg * kWsLine] = 0;
                    ws[seg * kWsLine + 1] = 0;
                    ws[seg * kWsLine + 2] = 0;
                    ws[seg * kWsLine + 3] = 0;
                    ws[seg * kWsLine + 4] = 0;
                    ws[seg * kWsLine + 5] = 0;
#if APGPU_REDO_RELAXED
                    const int segs_done = atomicAdd(ws + kWsCall, sync);
#else
                    const int segs_done = __hip_atomic_fetch_add(ws + kWsCall, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
                    if (segs_done == kRedoSegs - 1) {        // the call is complete: totals -> statistics, and the next call's mode
                        const long long listed = ws_load(ws + kWsCall + 1), blocks = ws_load(ws + kWsCall + 2);
                        atomicAdd(stats + 0, 1ull);
                        atomicAdd(stats + 1, (unsigned long long)kp->P);
                        // quiet: under an eighth of the pixels went to the redo pass, listed or in blocks given up - below that the
                        // list route costs within 2 % of the complete kernel (profiles/r05/redo_sweep.txt) and the guard has nothing
                        // to save, while its look costs every wavefront of a short kernel a memory round trip (C5's 16-frame stacks
                        // list 6 % of their pixels, the NaN blocks around bad pixels, in every call)
                        ws[kWsCall + 3] = ((blocks * 64 + listed) * APGPU_QUIET_DEN < kp->P) ? kModeQuiet : kModeAlert;
                        ws[kWsCall] = 0;
                        ws[kWsCall + 1] = 0;
                        ws[kWsCall + 2] = 0;
                    }
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Round 4: the FAST kernel and its redo list.  stack_sigclip_kernel carries, next to the float32 fast path, everything that
// path may have to fall back to - the exact calibration with its reload, the complete sorting network, the float64 clip,
// the pedestal body - and the register allocation of the whole function (166 VGPRs: three wavefronts per SIMD) is set by
// code the benchmark data runs in 0.6 % of its wavefronts.  This kernel holds ONLY the fast path (float32 frames, median
// centre / std deviation, lean outputs, no pedestals): masters + frame loads, packed calibration with scalar-load ratios,
// pruned network, clip_fast32, outputs - 104 VGPRs, FOUR wavefronts per SIMD, no LDS, no barrier.  A PIXEL that cannot
// finish (outside the calibration's guards, too many non-finite values, masked, a comparison inside the float32 margins, a
// fifth value to trim, the image's last partial tile) stores nothing and is appended to prm.redo; stack_redo_kernel - the
// complete path of stack_sigclip_kernel, exact clip only, one listed pixel per lane - walks that list.  The list is per
// pixel, not per wavefront: the exact path costs a wavefront the same whether one lane needs it or all 64, so the failing
// lanes of many wavefronts are compacted into few.  Results are those of stack_sigclip_kernel bit for bit: the same
// functions run, only in two kernels.  Same-box A/B on the 64-frame benchmark: profiles/r04/ab_fast_kernel.txt.
//
// Stacks WITHOUT the fused calibration (CALIB = false: frames that have been through ApCalibrate or a resample) may hold
// non-finite values anywhere - a resampled frame has a 6 x 6 block of NaNs around every masked input pixel, so with 16 frames
// nearly every wavefront of the C5 share holds one.  There every non-finite value becomes the +inf sentinel BEFORE the sort
// (two instructions per value), sorts to the top like a padding slot, and the lane's clip starts with its own count of them
// trimmed (clip_fast32, MODE 2; up to T - 1 per column, more go to the list).
// -------------------------------------------------------------------------------------------------
#ifndef APGPU_FAST_MIN_BLOCKS
#define APGPU_FAST_MIN_BLOCKS 4
#endif

// Appends the failing lanes' pixels to the list: one atomic per wavefront that has any.  kRedoSegs counters, one cache line
// apart, each with its own stretch of the list (segment = workgroup % kRedoSegs, so a segment can never overflow its share):
// a stack whose every pixel fails would otherwise serialise on one address.
__device__ __forceinline__ void redo_push(bool fail, int64_t p, bool alert)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(fail);
    if (m == 0) return;
    LateParams *const kp = late_params();
    int32_t *const redo = kp->redo;
    const int seg = blockIdx.x % kRedoSegs;
    int first = 0;
    if (alert && (threadIdx.x & 63) == 0) atomicAdd(&redo[seg * kWsLine + 4], (int)__builtin_popcountll(m));      // the guard's window
    if ((threadIdx.x & 63) == 0) first = atomicAdd(&redo[seg * kWsLine], (int)__builtin_popcountll(m));
    first = __builtin_amdgcn_readfirstlane(first);
    const int mine = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    if (fail) redo[ws_list_off(kp->P) + (int64_t)seg * redo_seg_capacity(kp->P) + first + mine] = (int32_t)p;
}

// The guard against the two-kernel scheme's worst case (round 5).  A listed pixel is paid for twice - the fast kernel's work on
// it is lost, and the redo pass gathers it through a list - which is a bargain at the benchmark's 0.007 % and a loss when most
// pixels fail (5 or more values to trim per side, frames full of NaNs, pedestal-free data outside the division's guards ...):
// measured on 64 x 4096^2 with a share f of the columns forced off the fast path (profiles/r05/redo_sweep.txt), the list route
// costs 1.65 + 0.8 f ms against 1.70 for the complete kernel alone - 1.4 x at f = 1.  So a WAVEFRONT first looks at how its
// segment has fared: the segment's line holds, for the current WINDOW of kBailWindow consecutive workgroups of the segment, the
// pixels its wavefronts listed and the 64-pixel blocks they finished; once more than kBailNum / kBailDen of the finished pixels
// were listed (and at least kBailMinBlocks blocks are in) this and every later wavefront of the window GIVES ITS BLOCK UP at
// once - one flag store, no loads, no arithmetic - and the redo pass reduces the flagged blocks whole and coalesced with the
// complete kernel's body.  A window's first workgroup starts its counters afresh, so every window begins with kBailMinBlocks
// blocks of honest attempts: a bad REGION (the NaN border of resampled frames, a satellite-free corner of a mosaic) costs its
// own windows, not the image - a first version with one cumulative count and a sticky flag per segment gave up C5's whole
// 8192^2 share because its first 30 rows are the frames' NaN border (8.2 against 5.6 ms per step).  Segments interleave the
// image's tiles (segment = tile % 256), so each samples its window's rows evenly and they tip together.  Flags are per
// wavefront (a 64-pixel block; two for a pair kernel's wavefront): the four wavefronts of a workgroup decide on their own -
// they read the line at different moments and there is no barrier to agree behind.
// The look costs one coherent 8-byte load and its round trip to L2 BEFORE the wavefront's frame loads can be issued: measured
// on the benchmark, +1.5 % (profiles/r05/ab_bail.txt) - for a guard that data like the benchmark's never needs.  Hence the MODE
// word (workspace layout above): the redo pass of every call leaves behind whether the NEXT call on this workspace may skip
// the look (quiet: under an eighth of the pixels redone) or must take it (alert; also the state of a fresh, zeroed workspace
// and of the temporary a call without workspace makes).  The word is read with a scalar load - written by an earlier
// kernel, never during this one - so a quiet call pays nothing (and counts no finished blocks).  The price: the FIRST call
// after the data turned bad runs unguarded (both kernels in full); from the second on the guard holds.
#ifndef APGPU_BAIL_MIN_BLOCKS
#define APGPU_BAIL_MIN_BLOCKS 16
#define APGPU_BAIL_NUM 2
#define APGPU_BAIL_DEN 5
#define APGPU_BAIL_WINDOW_LOG2 7
#endif
constexpr int kBailMinBlocks = APGPU_BAIL_MIN_BLOCKS, kBailNum = APGPU_BAIL_NUM, kBailDen = APGPU_BAIL_DEN;
constexpr int kBailWindowLog2 = APGPU_BAIL_WINDOW_LOG2;    // workgroups of a segment per window (128: 32768 tiles of the image)

// the mode the previous call left behind: a scalar load (constant address space: nothing writes it while this kernel runs)
__device__ __forceinline__ bool fast_kernel_alert()
{
#ifdef APGPU_VARIANT_NO_BAIL
    return false;
#endif
    typedef const int __attribute__((address_space(4))) cint;
    return ((cint *)(uintptr_t)late_params()->redo)[kWsCall + 3] != kModeQuiet;
}

// a wavefront that ran the fast path reports its blocks as finished (alert mode only: the guard's denominator)
template <int TILE>
__device__ __forceinline__ void fast_blocks_done(bool alert, int32_t *ws)
{
    if (alert && (threadIdx.x & 63) == 0) atomicAdd(ws + (blockIdx.x % kRedoSegs) * kWsLine + 5, TILE / 256);
}

// TILE: pixels per workgroup (256; 512 in the pair kernels, whose wavefront covers two blocks).  Block b = pixels [64 b, 64 b + 64);
// its tile (b / 4) belongs to the tile walk of segment (b / 4) % 256 - hence word [3], "this segment has flagged blocks", next to
// word [1], "this segment's wavefronts give up" (segment = workgroup % 256: the same thing only for the 256-pixel kernels).
template <int TILE>
__device__ __forceinline__ void give_blocks_up()
{
    if ((threadIdx.x & 63) != 0) return;
    LateParams *const kp = late_params();
    int32_t *const ws = kp->redo;
    constexpr int BPW = TILE / 256;                          // blocks per wavefront
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * BPW;
    const int64_t nblocks = (kp->P + 63) / 64;
#pragma unroll
    for (int k = 0; k < BPW; k++) {
        if (b0 + k < nblocks) {
            ws[kWsFlags + b0 + k] = 1;
            ws[(int)(((b0 + k) >> 2) % kRedoSegs) * kWsLine + 3] = 1;
        }
    }
}

template <int TILE>
__device__ __forceinline__ bool fast_block_bails(bool alert, int32_t *ws)
{
    if (!alert) return false;                                // (scalar: a quiet call loads nothing)
    const int j = blockIdx.x / kRedoSegs;                    // this workgroup's ordinal within its segment
    const int window = j >> kBailWindowLog2;
    int32_t *const line = ws + (blockIdx.x % kRedoSegs) * kWsLine;
    if ((j & ((1 << kBailWindowLog2) - 1)) == 0) {           // a window's first workgroup starts the counters afresh
        if (threadIdx.x == 0) {
            line[4] = 0;
            line[5] = window << 16;
        }
        return false;
    }
    const uint64_t wc = __hip_atomic_load(reinterpret_cast<const uint64_t *>(line + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int listed = __builtin_amdgcn_readfirstlane((int)(uint32_t)wc);
    const int w5 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(wc >> 32));
    const int done = w5 & 0xffff;
    // (counts of another window - its first workgroup has not got here yet, or stragglers of the last one - say nothing)
    return (w5 >> 16) == window && done >= kBailMinBlocks && (int64_t)listed * kBailDen > (int64_t)done * 64 * kBailNum;
}

// clip_fast32 + the outputs of reduce_and_store's fast branch for the lanes that complete; `good` is cleared for the others.
// PLUS: the median and std planes of the final survivors as well (the three planes of sigma_clipped_stats(axis = 0)).  The
// median: the middle pair of [a, b), inside the window the pruned network delivers.  The std: numpy's two-pass definition,
// sqrt(sum((x - mean)^2) / n) in float64 - with the mean taken from the fast path's own sum, c + S / n (off the exact mean by
// at most ~1e-6 of the column's rms: |dS| <= 19u sqrt(n Q), clip_fast32's budget - a relative 1e-12 on the variance), so only
// ONE float64 pass remains (3 instructions per value where the complete kernel's two passes take 7); identical survivors have
// S = 0 exactly and give exactly 0.
template <int NP, int T, bool CALIB, int PLO = -1, int PHI = -1, bool PLUS = false>
__device__ __forceinline__ void finish_fast_column(const float (&v)[NP], bool &good, int64_t p, int plo, int phi)
{
    LateParams *const kp = late_params();
    int a, b;
    float cf, Sf, Qf;
    const bool done = clip_fast32<NP, T, CALIB ? 1 : 2, PLO, PHI>(v, (float)kp->sl2, (float)kp->su2, kp->maxiters, a, b, cf, Sf, Qf, plo, phi);
    good = good && done;
    if (good) {
        LateParams *const ko = late_params();
        const int cnt = b - a;
        const float nf32 = (float)cnt;
        const float y = __builtin_amdgcn_rcpf(nf32);
        const float q0 = Sf * y;
        const float ms32 = __builtin_fmaf(__builtin_fmaf(-nf32, q0, Sf), y, q0);
        if (ko->mean) ko->mean[p] = cf + ms32;               // cnt >= 8 here (the core survives)
        if (ko->count) ko->count[p] = cnt;
        if (ko->moments) store_moments(ko->moments, ko->moments64, ko->P, p, cnt, (double)cf, (double)Sf, (double)Qf);
    }
    if constexpr (PLUS) {
        if (wave_any(good)) {
            LateParams *const ko = late_params();
            if (ko->median) {
                constexpr int LO1 = (NP - T - 1) >> 1, LO2 = (NP - T) >> 1;
                const float m1 = pick_rel<LO1, T + 1, NP>(v, ((a + b - 1) >> 1) - LO1);
                const float m2 = pick_rel<LO2, T + 1, NP>(v, ((a + b) >> 1) - LO2);
                if (good) ko->median[p] = (float)(((double)m1 + (double)m2) / 2.0);
            }
            if (ko->std) {
                const float nf32 = (float)(b - a);
                const float y = __builtin_amdgcn_rcpf(nf32);
                const float q0 = Sf * y;
                const double mean64 = (double)cf + (double)__builtin_fmaf(__builtin_fmaf(-nf32, q0, Sf), y, q0);
                // (a wave-uniform test per slot keeps the pass as NP short blocks: as one straight-line block the register
                // allocator keeps every converted value alive)
                int nslots = NP;
                asm volatile("" : "+s"(nslots));
                double q1 = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    if (i >= nslots) continue;
                    const double dd = widen(v[i]) - mean64;
                    const bool in = (i >= T && i < NP - T) || ((i >= a) && (i < b));     // the core always survives
                    const double d = in ? dd : 0.0;
                    q1 = fma(d, d, q1);
                }
                if (good) ko->std[p] = (float)sqrt(q1 > 0.0 ? q1 / (double)(b - a) : 0.0);
            }
        }
    }
}

// FULL = false (since the end of round 4): a padded stack (N between two slot counts) on the same kernel - the padding slots
// are not loaded, the pads are split (-inf below, +inf above the real values, fast32_possible_padded) and clip_fast32's padded
// form starts with them trimmed.
// PADS > 0 (slot counts up to 64, where a padded stack has 1 .. 3 pads): the pad count is a compile-time value - the loads, the
// sentinels and the clip's starting cursors are static, nothing walks over the pads at run time (clip_fast32, PLO / PHI).
template <int NP, typename RawT, bool CALIB, bool FULL = true, int PADS = 0, bool PLUS = false>
__global__ __launch_bounds__(256, NP <= 64 ? APGPU_FAST_MIN_BLOCKS : ((FULL || PADS > 0 || NP <= 96) ? 3 : 2)) void stack_fast_kernel(const StackParams prm)
{
    static_assert(PADS == 0 || (!FULL && PADS <= NP - prev_slots(NP)), "static pads: a padded stack");
    constexpr int MINN = fast_kernel_minn(NP, FULL);
    constexpr int PLO = PADS > 0 ? (PADS >> 1) : -1, PHI = PADS > 0 ? PADS - (PADS >> 1) : -1;
    static_assert(fast_kernel_slots(NP), "the fast kernel is the float32 fast path");
    static_assert(CALIB || sizeof(RawT) == 4, "unfused stacks: float32 frames");
    // tails: what the clip can trim per side.  Unfused stacks spend tail entries on their non-finite values (up to T - 1 per
    // column), so theirs are longer (a core of four values is enough for the sums' four chains)
    constexpr int T = !CALIB ? (NP >= 20 ? 8 : 6) : fast_kernel_tail(NP, FULL);
    constexpr bool HALVES = CALIB && NP >= 104;             // the raw column in two halves: v[] + half of raw[] fit two wavefronts per SIMD
    const int64_t base = (int64_t)blockIdx.x * 256;
    const int lane = threadIdx.x;
    const int64_t p = base + lane;
    __shared__ FrameScalars<NP> fs;                         // (never touched: the ratios come by scalar loads; no LDS is allocated)
    const int N = FULL ? NP : (PADS > 0 ? NP - PADS : prm.N);
    const int plo = FULL ? 0 : (NP - N) >> 1, phi = FULL ? 0 : NP - N - plo;
    const bool alert = fast_kernel_alert();
    if (fast_block_bails<256>(alert, prm.redo)) {           // the segment's pixels mostly fail: the redo pass takes the block whole
        give_blocks_up<256>();
        return;
    }
    bool good = false;
    if (base + 256 <= prm.P) {                              // the last, partial tile goes to the redo list whole
        float v[NP];
        bool dodiv = false;
        int nonfin = 0;                                     // (CALIB = false) the lane's non-finite values
        if constexpr (HALVES) {
            constexpr int HN = NP / 2, NS = PADS > 0 ? NP - PADS : 0;
            const float b = prm.bias[p];
            const float D = prm.still_biased ? prm.dark[p] - b : prm.dark[p];      // ApCalibrate.py:440-445
            float nf = 1.f;
            if (prm.nflat) {
                nf = prm.nflat[p];
                dodiv = (nf != 0.f);                        // ApCalibrate.py:462 (NaN != 0 is True)
            }
            good = !(prm.pixmask && prm.pixmask[p]);
            __builtin_amdgcn_sched_barrier(0);
            RawT half[HN];
            load_raw<NP, RawT, FULL, 0, HN, MINN, NS>(prm, base, lane, half);
            good = calibrate_fast<NP, RawT, false, 0, HN, false, MINN, false, true>(fs, half, b, D, nf, dodiv, v, N, plo, prm.exp_ratio) && good;
            load_raw<NP, RawT, FULL, HN, HN, MINN, NS>(prm, base, lane, half);
            good = calibrate_fast<NP, RawT, false, HN, HN, false, MINN, false, true>(fs, half, b, D, nf, dodiv, v, N, plo, prm.exp_ratio) && good;
        }
        EarlyLoads<NP, RawT> L;
        if constexpr (!HALVES) {
            issue_early_loads<NP, RawT, CALIB, FULL, MINN, (PADS > 0 ? NP - PADS : 0)>(prm, base, lane, L);
            good = !L.skip;
        }
        if constexpr (HALVES) {
        } else if constexpr (CALIB) {
            const float b = L.b;
            const float D = prm.still_biased ? L.d - b : L.d;             // ApCalibrate.py:440-445
            float nf = 1.f;
            if (prm.nflat) {
                nf = L.nf;
                dodiv = (nf != 0.f);                        // ApCalibrate.py:462 (NaN != 0 is True)
            }
            // (measured again on this kernel: a second calibration body for "one exposure ratio", e * D formed once per pixel,
            // -32 instructions - the two bodies' v[] meet in one register set: 128 VGPRs + 36 spilled; not built)
            good = calibrate_fast<NP, RawT, false, 0, NP, false, MINN, false, true>(fs, L.raw, b, D, nf, dodiv, v, N, plo, prm.exp_ratio) && good;
        } else {
            // non-finite values (astropy masks them) become +inf sentinels; padding slots: the first plo -inf, the others +inf
#pragma unroll
            for (int f = 0; f < NP; f++) {
                if (FULL || f < MINN || f < N) {
                    const float x = L.raw[f];
                    v[f] = fabsf(x) < __builtin_inff() ? x : __builtin_inff();
                } else {
                    v[f] = (f < N + plo) ? -__builtin_inff() : __builtin_inff();
                }
            }
        }
#ifdef APGPU_VARIANT_STRIP
        // measurement only (tools/variant_lib.sh): THIS kernel without its clip - 2: loads + calibration, the column summed; 1: the
        // pruned sort and the range test as well, then the column summed - the floor rows of DESIGN 4.1 (round 5: measured on the
        // kernel that ships, same control flow up to the cut, same residency: the LDS block below holds it at four workgroups per CU)
        if (wave_any(good)) {
            if (APGPU_VARIANT_STRIP == 1) {
                sort_column<NP, T>(v);
                if constexpr (CALIB) good = good && range_ok_sorted<NP, MINN>(v, dodiv, N, plo);
            }
            float acc[4] = {good ? 0.f : 1.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NP; i++) acc[i & 3] += v[i];
            if (prm.mean) prm.mean[p] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        __shared__ char geometry[40 * 1024];
        if (prm.N < 0) geometry[lane] = 1;
        return;
#endif
        if (wave_any(good)) {
            sort_column<NP, T>(v);
            if constexpr (CALIB) good = good && range_ok_sorted<NP, MINN>(v, dodiv, N, plo);
            else {
                // the sentinels sit at the top: count them in the upper tail (pads included); a full tail = too many
#pragma unroll
                for (int k = 1; k <= T; k++) nonfin += (v[NP - k] == __builtin_inff()) ? 1 : 0;
                good = good && nonfin < T;
            }
            if (wave_any(good)) finish_fast_column<NP, T, CALIB, PLO, PHI, PLUS>(v, good, p, plo, CALIB ? phi : nonfin);
        }
    }
    // (finished blocks are counted BEFORE their pixels are listed: whoever reads the line in between underestimates the rate)
    fast_blocks_done<256>(alert, prm.redo);
    redo_push(!good && p < late_params()->P, p, alert);
}

// np.nanmedian(axis=0): NaNs dropped, +/-inf are ordinary values.
template <int NP, typename RawT, bool CALIB, bool FULL>
__global__ __launch_bounds__(256) void stack_median_kernel(const StackParams prm)
{
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int lane = threadIdx.x;
    const int64_t p = base + lane;
    __shared__ FrameScalars<NP> fs;
    if (needs_staging<CALIB, FULL, NP>(prm)) stage_frame_scalars<NP>(prm, fs);
    if (p >= prm.P) return;
    float v[NP];
    const int n = load_sorted_column<NP, RawT, CALIB, false, FULL>(prm, fs, base, lane, v);
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
            // (rare path: the plain padding scheme, MINN = NP - the per-slot tests of the other one would set this kernel's
            // register count, and with it the occupancy of the packed path above)
            const int n = load_sorted_column<NP, uint16_t, CALIB, false, FULL, NP>(q, fs, base, lane, v);
            const float m1 = pick_at<NP>(v, (n - 1) >> 1);
            const float m2 = pick_at<NP>(v, n >> 1);
            const double med = ((double)m1 + (double)m2) / 2.0;
            if (prm.median) prm.median[p] = n > 0 ? (float)med : __builtin_nanf("");
            if (prm.count) prm.count[p] = n;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// uint16 clipped stacks, two pixels per lane (the pair scheme).  With one exposure ratio for all frames, no pedestal and a
// positive flat, calibration is one non-decreasing function per pixel, so sorting the RAW uint16 column sorts the calibrated
// column: the raw columns of two neighbouring pixels are sorted together with packed 16-bit compare-exchanges (v_pk_min_u16 /
// v_pk_max_u16 on both pixels at once - half the sort cost per pixel, and one 4-byte load per lane per frame), and each pixel's
// sorted raw column is then calibrated and clipped WITHOUT a second sort.
// This is the FAST kernel of that scheme (round 4; since round 5 the only one, for every full slot count up to 128: the complete
// pair kernel of rounds 2-4, stack_sigclip_u16_pairs_kernel, is gone - whatever this kernel cannot take or finish runs one pixel
// per lane on the complete lean kernel): what stack_fast_kernel is to stack_sigclip_kernel.  Only the common
// path - packed sort of two raw columns (pruned to what the clip reads), one column parked in
// LDS, packed calibration with one scalar-load exposure ratio, range guards off the ends of the sorted column, clip_fast32 per
// lane, outputs; no staging pass, no barrier, no exact calibration, no second sort, no float64 clip: four wavefronts per SIMD
// without spills where the complete kernel holds three with 2-8 spilled registers.  A workgroup whose frames do not share one
// exposure ratio (wave-uniform scalar test), the image's last partial workgroup, and every pixel that cannot finish here
// (non-finite or non-positive masters, a guard, a masked pixel, an unsure comparison, a fifth value to trim) go to the redo list
// - stack_redo_kernel<NP, uint16_t, CALIB, FULL>, the one-pixel-per-lane path, redoes them exactly.
// -------------------------------------------------------------------------------------------------
template <int NP, bool CALIB, bool FULL, int T, int MINN, int PLO = -1, int PHI = -1>
__device__ __forceinline__ bool fast_raw_column(const FrameScalars<NP> &fs, const uint32_t (&cur)[NP / 2], float b, float D, float nf,
                                                bool dv, bool masked, const float *eg, int64_t p, int N, int plo, int phi)
{
    float v[NP];
    bool good = !masked;
    {
        float rawf[NP];
#pragma unroll
        for (int k = 0; k < NP / 2; k++) {
            rawf[2 * k] = (float)(cur[k] & 0xffffu);
            rawf[2 * k + 1] = (float)(cur[k] >> 16);
        }
        if constexpr (CALIB) {
            // non-decreasing map (finite masters, a positive or unused flat): the calibrated column is sorted like the raw one
            const bool increasing = (fabsf(b) < __builtin_inff()) && (fabsf(D) < __builtin_inff()) && (!dv || (nf > 0.f && nf < __builtin_inff()));
            good = calibrate_fast<NP, float, false, 0, NP, false, NP, true, true>(fs, rawf, b, D, nf, dv, v, NP, 0, eg) && increasing && good;
        } else {
#pragma unroll
            for (int f = 0; f < NP; f++) v[f] = rawf[f];
        }
    }
    if constexpr (!FULL) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (i < NP - MINN && i < plo) v[i] = -__builtin_inff();                  // (wave-uniform tests, the end slots only)
            if (i >= MINN && i >= NP - phi) v[i] = __builtin_inff();
        }
    }
    if constexpr (CALIB) good = good && range_ok_sorted<NP, (MINN < NP ? MINN : 0)>(v, dv, N, plo);
    if (wave_any(good)) finish_fast_column<NP, T, true, PLO, PHI>(v, good, p, plo, phi);
    return good;
}

// padded stacks on the pair kernel: split pads, tails of 6 / 8 (fast32_possible_padded), and 104 slots as well (97 .. 103 frames)
constexpr bool pairs_padded_slots(int np) { return fast32_possible_padded(np, padded_minn(np, false)) || np == 104; }

template <int NP, bool CALIB, bool FULL, int PADS = 0>
__global__ __launch_bounds__(256, NP <= 80 ? APGPU_FAST_MIN_BLOCKS : (NP <= 104 ? 3 : 2)) void stack_fast_u16_pairs_kernel(const StackParams prm)
{
    static_assert(PADS == 0 || (!FULL && PADS <= NP - padded_minn(NP, false)), "static pads: a padded stack");
    constexpr int MINN = padded_minn(NP, FULL);
    constexpr int PLO = PADS > 0 ? (PADS >> 1) : -1, PHI = PADS > 0 ? PADS - (PADS >> 1) : -1;
    static_assert(FULL ? fast_kernel_slots(NP) : pairs_padded_slots(NP), "the fast kernel is the float32 fast path");
    constexpr int T = FULL ? kFastTail : fast_tail_padded(NP);
    constexpr int HP = NP / 2;
    __shared__ uint32_t parked[HP][256];
    __shared__ FrameScalars<NP> fs;                         // (never touched: one exposure ratio, read by a scalar load)
    const int lane = threadIdx.x;
    const int N = FULL ? NP : (PADS > 0 ? NP - PADS : prm.N);
    const int plo = FULL ? 0 : (NP - N) >> 1, phi = FULL ? 0 : NP - N - plo;
    const int64_t p2 = ((int64_t)blockIdx.x * 256 + lane) * 2;         // this lane's pixel pair (P is even here)
    bool good0 = false, good1 = false;
    const bool tile_ok = ((int64_t)blockIdx.x + 1) * 512 <= prm.P;
    // Frames that do not share one exposure ratio (a wave-uniform scalar test) cannot take the pair scheme at all, and a segment
    // whose pixels mostly fail should not try: either way the workgroup gives its two tiles up and the redo pass reduces them
    // whole with the complete one-pixel-per-lane body, float32 fast path included (round 4 listed every pixel of such a stack
    // and redid it through the gather with the float64 clip only - the advisor's "performance cliff").
    const bool alert = fast_kernel_alert();
    bool give_up = fast_block_bails<512>(alert, prm.redo);
    if constexpr (CALIB) give_up = give_up || !ratios_uniform<NP>(prm.exp_ratio, N);
    if (give_up) {
        give_blocks_up<512>();
        return;
    }
    if (tile_ok) {
        float bb[2] = {0.f, 0.f}, dd[2] = {0.f, 0.f}, nn[2] = {1.f, 1.f};
        bool dodiv[2] = {false, false};
        bool skip[2] = {false, false};
        if constexpr (CALIB) {                              // masters first: the calibration's first instruction needs them
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
        if (prm.pixmask) {
            skip[0] = prm.pixmask[p2] != 0;
            skip[1] = prm.pixmask[p2 + 1] != 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t w[NP];
        {
            const uint32_t *fp = reinterpret_cast<const uint32_t *>(static_cast<const uint16_t *>(prm.frames) + (int64_t)blockIdx.x * 512);
            const int64_t step = prm.stride / 2;
            int nframes = N;
            if constexpr (!FULL && PADS == 0) asm volatile("" : "+s"(nframes));  // see load_raw
#pragma unroll
            for (int f = 0; f < NP; f++) {
                if (FULL || f < MINN || f < nframes) w[f] = fp[lane];
                else w[f] = (f < nframes + plo) ? 0u : 0xffffffffu;    // split pads: the first plo below, the others above the data
                if (FULL || f + 1 < MINN || f + 1 < nframes) fp += step;
                if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pruned_net_pk16<NP, T>(w);
        // re-pack: the first pixel's column stays in registers, the second one's is parked in LDS meanwhile
        uint32_t cur[HP];
#pragma unroll
        for (int k = 0; k < HP; k++) {
            cur[k] = (w[2 * k] & 0xffffu) | (w[2 * k + 1] << 16);
            parked[k][lane] = (w[2 * k] >> 16) | (w[2 * k + 1] & 0xffff0000u);
        }
        good0 = fast_raw_column<NP, CALIB, FULL, T, MINN, PLO, PHI>(fs, cur, bb[0], dd[0], nn[0], dodiv[0], skip[0], prm.exp_ratio, p2, N, plo, phi);
        int slot = lane;
        asm volatile("" : "+v"(slot) : : "memory");         // opaque index: no store-to-load forwarding in registers
#pragma unroll
        for (int k = 0; k < HP; k++) cur[k] = parked[k][slot];
        good1 = fast_raw_column<NP, CALIB, FULL, T, MINN, PLO, PHI>(fs, cur, bb[1], dd[1], nn[1], dodiv[1], skip[1], prm.exp_ratio, p2 + 1, N, plo, phi);
    }
    fast_blocks_done<512>(alert, prm.redo);                  // (before the pushes: see stack_fast_kernel)
    const int64_t P = late_params()->P;
    redo_push(!good0 && p2 < P, p2, alert);
    redo_push(!good1 && p2 + 1 < P, p2 + 1, alert);
}

// Whether a call takes stack_fast_kernel + stack_redo_kernel (host side; the same conditions as fast32_wanted, plus: lean
// outputs, no pedestals, pixel indices that fit the list's int32 entries).
inline bool fast_kernel_eligible(const StackParams &prm, bool median_only, bool rich, bool plus)
{
#ifdef APGPU_VARIANT_NO_FAST_KERNEL
    return false;
#endif
    if (median_only || rich || plus || prm.single_kernel) return false;
    if (prm.fast32 == 0 || prm.center != APGPU_CENTER_MEDIAN || prm.dev != APGPU_DEV_STD) return false;
    if (!(prm.moments == nullptr || prm.moments64 == 0 || prm.fast32 == 2)) return false;
    if (prm.pedestal) return false;
    return prm.P < 0x7fffffffLL;
}

// fast(prm) launches the fast kernel, redo(prm, workgroups) the redo pass (the complete kernel with prm.redo set).  The
// workspace is the caller's (apgpu_stack_args.workspace, checked by stack_dispatch: large enough, its prefix zero) - then the
// call is exactly these two dispatches; without one the list is a stream-ordered temporary (allocate, clear the prefix,
// free: three more runtime calls, and whatever the pool does with 4 bytes per pixel - the library no longer touches the
// pool's settings).  Returns kNoRedoList (> 0, nothing launched) when neither can be had: the caller goes on to the
// complete kernels, which need no list.
constexpr int kNoRedoList = 1;
bool mad_fast_eligible(const StackParams &prm, bool calib);                                  // stack_mad.hip
int launch_mad_fast(const StackParams &prm, bool u16, hipStream_t st);

// Workgroups of 256 threads of `Kernel` that fit a CU (2 or 3 for the complete kernels, by their register count): the redo pass
// launches exactly one residency round of them.  Asked once per kernel.
template <auto Kernel>
int blocks_per_cu()
{
    static int cached = 0;
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, Kernel, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        cached = n > 4 ? 4 : n;
    }
    return cached;
}

template <typename FastLaunch, typename RedoLaunch>
int launch_with_redo(const StackParams &prm0, hipStream_t st, int redo_blocks_per_cu, FastLaunch fast, RedoLaunch redo_launch)
{
    const int64_t ntiles = (prm0.P + 255) / 256;
    int32_t *ws = prm0.redo;
    const bool own = ws == nullptr;
    if (own) {
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&ws), (size_t)ws_total_words(prm0.P) * sizeof(int32_t), st);
        if (e != hipSuccess) {                              // no room for the list (4 bytes per pixel): the complete kernel needs none
            (void)hipGetLastError();
            return kNoRedoList;
        }
        e = hipMemsetAsync(ws, 0, (size_t)ws_list_off(prm0.P) * sizeof(int32_t), st);
        if (e != hipSuccess) {
            (void)hipFreeAsync(ws, st);
            return fail(APGPU_ELAUNCH, "stack (fast): memset: %s", hipGetErrorString(e));
        }
    }
    StackParams prm = prm0;
    prm.redo = ws;
    fast(prm);
    int rc = check_launch("stack kernel (fast)");
    if (rc == APGPU_OK) {
        // One residency round of the complete kernel: as many workgroups per segment as fit a CU (256 CUs = 256 segments; one
        // per segment for small images), so a pass with every block flagged walks them at full width, while a pass with next to
        // nothing to do - the usual one - is over as soon as its one busy wavefront per segment is (round 4 launched up to
        // 4096 workgroups: several rounds that only look at a counter and leave)
        const int64_t per_seg = ntiles >= 4 * kRedoSegs ? redo_blocks_per_cu : 1;
        redo_launch(prm, (unsigned)(kRedoSegs * per_seg));
        rc = check_launch("stack kernel (redo pass)");
    }
    if (own) {
        const hipError_t ef = hipFreeAsync(ws, st);
        if (rc == APGPU_OK && ef != hipSuccess) return fail(APGPU_ELAUNCH, "stack (fast): free: %s", hipGetErrorString(ef));
    }
    return rc;
}

template <int NP, typename RawT, bool CALIB, bool FULL, int D = 0, bool PLUS = false>
int launch_fast(const StackParams &prm0, dim3 grid, hipStream_t st)
{
    return launch_with_redo(
        prm0, st, blocks_per_cu<stack_sigclip_kernel<NP, RawT, CALIB, false, FULL, PLUS>>(),
        [&](const StackParams &q) { hipLaunchKernelGGL((stack_fast_kernel<NP, RawT, CALIB, FULL, D, PLUS>), grid, dim3(256), 0, st, q); },
        [&](const StackParams &q, unsigned wgs) { hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, FULL, PLUS>), dim3(wgs), dim3(256), 0, st, q); });
}

template <int NP, bool CALIB, bool FULL, int D = 0>
int launch_fast_u16_pairs(const StackParams &prm0, dim3 grid, hipStream_t st)
{
    return launch_with_redo(
        prm0, st, blocks_per_cu<stack_sigclip_kernel<NP, uint16_t, CALIB, false, FULL, false>>(),
        [&](const StackParams &q) { hipLaunchKernelGGL((stack_fast_u16_pairs_kernel<NP, CALIB, FULL, D>), grid, dim3(256), 0, st, q); },
        [&](const StackParams &q, unsigned wgs) { hipLaunchKernelGGL((stack_sigclip_kernel<NP, uint16_t, CALIB, false, FULL, false>), dim3(wgs), dim3(256), 0, st, q); });
}

// `describe` != nullptr: write the name of the kernel variant this call would launch (as rocprofv3 prints it, without
// the namespace) into describe[0..255] and launch nothing - the bench line and the profiles name the dominant kernel
// from the dispatch itself instead of a literal.
template <int NP, typename RawT, bool CALIB>
int launch_one(const StackParams &prm, bool median_only, hipStream_t st, char *describe)
{
    const char *rawname = sizeof(RawT) == 2 ? "unsigned short" : "float";
    const char *tf[2] = {"false", "true"};
    if constexpr (sizeof(RawT) == 2) {
        // uint16 median: pixel pairs per lane need 4-byte aligned frame rows and 8-byte aligned planes
        const bool pairs = median_only && (prm.P % 2 == 0) && (prm.stride % 2 == 0) &&
                           ((reinterpret_cast<uintptr_t>(prm.frames) & 3) == 0) &&
                           ((reinterpret_cast<uintptr_t>(prm.bias) | reinterpret_cast<uintptr_t>(prm.dark) |
                             reinterpret_cast<uintptr_t>(prm.nflat) | reinterpret_cast<uintptr_t>(prm.median) |
                             reinterpret_cast<uintptr_t>(prm.count)) & 7) == 0;
        const bool rich_out = !median_only && (prm.median || prm.std || prm.mean64 || prm.std64 || prm.dev == APGPU_DEV_MAD_STD);
        const bool pairs_clip = !median_only && !rich_out && (prm.P % 2 == 0) && (prm.stride % 2 == 0) &&
                                ((reinterpret_cast<uintptr_t>(prm.frames) & 3) == 0) &&
                                ((reinterpret_cast<uintptr_t>(prm.bias) | reinterpret_cast<uintptr_t>(prm.dark) |
                                  reinterpret_cast<uintptr_t>(prm.nflat)) & 7) == 0;
        // FULL stacks of every slot count, and padded ones up to 104 slots (one instantiation per pad count), take the fast pair
        // kernel (round 5: 72 .. 128 slots too - 101 .. 174 VGPRs, 36 .. 64 KB of LDS, no spills: 72 frames 1.08 -> 0.85 ms, 128:
        // 1.91 -> 1.68); a PADDED stack of 112 .. 128 slots runs one pixel per lane on stack_fast_kernel (static pads, half-column
        // loads)
        const bool wide_fast = CALIB && NP > 104 && fast_kernel_slots(NP) && prm.N != NP && fast_kernel_eligible(prm, median_only, rich_out, false);
        if (!wide_fast && pairs_clip) {
            const int64_t grid = (prm.P + 511) / 512;
            if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
            // the fast kernel of the pair scheme + redo pass (stack_fast_u16_pairs_kernel).  A stack that does not take it (exact =
            // True, mean centre, rich outputs, no list to be had) runs the one-pixel-per-lane kernels below.
            constexpr bool kFastPairsFull = fast_kernel_slots(NP);
            constexpr bool kFastPairsPadded = pairs_padded_slots(NP);
            const bool fastp = (prm.N == NP ? kFastPairsFull : kFastPairsPadded) && fast_kernel_eligible(prm, false, false, false);
            if (describe && fastp) {
                snprintf(describe, 256, "stack_fast_u16_pairs_kernel<%d, %s, %s, %d>", NP, tf[CALIB], tf[prm.N == NP], NP - prm.N);
                return APGPU_OK;
            }
            if (!describe && fastp) {
                int frc = kNoRedoList;
                if constexpr (kFastPairsFull) {
                    if (prm.N == NP) frc = launch_fast_u16_pairs<NP, CALIB, true>(prm, dim3((unsigned)grid), st);
                }
                if constexpr (kFastPairsPadded) {            // (1 .. 3 pads up to 64 slots, 1 .. 7 beyond: one instantiation each)
                    if (prm.N == NP - 1) frc = launch_fast_u16_pairs<NP, CALIB, false, 1>(prm, dim3((unsigned)grid), st);
                    if (prm.N == NP - 2) frc = launch_fast_u16_pairs<NP, CALIB, false, 2>(prm, dim3((unsigned)grid), st);
                    if (prm.N == NP - 3) frc = launch_fast_u16_pairs<NP, CALIB, false, 3>(prm, dim3((unsigned)grid), st);
                    if constexpr (NP > 64) {
                        if (prm.N == NP - 4) frc = launch_fast_u16_pairs<NP, CALIB, false, 4>(prm, dim3((unsigned)grid), st);
                        if (prm.N == NP - 5) frc = launch_fast_u16_pairs<NP, CALIB, false, 5>(prm, dim3((unsigned)grid), st);
                        if (prm.N == NP - 6) frc = launch_fast_u16_pairs<NP, CALIB, false, 6>(prm, dim3((unsigned)grid), st);
                        if (prm.N == NP - 7) frc = launch_fast_u16_pairs<NP, CALIB, false, 7>(prm, dim3((unsigned)grid), st);
                    }
                }
                if (frc != kNoRedoList) return frc;
            }
        }
        if (pairs) {
            const int64_t grid = (prm.P + 511) / 512;
            if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
            if (describe) {
                snprintf(describe, 256, "stack_median_u16_kernel<%d, %s, %s>", NP, tf[CALIB], tf[prm.N == NP]);
                return APGPU_OK;
            }
            if (prm.N == NP) hipLaunchKernelGGL((stack_median_u16_kernel<NP, CALIB, true>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            else hipLaunchKernelGGL((stack_median_u16_kernel<NP, CALIB, false>), dim3((unsigned)grid), dim3(256), 0, st, prm);
            return check_launch("stack median kernel (uint16 pairs)");
        }
    }

#ifdef APGPU_VARIANT_FORCE_RICH                             // tools/variant_lib.sh experiment: LDS-resident column for every output set
    const bool rich = !median_only;
    const bool plus = false;
#else
    // median / std planes come from the lean kernel's longer epilogue (PLUS) for slot counts whose column leaves room in the
    // registers; mad_std, the float64 planes and the largest slot counts take the LDS-resident (rich) kernel
    const bool wants_planes = !median_only && (prm.median || prm.std);
    const bool needs_lds = !median_only && (prm.mean64 || prm.std64 || prm.dev == APGPU_DEV_MAD_STD);
    const bool plus = wants_planes && !needs_lds && NP <= 96;
    const bool rich = needs_lds || (wants_planes && !plus);
#endif
    const bool full = prm.N == NP;
    const int block = rich ? rich_block<NP>() : 256;
    const int64_t grid = (prm.P + block - 1) / block;
    if (grid > 0x7fffffffLL) return fail(APGPU_EUNSUPPORTED, "stack: too many pixels (%lld)", (long long)prm.P);
    // the fast kernel + redo list (see stack_fast_kernel): float32 stacks on the float32 fast path, lean outputs, no pedestals
    // (uint16 frames: only the cases of `wide_fast` above - everything else went to the pair kernels)
    constexpr bool kFastSlots = (sizeof(RawT) == 4 || CALIB) && fast_kernel_slots(NP);
    constexpr int kMaxPads = NP - prev_slots(NP) - 1;                // (the next smaller slot count serves fewer frames)
    // mean + median + std planes (PLUS, up to 96 slots) have their fast kernel too: float32 stacks of any frame count, uint16
    // stacks (fused calibration, one pixel per lane) when full
    constexpr bool kPlusSlots = NP <= 96 && (sizeof(RawT) == 4 || CALIB);
    const bool fastplus = kPlusSlots && plus && (sizeof(RawT) == 4 || full);
    // lean uint16 stacks come here only as `wide_fast` (above): padded beyond 64 slots, or 120 / 128 slots
    const bool u16_lean_ok = !full ? NP > 64 : NP > 112;
    const bool fastk = kFastSlots && (full || kMaxPads > 0) && fast_kernel_eligible(prm, median_only, rich, plus && !fastplus) &&
                       (sizeof(RawT) == 4 || fastplus || u16_lean_ok);
    if (describe) {
        if (fastk) snprintf(describe, 256, "stack_fast_kernel<%d, %s, %s, %s, %d, %s>", NP, rawname, tf[CALIB], tf[full], NP - prm.N, tf[fastplus]);
        else if (median_only) snprintf(describe, 256, "stack_median_kernel<%d, %s, %s, %s>", NP, rawname, tf[CALIB], tf[full]);
        else if (plus) snprintf(describe, 256, "stack_sigclip_kernel<%d, %s, %s, false, %s, true>", NP, rawname, tf[CALIB], tf[full]);
        else snprintf(describe, 256, "stack_sigclip_kernel<%d, %s, %s, %s, %s, false>", NP, rawname, tf[CALIB], tf[rich], tf[full]);
        return APGPU_OK;
    }
    const dim3 g((unsigned)grid), b(block);
    StackParams plain = prm;                                 // the complete kernels' plain launches: no workspace = not a redo pass
    plain.redo = nullptr;
    if constexpr (kFastSlots) {
        int frc = kNoRedoList;
        const int pads = NP - prm.N;
        if (fastk && fastplus) {
            if constexpr (kPlusSlots) {
                if (pads == 0) frc = launch_fast<NP, RawT, CALIB, true, 0, true>(prm, g, st);
                if constexpr (sizeof(RawT) == 4) {
                    if constexpr (kMaxPads >= 1) if (pads == 1) frc = launch_fast<NP, RawT, CALIB, false, 1, true>(prm, g, st);
                    if constexpr (kMaxPads >= 2) if (pads == 2) frc = launch_fast<NP, RawT, CALIB, false, 2, true>(prm, g, st);
                    if constexpr (kMaxPads >= 3) if (pads == 3) frc = launch_fast<NP, RawT, CALIB, false, 3, true>(prm, g, st);
                    if constexpr (kMaxPads >= 4) if (pads == 4) frc = launch_fast<NP, RawT, CALIB, false, 4, true>(prm, g, st);
                    if constexpr (kMaxPads >= 5) if (pads == 5) frc = launch_fast<NP, RawT, CALIB, false, 5, true>(prm, g, st);
                    if constexpr (kMaxPads >= 6) if (pads == 6) frc = launch_fast<NP, RawT, CALIB, false, 6, true>(prm, g, st);
                    if constexpr (kMaxPads >= 7) if (pads == 7) frc = launch_fast<NP, RawT, CALIB, false, 7, true>(prm, g, st);
                }
            }
        } else if (fastk) {
            if constexpr (sizeof(RawT) == 4 || NP > 112) {
                if (pads == 0) frc = launch_fast<NP, RawT, CALIB, true>(prm, g, st);
            }
            if constexpr (sizeof(RawT) == 4 || NP > 64) {
                if constexpr (kMaxPads >= 1) if (pads == 1) frc = launch_fast<NP, RawT, CALIB, false, 1>(prm, g, st);
                if constexpr (kMaxPads >= 2) if (pads == 2) frc = launch_fast<NP, RawT, CALIB, false, 2>(prm, g, st);
                if constexpr (kMaxPads >= 3) if (pads == 3) frc = launch_fast<NP, RawT, CALIB, false, 3>(prm, g, st);
                if constexpr (kMaxPads >= 4) if (pads == 4) frc = launch_fast<NP, RawT, CALIB, false, 4>(prm, g, st);
                if constexpr (kMaxPads >= 5) if (pads == 5) frc = launch_fast<NP, RawT, CALIB, false, 5>(prm, g, st);
                if constexpr (kMaxPads >= 6) if (pads == 6) frc = launch_fast<NP, RawT, CALIB, false, 6>(prm, g, st);
                if constexpr (kMaxPads >= 7) if (pads == 7) frc = launch_fast<NP, RawT, CALIB, false, 7>(prm, g, st);
            }
        }
        if (frc != kNoRedoList) return frc;
    }
    if constexpr (!CALIB) {
        // the ccdproc.combine configuration (one pass of median / mad_std): register-resident fast kernel, then the
        // rich kernel for the 64-pixel blocks it flagged (stack_mad.hip)
        if (rich && mad_fast_eligible(prm, CALIB)) {
            const int frc = launch_mad_fast(prm, sizeof(RawT) == 2, st);
            if (frc == APGPU_OK) {
                StackParams fl = prm;
                fl.flag_mode = 1;
                if (full) hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, true>), g, b, 0, st, fl);
                else hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, false>), g, b, 0, st, fl);
                return check_launch("stack kernel (flagged blocks)");
            }
            if (frc != kNoRedoList) return frc;
        }
    }
    if (median_only) {
        if (full) hipLaunchKernelGGL((stack_median_kernel<NP, RawT, CALIB, true>), g, b, 0, st, plain);
        else hipLaunchKernelGGL((stack_median_kernel<NP, RawT, CALIB, false>), g, b, 0, st, plain);
    } else if (rich) {
        if (full) hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, true>), g, b, 0, st, plain);
        else hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, true, false>), g, b, 0, st, plain);
    } else if (plus) {
        if constexpr (NP <= 96) {
            if (full) hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, true, true>), g, b, 0, st, plain);
            else hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, false, true>), g, b, 0, st, plain);
        }
    } else if (full) {
        hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, true>), g, b, 0, st, plain);
    } else {
        hipLaunchKernelGGL((stack_sigclip_kernel<NP, RawT, CALIB, false, false>), g, b, 0, st, plain);
    }
    return check_launch("stack kernel");
}

// launch_one<NP, RawT, CALIB> is explicitly instantiated in the stack_inst_*.hip translation units (one group of
// slot counts each, so that the build parallelises); everybody else only sees these declarations.
#ifndef APGPU_STACK_INSTANTIATE
#define APGPU_DECLARE_LAUNCH(NP)                                                                          \
    extern template int launch_one<NP, float, true>(const StackParams &, bool, hipStream_t, char *);      \
    extern template int launch_one<NP, float, false>(const StackParams &, bool, hipStream_t, char *);     \
    extern template int launch_one<NP, uint16_t, true>(const StackParams &, bool, hipStream_t, char *);   \
    extern template int launch_one<NP, uint16_t, false>(const StackParams &, bool, hipStream_t, char *);
APGPU_DECLARE_LAUNCH(1) APGPU_DECLARE_LAUNCH(4) APGPU_DECLARE_LAUNCH(8) APGPU_DECLARE_LAUNCH(12) APGPU_DECLARE_LAUNCH(16)
APGPU_DECLARE_LAUNCH(20) APGPU_DECLARE_LAUNCH(24) APGPU_DECLARE_LAUNCH(28) APGPU_DECLARE_LAUNCH(32) APGPU_DECLARE_LAUNCH(36)
APGPU_DECLARE_LAUNCH(40) APGPU_DECLARE_LAUNCH(44) APGPU_DECLARE_LAUNCH(48) APGPU_DECLARE_LAUNCH(52) APGPU_DECLARE_LAUNCH(56)
APGPU_DECLARE_LAUNCH(60) APGPU_DECLARE_LAUNCH(64) APGPU_DECLARE_LAUNCH(72) APGPU_DECLARE_LAUNCH(80) APGPU_DECLARE_LAUNCH(88)
APGPU_DECLARE_LAUNCH(96) APGPU_DECLARE_LAUNCH(104) APGPU_DECLARE_LAUNCH(112) APGPU_DECLARE_LAUNCH(120) APGPU_DECLARE_LAUNCH(128)
#undef APGPU_DECLARE_LAUNCH
#endif

template <typename RawT, bool CALIB>
int launch_np(const StackParams &prm, bool median_only, hipStream_t st, char *describe = nullptr)
{
    const int N = prm.N;
    // slot counts: every multiple of 4 up to 64, every multiple of 8 from there to 128 (round 4: a stack between two slot
    // counts runs the padded kernel, which costs more than the next full one - so the gaps are at most 3 / 7 frames wide)
    if (N <= 1) return launch_one<1, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 4) return launch_one<4, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 8) return launch_one<8, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 12) return launch_one<12, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 16) return launch_one<16, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 20) return launch_one<20, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 24) return launch_one<24, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 28) return launch_one<28, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 32) return launch_one<32, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 36) return launch_one<36, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 40) return launch_one<40, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 44) return launch_one<44, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 48) return launch_one<48, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 52) return launch_one<52, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 56) return launch_one<56, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 60) return launch_one<60, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 64) return launch_one<64, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 72) return launch_one<72, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 80) return launch_one<80, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 88) return launch_one<88, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 96) return launch_one<96, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 104) return launch_one<104, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 112) return launch_one<112, RawT, CALIB>(prm, median_only, st, describe);
    if (N <= 120) return launch_one<120, RawT, CALIB>(prm, median_only, st, describe);
    return launch_one<128, RawT, CALIB>(prm, median_only, st, describe);
}


}  // namespace apgpu_stack
