// resample_core.h - the device code the resample kernels share: tile records and the tile pass, the footprint fills, the
// Lanczos-3 window evaluation (fast and general paths), the bad-pixel list.  Included by resample.hip (resample_affine_kernel:
// one frame-tile per workgroup, output to memory) and resample_stack.hip (round 6: resample_clip_kernel - the N frames of an
// output tile resampled into registers and clipped there, the co-add of scripts/resample_all.sh:330-342 in one launch).
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"

typedef float apgpu_v2f __attribute__((ext_vector_type(2)));
typedef float apgpu_v4f __attribute__((ext_vector_type(4)));
typedef int apgpu_v4i __attribute__((ext_vector_type(4)));
// Raw buffer instructions by their LLVM names (this clang's __builtin_amdgcn_raw_buffer_load_b128 / _b64 emit a ONE-dword
// load): resource in four SGPRs, 32-bit byte offset per lane, bounds-checked against the resource's size.
__device__ float apgpu_buffer_load_f32(apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ apgpu_v2f apgpu_buffer_load_v2f32(apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ apgpu_v4f apgpu_buffer_load_v4f32(apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ char apgpu_buffer_load_i8(apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i8");
__device__ void apgpu_buffer_store_f32(float v, apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void apgpu_buffer_store_i8(char v, apgpu_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i8");

namespace {
using namespace apgpu;

constexpr int kTileW = APGPU_RESAMPLE_TILE_W, kTileH = APGPU_RESAMPLE_TILE_H;
constexpr int kGenericFloats = 4096;                 // general path: footprint up to 16 KB at its own pitch
constexpr int kFastPitch = 80;                       // fast path: fixed pitch (320 B: row j of a window = immediate offset)
// TH = output rows per workgroup: 16 (one API tile: the only choice with one transform per tile) or 32 (two vertically
// adjacent tiles of a frame that has ONE transform: half the workgroup launches and tile set-up, 42 instead of 2 x 26
// footprint rows).  The fast path takes footprints of up to TH + 10 rows (rotations up to ~2.5 degrees).
// (Round 5, measured and dropped: ONE copy of the footprint, odd-start windows read with 4-byte aligned pairs - ds_read2_b32 -
// for half the LDS per workgroup and half the fill's stores: 9.0 against 4.0 ms per 16 x 8192^2.)
// (and: a wavefront shaped 16 columns x 4 row groups instead of 64 x 1 - a third of the distinct table rows per weight load, LDS
// pitch 84 and copy B at 16 mod 64 banks to keep the reads conflict-free - no change: 3.94-4.00 against 3.95-3.96 ms.)
template <int TH>
struct FastGeom {
    static constexpr int kRows = TH + 10;
    static constexpr int kTrips = (kRows + 2) / 3;                       // 3 footprint rows per fill trip
    static constexpr int kCopy = kFastPitch * 3 * kTrips;                // floats per copy: the fill writes whole trips
    static constexpr int kOffB = ((kCopy + 1 + 31) / 64) * 64 + 32;      // copy B (shifted by one float) starts 32 banks after copy A, behind a gap
    static constexpr int kLdsFloats = kOffB + kCopy > kGenericFloats ? kOffB + kCopy : kGenericFloats;   // 17.6 KB (TH 16) / 29.1 KB (TH 32)
    static_assert(kOffB % 64 == 32 && kOffB > kCopy && kFastPitch % 2 == 0, "LDS layout");
};
constexpr unsigned kRsrcFlags = 0x00020000;          // raw buffer, 32-bit elements (gfx9 family word 3)
typedef apgpu_v2f v2f;
typedef apgpu_v4f v4f;
typedef apgpu_v4i v4i;

// p and bytes are wave-uniform
__device__ __forceinline__ v4i make_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));     // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = (int)kRsrcFlags;
    return r;
}
#ifdef APGPU_VARIANT_RESAMPLE_READ2
typedef __attribute__((address_space(3))) const v2f *lds_pair_p;            // the compiler pairs them into ds_read2_b64
#else
typedef __attribute__((address_space(3))) const volatile v2f *lds_pair_p;   // volatile: one ds_read_b64 each (256 B/clk)
#endif
#ifndef APGPU_RESAMPLE_ROWS_PER_BATCH
#define APGPU_RESAMPLE_ROWS_PER_BATCH 3
#endif

enum : unsigned { kStaged = 1, kSane = 2, kInterior = 4, kFast = 8, kSaneTop = 16, kSaneBot = 32, kInlineMask = 64 };   // kSane: every 16-row half of the workgroup's tile is defined

// What the tile pass works out per output tile (everything tile-uniform), 64 bytes.
struct alignas(64) TileRec {
    long long F[6];             // the transform in 32.32 fixed point (of the fine grid when oversampling)
    int bx0, by0;               // first input column / row of the footprint
    unsigned dims;              // footprint width | height << 12 | flags << 24 (a staged footprint is at most 682 wide / tall)
    float fs;                   // flux scale
};

struct FrameView {
    const float *src;
    const uint8_t *mask;
    int h_in, w_in;
};

// src[row][col], or NaN if the pixel is outside the frame, masked or not finite
__device__ __forceinline__ float fetch_global(const FrameView &fv, int row, int col)
{
    if (row < 0 || row >= fv.h_in || col < 0 || col >= fv.w_in) return __builtin_nanf("");
    const int64_t q = (int64_t)row * fv.w_in + col;
    const float v = fv.src[q];
    const bool bad = !(fabsf(v) < __builtin_inff()) || (fv.mask && fv.mask[q] != 0);
    return bad ? __builtin_nanf("") : v;
}

// One output pixel from its 6 x 6 window; `row(j)` returns the 6 samples of window row j as three pairs.
// Evaluation order (restated in the oracle): per row the even and the odd taps are two fmaf chains, the rows
// are combined by two fmaf chains over j, and the two halves are added last - which is exactly a sequence of
// packed float32 operations on (even, odd) pairs: 4 instructions per row for 6 taps.
struct Weights {
    v2f wx01, wx23, wx45;       // x taps as (even, odd) pairs
    v2f wy01, wy23, wy45;
};

__device__ __forceinline__ Weights load_weights(const float *__restrict__ lut, int px, int py)
{
    const v2f *wxp = reinterpret_cast<const v2f *>(lut + 6 * px);
    const v2f *wyp = reinterpret_cast<const v2f *>(lut + 6 * py);
    Weights w;
    w.wx01 = wxp[0]; w.wx23 = wxp[1]; w.wx45 = wxp[2];
    w.wy01 = wyp[0]; w.wy23 = wyp[1]; w.wy45 = wyp[2];
    return w;
}

// the same rows through a buffer resource: one 32-bit offset per row, 16 + 8 bytes
__device__ __forceinline__ Weights load_weights(v4i lut, int px, int py)
{
    const int ox = (int)__umul24((unsigned)px, 24u), oy = (int)__umul24((unsigned)py, 24u);
    const v4f a = apgpu_buffer_load_v4f32(lut, ox, 0, 0);
    const v2f b = apgpu_buffer_load_v2f32(lut, ox + 16, 0, 0);
    const v4f c = apgpu_buffer_load_v4f32(lut, oy, 0, 0);
    const v2f d = apgpu_buffer_load_v2f32(lut, oy + 16, 0, 0);
    Weights w;
    w.wx01 = v2f{a.x, a.y};
    w.wx23 = v2f{a.z, a.w};
    w.wx45 = b;
    w.wy01 = v2f{c.x, c.y};
    w.wy23 = v2f{c.z, c.w};
    w.wy45 = d;
    return w;
}

template <typename RowFn>
__device__ __forceinline__ float window_sum(const Weights &w, RowFn row)
{
    const float wy[6] = {w.wy01.x, w.wy01.y, w.wy23.x, w.wy23.y, w.wy45.x, w.wy45.y};
    v2f V = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; j++) {
        v2f s01, s23, s45;
        row(j, s01, s23, s45);
        v2f acc = w.wx01 * s01;
        acc = __builtin_elementwise_fma(w.wx23, s23, acc);
        acc = __builtin_elementwise_fma(w.wx45, s45, acc);
        const v2f wyj = {wy[j], wy[j]};
        V = (j == 0) ? wyj * acc : __builtin_elementwise_fma(wyj, acc, V);
    }
    return V.x + V.y;
}

__device__ __forceinline__ void phases(unsigned long long X, unsigned long long Y, int sh, int &jx, int &jy, int &px, int &py)
{
    // a sane tile keeps the coordinates within +-1e9: the integer part IS the high dword (no 64-bit compares or selects)
    jx = (int)((long long)X >> 32);
    jy = (int)((long long)Y >> 32);
    const unsigned frx = (unsigned)X, fry = (unsigned)Y;
    px = (int)(((frx >> (sh - 1)) + 1u) >> 1);              // = (frx >> sh) + ((frx >> (sh - 1)) & 1)
    py = (int)(((fry >> (sh - 1)) + 1u) >> 1);
}

struct TileCtx {
    long long F[6];
    int bx0, by0, fw, fh;
    bool staged, sane_top, sane_bot;                     // sane_*: the 16-row API tile holding the workgroup's upper / lower rows is defined
    float fs;
};

// ---- FAST tiles ------------------------------------------------------------------------------------------------------
// One sample in two steps, so that the table rows of the NEXT sample are on their way while this one is evaluated:
// prep: phases, the two table rows (buffer loads), the LDS address; eval: 18 aligned LDS reads, 20 packed multiply-adds.
// cxo / cyo: input column / row of the footprint's origin + 2 (the window starts two taps before floor()).
struct FastPrep {
    Weights w;
    int idx;                    // float index of the window's first pair (even)
};

// a * b + c with a, b < 2^24: one instruction (the compiler's own choice for `(s & 1) * constant` was compare + select)
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// Xr, Yr: the fixed-point input coordinates minus the footprint origin + 2 (the window starts two taps before floor()), so
// that their high dwords ARE the window's first column / row inside the footprint; the fractions are those of X, Y.
template <int OFFB, typename LutT>
__device__ __forceinline__ FastPrep prep_fast(unsigned long long Xr, unsigned long long Yr, int sh, LutT lut)
{
    int js, jr, px, py;
    phases(Xr, Yr, sh, js, jr, px, py);
    FastPrep p;
    p.w = load_weights(lut, px, py);
    const unsigned s = (unsigned)js, r = (unsigned)jr;
    // an odd first column reads copy B, where element k holds the footprint's k + 1: the same six taps from the even k = s - 1
    p.idx = (int)mad_u24(s & 1u, (unsigned)(OFFB - 1), mad_u24(r, (unsigned)kFastPitch, s));
    return p;
}

// the value before the flux scale.  The window is read and reduced in two halves of three rows (9 reads each) with a
// scheduling fence between them: 18 instead of 36 sample registers alive, which is what lets 8+ waves share a SIMD.
__device__ __forceinline__ float eval_fast(const FastPrep &p, const float *tile)
{
    lds_pair_p t = (lds_pair_p)(tile + p.idx);
    const Weights &w = p.w;
    const float wy[6] = {w.wy01.x, w.wy01.y, w.wy23.x, w.wy23.y, w.wy45.x, w.wy45.y};
    v2f V = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 6; h += APGPU_RESAMPLE_ROWS_PER_BATCH) {
        v2f smp[APGPU_RESAMPLE_ROWS_PER_BATCH][3];
#pragma unroll
        for (int j = 0; j < APGPU_RESAMPLE_ROWS_PER_BATCH; j++) {
            smp[j][0] = t[(h + j) * (kFastPitch / 2) + 0];
            smp[j][1] = t[(h + j) * (kFastPitch / 2) + 1];
            smp[j][2] = t[(h + j) * (kFastPitch / 2) + 2];
        }
#pragma unroll
        for (int j = 0; j < APGPU_RESAMPLE_ROWS_PER_BATCH; j++) {
            v2f acc = w.wx01 * smp[j][0];
            acc = __builtin_elementwise_fma(w.wx23, smp[j][1], acc);
            acc = __builtin_elementwise_fma(w.wx45, smp[j][2], acc);
            const v2f wyj = {wy[h + j], wy[h + j]};
            V = (h + j == 0) ? wyj * acc : __builtin_elementwise_fma(wyj, acc, V);
        }
        if (h + APGPU_RESAMPLE_ROWS_PER_BATCH < 6) __builtin_amdgcn_sched_barrier(0);
    }
    return V.x + V.y;
}

template <bool OVERSAMPLED, int TH, int UNR, typename LutT>
__device__ __forceinline__ void pixels_fast(const TileCtx &tc, const float *tile, LutT lut, int sh, int os, int x0, int y0, int lx, int ly,
                                            v4i orsrc, v4i wrsrc, bool want_w, int ooff, int ostep)
{
    const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
    const float fs = tc.fs;
    const unsigned long long n = OVERSAMPLED ? (unsigned long long)os : 1ull;
    // 64-bit two's-complement sums: exact, because the true coordinates fit (sane), whatever the partial products do.
    // Tile-uniform part on the scalar unit (the tile's first pixel, minus the footprint origin), the lane's offset inside
    // the tile - at most 63 columns and 3 rows - as two small products.
    const unsigned long long us = (unsigned long long)(long long)x0 * n, vs = (unsigned long long)(long long)y0 * n;
    const unsigned long long Xs = F0 * us + F1 * vs + F2 - ((unsigned long long)(unsigned)(tc.bx0 + 2) << 32);
    const unsigned long long Ys = F3 * us + F4 * vs + F5 - ((unsigned long long)(unsigned)(tc.by0 + 2) << 32);
    const unsigned long long ul = (unsigned long long)(unsigned)lx * n, vl = (unsigned long long)(unsigned)ly * n;
    unsigned long long X = Xs + F0 * ul + F1 * vl;
    unsigned long long Y = Ys + F3 * ul + F4 * vl;
    const unsigned long long dX = F1 * (4ull * n), dY = F4 * (4ull * n);
    if constexpr (!OVERSAMPLED) {
        // one pixel per trip, table rows loaded in the trip that uses them: a variant that fetched the next pixel's rows
        // one trip ahead (12 more registers) measured 3 % slower, two pixels per trip no faster
#pragma unroll UNR
        for (int k = 0; k < TH / 4; k++) {
            const FastPrep cur = prep_fast<FastGeom<TH>::kOffB>(X, Y, sh, lut);
            X += dX;
            Y += dY;
            const float v = eval_fast(cur, tile);
            const float res = (v == v) ? v * fs : __builtin_nanf("");
            apgpu_buffer_store_f32(res, orsrc, ooff, 0, 0);
            if (want_w) apgpu_buffer_store_i8((char)((res == res) ? 1 : 0), wrsrc, ooff >> 2, 0, 0);
            ooff += ostep;
        }
    } else {
        const double inv = 1.0 / (double)(os * os);
#pragma unroll 1
        for (int k = 0; k < TH / 4; k++) {
            double acc = 0.0;                                     // a NaN sub-sample makes the sum, and the pixel, NaN
            unsigned long long Xa = X, Ya = Y;
#pragma unroll 1
            for (int a = 0; a < os; a++) {
                unsigned long long Xb = Xa, Yb = Ya;
#pragma unroll 1
                for (int b = 0; b < os; b++) {
                    const FastPrep cur = prep_fast<FastGeom<TH>::kOffB>(Xb, Yb, sh, lut);
                    const float v = eval_fast(cur, tile);
                    acc += (double)((v == v) ? v * fs : __builtin_nanf(""));
                    Xb += F0;
                    Yb += F3;
                }
                Xa += F1;
                Ya += F4;
            }
            const float res = (float)(acc * inv);
            X += dX;
            Y += dY;
            apgpu_buffer_store_f32(res, orsrc, ooff, 0, 0);
            if (want_w) apgpu_buffer_store_i8((char)((res == res) ? 1 : 0), wrsrc, ooff >> 2, 0, 0);
            ooff += ostep;
        }
    }
}

// ---- FAST tiles, rolling window (round 5) ------------------------------------------------------------------------------
// A lane produces TH / 4 CONSECUTIVE output rows of one column and keeps its 6 x 6 window in registers between them.  For a
// registration-sized transform the next output row's window is the same six columns one input row further down - then five
// of its six rows are already in registers and only ONE row (three ds_read_b64) is read; otherwise (the column or the copy
// changed: every 1 / |sin(rotation)| rows; or the row step was not exactly one) the lane re-reads its whole window.  The test
// is one compare - the window's LDS index went up by exactly the row pitch - and the re-read sits behind a wave vote, so a
// wave pays for it only in the steps in which one of its lanes needs it (0.2 degrees: one step in five).  The window rows
// live in six register triples in a STATIC rotation (step k keeps window row j in triple (j + k) % 6: the loop is unrolled,
// nothing is ever moved - round 3's first attempt at this shuffled registers around the vote and lost more VALU work than
// the LDS reads saved).  LDS reads per pixel: 18 -> 3 + (18 + 15 x re-reads) / (TH / 4), e.g. ~7 at 0.2 degrees.  Same
// arithmetic in the same order as eval_fast: the oracle is untouched.
#ifndef APGPU_RESAMPLE_ROLLING
#define APGPU_RESAMPLE_ROLLING 1
#endif


// The per-pixel table rows (four buffer loads) are fetched kAhead pixels AHEAD of their use: a pixel's wait for its weights was
// a memory round trip behind the previous pixel's output store (one in-order counter for both), eight of them in a row per
// wavefront - with 72 registers in use and the LDS holding the kernel at five wavefronts per SIMD, the 12 registers per pixel in
// flight are free.  The first kAhead pixels are prepared BEFORE the footprint's barrier (their loads overlap the fill's).
#ifndef APGPU_RESAMPLE_AHEAD
#define APGPU_RESAMPLE_AHEAD 1
#endif
#ifndef APGPU_RESAMPLE_ONE_LDS_WAIT
#define APGPU_RESAMPLE_ONE_LDS_WAIT 0
#endif
#ifndef APGPU_RESAMPLE_REREAD_VOTE
#define APGPU_RESAMPLE_REREAD_VOTE 0
#endif
// Round 5, second step - the table rows are what the kernel waits for: with every lane reading row 0 (wrong weights, timing only)
// the same launch takes 2.92 instead of 3.75 ms, with 16 distinct rows the full 3.75, and without the second (8-byte) load of
// each row nothing changes (profiles/r05_c5/ab_resample.txt): a gather is paid per distinct row its lanes address, and the y rows
// of a wavefront's 64 columns are 64 different ones (the y phase moves by sin(rotation) x phases per column), fetched for
// every pixel.  But a lane's CONSECUTIVE output rows differ in Y by F4 = cos(rotation) x scale - an integer plus a few millionths:
// on a "steady" tile (the phase drifts by less than one table row over the lane's R rows; rotations up to ~0.9 degrees at unit
// scale, scale errors up to ~1.4e-4) the y phase is monotonic and takes the value of the lane's first row, then that of its last.
// Both rows are fetched once per tile, during the footprint fill, and a pixel picks one (six selects): 2 y gathers per tile instead
// of 8.  Where the fraction wraps, a third value can appear (.. 1023, 1024 | 0, 1 ..: rows 1024 and 0 are half a phase wide): that
// lane fetches on the spot, behind a wave vote.  3.80 -> 3.45 ms at +-0.2 degrees, and the same 3.5 ms at 0.1 / 0.6 degrees and at
// scale 1.0001.
// Measured beside it and dropped (same file): (a) ONE kept row per lane, re-fetched behind a vote when the phase moves on, with
// its own full wait: 3.41 / 3.28 ms at 0.2 / 0.1 degrees but 3.73 at 0.6 and 3.9 at scale 1.0001 - some lane of the 64 moves
// on in 40 % of the steps at 0.2 degrees and in all of them at scale 1.0001; (b) that re-fetch issued two pixels ahead into
// per-stage copies, nothing waits: 3.50 / 3.83 ms, 92 VGPRs (a gather with a tenth of its lanes active costs what a full one
// costs); (c) the x row through the scalar unit when a wavefront's 64 columns share one x phase (s_buffer_load + six moves):
// 3.50 against 3.40 ms, 3.42 against 3.20 for pure translations - a gather whose lanes all read ONE address is already cheap;
// (d) the x rows of a steady tile staged in LDS - its range of (cyclic) table rows from the tile's four corners, up to 224 rows =
// 5.4 KB copied with coalesced loads during the fill, a pixel reading its row with three ds_read_b64: bit-identical, and 4.4 ms
// (3.9 with room for 96 or 160 rows) against 3.45 - also for tiles that did not stage: the copy's registers and the larger LDS
// block cost every tile more than the gathers cost the staged ones.
#ifndef APGPU_RESAMPLE_KEEP_WY
#define APGPU_RESAMPLE_KEEP_WY 1
#endif

struct RowsY {
    v2f wy01, wy23, wy45;
};

struct RollPrep {
    v2f wx01, wx23, wx45;       // x taps as (even, odd) pairs
    RowsY y;                    // !KEEP: the y rows travel with the x rows (KEEP: not used)
    int py;                     // y phase (table row)
    int idx;                    // float index of the window's first pair (even)
};

__device__ __forceinline__ RowsY load_rows_y(v4i lut, int py)
{
    const int oy = (int)__umul24((unsigned)py, 24u);
    const v4f c = apgpu_buffer_load_v4f32(lut, oy, 0, 0);
    const v2f d = apgpu_buffer_load_v2f32(lut, oy + 16, 0, 0);
    RowsY r;
    r.wy01 = v2f{c.x, c.y};
    r.wy23 = v2f{c.z, c.w};
    r.wy45 = d;
    return r;
}

template <int OFFB, bool KEEP>
__device__ __forceinline__ RollPrep prep_roll(unsigned long long Xr, unsigned long long Yr, int sh, v4i lut)
{
    int js, jr, px, py;
    phases(Xr, Yr, sh, js, jr, px, py);
    RollPrep p;
    {
        const int ox = (int)__umul24((unsigned)px, 24u);
        const v4f a = apgpu_buffer_load_v4f32(lut, ox, 0, 0);
        const v2f b = apgpu_buffer_load_v2f32(lut, ox + 16, 0, 0);
        p.wx01 = v2f{a.x, a.y};
        p.wx23 = v2f{a.z, a.w};
        p.wx45 = b;
    }
    if constexpr (!KEEP) p.y = load_rows_y(lut, py);
    p.py = py;
    const unsigned sx = (unsigned)js, r = (unsigned)jr;
    p.idx = (int)mad_u24(sx & 1u, (unsigned)(OFFB - 1), mad_u24(r, (unsigned)kFastPitch, sx));
    return p;
}

template <int TH, int AHEAD>
struct Rolling {
    static constexpr int R = TH / 4;                         // consecutive rows per lane
    static constexpr int kAhead = AHEAD < R ? AHEAD : R - 1;
    unsigned long long X, Y;                                 // coordinates of the next pixel to prepare
    RollPrep nxt[kAhead > 0 ? kAhead : 1];
    RowsY wyA, wyB;                                          // KEEP: the y rows of the lane's first and last output row, and their phases
    int pyA, pyB;
};

template <int TH, bool KEEP, int AHEAD, int AS, typename LutT>          // AHEAD <= AS: pixels prepared ahead / slots in the state
__device__ __forceinline__ void rolling_begin(Rolling<TH, AS> &ro, const TileCtx &tc, LutT lut, int sh, int x0, int y0, int lx, int ly)
{
    constexpr int R = Rolling<TH, AHEAD>::R;
    const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
    const unsigned long long us = (unsigned long long)(long long)x0, vs = (unsigned long long)(long long)y0;
    const unsigned long long Xs = F0 * us + F1 * vs + F2 - ((unsigned long long)(unsigned)(tc.bx0 + 2) << 32);
    const unsigned long long Ys = F3 * us + F4 * vs + F5 - ((unsigned long long)(unsigned)(tc.by0 + 2) << 32);
    const unsigned long long ul = (unsigned long long)(unsigned)lx, vl = (unsigned long long)(unsigned)(ly * R);
    ro.X = Xs + F0 * ul + F1 * vl;
    ro.Y = Ys + F3 * ul + F4 * vl;
    if constexpr (KEEP) {
        // steady tiles: the y phase moves by less than one table row over the lane's R rows, and monotonically - it takes at most
        // TWO values, those of the first and of the last row.  Both rows are fetched here, during the fill; a pixel picks one.
        int js, jr, px, py;
        phases(ro.X, ro.Y, sh, js, jr, px, py);
        ro.pyA = py;
        phases(ro.X + F1 * (unsigned long long)(R - 1), ro.Y + F4 * (unsigned long long)(R - 1), sh, js, jr, px, py);
        ro.pyB = py;
        ro.wyA = load_rows_y(lut, ro.pyA);
        ro.wyB = load_rows_y(lut, ro.pyB);
    }
#pragma unroll
    for (int k = 0; k < Rolling<TH, AHEAD>::kAhead; k++) {
        ro.nxt[k] = prep_roll<FastGeom<TH>::kOffB, KEEP>(ro.X, ro.Y, sh, lut);
        ro.X += F1;
        ro.Y += F4;
    }
}

template <int TH, bool KEEP, int AHEAD, int AS, typename LutT>
__device__ __forceinline__ void pixels_fast_rolling(Rolling<TH, AS> &ro, const TileCtx &tc, const float *tile, LutT lut, int sh, int lx, int ly,
                                                    v4i orsrc, v4i wrsrc, bool want_w, int w_out)
{
    constexpr int R = Rolling<TH, AHEAD>::R, A = Rolling<TH, AHEAD>::kAhead;
    const unsigned long long F1 = tc.F[1], F4 = tc.F[4];
    const float fs = tc.fs;
    int ooff = (ly * R * w_out + lx) * 4;
    const int ostep = 4 * w_out;
    v2f win[6][3];
    int prev = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        RollPrep cur;
        if constexpr (A > 0) {
            cur = ro.nxt[k % A];
            if (k + A < R) {
                ro.nxt[k % A] = prep_roll<FastGeom<TH>::kOffB, KEEP>(ro.X, ro.Y, sh, lut);
                ro.X += F1;
                ro.Y += F4;
            }
        } else {
            cur = prep_roll<FastGeom<TH>::kOffB, KEEP>(ro.X, ro.Y, sh, lut);
            ro.X += F1;
            ro.Y += F4;
        }
        RowsY wyr;
        if constexpr (KEEP) {
            const bool isA = cur.py == ro.pyA;
            wyr.wy01 = isA ? ro.wyA.wy01 : ro.wyB.wy01;
            wyr.wy23 = isA ? ro.wyA.wy23 : ro.wyB.wy23;
            wyr.wy45 = isA ? ro.wyA.wy45 : ro.wyB.wy45;
            // a third value is possible only where the fraction wraps (.. 1023, 1024 | 0, 1 ..: rows 1024 and 0 are half a phase
            // wide each): such a lane fetches its rows on the spot, with its own full wait inside the rare block so that the
            // compiler's wait counters at the join stay those of the prefetched x rows
            const bool other = !isA && cur.py != ro.pyB;
            if (__builtin_amdgcn_ballot_w64(other) != 0) {
                if (other) wyr = load_rows_y(lut, cur.py);
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
        } else {
            wyr = cur.y;
        }
        lds_pair_p t = (lds_pair_p)(tile + cur.idx);
        const bool reread = (k == 0) || (cur.idx != prev + kFastPitch);
        prev = cur.idx;
#if APGPU_RESAMPLE_REREAD_VOTE
        if (k == 0 || __builtin_amdgcn_ballot_w64(reread) != 0)
#endif
        {
            if (reread) {                                         // (divergent branch: skipped by the wave when no lane takes it)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    win[(j + k) % 6][0] = t[j * (kFastPitch / 2) + 0];
                    win[(j + k) % 6][1] = t[j * (kFastPitch / 2) + 1];
                    win[(j + k) % 6][2] = t[j * (kFastPitch / 2) + 2];
                }
            }
        }
        win[(5 + k) % 6][0] = t[5 * (kFastPitch / 2) + 0];
        win[(5 + k) % 6][1] = t[5 * (kFastPitch / 2) + 1];
        win[(5 + k) % 6][2] = t[5 * (kFastPitch / 2) + 2];
#if APGPU_RESAMPLE_ONE_LDS_WAIT
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): one wait for the window instead of one per read
#endif
        const float wy[6] = {wyr.wy01.x, wyr.wy01.y, wyr.wy23.x, wyr.wy23.y, wyr.wy45.x, wyr.wy45.y};
        v2f V = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const v2f(&row)[3] = win[(j + k) % 6];
            v2f acc = cur.wx01 * row[0];
            acc = __builtin_elementwise_fma(cur.wx23, row[1], acc);
            acc = __builtin_elementwise_fma(cur.wx45, row[2], acc);
            const v2f wyj = {wy[j], wy[j]};
            V = (j == 0) ? wyj * acc : __builtin_elementwise_fma(wyj, acc, V);
        }
        const float v = V.x + V.y;
        const float res = (v == v) ? v * fs : __builtin_nanf("");
        apgpu_buffer_store_f32(res, orsrc, ooff, 0, 0);
        if (want_w) apgpu_buffer_store_i8((char)((res == res) ? 1 : 0), wrsrc, ooff >> 2, 0, 0);
        ooff += ostep;
    }
}

// ---- general tiles ---------------------------------------------------------------------------------------------------
// INTERIOR: every window of the tile lies inside the frame (decided once per tile), so the per-sample frame tests and the
// selects they feed disappear.  y_ok: the pixel's row is inside the output.
template <bool INTERIOR>
__device__ __forceinline__ float sample_general(const TileCtx &tc, const FrameView &fv, const float *tile, const float *__restrict__ lut,
                                                unsigned long long X, unsigned long long Y, int sh, bool sane)
{
    int jx, jy, px, py;
    phases(X, Y, sh, jx, jy, px, py);
    // 2 <= ix <= w_in - 4 (the 6 x 6 window inside the frame)
    const bool inside = INTERIOR || (sane && (unsigned)(jx - 2) < (unsigned)(fv.w_in - 5) && (unsigned)(jy - 2) < (unsigned)(fv.h_in - 5));
    const int ix = inside ? jx : 0, iy = inside ? jy : 0;
    const Weights wts = load_weights(lut, inside ? px : 0, inside ? py : 0);
    float v;
    if (INTERIOR || tc.staged) {
        // pixels outside the frame read (and discard) the tile origin
        const int off = inside ? (iy - 2 - tc.by0) * tc.fw + (ix - 2 - tc.bx0) : 0;
        const int stride = inside ? tc.fw : 0;
        const float *t = tile + off;
        // all 18 ds_read2_b32 of the window are issued before the first product
        v2f smp[6][3];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const float *r = t + j * stride;
            smp[j][0] = v2f{r[0], r[1]};
            smp[j][1] = v2f{r[2], r[3]};
            smp[j][2] = v2f{r[4], r[5]};
        }
        v = window_sum(wts, [&](int j, v2f &s01, v2f &s23, v2f &s45) {
            s01 = smp[j][0];
            s23 = smp[j][1];
            s45 = smp[j][2];
        });
    } else {
        v = window_sum(wts, [&](int j, v2f &s01, v2f &s23, v2f &s45) {
            s01 = v2f{fetch_global(fv, iy - 2 + j, ix - 2), fetch_global(fv, iy - 2 + j, ix - 1)};
            s23 = v2f{fetch_global(fv, iy - 2 + j, ix), fetch_global(fv, iy - 2 + j, ix + 1)};
            s45 = v2f{fetch_global(fv, iy - 2 + j, ix + 2), fetch_global(fv, iy - 2 + j, ix + 3)};
        });
    }
    // invalid taps arrive as NaN and poison v; the weight plane is "out is not NaN" in the oracle too
    return (inside && v == v) ? v * tc.fs : __builtin_nanf("");
}

// The pixels of one lane: column x, rows yb0, yb0 + 4, ...
template <bool INTERIOR, bool OVERSAMPLED, int TH>
__device__ __forceinline__ void pixels_general(const TileCtx &tc, const FrameView &fv, const float *tile, const float *__restrict__ lut,
                                               int sh, int os, int x, int yb0, int h_out, int64_t row_stride, float *op, uint8_t *wp)
{
    const unsigned long long F0 = tc.F[0], F1 = tc.F[1], F2 = tc.F[2], F3 = tc.F[3], F4 = tc.F[4], F5 = tc.F[5];
    const unsigned long long n = OVERSAMPLED ? (unsigned long long)os : 1ull;
    const unsigned long long u0 = (unsigned long long)(long long)x * n, v0 = (unsigned long long)(long long)yb0 * n;
    unsigned long long X = F0 * u0 + F1 * v0 + F2;
    unsigned long long Y = F3 * u0 + F4 * v0 + F5;
    const unsigned long long dX = F1 * (4ull * n), dY = F4 * (4ull * n);
    const double inv = 1.0 / (double)(os * os);
    // One pixel per trip (not unrolled): residency hides latency better than batching (round 1 measurement).
#pragma unroll 1
    for (int k = 0; k < TH / 4; k++) {
        const int y = yb0 + 4 * k;
        const bool sane = (TH == 16 || k < 4) ? tc.sane_top : tc.sane_bot;   // rows ly + 4k: k < 4 is the upper API tile
        if (!INTERIOR && y >= h_out) break;
        float res;
        if constexpr (!OVERSAMPLED) {
            res = sample_general<INTERIOR>(tc, fv, tile, lut, X, Y, sh, sane);
        } else {
            double acc = 0.0;
            unsigned long long Xa = X, Ya = Y;
#pragma unroll 1
            for (int a = 0; a < os; a++) {
                unsigned long long Xb = Xa, Yb = Ya;
#pragma unroll 1
                for (int b = 0; b < os; b++) {
                    acc += (double)sample_general<INTERIOR>(tc, fv, tile, lut, Xb, Yb, sh, sane);
                    Xb += F0;
                    Yb += F3;
                }
                Xa += F1;
                Ya += F4;
            }
            res = (float)(acc * inv);
        }
        X += dX;
        Y += dY;
        *op = res;
        op += row_stride;
        if (wp) {
            *wp = (res == res) ? 1 : 0;
            wp += row_stride;
        }
    }
}

// ---- the tile pass ---------------------------------------------------------------------------------------------------
// One thread per (frame, workgroup tile).  th = 16 or 32 output rows per workgroup tile (gy counts those).  os = 1, or the
// oversampling factor: the transform then belongs to the os-times finer grid and a tile covers the fine pixels of its output
// pixels.  Whether a pixel is DEFINED is decided per 64 x 16 API tile (the oracle's rule: corner coordinates within +-1e9,
// coefficients below 2^30); a 32-row workgroup tile carries that flag for its upper and its lower half.
// ---- bad-pixel mask by scatter ---------------------------------------------------------------------------------------
// A masked input pixel makes NaN every output pixel whose 6 x 6 window holds it.  With the usual handful of bad pixels per
// ten thousand it is cheaper to resample WITHOUT the mask (the mask bytes double the footprint loads of every tile: 5.8
// against 4.2 ms for C5's 16 x 8192^2 share) and to poison those output pixels afterwards: mask_list_kernel compacts the bad
// pixels into a list (capacity: 1 / 64 of the pixels, at most 2^20; more than that, or per-tile transforms, or a strongly magnifying transform whose
// preimage of a 6 x 6 input square is large: the mask is applied in the resample kernel as before - decided on the device,
// tile by tile, nothing synchronises with the host), mask_scatter_kernel walks list x frames, inverts the frame's transform
// in float64 to bound the candidate output pixels and tests each candidate with the kernel's own fixed-point coordinates.
constexpr int kMaskListCapMax = 1 << 20;                 // the list holds up to 1 / 64 of the frame's pixels, at most this many

// the frame's transform can be handled by the scatter: coefficients usable, invertible, preimage of a 6 x 6 input square at
// most ~50 (fine) output pixels wide and tall
__device__ __forceinline__ bool mask_scatter_ok(const double *a)
{
    const double amax = fmax(fmax(fmax(fabs(a[0]), fabs(a[1])), fmax(fabs(a[2]), fabs(a[3]))), fmax(fabs(a[4]), fabs(a[5])));
    const double det = fma(a[0], a[4], -(a[1] * a[3]));
    const bool coef_ok = (amax < 1073741824.0) && (a[0] == a[0]) && (a[1] == a[1]) && (a[2] == a[2]) && (a[3] == a[3]) && (a[4] == a[4]) && (a[5] == a[5]);
    const double hw = 3.0 * (fabs(a[4]) + fabs(a[1])), hh = 3.0 * (fabs(a[3]) + fabs(a[0]));     // half extents * |det|
    return coef_ok && fabs(det) > 1e-300 && hw <= 24.0 * fabs(det) && hh <= 24.0 * fabs(det);
}

// Wave-aggregated append (round 4): one atomicAdd per wavefront that holds bad pixels instead of one per bad pixel - every
// atomic of the per-pixel form hit the same counter and serialised (0.24 ms for an 8192 x 8192 mask with ~0.1 % bad pixels,
// against the 13 us it takes to read the mask).  `c` bad pixels of this lane -> the lane's first slot in the list.
__device__ __forceinline__ int mask_list_reserve(int c, int *ctl)
{
    int incl = c;                                           // inclusive scan over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += up;
    }
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if ((threadIdx.x & 63) == 63) base = atomicAdd(&ctl[0], total);
    base = __shfl(base, 63, 64);
    return base + incl - c;
}

// number of non-zero bytes of a word
__device__ __forceinline__ int nonzero_bytes(unsigned w)
{
    const unsigned t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
    return __builtin_popcount(t);
}

__global__ __launch_bounds__(256) void mask_list_kernel(const uint8_t *__restrict__ mask, int64_t n, int cap, int *__restrict__ ctl,
                                                       int *__restrict__ list)
{
    // 16 mask bytes per lane and load (the frames' masks are 16-byte aligned like every plane here; a misaligned one takes the
    // byte loop).  TWO passes over the wavefront's share - count, ONE reservation (mask_list_reserve: a scan and one atomicAdd),
    // fill: with one reservation per loop trip the 65,000 atomics of an 8192 x 8192 mask (0.2 % bad pixels: nearly every
    // 1024-pixel trip holds one) serialised on the one counter and cost 0.21 ms where reading the mask takes 13 us; the second
    // read of the share comes from the cache.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
    const int64_t n16 = vec ? n / 16 : 0;
    const int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    for (int64_t q = q0; q < n16; q += stride) {
        const uint4 w = reinterpret_cast<const uint4 *>(mask)[q];
        if ((w.x | w.y | w.z | w.w) != 0) c += nonzero_bytes(w.x) + nonzero_bytes(w.y) + nonzero_bytes(w.z) + nonzero_bytes(w.w);
    }
    if (__builtin_amdgcn_ballot_w64(c != 0) != 0) {         // (wave-uniform: the scan needs all lanes)
        int i = mask_list_reserve(c, ctl);
        for (int64_t q = q0; q < n16 && c; q += stride) {
            const uint4 w = reinterpret_cast<const uint4 *>(mask)[q];
            if ((w.x | w.y | w.z | w.w) == 0) continue;
            const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if ((ws[k >> 2] >> (8 * (k & 3))) & 0xffu) {
                    if (i < cap) list[i] = (int)(q * 16 + k);
                    i++;
                }
            }
        }
    }
    for (int64_t p = n16 * 16 + q0; p < n; p += stride) {
        if (mask[p]) {
            const int i = atomicAdd(&ctl[0], 1);
            if (i < cap) list[i] = (int)p;
        }
    }
}

__global__ __launch_bounds__(256) void mask_scatter_kernel(const int *__restrict__ ctl, const int *__restrict__ list, int cap,
                                                          const double *__restrict__ affines, int os, int w_in, float *__restrict__ out,
                                                          uint8_t *__restrict__ wout, int h_out, int w_out)
{
    const int count = ctl[0];
    if (count > cap) return;                                   // the resample kernel applied the mask itself
    const int64_t f = blockIdx.y;
    const double *A = affines + 6 * f;
    const double a[6] = {A[0], A[1], A[2], A[3], A[4], A[5]};
    if (!mask_scatter_ok(a)) return;                           // (ditto, for this frame)
    long long F[6];
#pragma unroll
    for (int k = 0; k < 6; k++) F[k] = __double2ll_rn(a[k] * 4294967296.0);
    const double idet = 1.0 / fma(a[0], a[4], -(a[1] * a[3]));
    const long long wf = (long long)w_out * os, hf = (long long)h_out * os;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
        const int p = list[e];
        const int by = p / w_in, bx = p - by * w_in;
        // preimage of the input square [bx - 3, bx + 4) x [by - 3, by + 4) on the (fine) output grid: a parallelogram; its box
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
        ub = ub > wf - 1 ? wf - 1 : ub; vb = vb > hf - 1 ? hf - 1 : vb;
        for (long long v = va; v <= vb; v++) {
            for (long long u = ua; u <= ub; u++) {
                // the kernel's own coordinates: the window of (u, v) covers columns jx - 2 .. jx + 3, rows jy - 2 .. jy + 3
                const unsigned long long X = (unsigned long long)F[0] * (unsigned long long)u + (unsigned long long)F[1] * (unsigned long long)v + (unsigned long long)F[2];
                const unsigned long long Y = (unsigned long long)F[3] * (unsigned long long)u + (unsigned long long)F[4] * (unsigned long long)v + (unsigned long long)F[5];
                const long long jx = (long long)X >> 32, jy = (long long)Y >> 32;
                if (jx >= bx - 3 && jx <= bx + 2 && jy >= by - 3 && jy <= by + 2) {
                    const int64_t o = (f * h_out + v / os) * (int64_t)w_out + u / os;
                    out[o] = __builtin_nanf("");
                    if (wout) wout[o] = 0;
                }
            }
        }
    }
}

__device__ __forceinline__ bool tile_corners_ok(const double *a, long long ua, long long ub, long long va, long long vb)
{
    const long long cu[4] = {ua, ub, ua, ub}, cv[4] = {va, va, vb, vb};
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double xi = fma(a[0], (double)cu[k], fma(a[1], (double)cv[k], a[2]));
        const double yi = fma(a[3], (double)cu[k], fma(a[4], (double)cv[k], a[5]));
        ok = ok && (xi > -1e9) && (xi < 1e9) && (yi > -1e9) && (yi < 1e9);      // false for NaN
    }
    return ok;
}

__global__ __launch_bounds__(256) void resample_tiles_kernel(const double *__restrict__ affines, int per_tile, int conserve_flux,
                                                            const float *__restrict__ fscale, int os, int th, int gx, int gy, int64_t ntiles,
                                                            int h_in, int w_in, int h_out, int w_out, int fast_ok, int mask_scatter, TileRec *__restrict__ recs)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ntiles) return;
    const int tx = (int)(idx % gx), ty = (int)((idx / gx) % gy);
    const int64_t f = idx / ((int64_t)gx * gy);
    const int x0 = tx * kTileW, y0 = ty * th;
    // one transform per frame, or one per output tile (th = 16 then)
    const double *A = affines + 6 * (per_tile ? idx : f);
    const double a[6] = {A[0], A[1], A[2], A[3], A[4], A[5]};
    float fs = fscale ? fscale[f] : 1.0f;
    if (conserve_flux) fs = (float)((double)fs * fabs(fma(a[0], a[4], -(a[1] * a[3]))));   // (fine) output pixel area in input pixels
    // the coefficients below 2^30 (the fixed-point evaluation is then exact while the true sums fit 64 bits); false for NaN
    const double amax = fmax(fmax(fmax(fabs(a[0]), fabs(a[1])), fmax(fabs(a[2]), fabs(a[3]))), fmax(fabs(a[4]), fabs(a[5])));
    const bool coef_ok = (amax < 1073741824.0) && (a[0] == a[0]) && (a[1] == a[1]) && (a[2] == a[2]) && (a[3] == a[3]) && (a[4] == a[4]) && (a[5] == a[5]);
    // corner pixels on the (fine) output grid, per API tile: an affine map takes its extremes there
    const int xl = x0 + kTileW - 1 < w_out - 1 ? x0 + kTileW - 1 : w_out - 1;
    const long long ua = (long long)x0 * os, ub = (long long)xl * os + (os - 1);
    const int yl_top = y0 + kTileH - 1 < h_out - 1 ? y0 + kTileH - 1 : h_out - 1;
    const bool has_bot = th > kTileH && y0 + kTileH < h_out;
    const int yl_bot = y0 + th - 1 < h_out - 1 ? y0 + th - 1 : h_out - 1;
    const bool sane_top = coef_ok && tile_corners_ok(a, ua, ub, (long long)y0 * os, (long long)yl_top * os + (os - 1));
    const bool sane_bot = has_bot ? coef_ok && tile_corners_ok(a, ua, ub, (long long)(y0 + kTileH) * os, (long long)yl_bot * os + (os - 1)) : sane_top;
    const bool sane = sane_top && sane_bot;
    TileRec rec;
#pragma unroll
    for (int k = 0; k < 6; k++) rec.F[k] = coef_ok ? __double2ll_rn(a[k] * 4294967296.0) : 0;
    int bx0 = 0, by0 = 0, w = 0, h = 0;
    unsigned flags = (sane_top ? kSaneTop : 0u) | (sane_bot ? kSaneBot : 0u);
    if (!(mask_scatter && !per_tile && mask_scatter_ok(a))) flags |= kInlineMask;     // a mask, if any, is applied in the resample kernel
    if (sane) {
        flags |= kSane;
        // the footprint from the SAME integer coordinates the pixels will use (linear: extremes at the corners)
        const int yl = has_bot ? yl_bot : yl_top;
        const long long va = (long long)y0 * os, vb = (long long)yl * os + (os - 1);
        const long long cu[4] = {ua, ub, ua, ub}, cv[4] = {va, va, vb, vb};
        long long jx0 = 0x7fffffffffffffffLL, jx1 = -jx0, jy0 = jx0, jy1 = -jx0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long X = (unsigned long long)rec.F[0] * (unsigned long long)cu[k] + (unsigned long long)rec.F[1] * (unsigned long long)cv[k] + (unsigned long long)rec.F[2];
            const unsigned long long Y = (unsigned long long)rec.F[3] * (unsigned long long)cu[k] + (unsigned long long)rec.F[4] * (unsigned long long)cv[k] + (unsigned long long)rec.F[5];
            const long long jx = (long long)X >> 32, jy = (long long)Y >> 32;
            jx0 = jx < jx0 ? jx : jx0; jx1 = jx > jx1 ? jx : jx1;
            jy0 = jy < jy0 ? jy : jy0; jy1 = jy > jy1 ? jy : jy1;
        }
        bx0 = (int)jx0 - 2;
        by0 = (int)jy0 - 2;
        const long long wl = jx1 - jx0 + 6, hl = jy1 - jy0 + 6;
        const bool staged = wl <= kGenericFloats && hl <= kGenericFloats && wl * hl <= kGenericFloats;
        if (staged) {
            w = (int)wl;
            h = (int)hl;
            flags |= kStaged;
            const bool interior = bx0 >= 0 && by0 >= 0 && bx0 + w <= w_in && by0 + h <= h_in && y0 + th <= h_out;
            if (interior) flags |= kInterior;
#ifndef APGPU_VARIANT_RESAMPLE_NO_FAST
            if (interior && fast_ok && w <= kFastPitch && h <= th + 10) flags |= kFast;
#endif
        }
    }
    rec.bx0 = bx0;
    rec.by0 = by0;
    rec.dims = (unsigned)w | ((unsigned)h << 12) | (flags << 24);
    rec.fs = fs;
    recs[idx] = rec;
}

// The footprint of a FAST tile on its way from global memory to LDS: 3 rows of 80 columns per trip (240 of the 256 lanes),
// every load of the tile in flight before the first LDS store.  Columns beyond the footprint's width are read too (inside
// the frame's buffer, or returned as 0 by the bounds check) and never used.
#ifndef APGPU_RESAMPLE_FILL_UNCONDITIONAL
#define APGPU_RESAMPLE_FILL_UNCONDITIONAL 1
#endif
template <bool HAS_MASK, int TRIPS>
struct FastFill {
    float val[TRIPS];
    char mk[TRIPS];
};

template <bool HAS_MASK, int TRIPS>
__device__ __forceinline__ void fast_fill_issue(FastFill<HAS_MASK, TRIPS> &ff, const float *src, const uint8_t *mask, int bx0, int by0, int fh,
                                                int h_in, int w_in, int tid)
{
    const v4i irsrc = make_rsrc(src, (unsigned)(h_in * w_in) * 4u);
    const v4i mrsrc = make_rsrc(mask, (unsigned)(h_in * w_in));
    const int r = tid / kFastPitch, c = tid - r * kFastPitch;
    const int e0 = (by0 + r) * w_in + bx0 + c;
    const int estep = 3 * w_in;
#if APGPU_RESAMPLE_FILL_UNCONDITIONAL
    // Round 6: EVERY lane issues EVERY trip's load (rows beyond the footprint's height and the lanes 240 .. 255 read inside the
    // frame's buffer or get the bounds check's 0; fast_fill_store drops them).  Behind "if (row < fh && tid < 240)" each load sat in
    // its own exec-masked block, the compiler's wait-count pass could not count them, and the wait in front of the LDS stores was
    // vmcnt(0) - a wait for whatever else was in flight as well, i.e. the first pixels' weight rows could not be fetched beside the
    // footprint (resample_affine_kernel issues them between these loads and the stores since this round).
#pragma unroll
    for (int k = 0; k < TRIPS; k++) {
        ff.val[k] = apgpu_buffer_load_f32(irsrc, (e0 + k * estep) * 4, 0, 0);
        if constexpr (HAS_MASK) ff.mk[k] = apgpu_buffer_load_i8(mrsrc, e0 + k * estep, 0, 0);
        else ff.mk[k] = 0;
    }
#else
#pragma unroll
    for (int k = 0; k < TRIPS; k++) {
        ff.val[k] = 0.f;
        ff.mk[k] = 0;
        if (3 * k < fh && tid < 3 * kFastPitch) {               // (the first test is scalar)
            ff.val[k] = apgpu_buffer_load_f32(irsrc, (e0 + k * estep) * 4, 0, 0);
            if constexpr (HAS_MASK) ff.mk[k] = apgpu_buffer_load_i8(mrsrc, e0 + k * estep, 0, 0);
        }
    }
#endif
}

template <bool HAS_MASK, int TRIPS, int OFFB>
__device__ __forceinline__ void fast_fill_store(const FastFill<HAS_MASK, TRIPS> &ff, int fh, float *tile, int tid)
{
#pragma unroll
    for (int k = 0; k < TRIPS; k++) {
        if (3 * k < fh && tid < 3 * kFastPitch) {               // (a trip's rows beyond fh land in the spare rows)
            const bool good = (fabsf(ff.val[k]) < __builtin_inff()) && ff.mk[k] == 0;
            const float xv = good ? ff.val[k] : __builtin_nanf("");
            tile[tid + 3 * kFastPitch * k] = xv;                       // copy A
            tile[OFFB - 1 + tid + 3 * kFastPitch * k] = xv;            // copy B: element e - 1 (e = 0 lands in the gap)
        }
    }
}

// General staged fill (validity by position applied here): a wave takes every 4th footprint row (row address math is
// scalar), 3 rows and up to 2 x 64 columns per trip with clamped - always valid - addresses, so that all the loads of a trip
// are in flight together.  HAS_MASK is a template flag: as a run-time test every mask load became a branch followed by a
// full wait, which serialised the whole batch of loads.
template <bool HAS_MASK>
__device__ __forceinline__ void general_fill(const TileCtx &tc, const FrameView &fv, float *tile, int tid)
{
    const int bx0 = tc.bx0, by0 = tc.by0, fw = tc.fw, h = tc.fh;
    const int h_in = fv.h_in, w_in = fv.w_in;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
    constexpr int RU = 3;
    for (int r0 = wave; r0 < h; r0 += 4 * RU) {
        for (int c0 = 0; c0 < fw; c0 += 2 * kWave) {
            float val[RU][2];
            uint8_t mk[RU][2];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int row = by0 + r0 + 4 * u;
                const int rc = row < 0 ? 0 : (row >= h_in ? h_in - 1 : row);
                const float *rp2 = fv.src + (int64_t)rc * w_in;
                const uint8_t *mp = HAS_MASK ? fv.mask + (int64_t)rc * w_in : nullptr;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int col = bx0 + c0 + q * kWave + lane;
                    const int cc = col < 0 ? 0 : (col >= w_in ? w_in - 1 : col);
                    val[u][q] = rp2[cc];
                    if constexpr (HAS_MASK) mk[u][q] = mp[cc];
                    else mk[u][q] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int r = r0 + 4 * u;
                const int row = by0 + r;
                const bool row_ok = row >= 0 && row < h_in;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int c = c0 + q * kWave + lane;
                    const int col = bx0 + c;
                    const bool good = row_ok && col >= 0 && col < w_in && (fabsf(val[u][q]) < __builtin_inff()) && mk[u][q] == 0;
                    if (r < h && c < fw) tile[r * fw + c] = good ? val[u][q] : __builtin_nanf("");
                }
            }
        }
    }
}

}  // namespace
