// sigclip_global.hip - A3: astropy.stats.sigma_clipped_stats(data, sigma) with axis=None, the call
// ApFindBadPixels makes on a master dark (core/ApFindBadPixels.py:191), on gfx950.
//
// Reference algorithm (astropy/stats/sigma_clipping.py:385-433 _sigmaclip_noaxis with numpy
// nan-functions - the reference environment has no bottleneck):
//     x = finite values of data (order kept)
//     repeat <= maxiters:  c = np.nanmedian(x); s = np.nanstd(x)
//                          x = x[(x >= c - s*sigma_lower) & (x <= c + s*sigma_upper)]
//                          stop when nothing was removed
//     return np.nanmean(x), np.nanmedian(x), np.nanstd(x)
// For float32 input numpy does all of this in float32 and the result depends on numpy's summation
// tree, so the tree is reproduced exactly (bit-exact statistics -> bit-exact bad-pixel mask):
//     np.sum      = sequential float32 fold of 8192-element pieces, each piece a pairwise tree over
//                   128-element leaves with 8 strided accumulators  (numpy loops_utils.h.src)
//     np.median   = exact order statistic (radix select on order-preserving keys, 11-bit digits: 3 levels for
//                   float32, 6 for float64); even count -> float32(a + b) / 2
//     np.var      = mean = sum / float32(n); float32 sum of (x - mean)^2; float32(float64(sum) / n)
//     bounds      = float64 (np.float32 * python float promotes to float64 under numpy 1.26), demoted
//                   to float32 when compared with the float32 array
// The surviving values are kept compacted IN ORDER in a ping-pong pair of workspace buffers because
// the summation tree depends on element positions.
//
// Everything runs on the stream without host synchronisation: the iteration count is a launch-time
// constant and a device-side `done` flag turns the remaining iterations into no-ops.
// Traffic per iteration (float32): 3 fused statistics reads (each one select level, the first two also one of numpy's
// sums), the tile count and the compaction (1 read + 1 write) - 5 reads + 1 write of the surviving values, 67 MB at
// 4096^2 and resident in the 256 MB Infinity Cache after the first pass; 9 launches (round 1: 10 reads, 20 launches).
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kBlock = 256;
constexpr int kTile = 2048;            // elements per compaction tile (8 per thread)
constexpr int kPiece = 8192;           // numpy reduction buffer
constexpr int kLeaf = 128;             // numpy PW_BLOCKSIZE
constexpr int kMaxItersCap = 32;
constexpr int kDigit = 11;             // radix-select digit: float32 keys in 3 levels (11 + 11 + 10 bits), float64 in 6
constexpr int kBins = 1 << kDigit;
constexpr int kScanBlock = 512;        // workgroup of the fused statistics passes (8 wavefronts, one 8192-element piece each)
constexpr unsigned kSkip = 0xffffffffu;
constexpr int kMaxPassGroups = kNumCU * 2;  // workgroups of a statistics pass

// T = float : float32 data, numpy float32 statistics (float32 darks / flats)
// T = double: float64 data and statistics - what numpy computes for INTEGER images (np.median/np.var of
//             a uint16 array run in float64; the host widens integer images exactly) and for float64 data.
struct IterState {
    long long m;                // survivors in the current buffer
    int cur;                    // index (0/1) of the buffer holding them
    int done;                   // set when an iteration removed nothing: its statistics are final
    int iter;                   // iterations executed
    int pad;
};

struct GState {
    IterState it[2];            // it[i & 1] is read by iteration i and written by iteration i - 1's tile scan
    long long k;                // rank searched by the radix select (relative to the current prefix)
    unsigned long long prefix;          // key prefix found so far
    unsigned long long min_key, max_key;    // extremes of the survivors
    double tot;                 // np.sum of the survivors          (holds a float32 value when T = float)
    double s2;                  // np.sum((x - mean)^2)
    double med, sd;
    double lo, hi;
    unsigned hist[kBins];
    unsigned long long wg_lo[kMaxPassGroups + 1], wg_hi[kMaxPassGroups + 1];   // per-workgroup extremes of a statistics pass
};

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = unsigned;
    static constexpr int bits = 32;
    __device__ static unsigned to(float x)
    {
        const unsigned b = __float_as_uint(x);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    __device__ static float from(unsigned long long k64)
    {
        const unsigned k = (unsigned)k64;
        const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __uint_as_float(b);
    }
};
template <> struct KeyOf<double> {
    using type = unsigned long long;
    static constexpr int bits = 64;
    __device__ static unsigned long long to(double x)
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    __device__ static double from(unsigned long long k)
    {
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)b);
    }
};
template <typename T> constexpr int key_levels() { return (KeyOf<T>::bits + kDigit - 1) / kDigit; }
// level l looks at key bits [shift, shift + width)
template <typename T> __host__ __device__ constexpr int level_shift(int l)
{
    return KeyOf<T>::bits - kDigit * (l + 1) > 0 ? KeyOf<T>::bits - kDigit * (l + 1) : 0;
}
template <typename T> __host__ __device__ constexpr int level_width(int l)
{
    return l < key_levels<T>() - 1 ? kDigit : KeyOf<T>::bits - kDigit * (key_levels<T>() - 1);
}

template <typename T> __device__ __forceinline__ bool is_finite(T x) { return fabs((double)x) < __builtin_inf(); }

// ---- order-preserving compaction: tile counts -> scan -> scatter -------------------------------
// mode 0: keep finite values of `data` (length n, host-known); mode 1: keep lo <= x <= hi of the
// current survivors (iteration state `slot`).
template <int MODE, typename T>
__device__ __forceinline__ bool keep_pred(T x, T lof, T hif)
{
    if constexpr (MODE == 0) return is_finite<T>(x);
    else return (x >= lof) && (x <= hif);
}

template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void tile_count_kernel(const T *__restrict__ src0, const T *__restrict__ src1,
                                                           long long n_static, const GState *__restrict__ st, int slot,
                                                           unsigned *__restrict__ tile_counts)
{
    const IterState s = st->it[slot];
    if (MODE == 1 && s.done) return;
    const T *src = (MODE == 0) ? src0 : (s.cur ? src1 : src0);
    const long long m = (MODE == 0) ? n_static : s.m;
    // float32: the float64 bounds are demoted to float32 for the comparison, as numpy 1.26 does
    const T lof = (T)st->lo, hif = (T)st->hi;
    const long long ntiles = (m + kTile - 1) / kTile;
    __shared__ unsigned wsum[kBlock / kWave];
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T x[V]; };
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + (long long)threadIdx.x * 8;
        unsigned c = 0;
        if (aligned && base + 8 <= m) {
#pragma unroll
            for (int j = 0; j < 8 / V; j++) {
                const Vec v = reinterpret_cast<const Vec *>(src + base)[j];
#pragma unroll
                for (int q = 0; q < V; q++) c += keep_pred<MODE, T>(v.x[q], lof, hif) ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const long long i = base + j;
                if (i < m) c += keep_pred<MODE, T>(src[i], lof, hif) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) c += __shfl_down(c, d);
        if ((threadIdx.x % kWave) == 0) wsum[threadIdx.x / kWave] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// Exclusive scan of the tile counts; publishes the NEXT iteration's state (slot ^ 1) - there is no separate commit
// kernel - and resets the per-iteration extremes.  Runs (as a copy) even when the iteration is a no-op, so that the
// state a later iteration or the final pass reads is always current.
template <int MODE>
__global__ __launch_bounds__(1024) void tile_scan_kernel(long long n_static, GState *__restrict__ st, int slot,
                                                        const unsigned *__restrict__ tile_counts,
                                                        unsigned long long *__restrict__ tile_offsets)
{
    const IterState s = st->it[slot];
    IterState *next = MODE == 0 ? &st->it[0] : &st->it[slot ^ 1];
    if (MODE == 1 && s.done) {
        if (threadIdx.x == 0) *next = s;
        return;
    }
    if (MODE == 1) {                                        // the iteration's last histogram was read by the count kernel
        for (int t = threadIdx.x; t < kBins; t += 1024) st->hist[t] = 0;
    }
    const long long m = (MODE == 0) ? n_static : s.m;
    const long long ntiles = (m + kTile - 1) / kTile;
    // rounds of 4096 tiles: four consecutive counts per thread (one 16-byte load), wavefront scan, 16 wavefront totals
    __shared__ unsigned long long wtot[1024 / kWave];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    unsigned long long carry = 0;
    for (long long base = 0; base < ntiles; base += 4096) {
        const long long t0 = base + (long long)threadIdx.x * 4;
        unsigned c[4] = {0, 0, 0, 0};
        if (t0 + 4 <= ntiles) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tile_counts + t0);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
            for (int j = 0; j < 4; j++)
                if (t0 + j < ntiles) c[j] = tile_counts[t0 + j];
        }
        const unsigned long long local = (unsigned long long)c[0] + c[1] + c[2] + c[3];
        unsigned long long inc = local;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wtot[wave] = inc;
        __syncthreads();
        unsigned long long woff = carry;
        for (int w = 0; w < wave; w++) woff += wtot[w];
        unsigned long long run = woff + inc - local;
        if (t0 + 4 <= ntiles) {
            ulonglong2 o0, o1;
            o0.x = run; o0.y = run + c[0];
            o1.x = o0.y + c[1]; o1.y = o1.x + c[2];
            reinterpret_cast<ulonglong2 *>(tile_offsets + t0)[0] = o0;
            reinterpret_cast<ulonglong2 *>(tile_offsets + t0)[1] = o1;
        } else {
            for (int j = 0; j < 4; j++)
                if (t0 + j < ntiles) { tile_offsets[t0 + j] = run; run += c[j]; }
        }
        if (threadIdx.x == 1023) carry_s = woff + inc;
        __syncthreads();
        carry = carry_s;
    }
    if (threadIdx.x == 1023) {
        const long long m_next = (long long)carry;
        IterState nx;
        nx.pad = 0;
        if (MODE == 0) {
            nx.m = m_next; nx.cur = 0; nx.done = 0; nx.iter = 0;
            st->lo = __builtin_nan("");
            st->hi = __builtin_nan("");
        } else if (m_next == s.m) {             // nothing removed: survivors stay in buffer `cur`, statistics are final
            nx.m = s.m; nx.cur = s.cur; nx.done = 1; nx.iter = s.iter + 1;
        } else {
            nx.m = m_next; nx.cur = s.cur ^ 1; nx.done = 0; nx.iter = s.iter + 1;
        }
        *next = nx;
    }
}

template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void tile_scatter_kernel(const T *__restrict__ src0, const T *__restrict__ src1,
                                                             T *__restrict__ dst0, T *__restrict__ dst1,
                                                             long long n_static, const GState *__restrict__ st, int slot,
                                                             const unsigned long long *__restrict__ tile_offsets)
{
    const IterState s = st->it[slot];
    if (MODE == 1 && (s.done || st->it[slot ^ 1].done)) return;     // nothing to remove: no copy
    const T *src = (MODE == 0) ? src0 : (s.cur ? src1 : src0);
    T *dst = (MODE == 0) ? dst0 : (s.cur ? dst0 : dst1);
    const long long m = (MODE == 0) ? n_static : s.m;
    const T lof = (T)st->lo, hif = (T)st->hi;
    const long long ntiles = (m + kTile - 1) / kTile;
    __shared__ unsigned wsum[kBlock / kWave];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T x[V]; };
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + (long long)threadIdx.x * 8;
        T x[8];
        bool kp[8];
        unsigned c = 0;
        if (aligned && base + 8 <= m) {
#pragma unroll
            for (int j = 0; j < 8 / V; j++) {
                const Vec v = reinterpret_cast<const Vec *>(src + base)[j];
#pragma unroll
                for (int q = 0; q < V; q++) x[j * V + q] = v.x[q];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                kp[j] = keep_pred<MODE, T>(x[j], lof, hif);
                c += kp[j] ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const long long i = base + j;
                x[j] = i < m ? src[i] : (T)0;
                kp[j] = (i < m) && keep_pred<MODE, T>(x[j], lof, hif);
                c += kp[j] ? 1u : 0u;
            }
        }
        // exclusive scan of c over the block: wave inclusive scan + wave totals
        unsigned inc = c;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wsum[wave] = inc;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        unsigned long long pos = tile_offsets[tile] + woff + (inc - c);
        if (c == 8 && (pos % V) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
#pragma unroll
            for (int j = 0; j < 8 / V; j++) {
                Vec v;
#pragma unroll
                for (int q = 0; q < V; q++) v.x[q] = x[j * V + q];
                reinterpret_cast<Vec *>(dst + pos)[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (kp[j]) dst[pos++] = x[j];
        }
        __syncthreads();
    }
}

// ---- numpy pairwise sums --------------------------------------------------------------------------
template <int SQ, typename T>
__device__ __forceinline__ T tr(T x, T mean)
{
    if constexpr (SQ) { const T d = x - mean; return d * d; }
    else return x;
}

template <typename T>
__device__ __forceinline__ T var_mean(const GState *st, long long m)
{
    return (T)st->tot / (T)m;               // np.var: arrmean = true_divide(sum, n) in the array's dtype
}

// LDS index of element i of the ragged piece: one pad word per 64 elements, so that lanes walking different leaves
// (starts 64 .. 128 elements apart) do not all sit on one bank.
__device__ __forceinline__ int ridx(int i) { return i + (i >> 6); }
constexpr int kRaggedLds = kPiece + kPiece / 64;

template <int SQ, typename T>
__device__ T ragged_leaf(const T *rag, int off, int n, T mean)
{
    if (n < 8) {
        T res = 0;
        for (int i = 0; i < n; i++) res = res + tr<SQ, T>(rag[ridx(off + i)], mean);
        return res;
    }
    T r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = tr<SQ, T>(rag[ridx(off + k)], mean);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = r[k] + tr<SQ, T>(rag[ridx(off + i + k)], mean);
    }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + tr<SQ, T>(rag[ridx(off + i)], mean);
    return res;
}

// numpy's recursion over the ragged piece (rem < 8192 values staged in LDS), in parallel: every 64th element descends the
// split tree to its leaf (leaves are 64 .. 128 elements long, so each leaf is found by the thread at the first multiple of
// 64 inside it), the leaves are summed side by side, and the tree is folded level by level, deepest first; a node's value
// lives in the slot of the thread that owns its leftmost leaf.  Called by every thread of the workgroup (>= 128 threads);
// the sum is returned to all of them.
template <int SQ, typename T>
__device__ T ragged_tree_sum(const T *ragged, int rem, T mean, T *vals /* LDS [kPiece / 64] */)
{
    const int t = threadIdx.x;
    const int p = t * 64;
    const bool valid = t < kPiece / 64 && p < rem;
    int off = 0, n = rem;
    int path_off[8], path_n[8];
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const bool internal = valid && n > kLeaf;
        path_off[d] = off;
        path_n[d] = internal ? n : 0;
        if (internal) {
            int n2 = n / 2;
            n2 -= n2 % 8;
            if (p < off + n2) n = n2;
            else { off += n2; n -= n2; }
        }
    }
    const bool owner = valid && (off + 63) / 64 == t;
    if (owner) vals[t] = ragged_leaf<SQ, T>(ragged, off, n, mean);
    __syncthreads();
#pragma unroll
    for (int d = 7; d >= 0; d--) {
        if (owner && path_n[d] > 0 && path_off[d] == off) {
            int n2 = path_n[d] / 2;
            n2 -= n2 % 8;
            vals[t] = vals[t] + vals[(path_off[d] + n2 + 63) / 64];
        }
        __syncthreads();
    }
    return vals[0];
}

// One wavefront finds the bin of rank k in h[0 .. kBins): 32 bins per lane + a wavefront scan.
struct DigitPick {
    int digit;              // bin that holds rank k
    long long newk;         // rank inside that bin
    int dlow;               // highest occupied bin below it (-1: none)
};

__device__ void wave_pick_digit(const unsigned *h, long long k, int lane, DigitPick *pick /* LDS, pre-set to {0, 0, -1} */)
{
    unsigned long long c = 0;
    constexpr int per = kBins / kWave;
#pragma unroll 4
    for (int j = 0; j < per; j++) c += h[lane * per + j];
    unsigned long long inc = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    const unsigned long long exc = inc - c;
    int dlow = -1;
    if ((unsigned long long)k >= exc && (unsigned long long)k < inc) {
        unsigned long long run = exc;
        for (int j = 0; j < per; j++) {
            const unsigned cnt = h[lane * per + j];
            if ((unsigned long long)k < run + cnt) { pick->digit = lane * per + j; pick->newk = k - (long long)run; break; }
            if (cnt) dlow = lane * per + j;
            run += cnt;
        }
    } else if ((unsigned long long)k >= inc) {
        for (int j = per - 1; j >= 0; j--)
            if (h[lane * per + j]) { dlow = lane * per + j; break; }
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const int o = __shfl_down(dlow, d);
        dlow = o > dlow ? o : dlow;
    }
    if (lane == 0) pick->dlow = dlow;
}

// np.median from the last select level: `prefix` = the key bits found before it, `below` = the largest key under that prefix.
template <typename T>
__device__ double median_of_pick(long long m, unsigned long long prefix, int width, const DigitPick &pk, unsigned long long below)
{
    using K = KeyOf<T>;
    if (m <= 0) return __builtin_nan("");
    const T vhi = K::from((prefix << width) | (unsigned long long)pk.digit);
    if (m & 1) return (double)vhi;
    // rank m/2 - 1: the same key again if the searched rank is not the first of its bin; otherwise the highest occupied
    // lower bin of this prefix, otherwise the largest key below the prefix
    T vlo;
    if (pk.newk >= 1) vlo = vhi;
    else if (pk.dlow >= 0) vlo = K::from((prefix << width) | (unsigned long long)pk.dlow);
    else vlo = K::from(below);
    const T t = vlo + vhi;                                  // np.mean of the two middle values, in T
    return (double)(T)((double)t / 2.0);
}

// ---- fused statistics pass ---------------------------------------------------------------------------
// One read of the survivors does one level of the radix select (exact median) and, on the first two levels, one of
// numpy's two sums:
//     level 0:  histogram of the top 11 key bits          + piece sums of x              (+ min / max key)
//     level 1:  histogram of the next 11 bits (prefix)    + piece sums of (x - mean)^2
//     level 2+: histogram of the next digit               (last level: + largest key below the prefix, for the lower
//                                                           middle element of an even count)
// so an iteration reads its data 3 (float32) or 6 (float64) times for the statistics instead of 7 or 11.
// A wavefront owns an 8192-element piece and a lane one 128-element leaf of it (numpy's order); the lane's digits
// are run-length coded before they go to the LDS histogram, because a dark frame's values share their leading
// digits and 64 lanes adding to one bin serialise.
// KIND: 0 first level, 1 middle, 2 last.
template <typename T, int SUM, int KIND>
__global__ __launch_bounds__(kScanBlock) void pass_kernel(const T *__restrict__ b0, const T *__restrict__ b1,
                                                         GState *__restrict__ st, T *__restrict__ piece_sums, int slot, int level)
{
    using K = KeyOf<T>;
    using KT = typename K::type;
    const IterState s = st->it[slot];
    if (s.done) return;
    const T *src = s.cur ? b1 : b0;
    const long long m = s.m;
    const T mean = SUM == 2 ? var_mean<T>(st, m) : (T)0;
    const int shift = level_shift<T>(level), width = level_width<T>(level);
    const KT prefix = (KT)st->prefix;
    const unsigned dmask = (1u << width) - 1u;
    __shared__ unsigned h[kBins];
    for (int t = threadIdx.x; t < kBins; t += kScanBlock) h[t] = 0;
    __syncthreads();
    const long long npieces_full = m / kPiece;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    constexpr int wpb = kScanBlock / kWave;
    // Lane (g, k) = (lane / 8, lane % 8) owns numpy's accumulator k of the leaves 8g .. 8g+7 of the piece, one leaf per
    // round: a load instruction then reads 8 runs of 32 (64) contiguous bytes and a cache line is used up within 4 (2)
    // consecutive loads - a lane walking a whole leaf on its own would keep 64 lines per wavefront alive and thrash
    // the 32 KB vector cache.
    const int g = lane / 8, kacc = lane % 8;
    unsigned cur_d = kSkip, cur_n = 0;
    KT below = 0, kmin = ~(KT)0, kmax = 0;
    // the last workgroup takes the ragged piece (below); the others share the full pieces
    const bool tail_group = blockIdx.x == gridDim.x - 1;
    const long long nworkers = gridDim.x - 1;
    for (long long piece = tail_group ? npieces_full : (long long)blockIdx.x * wpb + wave; piece < npieces_full;
         piece += nworkers * wpb) {
        const T *pbase = src + piece * kPiece + (long long)(8 * g) * kLeaf + kacc;
        constexpr int E = kLeaf / 8;                            // values of one leaf per lane
        // sum (numpy's leaf order) and bin one leaf's values
        auto leaf = [&](const T(&x)[E]) -> T {
            T r = 0;
            if constexpr (SUM != 0) {
                r = tr<SUM == 2, T>(x[0], mean);
#pragma unroll
                for (int i = 1; i < E; i++) r = r + tr<SUM == 2, T>(x[i], mean);
                r = r + __shfl_xor(r, 1);                       // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
            }
#pragma unroll
            for (int i = 0; i < E; i++) {
                const KT key = K::to(x[i]);
                if constexpr (KIND == 0) {
                    // leading digits: nearly every value of a dark frame shares them - run-length coded per lane
                    const unsigned d = (unsigned)(key >> shift);
                    kmin = key < kmin ? key : kmin;
                    kmax = key > kmax ? key : kmax;
                    if (d != cur_d) {
                        if (cur_d != kSkip) atomicAdd(&h[cur_d], cur_n);
                        cur_d = d;
                        cur_n = 0;
                    }
                    cur_n++;
                } else {
                    // lower digits are spread out: plain LDS atomics (a constant image serialises them, at about twice
                    // the time of the pass)
                    const KT top = key >> (shift + width);
                    if (top == prefix) atomicAdd(&h[(unsigned)(key >> shift) & dmask], 1u);
                    if constexpr (KIND == 2)
                        if (top < prefix) below = key > below ? key : below;
                }
            }
            return r;
        };
        // Leaves 8g .. 8g+7 in four rounds of two; the next leaf's loads are in flight while the current one is processed.
        // The round loop is NOT unrolled: fully unrolled, the compiler hoists all 128 loads (256 VGPRs and scratch).
        T xa[E], xb[E];
#pragma unroll
        for (int i = 0; i < E; i++) xa[i] = pbase[8 * i];
        T q = 0, h0 = 0, sum = 0;
#pragma unroll 1
        for (int pp = 0; pp < 4; pp++) {
            const T *a1 = pbase + (2 * pp + 1) * kLeaf;
#pragma unroll
            for (int i = 0; i < E; i++) xb[i] = a1[8 * i];
            const T s0 = leaf(xa);
            if (pp < 3) {
#pragma unroll
                for (int i = 0; i < E; i++) xa[i] = a1[kLeaf + 8 * i];
            }
            const T s1 = leaf(xb);
            if constexpr (SUM != 0) {
                // the balanced tree over the lane group's 8 leaves: ((p0 + p1) + (p2 + p3)), p = leaf pair
                const T pr = s0 + s1;
                if (pp & 1) {
                    q = q + pr;
                    if (pp == 1) h0 = q;
                    else sum = h0 + q;
                } else {
                    q = pr;
                }
            }
        }
        if constexpr (SUM != 0) {
            // ... then across the 8 lane groups
            sum = sum + __shfl_xor(sum, 8);
            sum = sum + __shfl_xor(sum, 16);
            sum = sum + __shfl_xor(sum, 32);
            if (lane == 0) piece_sums[piece] = sum;
        }
    }
    if (KIND == 0 && cur_d != kSkip) atomicAdd(&h[cur_d], cur_n);
    if (tail_group) {
        // the ragged last piece (m % 8192 values): its share of the histogram and the extremes, and its sum by numpy's
        // irregular split tree - stored as one more piece sum, which the ordered fold adds last
        __shared__ T ragged[kRaggedLds];
        __shared__ T vals[kPiece / 64];
        const int rem = (int)(m - npieces_full * kPiece);
        for (int t = threadIdx.x; t < rem; t += kScanBlock) ragged[ridx(t)] = src[npieces_full * kPiece + t];
        __syncthreads();
        for (int t = threadIdx.x; t < rem; t += kScanBlock) {
            const KT key = K::to(ragged[ridx(t)]);
            if constexpr (KIND == 0) {
                atomicAdd(&h[(unsigned)(key >> shift)], 1u);
                kmin = key < kmin ? key : kmin;
                kmax = key > kmax ? key : kmax;
            } else {
                const KT top = key >> (shift + width);
                if (top == prefix) atomicAdd(&h[(unsigned)(key >> shift) & dmask], 1u);
                if constexpr (KIND == 2)
                    if (top < prefix) below = key > below ? key : below;
            }
        }
        if constexpr (SUM != 0) {
            if (rem > 0) {
                const T tail = ragged_tree_sum<SUM == 2, T>(ragged, rem, mean, vals);
                if (threadIdx.x == 0) piece_sums[npieces_full] = tail;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kBins; t += kScanBlock)
        if (h[t]) atomicAdd(&st->hist[t], h[t]);
    if constexpr (KIND == 0 || KIND == 2) {
        // extremes: one slot per workgroup, reduced by after_kernel (thousands of atomics on one address serialise)
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            if constexpr (KIND == 0) {
                const KT l2 = __shfl_down(kmin, d), h2 = __shfl_down(kmax, d);
                kmin = l2 < kmin ? l2 : kmin;
                kmax = h2 > kmax ? h2 : kmax;
            } else {
                const KT o = __shfl_down(below, d);
                below = o > below ? o : below;
            }
        }
        __shared__ KT w_lo[wpb], w_hi[wpb];
        if (lane == 0) {
            w_lo[wave] = kmin;
            w_hi[wave] = KIND == 0 ? kmax : below;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            KT lo = w_lo[0], hi = w_hi[0];
            for (int w = 1; w < wpb; w++) {
                lo = w_lo[w] < lo ? w_lo[w] : lo;
                hi = w_hi[w] > hi ? w_hi[w] : hi;
            }
            // widened so that "nothing seen" stays ~0 / 0 for both key widths
            st->wg_lo[blockIdx.x] = (KIND == 0 && hi >= lo) ? (unsigned long long)lo : ~0ull;
            st->wg_hi[blockIdx.x] = (KIND == 2 || hi >= lo) ? (unsigned long long)hi : 0ull;
        }
    }
}

// Closes a statistics pass (one workgroup): the sequential fold of the piece sums in numpy's order (the ragged piece's sum
// is the last of them), the digit of the searched rank - and on the last level the median and, inside a clipping iteration,
// the bounds.  (Inside an iteration the last level is closed by the tile count kernel instead.)
template <typename T, int SUM, int KIND>
__global__ __launch_bounds__(kBlock) void after_kernel(GState *__restrict__ st, const T *__restrict__ piece_sums, int slot, int level,
                                                      int final_pass, int ngroups, double sigma_lower, double sigma_upper)
{
    if (blockIdx.x != 0) return;
    const IterState s = st->it[slot];
    if (s.done) return;
    const long long m = s.m;
    const int width = level_width<T>(level);
    const unsigned long long prefix = st->prefix;
    const long long k = KIND == 0 ? m / 2 : st->k;          // upper median rank (0-based); odd m: the median itself
    const long long nsums = m / kPiece + ((m % kPiece) ? 1 : 0);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;

    __shared__ unsigned h[kBins];
    __shared__ unsigned long long sh_lo, sh_hi;
    __shared__ DigitPick pick;
    for (int t = threadIdx.x; t < kBins; t += kBlock) {
        h[t] = st->hist[t];
        st->hist[t] = 0;                            // ready for the next pass
    }
    if (threadIdx.x == 0) { sh_lo = ~0ull; sh_hi = 0; pick.digit = 0; pick.newk = 0; pick.dlow = -1; }
    __syncthreads();
    if constexpr (KIND == 0 || KIND == 2) {         // the workgroups' extremes
        unsigned long long lo = ~0ull, hi = 0;
        for (int t = threadIdx.x; t < ngroups; t += kBlock) {
            const unsigned long long l = st->wg_lo[t], u = st->wg_hi[t];
            lo = l < lo ? l : lo;
            hi = u > hi ? u : hi;
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            const unsigned long long l2 = __shfl_down(lo, d), h2 = __shfl_down(hi, d);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) { atomicMin(&sh_lo, lo); atomicMax(&sh_hi, hi); }
    }

    // wavefront 1 picks the digit (32 bins per lane) while lane 0 of wavefront 0 folds
    constexpr int kStage = 2048;
    constexpr int kBatch = 32;
    __shared__ __attribute__((aligned(16))) T stage[kStage];
    T res = 0;
    bool scanned = false;
    if constexpr (SUM != 0) {
        for (long long i0 = 0; i0 < nsums; i0 += kStage) {
            const int cnt = (int)((nsums - i0) < kStage ? (nsums - i0) : kStage);
            for (int t = threadIdx.x; t < cnt; t += blockDim.x) stage[t] = piece_sums[i0 + t];
            __syncthreads();
            if (threadIdx.x == 0) {
                // the one serial chain of the algorithm (numpy adds the pieces in order): 32 values per batch in
                // registers, the next batch's LDS reads issued before the current batch's dependent adds
                T cur[kBatch], nxt[kBatch];
                int t = 0;
                if (cnt >= kBatch) {
#pragma unroll
                    for (int q = 0; q < kBatch; q++) cur[q] = stage[q];
                    for (; t + kBatch <= cnt; t += kBatch) {
                        const bool more = t + 2 * kBatch <= cnt;
                        if (more) {
#pragma unroll
                            for (int q = 0; q < kBatch; q++) nxt[q] = stage[t + kBatch + q];
                        }
#pragma unroll
                        for (int q = 0; q < kBatch; q++) res = res + cur[q];
                        if (more) {
#pragma unroll
                            for (int q = 0; q < kBatch; q++) cur[q] = nxt[q];
                        }
                    }
                }
                for (; t < cnt; t++) res = res + stage[t];
            }
            if (wave == 1 && !scanned) wave_pick_digit(h, k, lane, &pick);
            scanned = true;
            __syncthreads();
        }
    }
    if (!scanned && wave == 1) wave_pick_digit(h, k, lane, &pick);
    __syncthreads();
    if (threadIdx.x != 0) return;
    if constexpr (SUM == 1) st->tot = (double)res;
    if constexpr (SUM == 2) {
        st->s2 = (double)res;
        const T var = (T)((double)res / (double)m);         // ret.dtype.type(ret / rcount)
        st->sd = (double)(T)sqrt((double)var);              // sqrt of a T value rounded to T (exact for float32 via float64)
    }
    if constexpr (KIND == 0) {
        st->min_key = sh_lo;
        st->max_key = sh_hi;
    }
    st->prefix = KIND == 0 ? (unsigned long long)pick.digit : ((prefix << width) | (unsigned long long)pick.digit);
    st->k = pick.newk;
    if constexpr (KIND == 2) {
        st->med = median_of_pick<T>(m, prefix, width, pick, sh_hi);
        if (!final_pass) {
            if (m <= 0) { st->lo = st->hi = __builtin_nan(""); }
            else {
                // SigmaClip._compute_bounds: float32 scalars * python float -> float64 (numpy 1.26)
                st->lo = st->med - st->sd * sigma_lower;
                st->hi = st->med + st->sd * sigma_upper;
            }
        }
    }
}

// Last select level of a clipping iteration, closed inside the tile count: every workgroup redoes the small closing step
// (2048-bin histogram -> digit -> median -> bounds; 8 KB of L2-resident reads) instead of waiting for a one-workgroup
// kernel, then counts its tiles against the bounds.  Workgroup 0 also publishes median and bounds for the scatter;
// the tile scan clears the histogram.
template <typename T>
__global__ __launch_bounds__(kBlock) void close_count_kernel(const T *__restrict__ src0, const T *__restrict__ src1, GState *__restrict__ st,
                                                            int slot, int level, int ngroups, double sigma_lower, double sigma_upper,
                                                            unsigned *__restrict__ tile_counts)
{
    const IterState s = st->it[slot];
    if (s.done) return;
    const T *src = s.cur ? src1 : src0;
    const long long m = s.m;
    const int width = level_width<T>(level);
    const unsigned long long prefix = st->prefix;
    const long long k = st->k;
    const double sd = st->sd;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    __shared__ unsigned h[kBins];
    __shared__ unsigned long long sh_hi;
    __shared__ DigitPick pick;
    __shared__ double sh_bounds[2];
    for (int t = threadIdx.x; t < kBins; t += kBlock) h[t] = st->hist[t];
    if (threadIdx.x == 0) { sh_hi = 0; pick.digit = 0; pick.newk = 0; pick.dlow = -1; }
    __syncthreads();
    if (wave == 1) wave_pick_digit(h, k, lane, &pick);
    __syncthreads();
    // the largest key below the prefix is needed only when the searched rank opens its key AND its bin AND the count is even
    if ((m & 1) == 0 && pick.newk == 0 && pick.dlow < 0) {
        unsigned long long hi = 0;
        for (int t = threadIdx.x; t < ngroups; t += kBlock) {
            const unsigned long long u = st->wg_hi[t];
            hi = u > hi ? u : hi;
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            const unsigned long long h2 = __shfl_down(hi, d);
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0 && hi) atomicMax(&sh_hi, hi);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double med = median_of_pick<T>(m, prefix, width, pick, sh_hi);
        // SigmaClip._compute_bounds: float32 scalars * python float -> float64 (numpy 1.26)
        const double lo = m > 0 ? med - sd * sigma_lower : __builtin_nan(""), hi = m > 0 ? med + sd * sigma_upper : __builtin_nan("");
        sh_bounds[0] = lo;
        sh_bounds[1] = hi;
        if (blockIdx.x == 0) {
            st->med = med;                                  // outputs only: prefix / k / sd are still being read by
            st->lo = lo;                                    // workgroups that start later
            st->hi = hi;
        }
    }
    __syncthreads();
    // float32: the float64 bounds are demoted to float32 for the comparison, as numpy 1.26 does
    const T lof = (T)sh_bounds[0], hif = (T)sh_bounds[1];
    const long long ntiles = (m + kTile - 1) / kTile;
    __shared__ unsigned wsum[kBlock / kWave];
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T x[V]; };
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + (long long)threadIdx.x * 8;
        unsigned c = 0;
        if (base + 8 <= m) {                                // the survivors live in the 256-byte aligned workspace buffers
#pragma unroll
            for (int j = 0; j < 8 / V; j++) {
                const Vec v = reinterpret_cast<const Vec *>(src + base)[j];
#pragma unroll
                for (int q = 0; q < V; q++) c += keep_pred<1, T>(v.x[q], lof, hif) ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const long long i = base + j;
                if (i < m) c += keep_pred<1, T>(src[i], lof, hif) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) c += __shfl_down(c, d);
        if (lane == 0) wsum[wave] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

template <typename T>
__global__ void publish_kernel(const GState *st, int slot, double *out)
{
    using K = KeyOf<T>;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const IterState s = st->it[slot];
    const long long m = s.m;
    const double nan = __builtin_nan("");
    out[0] = m > 0 ? (double)(T)(st->tot / (double)m) : nan;        // np.mean: T(float64(sum) / n)
    out[1] = m > 0 ? st->med : nan;
    out[2] = m > 0 ? st->sd : nan;
    out[3] = st->lo;
    out[4] = st->hi;
    out[5] = (double)s.iter;
    out[6] = (double)m;
    out[7] = m > 0 ? (double)K::from(st->min_key) : nan;
    out[8] = m > 0 ? (double)K::from(st->max_key) : nan;
    out[9] = 0.0;
}

struct WsLayout {
    size_t state, buf0, buf1, pieces, tcounts, toffsets, total;
};

WsLayout layout(int64_t n, size_t elem)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    WsLayout L;
    size_t off = 0;
    L.state = off; off = up(off + sizeof(GState));
    L.buf0 = off; off = up(off + elem * (size_t)n);
    L.buf1 = off; off = up(off + elem * (size_t)n);
    L.pieces = off; off = up(off + elem * (size_t)(n / kPiece + 2));
    const size_t ntiles = (size_t)((n + kTile - 1) / kTile);
    L.tcounts = off; off = up(off + sizeof(unsigned) * (ntiles + 1));
    L.toffsets = off; off = up(off + sizeof(unsigned long long) * (ntiles + 1));
    L.total = off;
    return L;
}

template <typename T>
int run_sigclip_global(const T *data, int64_t n_pixels, double sigma_lower, double sigma_upper, int maxiters, double *stats_out,
                       void *ws, size_t ws_bytes, void *stream)
{
    if (!data || !stats_out || !ws) return fail(APGPU_EINVAL, "sigclip_global: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "sigclip_global: n_pixels = %lld", (long long)n_pixels);
    if (maxiters == 0) return fail(APGPU_EINVAL, "sigclip_global: maxiters must be >= 1 or < 0");
    const WsLayout L = layout(n_pixels, sizeof(T));
    if (ws_bytes < L.total) return fail(APGPU_EWORKSPACE, "sigclip_global: workspace %zu < %zu bytes", ws_bytes, L.total);
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail(APGPU_EINVAL, "sigclip_global: workspace must be 16-byte aligned");
    // Up to 32 iterations are queued without looking back (the `done` flag turns the ones after convergence into no-ops).
    // More than that - maxiters < 0 = until convergence, which a small sigma stretches to hundreds of passes - goes in
    // batches of 32 with one read of the flag in between (the only host synchronisation of this entry point).
    const long long iters_max = maxiters < 0 ? n_pixels + 1 : maxiters;       // every pass but the last removes a value
    char *w = static_cast<char *>(ws);
    GState *st = reinterpret_cast<GState *>(w + L.state);
    T *b0 = reinterpret_cast<T *>(w + L.buf0);
    T *b1 = reinterpret_cast<T *>(w + L.buf1);
    T *pieces = reinterpret_cast<T *>(w + L.pieces);
    unsigned *tcounts = reinterpret_cast<unsigned *>(w + L.tcounts);
    unsigned long long *toffs = reinterpret_cast<unsigned long long *>(w + L.toffsets);
    hipStream_t s = as_stream(stream);
    const long long n = n_pixels;
    const long long ntiles = (n + kTile - 1) / kTile;
    const unsigned gtile = (unsigned)(ntiles < kNumCU * 8 ? ntiles : kNumCU * 8);
    // statistics passes: one piece per wavefront, at most two workgroups per CU (each ends in up to 2048 global atomics)
    constexpr int wpb = kScanBlock / kWave;
    const long long wg_pieces = (n / kPiece + wpb - 1) / wpb;
    const unsigned gpass = (unsigned)(wg_pieces < 1 ? 1 : (wg_pieces < kNumCU * 2 ? wg_pieces : kNumCU * 2));
    constexpr int levels = key_levels<T>();

    if (hipMemsetAsync(st, 0, sizeof(GState), s) != hipSuccess) return fail(APGPU_ELAUNCH, "sigclip_global: memset failed");
    // finite values of data -> buffer 0
    hipLaunchKernelGGL((tile_count_kernel<0, T>), dim3(gtile), dim3(kBlock), 0, s, data, data, n, st, 0, tcounts);
    hipLaunchKernelGGL(tile_scan_kernel<0>, dim3(1), dim3(1024), 0, s, n, st, 0, tcounts, toffs);
    hipLaunchKernelGGL((tile_scatter_kernel<0, T>), dim3(gtile), dim3(kBlock), 0, s, data, data, b0, b0, n, st, 0, toffs);
    if (int rc = check_launch("sigclip_global: compact finite")) return rc;

    const unsigned gp1 = gpass + 1;                          // + the workgroup of the ragged piece
    // close + count: every workgroup repeats the closing step, so few, fat workgroups (16 tiles each at 4096^2)
    const unsigned gclose = (unsigned)(ntiles < kNumCU * 2 ? ntiles : kNumCU * 2);
    // close_last: the last level is closed by after_kernel (final statistics) or left to close_count_kernel (iterations)
    auto stats_pass = [&](int slot, int final_pass, bool close_last) {
        hipLaunchKernelGGL((pass_kernel<T, 1, 0>), dim3(gp1), dim3(kScanBlock), 0, s, b0, b1, st, pieces, slot, 0);
        hipLaunchKernelGGL((after_kernel<T, 1, 0>), dim3(1), dim3(kBlock), 0, s, st, pieces, slot, 0, final_pass, (int)gp1, sigma_lower,
                           sigma_upper);
        hipLaunchKernelGGL((pass_kernel<T, 2, 1>), dim3(gp1), dim3(kScanBlock), 0, s, b0, b1, st, pieces, slot, 1);
        hipLaunchKernelGGL((after_kernel<T, 2, 1>), dim3(1), dim3(kBlock), 0, s, st, pieces, slot, 1, final_pass, (int)gp1, sigma_lower,
                           sigma_upper);
        for (int level = 2; level < levels - 1; level++) {
            hipLaunchKernelGGL((pass_kernel<T, 0, 1>), dim3(gp1), dim3(kScanBlock), 0, s, b0, b1, st, pieces, slot, level);
            hipLaunchKernelGGL((after_kernel<T, 0, 1>), dim3(1), dim3(kBlock), 0, s, st, pieces, slot, level, final_pass, (int)gp1,
                               sigma_lower, sigma_upper);
        }
        hipLaunchKernelGGL((pass_kernel<T, 0, 2>), dim3(gp1), dim3(kScanBlock), 0, s, b0, b1, st, pieces, slot, levels - 1);
        if (close_last)
            hipLaunchKernelGGL((after_kernel<T, 0, 2>), dim3(1), dim3(kBlock), 0, s, st, pieces, slot, levels - 1, final_pass, (int)gp1,
                               sigma_lower, sigma_upper);
    };

    long long iters = 0;
    for (long long it = 0; it < iters_max; it++) {
        if (it > 0 && it % kMaxItersCap == 0) {
            int done = 0;
            if (hipMemcpyAsync(&done, &st->it[it & 1].done, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                return fail(APGPU_ELAUNCH, "sigclip_global: reading the convergence flag failed");
            if (done) break;
        }
        iters = it + 1;
        const int slot = (int)(it & 1);
        stats_pass(slot, 0, false);
        hipLaunchKernelGGL(close_count_kernel<T>, dim3(gclose), dim3(kBlock), 0, s, b0, b1, st, slot, levels - 1, (int)gp1, sigma_lower,
                           sigma_upper, tcounts);
        hipLaunchKernelGGL(tile_scan_kernel<1>, dim3(1), dim3(1024), 0, s, n, st, slot, tcounts, toffs);
        hipLaunchKernelGGL((tile_scatter_kernel<1, T>), dim3(gtile), dim3(kBlock), 0, s, b0, b1, b0, b1, n, st, slot, toffs);
        if (int rc = check_launch("sigclip_global: iteration")) return rc;
    }
    // statistics of the survivors: the last iteration's statistics belong to the pre-clip set unless it removed nothing
    // (state `done`), in which case they are final already and these launches return at once
    stats_pass((int)(iters & 1), 1, true);
    hipLaunchKernelGGL(publish_kernel<T>, dim3(1), dim3(64), 0, s, st, (int)(iters & 1), stats_out);
    return check_launch("sigclip_global: publish");
}

// F2 helper: float64 difference of two images where neither is flagged bad, NaN elsewhere (the NaNs are
// dropped, in order, by the finite-value compaction of the global statistics pass).
template <typename RawT>
__global__ __launch_bounds__(kBlock) void image_difference_kernel(const RawT *__restrict__ a, const RawT *__restrict__ b,
                                                                 const uint8_t *__restrict__ bad1, const uint8_t *__restrict__ bad2,
                                                                 double *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool bad = (bad1 && bad1[i]) || (bad2 && bad2[i]);
        out[i] = bad ? __builtin_nan("") : (double)a[i] - (double)b[i];
    }
}

}  // namespace

extern "C" size_t apgpu_sigclip_global_ws_bytes(int64_t n_pixels)
{
    if (n_pixels <= 0) return 0;
    return layout(n_pixels, sizeof(float)).total;
}

extern "C" size_t apgpu_sigclip_global_f64_ws_bytes(int64_t n_pixels)
{
    if (n_pixels <= 0) return 0;
    return layout(n_pixels, sizeof(double)).total;
}

extern "C" int apgpu_sigclip_global_f32(const float *data, int64_t n_pixels, double sigma_lower, double sigma_upper, int maxiters,
                                        double *stats_out, void *ws, size_t ws_bytes, void *stream)
{
    return run_sigclip_global<float>(data, n_pixels, sigma_lower, sigma_upper, maxiters, stats_out, ws, ws_bytes, stream);
}

extern "C" int apgpu_sigclip_global_f64(const double *data, int64_t n_pixels, double sigma_lower, double sigma_upper, int maxiters,
                                        double *stats_out, void *ws, size_t ws_bytes, void *stream)
{
    return run_sigclip_global<double>(data, n_pixels, sigma_lower, sigma_upper, maxiters, stats_out, ws, ws_bytes, stream);
}

extern "C" int apgpu_image_difference_f64(const void *a, const void *b, int dtype, const uint8_t *bad1, const uint8_t *bad2,
                                          double *out, int64_t n_pixels, void *stream)
{
    if (!a || !b || !out) return fail(APGPU_EINVAL, "image_difference: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "image_difference: n_pixels = %lld", (long long)n_pixels);
    int64_t grid = (n_pixels + kBlock - 1) / kBlock;
    if (grid > kNumCU * 8) grid = kNumCU * 8;
    hipStream_t s = as_stream(stream);
    if (dtype == APGPU_F32)
        hipLaunchKernelGGL(image_difference_kernel<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, (const float *)a, (const float *)b,
                           bad1, bad2, out, n_pixels);
    else if (dtype == APGPU_U16)
        hipLaunchKernelGGL(image_difference_kernel<uint16_t>, dim3((unsigned)grid), dim3(kBlock), 0, s, (const uint16_t *)a,
                           (const uint16_t *)b, bad1, bad2, out, n_pixels);
    else
        return fail(APGPU_EINVAL, "image_difference: bad dtype %d", dtype);
    return check_launch("image_difference");
}
