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
//     np.median   = exact order statistic (radix select on order-preserving keys, 4 x 8-bit passes);
//                   even count -> float32(a + b) / 2
//     np.var      = mean = sum / float32(n); float32 sum of (x - mean)^2; float32(float64(sum) / n)
//     bounds      = float64 (np.float32 * python float promotes to float64 under numpy 1.26), demoted
//                   to float32 when compared with the float32 array
// The surviving values are kept compacted IN ORDER in a ping-pong pair of workspace buffers because
// the summation tree depends on element positions.
//
// Everything runs on the stream without host synchronisation: the iteration count is a launch-time
// constant and a device-side `done` flag turns the remaining iterations into no-ops.
// Traffic: ~9 reads + 1 write of the surviving values per iteration (67 MB at 4096^2 - resident in the
// 256 MB Infinity Cache after the first pass).
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kBlock = 256;
constexpr int kTile = 2048;            // elements per compaction tile (8 per thread)
constexpr int kPiece = 8192;           // numpy reduction buffer
constexpr int kLeaf = 128;             // numpy PW_BLOCKSIZE
constexpr int kMaxItersCap = 32;

// T = float : float32 data, numpy float32 statistics (float32 darks / flats)
// T = double: float64 data and statistics - what numpy computes for INTEGER images (np.median/np.var of
//             a uint16 array run in float64; the host widens integer images exactly) and for float64 data.
struct GState {
    long long m;                // survivors in the current buffer
    long long m_next;
    long long k;                // rank searched by the radix select
    unsigned long long cnt_less;
    unsigned long long prefix;          // key prefix found so far
    unsigned long long max_less_key;    // order-preserving key of max{x < v_hi}
    unsigned long long min_key, max_key;    // extremes of the survivors (final pass)
    int done;                   // set when an iteration removed nothing
    int iter;                   // iterations executed
    int cur;                    // index (0/1) of the buffer holding the survivors
    int pad;
    double tot;                 // np.sum of the survivors          (holds a float32 value when T = float)
    double s2;                  // np.sum((x - mean)^2)
    double med, sd;
    double lo, hi;
    unsigned hist[256];
};

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    using type = unsigned;
    static constexpr int passes = 4;
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
    static constexpr int passes = 8;
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

template <typename T> __device__ __forceinline__ bool is_finite(T x) { return fabs((double)x) < __builtin_inf(); }

// ---- order-preserving compaction: tile counts -> scan -> scatter -------------------------------
// mode 0: keep finite values of `data` (length n, host-known); mode 1: keep lo <= x <= hi of the
// current survivors (length st->m).
template <int MODE, typename T>
__device__ __forceinline__ bool keep_pred(T x, T lof, T hif)
{
    if constexpr (MODE == 0) return is_finite<T>(x);
    else return (x >= lof) && (x <= hif);
}

template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void tile_count_kernel(const T *__restrict__ src0, const T *__restrict__ src1,
                                                           long long n_static, const GState *__restrict__ st,
                                                           unsigned *__restrict__ tile_counts)
{
    if (MODE == 1 && st->done) return;
    const T *src = (MODE == 0) ? src0 : (st->cur ? src1 : src0);
    const long long m = (MODE == 0) ? n_static : st->m;
    // float32: the float64 bounds are demoted to float32 for the comparison, as numpy 1.26 does
    const T lof = (T)st->lo, hif = (T)st->hi;
    const long long ntiles = (m + kTile - 1) / kTile;
    __shared__ unsigned wsum[kBlock / kWave];
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + (long long)threadIdx.x * 8;
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const long long i = base + j;
            if (i < m) c += keep_pred<MODE, T>(src[i], lof, hif) ? 1u : 0u;
        }
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) c += __shfl_down(c, d);
        if ((threadIdx.x % kWave) == 0) wsum[threadIdx.x / kWave] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void tile_scan_kernel(long long n_static, GState *__restrict__ st,
                                                        const unsigned *__restrict__ tile_counts,
                                                        unsigned long long *__restrict__ tile_offsets)
{
    if (MODE == 1 && st->done) return;
    const long long m = (MODE == 0) ? n_static : st->m;
    const long long ntiles = (m + kTile - 1) / kTile;
    const long long per = (ntiles + 1023) / 1024;
    const long long t0 = (long long)threadIdx.x * per;
    const long long t1 = t0 + per < ntiles ? t0 + per : ntiles;
    unsigned long long local = 0;
    for (long long t = t0; t < t1; t++) local += tile_counts[t];
    __shared__ unsigned long long sums[1024];
    sums[threadIdx.x] = local;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned long long v = threadIdx.x >= d ? sums[threadIdx.x - d] : 0ull;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = sums[threadIdx.x] - local;
    for (long long t = t0; t < t1; t++) {
        tile_offsets[t] = run;
        run += tile_counts[t];
    }
    if (threadIdx.x == 1023) st->m_next = (long long)sums[1023];
}

template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void tile_scatter_kernel(const T *__restrict__ src0, const T *__restrict__ src1,
                                                             T *__restrict__ dst0, T *__restrict__ dst1,
                                                             long long n_static, const GState *__restrict__ st,
                                                             const unsigned long long *__restrict__ tile_offsets)
{
    if (MODE == 1 && st->done) return;
    const T *src = (MODE == 0) ? src0 : (st->cur ? src1 : src0);
    T *dst = (MODE == 0) ? dst0 : (st->cur ? dst0 : dst1);
    const long long m = (MODE == 0) ? n_static : st->m;
    const T lof = (T)st->lo, hif = (T)st->hi;
    const long long ntiles = (m + kTile - 1) / kTile;
    __shared__ unsigned wsum[kBlock / kWave];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + (long long)threadIdx.x * 8;
        T x[8];
        bool kp[8];
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const long long i = base + j;
            x[j] = i < m ? src[i] : (T)0;
            kp[j] = (i < m) && keep_pred<MODE, T>(x[j], lof, hif);
            c += kp[j] ? 1u : 0u;
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
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (kp[j]) dst[pos++] = x[j];
        __syncthreads();
    }
}

// commit a compaction: MODE 0 initialises the state, MODE 1 closes a clipping iteration
template <int MODE>
__global__ void commit_kernel(GState *st)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (MODE == 0) {
        st->m = st->m_next;
        st->cur = 0;
        st->done = 0;
        st->iter = 0;
        st->lo = __builtin_nan("");
        st->hi = __builtin_nan("");
        for (int i = 0; i < 256; i++) st->hist[i] = 0;
    } else {
        if (st->done) return;
        st->iter += 1;
        if (st->m_next == st->m) st->done = 1;      // nothing removed: survivors stay in buffer `cur`
        else { st->m = st->m_next; st->cur ^= 1; }
    }
}

// ---- exact median: radix select ------------------------------------------------------------------
__global__ void select_begin_kernel(GState *st, int final_pass)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!final_pass && st->done) return;
    st->k = st->m / 2;              // upper median rank (0-based); odd m: the median itself
    st->prefix = 0;
    st->cnt_less = 0;
    st->max_less_key = 0;
    st->min_key = ~0ull;
    st->max_key = 0;
}

// Visits src[0 .. m) with 16-byte loads where the pointer allows it (grid-stride over vectors, scalar tail).
template <typename T, typename F>
__device__ __forceinline__ void for_each_value(const T *__restrict__ src, long long m, F f)
{
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T x[V]; };
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    long long done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const long long nvec = m / V;
        constexpr int U = 4;                                // 4 x 16 bytes in flight per lane: these scans run with few,
        long long i = tid;                                  // fat workgroups and are latency-bound otherwise
        for (; i + (U - 1) * nthreads < nvec; i += U * nthreads) {
            Vec v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = reinterpret_cast<const Vec *>(src)[i + u * nthreads];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < V; k++) f(v[u].x[k]);
        }
        for (; i < nvec; i += nthreads) {
            const Vec v = reinterpret_cast<const Vec *>(src)[i];
#pragma unroll
            for (int k = 0; k < V; k++) f(v.x[k]);
        }
        done = nvec * V;
    }
    for (long long i = done + tid; i < m; i += nthreads) f(src[i]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void hist_kernel(const T *__restrict__ b0, const T *__restrict__ b1,
                                                     GState *__restrict__ st, int pass, int final_pass)
{
    using K = KeyOf<T>;
    if (!final_pass && st->done) return;
    const T *src = st->cur ? b1 : b0;
    const long long m = st->m;
    const int shift = 8 * (K::passes - 1 - pass);
    const typename K::type prefix = (typename K::type)st->prefix;
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for_each_value<T>(src, m, [&](T x) {
        const typename K::type key = K::to(x);
        const bool match = (pass == 0) || ((key >> (shift + 8)) == prefix);
        const unsigned d = (unsigned)(key >> shift) & 0xffu;
        // A dark frame's values share their leading digits: nearly every lane of a wave hits the same bin and
        // the LDS atomics serialise.  Up to 4 rounds of "leader adds the population count of its digit",
        // then plain atomics for whatever is left (the spread-out low digits).
        unsigned long long todo = __ballot(match);
        for (int round = 0; round < 4 && todo; round++) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned dl = (unsigned)__shfl((int)d, leader);
            const unsigned long long same = __ballot(match && d == dl) & todo;
            if ((int)(threadIdx.x % kWave) == leader) atomicAdd(&h[dl], (unsigned)__popcll(same));
            todo &= ~same;
        }
        if ((todo >> (threadIdx.x % kWave)) & 1ull) atomicAdd(&h[d], 1u);
    });
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void select_digit_kernel(GState *st, int final_pass)
{
    // one lane per bin: exclusive prefix over the 256 counts in LDS, the lane whose interval holds rank k wins
    if (blockIdx.x != 0) return;
    if (!final_pass && st->done) return;
    __shared__ long long cum[257];
    __shared__ int digit;
    const int t = threadIdx.x;
    const long long c = st->hist[t];
    cum[t + 1] = c;
    if (t == 0) { cum[0] = 0; digit = 255; }
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                     // Hillis-Steele inclusive scan over cum[1..256]
        const long long add = (t + 1 - d >= 1) ? cum[t + 1 - d] : 0;
        __syncthreads();
        cum[t + 1] += add;
        __syncthreads();
    }
    const long long k = st->k;
    if (k >= cum[t] && k < cum[t + 1]) digit = t;           // at most one lane (intervals are disjoint)
    st->hist[t] = 0;
    __syncthreads();
    if (t == 0) {
        const int d = digit;
        st->k = k - cum[d];
        st->prefix = (st->prefix << 8) | (unsigned long long)d;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void less_stats_kernel(const T *__restrict__ b0, const T *__restrict__ b1,
                                                           GState *__restrict__ st, int final_pass)
{
    using K = KeyOf<T>;
    if (!final_pass && st->done) return;
    const T *src = st->cur ? b1 : b0;
    const long long m = st->m;
    const unsigned long long vkey = st->prefix;
    unsigned cnt = 0;
    unsigned long long mx = 0, lo = ~0ull, hi = 0;
    for_each_value<T>(src, m, [&](T x) {
        const unsigned long long key = K::to(x);
        if (key < vkey) { cnt++; mx = key > mx ? key : mx; }
        lo = key < lo ? key : lo;
        hi = key > hi ? key : hi;
    });
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        cnt += __shfl_down(cnt, d);
        const unsigned long long o = __shfl_down(mx, d);
        mx = o > mx ? o : mx;
        const unsigned long long l2 = __shfl_down(lo, d), h2 = __shfl_down(hi, d);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    // block-level combine in LDS: one set of global atomics per workgroup, not per wave
    __shared__ unsigned s_cnt[kBlock / kWave];
    __shared__ unsigned long long s_mx[kBlock / kWave], s_lo[kBlock / kWave], s_hi[kBlock / kWave];
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x % kWave) == 0) {
        s_cnt[wave] = cnt;
        s_mx[wave] = mx;
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0;
        for (int w = 0; w < kBlock / kWave; w++) {
            c += s_cnt[w];
            mx = s_mx[w] > mx ? s_mx[w] : mx;
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        if (c) {
            atomicAdd(&st->cnt_less, c);
            atomicMax(&st->max_less_key, mx);
        }
        if (final_pass && hi >= lo) {
            atomicMin(&st->min_key, lo);
            atomicMax(&st->max_key, hi);
        }
    }
}

template <typename T>
__global__ void median_finish_kernel(GState *st, int final_pass)
{
    using K = KeyOf<T>;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!final_pass && st->done) return;
    const long long m = st->m;
    if (m <= 0) { st->med = __builtin_nan(""); return; }
    const T vhi = K::from(st->prefix);
    if (m & 1) { st->med = (double)vhi; return; }
    const long long k2 = m / 2;
    // rank k2-1 holds vhi again if fewer than k2 values are strictly below vhi
    const T vlo = ((long long)st->cnt_less <= k2 - 1) ? vhi : K::from(st->max_less_key);
    const T t = vlo + vhi;                                  // np.mean of the two middle values, in T
    st->med = (double)(T)((double)t / 2.0);
}

// ---- numpy float32 pairwise sums --------------------------------------------------------------------
template <int SQ, typename T>
__device__ __forceinline__ T tr(T x, T mean)
{
    if constexpr (SQ) { const T d = x - mean; return d * d; }
    else return x;
}

template <int SQ, typename T>
__device__ T leaf_sum(const T *a, int n, T mean)
{
    T r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = tr<SQ, T>(a[k], mean);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = r[k] + tr<SQ, T>(a[i + k], mean);
    }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + tr<SQ, T>(a[i], mean);
    return res;
}

// numpy's recursion for a ragged piece (n < 8192; depth <= 7), split in two walks so that the leaves -
// which are independent - can be summed by different lanes: enumerate the leaves, sum them in parallel,
// then combine the leaf sums in the recursion's order.
struct LeafList {
    int off[256], len[256];
    int n;
};

__device__ void enumerate_leaves(LeafList &ll, int off, int n)
{
    if (n < 8 || n <= kLeaf) {
        ll.off[ll.n] = off;
        ll.len[ll.n] = n;
        ll.n++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    enumerate_leaves(ll, off, n2);
    enumerate_leaves(ll, off + n2, n - n2);
}

template <typename T>
__device__ T combine_leaves(const T *vals, int &next, int n)
{
    if (n < 8 || n <= kLeaf) return vals[next++];
    int n2 = n / 2;
    n2 -= n2 % 8;
    const T l = combine_leaves<T>(vals, next, n2);
    const T r = combine_leaves<T>(vals, next, n - n2);
    return l + r;
}

template <int SQ, typename T>
__device__ T one_leaf(const T *a, int n, T mean)
{
    if (n < 8) {
        T res = 0;
        for (int i = 0; i < n; i++) res = res + tr<SQ, T>(a[i], mean);
        return res;
    }
    return leaf_sum<SQ, T>(a, n, mean);
}

template <typename T>
__device__ __forceinline__ T var_mean(const GState *st)
{
    return (T)st->tot / (T)st->m;           // np.var: arrmean = true_divide(sum, n) in the array's dtype
}

template <int SQ, typename T>
__global__ __launch_bounds__(kBlock) void piece_sums_kernel(const T *__restrict__ b0, const T *__restrict__ b1,
                                                           const GState *__restrict__ st, T *__restrict__ piece_sums,
                                                           int final_pass)
{
    if (!final_pass && st->done) return;
    const T *src = st->cur ? b1 : b0;
    const long long m = st->m;
    const T mean = SQ ? var_mean<T>(st) : (T)0;
    const long long npieces_full = m / kPiece;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int wpb = kBlock / kWave;
    for (long long piece = (long long)blockIdx.x * wpb + wave; piece < npieces_full; piece += (long long)gridDim.x * wpb) {
        T s = leaf_sum<SQ, T>(src + piece * kPiece + lane * kLeaf, kLeaf, mean);
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const T other = __shfl_xor(s, d);
            s = (lane & d) ? other + s : s + other;
        }
        if (lane == 0) piece_sums[piece] = s;
    }
}

template <int SQ, typename T>
__global__ void fold_kernel(const T *__restrict__ b0, const T *__restrict__ b1, GState *__restrict__ st,
                            const T *__restrict__ piece_sums, int final_pass)
{
    // one workgroup: the piece sums are accumulated sequentially (numpy's order), but fetched by all
    // lanes into LDS first - a lone lane walking global memory pays a full miss latency per piece
    if (blockIdx.x != 0) return;
    if (!final_pass && st->done) return;
    const T *src = st->cur ? b1 : b0;
    const long long m = st->m;
    const T mean = SQ ? var_mean<T>(st) : (T)0;
    const long long npieces_full = m / kPiece;
    constexpr int kStage = 2048;
    __shared__ T stage[kStage];
    T res = 0;
    for (long long i0 = 0; i0 < npieces_full; i0 += kStage) {
        const int cnt = (int)((npieces_full - i0) < kStage ? (npieces_full - i0) : kStage);
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) stage[t] = piece_sums[i0 + t];
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (; t + 8 <= cnt; t += 8) {                  // 8 LDS reads in flight, then the 8 ordered adds
                T x[8];
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = stage[t + k];
#pragma unroll
                for (int k = 0; k < 8; k++) res = res + x[k];
            }
            for (; t < cnt; t++) res = res + stage[t];
        }
        __syncthreads();
    }
    const int rem = (int)(m - npieces_full * kPiece);
    if (rem > 0) {
        __shared__ LeafList ll;
        __shared__ T leaf_vals[256];
        if (threadIdx.x == 0) {
            ll.n = 0;
            enumerate_leaves(ll, 0, rem);
        }
        __syncthreads();
        // the ragged piece goes through LDS as well: a leaf is up to 128 elements walked by one lane
        __shared__ T ragged[kPiece];
        for (int t = threadIdx.x; t < rem; t += blockDim.x) ragged[t] = src[npieces_full * kPiece + t];
        __syncthreads();
        const T *tail = ragged;
        for (int l = threadIdx.x; l < ll.n; l += blockDim.x) leaf_vals[l] = one_leaf<SQ, T>(tail + ll.off[l], ll.len[l], mean);
        __syncthreads();
        if (threadIdx.x == 0) {
            int next = 0;
            res = res + combine_leaves<T>(leaf_vals, next, rem);
        }
    }
    if (threadIdx.x != 0) return;
    if (SQ) {
        st->s2 = (double)res;
        const T var = (T)((double)res / (double)m);         // ret.dtype.type(ret / rcount)
        st->sd = (double)(T)sqrt((double)var);              // sqrt of a T value rounded to T (exact for float32 via float64)
    } else {
        st->tot = (double)res;
    }
}

__global__ void bounds_kernel(GState *st, double sigma_lower, double sigma_upper)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->done) return;
    if (st->m <= 0) { st->lo = st->hi = __builtin_nan(""); return; }
    // SigmaClip._compute_bounds: float32 scalars * python float -> float64 (numpy 1.26)
    st->lo = st->med - st->sd * sigma_lower;
    st->hi = st->med + st->sd * sigma_upper;
}

template <typename T>
__global__ void publish_kernel(const GState *st, double *out)
{
    using K = KeyOf<T>;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long m = st->m;
    const double nan = __builtin_nan("");
    out[0] = m > 0 ? (double)(T)(st->tot / (double)m) : nan;        // np.mean: T(float64(sum) / n)
    out[1] = m > 0 ? st->med : nan;
    out[2] = m > 0 ? st->sd : nan;
    out[3] = st->lo;
    out[4] = st->hi;
    out[5] = (double)st->iter;
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
    L.pieces = off; off = up(off + elem * (size_t)(n / kPiece + 1));
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
    using K = KeyOf<T>;
    if (!data || !stats_out || !ws) return fail(APGPU_EINVAL, "sigclip_global: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "sigclip_global: n_pixels = %lld", (long long)n_pixels);
    if (maxiters == 0) return fail(APGPU_EINVAL, "sigclip_global: maxiters must be >= 1 or < 0");
    const WsLayout L = layout(n_pixels, sizeof(T));
    if (ws_bytes < L.total) return fail(APGPU_EWORKSPACE, "sigclip_global: workspace %zu < %zu bytes", ws_bytes, L.total);
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail(APGPU_EINVAL, "sigclip_global: workspace must be 16-byte aligned");
    // maxiters < 0 (until convergence): the device loop is a launch-time constant; 32 passes always
    // converge in practice (every pass but the last removes at least one value).
    const int iters = (maxiters < 0 || maxiters > kMaxItersCap) ? kMaxItersCap : maxiters;
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
    const unsigned gflat = (unsigned)((n + kBlock - 1) / kBlock < kNumCU * 8 ? (n + kBlock - 1) / kBlock : kNumCU * 8);
    // histogram / rank scans end in global atomics per workgroup: fewer, fatter workgroups (16-byte loads)
    const unsigned gscan = gflat < kNumCU * 2 ? gflat : kNumCU * 2;
    const unsigned gpiece = (unsigned)((n / kPiece) / (kBlock / kWave) + 1 < kNumCU * 8 ? (n / kPiece) / (kBlock / kWave) + 1 : kNumCU * 8);

    if (hipMemsetAsync(st, 0, sizeof(GState), s) != hipSuccess) return fail(APGPU_ELAUNCH, "sigclip_global: memset failed");
    // finite values of data -> buffer 0
    hipLaunchKernelGGL((tile_count_kernel<0, T>), dim3(gtile), dim3(kBlock), 0, s, data, data, n, st, tcounts);
    hipLaunchKernelGGL(tile_scan_kernel<0>, dim3(1), dim3(1024), 0, s, n, st, tcounts, toffs);
    hipLaunchKernelGGL((tile_scatter_kernel<0, T>), dim3(gtile), dim3(kBlock), 0, s, data, data, b0, b0, n, st, toffs);
    hipLaunchKernelGGL(commit_kernel<0>, dim3(1), dim3(64), 0, s, st);
    if (int rc = check_launch("sigclip_global: compact finite")) return rc;

    auto stats_pass = [&](int final_pass) {
        hipLaunchKernelGGL(select_begin_kernel, dim3(1), dim3(64), 0, s, st, final_pass);
        for (int pass = 0; pass < K::passes; pass++) {
            hipLaunchKernelGGL(hist_kernel<T>, dim3(gscan), dim3(kBlock), 0, s, b0, b1, st, pass, final_pass);
            hipLaunchKernelGGL(select_digit_kernel, dim3(1), dim3(256), 0, s, st, final_pass);
        }
        hipLaunchKernelGGL(less_stats_kernel<T>, dim3(gscan), dim3(kBlock), 0, s, b0, b1, st, final_pass);
        hipLaunchKernelGGL(median_finish_kernel<T>, dim3(1), dim3(64), 0, s, st, final_pass);
        hipLaunchKernelGGL((piece_sums_kernel<0, T>), dim3(gpiece), dim3(kBlock), 0, s, b0, b1, st, pieces, final_pass);
        hipLaunchKernelGGL((fold_kernel<0, T>), dim3(1), dim3(kBlock), 0, s, b0, b1, st, pieces, final_pass);
        hipLaunchKernelGGL((piece_sums_kernel<1, T>), dim3(gpiece), dim3(kBlock), 0, s, b0, b1, st, pieces, final_pass);
        hipLaunchKernelGGL((fold_kernel<1, T>), dim3(1), dim3(kBlock), 0, s, b0, b1, st, pieces, final_pass);
    };

    for (int it = 0; it < iters; it++) {
        stats_pass(0);
        hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(64), 0, s, st, sigma_lower, sigma_upper);
        hipLaunchKernelGGL((tile_count_kernel<1, T>), dim3(gtile), dim3(kBlock), 0, s, b0, b1, n, st, tcounts);
        hipLaunchKernelGGL(tile_scan_kernel<1>, dim3(1), dim3(1024), 0, s, n, st, tcounts, toffs);
        hipLaunchKernelGGL((tile_scatter_kernel<1, T>), dim3(gtile), dim3(kBlock), 0, s, b0, b1, b0, b1, n, st, toffs);
        hipLaunchKernelGGL(commit_kernel<1>, dim3(1), dim3(64), 0, s, st);
        if (int rc = check_launch("sigclip_global: iteration")) return rc;
    }
    // statistics of the survivors (the last iteration's statistics belong to the pre-clip set unless
    // it removed nothing)
    stats_pass(1);
    hipLaunchKernelGGL(publish_kernel<T>, dim3(1), dim3(64), 0, s, st, stats_out);
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
