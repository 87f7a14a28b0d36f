// elementwise.hip - the streaming (one pass, HBM-bound) kernels of the calibrate path on gfx950:
//   A1 flat normalisation     ApCalibrate._generate_flat            core/ApCalibrate.py:166-190
//   A2 calibrate              ApCalibrate.calibrate                 core/ApCalibrate.py:439-464
//   A4 threshold mask         ApFindBadPixels._generate_sigmaclip_mask  core/ApFindBadPixels.py:194-216
//   A4 user overlays          ApFindBadPixels._add_bad_*            core/ApFindBadPixels.py:70-158
//   A8 image arithmetic       ApImArith.process_files               core/ApImArith.py:320-333
//   A9 Bayer split            RawConv._build_raw_channel_images     core/RawConv.py:111-128
// All float arithmetic is separately rounded IEEE float32 (built with -ffp-contract=off, __fdiv_rn)
// because the reference's NumPy expressions round after every ufunc.
//
// Access pattern: 16 bytes per lane per load (float4 / 8 x u16), grid-stride over 2048 workgroups so
// every CU holds several waves with independent 1 KiB wave-loads in flight.
#include "common.h"

namespace {
using namespace apgpu;

constexpr int kBlock = 256;
constexpr int kMaxGrid = kNumCU * 8;

inline unsigned grid_for(int64_t work_items)
{
    int64_t g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------------
// A2 calibrate.  One (frame, 4-pixel group) per lane-iteration; masters are re-read per frame from
// L2/Infinity Cache (3 x 64 MB at 4096^2 stay resident in the 256 MB Infinity Cache).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float calib_one(float x, float ped, bool has_ped, float b, float D, float e,
                                           bool has_flat, float nf)
{
    if (has_ped) x = x + ped;                 // ApCalibrate.py:318-326 (PEDESTAL is added at read time)
    x = x - b;                                // :439
    const float ds = e * D;                   // :450  float32(exp_ratio) * dark
    x = x - ds;                               // :451
    if (has_flat && nf != 0.f) x = __fdiv_rn(x, nf);   // :462-464 (NaN != 0 -> divide -> NaN)
    return x;
}

// x / nf through y = RN(1 / nf) and two FMA-residual corrections (exact when nothing over/underflows,
// see stack_kernels.h div_by_recip); anything outside the guarded ranges takes the IEEE sequence.
__device__ __forceinline__ float guarded_div(float x, float nf, float y, bool nf_ok)
{
    const float q0 = x * y;
    const float r0 = __builtin_fmaf(-nf, q0, x);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-nf, q1, x);
    const float q = __builtin_fmaf(r1, y, q1);
    const float aq = fabsf(q);
    const bool safe = nf_ok && (aq < 0x1p50f) && (aq > 0x1p-50f);     // false for NaN / Inf / 0 as well
    return safe ? q : __fdiv_rn(x, nf);
}

// One lane owns 4 consecutive pixels and walks over the N frames: the masters are read once per pixel
// (not once per frame), the reciprocal of the flat is formed once, and every access is a 16-byte (f32)
// or 8-byte (u16) coalesced vector.  vec = 0 (P % 4 != 0: frame starts are not 16-byte aligned) sends
// every pixel through the scalar tail.
template <typename RawT>
__global__ __launch_bounds__(kBlock) void calibrate_kernel(const RawT *__restrict__ raw, const float *__restrict__ bias,
                                                          const float *__restrict__ dark,
                                                          const float *__restrict__ nflat,
                                                          const float *__restrict__ exp_ratio,
                                                          const float *__restrict__ pedestal, int still_biased,
                                                          float *__restrict__ out, int64_t N, int64_t P, int vec)
{
    const int64_t groups = vec ? P / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool has_flat = nflat != nullptr;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const float4 b4 = reinterpret_cast<const float4 *>(bias)[g];
        const float4 d4 = reinterpret_cast<const float4 *>(dark)[g];
        float4 n4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (has_flat) n4 = reinterpret_cast<const float4 *>(nflat)[g];
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
        const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
        float D[4], y[4];
        bool dodiv[4], nf_ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            D[k] = still_biased ? dd[k] - bb[k] : dd[k];               // ApCalibrate.py:440-445
            dodiv[k] = has_flat && (nn[k] != 0.f);                     // :462 (NaN != 0 -> divide -> NaN)
            y[k] = __fdiv_rn(1.0f, nn[k]);
            const float a = fabsf(nn[k]);
            nf_ok[k] = (a >= 0x1p-40f) && (a <= 0x1p40f);
        }
        // kU frames per trip: all kU loads are issued before the first result is needed, which keeps
        // kU x 16 bytes per lane in flight (the loop is otherwise one outstanding load per lane).
        constexpr int kU = 16;                             // 16 x 16 B per lane in flight: 1.91 -> 1.78 ms against 8 on C2
        for (int64_t f0 = 0; f0 < N; f0 += kU) {
            float x[kU][4];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int64_t f = f0 + u < N ? f0 + u : N - 1;         // clamped reload, result discarded below
                const RawT *rp = raw + f * P + g * 4;
                if constexpr (sizeof(RawT) == 4) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(rp);
                    x[u][0] = r4.x; x[u][1] = r4.y; x[u][2] = r4.z; x[u][3] = r4.w;
                } else {
                    const ushort4 r4 = *reinterpret_cast<const ushort4 *>(rp);
                    x[u][0] = (float)r4.x; x[u][1] = (float)r4.y; x[u][2] = (float)r4.z; x[u][3] = (float)r4.w;
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int64_t f = f0 + u;
                if (f >= N) break;
                const float e = exp_ratio[f];
                const float ped = pedestal ? pedestal[f] : 0.f;
                const bool has_ped = ped != 0.f;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v = x[u][k];
                    if (has_ped) v = v + ped;                           // ApCalibrate.py:318-326
                    v = v - bb[k];                                      // :439
                    const float ds = e * D[k];                          // :450
                    v = v - ds;                                         // :451
                    o[k] = dodiv[k] ? guarded_div(v, nn[k], y[k], nf_ok[k]) : v;
                }
                float *op = out + f * P + g * 4;                        // streamed once, never re-read: bypass L2 retention
#pragma unroll
                for (int k = 0; k < 4; k++) __builtin_nontemporal_store(o[k], op + k);
            }
        }
    }
    // tail pixels (P % 4, or every pixel when vec == 0)
    const int64_t tail0 = groups * 4;
    const int64_t ntail = P - tail0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * ntail; t += stride) {
        const int64_t f = t / ntail;
        const int64_t p = tail0 + (t - f * ntail);
        const float e = exp_ratio[f];
        const float ped = pedestal ? pedestal[f] : 0.f;
        const float b = bias[p], d = dark[p];
        const float D = still_biased ? d - b : d;
        const float nf = has_flat ? nflat[p] : 1.f;
        out[f * P + p] = calib_one((float)raw[f * P + p], ped, ped != 0.f, b, D, e, has_flat, nf);
    }
}

// ------------------------------------------------------------------------------------------------
// A1 flat normalisation.  np.nanmean(float32) = numpy's pairwise float32 sum taken over 8192-element
// buffer pieces that are accumulated sequentially (verified against numpy 1.26.4 / 2.2.6, golden G7):
//   piece sum  : binary tree over 64 leaves of 128 elements; a leaf keeps 8 strided accumulators
//                r[k] += a[8j+k] and reduces ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
//   total      : ((0 + piece0) + piece1) + ...   in float32;  norm = f32(f64(total) / count)
// Kernel 1: one wavefront per full piece, one lane per leaf, in-wave tree with the same pairing as
//           numpy's recursion (halves of equal size).  NaNs count as 0 (np.nanmean) and are counted.
// Kernel 2: one thread folds the piece sums in order, handles the ragged last piece with the
//           general recursion, and writes norm.   Kernel 3: nflat = flat / norm.
// ------------------------------------------------------------------------------------------------
constexpr int kPiece = 8192;
constexpr int kLeaf = 128;

// The same kernels serve float32 and float64 flats (a ccdproc-written master flat is float64 and the reference then
// normalises it in float64, core/ApCalibrate.py:301-305, 181-188; golden G11): T is the array's own type.
template <typename T>
__device__ __forceinline__ T nan0(T x, int &nans)
{
    if (x != x) { nans++; return (T)0; }
    return x;
}

template <typename T>
__device__ T leaf_sum(const T *a, int n, int &nans)
{
    // numpy pairwise_sum for 8 <= n <= 128
    T r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = nan0(a[k], nans);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = r[k] + nan0(a[i + k], nans);
    }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + nan0(a[i], nans);
    return res;
}

struct PairwiseItem { int off, len; };

// (one thread; its explicit stack lives in the caller's LDS: as local arrays with run-time indices it was 272 / 528 bytes of
// scratch per thread of the kernel - round 5)
template <typename T>
__device__ T pairwise_generic(const T *a, int n, int &nans, PairwiseItem *stack, T *vals, int *state)
{
    // iterative form of numpy's recursion for the ragged last piece (n < 8192): explicit stack
    typedef PairwiseItem Item;
    int sp = 0, vp = 0;
    stack[sp] = {0, n}; state[sp] = 0; sp++;
    // post-order evaluation
    while (sp > 0) {
        Item it = stack[sp - 1];
        int stt = state[sp - 1];
        if (it.len < 8) {
            T res = (T)0;
            for (int i = 0; i < it.len; i++) res = res + nan0(a[it.off + i], nans);
            vals[vp++] = res; sp--;
        } else if (it.len <= kLeaf) {
            vals[vp++] = leaf_sum(a + it.off, it.len, nans); sp--;
        } else if (stt == 0) {
            int n2 = it.len / 2; n2 -= n2 % 8;
            state[sp - 1] = 1;
            // evaluate left first, then right
            stack[sp] = {it.off + n2, it.len - n2}; state[sp] = 0; sp++;
            stack[sp] = {it.off, n2}; state[sp] = 0; sp++;
        } else {
            // both children evaluated: left was pushed last so it was evaluated first
            T right = vals[--vp];
            T left = vals[--vp];
            vals[vp++] = left + right;
            sp--;
        }
    }
    return vals[0];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void flat_piece_sums_kernel(const T *__restrict__ flat, int64_t n,
                                                                T *__restrict__ piece_sums,
                                                                unsigned long long *__restrict__ nan_count)
{
    const int64_t npieces_full = n / kPiece;
    const int wave = threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    const int waves_per_block = kBlock / kWave;
    for (int64_t piece = (int64_t)blockIdx.x * waves_per_block + wave; piece < npieces_full;
         piece += (int64_t)gridDim.x * waves_per_block) {
        int nans = 0;
        T s = leaf_sum(flat + piece * kPiece + lane * kLeaf, kLeaf, nans);
        // numpy's recursion on 8192 = 64 leaves halves evenly: pair neighbours, then pairs of pairs ...
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const T other = __shfl_xor(s, d);
            // the lower lane of each pair holds the left operand
            s = (lane & d) ? other + s : s + other;
        }
        int tot_nans = nans;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) tot_nans += __shfl_xor(tot_nans, d);
        if (lane == 0) {
            piece_sums[piece] = s;
            if (tot_nans) atomicAdd(nan_count, (unsigned long long)tot_nans);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void flat_norm_kernel(const T *__restrict__ flat, int64_t n,
                                                          const T *__restrict__ piece_sums,
                                                          const unsigned long long *__restrict__ nan_count,
                                                          T *__restrict__ norm_out)
{
    // One workgroup.  The piece sums are accumulated sequentially (numpy's order) by lane 0, but all lanes fetch
    // them - and the ragged last piece - into LDS first: a lone lane walking global memory pays a full miss
    // latency per element.
    if (blockIdx.x != 0) return;
    constexpr int kStage = 2048;
    __shared__ T stage[kPiece > kStage ? kPiece : kStage];
    const int64_t npieces_full = n / kPiece;
    T res = (T)0;
    for (int64_t i0 = 0; i0 < npieces_full; i0 += kStage) {
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
    const int rem = (int)(n - npieces_full * kPiece);
    for (int t = threadIdx.x; t < rem; t += blockDim.x) stage[t] = flat[npieces_full * kPiece + t];
    __syncthreads();
    __shared__ PairwiseItem pw_stack[32];
    __shared__ T pw_vals[32];
    __shared__ int pw_state[32];
    if (threadIdx.x != 0) return;
    int nans = 0;
    if (rem > 0) res = res + pairwise_generic(stage, rem, nans, pw_stack, pw_vals, pw_state);
    const unsigned long long bad = *nan_count + (unsigned long long)nans;
    const double cnt = (double)(n - (int64_t)bad);
    norm_out[0] = (T)((double)res / cnt);              // numpy 1.26: float32 / int -> float64 -> float32; float64 / int -> float64
}

__global__ __launch_bounds__(kBlock) void divide_by_scalar_kernel(const float *__restrict__ in,
                                                                 const float *__restrict__ scalar,
                                                                 float *__restrict__ out, int64_t n)
{
    const float s = scalar[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(in)[i];
        reinterpret_cast<float4 *>(out)[i] =
            make_float4(__fdiv_rn(a.x, s), __fdiv_rn(a.y, s), __fdiv_rn(a.z, s), __fdiv_rn(a.w, s));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = __fdiv_rn(in[i], s);
}

__global__ __launch_bounds__(kBlock) void divide_by_scalar_f64_kernel(const double *__restrict__ in,
                                                                     const double *__restrict__ scalar,
                                                                     double *__restrict__ out, int64_t n)
{
    const double s = scalar[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i] / s;   // IEEE (no fast-math)
}

// ------------------------------------------------------------------------------------------------
// A2 with float64 inputs (golden G11): NumPy promotes per operation - the result of an operation is float64 if
// either operand ARRAY is float64, else float32; the python float exp_ratio takes the type of the dark it multiplies
// (core/ApCalibrate.py:301-305 keeps float64 FITS data as it is, :439-464 the expressions).  Every operation is
// evaluated in float64 and rounded to float32 where NumPy's result type is float32: for + - * / of float32 operands
// this is exactly the float32 operation (53 >= 2*24 + 2 significand bits: double rounding is innocuous).
// ------------------------------------------------------------------------------------------------
struct MixedTypes {
    int raw, bias, dark, nflat;            // APGPU_F32 / APGPU_U16 (raw only) / APGPU_F64
    int t1, t2, t3, t4;                    // 1 = the operation's NumPy result type is float64
};

__device__ __forceinline__ double load_as_f64(const void *a, int dt, int64_t i)
{
    if (dt == APGPU_F64) return static_cast<const double *>(a)[i];
    if (dt == APGPU_U16) return (double)static_cast<const uint16_t *>(a)[i];
    return (double)static_cast<const float *>(a)[i];
}

__device__ __forceinline__ double round_as(double x, int is64) { return is64 ? x : (double)(float)x; }

__global__ __launch_bounds__(kBlock) void calibrate_mixed_kernel(const void *__restrict__ raw, const void *__restrict__ bias,
                                                                const void *__restrict__ dark, const void *__restrict__ nflat,
                                                                const double *__restrict__ exp_ratio,
                                                                const double *__restrict__ pedestal, int still_biased,
                                                                void *__restrict__ out, int64_t N, int64_t P, MixedTypes ty)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int r64 = ty.raw == APGPU_F64;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const double b = load_as_f64(bias, ty.bias, p), d = load_as_f64(dark, ty.dark, p);
        const double D = still_biased ? round_as(d - b, ty.t2) : d;          // ApCalibrate.py:440-445
        const bool has_flat = nflat != nullptr;
        const double nf = has_flat ? load_as_f64(nflat, ty.nflat, p) : 1.0;
        for (int64_t f = 0; f < N; f++) {
            double r = load_as_f64(raw, ty.raw, f * P + p);
            const double ped = pedestal ? pedestal[f] : 0.0;
            if (ped != 0.0) r = round_as(r + round_as(ped, r64), r64);       // :318-326, in the raw array's own type
            const double e = round_as(exp_ratio[f], ty.t2);                  // weak python scalar
            const double x = round_as(r - b, ty.t1);                         // :439
            const double ds = round_as(e * D, ty.t2);                        // :450
            double y = round_as(x - ds, ty.t3);                              // :451
            if (has_flat && nf != 0.0) y = round_as(y / nf, ty.t4);          // :462-464 (NaN != 0 -> divide -> NaN)
            if (ty.t4) static_cast<double *>(out)[f * P + p] = y;
            else static_cast<float *>(out)[f * P + p] = (float)y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A4 threshold mask + count
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void threshold_mask_kernel(const float *__restrict__ data, int64_t n, double lo_d,
                                                               double hi_d, const double *__restrict__ thr_dev,
                                                               uint8_t *__restrict__ mask,
                                                               unsigned long long *__restrict__ nbad)
{
    if (thr_dev) { lo_d = thr_dev[0]; hi_d = thr_dev[1]; }
    const float lo = (float)lo_d, hi = (float)hi_d;     // numpy 1.26 demotes the float64 scalar to float32
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n16 = n / 16;             // 16 pixels per lane: 4 x 16-byte loads, one 16-byte mask store
    unsigned cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 a = reinterpret_cast<const float4 *>(data)[i * 4 + q];
            const unsigned b0 = (a.x < lo) || (a.x > hi), b1 = (a.y < lo) || (a.y > hi);
            const unsigned b2 = (a.z < lo) || (a.z > hi), b3 = (a.w < lo) || (a.w > hi);
            cnt += b0 + b1 + b2 + b3;
            w[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        reinterpret_cast<uint4 *>(mask)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (int64_t i = n16 * 16 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t b = (data[i] < lo) || (data[i] > hi);
        mask[i] = b;
        cnt += b;
    }
    // wave reduction, then ONE atomic per workgroup (every atomic lands on the same counter: 8192 of them cost more than
    // reading the image)
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
    __shared__ unsigned s_cnt[kBlock / kWave];
    if ((threadIdx.x % kWave) == 0) s_cnt[threadIdx.x / kWave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; w++) tot += s_cnt[w];
        if (tot) atomicAdd(nbad, (unsigned long long)tot);
    }
}

__global__ __launch_bounds__(kBlock) void mask_add_rects_kernel(uint8_t *mask, int64_t W, const int32_t *__restrict__ rects,
                                                               int value)
{
    // blockIdx.y = rectangle; rectangles are applied one launch-row each, overlapping rectangles
    // are serialised by launching them in separate kernel calls from the host side (see C entry).
    const int r0 = rects[0], r1 = rects[1], c0 = rects[2], c1 = rects[3];
    const int64_t w = c1 - c0, h = r1 - r0;
    const int64_t total = w * h;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = r0 + t / w, c = c0 + t % w;
        mask[r * W + c] = (uint8_t)(mask[r * W + c] + value);
    }
}

// ------------------------------------------------------------------------------------------------
// A8 image arithmetic
// ------------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ float arith(float a, float b)
{
    if constexpr (OP == APGPU_OP_ADD) return a + b;
    if constexpr (OP == APGPU_OP_SUB) return a - b;
    if constexpr (OP == APGPU_OP_MUL) return a * b;
    return __fdiv_rn(a, b);
}

template <int OP>
__global__ __launch_bounds__(kBlock) void imarith_f32_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            float scalar, float *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i];
        float4 y = make_float4(scalar, scalar, scalar, scalar);
        if (b) y = reinterpret_cast<const float4 *>(b)[i];
        reinterpret_cast<float4 *>(out)[i] =
            make_float4(arith<OP>(x.x, y.x), arith<OP>(x.y, y.y), arith<OP>(x.z, y.z), arith<OP>(x.w, y.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = arith<OP>(a[i], b ? b[i] : scalar);
}

// float64 images, and float32 (op) float64 pairs: np.add(data1, data2, out=result) computes in float64 as soon as one
// operand array is float64 and stores in data1's dtype (core/ApImArith.py:321-333, same_kind casting).
template <int OP>
__global__ __launch_bounds__(kBlock) void imarith_f64_kernel(const void *__restrict__ a, int a64, const void *__restrict__ b, int b64,
                                                            double scalar, void *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double x = a64 ? static_cast<const double *>(a)[i] : (double)static_cast<const float *>(a)[i];
        const double y = b ? (b64 ? static_cast<const double *>(b)[i] : (double)static_cast<const float *>(b)[i]) : scalar;
        double r;
        if constexpr (OP == APGPU_OP_ADD) r = x + y;
        else if constexpr (OP == APGPU_OP_SUB) r = x - y;
        else if constexpr (OP == APGPU_OP_MUL) r = x * y;
        else r = x / y;
        if (a64) static_cast<double *>(out)[i] = r;
        else static_cast<float *>(out)[i] = (float)r;
    }
}

template <int OP>
__global__ __launch_bounds__(kBlock) void imarith_u16_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                            uint16_t *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned x = a[i], y = b[i];
        unsigned r;
        if constexpr (OP == APGPU_OP_ADD) r = x + y;
        else if constexpr (OP == APGPU_OP_SUB) r = x - y;
        else r = x * y;
        out[i] = (uint16_t)r;                          // numpy integer ufuncs wrap modulo 2^16
    }
}

// ------------------------------------------------------------------------------------------------
// A9 Bayer split
// ------------------------------------------------------------------------------------------------
struct BayerParams {
    int pattern[4];
    int black[4];
};

__global__ __launch_bounds__(kBlock) void bayer_split_kernel(const uint16_t *__restrict__ raw, int64_t H, int64_t W,
                                                            BayerParams bp, uint16_t *__restrict__ planes)
{
    const int64_t P = H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const int64_t r = i / W, c = i - r * W;
        const int k = bp.pattern[(int)(r & 1) * 2 + (int)(c & 1)];
        int v = (int)raw[i] - bp.black[k];
        v = v < 0 ? 0 : v;
#pragma unroll
        for (int q = 0; q < 4; q++) planes[q * P + i] = (q == k) ? (uint16_t)v : (uint16_t)0;
    }
}

// ------------------------------------------------------------------------------------------------
// F1  FITS payload decode / encode on the device (the step either side of the path: astropy's read of
//     big-endian BITPIX 16 / -32 data with the BZERO = 32768 unsigned convention, core/ApCalibrate.py:
//     270-277, and the float32 write of :392-399).  The host only moves raw bytes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bswap32(unsigned x) { return __builtin_bswap32(x); }
__device__ __forceinline__ unsigned bswap16x2(unsigned x) { return ((x & 0x00ff00ffu) << 8) | ((x & 0xff00ff00u) >> 8); }

// BITPIX 16: big-endian int16 + 32768 -> uint16 (flip the sign bit after the byte swap)
__global__ __launch_bounds__(kBlock) void fits_decode_u16_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out,
                                                                int64_t nwords, const uint8_t *__restrict__ tail_in,
                                                                uint16_t *__restrict__ tail_out, int has_tail)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        out[i] = bswap16x2(in[i]) ^ 0x80008000u;
    if (has_tail && blockIdx.x == 0 && threadIdx.x == 0)
        tail_out[0] = (uint16_t)((((unsigned)tail_in[0] << 8) | tail_in[1]) ^ 0x8000u);
}

// BITPIX 16 without the unsigned convention: big-endian int16 -> float32 (ApCalibrate.py:304-307)
__global__ __launch_bounds__(kBlock) void fits_decode_i16_f32_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
                                                                    int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const short v = (short)(((unsigned)in[2 * i] << 8) | in[2 * i + 1]);
        out[i] = (float)v;
    }
}

// BITPIX -32 / 32: 4-byte swap (decode and encode are the same operation)
__global__ __launch_bounds__(kBlock) void fits_swap32_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        v.x = bswap32(v.x); v.y = bswap32(v.y); v.z = bswap32(v.z); v.w = bswap32(v.w);
        reinterpret_cast<uint4 *>(out)[i] = v;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = bswap32(in[i]);
}

// BITPIX -64 / 64: 8-byte swap (decode and encode are the same operation)
__global__ __launch_bounds__(kBlock) void fits_swap64_kernel(const unsigned long long *__restrict__ in,
                                                            unsigned long long *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = __builtin_bswap64(in[i]);
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// =================================================================================================
extern "C" int apgpu_calibrate(const void *raw, int raw_dtype, const float *bias, const float *dark, const float *nflat,
                               const float *exp_ratio, const float *pedestal, int dark_still_biased, float *out,
                               int64_t n_frames, int64_t n_pixels, void *stream)
{
    if (!raw || !bias || !dark || !exp_ratio || !out) return fail(APGPU_EINVAL, "calibrate: NULL pointer argument");
    if (n_frames <= 0 || n_pixels <= 0) return fail(APGPU_EINVAL, "calibrate: empty slab (%lld x %lld)", (long long)n_frames, (long long)n_pixels);
    if (raw_dtype != APGPU_F32 && raw_dtype != APGPU_U16) return fail(APGPU_EINVAL, "calibrate: bad raw dtype %d", raw_dtype);
    if (!aligned16(raw) || !aligned16(bias) || !aligned16(dark) || !aligned16(out) || (nflat && !aligned16(nflat)))
        return fail(APGPU_EINVAL, "calibrate: buffers must be 16-byte aligned");
    const int vec = (n_pixels % 4) == 0 || n_frames == 1;
    const unsigned grid = grid_for(vec ? (n_pixels + 3) / 4 : n_frames * n_pixels);
    hipStream_t st = as_stream(stream);
    if (raw_dtype == APGPU_F32)
        hipLaunchKernelGGL(calibrate_kernel<float>, dim3(grid), dim3(kBlock), 0, st, (const float *)raw, bias, dark, nflat,
                           exp_ratio, pedestal, dark_still_biased, out, n_frames, n_pixels, vec);
    else
        hipLaunchKernelGGL(calibrate_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, st, (const uint16_t *)raw, bias, dark,
                           nflat, exp_ratio, pedestal, dark_still_biased, out, n_frames, n_pixels, vec);
    return check_launch("calibrate");
}

extern "C" int apgpu_calibrate_mixed(const void *raw, int raw_dtype, const void *bias, int bias_dtype, const void *dark,
                                     int dark_dtype, const void *nflat, int nflat_dtype, const double *exp_ratio,
                                     const double *pedestal, int dark_still_biased, void *out, int out_dtype,
                                     int64_t n_frames, int64_t n_pixels, void *stream)
{
    if (!raw || !bias || !dark || !exp_ratio || !out) return fail(APGPU_EINVAL, "calibrate_mixed: NULL pointer argument");
    if (n_frames <= 0 || n_pixels <= 0) return fail(APGPU_EINVAL, "calibrate_mixed: n_frames = %lld, n_pixels = %lld", (long long)n_frames, (long long)n_pixels);
    auto okm = [](int d) { return d == APGPU_F32 || d == APGPU_F64; };
    if (!(raw_dtype == APGPU_F32 || raw_dtype == APGPU_U16 || raw_dtype == APGPU_F64) || !okm(bias_dtype) || !okm(dark_dtype) ||
        (nflat && !okm(nflat_dtype)) || !okm(out_dtype))
        return fail(APGPU_EINVAL, "calibrate_mixed: bad dtype tag");
    MixedTypes ty;
    ty.raw = raw_dtype; ty.bias = bias_dtype; ty.dark = dark_dtype; ty.nflat = nflat ? nflat_dtype : APGPU_F32;
    const int r64 = raw_dtype == APGPU_F64, b64 = bias_dtype == APGPU_F64, d64 = dark_dtype == APGPU_F64;
    ty.t1 = r64 || b64;
    ty.t2 = dark_still_biased ? (d64 || b64) : d64;
    ty.t3 = ty.t1 || ty.t2;
    ty.t4 = nflat ? (ty.t3 || nflat_dtype == APGPU_F64) : ty.t3;
    if ((out_dtype == APGPU_F64) != (ty.t4 != 0))
        return fail(APGPU_EINVAL, "calibrate_mixed: NumPy's result type for these inputs is %s, out_dtype says otherwise",
                    ty.t4 ? "float64" : "float32");
    hipLaunchKernelGGL(calibrate_mixed_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, as_stream(stream), raw, bias, dark, nflat,
                       exp_ratio, pedestal, dark_still_biased, out, n_frames, n_pixels, ty);
    return check_launch("calibrate_mixed");
}

extern "C" size_t apgpu_flat_normalize_ws_bytes(int64_t n_pixels)
{
    if (n_pixels < 0) return 0;
    return 16 + sizeof(float) * (size_t)(n_pixels / kPiece + 1);
}

template <typename T>
static int flat_normalize_impl(const T *flat, T *nflat, T *norm_out, int64_t n_pixels, void *ws, size_t ws_bytes, void *stream);

extern "C" size_t apgpu_flat_normalize_f64_ws_bytes(int64_t n_pixels)
{
    if (n_pixels < 0) return 0;
    return 16 + sizeof(double) * (size_t)(n_pixels / kPiece + 1);
}

extern "C" int apgpu_flat_normalize_f64(const double *flat, double *nflat, double *norm_out, int64_t n_pixels, void *ws,
                                        size_t ws_bytes, void *stream)
{
    if (ws_bytes < apgpu_flat_normalize_f64_ws_bytes(n_pixels))
        return fail(APGPU_EWORKSPACE, "flat_normalize_f64: workspace %zu < %zu bytes", ws_bytes, apgpu_flat_normalize_f64_ws_bytes(n_pixels));
    return flat_normalize_impl<double>(flat, nflat, norm_out, n_pixels, ws, ws_bytes, stream);
}

extern "C" int apgpu_flat_normalize_f32(const float *flat, float *nflat, float *norm_out, int64_t n_pixels, void *ws,
                                        size_t ws_bytes, void *stream)
{
    if (ws_bytes < apgpu_flat_normalize_ws_bytes(n_pixels))
        return fail(APGPU_EWORKSPACE, "flat_normalize: workspace %zu < %zu bytes", ws_bytes, apgpu_flat_normalize_ws_bytes(n_pixels));
    return flat_normalize_impl<float>(flat, nflat, norm_out, n_pixels, ws, ws_bytes, stream);
}

template <typename T>
static int flat_normalize_impl(const T *flat, T *nflat, T *norm_out, int64_t n_pixels, void *ws, size_t ws_bytes, void *stream)
{
    if (!flat || !norm_out || !ws) return fail(APGPU_EINVAL, "flat_normalize: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "flat_normalize: n_pixels = %lld", (long long)n_pixels);
    if (!aligned16(flat) || (nflat && !aligned16(nflat)) || !aligned16(ws)) return fail(APGPU_EINVAL, "flat_normalize: buffers must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    unsigned long long *nan_count = static_cast<unsigned long long *>(ws);
    T *piece_sums = reinterpret_cast<T *>(static_cast<char *>(ws) + 16);
    if (hipMemsetAsync(nan_count, 0, 16, st) != hipSuccess) return fail(APGPU_ELAUNCH, "flat_normalize: memset failed");
    const int64_t npieces = n_pixels / kPiece;
    if (npieces > 0) {
        const unsigned grid = grid_for(npieces * kWave);
        hipLaunchKernelGGL(flat_piece_sums_kernel<T>, dim3(grid), dim3(kBlock), 0, st, flat, n_pixels, piece_sums, nan_count);
        if (int rc = check_launch("flat_piece_sums")) return rc;
    }
    hipLaunchKernelGGL(flat_norm_kernel<T>, dim3(1), dim3(kBlock), 0, st, flat, n_pixels, piece_sums, nan_count, norm_out);
    if (int rc = check_launch("flat_norm")) return rc;
    if (nflat) {
        if constexpr (sizeof(T) == 8)
            hipLaunchKernelGGL(divide_by_scalar_f64_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, st, flat, norm_out, nflat, n_pixels);
        else
            hipLaunchKernelGGL(divide_by_scalar_kernel, dim3(grid_for(n_pixels / 4 + 1)), dim3(kBlock), 0, st, flat, norm_out, nflat,
                               n_pixels);
        if (int rc = check_launch("flat_divide")) return rc;
    }
    return APGPU_OK;
}

extern "C" int apgpu_threshold_mask_f32(const float *data, int64_t n_pixels, double lothresh, double hithresh,
                                        const double *thresholds_dev, uint8_t *mask, int64_t *nbad_out, void *stream)
{
    if (!data || !mask || !nbad_out) return fail(APGPU_EINVAL, "threshold_mask: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "threshold_mask: n_pixels = %lld", (long long)n_pixels);
    if (!aligned16(data) || !aligned16(mask)) return fail(APGPU_EINVAL, "threshold_mask: buffers must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(nbad_out, 0, sizeof(int64_t), st) != hipSuccess) return fail(APGPU_ELAUNCH, "threshold_mask: memset failed");
    hipLaunchKernelGGL(threshold_mask_kernel, dim3(grid_for(n_pixels / 16 + 1)), dim3(kBlock), 0, st, data, n_pixels, lothresh,
                       hithresh, thresholds_dev, mask, reinterpret_cast<unsigned long long *>(nbad_out));
    return check_launch("threshold_mask");
}

extern "C" int apgpu_mask_add_rects_u8(uint8_t *mask, int64_t height, int64_t width, const int32_t *rects, int32_t n_rects,
                                       int32_t value, void *stream)
{
    if (!mask || (n_rects > 0 && !rects)) return fail(APGPU_EINVAL, "mask_add_rects: NULL pointer argument");
    if (height <= 0 || width <= 0 || n_rects < 0) return fail(APGPU_EINVAL, "mask_add_rects: bad shape");
    hipStream_t st = as_stream(stream);
    // one launch per rectangle: overlapping rectangles must accumulate (the reference sums flags),
    // stream order serialises the read-modify-write of shared pixels.
    for (int k = 0; k < n_rects; k++) {
        hipLaunchKernelGGL(mask_add_rects_kernel, dim3(256), dim3(kBlock), 0, st, mask, width, rects + 4 * k, value);
        if (int rc = check_launch("mask_add_rects")) return rc;
    }
    return APGPU_OK;
}

extern "C" int apgpu_imarith_f64(const void *a, int a_dtype, const void *b, int b_dtype, double scalar, int op, void *out,
                                 int64_t n_pixels, void *stream)
{
    if (!a || !out) return fail(APGPU_EINVAL, "imarith_f64: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "imarith_f64: n_pixels = %lld", (long long)n_pixels);
    if (op < APGPU_OP_ADD || op > APGPU_OP_DIV) return fail(APGPU_EINVAL, "imarith_f64: bad operation %d", op);
    auto okd = [](int d) { return d == APGPU_F32 || d == APGPU_F64; };
    if (!okd(a_dtype) || (b && !okd(b_dtype))) return fail(APGPU_EINVAL, "imarith_f64: bad dtype tag");
    if (a_dtype == APGPU_F32 && !(b && b_dtype == APGPU_F64))
        return fail(APGPU_EINVAL, "imarith_f64: nothing is float64 here - use apgpu_imarith");
    hipStream_t st = as_stream(stream);
    const unsigned grid = grid_for(n_pixels);
    const int a64 = a_dtype == APGPU_F64, b64 = b_dtype == APGPU_F64;
    switch (op) {
    case APGPU_OP_ADD: hipLaunchKernelGGL(imarith_f64_kernel<APGPU_OP_ADD>, dim3(grid), dim3(kBlock), 0, st, a, a64, b, b64, scalar, out, n_pixels); break;
    case APGPU_OP_SUB: hipLaunchKernelGGL(imarith_f64_kernel<APGPU_OP_SUB>, dim3(grid), dim3(kBlock), 0, st, a, a64, b, b64, scalar, out, n_pixels); break;
    case APGPU_OP_MUL: hipLaunchKernelGGL(imarith_f64_kernel<APGPU_OP_MUL>, dim3(grid), dim3(kBlock), 0, st, a, a64, b, b64, scalar, out, n_pixels); break;
    default: hipLaunchKernelGGL(imarith_f64_kernel<APGPU_OP_DIV>, dim3(grid), dim3(kBlock), 0, st, a, a64, b, b64, scalar, out, n_pixels); break;
    }
    return check_launch("imarith_f64");
}

extern "C" int apgpu_imarith(const void *a, const void *b, double scalar, int op, int dtype, void *out, int64_t n_pixels,
                             void *stream)
{
    if (!a || !out) return fail(APGPU_EINVAL, "imarith: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "imarith: n_pixels = %lld", (long long)n_pixels);
    if (op < APGPU_OP_ADD || op > APGPU_OP_DIV) return fail(APGPU_EINVAL, "imarith: bad operation %d", op);
    hipStream_t st = as_stream(stream);
    const unsigned grid = grid_for(n_pixels / 4 + 1);
    if (dtype == APGPU_F32) {
        if (!aligned16(a) || !aligned16(out) || (b && !aligned16(b))) return fail(APGPU_EINVAL, "imarith: buffers must be 16-byte aligned");
        const float *fa = (const float *)a, *fb = (const float *)b;
        float *fo = (float *)out;
        const float s = (float)scalar;
        switch (op) {
        case APGPU_OP_ADD: hipLaunchKernelGGL(imarith_f32_kernel<APGPU_OP_ADD>, dim3(grid), dim3(kBlock), 0, st, fa, fb, s, fo, n_pixels); break;
        case APGPU_OP_SUB: hipLaunchKernelGGL(imarith_f32_kernel<APGPU_OP_SUB>, dim3(grid), dim3(kBlock), 0, st, fa, fb, s, fo, n_pixels); break;
        case APGPU_OP_MUL: hipLaunchKernelGGL(imarith_f32_kernel<APGPU_OP_MUL>, dim3(grid), dim3(kBlock), 0, st, fa, fb, s, fo, n_pixels); break;
        default: hipLaunchKernelGGL(imarith_f32_kernel<APGPU_OP_DIV>, dim3(grid), dim3(kBlock), 0, st, fa, fb, s, fo, n_pixels); break;
        }
    } else if (dtype == APGPU_U16) {
        if (!b) return fail(APGPU_EUNSUPPORTED, "imarith: uint16 image (op) scalar is rejected (numpy raises UFuncTypeError)");
        if (op == APGPU_OP_DIV) return fail(APGPU_EUNSUPPORTED, "imarith: uint16 DIV is rejected (numpy raises UFuncTypeError)");
        const uint16_t *ua = (const uint16_t *)a, *ub = (const uint16_t *)b;
        uint16_t *uo = (uint16_t *)out;
        switch (op) {
        case APGPU_OP_ADD: hipLaunchKernelGGL(imarith_u16_kernel<APGPU_OP_ADD>, dim3(grid), dim3(kBlock), 0, st, ua, ub, uo, n_pixels); break;
        case APGPU_OP_SUB: hipLaunchKernelGGL(imarith_u16_kernel<APGPU_OP_SUB>, dim3(grid), dim3(kBlock), 0, st, ua, ub, uo, n_pixels); break;
        default: hipLaunchKernelGGL(imarith_u16_kernel<APGPU_OP_MUL>, dim3(grid), dim3(kBlock), 0, st, ua, ub, uo, n_pixels); break;
        }
    } else {
        return fail(APGPU_EINVAL, "imarith: bad dtype %d", dtype);
    }
    return check_launch("imarith");
}

extern "C" int apgpu_bayer_split_u16(const uint16_t *raw, int64_t height, int64_t width, const int32_t *pattern_host,
                                     const int32_t *black_host, uint16_t *planes, void *stream)
{
    if (!raw || !planes || !pattern_host) return fail(APGPU_EINVAL, "bayer_split: NULL pointer argument");
    if (height <= 0 || width <= 0) return fail(APGPU_EINVAL, "bayer_split: bad shape");
    BayerParams bp;
    for (int k = 0; k < 4; k++) {
        if (pattern_host[k] < 0 || pattern_host[k] > 3) return fail(APGPU_EINVAL, "bayer_split: pattern entries must be 0..3");
        bp.pattern[k] = pattern_host[k];
        bp.black[k] = black_host ? black_host[k] : 0;
    }
    hipLaunchKernelGGL(bayer_split_kernel, dim3(grid_for(height * width)), dim3(kBlock), 0, as_stream(stream), raw, height, width,
                       bp, planes);
    return check_launch("bayer_split");
}

extern "C" int apgpu_fits_decode(const void *payload, int bitpix, int unsigned16, void *out, int64_t n_pixels, void *stream)
{
    if (!payload || !out) return fail(APGPU_EINVAL, "fits_decode: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "fits_decode: n_pixels = %lld", (long long)n_pixels);
    if (!aligned16(payload) || !aligned16(out)) return fail(APGPU_EINVAL, "fits_decode: buffers must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (bitpix == 16 && unsigned16) {
        const int64_t nwords = n_pixels / 2;
        const int has_tail = (int)(n_pixels & 1);
        hipLaunchKernelGGL(fits_decode_u16_kernel, dim3(grid_for(nwords + 1)), dim3(kBlock), 0, st, (const unsigned *)payload,
                           (unsigned *)out, nwords, (const uint8_t *)payload + 4 * nwords, (uint16_t *)out + 2 * nwords, has_tail);
    } else if (bitpix == 16) {
        hipLaunchKernelGGL(fits_decode_i16_f32_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, st, (const uint8_t *)payload,
                           (float *)out, n_pixels);
    } else if (bitpix == -32 || bitpix == 32) {
        hipLaunchKernelGGL(fits_swap32_kernel, dim3(grid_for(n_pixels / 4 + 1)), dim3(kBlock), 0, st, (const unsigned *)payload,
                           (unsigned *)out, n_pixels);
    } else if (bitpix == -64 || bitpix == 64) {
        hipLaunchKernelGGL(fits_swap64_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, st, (const unsigned long long *)payload,
                           (unsigned long long *)out, n_pixels);
    } else {
        return fail(APGPU_EUNSUPPORTED, "fits_decode: BITPIX %d is decoded on the host", bitpix);
    }
    return check_launch("fits_decode");
}

extern "C" int apgpu_fits_encode_f32(const float *data, void *payload, int64_t n_pixels, void *stream)
{
    if (!data || !payload) return fail(APGPU_EINVAL, "fits_encode: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "fits_encode: n_pixels = %lld", (long long)n_pixels);
    if (!aligned16(data) || !aligned16(payload)) return fail(APGPU_EINVAL, "fits_encode: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(fits_swap32_kernel, dim3(grid_for(n_pixels / 4 + 1)), dim3(kBlock), 0, as_stream(stream),
                       (const unsigned *)data, (unsigned *)payload, n_pixels);
    return check_launch("fits_encode");
}

extern "C" int apgpu_fits_encode_f64(const double *data, void *payload, int64_t n_pixels, void *stream)
{
    if (!data || !payload) return fail(APGPU_EINVAL, "fits_encode_f64: NULL pointer argument");
    if (n_pixels <= 0) return fail(APGPU_EINVAL, "fits_encode_f64: n_pixels = %lld", (long long)n_pixels);
    if (!aligned16(data) || !aligned16(payload)) return fail(APGPU_EINVAL, "fits_encode_f64: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(fits_swap64_kernel, dim3(grid_for(n_pixels)), dim3(kBlock), 0, as_stream(stream),
                       (const unsigned long long *)data, (unsigned long long *)payload, n_pixels);
    return check_launch("fits_encode_f64");
}
