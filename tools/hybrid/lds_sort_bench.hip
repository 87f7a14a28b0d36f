// Development aid (round 4): does moving half of each compare-exchange onto the LDS pipe pay on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I astrophotography_amd/csrc tools/hybrid/lds_sort_bench.hip -o tools/hybrid/lds_sort_bench
// Part 1: DS instruction rates per CU (ds_min_rtn_f32, ds_read_b32, ds_write_b32, mixed with VALU min/max streams).
// Part 2: the pruned 64-slot network of the stack kernels as plain VALU code against hybrid plans (tools/hybrid/plan_hybrid.py):
//         a hybrid compare-exchange keeps one operand in LDS (lane-private column, bank = lane): ds_min_rtn_f32 / ds_max_rtn_f32
//         leaves one output there and returns the old value, one v_max / v_min on the returned value makes the other output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "stack_sort.h"

using namespace apgpu_stack;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float amin(float *p, float x) { return __hip_atomic_fetch_min(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ float amax(float *p, float x) { return __hip_atomic_fetch_max(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

// ---------------------------------------------------------------- part 1: rates
// KIND 0: ds_min_rtn_f32, 1: ds_read_b32, 2: ds_write_b32, 3: none.  NV: VALU min/max instructions per DS instruction slot.
template <int KIND, int NV>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int reps, float seed)
{
    extern __shared__ float lds[];
    float *col = lds + threadIdx.x;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) col[i * 256] = seed + i;
    float acc = 0.f;
    for (int r = 0; r < reps; r++) {
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if constexpr (KIND == 0) t[i] = amin(&col[i * 256], a[i & 7]);
            else if constexpr (KIND == 1) t[i] = col[i * 256];
            else if constexpr (KIND == 2) { col[i * 256] = a[i & 7]; t[i] = 0.f; }
            else t[i] = 0.f;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int x = (i * NV + j) & 7, y = (x + 1) & 7;
                if (j & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[x]) : "v"(a[y]));
                else asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[x]) : "v"(a[y]));
            }
        }
        if constexpr (KIND == 1) asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) acc += t[i];
    }
    float s = acc;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + col[0];
}

// ---------------------------------------------------------------- part 2: the sort
#define H_CE(a, b) cmpx(v[a], v[b]);
#define H_W(w, r) col[(r) * 256] = v[w];
#define H_R(w, r) v[w] = col[(r) * 256];
#define H_LO_ISSUE(a, b, r, t) const float t = amin(&col[(r) * 256], v[b]);
#define H_LO_DONE(a, b, t) asm("v_max_f32 %0, %1, %2" : "=v"(v[b]) : "v"(t), "v"(v[b]));
#define H_HI_ISSUE(a, b, r, t) const float t = amax(&col[(r) * 256], v[a]);
#define H_HI_DONE(a, b, t) asm("v_min_f32 %0, %1, %2" : "=v"(v[a]) : "v"(t), "v"(v[a]));

template <int VARIANT>
__device__ __forceinline__ void sort_variant(float (&v)[64], float *col)
{
#pragma unroll
    for (int g = 0; g < 64; g += 4) sort4(v[g], v[g + 1], v[g + 2], v[g + 3]);
    if constexpr (VARIANT == 0) {
        net_from<64, 4, 4, 0>(v);
    } else if constexpr (VARIANT == 1) {
#include "plan64_0.85.inc"
    } else if constexpr (VARIANT == 2) {
#include "plan64_0.7.inc"
    } else if constexpr (VARIANT == 3) {
#include "plan64_0.5.inc"
    } else if constexpr (VARIANT == 4) {
#include "plan64_0.2.inc"
    }
}

// frames[f][p]; every workgroup walks `reps` tiles of 256 pixels; out[p] = checksum of what the fast path reads + the sum of all
template <int VARIANT>
__global__ __launch_bounds__(256) void sort_kernel(const float *frames, int64_t stride, int64_t P, float *out, int reps)
{
    extern __shared__ float lds[];
    float *col = lds + threadIdx.x;
    for (int r = 0; r < reps; r++) {
        const int64_t base = (int64_t)__builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x + (int64_t)r * gridDim.x) % (P / 256))) * 256;
        const int64_t p = base + threadIdx.x;
        float v[64];
        const float *fp = frames + base;                     // wave-uniform frame pointer + lane offset, as in load_raw
#pragma unroll
        for (int f = 0; f < 64; f++) {
            v[f] = fp[threadIdx.x];
            fp += stride;
            if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        sort_variant<VARIANT>(v, col);
        float core = 0.f;
#pragma unroll
        for (int i = 4; i < 60; i++) core += v[i];
        const float chk = v[0] + 2.f * v[1] + 3.f * v[2] + 5.f * v[3] + 7.f * v[29] + 11.f * v[30] + 13.f * v[31] + 17.f * v[32] + 19.f * v[33] + 23.f * v[34] +
                          29.f * v[60] + 31.f * v[61] + 37.f * v[62] + 41.f * v[63] + core;
        if (r == 0) out[p] = chk;
        else if (chk == -12345.f) out[p] = 0.f;
    }
}

template <int KIND, int NV>
static void run_rate(const char *name, int waves_per_simd, int reps)
{
    float *out;
    const int blocks = 256 * waves_per_simd;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const size_t lds = 160 * 1024 / waves_per_simd - 1024;      // exactly waves_per_simd workgroups of 4 waves per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<KIND, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<KIND, NV>), dim3(blocks), dim3(256), lds, 0, out, 10, 1.f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<KIND, NV>), dim3(blocks), dim3(256), lds, 0, out, reps, 1.f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ds_per_cu = (double)waves_per_simd * 4 * reps * 16;            // DS wave-instructions per CU
    const double valu_per_simd = (double)waves_per_simd * reps * 16 * NV;
    printf("%-28s waves/SIMD %d: %8.3f ms  DS %6.3f wave-instr/CU/ns  VALU %6.3f wave-instr/SIMD/ns\n", name, waves_per_simd, ms,
           KIND == 3 ? 0.0 : ds_per_cu / (ms * 1e6), valu_per_simd / (ms * 1e6));
    CHECK(hipFree(out));
}

template <int VARIANT>
static void run_sort(const char *name, const float *frames, int64_t P, float *out, std::vector<float> &ref, int waves_per_simd, int reps)
{
    const size_t lds = 160 * 1024 / waves_per_simd - 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sort_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = (int)(P / 256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipMemset(out, 0, P * 4));
    hipLaunchKernelGGL((sort_kernel<VARIANT>), dim3(blocks), dim3(256), lds, 0, frames, P, P, out, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<float> h(P);
    CHECK(hipMemcpy(h.data(), out, P * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    if (ref.empty()) ref = h;
    else for (int64_t i = 0; i < P; i++) bad += (h[i] != ref[i]);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((sort_kernel<VARIANT>), dim3(blocks), dim3(256), lds, 0, frames, P, P, out, reps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double columns = (double)blocks * 256 * reps;
    printf("%-34s waves/SIMD %d: %8.3f ms  %7.2f Gcolumns/s  mismatches %ld\n", name, waves_per_simd, best, columns / (best * 1e6), bad);
}

int main(int argc, char **argv)
{
    const int reps = 2000;
    for (int w = 1; w <= 4; w++) {
        if (w == 4) w = 5;
        run_rate<0, 0>("ds_min_rtn_f32", w, reps);
        run_rate<1, 0>("ds_read_b32", w, reps);
        run_rate<2, 0>("ds_write_b32", w, reps);
        run_rate<3, 2>("valu min/max only (2/slot)", w, reps);
        run_rate<0, 1>("ds_min_rtn + 1 valu", w, reps);
        run_rate<0, 2>("ds_min_rtn + 2 valu", w, reps);
        run_rate<0, 4>("ds_min_rtn + 4 valu", w, reps);
        run_rate<1, 2>("ds_read + 2 valu", w, reps);
        run_rate<2, 2>("ds_write + 2 valu", w, reps);
    }
    // sort: 64 frames of 262144 pixels (64 MB: Infinity-Cache resident), every pixel's column distinct
    const int64_t P = 262144;
    std::vector<float> h((size_t)64 * P);
    uint32_t s = 12345u;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = 1000.f + (float)((s >> 8) & 0xffff) * 0.01f - ((s >> 30) == 3 ? 2000.f : 0.f); }
    float *frames, *out;
    CHECK(hipMalloc(&frames, h.size() * 4));
    CHECK(hipMalloc(&out, P * 4));
    CHECK(hipMemcpy(frames, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int w = 2; w <= 4; w++) {
        std::vector<float> ref;
        run_sort<0>("valu pruned network (890 valu)", frames, P, out, ref, w, 16);
        run_sort<1>("hybrid lambda 0.85 (534v + 280ds)", frames, P, out, ref, w, 16);
        run_sort<2>("hybrid lambda 0.7 (499v + 325ds)", frames, P, out, ref, w, 16);
        run_sort<3>("hybrid lambda 0.5 (434v + 428ds)", frames, P, out, ref, w, 16);
        run_sort<4>("hybrid lambda 0.2 (390v + 542ds)", frames, P, out, ref, w, 16);
    }
    return 0;
}
