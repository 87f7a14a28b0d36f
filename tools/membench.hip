// Development micro-benchmark (not part of the product): achievable HBM read rate for the stack
// kernel's access pattern - every lane reads one element from each of N frames - versus wider loads.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int NP, int VEC>
__global__ __launch_bounds__(256) void colsum(const float *__restrict__ frames, float *__restrict__ out, int64_t P)
{
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (g >= P) return;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] = 0.f;
    float v[NP][VEC];
#pragma unroll
    for (int f = 0; f < NP; f++) {
        const float *p = frames + (int64_t)f * P + g;
        if constexpr (VEC == 1) v[f][0] = p[0];
        else if constexpr (VEC == 2) { float2 t = *reinterpret_cast<const float2 *>(p); v[f][0] = t.x; v[f][1] = t.y; }
        else { float4 t = *reinterpret_cast<const float4 *>(p); v[f][0] = t.x; v[f][1] = t.y; v[f][2] = t.z; v[f][3] = t.w; }
    }
#pragma unroll
    for (int f = 0; f < NP; f++)
#pragma unroll
        for (int k = 0; k < VEC; k++) acc[k] += v[f][k];
#pragma unroll
    for (int k = 0; k < VEC; k++) out[g + k] = acc[k];
}

__global__ __launch_bounds__(256) void stream_read(const float4 *__restrict__ in, float *__restrict__ out, int64_t n4)
{
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 t = in[i];
        acc += t.x + t.y + t.z + t.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const int N = 64;
    const int64_t P = 4096LL * 4096;
    float *frames, *out;
    hipMalloc(&frames, sizeof(float) * N * P);
    hipMalloc(&out, sizeof(float) * P);
    hipMemset(frames, 0, sizeof(float) * N * P);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int i = 0; i < 10; i++) {
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
            tot += ms;
        }
        const double bytes = 4.0 * N * P + 4.0 * P;
        printf("%-28s avg %.3f ms  min %.3f ms  -> %.0f GB/s (min)\n", name, tot / 10, best, bytes / best / 1e6);
    };
    timeit("colsum 1 px/lane (dword)", [&] { hipLaunchKernelGGL((colsum<64, 1>), dim3(P / 256), dim3(256), 0, 0, frames, out, P); });
    timeit("colsum 2 px/lane (dwordx2)", [&] { hipLaunchKernelGGL((colsum<64, 2>), dim3(P / 512), dim3(256), 0, 0, frames, out, P); });
    timeit("colsum 4 px/lane (dwordx4)", [&] { hipLaunchKernelGGL((colsum<64, 4>), dim3(P / 1024), dim3(256), 0, 0, frames, out, P); });
    timeit("stream float4 grid-stride", [&] { hipLaunchKernelGGL(stream_read, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)frames, out, (int64_t)N * P / 4); });
    return 0;
}
