// Development aid (round 4): is  q0 = fma(x, y, x * yl);  r = fma(-d, q0, x);  q = fma(r, y, q0)  (y = RN(1 / d), yl = fma(-d, y, 1) * y)
// the correctly rounded x / d?  Compares with __fdiv_rn on random and adversarial operands inside the kernels' range guards.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_check.hip -o tools/div_check && tools/div_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float div4(float x, float d)
{
    const float y = __fdiv_rn(1.0f, d);
    const float yl = __builtin_fmaf(-d, y, 1.0f) * y;
    const float t = x * yl;
    const float q0 = __builtin_fmaf(x, y, t);
    const float r = __builtin_fmaf(-d, q0, x);
    return __builtin_fmaf(r, y, q0);
}

__device__ __forceinline__ float div5(float x, float d)
{
    const float y = __fdiv_rn(1.0f, d);
    const float q0 = x * y;
    const float r0 = __builtin_fmaf(-d, q0, x);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, x);
    return __builtin_fmaf(r1, y, q1);
}

__device__ __forceinline__ uint32_t rng(uint64_t &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 32);
}

// mode 0: random significands, exponents within the guards; 1: divisor significand from a special set; 2: numerators next to
// the product of the divisor with a MIDPOINT of two adjacent floats (the quotient lands as close to a rounding boundary as
// the operands allow)
__global__ void check(int mode, int iters, unsigned long long *bad4, unsigned long long *bad5, float *ex)
{
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + mode * 77 + 1;
    unsigned long long b4 = 0, b5 = 0;
    for (int i = 0; i < iters; i++) {
        uint32_t a = rng(s), b = rng(s), c = rng(s);
        uint32_t dm = b & 0x7FFFFF;
        if (mode == 1) {
            const uint32_t sp[8] = {0x7FFFFF, 0x000001, 0x7FFFFE, 0x400000, 0x3FFFFF, 0x000000, 0x555555, 0x2AAAAA};
            dm = sp[c & 7] ^ ((c >> 3) & 3);
        }
        const int de = 127 + (int)((c >> 8) % 61) - 30;                 // 2^-30 .. 2^30
        float d = __uint_as_float(((uint32_t)de << 23) | dm);
        if (c & 0x80000000u) d = -d;
        float x;
        if (mode == 2) {
            // q* = midpoint between two floats with a random significand; x = RN(q* d) moved by -2 .. +2 ulps
            const uint32_t qm = a & 0x7FFFFF;
            const double qmid = (double)__uint_as_float((127u << 23) | qm) + 0x1p-24;
            const float x0 = (float)(qmid * (double)d);
            x = __uint_as_float(__float_as_uint(x0) + ((a >> 23) % 5) - 2);
            x = x * __uint_as_float((uint32_t)(127 + (int)((c >> 16) % 41) - 20) << 23);
        } else {
            const int xe = 127 + (int)((c >> 16) % 81) - 40;
            x = __uint_as_float(((uint32_t)xe << 23) | (a & 0x7FFFFF));
            if (a & 0x80000000u) x = -x;
        }
        const float ref = __fdiv_rn(x, d);
        const float aq = fabsf(ref);
        if (!(aq > 0x1p-50f && aq < 0x1p50f)) continue;                  // the kernels' quotient guard
        const float q4 = div4(x, d), q5 = div5(x, d);
        if (__float_as_uint(q4) != __float_as_uint(ref)) { b4++; ex[0] = x; ex[1] = d; }
        if (__float_as_uint(q5) != __float_as_uint(ref)) b5++;
    }
    atomicAdd(bad4, b4);
    atomicAdd(bad5, b5);
}

int main()
{
    unsigned long long *bad;
    float *ex;
    hipMalloc(&bad, 16);
    hipMalloc(&ex, 8);
    for (int mode = 0; mode < 3; mode++) {
        hipMemset(bad, 0, 16);
        const int iters = 20000;
        hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, mode, iters, bad, bad + 1, ex);
        unsigned long long h[2];
        float hx[2];
        hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        hipMemcpy(hx, ex, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.3g operand pairs, mismatches with __fdiv_rn: 4-op %llu, 5-op %llu", mode, 8192.0 * 256 * iters, h[0], h[1]);
        if (h[0]) printf("   (e.g. x = %a, d = %a)", hx[0], hx[1]);
        printf("\n");
    }
    return 0;
}
