// Development aid (round 6, VERDICT r5 next #1b): can the clip's core sums ride on the matrix pipe for free?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_coissue.hip -o tools/mfma_coissue && tools/mfma_coissue
// The fused calibrate + clip kernel is bound by vector-instruction issue (1581 VALU instructions per wavefront, 0.98 busy).  Its
// core moments are, per sorted value, one v_sub (d = x - c), one v_add (S += d) and one v_fmac (Q += d d): the last two could be
// v_mfma_f32_4x4x1_16b_f32 (A = d, B = 1 -> row sums; A = B = d -> the diagonal holds sum d^2) IF an MFMA costs the SIMD's vector
// issue less than the 2-cycle instruction it replaces.  This program measures exactly that, chip-wide (256 x waves-per-SIMD
// workgroups of 4 wavefronts, wall time by events), as wave-instructions per SIMD per ns:
//   1. streams of ONE kind: v_min_f32 (the sort's instruction, 4-cycle class), v_add_f32 / v_fmac_f32 (2-cycle class), the MFMA alone;
//   2. the kernel's mix in miniature, per block of 12: 11 v_min_f32 + {nothing | 1 v_fmac_f32 | 1 MFMA} - what does the twelfth cost?
//   3. every wave runs v_min only except ONE wave per SIMD that runs MFMAs only (separate waves: the case MI355X_MICROARCH.md says
//      runs concurrently) - does the v_min rate of the others drop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 512
typedef float v4f __attribute__((ext_vector_type(4)));

#define MIN8                                                                                                                    \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));        \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a6) : "v"(a7));        \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a3) : "v"(a2));        \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a5) : "v"(a4)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a7) : "v"(a6));
#define MIN3                                                                                                                    \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));        \
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a4) : "v"(a5));
#define ADD8                                                                                                                    \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));        \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(a7));        \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(a2));        \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(a4)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(a6));
#define FMAC(acc) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a6), "v"(a7));
// four independent accumulators in rotation: an MFMA never waits for its own previous result (4x4x1: 2 passes)
#define MFMA(acc) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a6), "v"(a7));

#define KERNEL(name, body)                                                                                                      \
    __global__ void name(float *out, float seed)                                                                                \
    {                                                                                                                           \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;                                                                           \
        v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0, m3 = m0;                                                               \
        const int wave = threadIdx.x >> 6; (void)wave;                                                                          \
        for (int r = 0; r < REPS; r++) { body }                                                                                 \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + m0.x + m1.y + m2.z + m3.w;                        \
        if (s == 12345.678f) out[0] = s;                                                                                        \
    }

KERNEL(k_min, MIN8 MIN8 MIN8 MIN8)                                                            // 32 per trip
KERNEL(k_add, ADD8 ADD8 ADD8 ADD8)                                                            // 32
KERNEL(k_fmac, FMAC(f0) FMAC(f1) FMAC(f2) FMAC(f3) FMAC(f0) FMAC(f1) FMAC(f2) FMAC(f3)
               FMAC(f0) FMAC(f1) FMAC(f2) FMAC(f3) FMAC(f0) FMAC(f1) FMAC(f2) FMAC(f3))        // 16
KERNEL(k_mfma, MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3) MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3)
               MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3) MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3))        // 16
// round 6: is any form of a 32-bit minimum in the 2-cycle class?  (DPP / SDWA encodings with identity controls, the gfx950
// v_minimum3_f32, 64-bit and 16-bit minima, and v_min_f32 with the IEEE mode bit cleared - MODE[9], quieting of signalling NaNs)
#define OP8(ins, tail)                                                                                                          \
    asm volatile(ins " %0, %0, %1 " tail : "+v"(a0) : "v"(a1)); asm volatile(ins " %0, %0, %1 " tail : "+v"(a2) : "v"(a3));      \
    asm volatile(ins " %0, %0, %1 " tail : "+v"(a4) : "v"(a5)); asm volatile(ins " %0, %0, %1 " tail : "+v"(a6) : "v"(a7));      \
    asm volatile(ins " %0, %0, %1 " tail : "+v"(a1) : "v"(a0)); asm volatile(ins " %0, %0, %1 " tail : "+v"(a3) : "v"(a2));      \
    asm volatile(ins " %0, %0, %1 " tail : "+v"(a5) : "v"(a4)); asm volatile(ins " %0, %0, %1 " tail : "+v"(a7) : "v"(a6));
#define X4(x) x x x x
KERNEL(k_min_dpp, X4(OP8("v_min_f32_dpp", "quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf")))
KERNEL(k_min_sdwa, X4(OP8("v_min_f32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")))
KERNEL(k_min_u16, X4(OP8("v_min_u16", "")))
KERNEL(k_min_i16, X4(OP8("v_min_i16", "")))
KERNEL(k_minimum3, X4(asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
                      asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));
                      asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
                      asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));))
__global__ void k_min_noieee(float *out, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
    for (int r = 0; r < REPS; r++) { MIN8 MIN8 MIN8 MIN8 }
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1");
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[0] = s;
}
__global__ void k_min_f64(float *out, float seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    for (int r = 0; r < REPS; r++) { X4(OP8("v_min_f64", "")) }
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678) out[0] = (float)s;
}
KERNEL(k_min11, MIN8 MIN3 MIN8 MIN3 MIN8 MIN3 MIN8 MIN3)                                       // 44: four blocks of 11
KERNEL(k_min11_fmac, MIN8 MIN3 FMAC(f0) MIN8 MIN3 FMAC(f1) MIN8 MIN3 FMAC(f2) MIN8 MIN3 FMAC(f3))   // 44 + 4
KERNEL(k_min11_mfma, MIN8 MIN3 MFMA(m0) MIN8 MIN3 MFMA(m1) MIN8 MIN3 MFMA(m2) MIN8 MIN3 MFMA(m3))   // 44 + 4
// one wave of the workgroup (= one of the SIMD's resident waves per workgroup... a 256-thread workgroup has one wave per SIMD, so
// "wave 0 of every OTHER workgroup" is not placeable; instead: workgroups of 1024 threads = 4 waves per SIMD, waves 12 .. 15 - the
// fourth wave of each SIMD - run MFMAs, the other twelve v_min)
__global__ void k_split(float *out, float seed, int mfma_waves)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    v4f m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0, m3 = m0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 16 - mfma_waves) {
        for (int r = 0; r < REPS; r++) {
            MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3) MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3)
            MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3) MFMA(m0) MFMA(m1) MFMA(m2) MFMA(m3)
        }
    } else {
        for (int r = 0; r < REPS; r++) { MIN8 MIN8 MIN8 MIN8 }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0.x + m1.y + m2.z + m3.w;
    if (s == 12345.678f) out[0] = s;
}

struct K { const char *name; void (*fn)(float *, float); int valu, mfma; };

int main()
{
    K ks[] = {{"v_min_f32", k_min, 32, 0}, {"v_add_f32", k_add, 32, 0}, {"v_fmac_f32", k_fmac, 16, 0},
              {"v_min_f32_dpp (identity)", k_min_dpp, 32, 0}, {"v_min_f32_sdwa (dwords)", k_min_sdwa, 32, 0}, {"v_min_f32, MODE.IEEE = 0", k_min_noieee, 32, 0}, {"v_min_f32 (again)", k_min, 32, 0}, {"v_min_f32, MODE.IEEE = 0 (again)", k_min_noieee, 32, 0}, {"v_min_f32 (third)", k_min, 32, 0},
              {"v_minimum3_f32", k_minimum3, 32, 0}, {"v_min_f64", k_min_f64, 32, 0}, {"v_min_u16", k_min_u16, 32, 0}, {"v_min_i16", k_min_i16, 32, 0},
              {"v_mfma_f32_4x4x1_16b_f32", k_mfma, 0, 16}, {"11 v_min", k_min11, 44, 0}, {"11 v_min + 1 v_fmac", k_min11_fmac, 48, 0},
              {"11 v_min + 1 v_mfma_4x4x1", k_min11_mfma, 44, 4}};
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("chip-wide, 256-thread workgroups, waves per SIMD = blocks / 256; ns per trip of the loop body per SIMD-resident wave set\n");
    printf("%-28s %5s %5s | %9s %9s %9s %9s  (ns per loop trip per wave at 1 / 2 / 4 / 8 waves per SIMD)\n", "stream", "valu", "mfma", "1", "2", "4", "8");
    for (auto &k : ks) {
        printf("%-28s %5d %5d |", k.name, k.valu, k.mfma);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // time the SIMD spends per loop trip of ONE of its waves: total / (launches x trips x waves per SIMD)
            printf(" %9.2f", ms * 1e6 / (5.0 * REPS * wps));
        }
        printf("\n");
    }
    printf("\nsplit roles, 1024-thread workgroups (4 waves per SIMD), 256 workgroups: waves 16-m .. 15 run 16 MFMAs per trip, the others 32 v_min\n");
    printf("%-28s %12s\n", "MFMA waves per workgroup", "ns per trip");
    for (int m : {0, 4, 8, 16}) {
        hipLaunchKernelGGL(k_split, dim3(256), dim3(1024), 0, 0, d, 1.0f, m);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k_split, dim3(256), dim3(1024), 0, 0, d, 1.0f, m);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28d %12.2f\n", m, ms * 1e6 / (5.0 * REPS));
    }
    return 0;
}
