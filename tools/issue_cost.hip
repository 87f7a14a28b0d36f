// Development aid: issue cost (cycles per wave64 instruction on one SIMD) of the instruction kinds the stack kernels use.
//   hipcc --offload-arch=gfx950 -O2 tools/issue_cost.hip -o tools/issue_cost && tools/issue_cost
// Each kernel runs REPS x 64 independent copies of one instruction per wave; 1 or 3 waves per SIMD; cycles from s_memrealtime-free
// wall_clock64 around the whole launch are avoided: the kernel itself brackets the loop with s_memtime (core clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 256

#define KERNEL(name, body)                                                                               \
    __global__ void name(unsigned long long *out, float seed)                                            \
    {                                                                                                    \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;  \
        double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7; \
        typedef float v2f __attribute__((ext_vector_type(2)));                                           \
        v2f p0 = {seed, seed}, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3, p4 = p0 + 4, p5 = p0 + 5, p6 = p0 + 6, p7 = p0 + 7;           \
        unsigned long long t0 = __builtin_readcyclecounter();                                            \
        for (int r = 0; r < REPS; r++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; u++) { body }                                       \
        }                                                                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                                            \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y; \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                 \
        if (s == 12345.678f) out[0] = 0;                                                                 \
    }

#define X8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define F32OP(ins) asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(a3)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(a7)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(a2)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(a4)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(a6));
#define F64OP(ins) asm volatile(ins " %0, %0, %1" : "+v"(d0) : "v"(d1)); asm volatile(ins " %0, %0, %1" : "+v"(d2) : "v"(d3)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(d4) : "v"(d5)); asm volatile(ins " %0, %0, %1" : "+v"(d6) : "v"(d7)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(d1) : "v"(d0)); asm volatile(ins " %0, %0, %1" : "+v"(d3) : "v"(d2)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(d5) : "v"(d4)); asm volatile(ins " %0, %0, %1" : "+v"(d7) : "v"(d6));
#define PKOP(ins) asm volatile(ins " %0, %0, %1" : "+v"(p0) : "v"(p1)); asm volatile(ins " %0, %0, %1" : "+v"(p2) : "v"(p3)); \
                  asm volatile(ins " %0, %0, %1" : "+v"(p4) : "v"(p5)); asm volatile(ins " %0, %0, %1" : "+v"(p6) : "v"(p7)); \
                  asm volatile(ins " %0, %0, %1" : "+v"(p1) : "v"(p0)); asm volatile(ins " %0, %0, %1" : "+v"(p3) : "v"(p2)); \
                  asm volatile(ins " %0, %0, %1" : "+v"(p5) : "v"(p4)); asm volatile(ins " %0, %0, %1" : "+v"(p7) : "v"(p6));

KERNEL(k_add_f32, F32OP("v_add_f32"))
KERNEL(k_min_f32, F32OP("v_min_f32"))
KERNEL(k_max_f32, F32OP("v_max_f32"))
KERNEL(k_mul_f32, F32OP("v_mul_f32"))
KERNEL(k_add_f64, F64OP("v_add_f64"))
KERNEL(k_mul_f64, F64OP("v_mul_f64"))
KERNEL(k_pk_add, PKOP("v_pk_add_f32"))
KERNEL(k_pk_mul, PKOP("v_pk_mul_f32"))
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(d1), "v"(d2)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d3) : "v"(d4), "v"(d5));
                  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d6) : "v"(d7), "v"(d1)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d2) : "v"(d4), "v"(d7));
                  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(d1), "v"(d2)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d3) : "v"(d4), "v"(d5));
                  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d6) : "v"(d7), "v"(d1)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d2) : "v"(d4), "v"(d7));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p3) : "v"(p4), "v"(p5));
                 asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p6) : "v"(p7), "v"(p1)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(p4), "v"(p7));
                 asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p3) : "v"(p4), "v"(p5));
                 asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p6) : "v"(p7), "v"(p1)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(p4), "v"(p7));)
KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5));
                  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a7));
                  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5));
                  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a7));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(a0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a1));
                      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(a3));
                      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d4) : "v"(a4)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d5) : "v"(a5));
                      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d6) : "v"(a6)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d7) : "v"(a7));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3));
                  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(a5)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(a7));
                  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a0)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a2));
                  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(a4)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(a6));)
KERNEL(k_cmp_f64, asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d0), "v"(d1) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d2), "v"(d3) : "vcc");
                  asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d4), "v"(d5) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d6), "v"(d7) : "vcc");
                  asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d1), "v"(d0) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d3), "v"(d2) : "vcc");
                  asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d5), "v"(d4) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d7), "v"(d6) : "vcc");)
KERNEL(k_cmp_f32, asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a2), "v"(a3) : "vcc");
                  asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a4), "v"(a5) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a6), "v"(a7) : "vcc");
                  asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a1), "v"(a0) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a3), "v"(a2) : "vcc");
                  asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a5), "v"(a4) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a7), "v"(a6) : "vcc");)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
               asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));
               asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
               asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));)
#define I32OP(ins) asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(a3)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(a7)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(a2)); \
                   asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(a4)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(a6));
KERNEL(k_min_i32, I32OP("v_min_i32"))
KERNEL(k_max_i32, I32OP("v_max_i32"))
KERNEL(k_min_u32, I32OP("v_min_u32"))
KERNEL(k_max_u32, I32OP("v_max_u32"))
KERNEL(k_xor, I32OP("v_xor_b32"))
KERNEL(k_and, I32OP("v_and_b32"))
KERNEL(k_ashr, I32OP("v_ashrrev_i32"))
KERNEL(k_add_u32, I32OP("v_add_u32"))
KERNEL(k_sub_f32, F32OP("v_sub_f32"))
KERNEL(k_pk_min_u16, I32OP("v_pk_min_u16"))
KERNEL(k_pk_max_i16, I32OP("v_pk_max_i16"))
KERNEL(k_min_f16, I32OP("v_pk_min_f16"))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_mov_b32 %0, %1" : "=v"(a2) : "v"(a3)); asm volatile("v_mov_b32 %0, %1" : "=v"(a4) : "v"(a5)); asm volatile("v_mov_b32 %0, %1" : "=v"(a6) : "v"(a7));
              asm volatile("v_mov_b32 %0, %1" : "=v"(a1) : "v"(a0)); asm volatile("v_mov_b32 %0, %1" : "=v"(a3) : "v"(a2)); asm volatile("v_mov_b32 %0, %1" : "=v"(a5) : "v"(a4)); asm volatile("v_mov_b32 %0, %1" : "=v"(a7) : "v"(a6));)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
               asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));
               asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
               asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_gt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a0) : "v"(a1), "v"(a2) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a3) : "v"(a4), "v"(a5) : "vcc");
                  asm volatile("v_cmp_gt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a6) : "v"(a7), "v"(a0) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a1) : "v"(a2), "v"(a3) : "vcc");)
KERNEL(k_cnd_sgpr, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "s"(t0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "s"(t0));
                   asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "s"(t0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "s"(t0));
                   asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a0), "s"(t0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a2), "s"(t0));
                   asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(a4), "s"(t0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(a6), "s"(t0));)
KERNEL(k_snop, asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0");)
KERNEL(k_minmax_dep, asm volatile("v_min_f32 %0, %2, %3\nv_max_f32 %1, %2, %3" : "=&v"(a0), "=&v"(a1) : "v"(a2), "v"(a3)); asm volatile("v_min_f32 %0, %2, %3\nv_max_f32 %1, %2, %3" : "=&v"(a2), "=&v"(a3) : "v"(a0), "v"(a1));
                     asm volatile("v_min_f32 %0, %2, %3\nv_max_f32 %1, %2, %3" : "=&v"(a4), "=&v"(a5) : "v"(a6), "v"(a7)); asm volatile("v_min_f32 %0, %2, %3\nv_max_f32 %1, %2, %3" : "=&v"(a6), "=&v"(a7) : "v"(a4), "v"(a5));)


// round 3: candidates for a cheaper compare-exchange / cheaper clip arithmetic (3-operand integer forms, carries, VOP2 fma)
#define I3OP(ins) asm volatile(ins " %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5)); \
                  asm volatile(ins " %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7)); \
                  asm volatile(ins " %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5)); \
                  asm volatile(ins " %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a2) : "v"(a4), "v"(a7));
KERNEL(k_sad_u32, I3OP("v_sad_u32"))
KERNEL(k_add3_u32, I3OP("v_add3_u32"))
KERNEL(k_xad_u32, I3OP("v_xad_u32"))
KERNEL(k_and_or, I3OP("v_and_or_b32"))
KERNEL(k_or3, I3OP("v_or3_b32"))
KERNEL(k_bfi, I3OP("v_bfi_b32"))
KERNEL(k_lshl_add, I3OP("v_lshl_add_u32"))
KERNEL(k_alignbit, I3OP("v_alignbit_b32"))
KERNEL(k_perm, I3OP("v_perm_b32"))
KERNEL(k_mad_u24, I3OP("v_mad_u32_u24"))
KERNEL(k_bfe, I3OP("v_bfe_u32"))
KERNEL(k_min3_u32, I3OP("v_min3_u32"))
KERNEL(k_sub_u32, I32OP("v_sub_u32"))
KERNEL(k_mul_u24, I32OP("v_mul_u32_u24"))
KERNEL(k_lshlrev, I32OP("v_lshlrev_b32"))
KERNEL(k_or, I32OP("v_or_b32"))
KERNEL(k_fmac_f32, F32OP("v_fmac_f32"))
KERNEL(k_max_u16, I32OP("v_max_u16"))
KERNEL(k_add_f32_abs, asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a0) : "v"(a1)); asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a2) : "v"(a3));
                      asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a4) : "v"(a5)); asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a6) : "v"(a7));
                      asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a1) : "v"(a0)); asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a3) : "v"(a2));
                      asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a5) : "v"(a4)); asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(a7) : "v"(a6));)
KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a0) : "v"(a1) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a2) : "v"(a3) : "vcc");
                 asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a4) : "v"(a5) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a6) : "v"(a7) : "vcc");
                 asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a1) : "v"(a0) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a3) : "v"(a2) : "vcc");
                 asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a5) : "v"(a4) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a7) : "v"(a6) : "vcc");)
// the candidate compare-exchange: lo = v_min_f32(a, b); hi = v_sad_u32(a, b, lo) (bit patterns of non-negative floats)
KERNEL(k_min_sad, asm volatile("v_min_f32 %0, %2, %3\nv_sad_u32 %1, %2, %3, %0" : "=&v"(a0), "=&v"(a1) : "v"(a2), "v"(a3)); asm volatile("v_min_f32 %0, %2, %3\nv_sad_u32 %1, %2, %3, %0" : "=&v"(a2), "=&v"(a3) : "v"(a0), "v"(a1));
                  asm volatile("v_min_f32 %0, %2, %3\nv_sad_u32 %1, %2, %3, %0" : "=&v"(a4), "=&v"(a5) : "v"(a6), "v"(a7)); asm volatile("v_min_f32 %0, %2, %3\nv_sad_u32 %1, %2, %3, %0" : "=&v"(a6), "=&v"(a7) : "v"(a4), "v"(a5));)
// lo = v_min_f32(a, b); hi = (a + b) - lo on the bit patterns (two fast-class integer instructions)
KERNEL(k_min_addsub, asm volatile("v_min_f32 %0, %2, %3\nv_add_u32 %1, %2, %3\nv_sub_u32 %1, %1, %0" : "=&v"(a0), "=&v"(a1) : "v"(a2), "v"(a3)); asm volatile("v_min_f32 %0, %2, %3\nv_add_u32 %1, %2, %3\nv_sub_u32 %1, %1, %0" : "=&v"(a2), "=&v"(a3) : "v"(a0), "v"(a1));
                     asm volatile("v_min_f32 %0, %2, %3\nv_add_u32 %1, %2, %3\nv_sub_u32 %1, %1, %0" : "=&v"(a4), "=&v"(a5) : "v"(a6), "v"(a7)); asm volatile("v_min_f32 %0, %2, %3\nv_add_u32 %1, %2, %3\nv_sub_u32 %1, %1, %0" : "=&v"(a6), "=&v"(a7) : "v"(a4), "v"(a5));)

struct K { const char *name; void (*fn)(unsigned long long *, float); int per_iter; };

int main()
{
    K ks[] = {{"v_add_f32", k_add_f32, 64}, {"v_min_f32", k_min_f32, 64}, {"v_max_f32", k_max_f32, 64}, {"v_mul_f32", k_mul_f32, 64}, {"v_fma_f32", k_fma_f32, 64},
              {"v_max3_f32", k_max3, 64}, {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp_gt_f32", k_cmp_f32, 64},
              {"v_pk_add_f32", k_pk_add, 64}, {"v_pk_mul_f32", k_pk_mul, 64}, {"v_pk_fma_f32", k_pk_fma, 64},
              {"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_cmp_gt_f64", k_cmp_f64, 64},
              {"v_cvt_f64_f32", k_cvt_f64_f32, 64}, {"s_nop 0", k_snop, 64}, {"v_min_i32", k_min_i32, 64}, {"v_max_i32", k_max_i32, 64}, {"v_min_u32", k_min_u32, 64}, {"v_max_u32", k_max_u32, 64},
              {"v_xor_b32", k_xor, 64}, {"v_and_b32", k_and, 64}, {"v_ashrrev_i32", k_ashr, 64}, {"v_add_u32", k_add_u32, 64}, {"v_sub_f32", k_sub_f32, 64},
              {"v_pk_min_u16", k_pk_min_u16, 64}, {"v_pk_max_i16", k_pk_max_i16, 64}, {"v_pk_min_f16", k_min_f16, 64}, {"v_mov_b32", k_mov, 64}, {"v_med3_f32", k_med3, 64},
              {"v_cmp_gt_f32 + v_cndmask (vcc)", k_cmp_cnd, 64}, {"v_cndmask_b32 (sgpr mask)", k_cnd_sgpr, 64}, {"v_min+v_max dependent pairs", k_minmax_dep, 64},
              {"v_sad_u32", k_sad_u32, 64}, {"v_add3_u32", k_add3_u32, 64}, {"v_xad_u32", k_xad_u32, 64}, {"v_and_or_b32", k_and_or, 64}, {"v_or3_b32", k_or3, 64},
              {"v_bfi_b32", k_bfi, 64}, {"v_lshl_add_u32", k_lshl_add, 64}, {"v_alignbit_b32", k_alignbit, 64}, {"v_perm_b32", k_perm, 64}, {"v_mad_u32_u24", k_mad_u24, 64},
              {"v_bfe_u32", k_bfe, 64}, {"v_min3_u32", k_min3_u32, 64}, {"v_sub_u32", k_sub_u32, 64}, {"v_mul_u32_u24", k_mul_u24, 64}, {"v_lshlrev_b32", k_lshlrev, 64}, {"v_or_b32", k_or, 64},
              {"v_fmac_f32", k_fmac_f32, 64}, {"v_max_u16", k_max_u16, 64}, {"v_add_f32 |a|, -b (VOP3)", k_add_f32_abs, 64}, {"v_add_co + v_addc_co (per instr)", k_add_co, 64},
              {"v_min_f32 + v_sad_u32 (per instr)", k_min_sad, 64}, {"v_min_f32+v_add_u32+v_sub_u32 (per instr)", k_min_addsub, 96}};
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 4096);
    std::vector<unsigned long long> h(4096);
    printf("%-30s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (auto &k : ks) {
        double res[3];
        int wpb[3] = {256, 512, 1024};                  // threads per block: 4 / 8 / 16 waves = 1 / 2 / 4 per SIMD on one CU
        for (int c = 0; c < 3; c++) {
            hipLaunchKernelGGL(k.fn, dim3(1), dim3(wpb[c]), 0, 0, d, 1.0f);
            hipLaunchKernelGGL(k.fn, dim3(1), dim3(wpb[c]), 0, 0, d, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, sizeof(unsigned long long), hipMemcpyDeviceToHost);
            // cycles per instruction PER SIMD: (cycles of wave 0) / (instructions per wave * waves per SIMD)
            res[c] = (double)h[0] / ((double)REPS * k.per_iter) / (wpb[c] / 256.0);
        }
        printf("%-30s %12.2f %12.2f %12.2f\n", k.name, res[0], res[1], res[2]);
    }
    // chip-wide throughput: many 256-thread workgroups, wall time by events -> wave64 instructions per SIMD per microsecond
    printf("\nchip-wide (1024 SIMDs), workgroups of 4 waves; waves per SIMD = blocks / 256:\n%-30s %10s %10s %10s %10s   (wave-instructions per SIMD per ns)\n", "instruction", "1", "2", "4", "8");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        printf("%-30s", k.name);
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
            const double instr_per_simd = 5.0 * wps * (double)REPS * k.per_iter;
            printf(" %10.3f", instr_per_simd / (ms * 1e6));
        }
        printf("\n");
    }
    return 0;
}
