// stack_mad.h - what the translation units of the median / mad_std fast kernels share (stack_mad.hip, stack_mad_wide.hip,
// stack_mad_pairs.hip): the kernels' argument block.
#pragma once
#include "stack_kernels.h"

namespace apgpu_stack {

struct MadParams {
    const void *frames;
    int64_t stride, P;
    float *mean;
    int32_t *count;
    double *mean64, *std64;
    int32_t *ws;
    float cl, cu;               // thresh x 1.482602218505602 / 2 for the lower / upper bound
};

int launch_mad_wide(const MadParams &q, int np, bool u16, hipStream_t st);       // stack_mad_wide.hip: 65 .. 128 frames, one pixel per lane
int launch_mad_pairs(const MadParams &q, int np, hipStream_t st);                // stack_mad_pairs.hip: uint16 frames, two pixels per lane (3 .. 64 itself,
int launch_mad_pairs_wide(const MadParams &q, int np, hipStream_t st);           // 65 .. 128 in stack_mad_pairs_wide.hip)

}  // namespace apgpu_stack
