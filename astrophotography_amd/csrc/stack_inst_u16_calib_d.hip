// Explicit instantiation: raw dtype uint16_t, fused calibration true, slot counts 96.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<96, uint16_t, true>(const StackParams &, bool, hipStream_t);
}
