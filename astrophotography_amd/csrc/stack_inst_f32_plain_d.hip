// Explicit instantiation: raw dtype float, fused calibration false, slot counts 96.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<96, float, false>(const StackParams &, bool, hipStream_t);
}
