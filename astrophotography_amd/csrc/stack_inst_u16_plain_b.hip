// Explicit instantiation: raw dtype uint16_t, fused calibration false, slot counts 32, 48.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<32, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<48, uint16_t, false>(const StackParams &, bool, hipStream_t);
}
