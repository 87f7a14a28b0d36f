// Explicit instantiation: raw dtype uint16_t, fused calibration false, slot counts 64.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<64, uint16_t, false>(const StackParams &, bool, hipStream_t);
}
