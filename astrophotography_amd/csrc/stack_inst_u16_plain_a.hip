// Explicit instantiation: raw dtype uint16_t, fused calibration false, slot counts 1, 4, 8, 12, 16, 24.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<1, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<4, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<8, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<12, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<16, uint16_t, false>(const StackParams &, bool, hipStream_t);
template int launch_one<24, uint16_t, false>(const StackParams &, bool, hipStream_t);
}
