// Explicit instantiation: raw dtype float, fused calibration false, slot counts 1, 4, 8, 12, 16, 24.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<1, float, false>(const StackParams &, bool, hipStream_t);
template int launch_one<4, float, false>(const StackParams &, bool, hipStream_t);
template int launch_one<8, float, false>(const StackParams &, bool, hipStream_t);
template int launch_one<12, float, false>(const StackParams &, bool, hipStream_t);
template int launch_one<16, float, false>(const StackParams &, bool, hipStream_t);
template int launch_one<24, float, false>(const StackParams &, bool, hipStream_t);
}
