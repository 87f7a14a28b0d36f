// Explicit instantiation: raw dtype float, fused calibration true, slot counts 1, 4, 8, 12, 16, 24.
#define APGPU_STACK_INSTANTIATE
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_one<1, float, true>(const StackParams &, bool, hipStream_t);
template int launch_one<4, float, true>(const StackParams &, bool, hipStream_t);
template int launch_one<8, float, true>(const StackParams &, bool, hipStream_t);
template int launch_one<12, float, true>(const StackParams &, bool, hipStream_t);
template int launch_one<16, float, true>(const StackParams &, bool, hipStream_t);
template int launch_one<24, float, true>(const StackParams &, bool, hipStream_t);
}
