// Explicit instantiation: raw dtype uint16_t, fused calibration true.
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_np<uint16_t, true>(const StackParams &, bool, hipStream_t);
}
