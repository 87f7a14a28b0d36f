// Explicit instantiation: raw dtype float, fused calibration true.
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_np<float, true>(const StackParams &, bool, hipStream_t);
}
