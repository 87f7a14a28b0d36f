// Explicit instantiation: raw dtype float, fused calibration false.
#include "stack_kernels.h"
namespace apgpu_stack {
template int launch_np<float, false>(const StackParams &, bool, hipStream_t);
}
