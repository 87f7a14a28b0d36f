// stack_mad_wide.hip - the median / mad_std fast kernels of 65 .. 128 frames: stack_mad.hip compiled a second time, into a code
// object of its own (see the note there).
#define APGPU_MAD_WIDE
#include "stack_mad.hip"
