// stack_mad_pairs_wide.hip - the two-pixels-per-lane median / mad_std kernels of 65 .. 128 uint16 frames: stack_mad_pairs.hip compiled
// a second time, into a code object of its own (see the note in stack_mad.hip).
#define APGPU_MAD_WIDE
#include "stack_mad_pairs.hip"
