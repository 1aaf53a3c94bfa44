// tools/isa_probe/force_deep_tt2.hip -- ISA study only (see force_deep_mask.hip): the deep operand prefetch in the two-tile
// instances (256 VGPRs, the recurrence wave of <10, 9, SEG, 2> spills), compiled with -DAQ_FORCE_DEEP_DB=3|4 -DAQ_FORCE_DEEP_TT2.
#include "aq_core_sweep_la.h"
template __global__ void aq_core_sweep_la_kernel<10, 9, true, 2, false>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<10, 10, true, 2, false>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<11, 11, true, 2, false>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<11, 10, false, 2, false>(const AqCoreArgs);
