// tools/isa_probe/tt1_nt3.hip -- ISA study only: one-tile instances with 3 / 6 residual tiles on the recurrence wave (registers, scratch)
#include "aq_core_sweep_la.h"
template __global__ void aq_core_sweep_la_kernel<10, 10, true, 1, false, 3>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<10, 9, true, 1, false, 6>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<9, 9, true, 1, false, 9>(const AqCoreArgs);
