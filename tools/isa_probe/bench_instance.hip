// tools/isa_probe/bench_instance.hip -- ISA study only: the two instances bench.py's C3 run launches, for a quick look at
// registers, scratch and the helper's code while working on the kernel (seconds instead of the minutes of the full build).
#include "aq_core_sweep_la.h"
template __global__ void aq_core_sweep_la_kernel<10, 9, true, 2, false>(const AqCoreArgs);
template __global__ void aq_core_sweep_la_kernel<10, 10, true, 2, false>(const AqCoreArgs);
