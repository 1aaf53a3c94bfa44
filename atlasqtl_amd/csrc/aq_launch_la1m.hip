// aq_launch_la1m.hip -- instances of the look-ahead sweep kernel for Y with missing values (MASK, one trait tile per workgroup;
// see aq_launch_la.h).
#include "aq_launch_la.h"
#include "aq_core_sweep_la.h"

int aq_la_launch_mask(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (nt3x > 0) return -1;   // residual tiles on the recurrence wave of a MASK instance: tried (NT / NT / 3, chained), hung on the GPU; not built
#define AQ_LA(NT_)                                                                                                                 \
  if (NT == NT_ && (NT2 == NT_ || (NT2 == NT_ - 1 && NT_ > 1))) {                                                                  \
    if (NT2 == NT_) {                                                                                                              \
      if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_, true, 1, true>), dim3(grid), dim3(512), 0, st, a);            \
      else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_, false, 1, true>), dim3(grid), dim3(512), 0, st, a);               \
    } else {                                                                                                                       \
      if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, (NT_ > 1 ? NT_ - 1 : 1), true, 1, true>), dim3(grid), dim3(512), 0, st, a);  \
      else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, (NT_ > 1 ? NT_ - 1 : 1), false, 1, true>), dim3(grid), dim3(512), 0, st, a);     \
    }                                                                                                                              \
    return 0;                                                                                                                      \
  }
  AQ_LA(1) AQ_LA(2) AQ_LA(3) AQ_LA(4) AQ_LA(5) AQ_LA(6) AQ_LA(7) AQ_LA(8) AQ_LA(9) AQ_LA(10) AQ_LA(11)
#undef AQ_LA
#define AQ_LB(NT_)                                                                                                                 \
  if (!seg && NT == NT_ && NT2 == NT_) { hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_, false, 1, true>), dim3(grid), dim3(512), 0, st, a); return 0; } \
  if (!seg && NT == NT_ && NT2 == NT_ - 1) { hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_ - 1, false, 1, true>), dim3(grid), dim3(512), 0, st, a); return 0; }
  AQ_LB(12) AQ_LB(13) AQ_LB(14) AQ_LB(15) AQ_LB(16) AQ_LB(17) AQ_LB(18)
#undef AQ_LB
  return -1;
}
