// aq_launch_la1.hip -- instances of the look-ahead sweep kernel with 1 trait tile(s) per workgroup (see aq_launch_la.h).
#include "aq_launch_la.h"
#include "aq_core_sweep_la.h"

template <int NT, int NT2>
static void aq_la_go(bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, true, 1>), dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, false, 1>), dim3(grid), dim3(512), 0, st, a);
}

template <int NT, int NT2, int N3>
static void aq_la_go3(bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, true, 1, false, N3>), dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, false, 1, false, N3>), dim3(grid), dim3(512), 0, st, a);
}

int aq_la_launch_tt1(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (nt3x > 0) {
#define AQ_L3(NT_)                                                                                                  \
    if (NT == NT_ && nt3x == 3 && NT2 == NT_) { aq_la_go3<NT_, NT_, 3>(seg, grid, st, a); return 0; }               \
    if (NT == NT_ && nt3x == 6 && NT2 == NT_ - 1) { aq_la_go3<NT_, NT_ - 1, 6>(seg, grid, st, a); return 0; }       \
    if (NT == NT_ && nt3x == 9 && NT2 == NT_) { aq_la_go3<NT_, NT_, 9>(seg, grid, st, a); return 0; }
    AQ_L3(8) AQ_L3(9) AQ_L3(10) AQ_L3(11)
#undef AQ_L3
    return -1;
  }
#define AQ_LA(NT_)                                                                        \
  if (NT == NT_ && NT2 == NT_) { aq_la_go<NT_, NT_>(seg, grid, st, a); return 0; }        \
  if (NT == NT_ && NT2 == NT_ - 1 && NT_ > 1) { aq_la_go<NT_, (NT_ > 1 ? NT_ - 1 : 1)>(seg, grid, st, a); return 0; }
  AQ_LA(1) AQ_LA(2) AQ_LA(3) AQ_LA(4) AQ_LA(5) AQ_LA(6) AQ_LA(7) AQ_LA(8) AQ_LA(9) AQ_LA(10) AQ_LA(11)
#undef AQ_LA
  // up to 18 tiles per wave for the sample-split launches of large n (never chained: the parts of a group must be co-resident)
#define AQ_LB(NT_)                                                                                                              \
  if (!seg && NT == NT_ && NT2 == NT_) { hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_, false, 1>), dim3(grid), dim3(512), 0, st, a); return 0; } \
  if (!seg && NT == NT_ && NT2 == NT_ - 1) { hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT_, NT_ - 1, false, 1>), dim3(grid), dim3(512), 0, st, a); return 0; }
  AQ_LB(12) AQ_LB(13) AQ_LB(14) AQ_LB(15) AQ_LB(16) AQ_LB(17) AQ_LB(18)
#undef AQ_LB
  return -1;
}
