// aq_launch_la2.hip -- instances of the look-ahead sweep kernel with 2 trait tile(s) per workgroup (see aq_launch_la.h).
#include "aq_launch_la.h"
#include "aq_core_sweep_la.h"

template <int NT, int NT2>
static void aq_la_go(bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, true, 2>), dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT2, false, 2>), dim3(grid), dim3(512), 0, st, a);
}

template <int NT>
static void aq_la_go9(bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (seg) hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT, true, 2, false, 9>), dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((aq_core_sweep_la_kernel<NT, NT, false, 2, false, 9>), dim3(grid), dim3(512), 0, st, a);
}

int aq_la_launch_tt2(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a) {
  if (nt3x == 9) {
#ifdef AQ_DIAG_TIME
    return -1;   // the cycle counters of the diagnostic build push these instances over the register budget (the ISA proof stops them)
#else
    if (NT2 != NT) return -1;
    switch (NT) {
      case 8: aq_la_go9<8>(seg, grid, st, a); return 0;
      case 9: aq_la_go9<9>(seg, grid, st, a); return 0;
      case 10: aq_la_go9<10>(seg, grid, st, a); return 0;
      case 11: aq_la_go9<11>(seg, grid, st, a); return 0;
      default: return -1;
    }
#endif
  }
#define AQ_LA(NT_)                                                                        \
  if (NT == NT_ && NT2 == NT_) { aq_la_go<NT_, NT_>(seg, grid, st, a); return 0; }        \
  if (NT == NT_ && NT2 == NT_ - 1 && NT_ > 1) { aq_la_go<NT_, (NT_ > 1 ? NT_ - 1 : 1)>(seg, grid, st, a); return 0; }
  AQ_LA(1) AQ_LA(2) AQ_LA(3) AQ_LA(4) AQ_LA(5) AQ_LA(6) AQ_LA(7) AQ_LA(8) AQ_LA(9) AQ_LA(10) AQ_LA(11)
#undef AQ_LA
  return -1;
}
