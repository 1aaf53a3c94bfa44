// aq_launch_la.h -- launch dispatch of the look-ahead sweep kernel (aq_core_sweep_la.h).  The template instances are
// compiled in two translation units (aq_launch_la1.hip: one trait tile per workgroup, aq_launch_la2.hip: two) so that
// the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "aq_core_sweep.h"

// Launches aq_core_sweep_la_kernel<NT, NT2, seg, TT> on `grid` workgroups of 512 threads.  Returns 0, or -1 when there is
// no instantiation for (NT, NT2): NT in 1..11, NT2 in {NT, NT - 1}.
// nt3x = 3 / 6 / 9: one-tile instances in which the recurrence wave owns residual tiles as well (NT in 8..11; 3, 9: NT2 == NT,
// 6: NT2 == NT - 1); -1 / 0: none (the default for one tile per workgroup)
int aq_la_launch_tt1(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a);
// nt3x = 9: the instance with nine residual tiles on the recurrence wave (NT2 == NT, NT in 8..11); -1: aq_la_nt3's count
int aq_la_launch_tt2(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a);
// the MASK instances (Y with missing values; one trait tile per workgroup): aq_launch_la1m.hip
// nt3x = 3: three residual tiles on the recurrence wave (NT2 == NT, NT in 8..11); -1: none
int aq_la_launch_mask(int NT, int NT2, int nt3x, bool seg, unsigned grid, hipStream_t st, const AqCoreArgs &a);
