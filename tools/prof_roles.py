"""Summary of the per-role cycle counters of a -DAQ_DIAG_TIME build (make -C atlasqtl_amd/csrc diag):
AQ_LIB=atlasqtl_amd/libatlasqtl_hip_diag.so AQ_DIAG_DUMP=gpurun_out/roles.txt python bench.py ... ; python tools/prof_roles.py gpurun_out/roles.txt
Each line of the dump: workgroup, wave, cycles waiting on the matrix / recurrence counters, cycles waiting on the helper's and
the stagger counters, cycles in total (sweep 15)."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
d = d[d[:, 4] > 0]
roles = {"matrix 0-2": [0, 1, 2], "matrix 4-6": [4, 5, 6], "recurrence (3)": [3], "helper (7)": [7]}
print(f"{len(set(d[:, 0]))} workgroups; total cycles per workgroup: mean {d[:, 4].mean():.0f}")
for name, ws in roles.items():
    m = np.isin(d[:, 1], ws)
    wa, wb, tot = d[m, 2].astype(float).sum(), d[m, 3].astype(float).sum(), d[m, 4].astype(float).sum()
    print(f"{name:16s} busy {100 * (1 - (wa + wb) / tot):5.1f} %   waiting on matrix/recurrence {100 * wa / tot:5.1f} %   on helper/stagger {100 * wb / tot:5.1f} %")
