"""Timeline of workgroup 0 from a -DAQ_DIAG_TIME build (AQ_DIAG_DUMP=f writes f.timeline): per wave and phase the cycle counter
at phase entry, after the recurrence wave's chain, after the waits (tile loop starts) and after the S' store + announcement.
usage: python tools/prof_timeline.py f.timeline [first_phase n_phases]"""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
p0 = int(sys.argv[2]) if len(sys.argv) > 2 else 70
npz = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = d[(d[:, 1] == p0) & (d[:, 2] > 0), 2].min()
ghz = 2.4   # shader clock the counter ticks with on this pool (profiles/r01_f64_clock.txt)
print("times in us after the first wave entered phase", p0)
print("wave phase   enter  chain_done  loop_start  announced   (loop+store)")
for w in range(8):
    for ph in range(p0, p0 + npz):
        r = d[(d[:, 0] == w) & (d[:, 1] == ph)]
        if len(r) == 0 or r[0, 2] == 0:
            continue
        e, c, l, a = [(x - t0) / ghz / 1e3 for x in r[0, 2:6]]
        extra = ""
        if r.shape[1] >= 9 and r[0, 6] > 0:     # recurrence wave: S' summed, correction applied, half of the 16 steps
            s4, s5, s6 = [(x - t0) / ghz / 1e3 for x in r[0, 6:9]]
            extra = f"   chain: S' ready {s4 - e:5.2f}  corrected {s5 - e:5.2f}  8 steps {s6 - e:5.2f}  16 steps {c - e:5.2f}"
        print(f"{w:4d} {ph:5d} {e:8.2f} {c:10.2f} {l:10.2f} {a:10.2f}   {a - l:8.2f}{extra}")
periods = []
for w in (0, 3, 4):
    r = d[(d[:, 0] == w) & (d[:, 5] > 0)]
    if len(r) > 2:
        periods.append((w, np.diff(r[:, 5]).mean() / ghz / 1e3))
print("mean period per phase (us):", ", ".join(f"wave {w}: {p:.2f}" for w, p in periods))
