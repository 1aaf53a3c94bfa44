"""Build-time check for the hand-issued operand loads of aq_core_sweep_la.h.

The X operand stream is requested by inline-asm `global_load_dwordx4 v[a:b], v, s[c:d]` whose destination registers are written
when the data returns -- long after the asm statement, which the compiler cannot know.  If it spilled (or reloaded into) one of
those registers between the request and the first MFMA that consumes it, the returning load would overwrite an unrelated value
(or the MFMA would read a stale one).  This script scans the ISA of every kernel in an assembly file (hipcc -S
--cuda-device-only): for every such load it walks forward to the first v_mfma that reads one of the destination registers and
reports any scratch_store / scratch_load that touches them on the way.

usage: python tools/check_isa_spills.py file.s [...]      exit code 1 if any window has a hit"""
import re
import sys

LOAD = re.compile(r"\s+global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]")
SCR = re.compile(r"\s+scratch_(store|load)_dword(?:x\d)? (?:off, )?v(?:\[(\d+):(\d+)\]|(\d+))")
MFMA = re.compile(r"\s+v_mfma_f64_16x16x4_f64 v\[\d+:\d+\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]")
WINDOW = 1500   # lines: a request is consumed within a few tile steps


def check(path):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and "aq_core_sweep_la_kernel" in l and l.rstrip().endswith("AqCoreArgs")]
    starts += [len(lines)]
    bad = 0
    for k in range(len(starts) - 1):
        lo, hi = starts[k], starts[k + 1]
        nload = nhit = nopen = 0
        for i in range(lo, hi):
            m = LOAD.match(lines[i])
            if not m:
                continue
            nload += 1
            a, b = int(m.group(1)), int(m.group(2))
            consumed = False
            for j in range(i + 1, min(hi, i + WINDOW)):
                mm = MFMA.match(lines[j])
                if mm and (a <= int(mm.group(1)) <= b or a <= int(mm.group(3)) <= b):
                    consumed = True
                    break
                ms = SCR.match(lines[j])
                if ms:
                    r = set(range(int(ms.group(2)), int(ms.group(3)) + 1)) if ms.group(2) else {int(ms.group(4))}
                    if r & set(range(a, b + 1)):
                        nhit += 1
                        print(f"    line {j + 1}: scratch_{ms.group(1)} touches v[{a}:{b}] requested at line {i + 1}")
                ml = LOAD.match(lines[j])
                if ml and int(ml.group(1)) == a:
                    break   # requested again (a dangling prefetch that was never consumed: end of the sweep)
                if re.match(r"\s+s_waitcnt vmcnt\(0\)", lines[j]):
                    break   # every request has returned: the registers hold their data (or are free again)
            nopen += not consumed
        print(f"{lines[lo].split(':')[0][:80]}: {nload} operand requests, {nhit} spill hits between request and use, "
              f"{nopen} never consumed (dangling at the end of a sweep)")
        bad += nhit
    return bad


if __name__ == "__main__":
    total = sum(check(p) for p in sys.argv[1:])
    sys.exit(1 if total else 0)
