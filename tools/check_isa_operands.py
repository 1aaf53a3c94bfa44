"""Build-time proof obligation for the hand-issued vector-memory loads of aq_core_sweep_la.h.

The X operand stream (AQ_LD2 / AQ_LD2I) and the L2 warm-up touches are requested by inline-asm `global_load_*` whose destination
VGPRs are written when the data RETURNS -- long after the asm statement that names them as outputs, which the compiler cannot
know.  Between a request and the `s_waitcnt vmcnt(N)` that covers it the compiler may therefore do anything it likes with those
registers: copy them elsewhere (a live-range split: the copy holds stale bits, the original register is handed to another value
which the returning load then overwrites -- an address, say: a GPU fault), spill them, or re-use them outright.

This tool proves, per kernel of an assembly listing (hipcc -S --cuda-device-only), that none of that happened:

  * the kernel is cut into basic blocks and walked along its real control flow (every s_branch / s_cbranch_* edge, loop
    back-edges included) with a work list over (block, queue-of-outstanding-vector-memory-operations) states;
  * the queue replays `vmcnt` over the REAL instruction stream: every global / scratch / buffer / flat load, store, atomic and
    LDS-DMA transfer enters it -- the compiler's own loads and spill traffic too -- and every `s_waitcnt vmcnt(N)` (hand-written
    or inserted by the compiler) retires the oldest entries until N are left (gfx9: loads and stores retire in issue order);
  * while a hand-issued request sits in the queue, ANY instruction that reads or writes one of its destination VGPRs is a
    violation (a read sees stale bits, a write races with the returning data), whatever it is: v_mov, MFMA, scratch_store,
    ds_read destination, another request;
  * a window that reaches s_endpgm still open is reported as well (harmless to the hardware, but it means a wait is missing).

usage: python tools/check_isa_operands.py [--quiet] file.s [...]     exit code 1 on any violation."""
import re
import sys
from collections import deque

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
KERNEL = re.compile(r"^(_Z\w*aq_core_sweep_la_kernel\w*):")
FUNC_END = re.compile(r"^\.Lfunc_end\d+:")
VREG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
VMEM = re.compile(r"^(global|scratch|buffer|flat|tbuffer)_(load|store|atomic)")
BRANCH = re.compile(r"^s_(branch|cbranch_\w+)\s+(\.LBB\d+_\d+)")
CAP = 63   # vmcnt is a 6-bit counter: the hardware never has more in flight


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


class Ins:
    __slots__ = ("line", "op", "text", "regs", "asm", "vmem", "dest", "wait", "branch", "end")


def parse_kernel(lines, lo, hi):
    """-> list of basic blocks [(label or None, [Ins])], label -> block index"""
    blocks, cur, labels = [], [], {}
    cur_label = None
    in_asm = False
    for i in range(lo, hi):
        raw = lines[i]
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        m = LABEL.match(raw)
        if m:
            if cur or cur_label is not None:
                blocks.append((cur_label, cur))
            cur, cur_label = [], m.group(1)
            continue
        code = raw.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        ins = Ins()
        ins.line, ins.text, ins.asm = i + 1, code, in_asm
        ins.op = code.split()[0]
        ins.regs = vregs(code)
        ins.vmem = bool(VMEM.match(ins.op))
        ins.dest = None
        if ins.vmem and ins.asm and "_load" in ins.op and "lds" not in ins.op:
            first = code[len(ins.op):].split(",")[0]
            d = vregs(first)
            ins.dest = (min(d), max(d)) if d else None
        ins.wait = None
        if ins.op == "s_waitcnt":
            mm = VMCNT.search(code)
            if mm:
                ins.wait = int(mm.group(1))
        mb = BRANCH.match(code)
        ins.branch = (mb.group(1), mb.group(2)) if mb else None
        ins.end = ins.op == "s_endpgm"
        cur.append(ins)
        if ins.branch or ins.end:
            blocks.append((cur_label, cur))
            cur, cur_label = [], None
    if cur or cur_label is not None:
        blocks.append((cur_label, cur))
    for k, (lab, _) in enumerate(blocks):
        if lab is not None:
            labels[lab] = k
    return blocks, labels


def check_kernel(name, lines, lo, hi, quiet):
    blocks, labels = parse_kernel(lines, lo, hi)
    viol = {}          # (line of the offending instruction, line of the request) -> text
    open_at_end = set()
    requests = set()
    seen = set()
    visited = set()
    work = deque([(0, ())])
    nstates = 0
    while work:
        b, q = work.pop()
        if (b, q) in seen:
            continue
        seen.add((b, q))
        visited.add(b)
        nstates += 1
        q = list(q)
        fall = True
        for ins in blocks[b][1]:
            if ins.wait is not None:
                while len(q) > ins.wait:
                    q.pop(0)
                continue
            if ins.regs:
                for e in q:
                    if e is not None and any(e[0] <= r <= e[1] for r in ins.regs):
                        viol.setdefault((ins.line, e[2]), f"line {ins.line}: `{ins.text}` touches v[{e[0]}:{e[1]}] while the request of line {e[2]} is in flight")
            if ins.vmem:
                if ins.dest is not None:
                    requests.add(ins.line)
                    q.append((ins.dest[0], ins.dest[1], ins.line))
                else:
                    q.append(None)
                if len(q) > CAP:
                    q.pop(0)
            if ins.end:
                for e in q:
                    if e is not None:
                        open_at_end.add(e[2])
                fall = False
            elif ins.branch:
                kind, target = ins.branch
                if target in labels:
                    work.append((labels[target], tuple(q)))
                if kind == "branch":
                    fall = False
        if fall and b + 1 < len(blocks):
            work.append((b + 1, tuple(q)))
    tag = name.replace("_Z23aq_core_sweep_la_kernelI", "").replace("EEv10AqCoreArgs", "")
    unreached = len(blocks) - len(visited)
    print(f"{tag}: {len(requests)} hand-issued requests, {len(blocks)} blocks ({unreached} unreached), {nstates} (block, queue) states, "
          f"{len(viol)} violations, {len(open_at_end)} windows open at s_endpgm")
    if not quiet or viol:
        for k in sorted(viol):
            print("    " + viol[k])
    for ln in sorted(open_at_end):
        print(f"    request of line {ln} is never covered by a wait before s_endpgm")
    return len(viol) + len(open_at_end) + unreached      # a block the walk cannot reach is a block that was not checked


def check(path, quiet):
    lines = open(path).read().split("\n")
    bad = 0
    i = 0
    while i < len(lines):
        m = KERNEL.match(lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not FUNC_END.match(lines[j]):
                j += 1
            bad += check_kernel(m.group(1), lines, i + 1, j, quiet)
            i = j
        i += 1
    return bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--quiet"]
    total = sum(check(p, "--quiet" in sys.argv) for p in args)
    print(f"total: {total} findings")
    sys.exit(1 if total else 0)
