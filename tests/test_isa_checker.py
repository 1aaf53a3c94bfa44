"""tools/check_isa_operands.py, the build-time proof that no instruction touches the destination registers of a hand-issued
vector-memory load while that load is in flight (aq_core_sweep_la.h): the checker itself must see what it claims to see.
Synthetic listings in the shape hipcc -S emits, one hazard each, walked along real control flow (no GPU, no compiler)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_operands as C  # noqa: E402

HEAD = "_Z23aq_core_sweep_la_kernelILi1ELi1ELb0ELi1ELb0EEv10AqCoreArgs: ; @kernel\n; %bb.0:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"
REQ = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v52, s[4:5]\n\tglobal_load_dwordx4 v[14:17], v52, s[4:5] offset:1024\n\t;;#ASMEND\n"
WAIT0 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"
USE = "\tv_mfma_f64_16x16x4_f64 v[20:27], v[10:11], v[30:31], v[20:27]\n"


def run(body):
    lines = (HEAD + body + TAIL).split("\n")
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        bad = C.check_kernel("k", lines, 1, len(lines) - 2, True)
    return bad, buf.getvalue()


def test_clean_request_wait_use():
    bad, out = run(REQ + "\tv_add_f64 v[40:41], v[42:43], v[44:45]\n" + WAIT0 + USE)
    assert bad == 0, out


def test_register_ranges_are_parsed():
    assert C.vregs("global_load_dwordx4 v[46:49], v52, s[64:65] offset:1024") == {46, 47, 48, 49, 52}
    assert C.vregs("v_mfma_f64_16x16x4_f64 v[18:25], v[38:39], v[10:11], 0") == set(range(18, 26)) | {38, 39, 10, 11}
    assert C.vregs("s_mov_b32 s5, s13") == set()


@pytest.mark.parametrize("hazard", [
    "\tv_mov_b32_e32 v99, v12\n",                                   # live-range split: a copy of stale bits
    "\tscratch_store_dwordx2 off, v[10:11], off offset:16\n",        # spill of an in-flight destination
    "\tv_mov_b32_e32 v15, 0\n",                                     # the register handed to another value
    "\tds_read_b64 v[16:17], v3\n",                                 # an LDS read landing in it
    USE,                                                            # consumed before the wait
])
def test_touch_between_request_and_wait_is_flagged(hazard):
    bad, out = run(REQ + hazard + WAIT0 + USE)
    assert bad >= 1 and "while the request" in out


def test_wait_count_is_replayed_over_the_real_stream():
    # two requests (4 loads): vmcnt(2) covers the first request only
    req2 = REQ.replace("v[10:13]", "v[60:63]").replace("v[14:17]", "v[64:67]")
    use2 = "\tv_mfma_f64_16x16x4_f64 v[20:27], v[60:61], v[30:31], v[20:27]\n"
    w2 = "\ts_waitcnt vmcnt(2)\n"
    assert run(REQ + req2 + w2 + USE + WAIT0 + use2)[0] == 0
    assert run(REQ + req2 + w2 + use2 + WAIT0)[0] >= 1          # second request still in flight at its use
    # a compiler-issued load or a spill between request and wait only makes the hand-counted wait conservative
    assert run(REQ + req2 + "\tscratch_store_dword off, v200, off\n" + w2 + USE + WAIT0 + use2)[0] == 0
    assert run(REQ + "\tglobal_load_dwordx2 v[80:81], v[82:83], off\n\ts_waitcnt vmcnt(1)\n" + USE + WAIT0)[0] == 0
    # ... whereas a wait that retires only an EARLIER compiler load does not cover the request
    assert run("\tglobal_load_dwordx2 v[80:81], v[82:83], off\n" + REQ + "\ts_waitcnt vmcnt(2)\n" + USE + WAIT0)[0] >= 1


def test_back_edge_is_followed():
    """A request left dangling at the end of the loop body is consumed at the top of the next iteration: clean with the wait,
    flagged without it -- a linear scan sees neither."""
    def loop(wait):
        return (REQ + ".LBB0_1:\n" + wait + USE + "\tv_add_u32_e32 v1, 1, v1\n" + REQ +
                "\ts_cbranch_scc1 .LBB0_1\n; %bb.2:\n" + WAIT0)
    assert run(loop(WAIT0))[0] == 0
    bad, out = run(loop(""))
    assert bad >= 1
    # copy on the back edge (phi elimination): flagged
    body = (REQ + ".LBB0_1:\n" + WAIT0 + USE + REQ + "\tv_mov_b32_e32 v70, v10\n\ts_cbranch_scc1 .LBB0_1\n; %bb.2:\n" + WAIT0)
    assert run(body)[0] >= 1


def test_open_window_at_end_is_reported():
    bad, out = run(REQ)
    assert bad >= 1 and "never covered" in out


def test_branches_both_ways():
    # the hazard sits only on the taken path
    body = (REQ + "\ts_cbranch_vccz .LBB0_3\n; %bb.1:\n" + WAIT0 + USE + "\ts_branch .LBB0_4\n.LBB0_3:\n" + USE + WAIT0 +
            ".LBB0_4:\n")
    assert run(body)[0] >= 1


def test_cli_exit_code(tmp_path):
    good, badf = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(HEAD + REQ + WAIT0 + USE + TAIL)
    badf.write_text(HEAD + REQ + USE + WAIT0 + TAIL)
    tool = os.path.join(ROOT, "tools", "check_isa_operands.py")
    assert subprocess.run([sys.executable, tool, "--quiet", str(good)], capture_output=True).returncode == 0
    assert subprocess.run([sys.executable, tool, "--quiet", str(badf)], capture_output=True).returncode == 1
