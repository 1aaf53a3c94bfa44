"""aq_vb_partition (the trait sharding of aq_vb_run_multi, include/atlasqtl_hip.h): whole 16-trait tiles, contiguous, complete,
balanced to one tile -- no GPU needed."""
import ctypes as C

import pytest

from atlasqtl_amd import _lib


def part(q, n, r):
    k0, k1 = C.c_int32(), C.c_int32()
    rc = _lib.lib().aq_vb_partition(q, n, r, C.byref(k0), C.byref(k1))
    return rc, k0.value, k1.value


@pytest.mark.parametrize("q", [1, 15, 16, 17, 50, 1000, 10000, 20000, 12345])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8])
def test_partition_covers_q_in_whole_tiles(q, n):
    ntile = (q + 15) // 16
    if n > ntile:
        assert part(q, n, 0)[0] == 1                      # AQ_ERR_ARG: more parts than tiles
        assert b"more parts" in _lib.lib().aq_last_error()
        return
    cuts = [part(q, n, r) for r in range(n)]
    assert all(c[0] == 0 for c in cuts)
    assert cuts[0][1] == 0 and cuts[-1][2] == q
    sizes = []
    for a, b in zip(cuts, cuts[1:]):
        assert a[2] == b[1] and a[2] % 16 == 0            # contiguous, shards start on tile boundaries
    for c in cuts:
        assert c[2] > c[1]
        sizes.append((c[2] - c[1] + 15) // 16)
    assert max(sizes) - min(sizes) <= 1                   # balanced to one tile


def test_partition_argument_errors():
    assert part(0, 1, 0)[0] == 1 and part(10, 0, 0)[0] == 1 and part(10, 2, 2)[0] == 1 and part(100, 2, -1)[0] == 1
    assert _lib.lib().aq_vb_partition(10, 1, 0, None, None) == 1


def test_c3_c5_shards():
    """BASELINE configs C4 (q = 10 000 on 8 GPUs) and C5 (q = 20 000): 78 / 79 and 156 / 157 tiles per GPU."""
    for q, lo, hi in ((10000, 78, 79), (20000, 156, 157)):
        t = [(part(q, 8, r)[2] - part(q, 8, r)[1] + 15) // 16 for r in range(8)]
        assert min(t) == lo and max(t) == hi and sum(t) == (q + 15) // 16
