"""Host logic of the task-DAG factorization schedule (csrc/dag.hip: dag_build_tasks), without a GPU: the task list the
persistent bulk kernel consumes IN ORDER.  Progress of the schedule rests on two properties of that list -- every task waits
only for tasks in front of it (or for the pivot chain, which waits only for tasks whose operands the chain itself has
already produced), and the chunks of a tile cover its columns exactly once -- checked here for several matrix orders."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madnlp_jl_amd import _lib as L  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from dag_tasks import dag_tasks  # noqa: E402

BAND, FINAL, FIRST = 1, 2, 4


def _tasks(ntile, chunk, band_tiles, js2, taper0=1):
    cap = 400000
    out = np.zeros(4 * cap, dtype=np.int32)
    n1 = C.c_int(0)
    n = L.lib().mnk_debug_dag_tasks(ntile, chunk, band_tiles, js2, taper0, out.ctypes.data, cap, C.byref(n1))
    assert 0 <= n <= cap
    t = out[: 4 * n].reshape(n, 4)
    return [(int(a) & 255, int(a) >> 8, int(i), int(j), int(k) & 0xffff, int(k) >> 16) for a, i, j, k in t], n1.value


@pytest.mark.parametrize("ntile,chunk,band_tiles,js2,taper0", [(16, 64, 8, 8, 1), (17, 12, 8, 0, 1), (40, 64, 8, 20, 2), (88, 64, 8, 44, 1),
                                                               (88, 64, 8, 44, 2), (88, 12, 8, 44, 1), (45, 7, 6, 10, 3), (128, 64, 8, 64, 2)])
def test_task_list_covers_every_tile_once_and_waits_only_backwards(ntile, chunk, band_tiles, js2, taper0):
    ts, n1 = _tasks(ntile, chunk, band_tiles, js2, taper0)
    pos = {}
    tiles = {}
    closed = {}           # (I, J) -> queue index of the task that closes the bulk tile (publishes the row's front)
    for idx, (flags, q, I, J, kb, ke) in enumerate(ts):
        assert 0 <= J <= I < ntile and 0 <= kb <= ke
        tiles.setdefault((I, J), []).append((q, kb, ke, flags, idx))
        if (flags & FINAL) and not (flags & BAND):
            closed[(I, J)] = idx
    nsc = (ntile + 1) // 2
    for (I, J), lst in tiles.items():
        Js = J // 2
        bt = ntile if Js >= js2 else band_tiles
        band = I < 2 * Js + bt
        K = (2 * Js - 2) if band else J     # band tiles: the chain applies the previous strip-column itself
        lst.sort()
        # chunk indices 0, 1, ... in queue order (the kernel applies the chunks of a tile in this order) and contiguous cover
        assert [q for q, *_ in lst] == list(range(len(lst)))
        assert [x[4] for x in lst] == sorted(x[4] for x in lst)
        assert lst[0][1] == 0 and lst[-1][2] == K
        for a, b in zip(lst, lst[1:]):
            assert a[2] == b[1]
        assert (lst[0][3] & FIRST) and all(not (x[3] & FIRST) for x in lst[1:])
        assert (lst[-1][3] & FINAL) and all(not (x[3] & FINAL) for x in lst[:-1])
        assert all(bool(x[3] & BAND) == band for x in lst)
        # operands: tile columns [kb, ke) of rows I and J.  A column k of a row below the band is final once that row's tile
        # (row, k) is closed -- by a task that must sit in FRONT of this one; rows inside the band are the chain's.
        for q, kb, ke, flags, idx in lst:
            for row in (I, J):
                for k in range(kb, ke):
                    ks = k // 2
                    bt_k = ntile if ks >= js2 else band_tiles
                    if row >= 2 * ks + bt_k:
                        assert closed[(row, k)] < idx, ((I, J), row, k)
    # every tile below the diagonal band structure appears: bulk tiles for all J, band tiles once there is something to accumulate
    for J in range(ntile):
        Js = J // 2
        bt = ntile if Js >= js2 else band_tiles
        for I in range(2 * Js + bt, ntile):
            assert (I, J) in tiles
        if 2 * Js - 2 > 0:
            for I in range(max(2 * Js, J), min(2 * Js + bt, ntile)):
                assert (I, J) in tiles
    # first-phase tasks come first and need nothing the second phase's chain produces
    assert all(ke <= 2 * js2 for _, _, _, _, _, ke in ts[:n1])
    # the Python mirror the trace tools use is the same list
    mirror = dag_tasks(ntile, chunk, band_tiles, js2, taper0)
    assert [(f, q, I, J, kb, ke) for (_, _, J, I, f, q, kb, ke) in mirror] == ts


FILL = 8


def _merged(ntile, chunk, band_tiles, js2, taper0, fill, ninst, period):
    cap = 2000000
    out = np.zeros(4 * cap, dtype=np.int32)
    n = L.lib().mnk_debug_dag_merged_tasks(ntile, chunk, band_tiles, js2, taper0, int(fill), ninst, period, out.ctypes.data, cap)
    assert 0 <= n <= cap
    t = out[: 4 * n].reshape(n, 4)
    return [(int(a) & 255, int(a) >> 8, int(i) & 0xffff, int(i) >> 16, int(j), int(k) & 0xffff, int(k) >> 16) for a, i, j, k in t]


@pytest.mark.parametrize("ntile,chunk,band_tiles,js2,taper0,ninst,period", [(16, 64, 8, 8, 2, 3, 8), (40, 64, 8, 20, 2, 5, 20), (88, 64, 8, 44, 2, 16, 44),
                                                                            (45, 7, 6, 23, 3, 4, 23), (88, 64, 8, 44, 2, 2, 60)])
def test_merged_queue_of_a_batch(ntile, chunk, band_tiles, js2, taper0, ninst, period):
    """mnk_factorize_batch_*: ONE queue for several factorizations of the same order.  (1) Every instance's sub-sequence is the
    single-instance list (so the arithmetic per tile, and the order of its chunks, are those of a lone factorization:
    bit-identical factors) with the zero-fill tasks of the next factorization's buffer: every lower tile once.  (2) Two
    pivot chains run at a time -- chain i + 2 follows chain i on its stream -- so every task of instance i must sit in front
    of every task of instance i + 2 (a workgroup that holds a task of i + 2 may have to wait for the end of chain i, and
    chain i for tasks of instance i: they must not be stuck behind it).  (3) The instances overlap: tasks of i + 1 start
    before the tasks of i end."""
    base, _ = _tasks(ntile, chunk, band_tiles, js2, taper0)
    one = _merged(ntile, chunk, band_tiles, js2, taper0, True, 1, period)
    assert [x[:3] + x[4:] for x in one if not (x[0] & FILL)] == base
    fills = [(x[2], x[4]) for x in one if x[0] & FILL]
    assert sorted(fills) == sorted((i, j) for j in range(ntile) for i in range(j, ntile)) and len(set(fills)) == len(fills)
    assert all(x[3] == 0 for x in one)
    # fills are spread over the first phase, not bunched: no more than a few in a row
    run = mx = 0
    for x in one:
        run = run + 1 if x[0] & FILL else 0
        mx = max(mx, run)
    assert mx <= 1 + (len(fills) + len(base) - 1) // max(1, len(base))
    m = _merged(ntile, chunk, band_tiles, js2, taper0, True, ninst, period)
    assert len(m) == ninst * len(one)
    first, last = {}, {}
    for idx, x in enumerate(m):
        first.setdefault(x[3], idx)
        last[x[3]] = idx
    for i in range(ninst):
        sub = [x[:3] + (0,) + x[4:] for x in m if x[3] == i]
        assert sub == one, i
    for i in range(ninst - 2):
        assert last[i] < first[i + 2], (i, last[i], first[i + 2])
    if period < ntile:
        for i in range(ninst - 1):
            assert first[i + 1] < last[i]


def test_persistent_grids_are_sized_by_what_is_placed_at_launch():
    """`mnk_debug_grid_at_launch` (host arithmetic behind the bulk kernel's grid, csrc/ls.hip): the dispatcher deals a grid out
    evenly per (XCD, shader engine) -- mask bit b is XCD b % 8, engine (b / 8) % 4 -- so the engine with the fewest CUs under the
    mask bounds what starts at launch.  Workgroups placed later, in mid-kernel, were the rare time-out of the task-DAG schedule
    (DESIGN.md section 8 item -1)."""
    f = L.lib().mnk_debug_grid_at_launch
    assert f(0, 256, 16, 3) == 3 * 7 * 32 == 672      # beside the 16-CU chain: engines 0 and 1 of every XCD keep 7 CUs
    assert f(0, 256, 32, 3) == 3 * 224                # the batch mask (two chain partitions): one CU of every engine, balanced
    assert f(0, 256, 96, 3) == 3 * 160                # the deep-band mask: three CUs of every engine
    assert f(0, 256, 64, 3) == 3 * 192 and f(0, 256, 128, 3) == 3 * 128   # small-batch partitions: multiples of 64
    assert f(0, 256, 0, 1) == 256
    assert f(128, 128, 16, 3) == 3 * 3 * 32           # a half-device partition (CUs 16..31 of every XCD) with its own chain
    assert f(0, 256, 8, 3) == 3 * 7 * 32              # one engine short of a CU is enough
    assert f(0, 64, 16, 3) == 3 * 1 * 32              # a 64-CU partition: engines 0 and 1 are left ONE CU each
    assert f(0, 256, 256, 3) == 3                     # (an empty mask: never zero)
