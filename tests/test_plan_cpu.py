"""plan.py on the CPU: read / write sets -> dependencies -> two-lane schedule -> race check, on synthetic launches (the C entry points are only
looked up, never called)."""
import ctypes as C

import pytest
import torch

from audio_inpainting_diffusion_amd import _lib
from audio_inpainting_diffusion_amd.plan import Plan, _disjoint, _range


class _P(C.Structure):
    _fields_ = [("x", C.c_int)]


def _plan(ops):
    """ops: (lane, reads, writes) per launch"""
    pl = Plan()
    for lane, rd, wr in ops:
        pl.lane = lane
        pl.add("aid_add2", _P(), *rd, *wr, writes=tuple(wr))
    return pl


def test_view_disjointness():
    X = torch.zeros(2, 8, 12, 16)
    r = _range
    assert _disjoint(r(X[:, :, :4]), r(X[:, :, 4:])) and _disjoint(r(X[:, :4]), r(X[:, 4:])) and _disjoint(r(X[:, :, :, :8]), r(X[:, :, :, 8:]))
    assert not _disjoint(r(X[:, :, :4]), r(X[:, :, 3:6])) and not _disjoint(r(X[:, :, :4]), r(X[:, :4])) and not _disjoint(r(X[:, :, :4]), r(X))
    assert not _disjoint(r(X[:, :, :4]), r(X.view(2, 8, 192)[:, :, :64]))      # different stride tuples: conservatively overlapping
    assert not _disjoint(r(X[0:1].expand(2, -1, -1, -1)), r(X[1:2]))           # stride 0 (broadcast view): conservative


def test_dependencies_raw_war_waw_and_disjoint_slices():
    a, b, c = torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(2, 4, 6, 8)
    lo, hi = c[:, :, :3], c[:, :, 3:]
    pl = _plan([(0, [a], [b]),          # 0: b = f(a)
                (0, [b], [lo]),         # 1: RAW on b
                (0, [a], [hi]),         # 2: writes the OTHER rows of c: independent of 1
                (0, [c], [a]),          # 3: reads all of c (RAW on 1 and 2), writes a (WAR on 0 and 2)
                (0, [a], [b])])         # 4: WAW on b (0) + WAR (1 read b) + RAW on a (3)
    d = pl.dependencies()
    assert d[0] == [] and d[1] == [0] and d[2] == []
    assert d[3] == [0, 1, 2]
    assert d[4] == [0, 1, 3]


def test_two_lane_schedule_inserts_only_the_needed_events_and_check_catches_a_missing_one():
    a, b, c, d_, e = (torch.zeros(16) for _ in range(5))
    pl = _plan([(0, [a], [b]),          # 0 trunk
                (1, [a], [c]),          # 1 lane 1, independent of 0
                (1, [c], [d_]),         # 2 lane 1 (stream order covers 1 -> 2)
                (0, [b, d_], [e]),      # 3 trunk joins: needs 2 (lane 1) -> event
                (1, [e], [c]),          # 4 lane 1 needs 3 (lane 0) -> event; WAR on c vs 2 is lane order
                (0, [c], [a])])         # 5 trunk needs 4 -> event; WAR on a vs 1 (lane 1, before 4): covered transitively
    pl.lanes = 2
    waits, record, deps = pl.schedule()
    assert waits == [[], [], [], [2], [3], [4]] and record == {2, 3, 4}
    assert pl.check() == 3
    waits[4].clear()
    with pytest.raises(AssertionError, match="may run before"):
        pl.check()


def test_also_writes_adds_a_dependency_after_the_fact():
    a, b, ws = torch.zeros(8), torch.zeros(8), torch.zeros(8, dtype=torch.float64)
    pl = _plan([(0, [a], [b]), (1, [ws], [a])])
    assert pl.dependencies()[1] == [0]                       # WAR on a
    pl2 = _plan([(0, [a], [b]), (1, [ws], [torch.zeros(8)])])
    assert pl2.dependencies()[1] == []
    pl2.ops[0].also_writes(ws)                               # the first launch's epilogue was patched to write ws (epilogue statistics)
    pl2._sched = None
    assert pl2.dependencies()[1] == [0]
