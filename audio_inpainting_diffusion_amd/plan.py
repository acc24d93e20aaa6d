"""Static launch plans: a flat list of C-ABI calls with pre-filled parameter structs, their read / write sets, and a two-lane schedule.

Every op records which tensors it reads and which it writes (bounding address ranges per storage).  From those the plan derives the
data dependencies between ops (read-after-write, write-after-read, write-after-write) -- used for two things:

  * ``schedule``: ops are tagged with a LANE when they are emitted (lane 0 = the trunk of the U-Net; lane 1 = the per-octave init blocks,
    the pyramid path and the out blocks, whose few-channel launches are latency-bound at the reference's batch size of 1).  Lanes run on
    separate HIP streams (inside a captured HIP graph they become parallel branches); every dependency that crosses lanes gets an event.
    Within a lane, stream order covers the dependencies.  Results are bit-identical to the single-lane order: only the overlap changes.
  * ``check``: a race detector for the schedule itself (tests): every dependency must be covered by stream order or by an event chain.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib


def _range(t: torch.Tensor):
    """(storage key, first byte, one past the last byte, view signature) of the bounding range of a strided view.  The signature --
    (element offset in the storage, shape, strides) -- lets ``_disjoint`` tell apart slices of one buffer whose bounding ranges interleave
    (the U-Net writes octave rows / channel halves of the same concatenation buffer from different producers)."""
    lo = t.data_ptr()
    ext = 1
    for n, s in zip(t.shape, t.stride()):
        if n == 0:
            return (t.untyped_storage().data_ptr(), lo, lo, None)
        ext += (n - 1) * abs(s)
    return (t.untyped_storage().data_ptr(), lo, lo + ext * t.element_size(), (t.storage_offset(), tuple(t.shape), tuple(t.stride())))


def _disjoint(a, b) -> bool:
    """True when two views of the same storage provably share no element: same strides, nested (each stride covers the extent of the inner
    dimensions), and their index boxes miss each other along some dimension."""
    sa, sb = a[3], b[3]
    if sa is None or sb is None or sa[2] != sb[2] or len(sa[1]) != len(sb[1]):
        return False
    strides = sa[2]
    order = sorted(range(len(strides)), key=lambda d: -strides[d])
    if any(strides[d] <= 0 for d in order):
        return False
    oa, ob = sa[0], sb[0]
    ia, ib = {}, {}
    for d in order:
        ia[d], oa = divmod(oa, strides[d])
        ib[d], ob = divmod(ob, strides[d])
    if oa or ob:
        return False
    for k, d in enumerate(order):                     # nesting: the box of the inner dimensions must fit inside one step of dimension d
        inner = order[k + 1:]
        for idx, shp in ((ia, sa[1]), (ib, sb[1])):
            if sum((idx[e] + shp[e] - 1) * strides[e] for e in inner) >= strides[d]:
                return False
    return any(ia[d] + sa[1][d] <= ib[d] or ib[d] + sb[1][d] <= ia[d] for d in order)


# entry points that take an aid_conv2d_params block (the two launches of the 2-D Winograd form are separate plan nodes)
CONV_OPS = ("aid_conv2d", "aid_conv2d_wino2d_gemm", "aid_conv2d_wino2d_output")


class Op:
    __slots__ = ("fn", "addr", "name", "flops", "nbytes", "descr", "lane", "reads", "writes", "params")

    def __init__(self, fn, addr, name, flops, nbytes, descr, lane, reads, writes, params):
        self.fn, self.addr, self.name, self.flops, self.nbytes, self.descr = fn, addr, name, flops, nbytes, descr
        self.lane, self.reads, self.writes, self.params = lane, reads, writes, params

    def also_reads(self, *tensors):
        """a later emitter patched this op's parameter struct so that it reads one more tensor"""
        self.reads.extend(_range(t) for t in tensors if t is not None)

    def also_writes(self, *tensors):
        """a later emitter patched this op's parameter struct so that it writes one more tensor (epilogue statistics, fused copies)"""
        self.writes.extend(_range(t) for t in tensors if t is not None)


class Plan:
    """Flat list of ops executed in order on the current stream (lanes = 1) or on two streams with event edges (lanes = 2)."""

    def __init__(self):
        self.ops: List[Op] = []
        self.keep = []             # structs + tensors referenced by raw pointers
        self.flops = 0
        self.lane = 0              # lane given to the ops added next (set by the builder)
        self.lanes = 1             # streams used by run(): 1 = everything in order on the current stream
        self.timing = None         # set to a list to bracket every conv launch with HIP events (bench.py roofline)
        self.trace = None          # set to a list to bracket EVERY launch with HIP events (tools/plan_trace.py)
        self._sched = None
        self.side_stream = None    # stream of lane 1 (the owner may share one between plans that never run at the same time)
        self.zero_on_fail = []     # scratch whose contract is "zero before a launch, left zero by it" (arrival counters, split-K flags): re-zeroed when a launch fails

    # ---- construction ------------------------------------------------------------------------------------------------------
    def add(self, name, params, *tensors, flops=0, nbytes=0, writes: Sequence[torch.Tensor] = ()):
        """tensors: everything the launch touches through raw pointers (kept alive); writes: the subset it writes (the rest is read)."""
        fn = getattr(_lib.lib(), name)
        if name in CONV_OPS:
            q = params
            descr = "conv %dx%d d%-3d Cin%-4d Cout%-4d F%-3d T%-4d act%d epi%d" % (q.KH, q.KW, q.dilF, q.Cin, q.Cout, q.F, q.T, q.act, q.epi)
        else:
            descr = name
        wr = [w for w in writes if w is not None]
        rd = [t for t in tensors if t is not None and not any(t is w for w in wr)]
        op = Op(fn, C.addressof(params), name, flops, nbytes, descr, self.lane, [_range(t) for t in rd], [_range(t) for t in wr], params)
        self.ops.append(op)
        self.keep.append(params)
        self.keep.extend(t for t in tensors if t is not None)
        self.keep.extend(wr)
        self.flops += flops
        self._sched = None
        return op

    @property
    def descr(self):
        return [o.descr for o in self.ops]

    # ---- dependencies --------------------------------------------------------------------------------------------------------
    def dependencies(self) -> List[List[int]]:
        """deps[j] = ops that must complete before op j starts (RAW, WAR, WAW on overlapping views), nearest ones per view."""
        regions: Dict[int, list] = {}      # storage -> [[range, last_writer, [readers since]]]

        def hit(r, q):
            return r[1] < q[2] and q[1] < r[2] and not _disjoint(r, q)
        deps: List[List[int]] = []
        for j, op in enumerate(self.ops):
            d = set()
            for q in op.reads:
                for r in regions.get(q[0], ()):
                    if r[1] >= 0 and hit(r[0], q):
                        d.add(r[1])
            for q in op.writes:
                for r in regions.get(q[0], ()):
                    if hit(r[0], q):
                        if r[1] >= 0:
                            d.add(r[1])
                        d.update(r[2])
            d.discard(j)
            deps.append(sorted(d))
            for q in op.reads:
                lst = regions.setdefault(q[0], [])
                for r in lst:
                    if r[0][1] == q[1] and r[0][2] == q[2] and r[0][3] == q[3]:
                        r[2].append(j)
                        break
                else:
                    lst.append([q, -1, [j]])
            for q in op.writes:
                lst = regions.setdefault(q[0], [])
                for r in lst:
                    if r[0][1] == q[1] and r[0][2] == q[2] and r[0][3] == q[3]:
                        r[1], r[2] = j, []          # (earlier readers of this exact view are ordered before j, and later writers after j)
                        break
                else:
                    lst.append([q, j, []])
        return deps

    def schedule(self):
        """(waits[j] = ops whose events op j waits for, record = set of ops that record an event after themselves)"""
        if self._sched is None:
            deps = self.dependencies()
            waits: List[List[int]] = [[] for _ in self.ops]
            record = set()
            synced: Dict[Tuple[int, int], int] = {}
            for j, op in enumerate(self.ops):
                for i in deps[j]:
                    li = self.ops[i].lane
                    if li != op.lane and synced.get((li, op.lane), -1) < i:
                        waits[j].append(i)
                        record.add(i)
                        synced[(li, op.lane)] = i
            self._sched = (waits, record, deps)
        return self._sched

    def check(self) -> int:
        """Race detector: every dependency is covered by lane order or by an event edge (possibly through earlier waits).  Returns the
        number of cross-lane edges.  Raises AssertionError on an uncovered dependency."""
        waits, record, deps = self.schedule()
        nl = 1 + max((o.lane for o in self.ops), default=0)
        seen = [[-1] * nl for _ in self.ops]          # seen[j][l] = latest op of lane l known complete when op j starts
        last_on_lane = [-1] * nl
        for j, op in enumerate(self.ops):
            cur = list(seen[last_on_lane[op.lane]]) if last_on_lane[op.lane] >= 0 else [-1] * nl
            if last_on_lane[op.lane] >= 0:
                cur[op.lane] = last_on_lane[op.lane]
            for i in waits[j]:
                li = self.ops[i].lane
                cur[li] = max(cur[li], i)
                for l in range(nl):
                    cur[l] = max(cur[l], seen[i][l])
            seen[j] = cur
            for i in deps[j]:
                assert cur[self.ops[i].lane] >= i, f"op {j} ({op.descr}, lane {op.lane}) may run before op {i} ({self.ops[i].descr}, lane {self.ops[i].lane})"
            last_on_lane[op.lane] = j
        return sum(len(w) for w in waits)

    # ---- execution -----------------------------------------------------------------------------------------------------------
    def zero_scratch(self):
        """Re-establish the contract of the zero_on_fail scratch: arrival counters entirely, the flag region that leads the fp32 split-K scratch."""
        for t in self.zero_on_fail:
            try:
                t.zero_() if t.dtype != torch.float32 else t[:_lib.AID_CONV2D_SPLIT_FLAG_BYTES // 4].zero_()
            except Exception:
                pass
        self._dirty = False

    def _fail(self, op, rc):
        """A launcher returned an error (host-side validation / launch failure -- nothing of THIS op ran, but earlier ops of the run are in flight)."""
        msg = _lib.lib().aid_last_error().decode()
        try:                       # let the kernels already queued finish: they leave their counters / flags zero themselves
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        self.zero_scratch()
        raise _lib.AidError(f"{op.name} failed rc={rc}: {msg}")

    def run(self):
        """Launch every op.  If ANYTHING raises while a run is under way (a launcher's error code, an asynchronous device error surfacing in a torch call
        of the multi-lane path, an interrupt between launches), the plan is marked dirty and the NEXT run first synchronises and re-zeroes the
        counters / flags (zero_scratch): a kernel that was cut short may have left them half-way.  A device FAULT proper (memory violation)
        aborts the process on this stack -- there is no next run to protect."""
        if getattr(self, "_dirty", False):
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.zero_scratch()
        self._dirty = True
        self._run()
        self._dirty = False

    def _run(self):
        cur = torch.cuda.current_stream()
        multi = self.lanes > 1 and self.timing is None and self.trace is None and any(o.lane for o in self.ops)
        if not multi:
            stream = cur.cuda_stream
            timing, trace = self.timing, self.trace
            for op in self.ops:
                if trace is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = op.fn(op.addr, stream)
                    e1.record()
                    trace.append((e0, e1, op.name, op.descr, op.addr, _lib.lib().aid_last_kernel().decode() if op.name in CONV_OPS else "", op.lane))
                elif timing is not None and op.name in CONV_OPS and (op.flops > 0 or op.name == "aid_conv2d_wino2d_output"):
                    # (the output-transform + epilogue pass of a 2-D Winograd layer carries no FLOPs of its own but IS conv time: the fused 1-D kernels do
                    #  that work inside the timed kernel -- ADVICE r5)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = op.fn(op.addr, stream)
                    e1.record()
                    timing.append((e0, e1, op.flops, op.descr, op.nbytes, _lib.lib().aid_last_kernel().decode()))
                else:
                    rc = op.fn(op.addr, stream)
                if rc != 0:
                    self._fail(op, rc)
            return
        waits, record, _ = self.schedule()
        if self.side_stream is None or self.side_stream.device != cur.device:
            self.side_stream = torch.cuda.Stream(device=cur.device)
        side = self.side_stream
        side.wait_stream(cur)
        streams = (cur, side)
        handles = (cur.cuda_stream, side.cuda_stream)
        ev = self.__dict__.setdefault("_events", {})      # one event per recording op, created once and re-recorded by every run (a wait enqueued by an earlier
        for j, op in enumerate(self.ops):                  # run captured the event's state at that time: re-recording does not disturb it)
            s = streams[op.lane]
            for i in waits[j]:
                s.wait_event(ev[i])
            rc = op.fn(op.addr, handles[op.lane])
            if rc != 0:
                self._fail(op, rc)
            if j in record:
                e = ev.get(j)
                if e is None:
                    e = ev[j] = torch.cuda.Event()
                e.record(s)
        cur.wait_stream(side)
