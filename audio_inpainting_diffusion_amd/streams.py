"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask) -- the spatial-partition experiment of round 6.

VERDICT r5 next-1(a): the sub-batch streams of an evaluation are free-running; an MFMA-bound kernel fills every CU's LDS / registers first and the
HBM-bound passes of the other sub-batch mostly WAIT (one kernel in flight 56 % of the time, profiles/r05_overlap_timeline.txt).  A CU mask is the one
schedule free-running streams cannot produce: MFMA-bound kernels on ~192 CUs while the passes of the other sub-batch own the remaining ~64.

Mask layout (gfx942 / gfx950, KFD mqd_symmetrically_map_cu_mask): bit i of the mask selects CU (i // n_xcc) of XCC (i % n_xcc) -- the bits go ROUND-ROBIN over
the 8 XCDs -- so "the last k CUs of every XCD" is the set {i : i // 8 >= 32 - k}.  Both halves of a partition keep CUs on every XCD: the workgroup -> XCD
round-robin the kernels' XCD-aware tile orders rely on is unchanged.  tools/cu_mask_probe.py prints which (XCC, SE, CU) a masked stream really ran on.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

N_XCC, CUS_PER_XCC = 8, 32          # MI355X: 256 CUs in 8 XCDs


def _hip():
    return C.CDLL("libamdhip64.so")                  # the runtime torch already loaded (same soname -> same handle)


def partition_masks(n_pass: int, interleaved: bool = True) -> Tuple[List[int], List[int]]:
    """(mask words of the MFMA partition, mask words of the pass partition): the pass partition owns the last n_pass / 8 CUs of every XCD.
    interleaved=False assumes the other bit order (bit = xcc * 32 + cu) -- only for the probe, which tells the two apart."""
    if n_pass % N_XCC or not 0 < n_pass < N_XCC * CUS_PER_XCC:
        raise ValueError("n_pass must be a multiple of 8 in (0, 256)")
    k = n_pass // N_XCC
    bits_pass = 0
    for xcc in range(N_XCC):
        for cu in range(CUS_PER_XCC - k, CUS_PER_XCC):
            bits_pass |= 1 << ((cu * N_XCC + xcc) if interleaved else (xcc * CUS_PER_XCC + cu))
    full = (1 << (N_XCC * CUS_PER_XCC)) - 1
    words = lambda b: [(b >> (32 * w)) & 0xFFFFFFFF for w in range(N_XCC * CUS_PER_XCC // 32)]
    return words(full & ~bits_pass), words(bits_pass)


def xcd_range_mask(lo: int, hi: int) -> List[int]:
    """mask words selecting CUs [lo, hi) of EVERY XCD (round-robin bit order: bit = cu * 8 + xcc)"""
    if not 0 <= lo < hi <= CUS_PER_XCC:
        raise ValueError("CU range must lie inside [0, 32]")
    bits = 0
    for cu in range(lo, hi):
        bits |= 0xFF << (cu * N_XCC)
    return [(bits >> (32 * w)) & 0xFFFFFFFF for w in range(N_XCC * CUS_PER_XCC // 32)]


def cu_masked_stream(words: List[int]) -> torch.cuda.ExternalStream:
    """A new HIP stream whose kernels may only run on the CUs set in ``words`` (uint32 little-endian bit mask), as a torch stream."""
    h = _hip()
    fn = h.hipExtStreamCreateWithCUMask
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = fn(C.byref(s), len(words), arr)
    if rc != 0 or not s.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (hipError {rc})")
    return torch.cuda.ExternalStream(s.value)
