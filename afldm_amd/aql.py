"""AQL-level control of the step's launches (csrc/aqlq.cpp, libafldm_aql.so).

The library is a ROCr tool: it must be named in HSA_TOOLS_LIB BEFORE the process touches the GPU (`install()` - call
it before the first torch.cuda use).  Afterwards every dispatch packet of the process passes its handler, which can
record it and re-head it by a policy armed from here:

    aql.install()                       # sets HSA_TOOLS_LIB (appends to an existing value)
    ... create model / engine, capture the step graph ...
    aql.arm(policy, total)              # policy: bytes, one per dispatch of a step, applied cyclically to `total` dispatches
    POLICY bits: 1 = clear the barrier bit, 2 = no acquire fence, 4 = no release fence
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "lib", "libafldm_aql.so")
NO_BARRIER, NO_ACQUIRE, NO_RELEASE = 1, 2, 4

REC = np.dtype([("header", "<u2"), ("setup", "<u2"), ("wg", "<u2", 3), ("pad", "<u2"), ("grid", "<u4", 3),
                ("priv_bytes", "<u4"), ("group_bytes", "<u4"), ("kernel_object", "<u8"), ("kernarg", "<u8"),
                ("completion", "<u8")])
assert REC.itemsize == 56

_lib = None


def install():
    """Name the tool library in HSA_TOOLS_LIB.  Must run before ROCr initialises (first GPU use of the process)."""
    if not os.path.exists(LIB):
        raise RuntimeError(f"{LIB} is missing: run `python -m afldm_amd.build`")
    cur = os.environ.get("HSA_TOOLS_LIB", "")
    if LIB not in cur.split(":"):
        os.environ["HSA_TOOLS_LIB"] = (cur + ":" if cur else "") + LIB
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB)
        L.afldm_aql_loaded.restype = ctypes.c_int
        L.afldm_aql_counts.argtypes = [ctypes.c_void_p]
        L.afldm_aql_record.argtypes = [ctypes.c_int]
        L.afldm_aql_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.afldm_aql_records.restype = ctypes.c_int
        L.afldm_aql_arm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
        L.afldm_aql_arm.restype = ctypes.c_int
        L.afldm_aql_armed_left.restype = ctypes.c_longlong
        _lib = L
    return _lib


def loaded():
    """True when ROCr called the tool's OnLoad in this process (the queues are intercept queues)."""
    return bool(lib().afldm_aql_loaded())


def counts():
    out = (ctypes.c_uint64 * 4)()
    lib().afldm_aql_counts(out)
    return {"dispatch": out[0], "other": out[1], "queues": out[2], "rewritten": out[3]}


def record(on=True):
    lib().afldm_aql_record(1 if on else 0)


def records(max_n=1 << 15):
    buf = np.zeros(max_n, dtype=REC)
    n = lib().afldm_aql_records(buf.ctypes.data, max_n)
    return buf[:n].copy()


def arm(policy, total):
    p = np.ascontiguousarray(np.asarray(policy, dtype=np.uint8))
    rc = lib().afldm_aql_arm(p.ctypes.data if p.size else None, int(p.size), int(total))
    if rc != 0:
        raise ValueError("policy too long")


def disarm():
    lib().afldm_aql_arm(None, 0, 0)


def armed_left():
    return int(lib().afldm_aql_armed_left())


# ----------------------------------------------------------------------------- launch-order tracing
# Which dispatches of a step may start without waiting for the launches in front of them is known to the HOST code
# that issues them (the model's blocks).  During ONE eager pass of the step (`trace_begin()` .. `trace_end()`), every
# `with aql.independent(tag):` region notes the dispatch indices it covers; the policy of a graph replay of the same
# step is built from those (the graph replays the same launches in the same order).
_trace = None


def trace_begin():
    global _trace
    _trace = {"base": counts()["dispatch"], "marks": []}


def trace_end():
    """-> (dispatches of the pass, [(tag, first, end)] relative to its first dispatch)"""
    global _trace
    t, _trace = _trace, None
    return counts()["dispatch"] - t["base"], t["marks"]


class independent:
    """The FIRST launch issued inside the region does not depend on the launch(es) issued immediately before the
    region (it may run beside them); what follows the region depends on both.  A no-op unless a trace is running."""

    def __init__(self, tag=""):
        self.tag = tag

    def __enter__(self):
        if _trace is not None:
            self.i0 = counts()["dispatch"] - _trace["base"]
        return self

    def __exit__(self, *exc):
        if _trace is not None:
            _trace["marks"].append((self.tag, self.i0, counts()["dispatch"] - _trace["base"]))
        return False


def policy_from_marks(n, marks, bits=NO_BARRIER):
    pol = np.zeros(n, dtype=np.uint8)
    for _, i0, i1 in marks:
        if i1 > i0:
            pol[i0] |= bits
    return pol
