"""Constants and dtype maps of the ReCoDe format (host side, no compute).
Mirrors the interface of reference pyrecode/misc.py (rc_cfg :4-38, map_dtype :41-71, get_dtype_code :74-82,
get_dtype_string :85-95); written table-first."""
import numpy as np


class rc_cfg:
    REQ_TYPE_QUERY, REQ_TYPE_COMMAND = 0, 1
    FILE_TYPE_BINARY, FILE_TYPE_MRC, FILE_TYPE_SEQ, FILE_TYPE_OTHER = 0, 1, 2, 255
    STATUS_CODE_BUSY, STATUS_CODE_AVAILABLE = 0, 1
    STATUS_CODE_ERROR, STATUS_CODE_NOT_READY, STATUS_CODE_IS_CLOSED = -1, -2, -3
    MESSAGE_TYPE_INFO, MESSAGE_TYPE_ERROR, MESSAGE_TYPE_STATUS, MESSAGE_TYPE_ACK = 0, -1, 1, 2


rc_cfg.STATUS_CODES = {k: v for k, v in vars(rc_cfg).items() if k.startswith("STATUS_CODE_")}
rc_cfg.MESSAGE_TYPES = {k: v for k, v in vars(rc_cfg).items() if k.startswith("MESSAGE_TYPE_")}

# (type code) -> ((max bit depth, numpy type), ...); type 0 unsigned, 1 signed, 2 float
_DTYPE_LADDER = {
    0: ((8, np.uint8), (16, np.uint16), (32, np.uint32), (64, np.uint64)),
    1: ((8, np.int8), (16, np.int16), (32, np.int32), (64, np.int64)),
    2: ((32, np.float32), (64, np.float64)),
}
_DTYPE_NAMES = ("uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64", "float32", "float64")


def map_dtype(type, bit_depth):
    """Smallest numpy type of the given kind that holds bit_depth bits."""
    for limit, np_type in _DTYPE_LADDER.get(type, ()):
        if bit_depth <= limit:
            return np_type
    raise ValueError("Unable to match a numpy dtype for type = %s (0=unsigned int, 1=signed int, 2=float) "
                     "with bit depth = %s" % (type, bit_depth))


def get_dtype_code(dtype):
    name = np.dtype(dtype).name
    if name not in _DTYPE_NAMES:
        raise ValueError("Unknown dtype")
    return _DTYPE_NAMES.index(name)


def get_dtype_string(dtype):
    code = int(dtype)
    if not 0 <= code < len(_DTYPE_NAMES):
        raise ValueError("Unknown dtype")
    return _DTYPE_NAMES[code]


def effective_cpus():
    """(CPUs this process may really use, CPUs it may be scheduled on): the scheduler affinity, capped by the cgroup's CPU bandwidth
    quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A container with 256 visible cores and a quota of 16 runs 256
    busy threads at 16 cores' worth - and is throttled for the rest of every accounting period once the quota is spent, which stalls
    pipelines that count on their helper threads: pools are sized by the first number."""
    import math
    import os
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    usable = visible if quota is None else max(1, min(visible, int(math.ceil(quota))))
    return usable, visible
