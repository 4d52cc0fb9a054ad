"""ctypes loader for oracle/libmsi_cpubase.so — the cpu_baseline leg of bench.py
and a second opinion in tests.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmsi_cpubase.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(path)
        vp = C.c_void_p
        L.cpb_row_norms.restype = None
        L.cpb_row_norms.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
        L.cpb_vs_topk_mt.restype = None
        L.cpb_vs_topk_mt.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, C.c_uint32,
                                     C.c_uint32, vp, vp, vp]
        L.cpb_dict_build.restype = vp
        L.cpb_dict_build.argtypes = [vp, vp, C.c_uint32]
        L.cpb_dict_free.restype = None
        L.cpb_dict_free.argtypes = [vp]
        L.cpb_dict_lookup_mt.restype = None
        L.cpb_dict_lookup_mt.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         vp, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def host_threads():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container with
    cpu.max = "1600000 100000" sees 256 logical CPUs but gets 16 CPUs' worth of time — threads beyond that only
    throttle each other, and reporting 256 "cores" would misstate the baseline's hardware)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


class CpuVectorScan:
    def __init__(self, rows, docids):
        self.rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.docids = np.ascontiguousarray(docids, dtype=np.uint32)
        self.n, self.d = self.rows.shape
        self.norms = np.zeros(self.n, dtype=np.float32)
        lib().cpb_row_norms(_p(self.rows), self.n, self.d, _p(self.norms))

    def search(self, queries, k, threads=None):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        out_d = np.zeros((nq, k), dtype=np.uint32)
        out_s = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        lib().cpb_vs_topk_mt(_p(self.rows), _p(self.norms), _p(self.docids), self.n, self.d, _p(q), nq,
                             k, threads or host_threads(), _p(out_d), _p(out_s), _p(cnt))
        return out_d, out_s, cnt


def pack_queries(queries):
    """[(word, max_typos, is_prefix)] -> (bytes u8, offsets u32, flags u8)."""
    bs = [w.encode("utf-8") if isinstance(w, str) else bytes(w) for w, _, _ in queries]
    qb = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if qb.size == 0:
        qb = np.zeros(1, np.uint8)
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    fl = np.array([(min(mt, 2) & 3) | (4 if pf else 0) for _, mt, pf in queries], dtype=np.uint8)
    return qb, off, fl


class CpuDictionary:
    def __init__(self, concat, offsets):
        self.concat = np.ascontiguousarray(concat, dtype=np.uint8)
        if self.concat.size == 0:
            self.concat = np.zeros(1, np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self.n = self.offsets.size - 1
        self._h = lib().cpb_dict_build(_p(self.concat), _p(self.offsets), self.n)

    def lookup_packed(self, qb, off, fl, cap_one=150, cap_two=50, threads=None):
        nq = fl.size
        one = np.zeros((nq, cap_one), dtype=np.uint32)
        two = np.zeros((nq, cap_two), dtype=np.uint32)
        c1 = np.zeros(nq, dtype=np.uint32)
        c2 = np.zeros(nq, dtype=np.uint32)
        lib().cpb_dict_lookup_mt(self._h, _p(qb), _p(off), _p(fl), nq, cap_one, cap_two,
                                 threads or host_threads(), _p(one), _p(c1), _p(two), _p(c2))
        return one, c1, two, c2

    def lookup(self, queries, cap_one=150, cap_two=50, threads=None):
        qb, off, fl = pack_queries(queries)
        one, c1, two, c2 = self.lookup_packed(qb, off, fl, cap_one, cap_two, threads)
        return [(one[i, :c1[i]].copy(), two[i, :c2[i]].copy()) for i in range(len(queries))]

    def __del__(self):
        try:
            if self._h:
                lib().cpb_dict_free(self._h)
                self._h = None
        except Exception:
            pass
