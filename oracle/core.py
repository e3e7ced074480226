"""ORACLE (test infrastructure) -- ctypes access to oracle/_build/liboracle_core.so, the plain-C
restatement of the onlyunitdiag hot path (oracle/oracle_core.c).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle_core.so")


class Stats(C.Structure):
    _fields_ = [("cost", C.c_double), ("gradnorm", C.c_double), ("Delta", C.c_double), ("iters", C.c_int),
                ("hessvecs", C.c_int), ("accepted", C.c_int), ("rejected", C.c_int), ("cost_evals", C.c_int),
                ("last_stop_inner", C.c_int)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _lib = C.CDLL(_PATH)
        _lib.oc_num_threads.restype = C.c_int
        # default for the parity tests: at most 16 threads, never more than the CPUs this process may run on
        # (bench.py's cpu_baseline calls set_threads(host_cpus()) -- all host cores)
        _lib.oc_set_threads(max(1, min(16, host_cpus())))
    return _lib


def host_cpus():
    """CPUs this process may run on (nproc)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def set_threads(n):
    """OpenMP threads of the C restatement from now on; returns the count in force."""
    load().oc_set_threads(max(1, int(n)))
    return num_threads()


def _csr(Cm):
    Cs = Cm.tocsr()
    Cs.sort_indices()
    return (np.ascontiguousarray(Cs.indptr, dtype=np.int64), np.ascontiguousarray(Cs.indices, dtype=np.int32),
            np.ascontiguousarray(Cs.data, dtype=np.float64))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def num_threads():
    return load().oc_num_threads()


def cost_state(Cm, Y):
    lib = load()
    rp, ci, cv = _csr(Cm)
    n, p = Y.shape
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    f = C.c_double()
    eG = np.empty(n)
    G = np.empty((n, p))
    lib.oc_cost_state(n, p, _p(rp, C.c_int64), _p(ci, C.c_int32), _p(cv, C.c_double), _p(Y, C.c_double), C.byref(f),
                      _p(eG, C.c_double), _p(G, C.c_double))
    return f.value, eG, G


def hessvec(Cm, Y, U, eG):
    lib = load()
    rp, ci, cv = _csr(Cm)
    n, p = Y.shape
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    U = np.ascontiguousarray(U, dtype=np.float64)
    H = np.empty((n, p))
    lib.oc_hessvec(n, p, _p(rp, C.c_int64), _p(ci, C.c_int32), _p(cv, C.c_double), _p(Y, C.c_double), _p(U, C.c_double),
                   _p(np.ascontiguousarray(eG), C.c_double), _p(H, C.c_double))
    return H


def rtr_onlyunitdiag(Cm, Y, maxiter, maxinner, tolgradnorm):
    """trustregions() of the onlyunitdiag problem in C; returns (Y, Stats)."""
    lib = load()
    rp, ci, cv = _csr(Cm)
    n, p = Y.shape
    Yc = np.array(Y, dtype=np.float64, order="C", copy=True)
    st = Stats()
    rc = lib.oc_rtr_onlyunitdiag(n, p, _p(rp, C.c_int64), _p(ci, C.c_int32), _p(cv, C.c_double), _p(Yc, C.c_double),
                                 int(maxiter), int(maxinner), C.c_double(tolgradnorm), C.byref(st))
    if rc != 0:
        raise MemoryError("oc_rtr_onlyunitdiag failed")
    return Yc, st
