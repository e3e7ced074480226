"""ctypes binding of ``include/manisdp_hip.h`` (libmanisdp_hip.so).

There is no CPU fallback: if the shared library is missing or no HIP device is
visible, every entry point raises.  The library is looked up in-tree
(``manisdp-matlab_amd/lib/libmanisdp_hip.so``) so the GPU box loads exactly the
file that ``__graft_entry__.build()`` produced.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmanisdp_hip.so")

KIND_ONLYUNITDIAG, KIND_UNITDIAG, KIND_UNITTRACE, KIND_GENERIC, KIND_MULTIBLOCK, KIND_DUAL_UNITDIAG = 1, 2, 3, 4, 5, 6


class RtrOpts(C.Structure):
    _fields_ = [("maxiter", C.c_int32), ("maxinner", C.c_int32), ("mininner", C.c_int32),
                ("reserved0", C.c_int32), ("tolgradnorm", C.c_double), ("kappa", C.c_double),
                ("theta", C.c_double), ("rho_prime", C.c_double), ("rho_regularization", C.c_double),
                ("Delta_bar", C.c_double), ("Delta0", C.c_double)]


class RtrStats(C.Structure):
    _fields_ = [("cost", C.c_double), ("gradnorm", C.c_double), ("Delta", C.c_double),
                ("seconds", C.c_double), ("iters", C.c_int32), ("hessvecs", C.c_int32),
                ("accepted", C.c_int32), ("rejected", C.c_int32), ("cost_evals", C.c_int32),
                ("last_stop_inner", C.c_int32), ("reserved", C.c_int32 * 2)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class MsdpError(RuntimeError):
    pass


_P = C.POINTER
_dp = _P(C.c_double)
_i64p = _P(C.c_int64)

# name -> (restype, argtypes); this table is also what tests/test_cabi_symbols.py
# checks against the declarations in include/manisdp_hip.h.
SIGNATURES = {
    "msdp_rtr_default_opts": (None, [_P(RtrOpts)]),
    "msdp_set_device": (C.c_int, [C.c_int32]),
    "msdp_device_count": (C.c_int, [_P(C.c_int32)]),
    "msdp_create_onlyunitdiag_csc": (C.c_int, [C.c_int64, _i64p, _i64p, _dp, C.c_int32, _P(C.c_void_p)]),
    "msdp_create_onlyunitdiag_dense": (C.c_int, [C.c_int64, _dp, C.c_int32, _P(C.c_void_p)]),
    "msdp_create_onlyunitdiag_dense_synthetic": (C.c_int, [C.c_int64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                                          _P(C.c_void_p)]),
    "msdp_synthetic_dense_entry": (C.c_double, [C.c_int64, C.c_int64, C.c_int64, C.c_uint64]),
    "msdp_debug_set_full_rows": (C.c_int, [C.c_void_p, _dp]),
    "msdp_debug_p2p_self": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp]),
    "msdp_create_affine": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp,
                                     C.c_int32, _P(C.c_void_p)]),
    "msdp_create_multiblock": (C.c_int, [C.c_int32, _i64p, C.c_int32, C.c_int64, _i64p, _i64p, _dp, _dp, _dp,
                                         C.c_int32, _P(C.c_void_p)]),
    "msdp_destroy": (C.c_int, [C.c_void_p]),
    "msdp_set_multipliers": (C.c_int, [C.c_void_p, _dp, C.c_double]),
    "msdp_set_point": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "msdp_get_point": (C.c_int, [C.c_void_p, _dp]),
    "msdp_get_p": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "msdp_get_kind": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "msdp_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "msdp_rtr": (C.c_int, [C.c_void_p, _P(RtrOpts), _P(RtrStats)]),
    "msdp_rtr_host": (C.c_int, [C.c_void_p, C.c_int32, _dp, _P(RtrOpts), _P(RtrStats)]),
    "msdp_cost": (C.c_int, [C.c_void_p, _dp]),
    "msdp_rgrad": (C.c_int, [C.c_void_p, _dp]),
    "msdp_hessvec": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_proj": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_retr": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_get_z": (C.c_int, [C.c_void_p, _dp]),
    "msdp_linesearch_cost": (C.c_int, [C.c_void_p, _dp, C.c_double, _dp]),
    "msdp_linesearch_accept": (C.c_int, [C.c_void_p]),
    "msdp_escape_eigs": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp, _P(C.c_int32)]),
    "msdp_escape_eigs_matrix": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp,
                                         _P(C.c_int32)]),
    "msdp_escape_info": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _dp]),
    "msdp_escape_lower_bound": (C.c_int, [C.c_void_p, _dp]),
    "msdp_escape_method": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "msdp_debug_collective_calls": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "msdp_debug_last_rtr_device_ms": (C.c_int, [C.c_void_p, _dp]),
    "msdp_debug_persist_trace": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_uint64), C.c_int64, _P(C.c_int32), _dp]),
    "msdp_debug_time_collective": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _P(C.c_double)]),
    "msdp_debug_get_tcg_step": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_debug_sym_eig": (C.c_int, [C.c_int32, _dp, _dp, _dp]),
    "msdp_debug_ritz": (C.c_int, [C.c_int32, _dp, _dp, _dp, _dp, _P(C.c_int32)]),
    "msdp_get_dual_slack": (C.c_int, [C.c_void_p, _dp]),
    "msdp_get_dual_slack_block": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "msdp_block_eigs": (C.c_int, [C.c_void_p, C.c_int32, _i64p, _i64p, C.c_int32, C.c_int32, _dp, _dp]),
    "msdp_release_cache": (C.c_int, []),
    "msdp_debug_pool_stats": (C.c_int, [_P(C.c_int64), _P(C.c_int64), _P(C.c_int64)]),
    "msdp_debug_mem_info": (C.c_int, [_P(C.c_int64), _P(C.c_int64)]),
    "msdp_comm_init_local": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "msdp_comm_init_ipc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    "msdp_get_point_all": (C.c_int, [C.c_void_p, _dp]),
    "msdp_get_z_all": (C.c_int, [C.c_void_p, _dp]),
    "msdp_factor_gram": (C.c_int, [C.c_void_p, _dp]),
    "msdp_factor_rotate": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "msdp_factor_append": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.c_double, C.c_int32]),
    "msdp_create_dual_unitdiag": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, _dp, C.c_int32, _i64p, _i64p, _dp,
                                            _dp, C.c_int32, C.POINTER(C.c_void_p)]),
    "msdp_dual_set_penalty": (C.c_int, [C.c_void_p, C.c_double, _dp]),
    "msdp_dual_outer_step": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "msdp_dual_get_y": (C.c_int, [C.c_void_p, _dp]),
    "msdp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "msdp_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "msdp_local_rows": (C.c_int, [C.c_void_p, _i64p, _i64p]),
    "msdp_debug_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "msdp_tcg_path": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "msdp_debug_persist_form": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "msdp_point_snapshot": (C.c_int, [C.c_void_p]),
    "msdp_point_restore": (C.c_int, [C.c_void_p]),
    "msdp_al_primal": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_al_dual": (C.c_int, [C.c_void_p, _dp, _dp]),
    "msdp_escape_eigs_dual": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp, _P(C.c_int32)]),
    "msdp_bench_hessvec": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "msdp_bench_tcg_trip": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "msdp_bench_kernel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "msdp_last_error": (C.c_char_p, []),
    "msdp_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libmanisdp_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsdpError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C manisdp-matlab_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # the library parks one escape workspace per process between handles (msdp_release_cache): give it back when the
    # interpreter exits, and let callers do so earlier (release_cache()) before large allocations of their own
    import atexit
    atexit.register(lambda: lib.msdp_release_cache())
    return lib


def release_cache():
    """Free the device memory the library keeps between handles (the parked Lanczos workspace of the escape; the arenas of
    uncached memory, when no handle of the process holds a block of them)."""
    _check(load().msdp_release_cache())


def pool_stats():
    """(bytes the uncached-memory arenas hold, bytes handed out to live handles, number of arenas) -- msdp_debug_pool_stats."""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    _check(load().msdp_debug_pool_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def mem_info():
    """(free, total) bytes of the current device -- msdp_debug_mem_info."""
    a, b = C.c_int64(), C.c_int64()
    _check(load().msdp_debug_mem_info(C.byref(a), C.byref(b)))
    return a.value, b.value


def _check(rc):
    if rc != 0:
        raise MsdpError(f"libmanisdp_hip error {rc}: {load().msdp_last_error().decode()}")


def _dptr(a):
    return a.ctypes.data_as(_dp)


def set_device(dev):
    _check(load().msdp_set_device(int(dev)))


def device_count():
    n = C.c_int32()
    _check(load().msdp_device_count(C.byref(n)))
    return n.value


def default_opts(**kw):
    o = RtrOpts()
    load().msdp_rtr_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Handle:
    """Owning wrapper of an ``msdp_handle``.  Factors cross this boundary in the
    reference layout: (n, p) C-contiguous NumPy arrays for the oblique kinds (the
    bytes of MATLAB's p x n column-major Y), (n, p) arrays for unittrace as well
    (converted to MATLAB's n x p column-major, i.e. Fortran order, on the way in)."""

    def __init__(self, ptr, kind, n):
        self._h = C.c_void_p(ptr)
        self.kind = kind
        self.n = n
        self.p = 0
        self._lib = load()

    # ---- construction
    @classmethod
    def onlyunitdiag(cls, Cmat, pcap=32):
        import scipy.sparse as sp
        lib = load()
        out = C.c_void_p()
        n = Cmat.shape[0]
        if sp.issparse(Cmat):
            Cc = Cmat.tocsc()
            Cc.sort_indices()
            jc = np.ascontiguousarray(Cc.indptr, dtype=np.int64)
            ir = np.ascontiguousarray(Cc.indices, dtype=np.int64)
            pr = np.ascontiguousarray(Cc.data, dtype=np.float64)
            _check(lib.msdp_create_onlyunitdiag_csc(n, jc.ctypes.data_as(_i64p), ir.ctypes.data_as(_i64p),
                                                    _dptr(pr), pcap, C.byref(out)))
        else:
            Cd = np.ascontiguousarray(Cmat, dtype=np.float64)
            _check(lib.msdp_create_onlyunitdiag_dense(n, _dptr(Cd), pcap, C.byref(out)))
        return cls(out.value, KIND_ONLYUNITDIAG, n)

    @classmethod
    def dense_synthetic(cls, n, seed, nranks=1, rank=0, pcap=32):
        """Pre-sharded synthetic dense-C problem (rank `rank` of `nranks` holds its rows only)."""
        lib = load()
        out = C.c_void_p()
        _check(lib.msdp_create_onlyunitdiag_dense_synthetic(n, seed, nranks, rank, pcap, C.byref(out)))
        return cls(out.value, KIND_ONLYUNITDIAG, n)

    def debug_shard(self, nranks, rank):
        """Test-only: rank `rank` of `nranks` without a communicator (sparse C)."""
        _check(self._lib.msdp_debug_shard(self._h, nranks, rank))

    def debug_p2p_self(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        _check(self._lib.msdp_debug_p2p_self(self._h, x.size, _dptr(x), _dptr(out)))
        return out

    def debug_set_full_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        _check(self._lib.msdp_debug_set_full_rows(self._h, _dptr(rows)))

    @classmethod
    def affine(cls, kind, At, b, c, n, pcap=32):
        lib = load()
        out = C.c_void_p()
        Atc = At.tocsc()
        Atc.sort_indices()
        jc = np.ascontiguousarray(Atc.indptr, dtype=np.int64)
        ir = np.ascontiguousarray(Atc.indices, dtype=np.int64)
        pr = np.ascontiguousarray(Atc.data, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        _check(lib.msdp_create_affine(kind, n, Atc.shape[1], jc.ctypes.data_as(_i64p), ir.ctypes.data_as(_i64p),
                                      _dptr(pr), _dptr(b), _dptr(c), pcap, C.byref(out)))
        return cls(out.value, kind, n)

    @classmethod
    def multiblock(cls, At, b, c, block_n, nob, pcap=32):
        """Block-diagonal X = diag(X_1..X_nb), unit diagonal on the first `nob` blocks (ManiSDP_multiblock.m).  At is
        (sum n_i^2) x m over the concatenated vecs of the blocks.  The factor is one (N, p) array, N = sum n_i."""
        lib = load()
        out = C.c_void_p()
        Atc = At.tocsc()
        Atc.sort_indices()
        jc = np.ascontiguousarray(Atc.indptr, dtype=np.int64)
        ir = np.ascontiguousarray(Atc.indices, dtype=np.int64)
        pr = np.ascontiguousarray(Atc.data, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        bn = np.ascontiguousarray(block_n, dtype=np.int64)
        _check(lib.msdp_create_multiblock(len(bn), bn.ctypes.data_as(_i64p), int(nob), Atc.shape[1], jc.ctypes.data_as(_i64p),
                                          ir.ctypes.data_as(_i64p), _dptr(pr), _dptr(b), _dptr(c), pcap, C.byref(out)))
        return cls(out.value, KIND_MULTIBLOCK, int(bn.sum()))

    @classmethod
    def dual_unitdiag(cls, A, b, c, dAAt, B=None, cf=None, pcap=32):
        """Dual approach, diag(S) = 1 (ManiDSDP_unitdiag.m).  ``A`` is the m x n^2 PSD part (rows = vec(A_k)), ``c`` its
        cost (n^2), ``B`` the m x nf free part with costs ``cf``, ``dAAt = diag(A A')``.  The factor of S is (n, p)."""
        import scipy.sparse as sp
        lib = load()
        out = C.c_void_p()
        Atc = sp.csc_matrix(sp.csr_matrix(A).T)
        Atc.sort_indices()
        n = int(round(np.sqrt(Atc.shape[0])))
        jc = np.ascontiguousarray(Atc.indptr, dtype=np.int64)
        ir = np.ascontiguousarray(Atc.indices, dtype=np.int64)
        pr = np.ascontiguousarray(Atc.data, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        d = np.ascontiguousarray(dAAt, dtype=np.float64)
        nf = 0 if B is None else int(B.shape[1])
        if nf:
            Bc = sp.csc_matrix(B)
            Bc.sort_indices()
            bjc = np.ascontiguousarray(Bc.indptr, dtype=np.int64)
            bir = np.ascontiguousarray(Bc.indices, dtype=np.int64)
            bpr = np.ascontiguousarray(Bc.data, dtype=np.float64)
            cfv = np.ascontiguousarray(cf, dtype=np.float64)
            args = (bjc.ctypes.data_as(_i64p), bir.ctypes.data_as(_i64p), _dptr(bpr), _dptr(cfv))
        else:
            args = (None, None, None, None)
        _check(lib.msdp_create_dual_unitdiag(n, Atc.shape[1], jc.ctypes.data_as(_i64p), ir.ctypes.data_as(_i64p), _dptr(pr), _dptr(d),
                                             _dptr(b), _dptr(c), nf, *args, pcap, C.byref(out)))
        hd = cls(out.value, KIND_DUAL_UNITDIAG, n)
        hd.m = Atc.shape[1]
        hd.nf = nf
        return hd

    def dual_set_penalty(self, sigma, w=None):
        w = np.ascontiguousarray(w if w is not None else np.zeros(max(self.nf, 1)), dtype=np.float64)
        _check(self._lib.msdp_dual_set_penalty(self._h, float(sigma), _dptr(w)))

    def dual_outer_step(self):
        """Outer step of ManiDSDP_unitdiag.m:70-81 on the device: returns (b'y, <C,eX>, |As|^2, Af, z)."""
        scal = np.zeros(3)
        Af = np.zeros(max(self.nf, 1))
        z = np.zeros(self.n)
        _check(self._lib.msdp_dual_outer_step(self._h, _dptr(scal), _dptr(Af), _dptr(z)))
        return float(scal[0]), float(scal[1]), float(scal[2]), Af[:self.nf], z

    def dual_get_y(self):
        y = np.zeros(self.m)
        _check(self._lib.msdp_dual_get_y(self._h, _dptr(y)))
        return y

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.msdp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- layout helpers
    def _to_boundary(self, Y):
        Y = np.asarray(Y, dtype=np.float64)
        if self.kind in (KIND_UNITTRACE, KIND_GENERIC):
            return np.asfortranarray(Y)           # MATLAB n x p column-major
        return np.ascontiguousarray(Y)            # bytes of MATLAB p x n column-major

    def _empty(self):
        # zero-initialised: a row-sharded handle (comm_init / debug_shard) only writes its own rows
        if self.kind in (KIND_UNITTRACE, KIND_GENERIC):
            return np.zeros((self.n, self.p), dtype=np.float64, order="F")
        return np.zeros((self.n, self.p), dtype=np.float64, order="C")

    # ---- point I/O
    def set_point(self, Y):
        Yb = self._to_boundary(Y)
        assert Yb.shape[0] == self.n
        self.p = Yb.shape[1]
        _check(self._lib.msdp_set_point(self._h, self.p, _dptr(Yb)))

    def get_point(self):
        out = self._empty()
        _check(self._lib.msdp_get_point(self._h, _dptr(out)))
        return np.ascontiguousarray(out)

    def get_point_all(self):
        """All n rows of the resident point on every rank of a row-sharded handle (one all-gather + download)."""
        out = self._empty()
        _check(self._lib.msdp_get_point_all(self._h, _dptr(out)))
        return np.ascontiguousarray(out)

    def factor_gram(self):
        """p x p Gram matrix of the columns of the resident factor (computed on the device)."""
        G = np.zeros((self.p, self.p))
        _check(self._lib.msdp_factor_gram(self._h, _dptr(G)))
        return G

    def factor_rotate(self, Q):
        """Resident factor <- factor @ Q (Q: p x r): the rank cut, on the device."""
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        assert Q.shape[0] == self.p
        _check(self._lib.msdp_factor_rotate(self._h, Q.shape[1], _dptr(Q)))
        self.p = Q.shape[1]

    def factor_append(self, V, alpha, normalize=True):
        """Resident factor <- [factor, alpha*V] (V: n x k), rows renormalised (oblique kinds), on the device."""
        Vf = np.asfortranarray(V, dtype=np.float64)
        assert Vf.shape[0] == self.n
        _check(self._lib.msdp_factor_append(self._h, Vf.shape[1], _dptr(Vf), float(alpha), 1 if normalize else 0))
        self.p += Vf.shape[1]

    def set_multipliers(self, y, sigma):
        y = np.ascontiguousarray(y, dtype=np.float64)
        _check(self._lib.msdp_set_multipliers(self._h, _dptr(y), float(sigma)))

    # ---- hot path
    def rtr(self, opts):
        st = RtrStats()
        _check(self._lib.msdp_rtr(self._h, C.byref(opts), C.byref(st)))
        return st

    # ---- fine-grained
    def cost(self):
        f = C.c_double()
        _check(self._lib.msdp_cost(self._h, C.byref(f)))
        return f.value

    def rgrad(self):
        out = self._empty()
        _check(self._lib.msdp_rgrad(self._h, _dptr(out)))
        return np.ascontiguousarray(out)

    def _vec_op(self, fn, U):
        Ub = self._to_boundary(U)
        out = self._empty()
        _check(fn(self._h, _dptr(Ub), _dptr(out)))
        return np.ascontiguousarray(out)

    def hessvec(self, U):
        return self._vec_op(self._lib.msdp_hessvec, U)

    def proj(self, U):
        return self._vec_op(self._lib.msdp_proj, U)

    def retr(self, U):
        return self._vec_op(self._lib.msdp_retr, U)

    def debug_get_tcg_step(self):
        """(eta, Heta) of the last tCG solve (test hook, see msdp_debug_get_tcg_step)."""
        eta, heta = self._empty(), self._empty()
        _check(self._lib.msdp_debug_get_tcg_step(self._h, _dptr(eta), _dptr(heta)))
        return np.ascontiguousarray(eta), np.ascontiguousarray(heta)

    def get_z(self):
        z = np.zeros(self.n)
        _check(self._lib.msdp_get_z(self._h, _dptr(z)))
        return z

    def get_z_all(self):
        z = np.zeros(self.n)
        _check(self._lib.msdp_get_z_all(self._h, _dptr(z)))
        return z

    def linesearch_cost(self, U, alpha):
        v = C.c_double()
        if U is None:
            _check(self._lib.msdp_linesearch_cost(self._h, None, 0.0, C.byref(v)))
        else:
            Ub = self._to_boundary(U)
            _check(self._lib.msdp_linesearch_cost(self._h, _dptr(Ub), float(alpha), C.byref(v)))
        return v.value

    def linesearch_accept(self):
        _check(self._lib.msdp_linesearch_accept(self._h))

    def escape_eigs(self, k, tol=1e-10, maxit=500):
        lam = np.empty(k)
        V = np.empty((self.n, k), order="F")
        lmax = C.c_double()
        its = C.c_int32()
        _check(self._lib.msdp_escape_eigs(self._h, k, tol, maxit, _dptr(lam), _dptr(V), C.byref(lmax), C.byref(its)))
        return lam, np.ascontiguousarray(V), lmax.value, its.value

    def escape_eigs_matrix(self, S, k, tol=1e-10, maxit=20000):
        """k bottom eigenpairs and lambda_max of an explicit dense symmetric S (device Lanczos, dense GEMV)."""
        S = np.ascontiguousarray(S, dtype=np.float64)
        lam = np.empty(k)
        V = np.empty((self.n, k), order="F")
        lmax = C.c_double()
        its = C.c_int32()
        _check(self._lib.msdp_escape_eigs_matrix(self._h, _dptr(S), k, tol, maxit, _dptr(lam), _dptr(V), C.byref(lmax),
                                                 C.byref(its)))
        return lam, np.ascontiguousarray(V), lmax.value, its.value

    def escape_info(self):
        """(nvalid, converged, residual) of the last escape_eigs* call: how many returned pairs are real, whether
        every Lanczos run passed a stop test, and the worst relative residual of the runs that did not."""
        nv, cv, res = C.c_int32(), C.c_int32(), C.c_double()
        _check(self._lib.msdp_escape_info(self._h, C.byref(nv), C.byref(cv), C.byref(res)))
        return nv.value, bool(cv.value), res.value

    def set_option(self, name, value):
        """Run-time switch of this handle (see msdp_set_option in include/manisdp_hip.h)."""
        _check(self._lib.msdp_set_option(self._h, name.encode(), int(value)))

    def time_collective(self, which, reps=200):
        """Average stream time (us) of one collective call (msdp_debug_time_collective)."""
        v = C.c_double(0)
        _check(self._lib.msdp_debug_time_collective(self._h, which, reps, C.byref(v)))
        return v.value

    def collective_calls(self):
        """Collective calls issued on this handle's communicator so far (a grouped RCCL launch counts once)."""
        v = C.c_int64(0)
        _check(self._lib.msdp_debug_collective_calls(self._h, C.byref(v)))
        return int(v.value)

    def escape_method(self):
        """Which eigen-solver the last escape call ran: 1 = block Chebyshev-filtered subspace iteration, 0 = Lanczos."""
        v = C.c_int32()
        _check(self._lib.msdp_escape_method(self._h, C.byref(v)))
        return v.value

    def escape_lower_bound(self):
        """Lower estimate of lambda_min(S) from the last escape call; -inf unless that call was cold-started and
        undeflated (see msdp_escape_lower_bound in include/manisdp_hip.h: an estimate, not a certificate)."""
        v = C.c_double()
        _check(self._lib.msdp_escape_lower_bound(self._h, C.byref(v)))
        return v.value

    def get_kind(self):
        k = C.c_int32()
        _check(self._lib.msdp_get_kind(self._h, C.byref(k)))
        return k.value

    def get_dual_slack_block(self, row0, nb):
        """The nb x nb diagonal block of S that starts at row0 (multiblock: eig(S_i) block by block)."""
        S = np.empty((nb, nb))
        _check(self._lib.msdp_get_dual_slack_block(self._h, int(row0), int(nb), _dptr(S)))
        return S

    def get_dual_slack(self):
        """Dense S (n x n) of the last al_dual call."""
        S = np.empty((self.n, self.n))
        _check(self._lib.msdp_get_dual_slack(self._h, _dptr(S)))
        return S

    # ---- AL bookkeeping on the device (affine handles)
    def block_eigs(self, row0, nblk, k, method=0):
        """eig of the diagonal blocks (row0[b], nblk[b]) of the dual slack of the last al_dual call, on the device (msdp_block_eigs):
        returns (w, V) -- all eigenvalues, block after block and ascending inside a block; V[(rows of block b), c] = eigenvector of
        the block's c-th smallest eigenvalue, c < k.  method: 0 = tridiagonalisation + bisection + inverse iteration (k <= 8), else
        Jacobi; 1 = Jacobi; 2 = the tridiagonal method."""
        r0 = np.ascontiguousarray(row0, dtype=np.int64)
        nb = np.ascontiguousarray(nblk, dtype=np.int64)
        tot = int(nb.sum())
        w = np.empty(tot)
        V = np.empty((tot, max(int(k), 1)))
        _check(self._lib.msdp_block_eigs(self._h, len(nb), r0.ctypes.data_as(_i64p), nb.ctypes.data_as(_i64p), int(k), int(method), _dptr(w), _dptr(V)))
        return w, V[:, :int(k)]

    def al_primal(self, m):
        """obj = c'x and A x (length m) at the resident point, without forming X = YY' on the host."""
        obj = C.c_double()
        Ax = np.empty(m)
        _check(self._lib.msdp_al_primal(self._h, C.byref(obj), _dptr(Ax)))
        return obj.value, Ax

    def al_dual(self, y):
        """Build the dual slack S for the multipliers y on the device; returns z (n values for unitdiag, a scalar for
        unittrace, None for the generic kind).  S stays resident for escape_eigs_dual."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        if self.kind in (KIND_UNITDIAG, KIND_MULTIBLOCK):
            z = np.empty(self.n)
        elif self.kind == KIND_UNITTRACE:
            z = np.empty(1)
        else:
            z = None
        _check(self._lib.msdp_al_dual(self._h, _dptr(y), _dptr(z) if z is not None else None))
        if z is None:
            return None
        return z if self.kind in (KIND_UNITDIAG, KIND_MULTIBLOCK) else float(z[0])

    def escape_eigs_dual(self, k, tol=1e-10, maxit=20000):
        """k bottom eigenpairs and lambda_max of the device-resident S of the last al_dual call."""
        lam = np.empty(k)
        V = np.empty((self.n, k), order="F")
        lmax = C.c_double()
        its = C.c_int32()
        _check(self._lib.msdp_escape_eigs_dual(self._h, k, tol, maxit, _dptr(lam), _dptr(V), C.byref(lmax), C.byref(its)))
        return lam, np.ascontiguousarray(V), lmax.value, its.value

    # ---- multi-GPU
    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        _check(load().msdp_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def comm_init(self, nranks, rank, uid):
        buf = (C.c_char * 128).from_buffer_copy(uid)
        _check(self._lib.msdp_comm_init(self._h, nranks, rank, C.cast(buf, C.c_void_p)))

    def comm_init_local(self, nranks, rank, group):
        """Member `rank` of an in-process group of `nranks` handles on one GPU (one host thread per handle): the N-rank code
        paths with local stand-ins for the RCCL collectives."""
        _check(self._lib.msdp_comm_init_local(self._h, int(nranks), int(rank), int(group)))

    def comm_init_ipc(self, nranks, rank, name):
        """Member `rank` of a group of `nranks` PROCESSES (ranks on one GPU, or one per GPU with peer access): `name` is a POSIX
        shared-memory name ("/..."), the same on every member, fresh per group (msdp_comm_init_ipc)."""
        _check(self._lib.msdp_comm_init_ipc(self._h, int(nranks), int(rank), name.encode()))

    def local_rows(self):
        a, b = C.c_int64(), C.c_int64()
        _check(self._lib.msdp_local_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def point_snapshot(self):
        _check(self._lib.msdp_point_snapshot(self._h))

    def point_restore(self):
        _check(self._lib.msdp_point_restore(self._h))

    def last_rtr_device_ms(self):
        """Device time of the last rtr() call (HIP events on the library's stream; the fused launch on the fused path)."""
        v = C.c_double()
        _check(self._lib.msdp_debug_last_rtr_device_ms(self._h, C.byref(v)))
        return v.value

    def tcg_path(self):
        """1: persistent single-launch tCG kernel, 0: chunked hipGraph (three kernels per trip)."""
        v = C.c_int32()
        _check(self._lib.msdp_tcg_path(self._h, C.byref(v)))
        return v.value

    def persist_form(self):
        """Trip form of the persistent tCG kernel: 2 one grid reduction per trip, 1 early gather, 0 two reductions, -1 not persistent."""
        v = C.c_int32()
        _check(self._lib.msdp_debug_persist_form(self._h, C.byref(v)))
        return v.value

    # ---- measurement
    def persist_trace(self, reps=256):
        """Phase stamps of the persistent tCG trip: (array [G, nj, 8] of s_memtime ticks, first traced trip, trip time in ms).
        reps <= 0: the fused launch -- ((stamps of the TR iterations, stamps of the trips of one iteration), 0, call time in ms)."""
        cap = 512 * 64 * 8
        buf = (C.c_uint64 * cap)()
        dims = (C.c_int32 * 3)()
        ms = C.c_double()
        _check(self._lib.msdp_debug_persist_trace(self._h, reps, buf, cap, dims, C.byref(ms)))
        G, nj, j0 = dims[0], dims[1], dims[2]
        a = np.frombuffer(buf, dtype=np.uint64, count=G * nj * 8).reshape(G, nj, 8).astype(np.int64)
        if reps <= 0:
            # the fused launch: stamps of the TR iterations, then the trips of one of them (msdp_pipe.h MSDP_TRACE_KSEL)
            b = np.frombuffer(buf, dtype=np.uint64, count=2 * G * nj * 8)[G * nj * 8:].reshape(G, nj, 8).astype(np.int64)
            return (a, b), j0, ms.value
        return a, j0, ms.value

    def bench_hessvec(self, reps):
        ms, by, fl = C.c_double(), C.c_double(), C.c_double()
        _check(self._lib.msdp_bench_hessvec(self._h, reps, C.byref(ms), C.byref(by), C.byref(fl)))
        return ms.value, by.value, fl.value

    def bench_kernel(self, which, reps):
        ms = C.c_double()
        _check(self._lib.msdp_bench_kernel(self._h, which, reps, C.byref(ms)))
        return ms.value

    def bench_tcg_trip(self, reps):
        ms = C.c_double()
        _check(self._lib.msdp_bench_tcg_trip(self._h, reps, C.byref(ms)))
        return ms.value
