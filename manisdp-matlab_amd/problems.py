"""Instance readers and generators that feed the three ManiSDP entry points.

These are the callers' side of the hot path: they emit
exactly the SeDuMi-format data ``(At, b, c, K)`` / sparse ``C`` that the
reference entry points consume.  Each function cites the reference file it
restates.  Nothing here touches the GPU.

Conventions
-----------
* ``At`` is ``scipy.sparse.csc_matrix`` of shape ``(n*n, m)``; column ``k`` is the
  column-major ``vec`` of the symmetric constraint matrix ``A_k``
  (reference: src/basicfunction/bqpmom.m:91, example/example_theta.m:39).
* ``b`` is a dense ``(m,)`` float64 vector, ``c`` a dense ``(n*n,)`` vector or a
  sparse ``(n*n, 1)`` matrix, ``K = {"s": n}``.
"""
from __future__ import annotations

import gzip
import io
import os
import re
from itertools import combinations
from math import comb

import numpy as np
import scipy.sparse as sp

__all__ = [
    "read_gset", "gset_laplacian", "maxcut_cost_matrix", "toroidal_grid_maxcut",
    "from_sdpa", "get_basis", "bqpmom", "qsmom", "theta_problem", "generate_hamming",
    "dense_unitdiag_cost", "vec_index",
]


def _open_text(path):
    if str(path).endswith(".gz"):
        return io.TextIOWrapper(gzip.open(path, "rb"))
    return open(path, "r")


# --------------------------------------------------------------------------- Gset
def read_gset(path):
    """Read a Gset graph file: header ``nv ne`` then ``i j w`` lines, 1-based.

    Reference: src/basicfunction/Laplacian.m:2-4 (``readmatrix``)."""
    with _open_text(path) as fh:
        head = fh.readline().split()
        nv, ne = int(head[0]), int(head[1])
        edges = np.loadtxt(fh, ndmin=2, max_rows=ne)
    i = edges[:, 0].astype(np.int64) - 1
    j = edges[:, 1].astype(np.int64) - 1
    w = edges[:, 2].astype(np.float64)
    return nv, i, j, w


def gset_laplacian(path):
    """Graph Laplacian with the reference's semantics (Laplacian.m:5-11):
    off-diagonals are *assigned* ``-w`` (a repeated edge overwrites), the
    diagonal accumulates ``+w`` for every listed edge."""
    nv, i, j, w = read_gset(path)
    # "last assignment wins" for repeated (i,j): keep the last occurrence
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    key = lo * nv + hi
    _, last = np.unique(key[::-1], return_index=True)
    last = len(key) - 1 - last
    oi, oj, ow = i[last], j[last], w[last]
    diag = np.zeros(nv)
    np.add.at(diag, i, w)
    np.add.at(diag, j, w)
    rows = np.concatenate([oi, oj, np.arange(nv)])
    cols = np.concatenate([oj, oi, np.arange(nv)])
    vals = np.concatenate([-ow, -ow, diag])
    L = sp.coo_matrix((vals, (rows, cols)), shape=(nv, nv)).tocsr()
    return L


def maxcut_cost_matrix(path):
    """``C = -L/4`` as in example/example_maxcut.m:10-11 (sparse CSR, symmetric).

    Explicit zeros on the diagonal are dropped, as MATLAB's ``sparse`` does."""
    C = (-0.25 * gset_laplacian(path)).tocsr()
    C.eliminate_zeros()
    C.sort_indices()
    return C


def toroidal_grid_maxcut(rows, cols, seed=0):
    """Synthetic G81-shaped instance: 2-D toroidal grid, every vertex of degree 4,
    i.i.d. +-1 edge weights (Gset G81 is the 100 x 200 instance of this family).
    Returns ``C = -L/4``."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    idx = np.arange(n).reshape(rows, cols)
    right = np.roll(idx, -1, axis=1)
    down = np.roll(idx, -1, axis=0)
    i = np.concatenate([idx.ravel(), idx.ravel()])
    j = np.concatenate([right.ravel(), down.ravel()])
    w = rng.choice([-1.0, 1.0], size=i.size)
    diag = np.zeros(n)
    np.add.at(diag, i, w)
    np.add.at(diag, j, w)
    r = np.concatenate([i, j, np.arange(n)])
    c = np.concatenate([j, i, np.arange(n)])
    v = np.concatenate([-w, -w, diag])
    C = (-0.25 * sp.coo_matrix((v, (r, c)), shape=(n, n))).tocsr()
    C.eliminate_zeros()
    C.sort_indices()
    return C


# --------------------------------------------------------------------------- SDPA
def from_sdpa(path):
    """SDPA sparse format -> SeDuMi ``(At, b, c, K)`` for a single PSD block.

    Restates src/basicfunction/fromsdpa.m:40-155 for the case the three entry
    points accept (one semidefinite block, ``K.s = n``): ``c = -vec(F0)``
    (:127-130), matrices ``1..m`` become the columns of ``At`` (:141-144), both
    symmetrised from the upper triangle."""
    with _open_text(path) as fh:
        lines = [ln for ln in fh.read().splitlines()]
    # skip comment lines (start with " or *)
    lines = [ln for ln in lines if ln.strip() and ln.lstrip()[0] not in '"*']
    m = int(re.split(r"[\s=]+", lines[0].strip())[0])
    nblocks = int(re.split(r"[\s=]+", lines[1].strip())[0])
    dims = [int(t) for t in re.sub(r"[\.\,(){}]", " ", lines[2]).split()[:nblocks]]
    if nblocks != 1 or dims[0] <= 1:
        raise ValueError("from_sdpa: only a single semidefinite block is supported "
                         "(the three ManiSDP entry points take K.s = n)")
    n = dims[0]
    b = np.array([float(t) for t in re.sub(r"[\,(){}]", " ", lines[3]).split()[:m]])
    if b.size != m:
        raise ValueError("from_sdpa: right-hand side has the wrong dimension")
    E = np.loadtxt(io.StringIO("\n".join(lines[4:])), ndmin=2)
    matno = E[:, 0].astype(np.int64)
    ii = E[:, 2].astype(np.int64) - 1
    jj = E[:, 3].astype(np.int64) - 1
    vv = E[:, 4].astype(np.float64)
    off = ii != jj
    # column-major vec index of (i, j): i + j*n ; mirror entry for off-diagonals
    r = np.concatenate([ii + jj * n, (jj + ii * n)[off]])
    k = np.concatenate([matno, matno[off]])
    v = np.concatenate([vv, vv[off]])
    obj = k == 0
    c = -sp.coo_matrix((v[obj], (r[obj], np.zeros(obj.sum(), dtype=np.int64))),
                       shape=(n * n, 1)).tocsc()
    At = sp.coo_matrix((v[~obj], (r[~obj], k[~obj] - 1)), shape=(n * n, m)).tocsc()
    At.eliminate_zeros()
    c.eliminate_zeros()
    return At, b, c, {"s": n}


# ------------------------------------------------------------------ monomial bases
def get_basis(n, d):
    """All exponent vectors of degree <= d in n variables, as columns, in the
    reference's order (src/basicfunction/get_basis.m:1-33; order relation
    comp.m:1-23): graded; inside one degree ascending in
    ``(a_n, a_{n-1}, ..., a_1)``.  Returned as an ``(n, L)`` int8 array."""
    cols = []
    for deg in range(d + 1):
        # enumerate multisets of size deg over variables; sort by the comp order
        block = []
        for combo in _multisets(n, deg):
            a = np.zeros(n, dtype=np.int8)
            for v in combo:
                a[v] += 1
            block.append(a)
        block.sort(key=lambda a: tuple(a[::-1]))
        cols.extend(block)
    return np.array(cols, dtype=np.int8).T.reshape(n, -1)


def _multisets(n, k):
    from itertools import combinations_with_replacement
    return combinations_with_replacement(range(n), k)


def _get_basis_literal(n, d):
    """Line-by-line restatement of get_basis.m:1-33 (used by the tests to pin
    :func:`get_basis`'s ordering against the reference's own iteration)."""
    lb = comb(n + d, d)
    basis = np.zeros((n, lb), dtype=np.int64)
    i = 0
    t = 0  # 0-based column of "t" (MATLAB t = 1)
    while i < d + 1:
        t += 1
        if t >= lb:
            # MATLAB grows the array by one zero column here and leaves the loop
            # right after (i reaches d+1); we simply stop.
            if basis[n - 1, t - 1] == i:
                i += 1
                continue
            raise AssertionError("get_basis: ran past the expected length")
        if basis[n - 1, t - 1] == i:
            if i < d:
                basis[0, t] = i + 1
            i += 1
        else:
            j = 0
            while basis[j, t - 1] == 0:
                j += 1
            basis[:, t] = basis[:, t - 1]
            if j == 0:
                basis[0, t] -= 1
                basis[1, t] += 1
            else:
                basis[0, t] = basis[j, t] - 1
                basis[j, t] = 0
                basis[j + 1, t] += 1
    return basis


def vec_index(i, j, n):
    """0-based column-major linear index of entry (i, j) of an n x n matrix."""
    return i + j * n


def _mono_key(vars_sorted):
    return tuple(vars_sorted)


def _sp_order_key(mono, n):
    """Sort key implementing comp.m for a monomial given as a sorted variable
    multiset: (degree, a_n, ..., a_1)."""
    a = [0] * n
    for v in mono:
        a[v] += 1
    return (len(mono),) + tuple(a[::-1])


def bqpmom(n, Q, e):
    """Second-order moment relaxation of ``min x'Qx + e'x, x_i^2 = 1`` in SeDuMi
    format.  Restates src/basicfunction/bqpmom.m:6-117 (same basis order, same
    constraint order, same coefficients).  Returns ``(At, b, c, K)`` with ``c``
    sparse ``(mb^2, 1)`` like the reference (:114-115)."""
    Q = np.asarray(Q, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64).ravel()
    # basis: multilinear monomials of degree <= 2 (bqpmom.m:7-15), in get_basis order
    basis = [()] + [(k,) for k in range(n)]
    # degree 2, ordered by (a_n..a_1): for i=1..n-1 (0-based second var), j<i
    basis += [(j, i) for i in range(1, n) for j in range(i)]
    mb = len(basis)
    assert mb == 1 + n + n * (n - 1) // 2
    # sp: monomials of degree <= 4, exponents <= 2, at least one odd exponent (:16-23)
    sp_list = []
    for deg in range(1, 5):
        for combo in _multisets(n, deg):
            cnt = {}
            for v in combo:
                cnt[v] = cnt.get(v, 0) + 1
            if max(cnt.values()) > 2:
                continue
            if all(c % 2 == 0 for c in cnt.values()):
                continue
            sp_list.append(combo)
    sp_list.sort(key=lambda mno: _sp_order_key(mno, n))
    lsp = len(sp_list)
    sp_index = {mno: k for k, mno in enumerate(sp_list)}
    # mm{ind}: pairs (i<j) of basis elements whose product is sp(ind) (:25-32)
    mm = [[] for _ in range(lsp)]
    for i in range(mb):
        bi = basis[i]
        for j in range(i + 1, mb):
            mono = tuple(sorted(bi + basis[j]))
            mm[sp_index[mono]].append((i, j))
    ncons = mb * (mb + 1) // 2 - lsp + n * (mb - 1) - mb + 1
    rows, cols, vals = [0], [0], [1.0]
    b = np.zeros(ncons)
    b[0] = 1.0
    # X11/2 - Xii/2 = 0 for the degree-1 diagonal (:39-43)
    for i in range(1, n + 1):
        rows += [0, i * mb + i]
        cols += [i, i]
        vals += [0.5, -0.5]
    l = n + 1
    # two diagonal ties per degree-2 basis element (:46-52)
    for i in range(n + 1, mb):
        c1, c2 = basis[i][0] + 1, basis[i][1] + 1
        rows += [c1 * mb + c1, i * mb + i, c2 * mb + c2, i * mb + i]
        cols += [l, l, l + 1, l + 1]
        vals += [0.5, -0.5, 0.5, -0.5]
        l += 2
    # loa{i}: linear indices of both orientations of every pair (:53-59)
    loa = []
    for k in range(lsp):
        arr = []
        for (i, j) in mm[k]:
            arr += [j * mb + i, i * mb + j]
        loa.append(arr)
    # x_k^2 * x^alpha == x^alpha (:60-78)
    for k in range(n):
        for i in range(1, mb):
            if k not in basis[i]:
                ind1 = sp_index[tuple(sorted(basis[i] + (k, k)))]
                ind2 = sp_index[basis[i]]
                l1, l2 = len(loa[ind1]), len(loa[ind2])
                rows += loa[ind1] + loa[ind2]
                cols += [l] * (l1 + l2)
                if l1 < l2:
                    vals += [1.0] * l1 + [-l1 / l2] * l2
                else:
                    vals += [l2 / l1] * l1 + [-1.0] * l2
                l += 1
    # equal-monomial ties inside each class (:80-90)
    for k in range(lsp):
        firsts = [pr[0] for pr in mm[k]]
        idx = int(np.argmax(firsts))  # first maximum, like MATLAB's max
        for j in range(len(mm[k])):
            if j != idx:
                rows += loa[k][2 * idx:2 * idx + 2] + loa[k][2 * j:2 * j + 2]
                cols += [l, l, l, l]
                vals += [0.5, 0.5, -0.5, -0.5]
                l += 1
    assert l == ncons, (l, ncons)
    At = sp.coo_matrix((np.array(vals), (np.array(rows), np.array(cols))),
                       shape=(mb * mb, ncons)).tocsc()
    # objective (:93-114)
    rows = list(range(1, n + 1))
    cols = list(range(1, n + 1))
    vals = list(np.diag(Q))
    for i in range(n):
        for (a, bb) in mm[i]:
            rows += [a, bb]
            cols += [bb, a]
        vals += [e[i] / (2 * len(mm[i]))] * (2 * len(mm[i]))
    ind = n
    for i in range(1, n):
        for j in range(i):
            for (a, bb) in mm[ind]:
                rows += [a, bb]
                cols += [bb, a]
            vals += [Q[j, i] / len(mm[ind])] * (2 * len(mm[ind]))
            ind += 1
    C = sp.coo_matrix((np.array(vals), (np.array(rows), np.array(cols))), shape=(mb, mb)).tocsc()
    c = C.reshape((mb * mb, 1), order="F").tocsc()
    return At, b, c, {"s": mb}


def bqpsos(Q, e, n):
    """Second-order SOS relaxation of ``min x'Qx + e'x, x_i^2 = 1`` (the dual of :func:`bqpmom`'s problem) as
    ``(A, b, dAAt, mb)``: ``A`` is ``lsp x mb^2`` with one row per multilinear monomial of degree <= 4 (row 0 = the
    identity), ``dAAt = diag(A A')``.  Restates src/basicfunction/bqpsos.m:7-39 (same monomial order, same rows)."""
    Q = np.asarray(Q, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64).ravel()
    from itertools import combinations
    spl = [()]
    for deg in range(1, 5):                                # multilinear monomials of degree <= 4 (:8-11), comp.m order
        spl += sorted(combinations(range(n), deg), key=lambda mno: _sp_order_key(mno, n))
    lsp = len(spl)
    index = {mno: k for k, mno in enumerate(spl)}
    mb = comb(n + 2, 2) - n                                # :12  the monomials of degree <= 2 come first in sp
    rows = [0] * mb
    cols = [k * mb + k for k in range(mb)]                 # :20  row 1 = identity
    dAAt = np.zeros(lsp)
    dAAt[0] = mb
    for i in range(mb):                                    # :22-33
        si = set(spl[i])
        for j in range(i + 1, mb):
            loc = index[tuple(sorted(si.symmetric_difference(spl[j])))]    # mod(sp_i + sp_j, 2)
            rows += [loc, loc]
            cols += [i * mb + j, j * mb + i]
            dAAt[loc] += 2
    A = sp.coo_matrix((np.ones(len(rows)), (np.array(rows), np.array(cols))), shape=(lsp, mb * mb)).tocsr()
    b = np.zeros(lsp)                                      # :36-39
    b[0] = np.trace(Q)
    b[1:n + 1] = e
    b[n + 1:n + 1 + n * (n - 1) // 2] = [2.0 * Q[i, j] for j in range(1, n) for i in range(j)]
    return A, b, dAAt, mb


def bqpsos_dual_problem(Q, e, n):
    """The dual-form SeDuMi data example/dual/example_bqp_dual.m:21-37 passes to ``ManiDSDP_unitdiag``: one free
    variable (column ``e_1``, cost 1), ``b`` scaled by ``max|b|``.  Returns ``(A, b, c, K, dAAt, maxb)``; the optimum of
    the BQP relaxation is ``maxb * obj``."""
    A, b, dAAt, mb = bqpsos(Q, e, n)
    v = sp.csr_matrix(([1.0], ([0], [0])), shape=(A.shape[0], 1))
    Afull = sp.hstack([v, A]).tocsc()
    c = np.zeros(1 + mb * mb)
    c[0] = 1.0
    maxb = float(np.max(np.abs(b)))
    return Afull, b / maxb, c, {"f": 1, "s": mb}, dAAt, maxb


def qsmom(n, coe):
    """Second-order moment relaxation of ``min coe'[x]_4, |x|^2 = 1`` in SeDuMi
    format.  Restates src/basicfunction/qsmom.m:6-116 (the reference's own example solves it
    with the generic ManiSDP, example/example_qsphere.m:18-27)."""
    coe = np.asarray(coe, dtype=np.float64).ravel()
    basis_arr = get_basis(n, 2)
    mb = basis_arr.shape[1]
    basis = [tuple(v for v in range(n) for _ in range(int(basis_arr[v, k]))) for k in range(mb)]
    sp_arr = get_basis(n, 4)
    lsp = sp_arr.shape[1]
    sp_list = [tuple(v for v in range(n) for _ in range(int(sp_arr[v, k]))) for k in range(lsp)]
    sp_index = {mno: k for k, mno in enumerate(sp_list)}
    if coe.size != lsp:
        raise ValueError(f"qsmom: coe must have length C(n+4,4) = {lsp}")
    mm = [[] for _ in range(lsp)]
    for i in range(mb):
        for j in range(i, mb):
            mm[sp_index[tuple(sorted(basis[i] + basis[j]))]].append((i, j))
    ncons = mb * (mb + 1) // 2 - lsp + mb + 1
    rows, cols, vals = [0], [0], [1.0]
    b = np.zeros(ncons)
    b[0] = 1.0
    l = 1
    loa = []
    for k in range(lsp):
        arr = []
        for (i, j) in mm[k]:
            arr += [j * mb + i, i * mb + j]
        loa.append(arr)

    def class_rows(ind):
        out = []
        for t, (i, j) in enumerate(mm[ind]):
            if i == j:
                out.append(loa[ind][2 * t + 1])
            else:
                out += loa[ind][2 * t:2 * t + 2]
        return out

    # sphere constraint times every basis element (:33-65)
    for i in range(mb):
        for k in range(n):
            r1 = class_rows(sp_index[tuple(sorted(basis[i] + (k, k)))])
            rows += r1
            cols += [l] * len(r1)
            vals += [1.0 / len(r1)] * len(r1)
        r2 = class_rows(sp_index[basis[i]])
        rows += r2
        cols += [l] * len(r2)
        vals += [-1.0 / len(r2)] * len(r2)
        l += 1
    # equal-monomial ties (:67-92)
    for k in range(lsp):
        firsts = [pr[0] for pr in mm[k]]
        idx = int(np.argmax(firsts))
        for j in range(len(mm[k])):
            if j == idx:
                continue
            if mm[k][idx][0] == mm[k][idx][1]:
                rows += [loa[k][2 * idx + 1]]; cols += [l]; vals += [1.0]
            else:
                rows += loa[k][2 * idx:2 * idx + 2]; cols += [l, l]; vals += [0.5, 0.5]
            if mm[k][j][0] == mm[k][j][1]:
                rows += [loa[k][2 * j + 1]]; cols += [l]; vals += [-1.0]
            else:
                rows += loa[k][2 * j:2 * j + 2]; cols += [l, l]; vals += [-0.5, -0.5]
            l += 1
    assert l == ncons, (l, ncons)
    At = sp.coo_matrix((np.array(vals), (np.array(rows), np.array(cols))),
                       shape=(mb * mb, ncons)).tocsc()
    rows, cols, vals = [], [], []
    for k in range(lsp):
        s = 0
        for (i, j) in mm[k]:
            if i == j:
                rows.append(i); cols.append(j); s += 1
            else:
                rows += [i, j]; cols += [j, i]; s += 2
        vals += [coe[k] / s] * s
    C = sp.coo_matrix((np.array(vals), (np.array(rows), np.array(cols))), shape=(mb, mb)).tocsc()
    c = C.reshape((mb * mb, 1), order="F").tocsc()
    return At, b, c, {"s": mb}


# -------------------------------------------------------------------------- sensor network localization
def snl_polynomial(n, seed=1, radius2=0.5):
    """The quartic of example/Sensor_Network_Localization.m:2-31: ``n`` sensors at random positions in the unit square, the four anchors
    of the example, an edge wherever two sensors (or, for the LAST sensor as in the example's ``for i = n``, sensor and anchor) are
    within squared distance ``radius2``; ``f = sum (|p_i - p_j|^2 - d_ij^2)^2`` over the edges in the 2n variables
    ``(x_1..x_n, y_1..y_n)``.  Returns ``(f, loc)``: f as a dict {sorted variable tuple: coefficient} (degree <= 4), loc 2 x n.
    (NumPy's generator, not MATLAB's: the same family, not the same instance.)"""
    rng = np.random.default_rng(seed)
    loc = rng.random((2, n))
    anchors = np.array([[0.25, 0.75, 0.3, 0.8], [0.75, 0.25, 0.8, 0.3]])

    def mul(p, q):
        out = {}
        for ma, ca in p.items():
            for mq, cq in q.items():
                m = tuple(sorted(ma + mq))
                out[m] = out.get(m, 0.0) + ca * cq
        return out

    def add(p, q, sq=1.0):
        for m, cq in q.items():
            p[m] = p.get(m, 0.0) + sq * cq
        return p

    f = {}
    for i in range(n - 1):
        for j in range(i + 1, n):
            d2 = float(np.sum((loc[:, i] - loc[:, j]) ** 2))
            if d2 <= radius2:
                dx = {(i,): 1.0, (j,): -1.0}; dy = {(n + i,): 1.0, (n + j,): -1.0}
                q = add(add(mul(dx, dx), mul(dy, dy)), {(): -d2})
                add(f, mul(q, q))
    i = n - 1
    for j in range(anchors.shape[1]):
        d2 = float(np.sum((loc[:, i] - anchors[:, j]) ** 2))
        if d2 <= radius2:
            dx = {(i,): 1.0, (): -float(anchors[0, j])}; dy = {(n + i,): 1.0, (): -float(anchors[1, j])}
            q = add(add(mul(dx, dx), mul(dy, dy)), {(): -d2})
            add(f, mul(q, q))
    return {m: c for m, c in f.items() if c != 0.0}, loc


def snl_mom(f, nvars):
    """Second-order moment relaxation of ``min f(x)`` for a quartic ``f`` (dict {sorted variable tuple: coefficient}) in ``nvars``
    variables, one clique holding all of them, in SeDuMi format: what src/basicfunction/snl_mom_sparse.m:4-96 builds for
    ``cliques = {1:nvars}`` (example/Sensor_Network_Localization.m:34-35).  Basis = all monomials of degree <= 2 (get_basis order);
    constraints: ``X_11 = 1`` and one tie per extra entry of every monomial class (:46-74, the class representative is the entry
    with the largest first index, +-1 on a diagonal entry, +-1/2 on a symmetric pair); ``c`` spreads every coefficient of f evenly
    over the entries of its class (:78-93).  As in the reference, At has ``mb (mb + 1) / 2 - lsp + mb + 1`` columns (:38) of which the
    last ``mb`` stay empty."""
    n = int(nvars)
    basis_arr = get_basis(n, 2)
    mb = basis_arr.shape[1]
    basis = [tuple(v for v in range(n) for _ in range(int(basis_arr[v, k]))) for k in range(mb)]
    classes = {}
    for i in range(mb):
        for j in range(i, mb):
            classes.setdefault(tuple(sorted(basis[i] + basis[j])), []).append((i, j))
    # the reference sorts the degree-4 monomials by rows of exponents (sortrows): the order of the tie constraints
    def expo(m):
        e = [0] * n
        for v in m:
            e[v] += 1
        return tuple(e)
    order = sorted(classes, key=expo)
    lsp = len(order)
    assert lsp == comb(n + 4, 4)
    ncons = mb * (mb + 1) // 2 - lsp + mb + 1
    rows, cols, vals = [0], [0], [1.0]
    b = np.zeros(ncons); b[0] = 1.0
    l = 1
    for mon in order:
        lst = classes[mon]
        idx = int(np.argmax([pr[0] for pr in lst]))
        for q, (i, j) in enumerate(lst):
            if q == idx:
                continue
            for (ii, jj), sgn in ((lst[idx], 1.0), ((i, j), -1.0)):
                if ii == jj:
                    rows.append(ii * mb + ii); cols.append(l); vals.append(sgn)
                else:
                    rows += [jj * mb + ii, ii * mb + jj]; cols += [l, l]; vals += [0.5 * sgn, 0.5 * sgn]
            l += 1
    assert l == ncons - mb
    At = sp.coo_matrix((np.array(vals), (np.array(rows), np.array(cols))), shape=(mb * mb, ncons)).tocsc()
    c = np.zeros(mb * mb)
    for mon, coef in f.items():
        spots = []
        for (i, j) in classes[tuple(sorted(mon))]:
            spots += [i * mb + i] if i == j else [j * mb + i, i * mb + j]
        c[spots] = coef / len(spots)
    return At, b, sp.csc_matrix(c.reshape(-1, 1)), {"s": mb}


# -------------------------------------------------------------------------- theta
def theta_problem(n, ndraws=None, seed=1):
    """Lovasz-theta-like unit-trace SDP of example/example_theta.m:2-39:
    random edge set Omega (i<j, de-duplicated, sorted by rows), ``C = -ones``,
    constraints ``X_ij + X_ji = 0`` on Omega and ``tr X = 1`` as the LAST column,
    ``b = [0 ... 0 1]``.  (NumPy RNG replaces MATLAB's ``randi``.)"""
    rng = np.random.default_rng(seed)
    if ndraws is None:
        ndraws = 10 * n
    om = rng.integers(0, n, size=(ndraws, 2))
    om = om[om[:, 0] < om[:, 1]]
    om = np.unique(om, axis=0)
    m = om.shape[0]
    i, j = om[:, 0], om[:, 1]
    rows = np.concatenate([i * n + j, j * n + i, np.arange(n) * n + np.arange(n)])
    cols = np.concatenate([np.arange(m), np.arange(m), np.full(n, m)])
    vals = np.ones(rows.size)
    At = sp.coo_matrix((vals, (rows, cols)), shape=(n * n, m + 1)).tocsc()
    b = np.zeros(m + 1)
    b[m] = 1.0
    c = -np.ones(n * n)
    return At, b, c, {"s": n}


def generate_hamming(k, d):
    """Theta function of the Hamming graph H_{k,d} in SeDuMi format -- example/generate_hamming.m:24-59 (the SDPLIB ``hamming_*``
    generator): vertices = the 2^k bit patterns, an edge where the Hamming distance is in ``d``; constraint 1 is ``tr X = 1``
    (``b(1) = 1``), then one constraint ``X_ij + X_ji = 0`` per edge in the order the reference enumerates them (vertex
    ``i`` ascending, its neighbours ``j > i`` in the order of the bit patterns); ``c = -vec(1 - Adj)``.  A unit-trace problem:
    ``ManiSDP_unittrace(At, b, c, K)`` returns ``-theta(H_{k,d})``.  Returns ``At`` (the reference returns ``A``)."""
    n = 1 << int(k)
    bitpat = []
    for dist in np.atleast_1d(d):
        for comb_ in combinations(range(int(k)), int(dist)):       # nchoosek(1:k, i): rows in lexicographic order
            bitpat.append(sum(1 << q for q in comb_))
    bitpat = np.array(bitpat, dtype=np.int64)
    Adj = sp.lil_matrix((n, n))
    ai, aj = [], []
    start = 0
    for i in range(n):
        nb = np.bitwise_xor(i, bitpat)
        nb = nb[nb > i]
        if nb.size:
            Adj[i, nb] = 1.0
            Adj[nb, i] = 1.0
            rows = np.arange(start, start + nb.size)
            ai += [rows, rows]
            aj += [nb * n + i, nb + i * n]
            start += nb.size
    m = start + 1
    rows = np.concatenate([np.arange(n) * n + np.arange(n)] + ([np.concatenate(aj)] if aj else []))
    cols = np.concatenate([np.zeros(n, dtype=np.int64)] + ([np.concatenate(ai) + 1] if ai else []))
    At = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n * n, m)).tocsc()
    c = -(1.0 - np.asarray(Adj.todense())).ravel(order="F")
    b = np.zeros(m)
    b[0] = 1.0
    return At, b, c, {"s": n}


def _quat_rotation(q):
    """Rotation matrix of the unit quaternion q = (v, s), vector part first (the convention of the QUASAR papers)."""
    v, s = np.asarray(q[:3], float), float(q[3])
    V = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return (s * s - v @ v) * np.eye(3) + 2.0 * np.outer(v, v) + 2.0 * s * V


def wahba_with_outliers(N, outlier_rate=0.5, sigma=0.01, seed=1):
    """Synthetic rotation-search data in the manner of example/example_rotationsearch.m:10-20 (its generator, createWahbaProblem,
    belongs to the STRIDE package and is not in the reference tree; this is the set-up the example's parameters describe): N unit
    vectors a_i, a random rotation R_gt, b_i = R_gt a_i + noise of covariance sigma^2 I for the inliers, random unit vectors for
    round(N * outlier_rate) outliers; noise bound beta = sigma * sqrt(chi2inv(0.9999, 3)).  Returns (a, b, R_gt, beta, outlier mask)."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((N, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    R = _quat_rotation(q)
    b = a @ R.T + sigma * rng.standard_normal((N, 3))
    nout = int(round(N * outlier_rate))
    out = np.zeros(N, dtype=bool)
    out[rng.permutation(N)[:nout]] = True
    bo = rng.standard_normal((nout, 3)); bo /= np.linalg.norm(bo, axis=1, keepdims=True)
    b[out] = bo
    beta = sigma * np.sqrt(21.10751347)                            # chi2inv(0.9999, 3)
    return a, b, R, beta, out


def quasar_cost_blocks(a, b, betasq, cbar2=1.0):
    """The 4 x 4 matrices Q_i with ||b_i - R(q) a_i||^2 = q' Q_i q for a unit quaternion q = (v, s):
    b'R(q)a = q' M q, M = [[a b' + b a' - (a'b) I, a x b], [(a x b)', a'b]], Q_i = (|a_i|^2 + |b_i|^2) I - 2 M_i."""
    Q = []
    for ai, bi in zip(a, b):
        M = np.zeros((4, 4))
        M[:3, :3] = np.outer(ai, bi) + np.outer(bi, ai) - (ai @ bi) * np.eye(3)
        M[:3, 3] = M[3, :3] = np.cross(ai, bi)
        M[3, 3] = ai @ bi
        Q.append((ai @ ai + bi @ bi) * np.eye(4) - 2.0 * M)
    return Q


def quasar_problem(a, b, betasq, cbar2=1.0, redundant=True):
    """QUASAR relaxation of the truncated-least-squares rotation search (Yang & Carlone, ICCV 2019: the SDP that
    example/example_rotationsearch.m:26-28 builds with STRIDE's QUASAR_Problem) in SeDuMi format.
        min_{R, theta_i = +-1}  sum_i (1 + theta_i)/2 * ||b_i - R a_i||^2 / beta^2 + (1 - theta_i)/2 * cbar^2
    With x = [q; theta_1 q; ...; theta_N q] (4(N+1) entries) the cost is x'Cx and Z = xx' satisfies: tr Z_00 = 1; Z_ii = Z_00
    (10 equalities per i); Z_0i symmetric (6 per i); and, redundant, Z_ij symmetric for 0 < i < j (6 per pair).
    Blocks of C: C_00 = sum_i (Q_i / beta^2 + cbar^2 I) / 2, C_0i = C_i0 = (Q_i / beta^2 - cbar^2 I) / 4.
    tr Z = N + 1: ManiSDP_unittrace(At, b / (N + 1), c, K) solves for X = Z / (N + 1), as the example does (:37).
    Returns (At, b, c, K)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    N = a.shape[0]
    n = 4 * (N + 1)
    Q = quasar_cost_blocks(a, b, betasq, cbar2)
    C = np.zeros((n, n))
    for i, Qi in enumerate(Q, start=1):
        C[:4, :4] += 0.5 * (Qi / betasq + cbar2 * np.eye(4))
        blk = 0.25 * (Qi / betasq - cbar2 * np.eye(4))
        C[:4, 4 * i:4 * i + 4] = blk
        C[4 * i:4 * i + 4, :4] = blk.T
    rows, cols, vals = [], [], []
    k = 0

    def put(r, c_, v):
        rows.append(c_ * n + r); cols.append(k); vals.append(v)      # column-major vec, both triangles of a symmetric A_k

    for d_ in range(4):                                              # tr Z_00 = 1
        put(d_, d_, 1.0)
    k += 1
    for i in range(1, N + 1):                                        # Z_ii = Z_00
        o = 4 * i
        for p_ in range(4):
            for q_ in range(p_, 4):
                if p_ == q_:
                    put(o + p_, o + p_, 1.0); put(p_, p_, -1.0)
                else:
                    put(o + p_, o + q_, 0.5); put(o + q_, o + p_, 0.5); put(p_, q_, -0.5); put(q_, p_, -0.5)
                k += 1
    pairs = [(0, i) for i in range(1, N + 1)]
    if redundant:
        pairs += [(i, j) for i in range(1, N + 1) for j in range(i + 1, N + 1)]
    for i, j in pairs:                                               # Z_ij symmetric
        oi, oj = 4 * i, 4 * j
        for p_ in range(4):
            for q_ in range(p_ + 1, 4):
                put(oi + p_, oj + q_, 0.5); put(oj + q_, oi + p_, 0.5)
                put(oi + q_, oj + p_, -0.5); put(oj + p_, oi + q_, -0.5)
                k += 1
    m = k
    At = sp.coo_matrix((vals, (rows, cols)), shape=(n * n, m)).tocsc()
    bvec = np.zeros(m)
    bvec[0] = 1.0
    return At, bvec, C.reshape(-1, order="F"), {"s": n}


def quasar_recover(X, N):
    """Rotation and inlier decisions from a (near) rank-one solution X ~ xx' / (N + 1): q = leading eigenvector of the 00 block,
    theta_i = sign of tr X_0i."""
    w, V = np.linalg.eigh(0.5 * (X[:4, :4] + X[:4, :4].T))
    q = V[:, -1]
    theta = np.array([np.sign(np.trace(X[:4, 4 * i:4 * i + 4])) for i in range(1, N + 1)])
    return _quat_rotation(q / np.linalg.norm(q)), theta


def chain_cliques(t, q):
    """``t`` cliques of ``q`` consecutive variables, neighbours sharing two (reference example/example_bqp_sparse.m:3-10):
    ``n = q + (q-2)(t-1)`` variables, clique i = {(q-2) i, ..., (q-2) i + q - 1} (0-based)."""
    return [list(range((q - 2) * i, (q - 2) * i + q)) for i in range(t)], q + (q - 2) * (t - 1)


def bqp_sparse_monomials(cliques):
    """The monomials of a sparse BQP objective: ``x_a`` and ``x_a x_b`` (a < b) with both variables in one clique, as
    sorted tuples, in lexicographic order (example_bqp_sparse.m:11-18 without the constant)."""
    mons = set()
    for I in cliques:
        for a in I:
            mons.add((a,))
        for ia, a in enumerate(I):
            for bb in I[ia + 1:]:
                mons.add((a, bb))
    return sorted(mons)


def bqpmom_sparse(n, cliques, coe):
    """Second-order moment relaxation of a BQP with correlative sparsity, one PSD block per clique, for
    ``ManiSDP_multiblock`` (what src/basicfunction/bqpmom_sparse.m:6-133 builds; written from the definition, not its loops):

    * block k is the moment matrix of the basis ``[1, x_a (a in I_k), x_a x_b (a < b in I_k)]`` (pairs ordered by their
      larger variable), ``K['s'][k] = 1 + |I_k| + |I_k|(|I_k|-1)/2``, all blocks with unit diagonal (``K['nob'] = t``);
    * entries are indexed by UNREDUCED products (exponents 0..2); the constraints are: the (1,1) entry of block 1 is 1; every
      other diagonal entry of a constant / degree-1 basis element equals it; the diagonal entry of a pair equals those of its
      two variables; ``L(x_a^2 m) = L(m)`` for every clique variable a and basis monomial m != 1 of its block without a
      (both sides averaged over all entries that carry the monomial); all entries carrying one monomial are equal;
    * ``coe`` gives the coefficients of ``bqp_sparse_monomials(cliques)``, each spread evenly over the entries that carry
      its monomial.

    Returns ``At (sum s_k^2 x m, CSC), b, c, K``; every moment vector of a point in {-1, 1}^n satisfies ``At' x = b`` and
    ``c' x = f(x)`` (tests/test_problems.py)."""
    t = len(cliques)
    bases = []
    for I in cliques:
        bs = [()] + [(a,) for a in I]
        for jb in range(1, len(I)):
            for ia in range(jb):
                bs.append((I[ia], I[jb]))
        bases.append(bs)
    mb = [len(bs) for bs in bases]
    off = np.concatenate([[0], np.cumsum([v * v for v in mb])]).astype(np.int64)

    def entry(k, i, j):                                    # position of entry (i, j) of block k in the stacked vecs
        return int(off[k] + j * mb[k] + i)

    def product(u, v):                                     # exponent pattern of the unreduced product, as a sorted tuple
        return tuple(sorted(u + v))

    # every off-diagonal entry (i < j) grouped by the monomial it carries; diagonal entries carry squares
    where = {}
    for k, bs in enumerate(bases):
        for i in range(mb[k]):
            for j in range(i + 1, mb[k]):
                where.setdefault(product(bs[i], bs[j]), []).append((i, j, k))
    rows, cols, vals = [entry(0, 0, 0)], [0], [1.0]
    m = 1

    def add(pairs):
        nonlocal m
        for r, v in pairs:
            rows.append(r); cols.append(m); vals.append(v)
        m += 1

    for k, I in enumerate(cliques):                         # unit diagonal, constant and degree-1 part
        for i in range(1 if k == 0 else 0, len(I) + 1):
            add([(entry(0, 0, 0), 0.5), (entry(k, i, i), -0.5)])
    for k, I in enumerate(cliques):                         # unit diagonal, pairs: tied to both variables
        pos = {a: 1 + ia for ia, a in enumerate(I)}
        for i in range(len(I) + 1, mb[k]):
            for a in bases[k][i]:
                add([(entry(k, pos[a], pos[a]), 0.5), (entry(k, i, i), -0.5)])

    def spots(mon):
        out = []
        for (i, j, k) in where[mon]:
            out += [entry(k, i, j), entry(k, j, i)]
        return out

    for k, I in enumerate(cliques):                         # x_a^2 m = m
        for a in I:
            for i in range(1, mb[k]):
                mon = bases[k][i]
                if a in mon:
                    continue
                hi, lo = spots(product(mon, (a, a))), spots(mon)
                if len(hi) < len(lo):
                    add([(r, 1.0) for r in hi] + [(r, -len(hi) / len(lo)) for r in lo])
                else:
                    add([(r, len(lo) / len(hi)) for r in hi] + [(r, -1.0) for r in lo])
    for mon in sorted(where):                               # one value per monomial
        lst = where[mon]
        ref = max(range(len(lst)), key=lambda q: (lst[q][0], -q))      # the first entry with the largest row index
        i0, j0, k0 = lst[ref]
        for q, (i, j, k) in enumerate(lst):
            if q != ref:
                add([(entry(k0, i0, j0), 0.5), (entry(k0, j0, i0), 0.5), (entry(k, i, j), -0.5), (entry(k, j, i), -0.5)])
    At = sp.csc_matrix((vals, (rows, cols)), shape=(int(off[-1]), m))
    b = np.zeros(m); b[0] = 1.0
    c = np.zeros(int(off[-1]))
    mons = bqp_sparse_monomials(cliques)
    coe = np.asarray(coe, dtype=np.float64).ravel()
    if coe.size != len(mons):
        raise ValueError("bqpmom_sparse: %d coefficients for %d monomials" % (coe.size, len(mons)))
    for mon, v in zip(mons, coe):
        lo = spots(mon)
        c[lo] += v / len(lo)
    return At, b, c, {"s": mb, "nob": t}


def quartic_sparse_monomials(cliques):
    """All monomials of degree <= 4 whose variables lie in one clique, as sorted tuples (() = the constant), in lexicographic
    order: the support of the sparse quartic of example/example_qsphere_sparse.m:11-16."""
    import itertools
    mons = set()
    for I in cliques:
        for d in range(5):
            mons.update(itertools.combinations_with_replacement(I, d))
    return sorted(mons)


def qsmom_sparse(n, cliques, coe):
    """Second-order moment relaxation of a quartic with correlative sparsity whose clique sub-vectors all lie on unit spheres
    (``|x_{I_k}| = 1`` for every k), one PSD block per clique, for ``ManiSDP_multiblock`` with ``K['nob'] = 0`` (what
    src/basicfunction/qsmom_sparse.m:6-121 builds; written from the definition):

    * block k is the moment matrix of all monomials of degree <= 2 in ``x_{I_k}`` (``[1, x_a, x_a x_b (a <= b, ordered by b)]``);
    * constraints: ``L(1) = 1``; for every clique k and basis monomial m of its block ``sum_a L(x_a^2 m) - L(m) = 0`` (a over
      ``I_k``; every ``L`` the average of all entries that carry the monomial); all entries carrying one monomial are equal;
    * ``coe`` gives the coefficients of ``quartic_sparse_monomials(cliques)`` (constant included), each spread evenly over the
      entries that carry its monomial.

    Returns ``At, b, c, K``; the moments of every point with ``|x_{I_k}| = 1`` for all k satisfy ``At' x = b`` and give
    ``c' x = f(x)`` (tests/test_problems.py)."""
    t = len(cliques)
    bases = []
    for I in cliques:
        bs = [()] + [(a,) for a in I]
        for jb in range(len(I)):
            for ia in range(jb + 1):
                bs.append((I[ia], I[jb]))
        bases.append(bs)
    mb = [len(bs) for bs in bases]
    off = np.concatenate([[0], np.cumsum([v * v for v in mb])]).astype(np.int64)

    def entry(k, i, j):
        return int(off[k] + j * mb[k] + i)

    where = {}
    for k, bs in enumerate(bases):
        for i in range(mb[k]):
            for j in range(i, mb[k]):
                where.setdefault(tuple(sorted(bs[i] + bs[j])), []).append((i, j, k))

    def spots(mon):                                        # every entry carrying the monomial (diagonal ones once)
        out = []
        for (i, j, k) in where[mon]:
            out += [entry(k, i, i)] if i == j else [entry(k, i, j), entry(k, j, i)]
        return out

    rows, cols, vals = [entry(0, 0, 0)], [0], [1.0]
    m = 1
    for k, I in enumerate(cliques):                         # (|x_I|^2 - 1) m = 0
        for i in range(mb[k]):
            for a in I:
                hi = spots(tuple(sorted(bases[k][i] + (a, a))))
                rows += hi; cols += [m] * len(hi); vals += [1.0 / len(hi)] * len(hi)
            lo = spots(bases[k][i])
            rows += lo; cols += [m] * len(lo); vals += [-1.0 / len(lo)] * len(lo)
            m += 1
    for mon in sorted(where):                               # one value per monomial
        lst = where[mon]
        ref = max(range(len(lst)), key=lambda q: (lst[q][0], -q))
        for q, (i, j, k) in enumerate(lst):
            if q == ref:
                continue
            for (ii, jj, kk), sgn in ((lst[ref], 1.0), ((i, j, k), -1.0)):
                if ii == jj:
                    rows.append(entry(kk, ii, ii)); cols.append(m); vals.append(sgn)
                else:
                    rows += [entry(kk, ii, jj), entry(kk, jj, ii)]; cols += [m, m]; vals += [0.5 * sgn, 0.5 * sgn]
            m += 1
    At = sp.csc_matrix((vals, (rows, cols)), shape=(int(off[-1]), m))
    b = np.zeros(m); b[0] = 1.0
    c = np.zeros(int(off[-1]))
    mons = quartic_sparse_monomials(cliques)
    coe = np.asarray(coe, dtype=np.float64).ravel()
    if coe.size != len(mons):
        raise ValueError("qsmom_sparse: %d coefficients for %d monomials" % (coe.size, len(mons)))
    for mon, v in zip(mons, coe):
        lo = spots(mon)
        c[lo] += v / len(lo)
    return At, b, c, {"s": mb, "nob": 0}


def matrix_completion(p, q, k, m=None, seed=3):
    """Nuclear-norm matrix completion as an SDP for the generic ``ManiSDP`` (reference example/example_matrixcompletion.m:8-41):
    ``M = randn(p,k) randn(k,q)``, ``m`` sampled positions (default ``400 (p+q)`` draws with replacement, duplicates removed,
    as in the example), ``min <I, X>  s.t.  X[j, p+l] + X[p+l, j] = 2 M[j, l]`` for the sampled ``(j, l)``, ``X`` of order
    ``n = p + q``.  NumPy's generator replaces MATLAB's, so instances differ from the reference's; the structure does not.
    Returns ``At (n^2 x m, CSC), b, c, K, M, (rows, cols)``."""
    rng = np.random.default_rng(seed)
    n = p + q
    M = rng.standard_normal((p, k)) @ rng.standard_normal((k, q))
    m = 400 * n if m is None else int(m)
    omega = np.unique(rng.integers(0, p * q, size=m))              # :15-17 (row-major position j*q + l here)
    j, l = omega // q, omega % q
    m = omega.size
    b = 2.0 * M[j, l]                                              # :33
    col = np.repeat(np.arange(m), 2)
    row = np.empty(2 * m, dtype=np.int64)
    row[0::2] = j * n + (l + p)                                    # :34  vec index of (l+p, j), column-major
    row[1::2] = (l + p) * n + j                                    #      and of (j, l+p)
    At = sp.csc_matrix((np.ones(2 * m), (row, col)), shape=(n * n, m))
    c = np.eye(n).ravel()
    return At, b, c, {"s": n, "l": 0}, M, (j, l)


def dense_unitdiag_cost(n, seed=0):
    """Random dense symmetric cost ``C = (G + G')/(2 sqrt(n))``."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    return (G + G.T) / (2.0 * np.sqrt(n))


class SyntheticDenseC:
    """The synthetic dense symmetric cost of BASELINE config 5 (n = 100 000, p = 64: the matrix is 80 GB and never exists
    as a host array).  Entry (i, j) is a counter-based hash of (min(i,j), max(i,j), seed) mapped to U(-1, 1)/sqrt(n) --
    the generator every rank runs on the device for ITS rows (``msdp_create_onlyunitdiag_dense_synthetic``;
    ``msdp_synthetic_dense_entry`` is the same function on the host).  ``ManiSDP_onlyunitdiag`` accepts an instance in
    place of ``C``; ``rows`` / ``toarray`` restate the generator in NumPy for parity tests at small n."""

    def __init__(self, n, seed=0):
        self.n, self.seed = int(n), int(seed)
        self.shape = (self.n, self.n)

    def rows(self, rows):
        n = self.n
        rows = np.asarray(rows, dtype=np.uint64)[:, None]
        cols = np.arange(n, dtype=np.uint64)[None, :]
        a = np.minimum(rows, cols); bb = np.maximum(rows, cols)
        g = np.uint64(0x9E3779B97F4A7C15)
        with np.errstate(over="ignore"):
            x = a * np.uint64(n) + bb + np.uint64(self.seed) * g
            x = x + g
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
        u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return (2.0 * u - 1.0) / np.sqrt(float(n))

    def toarray(self):
        if self.n > 20000:
            raise ValueError("SyntheticDenseC.toarray: n = %d is a %0.f-GB matrix" % (self.n, 8e-9 * self.n * self.n))
        return self.rows(np.arange(self.n))
