"""Row sharding of the factor over the GPUs of one node.

The n points (rows of the n x p factor = MATLAB columns of the p x n Y) are split into
``nranks`` contiguous blocks of equal capacity ``cap = ceil(n / nranks)`` so that the thin
direction can be exchanged with ONE uniform all-gather (every rank contributes ``cap * p``
doubles; the last block is zero-padded).  Rank r owns rows ``[r*cap, min(n, (r+1)*cap))`` of
Y, U, H, eG and the same rows of C.  This file is the host-side statement of that layout; the
library applies the identical partition in ``msdp_comm_init`` (csrc/msdp_api.hip).
"""
from __future__ import annotations

import numpy as np


def row_capacity(n, nranks):
    return (n + nranks - 1) // nranks


def row_range(n, nranks, rank):
    cap = row_capacity(n, nranks)
    r0 = min(n, rank * cap)
    return r0, min(n, r0 + cap)


def shard_rows_csr(C, n, nranks, rank):
    """Rows [r0, r1) of a scipy CSR matrix, global column indices kept."""
    r0, r1 = row_range(n, nranks, rank)
    return C.tocsr()[r0:r1, :]


def pad_slab(local_rows, n, nranks):
    """Zero-pad a rank's (n_loc, p) block to the uniform all-gather slab (cap, p)."""
    cap = row_capacity(n, nranks)
    out = np.zeros((cap, local_rows.shape[1]), dtype=local_rows.dtype)
    out[: local_rows.shape[0]] = local_rows
    return out


def unpad_gathered(slabs, n):
    """Concatenated slabs (nranks*cap, p) -> the n real rows (pad rows sit at the tail of each slab)."""
    nranks = len(slabs)
    cap = slabs[0].shape[0]
    rows = []
    for r, s in enumerate(slabs):
        r0 = min(n, r * cap)
        r1 = min(n, r0 + cap)
        rows.append(s[: r1 - r0])
    return np.vstack(rows)


def halo_lists(C, nranks, rank):
    """Send / receive lists of the halo exchange for sparse C (option ``halo_exchange``; the host-side statement of
    ``halo_setup`` in csrc/msdp_api.hip).  Every rank holds the whole sparsity structure, so both ends of a pair compute
    the same lists: ``need[q]`` = the sorted global rows outside q's range that q's rows of C reference.

    Returns ``(send, recv)``: ``send[q]`` = local row indices this rank packs for peer q (in the order q unpacks them),
    ``recv[q]`` = global row indices this rank receives from peer q; both empty for ``q == rank``."""
    C = C.tocsr()
    n = C.shape[0]
    cap = row_capacity(n, nranks)
    m0, m1 = row_range(n, nranks, rank)
    send = [np.zeros(0, dtype=np.int64) for _ in range(nranks)]
    recv = [np.zeros(0, dtype=np.int64) for _ in range(nranks)]
    for q in range(nranks):
        q0, q1 = row_range(n, nranks, q)
        cols = np.unique(C.indices[C.indptr[q0]:C.indptr[q1]]) if q1 > q0 else np.zeros(0, dtype=np.int64)
        need = cols[(cols < q0) | (cols >= q1)]
        if q == rank:
            for o in range(nranks):
                recv[o] = need[need // cap == o].astype(np.int64)
        else:
            send[q] = (need[(need >= m0) & (need < m1)] - m0).astype(np.int64)
    return send, recv


def tcg_one_allreduce(Cl, Yl, gl, eGl, Delta, maxinner, exchange, allreduce, kappa=0.1, theta=1.0, mininner=1, refresh=32):
    """Host-side statement of the row-sharded tCG trip of ``csrc/msdp_trip1.hip`` (tCG.m:95-292 on the oblique manifold of
    ManiSDP_onlyunitdiag.m, Hess-vec :127-130): ONE exchange and ONE all-reduce per trip.

    ``Cl``: this rank's rows of C (global columns); ``Yl, gl, eGl``: its rows of the point, of the Riemannian gradient and of
    eG.  ``exchange(rows, sums)`` returns (all n rows, the list of every rank's ``sums`` in rank order) -- one collective;
    ``allreduce(x)`` sums a scalar over the ranks -- the other.  The residual is kept projected, the Hess-vec of the new
    direction follows by linearity, H mdelta' = H r' + beta * H mdelta (the Hessian is a linear map on the tangent space), and
    every ``refresh``-th trip exchanges the direction itself once more.  Returns (eta rows, Heta rows, inner iterations, stop code) like the oracle's tCG."""
    import math

    def tangent(v):                                         # obliquefactory: v - Y .* rowdot(Y, v)
        return v - Yl * np.sum(Yl * v, axis=1, keepdims=True)

    def hess_rows(cx, x):                                   # ManiSDP_onlyunitdiag.m:128-130 on the own rows, cx = rows of C*x
        return cx - Yl * np.sum(Yl * cx, axis=1, keepdims=True) - x * eGl

    eta = np.zeros_like(gl)
    r = gl.copy()
    md = gl.copy()
    r_r = allreduce(float(np.sum(gl * gl)))                 # (the library holds |grad|^2 from the gradient evaluation)
    norm_r0 = math.sqrt(r_r)
    z_r, d_Pd, e_Pd, e_Pe, model_value = r_r, r_r, 0.0, 0.0, 0.0
    full, _ = exchange(md, [0.0, 0.0, 0.0])                 # first trip: direct product with the gradient rows
    Hmd = hess_rows(Cl @ full, md)
    d_Hd = allreduce(float(np.sum(md * Hmd)))               # tCG.m:166
    stop, j = 5, 0
    for j in range(1, maxinner + 1):
        alpha = z_r / d_Hd if d_Hd != 0.0 else math.copysign(math.inf, z_r)
        e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd          # :173
        if d_Hd <= 0 or e_Pe_new >= Delta ** 2:                               # :183
            tau = (-e_Pd + math.sqrt(e_Pd * e_Pd + d_Pd * (Delta ** 2 - e_Pe))) / d_Pd
            return eta - tau * md, (r - tau * Hmd) - gl, j, (1 if d_Hd <= 0 else 2)      # Heta = r - grad (:198,220,238)
        new_eta = eta - alpha * md                                            # :215
        new_r = tangent(r - alpha * Hmd)                                      # :238, kept projected (rounding only)
        new_Heta = new_r - gl
        sums = [float(np.sum(new_eta * gl)), float(np.sum(new_eta * new_Heta)), float(np.sum(new_r * new_r))]
        full, all_sums = exchange(new_r, sums)                                # THE exchange: rows + every rank's sums
        s1 = s2 = r_r = 0.0
        for q in all_sums:                                                    # rank order: the same bits on every rank
            s1 += q[0]; s2 += q[1]; r_r += q[2]
        new_model = s1 + 0.5 * s2                                             # :227
        if new_model >= model_value:                                          # :228
            return eta, r - gl, j, 6
        eta, r, model_value, e_Pe = new_eta, new_r, new_model, e_Pe_new
        if j >= mininner and math.sqrt(r_r) <= norm_r0 * min(norm_r0 ** theta, kappa):   # :249
            return eta, r - gl, j, (3 if kappa < norm_r0 ** theta else 4)
        if j >= maxinner:
            break
        beta = r_r / z_r                                                      # :272
        md = tangent(r + beta * md)                                           # :273,283
        Hmd = hess_rows(Cl @ full, r) + beta * Hmd                            # linearity: H mdelta' = H r' + beta * H mdelta
        if refresh > 0 and j % refresh == 0:                                  # direct product every refresh-th trip
            full, _ = exchange(md, [0.0, 0.0, 0.0])
            Hmd = hess_rows(Cl @ full, md)
        e_Pd = beta * (e_Pd + alpha * d_Pd)                                   # :286
        d_Pd = r_r + beta * beta * d_Pd                                       # :287
        z_r = r_r
        d_Hd = allreduce(float(np.sum(md * Hmd)))                             # THE all-reduce (tCG.m:166)
    return eta, r - gl, j, stop
