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
