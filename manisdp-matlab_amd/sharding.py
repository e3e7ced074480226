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
