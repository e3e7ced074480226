// msdp_device.h -- device-side helpers shared by the kernel translation units.
//
// Reductions use DPP (data-parallel primitives on the VALU) instead of __shfl:
// on gfx950 a __shfl_xor of a double is two ds_bpermute round trips through the
// LDS pipe per step (measured: three workgroup sums added 4 us to a 6.8 us
// streaming kernel); the DPP forms are plain VALU moves.
#pragma once
#include "msdp_common.h"

// Windowed row traversal for gather kernels on vectors larger than the L2s (sweep != 0).  With msdp_chunk_rows every workgroup
// streams through its own chunk: the 64 workgroups of an XCD walk 64 windows that lie one chunk apart, and a row a neighbour
// row's gather brought into the L2 is gone long before its owner reaches it -- the two-launch tCG head, which gathers three
// vectors, fetched 2.66 GB for 1.28 GB of vectors at n = 10^6 (profiles/r3_pmc_chunked_n1e6_p32.json).  Here the workgroups of
// an XCD that are resident TOGETHER (32: one 1024-thread workgroup per CU; a grid of 512 runs as two phases, each on its own
// half of the XCD's rows) take consecutive blocks of BR rows and advance by 32 blocks: one window of 32*BR rows plus the
// graph's halo per XCD, which its 4-MB L2 holds.  Loop shape: for (row0 = lo + wave*RPW; row0 < hi; row0 += stride).
__device__ __forceinline__ void msdp_sweep_rows(int n_loc, int G, int BR, int& lo, int& hi, int& stride) {
    const int X = blockIdx.x & 7, s = blockIdx.x >> 3, S = G >> 3;
    const int W = S < 32 ? S : 32, P = (S + W - 1) / W;
    const int phase = s / W, sp = s - phase * W;
    const int xlo = (int)((int64_t)n_loc * X / 8), xhi = (int)((int64_t)n_loc * (X + 1) / 8);
    // phases split the XCD's rows at multiples of BR so that no block straddles two phases
    const int nb = (xhi - xlo + BR - 1) / BR;
    const int b0 = (int)((int64_t)nb * phase / P), b1 = (int)((int64_t)nb * (phase + 1) / P);
    lo = xlo + (b0 + sp) * BR;
    hi = xlo + b1 * BR < xhi ? xlo + b1 * BR : xhi;
    stride = W * BR;
}

// Workgroup b -> row chunk.  Workgroups b and b+8 share an XCD (round-robin
// dispatch, observed; a pure speed heuristic), so XCD x gets the contiguous
// chunk range [x*G/8, (x+1)*G/8): its L2 then serves one contiguous 1/8 of the
// rows of every vector.  G is a multiple of 8.
__device__ __forceinline__ void msdp_chunk_rows(int n_loc, int G, int& lo, int& hi, int plain = 0, int bx = -1) {
    const int b = bx < 0 ? (int)blockIdx.x : bx;            // bx: the workgroup's index inside ITS rank's share of a combined launch
    const int c = plain ? b : (b & 7) * (G >> 3) + (b >> 3);
    // balanced split with one 32-bit division (a 64-bit n_loc*c/G costs ~350 instructions of preamble in
    // every launch): the first n_loc % G chunks get one extra row
    const unsigned q = (unsigned)n_loc / (unsigned)G, r = (unsigned)n_loc - q * (unsigned)G;
    lo = (int)(c * q + ((unsigned)c < r ? (unsigned)c : r));
    hi = lo + (int)q + ((unsigned)c < r ? 1 : 0);
}

// v moved by a DPP control (both halves of the double).
template <int CTRL>
__device__ __forceinline__ double msdp_dpp(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
#define MSDP_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define MSDP_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define MSDP_DPP_HALF_MIRROR 0x141  // lane i <-> 7-i inside each 8 lanes
#define MSDP_DPP_MIRROR 0x140       // lane i <-> 15-i inside each 16-lane row

__device__ __forceinline__ double msdp_readlane(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

template <int W>
__device__ __forceinline__ double msdp_rowpair_sum(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)((unsigned long long)b >> 32);
    unsigned l0, l1, h0, h1;
    if (W == 16) {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    }
    const double a = __longlong_as_double((long long)(((unsigned long long)h0 << 32) | l0));
    const double c = __longlong_as_double((long long)(((unsigned long long)h1 << 32) | l1));
    return a + c;
}

// Sum over the LPR consecutive lanes that serve one row; valid in every lane of the
// group.  All lanes of a group must be active together (they are: a group = a row).
template <int LPR>
__device__ __forceinline__ double msdp_group_sum(double v) {
    if (LPR >= 2) v += msdp_dpp<MSDP_DPP_XOR1>(v);
    if (LPR >= 4) v += msdp_dpp<MSDP_DPP_XOR2>(v);
    if (LPR >= 8) v += msdp_dpp<MSDP_DPP_HALF_MIRROR>(v);
    if (LPR >= 16) v += msdp_dpp<MSDP_DPP_MIRROR>(v);
    // gfx950: v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows in the VALU (a __shfl_xor across
    // rows is a ds_bpermute round trip through the LDS crossbar).  swap(v, v) returns {even-row value everywhere in
    // the pair, odd-row value everywhere in the pair}; both rows add them in the same order.
    if (LPR >= 32) v = msdp_rowpair_sum<16>(v);
    if (LPR >= 64) v = msdp_rowpair_sum<32>(v);
    return v;
}

// Full-wave sum (all 64 lanes active), same value in every lane, fixed order.
__device__ __forceinline__ double msdp_wave_sum(double v) {
    v += msdp_dpp<MSDP_DPP_XOR1>(v);
    v += msdp_dpp<MSDP_DPP_XOR2>(v);
    v += msdp_dpp<MSDP_DPP_HALF_MIRROR>(v);
    v += msdp_dpp<MSDP_DPP_MIRROR>(v);
    const double r0 = msdp_readlane(v, 0), r1 = msdp_readlane(v, 16);
    const double r2 = msdp_readlane(v, 32), r3 = msdp_readlane(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// Deterministic workgroup sum; result valid in every thread (one barrier pair).
__device__ __forceinline__ double msdp_block_sum(double v, double* sh /* >= MSDP_WAVES doubles */) {
    v = msdp_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

// Up to three per-workgroup partial sums with ONE barrier.  sh (>= 3*MSDP_WAVES
// doubles) must not be in use by another phase of the kernel.
__device__ __forceinline__ void msdp_put_partials3(double* P, int w0, double a, int w1, double b, int w2, double c,
                                                   double* sh) {
    a = msdp_wave_sum(a);
    if (w1 >= 0) b = msdp_wave_sum(b);
    if (w2 >= 0) c = msdp_wave_sum(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[MSDP_WAVES + w] = b; sh[2 * MSDP_WAVES + w] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int which = threadIdx.x == 0 ? w0 : (threadIdx.x == 1 ? w1 : w2);
        if (which >= 0) {
            double s = 0.0;
            const int nw = blockDim.x >> 6;
            for (int i = 0; i < nw; ++i) s += sh[threadIdx.x * MSDP_WAVES + i];
            P[which * MSDP_MAX_GRID + blockIdx.x] = s;
        }
    }
}
__device__ __forceinline__ void msdp_put_partial(double* P, int which, double v, double* sh) {
    msdp_put_partials3(P, which, v, -1, 0.0, -1, 0.0, sh);
}

// Every WAVE re-reduces the <= 512 partials of the previous launch in the same fixed
// order (no LDS, no barrier): all waves, workgroups and ranks obtain bit-identical
// sums and therefore take identical decisions without atomics or fences.
__device__ __forceinline__ double msdp_sum_partials(const double* P, int which, int G) {
    const double* a = P + which * MSDP_MAX_GRID;
    const int lane = threadIdx.x & 63;
    double v = 0.0;
    for (int i = lane; i < G; i += 64) v += a[i];
    return msdp_wave_sum(v);
}

// Block-level variant: wave 0 re-reduces, everyone reads the result from LDS (one barrier);
// 16x fewer same-address requests to the L2 channels that hold the partials.
__device__ __forceinline__ double msdp_sum_partials_block(const double* P, int which, int G, double* shb) {
    if (threadIdx.x < 64) {
        const double s = msdp_sum_partials(P, which, G);
        if (threadIdx.x == 0) shb[0] = s;
    }
    __syncthreads();
    return shb[0];
}
__device__ __forceinline__ void msdp_sum_partials3_block(const double* P, int w0, int w1, int w2, int G, double* shb,
                                                         double& s0, double& s1, double& s2) {
    if (threadIdx.x < 64) {
        const double a = msdp_sum_partials(P, w0, G), b = msdp_sum_partials(P, w1, G), c = msdp_sum_partials(P, w2, G);
        if (threadIdx.x == 0) { shb[0] = a; shb[1] = b; shb[2] = c; }
    }
    __syncthreads();
    s0 = shb[0]; s1 = shb[1]; s2 = shb[2];
}

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }
// streaming accesses (nt: the line is not kept by the L2 beyond its use) for the operands a gather kernel touches exactly once,
// so that the rows other workgroups gather stay resident
typedef double msdp_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_nt(const double* p) {
    const msdp_d2v v = __builtin_nontemporal_load(reinterpret_cast<const msdp_d2v*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st2_nt(double* p, double2 v) {
    const msdp_d2v w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<msdp_d2v*>(p));
}

// Publish tCG progress to the host-mapped status word (one relaxed system-scope store by
// the lead thread): the host polls it to decide whether to enqueue another chunk.
__device__ __forceinline__ void msdp_publish(const Dev& d, int k, int j, int active) {
    if (!d.status) return;
    // bit 30 of the j field carries ctl->done so that the host can stop enqueuing TR iterations without a sync
    const unsigned jj = ((unsigned)j & 0x3fffffffu) | (d.ctl->done ? 0x40000000u : 0u);
    const unsigned long long v = ((unsigned long long)(unsigned)(k + 1) << 32) |
                                 ((unsigned long long)jj << 1) | (unsigned long long)(active & 1);
    __hip_atomic_store(d.status, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Frame fields are read one by one into scalars and written field by field by the lead
// thread: a by-value Frame copy is lowered through per-thread LDS/scratch by hipcc and
// cost 10 us per launch (measured: 19.3 -> 9.4 us for k_tcg_upd1).
__device__ __forceinline__ void frame_store(Frame* o, double z_r, double d_Pd, double e_Pd, double e_Pe,
                                            double model_value, double norm_r0, double alpha, double beta,
                                            int active, int j, int stop, int eta_idx, int md_idx = 0, int fresh = 0) {
    o->z_r = z_r; o->d_Pd = d_Pd; o->e_Pd = e_Pd; o->e_Pe = e_Pe; o->model_value = model_value;
    o->norm_r0 = norm_r0; o->alpha = alpha; o->beta = beta;
    o->active = active; o->j = j; o->stop = stop; o->eta_idx = eta_idx; o->md_idx = md_idx; o->fresh = fresh;
}


// Sum of the SK split-K slabs at offset o (two doubles), in slab order.  Four loads are requested together: with a
// plain loop hipcc keeps one load in flight per lane (SK is a run-time value), and the epilogues are latency-bound.
__device__ __forceinline__ double2 msdp_sum_slabs(const double* __restrict__ slab, int64_t slab_stride, int SK, int64_t o) {
    double2 acc = make_double2(0.0, 0.0);
    int s = 0;
    // eight, then four, then the last <= 3 loads in flight together; the additions stay in slab order (same bits as a plain loop).
    // Round 3: the affine Hess-vec of BQP d = 60 sums 26 slabs -- seven dependent round trips with batches of four.
    for (; s + 8 <= SK; s += 8) {
        double2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld2(slab + (int64_t)(s + q) * slab_stride + o);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc.x += v[q].x; acc.y += v[q].y; }
    }
    if (s + 4 <= SK) {
        double2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld2(slab + (int64_t)(s + q) * slab_stride + o);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc.x += v[q].x; acc.y += v[q].y; }
        s += 4;
    }
    if (s < SK) {
        double2 v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = (s + q < SK) ? ld2(slab + (int64_t)(s + q) * slab_stride + o) : make_double2(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < 3; ++q) if (s + q < SK) { acc.x += v[q].x; acc.y += v[q].y; }
    }
    return acc;
}

// ------------------------------------------------------------------ sparse gather
// acc[ch] += sum_k C[row,k] * X[k, cols of this lane]; the LPR lanes of a row each
// fetch one (col,val) pair of the CSR row (coalesced) and broadcast it by shuffle.
#define MSDP_ELL_MAXW 8
template <int LPR, int NCH, bool ELL>
__device__ __forceinline__ void spmm_row(const Dev& d, int row, int sub, const double* __restrict__ X,
                                         double2 (&acc)[NCH]) {
    if (ELL) {
        // ELL slices ([w][row], padded with (row, 0.0)): no rowptr in the dependency chain, all (col,val)
        // loads of a row are independent and coalesced across the rows of a wave, and all W neighbour-row
        // gathers are in flight together: two dependent memory round trips instead of three.
        int c[MSDP_ELL_MAXW];
        double v[MSDP_ELL_MAXW];
#pragma unroll
        for (int w = 0; w < MSDP_ELL_MAXW; ++w) {
            const bool ok = w < d.ellW;
            c[w] = ok ? d.ellc[(int64_t)w * d.ell_stride + row] : row;
            v[w] = ok ? d.ellv[(int64_t)w * d.ell_stride + row] : 0.0;
        }
#pragma unroll
        for (int w = 0; w < MSDP_ELL_MAXW; ++w) {
            if (w < d.ellW) {
                const double* src = X + (int64_t)c[w] * d.ld + 2 * sub;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (2 * sub + ch * 2 * LPR < d.ld) {
                        const double2 x = ld2(src + ch * 2 * LPR);
                        acc[ch].x = fma(v[w], x.x, acc[ch].x);
                        acc[ch].y = fma(v[w], x.y, acc[ch].y);
                    }
                }
            }
        }
        return;
    }
    // CSR: all LPR lanes of a row read the same (col,val) pair: one L1 line per row serves the
    // whole group, the loads are independent of each other (no shuffle in the chain) and
    // the compiler can keep 4 neighbour rows in flight per lane.
    const int start = d.rowptr[row], end = d.rowptr[row + 1];
    const int* __restrict__ ci = d.colind;
    const double* __restrict__ cv = d.cval;
#pragma unroll 4
    for (int k = start; k < end; ++k) {
        const int c = ci[k];
        const double v = cv[k];
        const double* src = X + (int64_t)c * d.ld + 2 * sub;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (2 * sub + ch * 2 * LPR < d.ld) {
                const double2 x = ld2(src + ch * 2 * LPR);
                acc[ch].x = fma(v, x.x, acc[ch].x);
                acc[ch].y = fma(v, x.y, acc[ch].y);
            }
        }
    }
}
