// msdp_dense.hip -- fp64 MFMA path for the dense tall-skinny contraction S*U.
//
//   onlyunitdiag, dense C :  eH = C*U                       (ManiSDP_onlyunitdiag.m:128, dense C)
//   unitdiag              :  eH = 2*eS*U + 4*sigma*AyU*Y    (ManiSDP_unitdiag.m:169)
//   unittrace             :  H  = 2*eS*U + 4*sigma*AyU*Y    (ManiSDP_unittrace.m:174)
//
// Every matrix is stored row-major with a padded leading dimension nS = roundup(n,16)
// (zero pad columns), so the k-loop has no tail and every A-fragment load is a 16-byte
// aligned global_load_dwordx4.  The matrix is symmetric, so "row-major rows" are also the
// columns the reference's U*C touches.
//
// Kernel 1 (k_dense_partial3): one wave owns 16 rows x all p columns as NT accumulator
// tiles of v_mfma_f64_16x16x4_f64; the matrix is streamed from HBM exactly once straight
// into the MFMA A layout (lane (g = lane>>4, i = lane&15) holds S[row0+i][k0+4g+t], t=0..3,
// two dwordx4 per 16 k -- each matrix row contributes a full 128-byte line per step); the
// thin panel tile U[k0:k0+KT, :] is staged through LDS once per workgroup (4 waves up to
// p = 32, 8 beyond) and read as the B operand (ds_read_b64: the 16 lanes of a k-group read
// 128 contiguous bytes; SQ_LDS_BANK_CONFLICT = 0).  The K range is split over blockIdx.y so
// that the launch fills the chip exactly once or a whole number of times (dense_plan); each
// slice writes a partial slab with plain stores (deterministic, no fp64 atomics).
// Kernel 2 (row-tiled epilogue): sums the slabs and applies the fused projection.
#include "msdp_device.h"
#include "msdp_affine_dev.h"
#include <math.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

typedef double double4_t __attribute__((ext_vector_type(4)));


struct DenseOp {
    const double* M[2];          // n_loc x nS row-major matrices (rows = local rows)
    const double* Mpk;           // M[0] again in MFMA-fragment order (k_pack_fragments), or null
    const double* X[2];          // n x ld panels (all rows)
    double scale[2];
    int nmat;
    int n;                       // true matrix order (panel rows)
    int nS;                      // padded leading dimension / k extent per matrix
    int n_loc, ld, ldl;          // rows, panel stride, LDS row stride
    int colofs, ncols;           // column block [colofs, colofs+ncols) of the panel handled by this launch (ncols <= 128)
    int SK;                      // k slices
    int kslice;                  // k extent per slice (multiple of 64) over the concatenated K = nmat*nS
    double* slab;                // SK x n_loc_cap x ld
    int64_t slab_stride;         // doubles between slabs
    const int* blk_lo; const int* blk_hi;   // block-diagonal operands (multiblock kind, Dev::blk_lo): column range per row, or null
};

// What shaped the kernel (all measured on MI355X, tools/dense_sweep.sh; a pure-MFMA loop reaches 72 TFLOP/s,
// tools/microbench_mfma64.hip):
//   * registers first: an earlier version with two fragment sets AND two staging sets and per-thread staging tables
//     needed 200-256 VGPRs = 1-2 waves per SIMD, so every barrier and LDS round trip idled the matrix pipe (26-35 TF).
//     Now: two statically named fragment sets (tile in work / next tile, 8 VGPRs per 16 k each; static names let
//     hipcc emit counted vmcnt(N) waits), ONE staging set, no tables (HP = pow2 >= ncols/2 lanes cover a panel row,
//     WAVES*64/HP rows per pass), the scale of a k-step as a wave-uniform scalar where it is used, KT = 32 beyond
//     p = 16: 112-128 VGPRs = 4 waves per SIMD;
//   * hipcc's occupancy-driven scheduler sinks prefetch loads next to their use whenever it can: the loop body has
//     no exits and no conditional stores, and beyond p = 32 a sched_barrier closes the prefetch block of each tile
//     (load_stg below describes the two forms);
//   * a single fragment ring refilled step by step was tried: the refills were sunk below the last MFMA of the tile,
//     and pinning them cost a second register set anyway.
template <int NT> struct Dense3Cfg {
    static constexpr int KT = NT <= 1 ? 64 : 32;
    static constexpr int HP = NT == 1 ? 8 : (NT == 2 ? 16 : (NT <= 4 ? 32 : 64));
    static constexpr int WAVES = NT <= 2 ? 4 : 8;              // 16 matrix rows each (8 waves beyond p = 32: the panel tile
                                                               // is shared by twice the rows: +6..14 % there, -3 % at p <= 32)
    static constexpr int RPP = WAVES * 64 / HP;                // panel rows staged per pass
    static constexpr int NPASS = KT / RPP;
    static constexpr int SS = KT / 16;
    static constexpr bool PIN = NT >= 3;                       // keep the prefetch block of a tile in place (see below)
};

// PK: the (single) matrix is read from its fragment-ordered copy: tile (rb, kb) of 16 rows x 16 k is 256 contiguous
// doubles, first the (t = 0,1) pairs of the 64 lanes, then the (t = 2,3) pairs, so each of the two dwordx4 loads of a
// k-step reads one contiguous kilobyte per wave instead of sixteen 128-byte lines 8*nS bytes apart.  Same values into
// the same MFMA sequence: results are bit-identical to the row-major form.
template <int NT, bool PK>
__device__ __forceinline__ void dense_partial3_body(const DenseOp& op, const int by) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 x KT x ldl + 128 (dummy slots)
    typedef Dense3Cfg<NT> Cfg;
    constexpr int KT = Cfg::KT, HP = Cfg::HP, RPP = Cfg::RPP, NPASS = Cfg::NPASS, SS = Cfg::SS;
    constexpr int NTHR = Cfg::WAVES * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int row0 = (blockIdx.x * Cfg::WAVES + wave) * 16;
    const int arow = min(row0 + i, op.n_loc - 1);
    const int nS = op.nS;
    const int Ktot = op.nmat * nS;
    int kbeg = by * op.kslice;
    int kend = min(kbeg + op.kslice, Ktot);
    if (op.blk_lo) {
        // block-diagonal operands: outside the column range of the blocks this workgroup's rows belong to the matrices hold
        // exact zeros -- trim the slice to the hull of its overlaps (whole tile pairs, so the tile sequence inside is the
        // one of the untrimmed loop), or to nothing: the slab rows are then written as zeros
        const int rfirst = blockIdx.x * Cfg::WAVES * 16, rlast = min(rfirst + Cfg::WAVES * 16, op.n_loc) - 1;
        const int clo = op.blk_lo[rfirst], chi = op.blk_hi[rlast];
        int lo = 0x7fffffff, hi = -1;
        for (int m = 0; m < op.nmat; ++m) {
            const int a = max(m * nS + clo, kbeg), b = min(m * nS + chi, kend);
            if (a < b) { lo = min(lo, a); hi = max(hi, b); }
        }
        if (hi < 0) kend = kbeg;
        else {
            const int k0 = kbeg + ((lo - kbeg) / (2 * KT)) * (2 * KT);
            const int k1 = kbeg + ((hi - kbeg + 2 * KT - 1) / (2 * KT)) * (2 * KT);
            kend = min(kend, k1);
            kbeg = k0;
        }
    }
    const int ld = op.ld, ldl = op.ldl;
    for (int e = threadIdx.x; e < 2 * KT * ldl; e += NTHR) lds[e] = 0.0;   // pad columns stay zero
    const int c2 = threadIdx.x & (HP - 1), sr0 = threadIdx.x / HP;
    const bool colok = 2 * c2 < op.ncols;
    const int sgo = op.colofs + (colok ? 2 * c2 : 0);
    const int slo = sr0 * ldl + 2 * c2;
    double4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double* a0 = PK ? op.Mpk + (int64_t)(arow >> 4) * nS * 16 + 2 * lane      // tile (rb, kb) starts at (rb*nS/16 + kb)*256
                          : op.M[0] + (int64_t)arow * nS + 4 * g;
    const double* a1 = op.M[1] + (int64_t)arow * nS + 4 * g - nS;      // indexed with the concatenated k
    double2 A0[SS][2], A1[SS][2];                              // fragments of the tile in work / of the next tile
    double2 stg[NPASS];

    auto load_a = [&](int k0, double2 (&A)[SS][2]) {           // branch-free: out-of-range steps read the slice start
#pragma unroll
        for (int s = 0; s < SS; ++s) {
            const int ks = k0 + 16 * s;
            const int kc = ks < kend ? ks : kbeg;
            if constexpr (PK) {
                const double* ap = a0 + (int64_t)kc * 16;
                A[s][0] = ld2(ap); A[s][1] = ld2(ap + 128);
            } else {
                const double* ap = (kc >= nS ? a1 : a0) + kc;
                A[s][0] = ld2(ap); A[s][1] = ld2(ap + 2);
            }
        }
    };
    // Rows outside the slice or in the pad range [n, nS) contribute nothing: the matching matrix entries are exact
    // zeros (pad columns) or are multiplied by sc = 0 (beyond the slice).  PIN form (p > 32): such rows read panel
    // row 0 (any finite value will do), the staged value is used as loaded and a sched_barrier keeps the whole
    // prefetch block at the top of the tile.  Up to p = 32 the kernel is bound by the bytes in flight rather than
    // by the matrix pipe, and the form below measured best of four (n = 20000, p = 32: 625 us; PIN form 658; neither
    // select nor pin 656; 32-bit offsets + select 690): hipcc keeps the fragment loads and the first staging load
    // at the top of the tile and fits 90 VGPRs = 5 waves per SIMD (PIN form: 127 = 4).
    auto load_stg = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int kk = k0 + sr0 + q * RPP;
            if constexpr (Cfg::PIN) {
                const bool inr = kk < kend;
                const int m = (inr && kk >= nS) ? 1 : 0;
                int kl = kk - m * nS;
                kl = (inr && kl < op.n) ? kl : 0;
                stg[q] = ld2((m ? op.X[1] : op.X[0]) + (unsigned)(kl * ld + sgo));
            } else {
                const int kc = kk < kend ? kk : kbeg;
                const int m = kc >= nS ? 1 : 0;
                const int kl = kc - m * nS;
                const bool ok = kk < kend && kl < op.n;
                const double2 v = ld2((m ? op.X[1] : op.X[0]) + (ok ? (int64_t)kl * ld + sgo : 0));
                stg[q] = ok ? v : make_double2(0.0, 0.0);
            }
        }
    };
    // branch-free (a conditional store lets hipcc sink the staging loads into the branch, next to their use):
    // lanes beyond the last column pair write a per-lane dummy slot behind the two tiles
    auto store_stg = [&](double* buf) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
            *reinterpret_cast<double2*>(colok ? &buf[slo + q * RPP * ldl] : &lds[2 * KT * ldl + 2 * lane]) = stg[q];
    };
    auto compute_tile = [&](const double* bt, int k0, const double2 (&A)[SS][2]) {
#pragma unroll
        for (int s = 0; s < SS; ++s) {
            const int ks = k0 + 16 * s;
            const double sc = ks < kend ? (ks >= nS ? op.scale[1] : op.scale[0]) : 0.0;
            const double av[4] = {A[s][0].x * sc, A[s][0].y * sc, A[s][1].x * sc, A[s][1].y * sc};
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const double* brow = &bt[(16 * s + 4 * g + t4) * ldl + i];
                double bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = brow[16 * t];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t4], bv[t], acc[t], 0, 0, 0);
            }
        }
    };

    double* buf0 = lds;
    double* buf1 = lds + KT * ldl;
    __syncthreads();                                           // zero fill done
    load_a(kbeg, A0);
    load_stg(kbeg);
    store_stg(buf0);
    // No exits inside the body: a break between the loads and the code that uses them lets hipcc sink the loads
    // past the branch, next to their use.  An odd tile count costs one tile of zero work (sc = 0, staged zeros).
    for (int k0 = kbeg; k0 < kend; k0 += 2 * KT) {
        __syncthreads();
        load_stg(k0 + KT);                                      // older than the fragment loads: its wait leaves them in flight
        load_a(k0 + KT, A1);
        if constexpr (Cfg::PIN) __builtin_amdgcn_sched_barrier(0);   // hipcc's occupancy-driven scheduler otherwise sinks the prefetch next to its use
        compute_tile(buf0, k0, A0);
        store_stg(buf1);
        __syncthreads();
        load_stg(k0 + 2 * KT);
        load_a(k0 + 2 * KT, A0);
        if constexpr (Cfg::PIN) __builtin_amdgcn_sched_barrier(0);
        compute_tile(buf1, k0 + KT, A1);
        store_stg(buf0);
    }
    double* out = op.slab + (int64_t)by * op.slab_stride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + g + 4 * r;
            if (row < op.n_loc && col < op.ncols) out[(int64_t)row * ld + op.colofs + col] = acc[t][r];
        }
    }
}

template <int NT, bool PK>
__global__ __launch_bounds__(Dense3Cfg<NT>::WAVES * 64, (NT <= 4 ? 3 : 1)) void k_dense_partial3(DenseOp op, const int* active_flag) {
    if (active_flag && !*active_flag) return;
    dense_partial3_body<NT, PK>(op, (int)blockIdx.y);
}
// The same launch with a SIDE JOB in its first `side_rows` rows of workgroups (they are dispatched first): the SDDMM of the
// sphere / Euclidean Hess-vec (msdp_affine_dev.h, msdp_sddmm_side) -- its chain of dependent round trips hides under the matrix
// stream of the other workgroups instead of standing in front of it as a launch of its own.
template <int NT, int LPR>
__global__ __launch_bounds__(Dense3Cfg<NT>::WAVES * 64, 3) void k_dense_partial3_side(DenseOp op, const int* active_flag, SideJob sj, int side_rows) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (active_flag && !*active_flag) return;
    if ((int)blockIdx.y < side_rows) {
        msdp_sddmm_side<LPR, 1>(sj, (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, lds);
        return;
    }
    dense_partial3_body<NT, false>(op, (int)blockIdx.y - side_rows);
}

// ---------------------------------------------------------------- epilogues (oblique)
// eH(row) = sum of the SK slabs; then ManiSDP_onlyunitdiag.m:129 / ManiSDP_unitdiag.m:170.
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_dense_hess_epi_obl(Dev d, const double* slab, int64_t slab_stride, int SK) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (!d.F[0].active) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const int cur = d.ctl->cur;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ eG = cur ? d.eG[1] : d.eG[0];
    double pd = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            // (eight column chunks per lane, p = 513..1024: the rows of Y and U are read a second time in the second loop instead of
            // being held -- 64 registers per lane, which spilled 140 bytes at the 128-register budget of a 1024-thread workgroup)
            constexpr bool HOLD = NCH < 8;
            double2 acc[NCH], y[HOLD ? NCH : 1], u[HOLD ? NCH : 1];
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                acc[ch] = make_double2(0.0, 0.0);
                if (HOLD) { y[HOLD ? ch : 0] = acc[ch]; u[HOLD ? ch : 0] = acc[ch]; }
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    acc[ch] = msdp_sum_slabs(slab, slab_stride, SK, o);
                    const double2 yy = ld2(Yl + o);
                    if (HOLD) { y[HOLD ? ch : 0] = yy; u[HOLD ? ch : 0] = ld2(d.md + o); }
                    dot += acc[ch].x * yy.x + acc[ch].y * yy.y;
                }
            }
            dot = msdp_group_sum<LPR>(dot);
            if (d.rowfree && d.rowfree[row]) dot = 0.0;           // Euclidean block of a multiblock problem (eG is 0 there)
            const double eg = eG[row];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    const double2 yy = HOLD ? y[HOLD ? ch : 0] : ld2(Yl + o), uu = HOLD ? u[HOLD ? ch : 0] : ld2(d.md + o);
                    double2 h;
                    h.x = acc[ch].x - yy.x * dot - uu.x * eg;
                    h.y = acc[ch].y - yy.y * dot - uu.y * eg;
                    st2(d.Hmd + o, h);
                    pd += uu.x * h.x + uu.y * h.y;
                }
            }
        }
    }
    msdp_put_partial(d.P, P_DHD, pd, sh);
}

// cost/grad epilogue for dense-C onlyunitdiag: YC = sum slabs; eG, G, f, |G|^2
// (ManiSDP_onlyunitdiag.m:118-124).
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_dense_costgrad_epi_obl(Dev d, int slot, const double* slab,
                                                                     int64_t slab_stride, int SK) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (d.ctl->done) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const double* __restrict__ Yl = slot ? d.Y[1] : d.Y[0];
    double* __restrict__ Gr = slot ? d.Gr[1] : d.Gr[0];
    double* __restrict__ eG = slot ? d.eG[1] : d.eG[0];
    double pf = 0.0, pgg = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 acc[NCH], y[NCH];
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                acc[ch] = make_double2(0.0, 0.0); y[ch] = acc[ch];
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    acc[ch] = msdp_sum_slabs(slab, slab_stride, SK, o);
                    y[ch] = ld2(Yl + o);
                    dot += acc[ch].x * y[ch].x + acc[ch].y * y[ch].y;
                }
            }
            dot = msdp_group_sum<LPR>(dot);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    double2 gq;
                    gq.x = acc[ch].x - y[ch].x * dot;
                    gq.y = acc[ch].y - y[ch].y * dot;
                    st2(Gr + (int64_t)row * d.ld + col, gq);
                    pgg += gq.x * gq.x + gq.y * gq.y;
                }
            }
            if (sub == 0) { eG[row] = dot; pf += 0.5 * dot; }
        }
    }
    msdp_put_partials3(d.P, P_F, pf, P_GG, pgg, -1, 0.0, sh);
}

// ---------------------------------------------------------------- host side
int msdp_dev_alloc_bytes(msdp_handle h, void** out, size_t bytes);

static int ensure_slab(msdp_handle h, size_t need) {
    if (h->slab_cap >= need) return 0;
    if (h->slab) {
        for (size_t i = 0; i < h->allocs.size(); ++i)
            if (h->allocs[i] == h->slab) { h->allocs.erase(h->allocs.begin() + i); break; }
        (void)hipFree(h->slab);
        h->slab = nullptr;
        h->slab_cap = 0;
    }
    void* p = nullptr;
    int rc = msdp_dev_alloc_bytes(h, &p, need * sizeof(double));
    if (rc) return rc;
    h->slab = (double*)p;
    h->slab_cap = need;
    return 0;
}

int msdp_dense_ensure_slab(msdp_handle h, size_t need) { return ensure_slab(h, need); }
int msdp_densesym_eligible(msdp_handle h, int nmat);                   // msdp_densesym.hip
int msdp_densesym_reserve(msdp_handle h, int nmat, size_t* slabs_out);
int msdp_densesym_gemm(msdp_handle h, hipStream_t stream, int nmat, const double* const* M, const double* const* X,
                       const double* scale, const int* active_flag);

int msdp_dense_nS(int n) { return ((n + 15) / 16) * 16; }

typedef void (*dense3_fn_t)(DenseOp, const int*);
static dense3_fn_t dense3_fn(int NT, bool pk = false) {
    if (pk)
        switch (NT) {
            case 1: return k_dense_partial3<1, true>; case 2: return k_dense_partial3<2, true>; case 3: return k_dense_partial3<3, true>;
            case 4: return k_dense_partial3<4, true>; case 5: return k_dense_partial3<5, true>; case 6: return k_dense_partial3<6, true>;
            case 7: return k_dense_partial3<7, true>; default: return k_dense_partial3<8, true>;
        }
    switch (NT) {
        case 1: return k_dense_partial3<1, false>; case 2: return k_dense_partial3<2, false>; case 3: return k_dense_partial3<3, false>;
        case 4: return k_dense_partial3<4, false>; case 5: return k_dense_partial3<5, false>; case 6: return k_dense_partial3<6, false>;
        case 7: return k_dense_partial3<7, false>; default: return k_dense_partial3<8, false>;
    }
}
static int dense3_kt(int NT) { return NT <= 1 ? 64 : 32; }
static int dense3_waves(int NT) { return NT <= 2 ? 4 : 8; }
static size_t dense3_lds(int NT, int ldl) { return ((size_t)2 * dense3_kt(NT) * ldl + 128) * sizeof(double); }   // two tiles + dummy slots
static int dense_ldl(int ncols) {
    int ldl = ((ncols + 15) / 16) * 16;                      // zero-padded to 16*NT columns (no column predicate)
    while ((ldl & 7) != 4) ldl += 2;                         // ldl = 4 (mod 8): conflict-free B reads
    return ldl;
}
// Workgroups of k_dense_partial3<NT> the chip holds at once (registers and LDS), cached per NT.
static int dense3_capacity(int NT, int ldl) {
    static int cap[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    if (!cap[NT]) {
        int per_cu = 0;
        const size_t shmem = dense3_lds(NT, ldl);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)dense3_fn(NT), dense3_waves(NT) * 64, shmem) != hipSuccess || per_cu < 1)
            per_cu = 1;
        cap[NT] = per_cu * cus;
    }
    return cap[NT];
}

// Split-K plan.  The k range is cut into SK slices so that the launch fills the chip: the workgroup count
// row_blocks*SK should sit just below a multiple of what the chip holds at once (a partial last round idles most
// CUs), the slices should be equal (multiples of 16 k), and every extra slab costs a write and a read of n_loc x ld.
static void dense_plan(msdp_handle h, int nmat, int* row_blocks_out, int* SK_out, int64_t* kslice_out) {
    const Dev& d = h->d;
    const int nS = msdp_dense_nS(d.n);
    const int NT0 = (std::min(128, d.ld) + 15) / 16;         // the plan follows the first column block
    const int row_blocks = (d.n_loc + dense3_waves(NT0) * 16 - 1) / (dense3_waves(NT0) * 16);
    const int64_t Ktot = (int64_t)nmat * nS;
    int SK;
    int64_t kslice;
    {
        const int ncols = std::min(128, d.ld);
        const int NT = (ncols + 15) / 16;
        double cap = (double)dense3_capacity(NT, dense_ldl(ncols));
        // The affine kinds feed the slabs to epilogues that are chains of dependent launches and row sums (k_sph_hess_fused,
        // k_dense_hess_epi_obl behind the adjoint): there every slab costs more than its bytes, and two workgroups per CU stream
        // a 30-200 MB operand as fast as five.  Round 4, tools/affine_chain_probe.py --sk=..: theta n = 5000 16 slabs 67.5 us,
        // 6-12 slabs 61-62 us; BQP d = 60 29 slabs 59.6 us, 16 slabs 56.5 us (the dense kinds keep the full-occupancy plan:
        // tools/archive/dense_sk_probe.py has it within 1-5 % of the best slice count at n = 1000..20000, p = 16..64)
        if (d.costkind == COST_AFFINE) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
            cap = std::min(cap, 2.0 * cus);
        }
        double best = 1e300;
        SK = 1; kslice = ((Ktot + 15) / 16) * 16;
        for (int cand = 1; cand <= 32; ++cand) {
            if (h->tune.dense_sk > 0 && cand != h->tune.dense_sk) continue;      // A/B switch: this many k slices
            int64_t ks = (Ktot + cand - 1) / cand;
            ks = ((ks + 63) / 64) * 64;                       // whole tile pairs
            if (ks < 128 && cand > 1 && h->tune.dense_sk <= 0) break;
            const int sk = (int)((Ktot + ks - 1) / ks);
            const double W = (double)row_blocks * sk;
            const double rounds = ceil(W / cap);
            const double fill = W / (rounds * cap);
            const double balance = (double)Ktot / ((double)sk * (double)ks);
            const double slab = 1.0 + 2.0 * sk * (double)d.ld / (double)Ktot;
            const double cost = slab / (fill * balance);
            if (cost < best * 0.999) { best = cost; SK = sk; kslice = ks; }
        }
    }
    *row_blocks_out = row_blocks; *SK_out = SK; *kslice_out = kslice;
}

// Reserve the split-K slab for the current p BEFORE any graph capture (hipMalloc is illegal
// while a stream is capturing).
int msdp_dense_reserve(msdp_handle h, int nmat) {
    if (h->blocked) return ensure_slab(h, (size_t)2 * h->d.n * (size_t)h->d.ld);      // per-block contraction: slab 0 only (msdp_affine.hip)
    // room for the plans of 1..nmat matrices plus one extra slab (the affine Hess-vec appends the sparse A'(w)*Y
    // product as one more slab when At touches few entries)
    int SKmax = 1;
    for (int q = 1; q <= nmat; ++q) {
        int SK, row_blocks; int64_t kslice;
        dense_plan(h, q, &row_blocks, &SK, &kslice);
        SKmax = std::max(SKmax, SK);
        if (q == 1) SKmax = std::max(SKmax, nmat * SK);       // nmat one-matrix contractions side by side (msdp_dense_gemm_at)
    }
    const int64_t cap_rows = (h->d.n + h->nranks - 1) / h->nranks;
    size_t slabs = (size_t)(SKmax + 1);
    if (msdp_densesym_eligible(h, nmat)) {
        size_t sym_slabs = 0;
        int rc = msdp_densesym_reserve(h, nmat, &sym_slabs);
        if (rc) return rc;
        slabs = std::max(slabs, sym_slabs);
    }
    return ensure_slab(h, slabs * cap_rows * (size_t)h->d.ld);
}

// Launch the partial GEMM for up to two (matrix, panel, scale) pairs; returns slab info.
// Number of split-K slabs a contraction of nmat (matrix, panel) pairs writes at the current factor width.
int msdp_dense_gemm_slabs(msdp_handle h, int nmat) {
    int SK; int64_t kslice; int row_blocks;
    dense_plan(h, nmat, &row_blocks, &SK, &kslice);
    return SK;
}
int msdp_dense_gemm_at(msdp_handle h, hipStream_t stream, int slab_first, int slabs_reserve, int nmat, const double* const* M,
                       const double* const* X, const double* scale, const int* active_flag, const double** slab_out,
                       int64_t* stride_out, int* SK_out);
int msdp_dense_gemm(msdp_handle h, int nmat, const double* const* M, const double* const* X, const double* scale,
                    const int* active_flag, const double** slab_out, int64_t* stride_out, int* SK_out) {
    return msdp_dense_gemm_at(h, h->stream, 0, 0, nmat, M, X, scale, active_flag, slab_out, stride_out, SK_out);
}
// The same on `stream`, writing its slabs behind the first `slab_first` ones of the handle's slab buffer.  slabs_reserve: how
// many slabs the caller will fill in total (several contractions whose slabs one epilogue sums; the buffer must not be
// re-allocated between them) -- 0: this call's own count.  *slab_out is the start of the WHOLE buffer.
int msdp_dense_gemm_at(msdp_handle h, hipStream_t stream, int slab_first, int slabs_reserve, int nmat, const double* const* M,
                       const double* const* X, const double* scale, const int* active_flag, const double** slab_out,
                       int64_t* stride_out, int* SK_out) {
    Dev& d = h->d;
    if (slab_first == 0 && slabs_reserve == 0 && msdp_densesym_eligible(h, nmat)) {
        // symmetric operands: the upper triangle only; the sum of all partial slabs arrives in slab 0
        int rc = msdp_densesym_gemm(h, stream, nmat, M, X, scale, active_flag);
        if (rc) return rc;
        *slab_out = h->slab;
        *stride_out = (int64_t)d.n * d.ld;
        *SK_out = 1;
        return 0;
    }
    DenseOp op;
    memset(&op, 0, sizeof(op));
    op.nmat = nmat;
    for (int m = 0; m < nmat; ++m) { op.M[m] = M[m]; op.X[m] = X[m]; op.scale[m] = scale[m]; }
    if (nmat == 1) { op.M[1] = M[0]; op.X[1] = X[0]; op.scale[1] = 0.0; }
    const bool pk = nmat == 1 && d.Cpk && M[0] == d.Cd && h->tune.dense_pack;                // the constant dense cost matrix has a fragment-ordered copy
    op.Mpk = pk ? d.Cpk : nullptr;
    op.n = d.n;
    op.nS = msdp_dense_nS(d.n);
    op.n_loc = d.n_loc;
    op.ld = d.ld;
    int SK; int64_t kslice; int row_blocks;
    dense_plan(h, nmat, &row_blocks, &SK, &kslice);
    op.SK = SK;
    op.kslice = (int)kslice;
    const int64_t cap_rows = (d.n + h->nranks - 1) / h->nranks;
    op.slab_stride = cap_rows * (int64_t)d.ld;
    const int want_slabs = std::max(slab_first + SK, slabs_reserve) + 1;      // + 1: see msdp_dense_reserve
    int rc = ensure_slab(h, (size_t)want_slabs * op.slab_stride);
    if (rc) return rc;
    op.slab = h->slab + (size_t)slab_first * op.slab_stride;
    if (h->tune.block_skip) { op.blk_lo = d.blk_lo; op.blk_hi = d.blk_hi; }
    // p > 128: column blocks of 128 (the matrix is re-streamed once per block; NT <= 8 accumulator tiles per wave)
    for (int colofs = 0; colofs < d.ld; colofs += 128) {
        op.colofs = colofs;
        op.ncols = std::min(128, d.ld - colofs);
        const int ldl = dense_ldl(op.ncols);
        op.ldl = ldl;
        const int NT = (op.ncols + 15) / 16;
        const int rows_wg = dense3_waves(NT) * 16;
        dim3 grid((d.n_loc + rows_wg - 1) / rows_wg, SK), block(dense3_waves(NT) * 64);
        hipLaunchKernelGGL(dense3_fn(NT, pk), grid, block, dense3_lds(NT, ldl), stream, op, active_flag);
    }
    HIPCHK(hipGetLastError());
    *slab_out = h->slab;
    *stride_out = op.slab_stride;
    *SK_out = SK;
    return 0;
}

// One-matrix contraction with the SDDMM side job (sphere Hess-vec on the SDDMM route; msdp_affine.hip).  *njobs_out = the number of
// side workgroups (= entries of P_T1..P_T3).  MSDP_EUNSUPPORTED when the shape does not fit (p > 32, symmetric route, sharding):
// the caller then launches k_sddmm1 and the plain contraction.
int msdp_dense_gemm_side(msdp_handle h, const double* M, const double* X, double scale, const int* active_flag, SideJob sj,
                         int* njobs_out, const double** slab_out, int64_t* stride_out, int* SK_out) {
    Dev& d = h->d;
    if (d.ld > 32 || h->nranks != 1 || h->use_comm || d.n_loc != d.n || msdp_densesym_eligible(h, 1) || (d.blk_lo && h->tune.block_skip)) return MSDP_EUNSUPPORTED;
    DenseOp op;
    memset(&op, 0, sizeof(op));
    op.nmat = 1;
    op.M[0] = M; op.X[0] = X; op.scale[0] = scale;
    op.M[1] = M; op.X[1] = X; op.scale[1] = 0.0;
    op.n = d.n; op.nS = msdp_dense_nS(d.n); op.n_loc = d.n_loc; op.ld = d.ld;
    int SK; int64_t kslice; int row_blocks;
    dense_plan(h, 1, &row_blocks, &SK, &kslice);
    op.SK = SK; op.kslice = (int)kslice;
    op.slab_stride = (int64_t)d.n * d.ld;
    int rc = ensure_slab(h, (size_t)(SK + 1) * op.slab_stride);      // what msdp_dense_reserve has set aside (no allocation inside a capture)
    if (rc) return rc;
    op.slab = h->slab;
    op.colofs = 0; op.ncols = d.ld;
    const int ldl = dense_ldl(op.ncols);
    op.ldl = ldl;
    const int NT = (op.ncols + 15) / 16;
    int half = d.ld / 2, lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    const int rows_wg = dense3_waves(NT) * 16;
    const int gx = (d.n_loc + rows_wg - 1) / rows_wg;
    // side workgroups: 4 waves x (64 / lpr) units per pass, about five passes each, at most 510 of them (partial-sum slots)
    const int64_t nunits = (int64_t)sj.a.nshort + sj.a.nlit;
    const int64_t per_job = (int64_t)4 * (64 / lpr) * 5;
    int side_rows = (int)((nunits + per_job * gx - 1) / (per_job * gx));
    if (side_rows < 1) side_rows = 1;
    while (side_rows > 1 && side_rows * gx > 510) --side_rows;
    if (side_rows * gx > 510) return MSDP_EUNSUPPORTED;
    sj.njobs = side_rows * gx; sj.n_loc = d.n_loc; sj.ld = d.ld;
    dim3 grid(gx, SK + side_rows), block(dense3_waves(NT) * 64);
    const size_t lds = dense3_lds(NT, ldl);
    if (NT == 2) hipLaunchKernelGGL((k_dense_partial3_side<2, 16>), grid, block, lds, h->stream, op, active_flag, sj, side_rows);
    else switch (lpr) {
        case 1: hipLaunchKernelGGL((k_dense_partial3_side<1, 1>), grid, block, lds, h->stream, op, active_flag, sj, side_rows); break;
        case 2: hipLaunchKernelGGL((k_dense_partial3_side<1, 2>), grid, block, lds, h->stream, op, active_flag, sj, side_rows); break;
        case 4: hipLaunchKernelGGL((k_dense_partial3_side<1, 4>), grid, block, lds, h->stream, op, active_flag, sj, side_rows); break;
        default: hipLaunchKernelGGL((k_dense_partial3_side<1, 8>), grid, block, lds, h->stream, op, active_flag, sj, side_rows); break;
    }
    HIPCHK(hipGetLastError());
    *njobs_out = sj.njobs;
    *slab_out = h->slab; *stride_out = op.slab_stride; *SK_out = SK;
    return 0;
}

#define DISPATCH_LPR_D(KERNEL, h, ...)                                                               \
    do {                                                                                             \
        int half = (h)->d.ld / 2, lpr = 1;                                                           \
        while (lpr < half && lpr < 64) lpr <<= 1;                                                    \
        int nch = (half + lpr - 1) / lpr; if (nch < 1) nch = 1;                                      \
        dim3 grid((h)->d.G), block(MSDP_BLOCK);                                                      \
        if (nch == 1) {                                                                              \
            switch (lpr) {                                                                           \
                case 1:  hipLaunchKernelGGL((KERNEL<1, 1>),  grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                case 2:  hipLaunchKernelGGL((KERNEL<2, 1>),  grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                case 4:  hipLaunchKernelGGL((KERNEL<4, 1>),  grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                case 8:  hipLaunchKernelGGL((KERNEL<8, 1>),  grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                case 16: hipLaunchKernelGGL((KERNEL<16, 1>), grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32, 1>), grid, block, 0, (h)->stream, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64, 1>), grid, block, 0, (h)->stream, __VA_ARGS__); break; \
            }                                                                                        \
        } else if (nch == 2) {                                                                       \
            hipLaunchKernelGGL((KERNEL<64, 2>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else if (nch <= 4) {                                                                       \
            hipLaunchKernelGGL((KERNEL<64, 4>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else if (nch <= 8) {                                                                       \
            hipLaunchKernelGGL((KERNEL<64, 8>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else {                                                                                     \
            msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", (h)->d.p);    \
            return MSDP_EUNSUPPORTED;                                                                \
        }                                                                                            \
    } while (0)

// Fragment-ordered copy of the local rows of the cost matrix (k_dense_partial3<NT, true> reads it); rows beyond
// n_loc in the last 16-row block are zero.
__global__ void k_pack_fragments(const double* __restrict__ Cd, double* __restrict__ Cpk, int nS, int n_loc, int64_t tot) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tile = e >> 8;
        const int w = (int)(e & 255);
        const int L = (w & 127) >> 1, t = ((w >> 7) << 1) | (w & 1);
        const int64_t rb = tile / (nS >> 4), kb = tile - rb * (nS >> 4);
        const int64_t row = rb * 16 + (L & 15), k = kb * 16 + 4 * (L >> 4) + t;
        Cpk[e] = row < n_loc ? Cd[row * nS + k] : 0.0;
    }
}
static int dense_pack(msdp_handle h) {
    Dev& d = h->d;
    const int nS = msdp_dense_nS(d.n);
    const int64_t tot = (int64_t)((d.n_loc + 15) / 16) * 16 * nS;
    void* p = nullptr;
    int rc = msdp_dev_alloc_bytes(h, &p, (size_t)tot * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_fragments, dim3(4096), dim3(256), 0, h->stream, d.Cd, (double*)p, nS, d.n_loc, tot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    d.Cpk = (double*)p;
    return 0;
}

// Upload a dense symmetric n x n matrix (host, column-major == row-major) with the padded
// leading dimension nS.
int msdp_dense_setup(msdp_handle h, const double* C) {
    Dev& d = h->d;
    const int nS = msdp_dense_nS(d.n);
    void* p = nullptr;
    int rc = msdp_dev_alloc_bytes(h, &p, (size_t)d.n * nS * sizeof(double));
    if (rc) return rc;
    d.Cd = (double*)p;
    HIPCHK(hipMemset(d.Cd, 0, (size_t)d.n * nS * sizeof(double)));
    HIPCHK(msdp_memcpy2d(d.Cd, (size_t)nS * sizeof(double), C, (size_t)d.n * sizeof(double), (size_t)d.n * sizeof(double),
                       d.n, hipMemcpyHostToDevice));
    {
        bool sym = true;
        for (int i = 0; i < d.n && sym; ++i)
            for (int j = i + 1; j < d.n; ++j)
                if (C[(size_t)i * d.n + j] != C[(size_t)j * d.n + i]) { sym = false; break; }
        h->dense_symmetric = sym;
    }
    return dense_pack(h);
}

// Counter-based generator of a dense symmetric test matrix: entry (i,j) depends only on (min,max,seed), so
// every rank can fill ITS rows on the device without any host array (the full
// n = 100000 matrix would be 80 GB).  Uniform in (-1,1)/sqrt(n).
__host__ __device__ inline double msdp_syn_entry(int64_t n, int64_t i, int64_t j, uint64_t seed) {
    const int64_t a = i < j ? i : j, b = i < j ? j : i;
    uint64_t x = (uint64_t)(a * n + b) + seed * 0x9E3779B97F4A7C15ULL;
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x = x ^ (x >> 31);
    const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);      // [0,1)
    return (2.0 * u - 1.0) / sqrt((double)n);
}
extern "C" double msdp_synthetic_dense_entry(int64_t n, int64_t i, int64_t j, uint64_t seed) {
    return msdp_syn_entry(n, i, j, seed);
}
__global__ void k_fill_dense_sym(double* __restrict__ Cd, int n, int nS, int row0, int n_loc, uint64_t seed) {
    const int64_t tot = (int64_t)n_loc * nS;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t il = e / nS, j = e - il * nS;
        Cd[e] = (j < n) ? msdp_syn_entry(n, row0 + il, j, seed) : 0.0;
    }
}
int msdp_dense_setup_synthetic(msdp_handle h, uint64_t seed) {
    Dev& d = h->d;
    const int nS = msdp_dense_nS(d.n);
    void* p = nullptr;
    const size_t rows = (size_t)((d.n + h->nranks - 1) / h->nranks);
    int rc = msdp_dev_alloc_bytes(h, &p, rows * nS * sizeof(double));
    if (rc) return rc;
    d.Cd = (double*)p;
    HIPCHK(hipMemsetAsync(d.Cd, 0, rows * nS * sizeof(double), h->stream));
    hipLaunchKernelGGL(k_fill_dense_sym, dim3(4096), dim3(256), 0, h->stream, d.Cd, d.n, nS, d.row0, d.n_loc, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->dense_symmetric = true;                                // entry (i,j) depends on (min, max) only
    return dense_pack(h);
}

int msdp_dense_costgrad(msdp_handle h, int slot) {
    Dev& d = h->d;
    const double* M[1] = {d.Cd};
    const double* X[1] = {d.full};
    const double sc[1] = {1.0};
    const double* slab; int64_t stride; int SK;
    int rc = msdp_dense_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK);
    if (rc) return rc;
    DISPATCH_LPR_D(k_dense_costgrad_epi_obl, h, d, slot, slab, stride, SK);
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_dense_hess(msdp_handle h) {
    Dev& d = h->d;
    const double* M[1] = {d.Cd};
    const double* X[1] = {d.full};
    const double sc[1] = {1.0};
    const double* slab; int64_t stride; int SK;
    int rc = msdp_dense_gemm(h, 1, M, X, sc, &d.F[0].active, &slab, &stride, &SK);
    if (rc) return rc;
    DISPATCH_LPR_D(k_dense_hess_epi_obl, h, d, slab, stride, SK);
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_dense_hess_epilogue_obl(msdp_handle h, const double* slab, int64_t stride, int SK) {
    Dev& d = h->d;
    DISPATCH_LPR_D(k_dense_hess_epi_obl, h, d, slab, stride, SK);
    HIPCHK(hipGetLastError());
    return 0;
}
