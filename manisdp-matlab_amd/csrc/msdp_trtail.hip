// msdp_trtail.hip -- everything of a trust-region iteration that follows the tCG solve, in ONE launch
// (sparse C, oblique manifold, the persistent path of msdp_persist.hip):
//   x_prop = retr(x, eta)                      ManiSDP_onlyunitdiag.m:142-145, trustregions.m:540
//   <eta, grad + .5*Heta>                      trustregions.m:549-550
//   cost / gradient at x_prop                  ManiSDP_onlyunitdiag.m:117-125, trustregions.m:544
//   rho, radius update, accept / reject, stop  trustregions.m:548-729
// Before: k_retract_obl, k_costgrad_*, k_rtr_decide + the slot-reset launch = 4 launches per TR iteration
// next to the persistent tCG kernel; now 1.  The kernel needs one grid barrier (the proposal rows must be
// in place before the S*Y_prop gathers) and one grid reduction (f, |grad|^2, model decrease), for which it
// uses region B of the synchronisation slots; it clears region A for the next tCG launch.
#include <type_traits>
#include "msdp_psync.h"
#include <math.h>
#include <cstdlib>

// XR (round 5): the members of a row-sharded group run the tail TOGETHER, like the cross-rank persistent tCG in front of it
// (msdp_persist.hip): the proposal rows travel through the members' exchange buffers (every member writes its rows into its own; the
// tCG does not touch them between its last reduction and its next launch), the barrier and the reduction run over the N x G slots of a slot
// region of the group (regions 2 and 3 of the shared block, alternating with the TR iteration; every launch clears the other one),
// and every member takes the decision of trustregions.m:548-729 from the same sums in the same order -- no collective.
template <int LPR, bool XR = false, bool XR2 = false>
__global__ __launch_bounds__(PB) void k_tr_tail_obl(Dev d, unsigned long long* slots, int* err, int rcap) {
    extern __shared__ double lds[];
    __shared__ double sh[3 * PWAVES];
    __shared__ double shb[8];
    constexpr int RPW = 64 / LPR;
    constexpr int RSTEP = PWAVES * RPW;
    double2* YPs = reinterpret_cast<double2*>(lds);            // [rcap][PB] proposal rows of this workgroup
    Ctl* c = d.ctl;
    if (c->done) return;
    // two-level form (round 6, d.xr2_on: members on devices of their own, up to 8; msdp_psync.h psync2): regions 2 / 3 of the member's block
    constexpr bool two = XR && XR2;
    __shared__ unsigned long long* shpeer[two ? 8 : 1];
    const int bid = (XR && !two) ? d.xr_gid0 + (int)blockIdx.x : (int)blockIdx.x;
    const int GS = (XR && !two) ? d.xr_gtot : d.G;
    const int xri = 2 + (c->k & 1);
    unsigned long long* sb = slots;
    if (two) {
        if (threadIdx.x < 8) shpeer[threadIdx.x] = threadIdx.x < d.xr2_n ? d.xr2_peers[threadIdx.x] : d.xr2_blk;
        psync2_reset_other(d.xr2_blk, 2 + ((c->k & 1) ^ 1), bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (XR) {
        sb = slots + (size_t)(2 + (c->k & 1)) * PSYNC_REGION;
        psync_reset_other(slots + (size_t)(2 + ((c->k & 1) ^ 1)) * PSYNC_REGION, bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        psync_reset_other(slots, bid, GS);                     // region A belongs to the persistent tCG kernel
        sb = slots + PSYNC_REGION;
    }
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    constexpr int RMAX = (LPR / 4 < 4) ? LPR / 4 : 4;            // row slots processed together (their loads overlap)
    const int R = (hi - lo + RSTEP - 1) / RSTEP;               // row slots actually needed (<= rcap: the host sizes the LDS for the largest chunk)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const bool colok = 2 * sub < d.ld;
    const int slot0 = wave * RPW + rsub;
    const int cur = c->cur, ix = d.F[0].eta_idx;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ gl = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ Heta = ix ? d.Heta[1] : d.Heta[0];
    const unsigned vec_bytes = (unsigned)((size_t)d.n_loc * d.ld * sizeof(double));
    // XR ("push" exchange, msdp_persist.hip): the proposal rows go to this member's own exchange buffer and to the halo slots of the
    // members that reference them; every gather is a local load with the buffer-local column indices d.xr_colind
    const unsigned xr_bytes = XR ? (unsigned)(((size_t)d.xr_cap + (size_t)d.xr_halo) * d.ld * sizeof(double)) : 0u;
    double* xr_own = d.xr_rows[0];
    if (XR) { if (d.xr_me == 1) xr_own = d.xr_rows[1]; if (d.xr_me == 2) xr_own = d.xr_rows[2]; if (d.xr_me == 3) xr_own = d.xr_rows[3]; }
    __amdgpu_buffer_rsrc_t rs_yp = XR ? __builtin_amdgcn_make_buffer_rsrc(xr_own, 0, xr_bytes, 0x00020000)
                                      : __builtin_amdgcn_make_buffer_rsrc(cur ? d.Y[0] : d.Y[1], 0, vec_bytes, 0x00020000);
    auto xr_push = [&](int row, bool ok, double2 v) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned long long a = ok ? d.xr_paddr[(int64_t)t * d.n_loc + row] : 0ULL;
            if (a == 0ULL) continue;
            double* ptr = reinterpret_cast<double*>(a) + 2 * sub;
            if (two) {                                             // the slot may live on another device: system scope
                __hip_atomic_store(ptr, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(ptr + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_store(ptr, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ptr + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    double* __restrict__ Ypl = cur ? d.Y[0] : d.Y[1];          // XR: the member's own copy of its proposal rows
    double* __restrict__ Gp = cur ? d.Gr[0] : d.Gr[1];
    double* __restrict__ eGp = cur ? d.eG[0] : d.eG[1];
    const double2 zz = make_double2(0.0, 0.0);
    double prd = 0.0, pf = 0.0, pgg = 0.0;
    for (int r0 = 0; r0 < R; r0 += RMAX) {
        double2 y[RMAX], g[RMAX], e[RMAX], he[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int row = lo + (r0 + q) * RSTEP + slot0;
            const int rc = row < hi ? row : lo;
            const int64_t o = (int64_t)rc * d.ld + (colok ? 2 * sub : 0);
            y[q] = ld2(Yl + o); g[q] = ld2(gl + o); e[q] = ld2(eta + o); he[q] = ld2(Heta + o);
        }
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int r = r0 + q;
            const int row = lo + r * RSTEP + slot0;
            const bool ok = row < hi && colok;
            double2 x = zz;
            if (ok) {
                prd += e[q].x * (g[q].x + 0.5 * he[q].x) + e[q].y * (g[q].y + 0.5 * he[q].y);
                x = make_double2(y[q].x + e[q].x, y[q].y + e[q].y);
            }
            double nn = sqrt(msdp_group_sum<LPR>(x.x * x.x + x.y * x.y));
            if (!(nn > 0.0)) nn = 1.0;
            const double2 ypr = ok ? make_double2(x.x / nn, x.y / nn) : zz;
            if (r < rcap) YPs[r * PB + threadIdx.x] = ypr;
            if (ok) st2_sc1(rs_yp, ((unsigned)row * (unsigned)d.ld + 2 * sub) * 8u, ypr);
            if (XR) xr_push(row < hi ? row : lo, ok, ypr);
            if (XR && ok) st2(Ypl + (int64_t)row * d.ld + 2 * sub, ypr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my proposal rows are performed before my workgroup arrives
    const int backoff = d.ctl->psync_backoff;
    if (two) {
        double z0 = 0.0, z1 = 0.0, z2 = 0.0;
        if (!psync2(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, 0, GS, 0, z0, z1, z2, sh, shb, err, bid, backoff)) return;
    } else if (!pbarrier(sb, 0, GS, shb, err, bid)) return;
    for (int r0 = 0; r0 < R; r0 += RMAX) {
        int s0[RMAX], s1[RMAX];
        double2 acc[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int row = lo + (r0 + q) * RSTEP + slot0;
            const bool rok = row < hi;
            s0[q] = rok ? d.rowptr[row] : 0;
            s1[q] = rok ? d.rowptr[row + 1] : 0;
            acc[q] = zz;
        }
        int len = 0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q) len = max(len, s1[q] - s0[q]);
        // Round 5: rows stored in ELL form too (width 5 or 8: the persistent kernels' copy) take ONE dependent load level -- column
        // index, then the row -- with all RMAX x width gathers in flight; the CSR walk below pays row pointer -> (column, value)
        // -> row per batch of four entries (k_tr_tail_obl<32>: 31.7 us per launch on G81 at p = 40, most of it these round trips)
        const int ellw = (d.ellW == 5 || d.ellW == 8) && d.ellc && (!XR || d.xr_ellc) ? d.ellW : 0;
        if (ellw) {
            len = 0;                                            // (skips the CSR walk)
            auto ell_rows = [&](auto wc) {
                constexpr int W = decltype(wc)::value;
                double2 x[RMAX][W];
                double cv[RMAX][W];
#pragma unroll
                for (int q = 0; q < RMAX; ++q) {
                    const int row = lo + (r0 + q) * RSTEP + slot0;
                    const int rc = row < hi ? row : lo;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const int col = XR ? d.xr_ellc[(int64_t)w * d.ell_stride + rc] : d.ellc[(int64_t)w * d.ell_stride + rc];
                        cv[q][w] = row < hi ? d.ellv[(int64_t)w * d.ell_stride + rc] : 0.0;
                        x[q][w] = ld2_sc1(rs_yp, ((unsigned)col * (unsigned)d.ld + (colok ? 2 * sub : 0)) * 8u);
                    }
                }
#pragma unroll
                for (int q = 0; q < RMAX; ++q)
#pragma unroll
                    for (int w = 0; w < W; ++w) { acc[q].x = fma(cv[q][w], x[q][w].x, acc[q].x); acc[q].y = fma(cv[q][w], x[q][w].y, acc[q].y); }
            };
            if (ellw == 5) ell_rows(std::integral_constant<int, 5>()); else ell_rows(std::integral_constant<int, 8>());
        }
        // CSR rows of C (static data: plain loads); the proposal rows of other workgroups through sc1.  The RMAX
        // rows advance together, 4 entries each per batch, so 4*RMAX gathers are in flight.
        for (int kb = 0; kb < len; kb += 4) {
            double2 x[RMAX][4];
            double cv[RMAX][4];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = s0[q] + kb + u < s1[q];
                    const int k = in ? s0[q] + kb + u : (s1[q] > s0[q] ? s1[q] - 1 : 0);
                    cv[q][u] = in ? d.cval[k] : 0.0;
                    const int col = (s1[q] > s0[q]) ? (XR ? d.xr_colind[k] : d.colind[k]) : lo;
                    x[q][u] = ld2_sc1(rs_yp, ((unsigned)col * (unsigned)d.ld + (colok ? 2 * sub : 0)) * 8u);
                }
            }
#pragma unroll
            for (int q = 0; q < RMAX; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc[q].x = fma(cv[q][u], x[q][u].x, acc[q].x); acc[q].y = fma(cv[q][u], x[q][u].y, acc[q].y); }
        }
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int r = r0 + q;
            const int row = lo + r * RSTEP + slot0;
            const bool rok = row < hi, ok = rok && colok;
            double2 a2 = colok ? acc[q] : zz;
            const double2 ypr = (r < rcap) ? YPs[r * PB + threadIdx.x] : zz;
            const double dot = msdp_group_sum<LPR>(a2.x * ypr.x + a2.y * ypr.y);     // eG(row) = sum(YC.*Y)
            const double2 gpr = ok ? make_double2(a2.x - ypr.x * dot, a2.y - ypr.y * dot) : zz;   // G = YC - Y.*eG
            pgg += gpr.x * gpr.x + gpr.y * gpr.y;
            if (ok) st2(Gp + (int64_t)row * d.ld + 2 * sub, gpr);
            if (sub == 0 && rok) { pf += 0.5 * dot; eGp[row] = dot; }
        }
    }
    if (two) { if (!psync2(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, 1, GS, 3, pf, pgg, prd, sh, shb, err, bid, backoff)) return; }
    else if (!psync(sb, 0, GS, 3, pf, pgg, prd, sh, shb, err, bid, backoff)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                 // trustregions.m:548-729 (same arithmetic as k_rtr_decide)
        const int f_stop = d.F[0].stop, f_j = d.F[0].j;
        const double fp = pf, ggp = pgg;
        double rhonum = c->fx - fp;                                          // :548
        double rhoden = -prd;                                                // :550
        const double rho_reg = fmax(1.0, fabs(c->fx)) * 2.220446049250313e-16 * c->rho_reg;   // :579
        rhonum += rho_reg;
        rhoden += rho_reg;
        const bool model_decreased = rhoden >= 0.0;                          // :614
        const double rho = rhonum / rhoden;                                  // :621
        if (rho < 0.25 || !model_decreased || isnan(rho)) {                  // :653
            c->Delta = c->Delta / 4.0;
        } else if (rho > 0.75 && (f_stop == 1 || f_stop == 2)) {             // :669
            c->Delta = fmin(2.0 * c->Delta, c->Delta_bar);
        }
        if (model_decreased && rho > c->rho_prime) {                         // :688
            c->cur ^= 1;
            c->fx = fp; c->gg = ggp; c->norm_grad = sqrt(ggp);
            c->accepted++;
        } else {
            c->rejected++;
        }
        c->rho = rho; c->rhonum = rhonum; c->rhoden = rhoden; c->fx_prop = fp; c->gg_prop = ggp;
        c->k++;                                                              // :729
        c->hessvecs += f_j;
        c->cost_evals++;
        c->last_stop_inner = f_stop;
        c->done = (c->norm_grad < c->tolgradnorm) || (c->k >= c->maxiter);
    }
}

static int tail_lpr(const Dev& d) {
    int half = d.ld / 2, lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    if (lpr < 8) lpr = 8;
    return lpr;
}

// Same grid as the persistent tCG kernel (one workgroup per CU, all co-resident); LDS = one proposal row set.
int msdp_tr_tail_grid(msdp_handle h);           // msdp_persist.hip (persist_grid)
int msdp_launch_tr_tail(msdp_handle h) {
    const int G = msdp_tr_tail_grid(h);
    const int lpr = tail_lpr(h->d);
    if (G < 8 || lpr > 32) { msdp_set_error("TR tail: not eligible"); return MSDP_ESTATE; }
    // one LDS row set per row slot of the largest chunk (3 slots for G81 at p = 32, 8 for the eight-slot instances)
    const int rstep = PWAVES * (64 / lpr);
    int rcap = ((h->d.n_loc + G - 1) / G + rstep - 1) / rstep;
    if (rcap < 1) rcap = 1;
    const size_t lds = (size_t)rcap * PB * sizeof(double2);
    if (lds > 128 * 1024) { msdp_set_error("TR tail: %d row slots do not fit the LDS", rcap); return MSDP_ESTATE; }
    Dev dp = h->d;
    dp.G = G;
    typedef void (*fn_t)(Dev, unsigned long long*, int*, int);
    fn_t fn = lpr == 8 ? k_tr_tail_obl<8> : (lpr == 16 ? k_tr_tail_obl<16> : k_tr_tail_obl<32>);
    static bool attr_set[3] = {false, false, false};
    const int ai = lpr == 8 ? 0 : (lpr == 16 ? 1 : 2);
    if (!attr_set[ai]) {
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set[ai] = true;
    }
    hipLaunchKernelGGL(fn, dim3(G), dim3(PB), lds, h->stream, dp, h->psync_slots, h->psync_err, rcap);
    HIPCHK(hipGetLastError());
    return 0;
}

// XR: this member's launch of the tail the group runs together (separate processes: msdp_comm_init_ipc).  `dv` = the member's Dev as
// msdp_xpersist_member filled it (G, xr_gid0, xr_gtot, xr_rows), slots = the group's shared block (regions 2 / 3).
int msdp_launch_tr_tail_xr(hipStream_t stream, const Dev& dv, unsigned long long* slots, int* err) {
    const int lpr = tail_lpr(dv);
    if (lpr > 32) { msdp_set_error("cross-rank TR tail: not eligible"); return MSDP_ESTATE; }
    const int rstep = PWAVES * (64 / lpr);
    int rcap = ((dv.n_loc + dv.G - 1) / dv.G + rstep - 1) / rstep;
    if (rcap < 1) rcap = 1;
    const size_t lds = (size_t)rcap * PB * sizeof(double2);
    if (lds > 128 * 1024) { msdp_set_error("cross-rank TR tail: %d row slots do not fit the LDS", rcap); return MSDP_ESTATE; }
    typedef void (*fn_t)(Dev, unsigned long long*, int*, int);
    fn_t fn = lpr == 8 ? k_tr_tail_obl<8, true> : (lpr == 16 ? k_tr_tail_obl<16, true> : k_tr_tail_obl<32, true>);
    if (dv.xr2_on) fn = lpr == 8 ? k_tr_tail_obl<8, true, true> : (lpr == 16 ? k_tr_tail_obl<16, true, true> : k_tr_tail_obl<32, true, true>);
    static bool attr_set[6] = {false, false, false, false, false, false};
    const int ai = (lpr == 8 ? 0 : (lpr == 16 ? 1 : 2)) + (dv.xr2_on ? 3 : 0);
    if (!attr_set[ai]) {
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set[ai] = true;
    }
    hipLaunchKernelGGL(fn, dim3(dv.G), dim3(PB), lds, stream, dv, slots, err, rcap);
    HIPCHK(hipGetLastError());
    return 0;
}
