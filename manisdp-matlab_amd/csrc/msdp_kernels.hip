// msdp_kernels.hip -- row-tiled fp64 kernels of the oblique sparse-C path and the
// manifold-generic tCG / RTR bookkeeping kernels.  gfx950 (wave64) only.
//
// Data layout: every n x p factor / tangent vector is row-major with row stride
// ld (even, pad columns identically zero) -- byte-identical to MATLAB's p x n
// column-major Y when ld == p.  A row is served by LPR = pow2 >= ld/2 lanes, each
// lane owning one double2 (16 B: the coalescing sweet spot), so a wave covers
// 64/LPR rows and the per-row dot products are LPR-lane shuffles.
//
// Reference expressions reproduced here:
//   cost/grad  ManiSDP_onlyunitdiag.m:117-125   YC=Y*C; eG=sum(YC.*Y); f=.5*sum(eG); G=YC-Y.*eG
//   hess       ManiSDP_onlyunitdiag.m:127-130   eH=U*C; H=eH-Y.*sum(Y.*eH)-U.*eG
//   manifold   ManiSDP_onlyunitdiag.m:132-156   proj/tangent, retr
//   tCG        manopt7.0/manopt/solvers/trustregions/tCG.m:160-289
//   RTR        manopt7.0/manopt/solvers/trustregions/trustregions.m:405-409,540-729
#include "msdp_device.h"
#include <math.h>
#include <cstdlib>

// spmm_row<LPR, NCH, ELL>: one row of C times the gathered vector -- msdp_device.h (shared with msdp_trip1.hip)

// cost + Riemannian gradient at Y[slot] (gather source: d.full holds all rows of Y[slot]).
template <int LPR, int NCH, bool ELL>
__device__ __forceinline__ void costgrad_sparse_obl_body(const Dev& d, int slot) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (d.ctl->done) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const bool rel = slot >= 2;                      // 2: current slot, 3: proposal slot, resolved on the device
    if (rel) slot = d.ctl->cur ^ (slot & 1);
    const double* __restrict__ Yl = slot ? d.Y[1] : d.Y[0];
    const double* __restrict__ Xf = rel ? Yl : d.full;
    double* __restrict__ Gr = slot ? d.Gr[1] : d.Gr[0];
    double* __restrict__ eG = slot ? d.eG[1] : d.eG[0];
    double pf = 0.0, pgg = 0.0;
    int stride = MSDP_WAVES * RPW;
    if (d.sweep) msdp_sweep_rows(d.n_loc, d.G, MSDP_WAVES * RPW, lo, hi, stride);      // windowed traversal, see hess_sparse_obl_body
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += stride) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 acc[NCH];
            double2 y[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[ch] = make_double2(0.0, 0.0);
                const int col = 2 * sub + ch * 2 * LPR;
                y[ch] = (col < d.ld) ? ld2(Yl + (int64_t)row * d.ld + col) : make_double2(0.0, 0.0);
            }
            spmm_row<LPR, NCH, ELL>(d, row, sub, Xf, acc);
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) dot += acc[ch].x * y[ch].x + acc[ch].y * y[ch].y;
            dot = msdp_group_sum<LPR>(dot);          // eG(row) = sum(YC.*Y)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    double2 g;
                    g.x = acc[ch].x - y[ch].x * dot;  // G = YC - Y.*eG
                    g.y = acc[ch].y - y[ch].y * dot;
                    st2(Gr + (int64_t)row * d.ld + col, g);
                    pgg += g.x * g.x + g.y * g.y;
                }
            }
            if (sub == 0) { eG[row] = dot; pf += 0.5 * dot; }
        }
    }
    msdp_put_partials3(d.P, P_F, pf, P_GG, pgg, -1, 0.0, sh);
}

// Hess-vec: Hmd = proj-fused (C*md - Y.*rowdot(Y, C*md) - md.*eG), partial <md, Hmd>.
template <int LPR, int NCH, bool ELL>
__device__ __forceinline__ void hess_sparse_obl_body(const Dev& d) {
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
    const double* __restrict__ Uf = d.full;
    const double* __restrict__ Ul = d.md;
    double* __restrict__ H = d.Hmd;
    double pd = 0.0;
    // vectors beyond the L2s (option sweep): the workgroups of an XCD walk ONE window of rows together, so that a row fetched
    // as somebody's neighbour is still in the L2 when its owner reaches it (msdp_sweep_rows; round 3 measured 1.32 x the
    // algorithmic bytes at n = 10^6 with the chunked order -- the far grid neighbours' rows were fetched twice).  Round 4:
    // the window alone left 1.30 x (PMC); with streaming (nt) accesses for what the launch touches once -- Y, the output --
    // the L2 keeps the gathered rows and the launch moves 1.00 x the algorithmic bytes (0.841 GB at n = 10^6, p = 32), and with
    // two 64-row steps per workgroup and window 242 -> 215 us.  What is left is not traffic: a software-pipelined form of this
    // loop (next step's loads in flight during the gather wait) was SLOWER (263 us), four steps per window and four row steps in
    // flight per wave change nothing -- the time follows the request count through the L2 (1.9 KB per row against 0.84 KB
    // from HBM), not the bytes from HBM.
    int stride = MSDP_WAVES * RPW;
    const int K = d.sweep ? ((d.sweep >> 4) & 15) + 1 : 1;       // 64-row steps per block of the windowed traversal
    if (d.sweep) msdp_sweep_rows(d.n_loc, d.G, K * MSDP_WAVES * RPW, lo, hi, stride);
    const bool nt = (d.sweep & 2) != 0;
    // two row steps per trip of the loop: the loads of both are issued before either is consumed (the loop is a chain of two
    // dependent round trips -- (col, val) -> neighbour rows -- and 16 waves per CU do not cover it at HBM latency)
    constexpr int UN = 2;
    auto step_row = [&](int t) { return lo + (K == 1 ? t * stride : (t / K) * stride + (t % K) * (MSDP_WAVES * RPW)) + wave * RPW; };
    for (int t = 0; step_row(t) < hi; t += UN) {
        double2 acc[UN][NCH], y[UN][NCH], u[UN][NCH];
        double eg[UN];
        int rows[UN];
        bool rok[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int row = step_row(t + q) + rsub;
            rok[q] = row < hi;
            rows[q] = rok[q] ? row : lo;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[q][ch] = make_double2(0.0, 0.0);
                const int col = 2 * sub + ch * 2 * LPR;
                const bool ok = col < d.ld;
                const int64_t o = (int64_t)rows[q] * d.ld + (ok ? col : 0);
                y[q][ch] = nt ? ld2_nt(Yl + o) : ld2(Yl + o);
                u[q][ch] = ld2(Ul + o);
                if (!ok) { y[q][ch] = make_double2(0.0, 0.0); u[q][ch] = make_double2(0.0, 0.0); }
            }
            eg[q] = eG[rows[q]];
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) spmm_row<LPR, NCH, ELL>(d, rows[q], sub, Uf, acc[q]);
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) dot += acc[q][ch].x * y[q][ch].x + acc[q][ch].y * y[q][ch].y;
            dot = msdp_group_sum<LPR>(dot);          // sum(Y.*eH)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld && rok[q]) {
                    double2 h;
                    h.x = acc[q][ch].x - y[q][ch].x * dot - u[q][ch].x * eg[q];
                    h.y = acc[q][ch].y - y[q][ch].y * dot - u[q][ch].y * eg[q];
                    if (nt) st2_nt(H + (int64_t)rows[q] * d.ld + col, h); else st2(H + (int64_t)rows[q] * d.ld + col, h);
                    pd += u[q][ch].x * h.x + u[q][ch].y * h.y;
                }
            }
        }
    }
    msdp_put_partial(d.P, P_DHD, pd, sh);
}

template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_costgrad_sparse_obl(Dev d, int slot) { costgrad_sparse_obl_body<LPR, NCH, false>(d, slot); }
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_costgrad_ell_obl(Dev d, int slot) { costgrad_sparse_obl_body<LPR, NCH, true>(d, slot); }
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_hess_sparse_obl(Dev d) { hess_sparse_obl_body<LPR, NCH, false>(d); }
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_hess_ell_obl(Dev d) { hess_sparse_obl_body<LPR, NCH, true>(d); }

// ------------------------------------------------------------------ tCG kernels
// tCG.m:102-157: eta=0, Heta=0, r=grad, mdelta=r, scalars.
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg_init(Dev d) {
    const Ctl* c = d.ctl;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int act = c->done ? 0 : 1;
        frame_store(&d.F[0], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 0, 1);
        frame_store(&d.F[1], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 0, 1);
        d.ctl->tcg_running = act;
        msdp_publish(d, c->k, 0, act);
    }
    if (c->done) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ g = c->cur ? d.Gr[1] : d.Gr[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    const double2 z = make_double2(0.0, 0.0);
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 gv = ld2(g + i);
        st2(d.r + i, gv);
        st2(d.md + i, gv);
        st2(d.eta[0] + i, z);
        st2(d.Heta[0] + i, z);
    }
}

// tCG.m:166-241 (everything between the Hess-vec and the residual norm).
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg_upd1(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    __shared__ double shb[2];
    const Frame* fi = &d.F[0];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha0 = fi->alpha;
    const int j = fi->j, stop0 = fi->stop, ix = fi->eta_idx, mi = fi->md_idx;
    if (!active) {
        if (lead) frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha0, beta0, 0, j, stop0, ix, mi, 0);
        return;
    }
    const double* __restrict__ mdp = mi ? d.md2 : d.md;
    const Ctl* c = d.ctl;
    const bool bench = c->bench_mode != 0;
    const double Delta = c->Delta;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ Heta = ix ? d.Heta[1] : d.Heta[0];
    double* __restrict__ neta = ix ? d.eta[0] : d.eta[1];
    double* __restrict__ nHeta = ix ? d.Heta[0] : d.Heta[1];
    const double* __restrict__ g = c->cur ? d.Gr[1] : d.Gr[0];
    const int64_t i0 = e0 + 2 * threadIdx.x;
    // Every vector load of the first two passes is issued BEFORE the partial sums are re-reduced: the
    // launch is a chain of dependent memory round trips (cold L2 at every kernel boundary), and this takes
    // {partials} -> {vectors, pass 1} -> {vectors, pass 2} down to a single round trip.
    constexpr int64_t STEP = 2 * MSDP_BLOCK;
    const double2 zz = make_double2(0.0, 0.0);
    double2 E[2], HE[2], M[2], HM[2], RR[2], GV[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t i = i0 + q * STEP;
        E[q] = HE[q] = M[q] = HM[q] = RR[q] = GV[q] = zz;
        if (i < e1) {
            E[q] = ld2(eta + i); HE[q] = ld2(Heta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i);
            RR[q] = ld2(d.r + i); GV[q] = ld2(g + i);
        }
    }
    const double d_Hd = msdp_sum_partials_block(d.P, P_DHD, d.G, shb);       // :166
    const double alpha = z_r / d_Hd;                                 // :170
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;  // :173
    if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {         // :183
        const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;  // :188
        for (int64_t ib = i0; ib < e1; ib += 2 * STEP) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t i = ib + q * STEP;
                if (i < e1) {
                    if (ib != i0) { E[q] = ld2(eta + i); HE[q] = ld2(Heta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i); }
                    st2(neta + i, make_double2(E[q].x - tau * M[q].x, E[q].y - tau * M[q].y));        // :192
                    st2(nHeta + i, make_double2(HE[q].x - tau * HM[q].x, HE[q].y - tau * HM[q].y));  // :198
                }
            }
        }
        if (lead) {
            frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha0, beta0, 0, j + 1,
                        (d_Hd <= 0.0) ? 1 : 2, ix ^ 1, mi, 0);
        }
        return;
    }
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t ib = i0; ib < e1; ib += 2 * STEP) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t i = ib + q * STEP;
            if (i < e1) {
                if (ib != i0) {
                    E[q] = ld2(eta + i); HE[q] = ld2(Heta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i);
                    RR[q] = ld2(d.r + i); GV[q] = ld2(g + i);
                }
                const double2 e = E[q], he = HE[q], m = M[q], hm = HM[q], rr = RR[q], gv = GV[q];
                const double2 ne = make_double2(e.x - alpha * m.x, e.y - alpha * m.y);         // :215
                const double2 nh = make_double2(he.x - alpha * hm.x, he.y - alpha * hm.y);     // :220
                const double2 nr = make_double2(rr.x - alpha * hm.x, rr.y - alpha * hm.y);     // :238
                st2(neta + i, ne);
                st2(nHeta + i, nh);
                st2(d.r + i, nr);
                s1 += ne.x * gv.x + ne.y * gv.y;      // <new_eta, grad>     :227
                s2 += ne.x * nh.x + ne.y * nh.y;      // <new_eta, new_Heta>
                s3 += nr.x * nr.x + nr.y * nr.y;      // r_r                 :241
            }
        }
    }
    msdp_put_partials3(d.P, P_S1, s1, P_S2, s2, P_S3, s3, sh);
    if (lead)   // :214
        frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe_new, model_value, norm_r0, alpha, beta0, 1, j, stop0, ix, mi, 0);
}

// tCG.m:227-287: model check, convergence test, new direction + tangent re-projection.
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg_upd2_obl(Dev d) {
    __shared__ double shb[4];
    const Frame* fi = &d.F[1];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha = fi->alpha;
    const int j0 = fi->j, stop0 = fi->stop, ix = fi->eta_idx;
    if (!active) {
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j0, stop0, ix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, d.ctl->k, j0, 0);
        }
        return;
    }
    const Ctl* c = d.ctl;
    const bool bench = c->bench_mode != 0;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const double* __restrict__ Yl = c->cur ? d.Y[1] : d.Y[0];
    // rows of the first pass (two rows per lane group when NCH == 1) are loaded before the partial sums
    // are re-reduced: one memory round trip instead of {partials} -> {rows, pass 1} -> {rows, pass 2}
    constexpr int R = (NCH == 1) ? 2 : 1;
    constexpr int RSTEP = MSDP_WAVES * RPW;
    const int rowf = lo + wave * RPW;
    double2 PR[R][NCH], PM[R][NCH], PY[R][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = rowf + r * RSTEP + rsub;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int col = 2 * sub + ch * 2 * LPR;
            PR[r][ch] = PM[r][ch] = PY[r][ch] = make_double2(0.0, 0.0);
            if (row < hi && col < d.ld) {
                const int64_t o = (int64_t)row * d.ld + col;
                PR[r][ch] = ld2(d.r + o); PM[r][ch] = ld2(d.md + o); PY[r][ch] = ld2(Yl + o);
            }
        }
    }
    double s1, s2, r_r;
    msdp_sum_partials3_block(d.P, P_S1, P_S2, P_S3, d.G, shb, s1, s2, r_r);
    const double new_model = s1 + 0.5 * s2;                 // :227
    const int j = j0 + 1;
    if (!bench && new_model >= model_value) {               // :228
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j, 6, ix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    const int nix = ix ^ 1;                                 // :233-235 commit new_eta/new_Heta
    const double norm_r = sqrt(r_r);
    const double nr0t = (c->theta == 1.0) ? norm_r0 : pow(norm_r0, c->theta);   // norm_r0^theta
    if (!bench && j >= c->mininner && norm_r <= norm_r0 * fmin(nr0t, c->kappa)) {   // :249
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, new_model, norm_r0, alpha, beta0, 0, j,
                        (c->kappa < nr0t) ? 3 : 4, nix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    if (j >= c->maxinner) {                                 // loop bound :160 (stop stays 5)
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, new_model, norm_r0, alpha, beta0, 0, j, stop0, nix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    const double beta = r_r / z_r;                          // :272
    if (lead) {
        frame_store(&d.F[0], r_r, r_r + beta * beta * d_Pd /* :287 */, beta * (e_Pd + alpha * d_Pd) /* :286 */, e_Pe,
                    new_model, norm_r0, alpha, beta, 1, j, stop0, nix);
        msdp_publish(d, c->k, j, 1);
    }
    // mdelta = tangent(z + beta*mdelta), z = r   :273,283
    for (int row0 = rowf; row0 < hi; row0 += R * RSTEP) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r * RSTEP + rsub;
            if (row < hi) {
                double2 v[NCH], y[NCH];
                double dot = 0.0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int col = 2 * sub + ch * 2 * LPR;
                    if (col < d.ld) {
                        const int64_t o = (int64_t)row * d.ld + col;
                        if (row0 != rowf) { PR[r][ch] = ld2(d.r + o); PM[r][ch] = ld2(d.md + o); PY[r][ch] = ld2(Yl + o); }
                        const double2 rr = PR[r][ch], m = PM[r][ch];
                        y[ch] = PY[r][ch];
                        v[ch] = make_double2(rr.x + beta * m.x, rr.y + beta * m.y);
                        dot += v[ch].x * y[ch].x + v[ch].y * y[ch].y;
                    } else { v[ch] = make_double2(0.0, 0.0); y[ch] = v[ch]; }
                }
                dot = msdp_group_sum<LPR>(dot);
                if (d.rowfree && d.rowfree[row]) dot = 0.0;           // Euclidean block: tangent = identity
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int col = 2 * sub + ch * 2 * LPR;
                    if (col < d.ld)
                        st2(d.md + (int64_t)row * d.ld + col,
                            make_double2(v[ch].x - y[ch].x * dot, v[ch].y - y[ch].y * dot));
                }
            }
        }
    }
}

// x_prop = retr(x, eta) (ManiSDP_onlyunitdiag.m:142-145) into the other slot, and the
// partial of <eta, grad + .5*Heta> (trustregions.m:549-550).
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_retract_obl(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    const Ctl* c = d.ctl;
    if (c->done) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const int cur = c->cur, ix = d.F[0].eta_idx;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ g = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ Heta = ix ? d.Heta[1] : d.Heta[0];
    double* __restrict__ Yp = cur ? d.Y[0] : d.Y[1];
    double prd = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 x[NCH];
            double nn = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    const double2 y = ld2(Yl + o), e = ld2(eta + o), he = ld2(Heta + o), gv = ld2(g + o);
                    x[ch] = make_double2(y.x + e.x, y.y + e.y);
                    nn += x[ch].x * x[ch].x + x[ch].y * x[ch].y;
                    prd += e.x * (gv.x + 0.5 * he.x) + e.y * (gv.y + 0.5 * he.y);
                } else x[ch] = make_double2(0.0, 0.0);
            }
            nn = sqrt(msdp_group_sum<LPR>(nn));
            if (d.rowfree && d.rowfree[row]) nn = 1.0;               // Euclidean block: retr(x, eta) = x + eta
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld)
                    st2(Yp + (int64_t)row * d.ld + col, make_double2(x[ch].x / nn, x[ch].y / nn));
            }
        }
    }
    msdp_put_partial(d.P, P_RD, prd, sh);
}

// ------------------------------------------------------------------ RTR scalars
// trustregions.m:405-409 after the first cost/grad.
__global__ __launch_bounds__(MSDP_BLOCK) void k_rtr_begin(Dev d) {
    const double f = msdp_sum_partials(d.P, P_F, d.G);
    const double gg = msdp_sum_partials(d.P, P_GG, d.G);
    if (threadIdx.x == 0) {
        Ctl* c = d.ctl;
        c->fx = f; c->gg = gg; c->norm_grad = sqrt(gg);
        c->Delta = c->Delta0;
        c->k = 0; c->hessvecs = 0; c->accepted = 0; c->rejected = 0; c->cost_evals = 1;
        c->last_stop_inner = 0;
        c->done = (c->norm_grad < c->tolgradnorm) || (c->k >= c->maxiter);   // stoppingcriterion.m:51-72
        c->tcg_running = 0;
    }
}

// trustregions.m:548-729: rho, radius update, accept/reject, stopping test.
__global__ __launch_bounds__(MSDP_BLOCK) void k_rtr_decide(Dev d) {
    if (d.ctl->done) return;
    const double fp = msdp_sum_partials(d.P, P_F, d.G);
    const double ggp = msdp_sum_partials(d.P, P_GG, d.G);
    const double rd = msdp_sum_partials(d.P, P_RD, d.G);
    if (threadIdx.x == 0) {
        Ctl* c = d.ctl;
        const int f_stop = d.F[0].stop, f_j = d.F[0].j;
        double rhonum = c->fx - fp;                                          // :548
        double rhoden = -rd;                                                 // :550
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

// ------------------------------------------------------------------ utilities
// Reference layout (row stride p) <-> device layout (row stride ld, zero pad).
__global__ void k_pack_rows(const double* __restrict__ src, double* __restrict__ dst, int n, int p, int ld) {
    const int64_t tot = (int64_t)n * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ld;
        const int col = (int)(i - row * ld);
        dst[i] = (col < p) ? src[row * p + col] : 0.0;
    }
}
__global__ void k_unpack_rows(const double* __restrict__ src, double* __restrict__ dst, int n, int p, int ld) {
    const int64_t tot = (int64_t)n * p;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / p;
        const int col = (int)(i - row * p);
        dst[i] = src[row * ld + col];
    }
}
// Same for the n x p column-major boundary layout of unittrace.
__global__ void k_pack_cols(const double* __restrict__ src, double* __restrict__ dst, int n, int p, int ld) {
    const int64_t tot = (int64_t)n * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ld;
        const int col = (int)(i - row * ld);
        dst[i] = (col < p) ? src[(int64_t)col * n + row] : 0.0;
    }
}
__global__ void k_unpack_cols(const double* __restrict__ src, double* __restrict__ dst, int n, int p, int ld) {
    const int64_t tot = (int64_t)n * p;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = i / n;
        const int64_t row = i - col * n;
        dst[i] = src[row * ld + col];
    }
}

// V = U - Y.*rowdot(Y,U) (M.proj, ManiSDP_onlyunitdiag.m:138); flat one-thread-per-row
// version for the test-only fine-grained entry point.
__global__ void k_proj_obl_simple(const double* __restrict__ Y, const double* __restrict__ U,
                                  double* __restrict__ V, int n, int ld, const unsigned char* __restrict__ rowfree) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
        double dot = 0.0;
        for (int c = 0; c < ld; ++c) dot += Y[(int64_t)row * ld + c] * U[(int64_t)row * ld + c];
        if (rowfree && rowfree[row]) dot = 0.0;
        for (int c = 0; c < ld; ++c) V[(int64_t)row * ld + c] = U[(int64_t)row * ld + c] - Y[(int64_t)row * ld + c] * dot;
    }
}
__global__ void k_retr_obl_simple(const double* __restrict__ Y, const double* __restrict__ U,
                                  double* __restrict__ Z, double alpha, int n, int ld, const unsigned char* __restrict__ rowfree) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
        double nn = 0.0;
        for (int c = 0; c < ld; ++c) {
            const double x = Y[(int64_t)row * ld + c] + alpha * U[(int64_t)row * ld + c];
            nn += x * x;
        }
        nn = (rowfree && rowfree[row]) ? 1.0 : sqrt(nn);
        for (int c = 0; c < ld; ++c)
            Z[(int64_t)row * ld + c] = (Y[(int64_t)row * ld + c] + alpha * U[(int64_t)row * ld + c]) / nn;
    }
}
__global__ void k_set_frame_active(Dev d, int active) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        d.F[0].active = active; d.F[1].active = active; d.ctl->done = 0;
    }
}
__global__ void k_sum_to(Dev d, int which, double* out) {
    const double s = msdp_sum_partials(d.P, which, d.G);
    if (threadIdx.x == 0) *out = s;
}

// ------------------------------------------------------------------ launchers
static inline void lpr_for(int ld, int& lpr, int& nch) {
    int half = ld / 2;
    lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    nch = (half + lpr - 1) / lpr;
    if (nch < 1) nch = 1;
}

// A workgroup walks its rows in passes of MSDP_WAVES*64/lpr rows.  When it owns between one and two passes' worth (G81 at
// p = 17..32 on 256 workgroups: 79 rows, 64 per pass) the second pass runs with most waves idle but costs a full dependent
// round of loads; half the lanes per row with two column chunks per lane puts all rows into ONE pass with twice the loads in
// flight per lane.
static inline void lpr_rebalance(msdp_handle h, int& lpr, int& nch) {
    // (k_hess_ell_obl on G81 at p = 32: 6.18 -> 5.58 us.)
    if (nch != 1 || (lpr != 16 && lpr != 32)) return;
    const int rows_wg = (h->d.n_loc + h->d.G - 1) / h->d.G, per_pass = MSDP_WAVES * (64 / lpr);
    if (rows_wg > per_pass && rows_wg <= 2 * per_pass) { lpr /= 2; nch = 2; }
}

#define DISPATCH_LPR(KERNEL, h, ...)                                                                 \
    do {                                                                                             \
        int lpr, nch;                                                                                \
        lpr_for((h)->d.ld, lpr, nch);                                                                \
        lpr_rebalance((h), lpr, nch);                                                                \
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
        } else if (nch == 2 && lpr == 8) {                                                           \
            hipLaunchKernelGGL((KERNEL<8, 2>), grid, block, 0, (h)->stream, __VA_ARGS__);            \
        } else if (nch == 2 && lpr == 16) {                                                          \
            hipLaunchKernelGGL((KERNEL<16, 2>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else if (nch == 4 && lpr == 4) {                                                           \
            hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, (h)->stream, __VA_ARGS__);            \
        } else if (nch == 2 && lpr == 64) {                                                          \
            hipLaunchKernelGGL((KERNEL<64, 2>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else if (nch <= 4 && lpr == 64) {                                                          \
            hipLaunchKernelGGL((KERNEL<64, 4>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else if (nch <= 8 && lpr == 64) {                                                          \
            hipLaunchKernelGGL((KERNEL<64, 8>), grid, block, 0, (h)->stream, __VA_ARGS__);           \
        } else {                                                                                     \
            msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", (h)->d.p);    \
            return MSDP_EUNSUPPORTED;                                                                \
        }                                                                                            \
    } while (0)

int msdp_window_eligible(msdp_handle h);            // msdp_window.hip
int msdp_window_hess(msdp_handle h);
int msdp_dense_costgrad(msdp_handle h, int slot);   // msdp_dense.hip
int msdp_dense_hess(msdp_handle h);
int msdp_affine_costgrad(msdp_handle h, int slot);  // msdp_affine.hip
int msdp_affine_hess(msdp_handle h);
int msdp_sphere_upd2(msdp_handle h);
int msdp_sphere_retract(msdp_handle h);

int msdp_launch_costgrad(msdp_handle h, int slot) {
    // gather source: all rows of Y[slot] (slot >= 2: resolved on the device, single rank, sparse C only)
    int rc = 0;
    if (slot >= 2) {
        if (h->d.costkind != COST_SPARSE || h->use_comm) { msdp_set_error("relative slot: sparse single-rank only"); return MSDP_EINVAL; }
    } else {
        rc = (h->d.costkind == COST_SPARSE) ? msdp_exchange_rows(h, h->d.Y[slot]) : msdp_allgather_rows(h, h->d.Y[slot]);
    }
    if (rc) return rc;
    if (h->d.costkind == COST_SPARSE) {
        if (h->d.ellW > 0) DISPATCH_LPR(k_costgrad_ell_obl, h, h->d, slot);
        else DISPATCH_LPR(k_costgrad_sparse_obl, h, h->d, slot);
    } else if (h->d.costkind == COST_DENSE) {
        rc = msdp_dense_costgrad(h, slot);
        if (rc) return rc;
    } else {
        rc = msdp_affine_costgrad(h, slot);
        if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    return msdp_allreduce_partials(h, P_F, 2);
}

int msdp_launch_hess(msdp_handle h) {
    int rc = (h->d.costkind == COST_SPARSE) ? msdp_exchange_rows(h, h->d.md) : msdp_allgather_rows(h, h->d.md);
    if (rc) return rc;
    if (h->d.costkind == COST_SPARSE) {
        if (msdp_window_eligible(h)) { if ((rc = msdp_window_hess(h))) return rc; }        // gathered rows staged in LDS (msdp_window.hip)
        else if (h->d.ellW > 0) DISPATCH_LPR(k_hess_ell_obl, h, h->d);
        else DISPATCH_LPR(k_hess_sparse_obl, h, h->d);
    } else if (h->d.costkind == COST_DENSE) {
        rc = msdp_dense_hess(h);
        if (rc) return rc;
    } else {
        rc = msdp_affine_hess(h);
        if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    return msdp_allreduce_partials(h, P_DHD, 1);
}

int msdp_launch_tcg_init(msdp_handle h) {
    hipLaunchKernelGGL(k_tcg_init, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_launch_upd1(msdp_handle h) {
    hipLaunchKernelGGL(k_tcg_upd1, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return msdp_allreduce_partials(h, P_S1, 3);
}

int msdp_launch_upd2(msdp_handle h) {
    if (h->d.manifold == MANI_OBLIQUE) {
        DISPATCH_LPR(k_tcg_upd2_obl, h, h->d);
        HIPCHK(hipGetLastError());
        return 0;
    }
    return msdp_sphere_upd2(h);
}

int msdp_launch_retract(msdp_handle h) {
    if (h->d.manifold == MANI_OBLIQUE) {
        DISPATCH_LPR(k_retract_obl, h, h->d);
        HIPCHK(hipGetLastError());
        return msdp_allreduce_partials(h, P_RD, 1);
    }
    return msdp_sphere_retract(h);
}

int msdp_launch_rtr_begin(msdp_handle h) {
    hipLaunchKernelGGL(k_rtr_begin, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_launch_rtr_decide(msdp_handle h) {
    hipLaunchKernelGGL(k_rtr_decide, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- the factor's p x p Gram matrix, rank cut and widening on the device (svd(Y), Y = V(:,1:r)'.*e(1:r),
// Y = [Y; alpha*vS'], Y = Y./sqrt(sum(Y.^2)) of ManiSDP_onlyunitdiag.m:52-54,70-83 without the factor leaving the GPU)
__global__ __launch_bounds__(256) void k_fgram_part(int n_loc, int ld, const double* __restrict__ Y, double* __restrict__ part) {
    const int rows = (n_loc + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows, r1 = min(n_loc, r0 + rows);
    for (int e = threadIdx.x; e < ld * ld; e += blockDim.x) {
        const int a = e / ld, b = e - a * ld;
        double acc = 0.0;
        for (int k = r0; k < r1; ++k) acc = fma(Y[(int64_t)k * ld + a], Y[(int64_t)k * ld + b], acc);
        part[(int64_t)blockIdx.x * ld * ld + e] = acc;
    }
}
__global__ void k_fgram_sum(int ld, int nblk, const double* __restrict__ part, double* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ld * ld; e += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int q = 0; q < nblk; ++q) acc += part[(int64_t)q * ld * ld + e];
        out[e] = acc;
    }
}
// Ynew (cap x ldn) = Y (n_loc x ld, first p columns) * Q (p x r row-major); pad rows / columns zero
__global__ void k_frotate(int cap, int n_loc, int ld, int p, int ldn, int r, const double* __restrict__ Y,
                          const double* __restrict__ Q, double* __restrict__ Yn) {
    const int64_t tot = (int64_t)cap * ldn;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / ldn; const int col = (int)(e - row * ldn);
        double acc = 0.0;
        if (row < n_loc && col < r)
            for (int a = 0; a < p; ++a) acc = fma(Y[row * ld + a], Q[a * r + col], acc);
        Yn[e] = acc;
    }
}
// Ynew = [Y, alpha*V] (V: n_loc x k column-major), rows scaled to unit norm when normalize != 0 (rows flagged free keep theirs)
__global__ void k_fappend(int cap, int n_loc, int ld, int p, int ldn, int k, const double* __restrict__ Y, const double* __restrict__ V,
                          double alpha, int normalize, const unsigned char* __restrict__ rowfree, double* __restrict__ Yn) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < cap; row += gridDim.x * blockDim.x) {
        double* yn = Yn + (int64_t)row * ldn;
        if (row >= n_loc) { for (int c = 0; c < ldn; ++c) yn[c] = 0.0; continue; }
        const double* y = Y + (int64_t)row * ld;
        double nn = 0.0;
        for (int c = 0; c < p; ++c) nn = fma(y[c], y[c], nn);
        for (int c = 0; c < k; ++c) { const double v = alpha * V[(int64_t)c * n_loc + row]; nn = fma(v, v, nn); }
        const double sc = (normalize && !(rowfree && rowfree[row]) && nn > 0.0) ? 1.0 / sqrt(nn) : 1.0;
        for (int c = 0; c < p; ++c) yn[c] = y[c] * sc;
        for (int c = 0; c < k; ++c) yn[p + c] = alpha * V[(int64_t)c * n_loc + row] * sc;
        for (int c = p + k; c < ldn; ++c) yn[c] = 0.0;
    }
}
int msdp_k_fgram(msdp_handle h, const double* Y, double* part, int nblk, double* out) {
    hipLaunchKernelGGL(k_fgram_part, dim3(nblk), dim3(256), 0, h->stream, h->d.n_loc, h->d.ld, Y, part);
    hipLaunchKernelGGL(k_fgram_sum, dim3((h->d.ld * h->d.ld + 255) / 256), dim3(256), 0, h->stream, h->d.ld, nblk, (const double*)part, out);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_frotate(msdp_handle h, int cap, int r, int ldn, const double* Y, const double* Q, double* Yn) {
    hipLaunchKernelGGL(k_frotate, dim3(1024), dim3(256), 0, h->stream, cap, h->d.n_loc, h->d.ld, h->d.p, ldn, r, Y, Q, Yn);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_fappend(msdp_handle h, int cap, int k, int ldn, const double* Y, const double* V, double alpha, int normalize, double* Yn) {
    hipLaunchKernelGGL(k_fappend, dim3((cap + 255) / 256), dim3(256), 0, h->stream, cap, h->d.n_loc, h->d.ld, h->d.p, ldn, k, Y, V, alpha,
                       normalize, h->d.rowfree, Yn);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- small launch wrappers used by the API unit ----
int msdp_k_pack(msdp_handle h, const double* src, double* dst, int n, int p, int ld, bool colmajor) {
    const int grid = 1024;
    if (colmajor) hipLaunchKernelGGL(k_pack_cols, dim3(grid), dim3(256), 0, h->stream, src, dst, n, p, ld);
    else hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, h->stream, src, dst, n, p, ld);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_unpack(msdp_handle h, const double* src, double* dst, int n, int p, int ld, bool colmajor) {
    const int grid = 1024;
    if (colmajor) hipLaunchKernelGGL(k_unpack_cols, dim3(grid), dim3(256), 0, h->stream, src, dst, n, p, ld);
    else hipLaunchKernelGGL(k_unpack_rows, dim3(grid), dim3(256), 0, h->stream, src, dst, n, p, ld);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_proj_obl(msdp_handle h, const double* Y, const double* U, double* V) {
    hipLaunchKernelGGL(k_proj_obl_simple, dim3(256), dim3(256), 0, h->stream, Y, U, V, h->d.n_loc, h->d.ld, h->d.rowfree);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_retr_obl(msdp_handle h, const double* Y, const double* U, double* Z, double alpha) {
    hipLaunchKernelGGL(k_retr_obl_simple, dim3(256), dim3(256), 0, h->stream, Y, U, Z, alpha, h->d.n_loc, h->d.ld, h->d.rowfree);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_set_active(msdp_handle h, int active) {
    hipLaunchKernelGGL(k_set_frame_active, dim3(1), dim3(64), 0, h->stream, h->d, active);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_k_sum_to(msdp_handle h, int which, double* out) {
    hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, h->d, which, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_k_sum_to_fwd(msdp_handle h, int which, double* out) { return msdp_k_sum_to(h, which, out); }
