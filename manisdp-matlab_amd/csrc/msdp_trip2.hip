// msdp_trip2.hip -- two-launch tCG trip of the chunked path (sparse C, oblique manifold, one rank).
//
// The three-launch trip (k_hess_* / k_tcg_upd1 / k_tcg_upd2_obl, msdp_kernels.hip) streams 17 vectors per trip through
// HBM: measured 4.44 GB per trip at n = 10^6, p = 32 against 0.84 GB for the Hess-vec alone
// (profiles/r2_pmc_chunked_n1e6_p32.json) -- the path every problem beyond the persistent kernel's reach takes.  The two
// synchronisation points of Manopt's CG arithmetic per trip (d_Hd -> alpha; model value and r_r -> beta) cannot be merged
// without changing its decisions, so a trip is two launches, and what crosses a launch boundary crosses HBM:
//
//   k_tcg2_upd  (tCG.m:166-241)  reads eta, mdelta, Hmdelta, r, grad; writes eta', r'            7 vectors
//   k_tcg2_head (tCG.m:227-287 of trip j, then tCG.m:163 of trip j+1)
//                                 reads r', mdelta, Y; writes mdelta', Hmdelta'                    5 vectors
//
// 12 vector passes instead of 17:
//   * Heta is not a stored vector: tCG.m:220,238 update Heta and r with the same -alpha*Hmdelta, so Heta = r - grad
//     (the identity msdp_persist.hip uses); the step's Heta is written once, when the tCG ends;
//   * the new direction mdelta' = tangent(r' + beta*mdelta) (tCG.m:273,283) is row-local: the head kernel forms it for its own
//     rows (and stores them) AND recomputes it for the neighbour rows its S*U gathers, from (r', mdelta, Y) with the same
//     operations in the same order -- bit-identical to the stored rows -- so no launch boundary separates "new direction"
//     from "S*U".  mdelta ping-pongs between two buffers (the neighbours still read the old rows).
// eta and r ping-pong too (tCG.m:228: a trial step whose model value went up is dropped and the OLD eta, Heta returned).
// The frames keep the convention of the three-launch trip: upd reads F[0] and writes F[1], head reads F[1] and writes F[0].
#include "msdp_device.h"
#include <math.h>

#define T2_ELL_MAXW 8

// direction of row k at my columns: fresh -> the gradient row; else tangent(r'[k] + beta*mdelta[k]) (tCG.m:273,283)
template <int LPR, int NCH>
__device__ __forceinline__ void t2_dir_row(const Dev& d, bool fresh, double beta, int k, int sub, const double* __restrict__ src0,
                                           const double* __restrict__ mdo, const double* __restrict__ Yl, double2 (&u)[NCH]) {
    if (fresh) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int col = 2 * sub + ch * 2 * LPR;
            u[ch] = (col < d.ld) ? ld2(src0 + (int64_t)k * d.ld + col) : make_double2(0.0, 0.0);
        }
        return;
    }
    double2 y[NCH];
    double dot = 0.0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int col = 2 * sub + ch * 2 * LPR;
        u[ch] = make_double2(0.0, 0.0); y[ch] = u[ch];
        if (col < d.ld) {
            const int64_t o = (int64_t)k * d.ld + col;
            const double2 rr = ld2(src0 + o), m = ld2(mdo + o);
            y[ch] = ld2(Yl + o);
            u[ch] = make_double2(rr.x + beta * m.x, rr.y + beta * m.y);
            dot += u[ch].x * y[ch].x + u[ch].y * y[ch].y;
        }
    }
    dot = msdp_group_sum<LPR>(dot);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { u[ch].x -= y[ch].x * dot; u[ch].y -= y[ch].y * dot; }
}

// eta = 0, r = grad into buffer 0; both frames = the start of a tCG (tCG.m:102-157).  The first direction (= grad) is not
// copied: the first head launch reads the gradient rows in its `fresh` mode.
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg2_init(Dev d) {
    const Ctl* c = d.ctl;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int act = c->done ? 0 : 1;
        frame_store(&d.F[0], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 1, 1);
        frame_store(&d.F[1], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 1, 1);
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
        st2(d.r + i, ld2(g + i));
        st2(d.eta[0] + i, z);
    }
}

// Second half of trip j (tCG.m:227-287: model check, stop tests, beta, new direction) and the Hess-vec of trip j+1
// (tCG.m:163; ManiSDP_onlyunitdiag.m:127-130).  Reads F[1] (written by k_tcg2_upd, or by k_tcg2_init: fresh), writes F[0].
template <int LPR, int NCH, bool ELL>
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg2_head(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    __shared__ double shb[4];
    const Frame* fi = &d.F[1];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha = fi->alpha;
    const int j0 = fi->j, stop0 = fi->stop, ix = fi->eta_idx, mi = fi->md_idx, fresh = fi->fresh;
    const Ctl* c = d.ctl;
    if (!active) {
        // the tCG ended earlier (in k_tcg2_upd: negative curvature / boundary, or in an earlier head launch): hand the final
        // frame on and do nothing
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j0, stop0, ix, mi, 0);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j0, 0);
        }
        return;
    }
    const bool bench = c->bench_mode != 0;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const int cur = c->cur;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ g = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eG = cur ? d.eG[1] : d.eG[0];
    int nix = ix, j = j0;
    double beta = 0.0, new_model = model_value;
    if (!fresh) {
        double s1, s2, r_r;
        msdp_sum_partials3_block(d.P, P_S1, P_S2, P_S3, d.G, shb, s1, s2, r_r);
        new_model = s1 + 0.5 * s2;                          // :227
        j = j0 + 1;
        int fin = 0, fstop = stop0, fix = ix;
        double fmodel = model_value;
        if (!bench && new_model >= model_value) { fin = 1; fstop = 6; }                     // :228 (the old eta, Heta stay)
        else {
            nix = ix ^ 1;                                   // :233-235 commit new_eta / new_Heta
            const double norm_r = sqrt(r_r);
            const double nr0t = (c->theta == 1.0) ? norm_r0 : pow(norm_r0, c->theta);
            if (!bench && j >= c->mininner && norm_r <= norm_r0 * fmin(nr0t, c->kappa)) {   // :249
                fin = 1; fstop = (c->kappa < nr0t) ? 3 : 4; fix = nix; fmodel = new_model;
            } else if (j >= c->maxinner) {                  // loop bound :160 (stop stays 5)
                fin = 1; fix = nix; fmodel = new_model;
            }
        }
        if (fin) {
            // the step's Heta = r - grad (tCG.m:220,238), written once, here
            const double* __restrict__ rf = fix ? d.r2 : d.r;
            double* __restrict__ Hout = fix ? d.Heta[1] : d.Heta[0];
            const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
            for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
                const double2 rr = ld2(rf + i), gv = ld2(g + i);
                st2(Hout + i, make_double2(rr.x - gv.x, rr.y - gv.y));
            }
            if (lead) {
                frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, fmodel, norm_r0, alpha, beta0, 0, j, fstop, fix, mi, 0);
                d.ctl->tcg_running = 0;
                msdp_publish(d, c->k, j, 0);
            }
            return;
        }
        beta = r_r / z_r;                                   // :272
        if (lead) {
            frame_store(&d.F[0], r_r, r_r + beta * beta * d_Pd /* :287 */, beta * (e_Pd + alpha * d_Pd) /* :286 */, e_Pe,
                        new_model, norm_r0, alpha, beta, 1, j, stop0, nix, mi ^ 1, 0);
            msdp_publish(d, c->k, j, 1);
        }
    } else if (lead) {
        frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 1, j0, stop0, ix, mi ^ 1, 0);
    }
    // ---- new direction of my rows (stored) and of the rows my S*U gathers (recomputed), then the Hess-vec
    const double* __restrict__ src0 = fresh ? g : (nix ? d.r2 : d.r);      // fresh: direction = gradient
    const double* __restrict__ mdo = mi ? d.md2 : d.md;                    // old direction (unused when fresh)
    double* __restrict__ mdn = mi ? d.md : d.md2;                          // new direction: the other buffer
    double* __restrict__ H = d.Hmd;
    double pd = 0.0;
    int stride = MSDP_WAVES * RPW;
    if (d.sweep) msdp_sweep_rows(d.n_loc, d.G, MSDP_WAVES * RPW, lo, hi, stride);     // (the streaming loops above keep the chunks)
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += stride) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 acc[NCH], y[NCH], u[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[ch] = make_double2(0.0, 0.0);
                const int col = 2 * sub + ch * 2 * LPR;
                y[ch] = (col < d.ld) ? ld2(Yl + (int64_t)row * d.ld + col) : make_double2(0.0, 0.0);
            }
            const double eg = eG[row];
            t2_dir_row<LPR, NCH>(d, fresh != 0, beta, row, sub, src0, mdo, Yl, u);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) st2(mdn + (int64_t)row * d.ld + col, u[ch]);
            }
            if (ELL) {
                int cw[T2_ELL_MAXW];
                double vw[T2_ELL_MAXW];
#pragma unroll
                for (int w = 0; w < T2_ELL_MAXW; ++w) {
                    const bool ok = w < d.ellW;
                    cw[w] = ok ? d.ellc[(int64_t)w * d.ell_stride + row] : row;
                    vw[w] = ok ? d.ellv[(int64_t)w * d.ell_stride + row] : 0.0;
                }
#pragma unroll
                for (int w = 0; w < T2_ELL_MAXW; ++w) {
                    if (w < d.ellW) {
                        double2 x[NCH];
                        if (cw[w] == row) {
#pragma unroll
                            for (int ch = 0; ch < NCH; ++ch) x[ch] = u[ch];
                        } else t2_dir_row<LPR, NCH>(d, fresh != 0, beta, cw[w], sub, src0, mdo, Yl, x);
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) { acc[ch].x = fma(vw[w], x[ch].x, acc[ch].x); acc[ch].y = fma(vw[w], x[ch].y, acc[ch].y); }
                    }
                }
            } else {
                const int start = d.rowptr[row], end = d.rowptr[row + 1];
                for (int k = start; k < end; ++k) {
                    const int cidx = d.colind[k];
                    const double v = d.cval[k];
                    double2 x[NCH];
                    if (cidx == row) {
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) x[ch] = u[ch];
                    } else t2_dir_row<LPR, NCH>(d, fresh != 0, beta, cidx, sub, src0, mdo, Yl, x);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) { acc[ch].x = fma(v, x[ch].x, acc[ch].x); acc[ch].y = fma(v, x[ch].y, acc[ch].y); }
                }
            }
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) dot += acc[ch].x * y[ch].x + acc[ch].y * y[ch].y;
            dot = msdp_group_sum<LPR>(dot);                 // sum(Y.*eH)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    double2 hq;
                    hq.x = acc[ch].x - y[ch].x * dot - u[ch].x * eg;
                    hq.y = acc[ch].y - y[ch].y * dot - u[ch].y * eg;
                    st2(H + (int64_t)row * d.ld + col, hq);
                    pd += u[ch].x * hq.x + u[ch].y * hq.y;
                }
            }
        }
    }
    msdp_put_partial(d.P, P_DHD, pd, sh);
}

// First half of a trip (tCG.m:166-241): alpha, negative-curvature / boundary exit, trial eta and r, their three inner
// products.  Reads F[0], writes F[1].  Heta is implied (r - grad); it is written only by the exit branch.
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg2_upd(Dev d) {
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
    const double* __restrict__ rold = ix ? d.r2 : d.r;
    double* __restrict__ neta = ix ? d.eta[0] : d.eta[1];
    double* __restrict__ rnew = ix ? d.r : d.r2;
    const double* __restrict__ g = c->cur ? d.Gr[1] : d.Gr[0];
    const int64_t i0 = e0 + 2 * threadIdx.x;
    // every vector load of the first two passes is issued before the partial sums are re-reduced (cf. k_tcg_upd1)
    constexpr int64_t STEP = 2 * MSDP_BLOCK;
    const double2 zz = make_double2(0.0, 0.0);
    double2 E[2], M[2], HM[2], RR[2], GV[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t i = i0 + q * STEP;
        E[q] = M[q] = HM[q] = RR[q] = GV[q] = zz;
        if (i < e1) { E[q] = ld2(eta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i); RR[q] = ld2(rold + i); GV[q] = ld2(g + i); }
    }
    const double d_Hd = msdp_sum_partials_block(d.P, P_DHD, d.G, shb);        // :166
    const double alpha = z_r / d_Hd;                                          // :170
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd; // :173
    if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {               // :183
        const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;   // :188
        double* __restrict__ Hout = ix ? d.Heta[0] : d.Heta[1];
        for (int64_t ib = i0; ib < e1; ib += 2 * STEP) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t i = ib + q * STEP;
                if (i < e1) {
                    if (ib != i0) { E[q] = ld2(eta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i); RR[q] = ld2(rold + i); GV[q] = ld2(g + i); }
                    st2(neta + i, make_double2(E[q].x - tau * M[q].x, E[q].y - tau * M[q].y));               // :192
                    // :198 Heta - tau*Hmdelta with Heta = r - grad
                    st2(Hout + i, make_double2((RR[q].x - tau * HM[q].x) - GV[q].x, (RR[q].y - tau * HM[q].y) - GV[q].y));
                }
            }
        }
        if (lead) {
            frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha0, beta0, 0, j + 1, (d_Hd <= 0.0) ? 1 : 2, ix ^ 1, mi, 0);
            d.ctl->tcg_running = 0;
        }
        return;
    }
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t ib = i0; ib < e1; ib += 2 * STEP) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t i = ib + q * STEP;
            if (i < e1) {
                if (ib != i0) { E[q] = ld2(eta + i); M[q] = ld2(mdp + i); HM[q] = ld2(d.Hmd + i); RR[q] = ld2(rold + i); GV[q] = ld2(g + i); }
                const double2 e = E[q], m = M[q], hm = HM[q], rr = RR[q], gv = GV[q];
                const double2 ne = make_double2(e.x - alpha * m.x, e.y - alpha * m.y);         // :215
                const double2 nr = make_double2(rr.x - alpha * hm.x, rr.y - alpha * hm.y);     // :238
                const double2 nh = make_double2(nr.x - gv.x, nr.y - gv.y);                     // new_Heta (:220)
                st2(neta + i, ne);
                st2(rnew + i, nr);
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

// ------------------------------------------------------------------ launchers
static inline void t2_lpr_for(int ld, int& lpr, int& nch) {
    int half = ld / 2;
    lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    nch = (half + lpr - 1) / lpr;
    if (nch < 1) nch = 1;
}

#define T2_LAUNCH(L, N)                                                                                          \
    do {                                                                                                         \
        if (h->d.ellW > 0) hipLaunchKernelGGL((k_tcg2_head<L, N, true>), grid, block, 0, h->stream, h->d);       \
        else hipLaunchKernelGGL((k_tcg2_head<L, N, false>), grid, block, 0, h->stream, h->d);                    \
    } while (0)

// Where it pays: the head kernel recomputes the direction of every gathered row from three vectors (15 row gathers per row on a
// grid graph instead of 5), which costs more than the five saved vector passes until the vectors are large -- measured
// (tools/archive/trip2_probe.py): G81 (n = 20000) p = 32: 23.6 against 23.9 us, p = 64: 49 against 45; n = 250 000, p = 64: 406 against 459;
// n = 10^6, p = 32: 858 against 940 us.  trip2 = 1 (default): from 2^21 vector entries on; 2: always (tests); 0: never.
int msdp_trip2_ok(msdp_handle h) {
    const Dev& d = h->d;
    if (!(h->tune.trip2 && d.costkind == COST_SPARSE && d.manifold == MANI_OBLIQUE && !h->use_comm && h->nranks == 1 && d.r2 && d.md2
          && !d.rowfree && d.ld <= 1024)) return 0;
    return h->tune.trip2 >= 2 || (int64_t)d.n_loc * d.ld >= ((int64_t)1 << 21);
}

int msdp_launch_trip2_init(msdp_handle h) {
    hipLaunchKernelGGL(k_tcg2_init, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_launch_trip2_head(msdp_handle h) {
    int lpr, nch;
    t2_lpr_for(h->d.ld, lpr, nch);
    const dim3 grid(h->d.G), block(MSDP_BLOCK);
    if (nch == 1) {
        switch (lpr) {
            case 1: T2_LAUNCH(1, 1); break;
            case 2: T2_LAUNCH(2, 1); break;
            case 4: T2_LAUNCH(4, 1); break;
            case 8: T2_LAUNCH(8, 1); break;
            case 16: T2_LAUNCH(16, 1); break;
            case 32: T2_LAUNCH(32, 1); break;
            default: T2_LAUNCH(64, 1); break;
        }
    } else if (nch == 2) T2_LAUNCH(64, 2);
    else if (nch <= 4) T2_LAUNCH(64, 4);
    else if (nch <= 8) T2_LAUNCH(64, 8);
    else { msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", h->d.p); return MSDP_EUNSUPPORTED; }
    HIPCHK(hipGetLastError());
    return 0;
}

int msdp_launch_trip2_upd(msdp_handle h) {
    hipLaunchKernelGGL(k_tcg2_upd, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}
