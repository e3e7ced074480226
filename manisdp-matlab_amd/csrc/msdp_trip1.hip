// msdp_trip1.hip -- row-sharded tCG trip with ONE exchange and ONE all-reduce (sparse C, oblique manifold, any communicator).
//
// The three-launch trip of the sharded path (msdp_kernels.hip) issues, per trip, the exchange of the direction rows in front of
// S*U and two all-reduces: <mdelta, H mdelta> behind the Hess-vec (tCG.m:166) and the three sums of tCG.m:227,241 behind the
// update.  Both reductions are needed -- alpha depends on the first, beta on the second -- but only one has to be a collective
// of its own: the second rides in the exchange.
//
//   k_tcg1_upd   (tCG.m:166-241)  alpha, exits, trial eta and r' (kept projected: r' = tangent(r - alpha*H mdelta)), this rank's
//                                  three sums (<eta',grad>, <eta',Heta'>, <r',r'>)
//   exchange                       rows of r' (all-gather or halo) + every rank's three sums, ONE grouped collective
//   k_tcg1_head  (tCG.m:227-287, then tCG.m:163 of the next trip)
//                                  adds the N triples in rank order (every rank: same bits, same decisions), model check, stop
//                                  tests, beta, mdelta' = tangent(r' + beta*mdelta) for its own rows, and the Hess-vec by
//                                  linearity (the trick of the persistent kernel, msdp_persist.hip:107-123, applied to the
//                                  Hessian itself -- a linear map on the tangent space):
//                                      H mdelta' = H r' + beta * H mdelta          (r', mdelta tangent)
//                                  with H mdelta of its rows from the previous trip; the partial <mdelta', H mdelta'>
//   all-reduce                     of those partials
//
// What linearity leaves out -- the re-projection of the old direction, 1e-16 per trip -- is reset every `persist_refresh`-th trip
// (default 32, the persistent kernel's schedule): that trip exchanges the rows of mdelta' themselves once more and multiplies
// directly (k_tcg1_head in its direct mode), as does the first trip of a tCG (direction = gradient).  The schedule depends on the
// trip count only, which the host knows: every rank issues the same collectives.
//
// eta and r ping-pong as in msdp_trip2.hip (tCG.m:228: a trial step whose model value went up is dropped); Heta is not
// stored (Heta = r - grad, tCG.m:220,238), the step's Heta is written when the tCG ends.
//
// Vector passes per trip: upd reads eta, mdelta, H mdelta, r, grad, Y and writes eta', r' (8); head reads r', mdelta, H mdelta, Y
// and writes mdelta', H mdelta' (6), its gather reads ONE vector (r') -- 14 passes, against 17 of the three-launch trip and 12 of
// the two-launch trip of msdp_trip2.hip, whose head gathers three vectors and is bound by the L2s instead (17 row loads per row:
// 530 us of its 830 at n = 10^6, p = 32).  With option trip1 = 2 a single rank without communicator runs this trip as well
// (the exchange is a no-op: the kernels read r' and the sums in place).
#include "msdp_device.h"
#include <math.h>

// eta = 0, r = grad, mdelta = grad (tCG.m:102-157); the first head launch multiplies directly
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg1_init(Dev d) {
    const Ctl* c = d.ctl;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int act = c->done ? 0 : 1;
        frame_store(&d.F[0], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 0, 0);
        frame_store(&d.F[1], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, act, 0, 5, 0, 0, 0);
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
    }
}

// First half of a trip.  Reads F[0] (and the all-reduced <mdelta, H mdelta>), writes F[1], the exchange rows and d.xs[0..2].
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg1_upd(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    __shared__ double shb[4];
    __shared__ int last;
    const Frame* fi = &d.F[0];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha0 = fi->alpha;
    const int j = fi->j, stop0 = fi->stop, ix = fi->eta_idx;
    if (!active) {
        if (lead) frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha0, beta0, 0, j, stop0, ix, 0, 0);
        return;
    }
    const Ctl* c = d.ctl;
    const bool bench = c->bench_mode != 0;
    const double Delta = c->Delta;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ rold = ix ? d.r2 : d.r;
    double* __restrict__ neta = ix ? d.eta[0] : d.eta[1];
    double* __restrict__ rnew = ix ? d.r : d.r2;
    const double* __restrict__ g = c->cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ Yl = c->cur ? d.Y[1] : d.Y[0];
    const double d_Hd = msdp_sum_partials_block(d.P, P_DHD, d.G, shb);        // :166 (all-reduced)
    const double alpha = z_r / d_Hd;                                          // :170
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd; // :173
    if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {               // :183
        const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;   // :188
        double* __restrict__ Hout = ix ? d.Heta[0] : d.Heta[1];
        const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
        for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
            const double2 e = ld2(eta + i), m = ld2(d.md + i), hm = ld2(d.Hmd + i), rr = ld2(rold + i), gv = ld2(g + i);
            st2(neta + i, make_double2(e.x - tau * m.x, e.y - tau * m.y));                               // :192
            st2(Hout + i, make_double2((rr.x - tau * hm.x) - gv.x, (rr.y - tau * hm.y) - gv.y));         // :198, Heta = r - grad
        }
        if (lead) {
            frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha0, beta0, 0, j + 1, (d_Hd <= 0.0) ? 1 : 2, ix ^ 1, 0, 0);
            d.ctl->tcg_running = 0;
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 nr[NCH], y[NCH], ne[NCH], gv[NCH];
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                nr[ch] = make_double2(0.0, 0.0); y[ch] = nr[ch]; ne[ch] = nr[ch]; gv[ch] = nr[ch];
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    const double2 e = ld2(eta + o), m = ld2(d.md + o), hm = ld2(d.Hmd + o), rr = ld2(rold + o);
                    gv[ch] = ld2(g + o);
                    y[ch] = ld2(Yl + o);
                    ne[ch] = make_double2(e.x - alpha * m.x, e.y - alpha * m.y);                   // :215
                    nr[ch] = make_double2(rr.x - alpha * hm.x, rr.y - alpha * hm.y);               // :238
                    dot += nr[ch].x * y[ch].x + nr[ch].y * y[ch].y;
                }
            }
            dot = msdp_group_sum<LPR>(dot);
            // the residual is kept PROJECTED: r and H*mdelta are tangent, so this removes rounding only (1e-16 |r| per trip) -- but
            // the neighbours' products are assembled from these rows by linearity, where a normal component would stay for good
            // (msdp_persist.hip:117-123 publishes a projected copy; here the row is the exchange buffer)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    nr[ch].x -= y[ch].x * dot; nr[ch].y -= y[ch].y * dot;
                    const double2 nh = make_double2(nr[ch].x - gv[ch].x, nr[ch].y - gv[ch].y);     // new_Heta (:220) = r' - grad
                    st2(neta + o, ne[ch]);
                    st2(rnew + o, nr[ch]);
                    s1 += ne[ch].x * gv[ch].x + ne[ch].y * gv[ch].y;      // <new_eta, grad>     :227
                    s2 += ne[ch].x * nh.x + ne[ch].y * nh.y;              // <new_eta, new_Heta>
                    s3 += nr[ch].x * nr[ch].x + nr[ch].y * nr[ch].y;      // r_r                 :241
                }
            }
        }
    }
    if (lead)   // :214
        frame_store(&d.F[1], z_r, d_Pd, e_Pd, e_Pe_new, model_value, norm_r0, alpha, beta0, 1, j, stop0, ix, 0, 0);
    // This rank's three sums for the exchange: the workgroup that arrives last adds the G partials in index order (the same
    // order whichever workgroup it is).  The partials cross XCDs inside the launch: agent-coherent (sc1) stores and loads and a
    // wait for the stores in front of the arrival count -- an agent-scope fence would write the XCD's whole L2 back (the
    // three vectors just stored): 60 us per launch, measured.
    s1 = msdp_wave_sum(s1); s2 = msdp_wave_sum(s2); s3 = msdp_wave_sum(s3);
    if (lane == 0) { sh[wave] = s1; sh[MSDP_WAVES + wave] = s2; sh[2 * MSDP_WAVES + wave] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int i = 0; i < MSDP_WAVES; ++i) s += sh[threadIdx.x * MSDP_WAVES + i];
        __hip_atomic_store(d.P + (P_S1 + threadIdx.x) * MSDP_MAX_GRID + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(d.xcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (last && threadIdx.x < 64) {
        double a = 0.0, b = 0.0, cc = 0.0;
        for (int i = threadIdx.x; i < d.G; i += 64) {
            a += __hip_atomic_load(d.P + P_S1 * MSDP_MAX_GRID + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b += __hip_atomic_load(d.P + P_S2 * MSDP_MAX_GRID + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cc += __hip_atomic_load(d.P + P_S3 * MSDP_MAX_GRID + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a = msdp_wave_sum(a); b = msdp_wave_sum(b); cc = msdp_wave_sum(cc);
        if (threadIdx.x == 0) { d.xs[0] = a; d.xs[1] = b; d.xs[2] = cc; d.xs[3] = 0.0; *d.xcount = 0u; }
    }
}

// Second half of trip j and the Hess-vec of trip j+1.
//   DIRECT = false: reads F[1] and the gathered sums, decides, writes F[0]; gathered rows = tangent(r'); product by linearity.
//   DIRECT = true : no decisions (first trip of a tCG, or right behind a linear launch on a refresh trip): reads F[0]; gathered
//                   rows = the direction itself; C*mdelta, H mdelta and the partial sums are computed afresh.
template <int LPR, int NCH, bool ELL, bool DIRECT>
__global__ __launch_bounds__(MSDP_BLOCK) void k_tcg1_head(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    const Frame* fi = DIRECT ? &d.F[0] : &d.F[1];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const Ctl* c = d.ctl;
    if (DIRECT) { if (!active) return; }
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha = fi->alpha;
    const int j0 = fi->j, stop0 = fi->stop, ix = fi->eta_idx;
    if (!DIRECT && !active) {
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j0, stop0, ix, 0, 0);
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
    int nix = ix;
    double beta = 0.0;
    if (!DIRECT) {
        double s1 = 0.0, s2 = 0.0, r_r = 0.0;
        for (int q = 0; q < d.xn; ++q) { s1 += d.xs_all[4 * q]; s2 += d.xs_all[4 * q + 1]; r_r += d.xs_all[4 * q + 2]; }   // rank order
        const double new_model = s1 + 0.5 * s2;             // :227
        const int j = j0 + 1;
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
            const double* __restrict__ rf = fix ? d.r2 : d.r;
            double* __restrict__ Hout = fix ? d.Heta[1] : d.Heta[0];
            const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
            for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
                const double2 rr = ld2(rf + i), gv = ld2(g + i);
                st2(Hout + i, make_double2(rr.x - gv.x, rr.y - gv.y));      // the step's Heta = r - grad (tCG.m:220,238)
            }
            if (lead) {
                frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, fmodel, norm_r0, alpha, beta0, 0, j, fstop, fix, 0, 0);
                d.ctl->tcg_running = 0;
                msdp_publish(d, c->k, j, 0);
            }
            return;
        }
        beta = r_r / z_r;                                   // :272
        if (lead) {
            frame_store(&d.F[0], r_r, r_r + beta * beta * d_Pd /* :287 */, beta * (e_Pd + alpha * d_Pd) /* :286 */, e_Pe,
                        new_model, norm_r0, alpha, beta, 1, j, stop0, nix, 0, 0);
            msdp_publish(d, c->k, j, 1);
        }
    }
    const double* __restrict__ rn = nix ? d.r2 : d.r;
    const double* __restrict__ Xf = d.full;
    double* __restrict__ H = d.Hmd;
    double pd = 0.0;
    int stride = MSDP_WAVES * RPW;
    if (d.sweep) msdp_sweep_rows(d.n_loc, d.G, MSDP_WAVES * RPW, lo, hi, stride);     // (the streaming loop above keeps the chunks)
    // streaming (nt) accesses for what this launch touches once -- Y, mdelta, H mdelta, eG -- so that the L2 keeps the rows of
    // the gathered vector (hess_sparse_obl_body, msdp_kernels.hip)
    const bool nt = (d.sweep & 2) != 0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += stride) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 acc[NCH], y[NCH], u[NCH], x[NCH];
            double udot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[ch] = make_double2(0.0, 0.0);
                const int col = 2 * sub + ch * 2 * LPR;
                const bool ok = col < d.ld;
                const int64_t o = (int64_t)row * d.ld + col;
                y[ch] = ok ? (nt ? ld2_nt(Yl + o) : ld2(Yl + o)) : make_double2(0.0, 0.0);
                u[ch] = ok ? ((nt && !DIRECT) ? ld2_nt(d.md + o) : ld2(d.md + o)) : make_double2(0.0, 0.0);
                x[ch] = u[ch];                                                          // the row the product is taken of
                if (!DIRECT) {
                    x[ch] = ok ? ld2(rn + o) : make_double2(0.0, 0.0);                  // r' (tangent)
                    u[ch] = make_double2(x[ch].x + beta * u[ch].x, x[ch].y + beta * u[ch].y);     // :273
                    udot += u[ch].x * y[ch].x + u[ch].y * y[ch].y;
                }
            }
            const double eg = eG[row];
            spmm_row<LPR, NCH, ELL>(d, row, sub, Xf, acc);
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) dot += acc[ch].x * y[ch].x + acc[ch].y * y[ch].y;
            dot = msdp_group_sum<LPR>(dot);                 // sum(Y.*eH)
            if (!DIRECT) udot = msdp_group_sum<LPR>(udot);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    // Hess(x) of the gathered vector (ManiSDP_onlyunitdiag.m:128-129)
                    double2 hq;
                    hq.x = acc[ch].x - y[ch].x * dot - x[ch].x * eg;
                    hq.y = acc[ch].y - y[ch].y * dot - x[ch].y * eg;
                    if (!DIRECT) {
                        // Hess is linear on the tangent space: Hess(mdelta') = Hess(r') + beta * Hess(mdelta)
                        const double2 ho = nt ? ld2_nt(H + o) : ld2(H + o);
                        hq.x = fma(beta, ho.x, hq.x); hq.y = fma(beta, ho.y, hq.y);
                        u[ch].x -= y[ch].x * udot; u[ch].y -= y[ch].y * udot;            // :283
                        if (nt) st2_nt(d.md + o, u[ch]); else st2(d.md + o, u[ch]);
                    }
                    if (nt) st2_nt(H + o, hq); else st2(H + o, hq);
                    pd += u[ch].x * hq.x + u[ch].y * hq.y;
                }
            }
        }
    }
    msdp_put_partial(d.P, P_DHD, pd, sh);
}

// ------------------------------------------------------------------ launchers
static inline void t1_lpr_for(int ld, int& lpr, int& nch) {
    int half = ld / 2;
    lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    nch = (half + lpr - 1) / lpr;
    if (nch < 1) nch = 1;
}

// trip1 = 1 (default): every row-sharded handle with sparse C on the oblique manifold, and the chunked path of a single rank
// (measured, tools/archive/trip1_single_probe.py and trip1_small_probe.py, linear / two-launch / three-launch: n = 10^6, p = 32: 773 /
// 889 / 943 us per trip; n = 250 000, p = 32: 190 / 227 / 216; n = 40 000, p = 40: 51 / 63 / 57; G81 with the persistent kernel
// off, p = 32: 21.7 / 25.4 / 23.7, p = 128: 62 / 67 / 75; G1 (CSR rows), p = 40: 34.4 / 48.2 / 34.7); 0: the two- / three-launch
// trips (msdp_trip2.hip, msdp_kernels.hip)
int msdp_trip1_ok(msdp_handle h) {
    const Dev& d = h->d;
    if (!(h->tune.trip1 && d.costkind == COST_SPARSE && d.manifold == MANI_OBLIQUE && d.r2 && d.xs && d.xs_all && d.xcount && !d.rowfree
          && d.ld <= 1024 && h->nranks <= MSDP_XS_MAX_RANKS)) return 0;
    if (h->use_comm) return 1;
    return h->nranks == 1;
}

int msdp_launch_trip1_init(msdp_handle h) {
    hipLaunchKernelGGL(k_tcg1_init, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}

#define T1_HEAD(L, N)                                                                                                  \
    do {                                                                                                               \
        if (direct) {                                                                                                  \
            if (h->d.ellW > 0) hipLaunchKernelGGL((k_tcg1_head<L, N, true, true>), grid, block, 0, h->stream, h->d);   \
            else hipLaunchKernelGGL((k_tcg1_head<L, N, false, true>), grid, block, 0, h->stream, h->d);                \
        } else {                                                                                                       \
            if (h->d.ellW > 0) hipLaunchKernelGGL((k_tcg1_head<L, N, true, false>), grid, block, 0, h->stream, h->d);  \
            else hipLaunchKernelGGL((k_tcg1_head<L, N, false, false>), grid, block, 0, h->stream, h->d);               \
        }                                                                                                              \
    } while (0)

int msdp_launch_trip1_head(msdp_handle h, bool direct) {
    int lpr, nch;
    t1_lpr_for(h->d.ld, lpr, nch);
    const dim3 grid(h->d.G), block(MSDP_BLOCK);
    if (nch == 1) {
        switch (lpr) {
            case 1: T1_HEAD(1, 1); break;
            case 2: T1_HEAD(2, 1); break;
            case 4: T1_HEAD(4, 1); break;
            case 8: T1_HEAD(8, 1); break;
            case 16: T1_HEAD(16, 1); break;
            case 32: T1_HEAD(32, 1); break;
            default: T1_HEAD(64, 1); break;
        }
    } else if (nch == 2) T1_HEAD(64, 2);
    else if (nch <= 4) T1_HEAD(64, 4);
    else if (nch <= 8) T1_HEAD(64, 8);
    else { msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", h->d.p); return MSDP_EUNSUPPORTED; }
    HIPCHK(hipGetLastError());
    return 0;
}

#define T1_UPD(L, N) hipLaunchKernelGGL((k_tcg1_upd<L, N>), grid, block, 0, h->stream, h->d)

int msdp_launch_trip1_upd(msdp_handle h) {
    int lpr, nch;
    t1_lpr_for(h->d.ld, lpr, nch);
    const dim3 grid(h->d.G), block(MSDP_BLOCK);
    if (nch == 1) {
        switch (lpr) {
            case 1: T1_UPD(1, 1); break;
            case 2: T1_UPD(2, 1); break;
            case 4: T1_UPD(4, 1); break;
            case 8: T1_UPD(8, 1); break;
            case 16: T1_UPD(16, 1); break;
            case 32: T1_UPD(32, 1); break;
            default: T1_UPD(64, 1); break;
        }
    } else if (nch == 2) T1_UPD(64, 2);
    else if (nch <= 4) T1_UPD(64, 4);
    else if (nch <= 8) T1_UPD(64, 8);
    else { msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", h->d.p); return MSDP_EUNSUPPORTED; }
    HIPCHK(hipGetLastError());
    return 0;
}
