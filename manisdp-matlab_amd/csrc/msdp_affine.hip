// msdp_affine.hip -- the sparse A(.) / A'(.) operators and the cost / gradient / Hess-vec
// of the two affine entry points, on the oblique (unitdiag) and sphere (unittrace) manifolds.
//
// Reference expressions (Y, U are n x p row-major here = MATLAB's p x n for unitdiag):
//   cost   ManiSDP_unitdiag.m:152-157   X=Y'*Y; Axb=A*x-b-y/sigma; f=c'*x+.5*sigma*(Axb'*Axb)
//   grad   ManiSDP_unitdiag.m:159-164   eS=reshape(c+sigma*At*Axb,n,n); eG=2*Y*eS; YeG=sum(Y.*eG); G=eG-Y.*YeG
//   hess   ManiSDP_unitdiag.m:166-171   YU=Y'*U; AyU=reshape(A'*(At'*YU(:)),n,n); eH=2*U*eS+4*sigma*(Y*AyU);
//                                       H=eH-Y.*sum(Y.*eH)-U.*YeG
//   cost   ManiSDP_unittrace.m:156-165  ... z=sum(X.*eS,'all'); G=2*eS*Y-2*z*Y
//   hess   ManiSDP_unittrace.m:171-177  H=2*eS*U+4*sigma*(AyU*Y); H=H-trace(H*Y')*Y-2*z*U
//
// The n x n matrices X, YU are never formed: A(Ya Yb') is an SDDMM over the pattern of At
// (one LPR-lane group per constraint column, p-wide row panels gathered from L2), A'(w) is a
// CSR-by-entry SpMV that writes the dense n x nS matrix the MFMA kernel then contracts.
// The constraint matrices A_k and C are symmetric (SeDuMi data), so row-major == column-major.
#include "msdp_device.h"
#include "msdp_affine_dev.h"
#include <math.h>
#include <algorithm>
#include <cstring>
#include <vector>

int msdp_dev_alloc_bytes(msdp_handle h, void** out, size_t bytes);
int msdp_dense_nS(int n);
int msdp_k_sum_to_fwd(msdp_handle h, int which, double* out);
int msdp_allreduce_partials(msdp_handle h, int first, int count);          // msdp_api.hip
int msdp_allgather_rows(msdp_handle h, const double* local_rows);
int msdp_allgather_vec(msdp_handle h, const double* local, double* all, size_t count_per_rank);
int msdp_dense_gemm(msdp_handle h, int nmat, const double* const* M, const double* const* X, const double* scale,
                    const int* active_flag, const double** slab_out, int64_t* stride_out, int* SK_out);

// item value = sum over the item's nonzeros (i,j,val) of val * <Ya_i, Yb_j>   (one LPR-lane group per item)
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_sddmm(AffineDev a, const double* __restrict__ Ya,
                                                    const double* __restrict__ Yb, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CPW = 64 / LPR;                       // items per wave
    const int sub = lane & (LPR - 1), csub = lane / LPR;
    const int64_t stride = (int64_t)gridDim.x * MSDP_WAVES * CPW;
    for (int64_t it = ((int64_t)blockIdx.x * MSDP_WAVES + wave) * CPW + csub; it < a.nitems; it += stride) {
        const int s0 = a.it0[it], s1 = a.it1[it];
        double acc = 0.0;
        // U nonzeros at a time (16 at a time was measured: 2x slower on BQP, whose constraints have ~4 nonzeros): their index loads and row gathers are independent, one at a time left a lane group
        // with a single request in flight (64 dependent round trips per item = 39 us for the trace row of a theta
        // problem, whatever the size of the rest)
        constexpr int U = NCH == 1 ? 4 : 2;
        for (int t = s0; t < s1; t += U) {
            int ii[U], jj[U];
            double vv[U], dd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = t + u < s1;
                const int tt = in ? t + u : s1 - 1;
                ii[u] = a.ci[tt]; jj[u] = a.cj[tt];
                vv[u] = in ? a.cv[tt] : 0.0;
                dd[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double* ya = Ya + (int64_t)ii[u] * a.ld + 2 * sub;
                const double* yb = Yb + (int64_t)jj[u] * a.ld + 2 * sub;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (2 * sub + ch * 2 * LPR < a.ld) {
                        const double2 x = ld2(ya + ch * 2 * LPR), z = ld2(yb + ch * 2 * LPR);
                        dd[u] += x.x * z.x + x.y * z.y;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma(vv[u], dd[u], acc);
        }
        acc = msdp_group_sum<LPR>(acc);
        if (sub == 0) a.ival[it] = acc;
    }
}
// w_k = sum of the items of constraint k (fixed order).  mode 0: store w.  mode 1 (cost): also
// Axb = w - b - y/sigma into axb_out and the partial sums of Axb^2 -> P_AXB (launch with MSDP_MAX_GRID workgroups).
__global__ __launch_bounds__(MSDP_BLOCK) void k_sddmm_finish(AffineDev a, int mode, double* axb_out, double sigma, double* P,
                                                           const int* skip_flag, int skip_when) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (skip_flag && *skip_flag == skip_when) return;
    double pss = 0.0;
    // short constraints: one thread each
    for (int64_t k = (int64_t)blockIdx.x * MSDP_BLOCK + threadIdx.x; k < a.m; k += (int64_t)gridDim.x * MSDP_BLOCK) {
        const int i0 = a.kit[k], i1 = a.kit[k + 1];
        if (i1 - i0 > FIN_SHORT) continue;
        double acc = 0.0;
        for (int it = i0; it < i1; ++it) acc += a.ival[it];
        a.w[k] = acc;
        if (mode == 1) {
            const double r = acc - a.b[k] - a.y[k] / sigma;
            axb_out[k] = r;
            pss += r * r;
        }
    }
    // long constraints (the trace row of a theta problem, the all-ones constraint of a gpp problem: up to n^2/16
    // items): one wave each, lanes stride over the items
    {
        const int lane = threadIdx.x & 63;
        const int wv = blockIdx.x * (MSDP_BLOCK / 64) + (threadIdx.x >> 6), nwv = gridDim.x * (MSDP_BLOCK / 64);
        for (int q = wv; q < a.nlong; q += nwv) {
            const int k = a.longk[q];
            const int i0 = a.kit[k], i1 = a.kit[k + 1];
            double a0 = 0.0, a1 = 0.0;
            int it = i0 + lane;
            for (; it + 64 < i1; it += 128) { a0 += a.ival[it]; a1 += a.ival[it + 64]; }
            if (it < i1) a0 += a.ival[it];
            const double acc = msdp_wave_sum(a0 + a1);
            if (lane == 0) {
                a.w[k] = acc;
                if (mode == 1) {
                    const double r = acc - a.b[k] - a.y[k] / sigma;
                    axb_out[k] = r;
                    pss += r * r;
                }
            }
        }
    }
    if (mode == 1) msdp_put_partial(P, P_AXB, pss, sh);
}

// k_sddmm + k_sddmm_finish in ONE launch (one rank).  A lane group sums a short constraint whole (its nonzeros are contiguous)
// and writes w_k; the items of the long constraints (trace rows) go to ival with agent-coherent stores and the workgroup that
// arrives last sums them per constraint in item order -- the same order whichever workgroup it is.
//   mode 0: w.   mode 1 (cost): also Axb = w - b - y/sigma -> axb_out and the partial sums of Axb^2 -> P_AXB: launch with
//   MSDP_MAX_GRID - 1 workgroups, slot MSDP_MAX_GRID - 1 takes the long constraints.   mode 2 (sphere Hess-vec, Ya = Y, Yb = U):
//   also the partial sums of w_k (A x)_k -> P_T3 (slot gridDim.x: the long constraints) with (A x)_k = Axb_k + b_k + y_k/sigma of
//   the current point, and of <U, G>, <U, Y> over this workgroup's rows -> P_T1, P_T2: together they give the sphere's
//   tr(H Y') without a pass over H (k_sph_hess_fused).
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_sddmm1(AffineDev a, Dev d, const double* __restrict__ Ya, const double* __restrict__ Yb,
                                                     int mode, double* axb_out, double sigma, const int* skip_flag, int skip_when, int nbl) {
    __shared__ double sh[3 * MSDP_WAVES + 8];
    __shared__ int lastflag;
    if (skip_flag && *skip_flag == skip_when) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CPW = 64 / LPR;
    const int sub = lane & (LPR - 1), csub = lane / LPR;
    const int64_t nunits = (int64_t)a.nshort + a.nlit;
    const int cur = mode == 2 ? d.ctl->cur : 0;
    const double* __restrict__ axc = cur ? a.Axb[1] : a.Axb[0];
    const int slotL = mode == 1 ? MSDP_MAX_GRID - 1 : (int)gridDim.x;
    double pacc = 0.0;
    int haslong = 0;
    const int64_t ustride = (int64_t)gridDim.x * MSDP_WAVES * CPW;
    // mode 2: the rows of <U, G>, <U, Y> are requested FIRST -- they depend on nothing and their latency hides behind the units'
    // chain (record -> nonzeros -> panel rows): every launch of this chain is a handful of dependent round trips, not bandwidth
    int lo = 0, hi = 0;
    double p1 = 0.0, p2 = 0.0;
    if (mode == 2) {
        const unsigned q = (unsigned)d.n_loc / gridDim.x, r = (unsigned)d.n_loc - q * gridDim.x, c = blockIdx.x;
        lo = (int)(c * q + (c < r ? c : r)); hi = lo + (int)q + (c < r ? 1 : 0);
        const double* __restrict__ Gr = cur ? d.Gr[1] : d.Gr[0];
        const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
        for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
            const double2 u = ld2(Yb + i), g = ld2(Gr + i), y = ld2(Ya + i);
            p1 += u.x * g.x + u.y * g.y;
            p2 += u.x * y.x + u.y * y.y;
        }
    }
    for (int64_t u = ((int64_t)blockIdx.x * MSDP_WAVES + wave) * CPW + csub; u < nunits; u += ustride) {
        const int s0 = a.us0[u], s1 = a.us1[u], kk = a.uk[u];
        const int k = kk >= 0 ? kk : -1, lq = kk >= 0 ? 0 : -1 - kk;
        if (kk < 0) haslong = 1;
        // what the constraint's value is combined with (known as soon as k is): requested together with the nonzeros
        double eb = 0.0, ey = 0.0, ea = 0.0;
        if (k >= 0 && mode != 0) { eb = a.b[k]; ey = a.y[k] / sigma; if (mode == 2) ea = axc[k]; }
        double acc = 0.0;
        constexpr int U = NCH == 1 ? 4 : 2;
        for (int t = s0; t < s1; t += U) {
            int ii[U], jj[U];
            double vv[U], dd[U];
#pragma unroll
            for (int u2 = 0; u2 < U; ++u2) {
                const bool in = t + u2 < s1;
                const int tt = in ? t + u2 : s1 - 1;
                ii[u2] = a.ci[tt]; jj[u2] = a.cj[tt];
                vv[u2] = in ? a.cv[tt] : 0.0;
                dd[u2] = 0.0;
            }
#pragma unroll
            for (int u2 = 0; u2 < U; ++u2) {
                const double* ya = Ya + (int64_t)ii[u2] * a.ld + 2 * sub;
                const double* yb = Yb + (int64_t)jj[u2] * a.ld + 2 * sub;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (2 * sub + ch * 2 * LPR < a.ld) {
                        const double2 x = ld2(ya + ch * 2 * LPR), z = ld2(yb + ch * 2 * LPR);
                        dd[u2] += x.x * z.x + x.y * z.y;
                    }
                }
            }
#pragma unroll
            for (int u2 = 0; u2 < U; ++u2) acc = fma(vv[u2], dd[u2], acc);
        }
        acc = msdp_group_sum<LPR>(acc);
        if (sub == 0) {
            if (k >= 0) {
                a.w[k] = acc;
                if (mode == 1) { const double r = acc - eb - ey; axb_out[k] = r; pacc += r * r; }      // ManiSDP_unitdiag.m:154, same order
                else if (mode == 2) pacc += acc * (ea + eb + ey);
            } else {
                __hip_atomic_store(a.ival + lq, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (mode == 2) {
        msdp_put_partials3(d.P, P_T1, p1, P_T2, p2, P_T3, pacc, sh);
    } else if (mode == 1) {
        msdp_put_partial(d.P, P_AXB, pacc, sh);
    }
    // ---- long constraints: the last workgroup to arrive sums their items
    if (a.nlit == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && mode != 0) d.P[(mode == 1 ? P_AXB : P_T3) * MSDP_MAX_GRID + slotL] = 0.0;
        return;
    }
    // only the `nbl` workgroups that summed items of long constraints (a host-known count: the items are consecutive units) take part
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!__syncthreads_or(haslong)) return;
    if (threadIdx.x == 0) lastflag = (__hip_atomic_fetch_add(a.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nbl - 1u) ? 1 : 0;
    __syncthreads();
    if (!lastflag) return;
    double pl = 0.0;
    for (int q = wave; q < a.nlong; q += MSDP_WAVES) {
        const int k = a.longk[q];
        const int i0 = a.lkit[q], i1 = a.lkit[q + 1];
        // four agent-coherent loads in flight per lane and ONE wait (the compiler puts a full s_waitcnt behind every atomic load:
        // the trace row of a theta problem -- 313 items -- cost 7 us of serialized round trips that way)
        double a0 = 0.0;
        for (int base = i0; base < i1; base += 256) {
            const double* q0 = a.ival + min(base + lane, i1 - 1);
            const double* q1 = a.ival + min(base + lane + 64, i1 - 1);
            const double* q2 = a.ival + min(base + lane + 128, i1 - 1);
            const double* q3 = a.ival + min(base + lane + 192, i1 - 1);
            double v0, v1, v2, v3;
            asm volatile(
                "global_load_dwordx2 %0, %4, off sc1\n\t"
                "global_load_dwordx2 %1, %5, off sc1\n\t"
                "global_load_dwordx2 %2, %6, off sc1\n\t"
                "global_load_dwordx2 %3, %7, off sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
                : "memory");
            if (base + lane < i1) a0 += v0;
            if (base + lane + 64 < i1) a0 += v1;
            if (base + lane + 128 < i1) a0 += v2;
            if (base + lane + 192 < i1) a0 += v3;
        }
        const double acc = msdp_wave_sum(a0);
        if (lane == 0) {
            a.w[k] = acc;
            if (mode == 1) { const double r = acc - a.b[k] - a.y[k] / sigma; axb_out[k] = r; pl += r * r; }
            else if (mode == 2) pl += acc * (axc[k] + a.b[k] + a.y[k] / sigma);
        }
    }
    __syncthreads();                                       // sh is free again (msdp_put_partials3 above has finished)
    if (lane == 0) sh[wave] = pl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < MSDP_WAVES; ++i) s += sh[i];
        if (mode != 0) d.P[(mode == 1 ? P_AXB : P_T3) * MSDP_MAX_GRID + slotL] = s;
        *a.cnt = 0u;
    }
}

// Sphere Hess-vec, everything behind the dense contraction in ONE launch (ManiSDP_unittrace.m:173-176, ManiSDP.m:160-162):
//   H_raw(i,:) = sum of the split-K slabs (2*eS*U, or 2*eS*U + 4 sigma AyU*Y when AyU is dense)
//              + 4 sigma * sum_j A'(w)_ij Y(j,:) over the entries At touches (k_support_spmm's sparse product, when nsup > 0)
//   H = H_raw - t*Y - 2 z U,  t = tr(H_raw Y'),   partial sums of <U, H> -> P_DHD.
// t needs no pass over H_raw: <2 eS U, Y> = <U, 2 eS Y> = <U, G> + 2 z <U, Y> (G = 2 eS Y - 2 z Y is the stored gradient, eS is
// symmetric) and <4 sigma A'(w) Y, Y> = 4 sigma <w, A(Y Y')> -- the three sums k_sddmm1 (mode 2) left in P_T1..P_T3.
// One wave per row; replaces k_support_spmm + k_sph_hess_raw + k_sph_hess_finish (20 us of launches at n = 5000).
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_hess_fused(Dev d, AffineDev a, const double* slab, int64_t slab_stride, int SK,
                                                             double sigma, int G2, int support, int cur, int hetero) {
    __shared__ double sh[3 * MSDP_WAVES + 8];
    __shared__ double wl[MSDP_WAVES + 1];                      // values of the long constraints (<= 16), and their share of the third sum
    if (!d.F[0].active) return;
    const bool euc = d.manifold == MANI_EUCLID;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    // `cur` is the host's (the affine kinds bake the slot into their launches: one graph per slot).  The three sums of t are
    // requested by the LAST wave up front and consumed behind the row work (one barrier there): nothing waits for them early
    double z = 0.0;
    if (!euc) {
        z = d.ctl->z_sphere[cur];
        if (wave == MSDP_WAVES - 1) {
            const double s1 = msdp_sum_partials(d.P, P_T1, G2), s2 = msdp_sum_partials(d.P, P_T2, G2), s3 = msdp_sum_partials(d.P, P_T3, G2 + 1);
            if (lane == 0) { sh[0] = s1; sh[1] = s2; sh[2] = s3; }
        }
    }
    // Long constraints (a trace row): wave q provides the value of number q -- hetero (the SDDMM ran as a side job of the contraction
    // launch, msdp_sddmm_side): the sum of its item values, in item order, the same in every workgroup; else what k_sddmm1 left in w
    // (only the side-job route needs wl, and it is taken for nlong <= MSDP_WAVES only; otherwise k_sddmm1's last workgroup has written
    // w[longk[q]] and the records below read it there -- any number of long constraints)
    if (support && hetero && wave < a.nlong) {
        double a0 = 0.0;
        for (int it = a.lkit[wave] + lane; it < a.lkit[wave + 1]; it += 64) a0 += a.ival[it];
        const double v = msdp_wave_sum(a0);
        if (lane == 0) wl[wave] = v;
    }
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ w = a.w;
    const double s4 = 4.0 * sigma;
    double pd = 0.0;
    // LPR lanes per row (one double2 each, NCH column chunks), 64 / LPR rows per wave: the lanes of a row take one touched entry
    // each (flattened record -> w), then the values and column indices are broadcast inside the group and all its lanes
    // accumulate v * Y(j, :).  raw = slabs + 4 sigma * sparse part; y, u = the row of the point and of the direction.
    // (eight column chunks per lane, p = 513..1024: only the sparse part is held across the row work; the slab sums and the rows of Y
    // and U are read in `finish` -- holding all four sets took 128 registers per lane and spilled 640 bytes)
    constexpr bool HOLD = NCH < 8;
    auto rowwork = [&](int i, bool rok, double2 (&raw)[NCH], double2 (&y)[HOLD ? NCH : 1], double2 (&u)[HOLD ? NCH : 1]) {
        double2 acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            acc[ch] = make_double2(0.0, 0.0); raw[ch] = acc[ch];
            if (HOLD) { y[HOLD ? ch : 0] = acc[ch]; u[HOLD ? ch : 0] = acc[ch]; }
            const int c = 2 * sub + ch * 2 * LPR;
            if (HOLD && rok && c < d.ld) {                         // independent of the sparse part: requested first
                const int64_t o = (int64_t)i * d.ld + c;
                raw[ch] = msdp_sum_slabs(slab, slab_stride, SK, o);
                y[HOLD ? ch : 0] = ld2(Yl + o); u[HOLD ? ch : 0] = ld2(d.md + o);
            }
        }
        if (support) {
            const int qb = rok ? a.suprow[i] : 0, qe = rok ? a.suprow[i + 1] : 0;
            for (int q0 = qb; __builtin_amdgcn_ballot_w64(q0 < qe) != 0ULL; q0 += LPR) {
                const int cnt = min(LPR, max(qe - q0, 0));
                int jl = 0;
                double vl = 0.0;
                if (sub < cnt) {
                    // flattened record of the entry: column, first (coefficient, constraint) pair; further pairs are rare
                    jl = a.sqj[q0 + sub];
                    const int k0 = a.sqk[q0 + sub], more = a.sqmore[q0 + sub];
                    vl = a.sqv[q0 + sub] * (k0 >= 0 ? w[k0] : (hetero ? wl[-1 - k0] : w[a.longk[-1 - k0]]));
                    if (more > 0) {
                        const int s0 = a.rp[a.sup[q0 + sub]] + 1;
                        for (int tt = s0; tt < s0 + more; ++tt) { const int kx = a.rkx[tt]; vl = fma(a.rv[tt], kx >= 0 ? w[kx] : (hetero ? wl[-1 - kx] : w[a.longk[-1 - kx]]), vl); }
                    }
                }
                for (int e0 = 0; __builtin_amdgcn_ballot_w64(e0 < cnt) != 0ULL; e0 += SPB) {      // until the longest row of the wave is done
                    int jj[SPB];
                    double v[SPB];
#pragma unroll
                    for (int q = 0; q < SPB; ++q) {
                        const int e = min(e0 + q, max(cnt - 1, 0));
                        jj[q] = __shfl(jl, e, LPR);
                        const double vv = __shfl(vl, e, LPR);
                        v[q] = (e0 + q < cnt) ? vv : 0.0;
                    }
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const int c = 2 * sub + ch * 2 * LPR;
                        if (c < d.ld) {
                            double2 yy[SPB];
#pragma unroll
                            for (int q = 0; q < SPB; ++q) yy[q] = ld2(Yl + (int64_t)jj[q] * d.ld + c);
#pragma unroll
                            for (int q = 0; q < SPB; ++q) { acc[ch].x = fma(v[q], yy[q].x, acc[ch].x); acc[ch].y = fma(v[q], yy[q].y, acc[ch].y); }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) { raw[ch].x += s4 * acc[ch].x; raw[ch].y += s4 * acc[ch].y; }
    };
    double t = 0.0;
    auto finish = [&](int i, bool rok, const double2 (&raw)[NCH], const double2 (&y)[HOLD ? NCH : 1], const double2 (&u)[HOLD ? NCH : 1]) {
        if (!rok) return;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = 2 * sub + ch * 2 * LPR;
            if (c < d.ld) {
                const int64_t o = (int64_t)i * d.ld + c;
                double2 rw = raw[ch];
                if (!HOLD) { const double2 sl = msdp_sum_slabs(slab, slab_stride, SK, o); rw.x += sl.x; rw.y += sl.y; }
                const double2 yy = HOLD ? y[HOLD ? ch : 0] : ld2(Yl + o), uu = HOLD ? u[HOLD ? ch : 0] : ld2(d.md + o);
                double2 hq;
                hq.x = rw.x - t * yy.x - 2.0 * z * uu.x;
                hq.y = rw.y - t * yy.y - 2.0 * z * uu.y;
                st2(d.Hmd + o, hq);
                pd += uu.x * hq.x + uu.y * hq.y;
            }
        }
    };
    if (support && hetero && a.nlong > 0) __syncthreads();         // wl is in place
    double2 raw[NCH], y[HOLD ? NCH : 1], u[HOLD ? NCH : 1];
    int row0 = lo + wave * RPW;                                    // wave-uniform
    const bool first = row0 < hi;
    if (first) rowwork(row0 + rsub, row0 + rsub < hi, raw, y, u);  // first pass in front of the barrier that hands out t
    if (!euc) {
        __syncthreads();
        double s3 = sh[2];
        if (hetero && support) {                                   // the long constraints' share of <w, A(YY')>, in constraint order
            const double* __restrict__ axc = cur ? a.Axb[1] : a.Axb[0];
            for (int q = 0; q < a.nlong; ++q) { const int k = a.longk[q]; s3 += wl[q] * (axc[k] + a.b[k] + a.y[k] / sigma); }
        }
        t = (sh[0] + 2.0 * z * sh[1]) + 4.0 * sigma * s3;
    }
    if (first) finish(row0 + rsub, row0 + rsub < hi, raw, y, u);
    for (row0 += MSDP_WAVES * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        rowwork(row0 + rsub, row0 + rsub < hi, raw, y, u);
        finish(row0 + rsub, row0 + rsub < hi, raw, y, u);
    }
    __syncthreads();                                               // sh[0..2] has been read by everybody
    msdp_put_partial(d.P, P_DHD, pd, sh + 8);
}

// fp64-MFMA version of k_gram (default): W = Ya * Yb' as 64 x 64 tiles, four waves per tile in a 2 x 2 arrangement,
// each wave 32 x 32 = 2 x 2 accumulator tiles of v_mfma_f64_16x16x4_f64.  Both operands are rows of a thin panel with k
// contiguous, so they are staged the same way (64 rows x 32 k per panel, row stride 36 doubles: the 16 rows of a
// fragment read start on distinct 32-byte bank groups) and read as fragments of 4 contiguous k per lane (two
// ds_read_b128), which feed four MFMAs -- the k order inside a 16-k step is permuted identically for A and B.
// In the BQP d = 60 solve the factor grows to p = 300 and the VALU kernel above was the largest single kernel of the
// RTR phase (228 us = 8.8 TFLOP/s at p = 300).  Prefetching the next tile's panel pieces into registers under the
// MFMAs was measured: 124 instead of 92 VGPRs (3 instead of 4 waves per SIMD) and 3 % slower.
typedef double gram_d4 __attribute__((ext_vector_type(4)));
#define GRAM_KT 32
#define GRAM_LDS (GRAM_KT + 4)
// sym != 0: Wsym = Ya*Yb' + Yb*Ya' on the tiles on and above the diagonal only -- the same flops as the full W, half the
// stores, and the gather that follows reads one entry per symmetric pair of a constraint.
// 512 threads = two groups of four waves, each with its own staging buffers: with the symmetric form group g computes
// product g (Ya_I*Yb_J' or Yb_I*Ya_J'); otherwise (one product) group g takes half of the k range.  The partial tiles
// are added through LDS at the end.  (Half as many workgroups as the full W would otherwise leave 1-2 four-wave
// workgroups per CU, too few to cover the staging latency: 79 instead of 57 us at p = 300.)
__global__ __launch_bounds__(512) void k_gram_mfma(int n, int nS, int ld, const double* __restrict__ Ya0,
                                                   const double* __restrict__ Yb0, double* __restrict__ W,
                                                   const int* skip_flag, int skip_when, int sym) {
    __shared__ __attribute__((aligned(16))) double Stage[2 * 2 * 64 * GRAM_LDS];      // [group][A|B][64 x GRAM_LDS]
    if (skip_flag && *skip_flag == skip_when) return;
    if (sym && blockIdx.y > blockIdx.x) return;
    const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const bool two = sym && Ya0 != Yb0;                                 // two products, one per group
    const double post = (sym && Ya0 == Yb0) ? 2.0 : 1.0;
    const double* __restrict__ Ya = (two && grp) ? Yb0 : Ya0;
    const double* __restrict__ Yb = (two && grp) ? Ya0 : Yb0;
    const int nk = (ld + GRAM_KT - 1) / GRAM_KT;
    const int niter = two ? nk : (nk + 1) / 2;                          // the same trip count in both groups (barriers)
    const int kbeg = two ? 0 : grp * niter * GRAM_KT;
    const int kend = two ? ld : min(ld, (grp + 1) * niter * GRAM_KT);
    double* As = Stage + grp * (2 * 64 * GRAM_LDS);
    double* Bs = As + 64 * GRAM_LDS;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;
    gram_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (gram_d4){0.0, 0.0, 0.0, 0.0};
    // staging: thread -> (row = tid / 16 + 16 * q, column pair = tid % 16), q = 0..3, both panels
    const int sr = tid >> 4, sc = 2 * (tid & 15);
    for (int it = 0; it < niter; ++it) {
        const int k0 = kbeg + it * GRAM_KT;
        double2 ra[4], rb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sr + 16 * q;
            const bool ck = k0 + sc < kend;                         // ld is even: a pair is in or out as a whole
            const bool oa = ck && ti + r < n, ob = ck && tj + r < n;
            const double2 va = ld2(Ya + (oa ? (int64_t)(ti + r) * ld + k0 + sc : 0));
            const double2 vb = ld2(Yb + (ob ? (int64_t)(tj + r) * ld + k0 + sc : 0));
            ra[q] = oa ? va : make_double2(0.0, 0.0);
            rb[q] = ob ? vb : make_double2(0.0, 0.0);
        }
        __syncthreads();                                            // previous tile's fragment reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sr + 16 * q;
            *reinterpret_cast<double2*>(&As[r * GRAM_LDS + sc]) = ra[q];
            *reinterpret_cast<double2*>(&Bs[r * GRAM_LDS + sc]) = rb[q];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GRAM_KT / 16; ++s) {
            double2 fa[2][2], fb[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const double* pa = &As[(wr * 32 + 16 * a + i) * GRAM_LDS + 16 * s + 4 * g];
                const double* pb = &Bs[(wc * 32 + 16 * a + i) * GRAM_LDS + 16 * s + 4 * g];
                fa[a][0] = *reinterpret_cast<const double2*>(pa); fa[a][1] = *reinterpret_cast<const double2*>(pa + 2);
                fb[a][0] = *reinterpret_cast<const double2*>(pb); fb[a][1] = *reinterpret_cast<const double2*>(pb + 2);
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const double av = (t4 == 0) ? fa[a][0].x : (t4 == 1) ? fa[a][0].y : (t4 == 2) ? fa[a][1].x : fa[a][1].y;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const double bv = (t4 == 0) ? fb[b][0].x : (t4 == 1) ? fb[b][0].y : (t4 == 2) ? fb[b][1].x : fb[b][1].y;
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[a][b], 0, 0, 0);
                    }
                }
            }
        }
    }
    // group 1 hands its partial tile to group 0 through LDS (4096 doubles, the staging space is 9216)
    __syncthreads();
    if (grp == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) Stage[(((wave * 2 + a) * 2 + b) * 4 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (grp == 1) return;
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = tj + wc * 32 + 16 * b + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti + wr * 32 + 16 * a + g + 4 * r;
                const double v = acc[a][b][r] + Stage[(((wave * 2 + a) * 2 + b) * 4 + r) * 64 + lane];
                if (row < n && col < nS) W[(int64_t)row * nS + col] = post * v;
            }
        }
}

// ---- Gram route for A(Ya Yb'): when At is dense in its rows (BQP moment relaxations: 4.8 M nonzeros for an
// 1831 x 1831 matrix) the SDDMM above gathers two p-wide rows per nonzero (2.5 GB of L2 traffic at p = 32,
// measured 154 us); forming W = Ya*Yb' once (n^2 p flops, 27 MB; k_gram_mfma above) and gathering ONE double per
// nonzero is what the reference does (ManiSDP_unitdiag.m:153,167: X = Y'*Y, YU = Y'*U) and moves 20x less.
// w_k = sum over the nonzeros of constraint k of val * W[cidx], straight from the Gram matrix: one thread per
// constraint (its nonzeros are contiguous), a whole wave for the few constraints with more than
// FIN_SHORT*SDDMM_CHUNK nonzeros (trace rows).  Four nonzeros are in flight per thread (one at a time left a thread
// with a single request outstanding: 33 us for the 1.16 M constraints of BQP d = 60).  mode as in k_sddmm_finish: the
// per-item partial values and the separate finish launch of the SDDMM route are not needed here.
__global__ __launch_bounds__(MSDP_BLOCK) void k_gram_apply(AffineDev a, const double* __restrict__ W, int mode, double* axb_out,
                                                         double sigma, double* P, const int* skip_flag, int skip_when) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (skip_flag && *skip_flag == skip_when) return;
    double pss = 0.0;
    constexpr int LONG = FIN_SHORT * SDDMM_CHUNK;
    for (int64_t k = (int64_t)blockIdx.x * MSDP_BLOCK + threadIdx.x; k < a.m; k += (int64_t)gridDim.x * MSDP_BLOCK) {
        const int s0 = a.cjc[k], s1 = a.cjc[k + 1];
        if (s1 - s0 > LONG) continue;
        double acc = 0.0;
        for (int t = s0; t < s1; t += 4) {
            int cc[4];
            double vv[4], ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = t + u < s1;
                const int tt = in ? t + u : s1 - 1;
                cc[u] = a.cidx[tt];
                vv[u] = in ? a.cv[tt] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ww[u] = W[cc[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fma(vv[u], ww[u], acc);
        }
        a.w[k] = acc;
        if (mode == 1) {
            const double r = acc - a.b[k] - a.y[k] / sigma;
            axb_out[k] = r;
            pss += r * r;
        }
    }
    {
        const int lane = threadIdx.x & 63;
        const int wv = blockIdx.x * (MSDP_BLOCK / 64) + (threadIdx.x >> 6), nwv = gridDim.x * (MSDP_BLOCK / 64);
        for (int q = wv; q < a.nlong; q += nwv) {
            const int k = a.longk[q];
            const int s0 = a.cjc[k], s1 = a.cjc[k + 1];
            double a0 = 0.0, a1 = 0.0;
            int t = s0 + lane;
            for (; t + 64 < s1; t += 128) { a0 = fma(a.cv[t], W[a.cidx[t]], a0); a1 = fma(a.cv[t + 64], W[a.cidx[t + 64]], a1); }
            if (t < s1) a0 = fma(a.cv[t], W[a.cidx[t]], a0);
            const double acc = msdp_wave_sum(a0 + a1);
            if (lane == 0) {
                a.w[k] = acc;
                if (mode == 1) {
                    const double r = acc - a.b[k] - a.y[k] / sigma;
                    axb_out[k] = r;
                    pss += r * r;
                }
            }
        }
    }
    if (mode == 1) msdp_put_partial(P, P_AXB, pss, sh);
}

// out[i][j] = (base ? base[i][j] : 0) + scale * sum_k At[(i,j),k] * vec[k]   (dense n x nS, zero pad)
__global__ void k_adjoint_dense(AffineDev a, const double* __restrict__ base, const double* __restrict__ vec,
                                double scale, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int64_t tot = (int64_t)a.n * a.nS;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / a.nS), j = (int)(e - (int64_t)i * a.nS);
        double v = 0.0;
        if (j < a.n) {
            const int64_t r = (int64_t)i * a.n + j;
            const int s0 = a.rp[r], s1 = a.rp[r + 1];
            double acc = 0.0;
            for (int t = s0; t < s1; ++t) acc = fma(a.rv[t], vec[a.rk[t]], acc);
            v = (base ? base[e] : 0.0) + scale * acc;
        }
        out[e] = v;
    }
}

// Same result for SYMMETRIC data (A_k = A_k', base = base': SeDuMi's convention, bqpmom.m:80-91, qsmom.m, fromsdpa.m),
// which is what the affine entry points are fed: only the 32 x 32 tiles on and above the diagonal are computed (half
// the (coefficient, constraint) reads and half the 8-byte gathers of `vec`) and every off-diagonal tile is stored
// twice, the second time transposed through LDS, so both stores are 256-byte row segments.  The CSR-by-entry
// arrays are laid out in tile order (trp/trk/trv), so a workgroup streams one contiguous range of them; each
// thread owns four entries of the tile and runs their chains of dependent loads (offsets -> coefficient and
// constraint -> vec) side by side.  The flat kernel above walked all n^2 entries with one chain per thread and a
// 64-bit division per entry: 85-104 us per call on BQP d = 60 (0.10 of the HBM roofline).
#define ADJ_T 32
#define ADJ_LONG 8
__global__ __launch_bounds__(256) void k_adjoint_tiled(AffineDev a, const double* __restrict__ base, const double* __restrict__ vec,
                                                       double scale, double* __restrict__ out, const int* skip_flag, int skip_when) {
    __shared__ double tile[ADJ_T][ADJ_T + 1];
    if (skip_flag && *skip_flag == skip_when) return;
    if ((int)blockIdx.x >= a.ntp) {
        // ---- long entries: one wave each, lanes stride over the (coefficient, constraint) pairs; both copies stored
        const int lane = threadIdx.x & 63;
        const int q = ((int)blockIdx.x - a.ntp) * 4 + (threadIdx.x >> 6);
        if (q >= a.nlong_e) return;
        const int t0 = a.ls0[q], t1 = a.ls1[q];
        double acc = 0.0;
        for (int t = t0 + lane; t < t1; t += 64) acc = fma(a.trv[t], vec[a.trk[t]], acc);
        acc = msdp_wave_sum(acc);
        if (lane == 0) {
            const int o = a.lpos[q], om = a.lmir[q];
            const double v = (base ? base[o] : 0.0) + scale * acc;
            out[o] = v;
            if (om != o) out[om] = v;
        }
        return;
    }
    const int bi = a.tp_i[blockIdx.x], bj = a.tp_j[blockIdx.x];
    const int lj = threadIdx.x & 31, li0 = threadIdx.x >> 5;          // rows li0 + 8 q, q = 0..3
    const int* __restrict__ trp = a.trp + (size_t)blockIdx.x * (ADJ_T * ADJ_T);
    int s0[4], s1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = (li0 + 8 * q) * ADJ_T + lj;
        s0[q] = trp[e]; s1[q] = trp[e + 1];
    }
    bool lng[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { lng[q] = s1[q] - s0[q] > ADJ_LONG; if (lng[q]) s1[q] = s0[q]; }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // the first two nonzeros of the four entries side by side (most entries have at most two); trk / trv carry one
    // padding element, so the clamped index of an exhausted entry is always readable
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int kk[4];
        double vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = s0[q] + r < s1[q];
            const int t = in ? s0[q] + r : s0[q];
            kk[q] = a.trk[t];
            vv[q] = in ? a.trv[t] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fma(vv[q], vec[kk[q]], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int t = s0[q] + 2; t < s1[q]; ++t) acc[q] = fma(a.trv[t], vec[a.trk[t]], acc[q]);   // <= ADJ_LONG - 2 trips
    const int j = bj * ADJ_T + lj;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = bi * ADJ_T + li0 + 8 * q;
        v[q] = 0.0;
        if (i < a.n && j < a.nS && !lng[q]) {
            const size_t o = (size_t)i * a.nS + j;
            if (j < a.n) v[q] = (base ? base[o] : 0.0) + scale * acc[q];
            out[o] = v[q];                                           // pad columns are written as zeros
        }
    }
    if (bi == bj) return;                                            // diagonal tile: both triangles were computed
    // mirrored copy; long entries are marked with a NaN and skipped (the long-entry waves store both copies)
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[li0 + 8 * q][lj] = lng[q] ? __longlong_as_double(0x7ff8000000000000LL) : v[q];
    __syncthreads();
    const int jm = bi * ADJ_T + lj;                                  // mirrored column = row index of the tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int im = bj * ADJ_T + li0 + 8 * q;                     // mirrored row = column index of the tile
        const double tv = tile[lj][li0 + 8 * q];
        if (im < a.n && jm < a.nS && tv == tv) out[(size_t)im * a.nS + jm] = (jm < a.n) ? tv : 0.0;
    }
}


// B route: out = scale * B * Wsym on the 32 x 32 tiles on and above the diagonal, mirrored like k_adjoint_tiled.  Wsym = Ya*Yb' +
// Yb*Ya' comes from k_gram_mfma (upper 64 x 64 tiles).  Replaces k_gram_apply (one gather per nonzero of the upper At, writes w)
// + k_adjoint_tiled (one gather of w per nonzero): for BQP d = 60 4.2 M coefficients instead of 2 x 2.4 M, two dependent round
// trips (ELL slice -> gather) instead of three + three, and no 9-MB m-vector in between.
template <int BW, bool PK>
__global__ __launch_bounds__(256) void k_adjoint_gram(AffineDev a, const double* __restrict__ Wsym, double scale, double* __restrict__ out,
                                                      const int* skip_flag, int skip_when) {
    __shared__ double tile[ADJ_T][ADJ_T + 1];
    __shared__ double dict[PK ? 256 : 1];
    if (skip_flag && *skip_flag == skip_when) return;
    if (PK && (int)blockIdx.x < a.ntp) { dict[threadIdx.x] = a.bdict[threadIdx.x]; __syncthreads(); }
    if ((int)blockIdx.x >= a.ntp) {
        const int lane = threadIdx.x & 63;
        const int q = ((int)blockIdx.x - a.ntp) * 4 + (threadIdx.x >> 6);
        if (q >= a.bnlong) return;
        const int t0 = a.bls0[q], t1 = a.bls1[q];
        double acc = 0.0;
        for (int t = t0 + lane; t < t1; t += 64) acc = fma(a.blv[t], Wsym[a.blk[t]], acc);
        acc = msdp_wave_sum(acc);
        if (lane == 0) {
            const int o = a.blpos[q], om = a.blmir[q];
            const double v = scale * acc;
            out[o] = v;
            if (om != o) out[om] = v;
        }
        return;
    }
    const int bi = a.tp_i[blockIdx.x], bj = a.tp_j[blockIdx.x];
    const int lj = threadIdx.x & 31, li0 = threadIdx.x >> 5;
    const size_t tb = (size_t)blockIdx.x * BW * (ADJ_T * ADJ_T);
    int idx[4][BW];
    double val[4][BW];
    bool lng[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = (li0 + 8 * q) * ADJ_T + lj;
        lng[q] = a.blong[(size_t)blockIdx.x * (ADJ_T * ADJ_T) + e] != 0;
#pragma unroll
        for (int w = 0; w < BW; ++w) {
            if (PK) {
                const unsigned pk = a.bpk[tb + (size_t)w * (ADJ_T * ADJ_T) + e];
                idx[q][w] = (int)(pk & 0xffffffu); val[q][w] = dict[pk >> 24];
            } else { idx[q][w] = a.bidx[tb + (size_t)w * (ADJ_T * ADJ_T) + e]; val[q][w] = a.bval[tb + (size_t)w * (ADJ_T * ADJ_T) + e]; }
        }
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int w = 0; w < BW; ++w)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fma(val[q][w], Wsym[idx[q][w]], acc[q]);
    const int j = bj * ADJ_T + lj;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = bi * ADJ_T + li0 + 8 * q;
        v[q] = 0.0;
        if (i < a.n && j < a.nS && !lng[q]) {
            const size_t o = (size_t)i * a.nS + j;
            if (j < a.n) v[q] = scale * acc[q];
            out[o] = v[q];
        }
    }
    if (bi == bj) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[li0 + 8 * q][lj] = lng[q] ? __longlong_as_double(0x7ff8000000000000LL) : v[q];
    __syncthreads();
    const int jm = bi * ADJ_T + lj;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int im = bj * ADJ_T + li0 + 8 * q;
        const double tv = tile[lj][li0 + 8 * q];
        if (im < a.n && jm < a.nS && tv == tv) out[(size_t)im * a.nS + jm] = (jm < a.n) ? tv : 0.0;
    }
}

// The same update restricted to the entries that occur in some constraint.  For the SDPLIB-type problems (theta, gpp:
// m = O(n) constraints of a few entries each) At touches a fraction of a percent of the n^2 entries, and `out` keeps
// its value everywhere else from one call to the next (eS = C there, AyU = 0 there), so the full sweep above moves
// 2-3 matrices through HBM to change 55 000 numbers (theta-like n = 5000: 55 us per call, 3 us here).
__global__ __launch_bounds__(256) void k_adjoint_support(AffineDev a, const double* __restrict__ base, const double* __restrict__ vec,
                                                         double scale, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < a.nsup; q += gridDim.x * blockDim.x) {
        const int r = a.sup[q];
        const int i = r / a.n, j = r - i * a.n;
        const int64_t e = (int64_t)i * a.nS + j;
        const int s0 = a.rp[r], s1 = a.rp[r + 1];
        double acc = 0.0;
        for (int t = s0; t < s1; ++t) acc = fma(a.rv[t], vec[a.rk[t]], acc);
        out[e] = (base ? base[e] : 0.0) + scale * acc;
    }
}

// out(i,:) = scale * sum_j (A'(w))_ij * Yp(j,:) over the entries (i,j) At touches: the product AyU*Y of the Hess-vec
// without forming AyU (ManiSDP_unitdiag.m:168-169, ManiSDP_unittrace.m:173-174).  One wave per matrix row; the
// adjoint values of the entries of a row are computed one per lane, the panel rows are read 16 bytes per lane.  The result is handed to the epilogue as one more split-K slab.
template <int NCH>
__global__ __launch_bounds__(256) void k_support_spmm(AffineDev a, const double* __restrict__ w, const double* __restrict__ Yp,
                                                      double scale, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < a.n; i += gridDim.x * 4) {
        double2 acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = make_double2(0.0, 0.0);
        // Phase 1: the lanes take one entry each (64 at a time) and run its chain of dependent loads -- entry -> row
        // pointer -> (coefficient, constraint) -> w -- side by side.  Phase 2: the values and column indices are
        // broadcast lane by lane and all lanes accumulate v * Yp(j,:); four panel rows are requested together.
        const int q1 = a.suprow[i + 1];
        for (int q0 = a.suprow[i]; q0 < q1; q0 += 64) {
            const int cnt = min(64, q1 - q0);
            int jl = 0;
            double vl = 0.0;
            if (lane < cnt) {
                const int r = a.sup[q0 + lane];
                jl = r - i * a.n;
                const int s0 = a.rp[r], s1 = a.rp[r + 1];
                for (int t = s0; t < s1; ++t) vl = fma(a.rv[t], w[a.rk[t]], vl);
            }
            for (int e0 = 0; e0 < cnt; e0 += SPB) {
                int jj[SPB];
                double v[SPB];
#pragma unroll
                for (int u = 0; u < SPB; ++u) {
                    const int e = min(e0 + u, cnt - 1);
                    jj[u] = __shfl(jl, e);
                    v[u] = (e0 + u < cnt) ? __shfl(vl, e) : 0.0;
                }
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int c = 2 * lane + 128 * ch;
                    if (c < a.ld) {
                        double2 y[SPB];
#pragma unroll
                        for (int u = 0; u < SPB; ++u) y[u] = ld2(Yp + (int64_t)jj[u] * a.ld + c);
#pragma unroll
                        for (int u = 0; u < SPB; ++u) { acc[ch].x = fma(v[u], y[u].x, acc[ch].x); acc[ch].y = fma(v[u], y[u].y, acc[ch].y); }
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = 2 * lane + 128 * ch;
            if (c < a.ld) st2(out + (int64_t)i * a.ld + c, make_double2(scale * acc[ch].x, scale * acc[ch].y));
        }
    }
}

// rows: t_i = <S_i, Y_i> where S = sum of slabs (S = M*Y); writes optional S to dst and partial sum -> P[which]
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_rowdot_slabs(Dev d, const double* __restrict__ Yl, const double* slab,
                                                           int64_t slab_stride, int SK, double scale_out,
                                                           double* dst, double* rowdot_out, int which) {
    __shared__ double sh[3 * MSDP_WAVES];
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    double ps = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            double dot = 0.0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    const double2 acc = msdp_sum_slabs(slab, slab_stride, SK, o);
                    const double2 y = ld2(Yl + o);
                    dot += acc.x * y.x + acc.y * y.y;
                    if (dst) st2(dst + o, make_double2(scale_out * acc.x, scale_out * acc.y));
                }
            }
            dot = msdp_group_sum<LPR>(dot);
            // rows of a Euclidean block (multiblock kind) carry no diagonal multiplier: YeG / z stay 0 there
            // (ManiSDP_multiblock.m:80-84,225-228)
            const bool freerow = d.rowfree && d.rowfree[row];
            if (sub == 0) { if (rowdot_out) rowdot_out[row] = freerow ? 0.0 : scale_out * dot; ps += dot; }
        }
    }
    msdp_put_partial(d.P, which, ps, sh);
}

// unitdiag gradient finish: eG (n x ld, = 2*eS*Y) in Gr, YeG rows known -> G = eG - Y.*YeG, |G|^2; and f.
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_obl_grad_finish(Dev d, int slot, double sigma, const double* f_given) {
    __shared__ double sh[3 * MSDP_WAVES + 8];
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const double* __restrict__ Yl = slot ? d.Y[1] : d.Y[0];
    double* __restrict__ Gr = slot ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ YeG = slot ? d.eG[1] : d.eG[0];
    double pgg = 0.0;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            const double t = YeG[row];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int col = 2 * sub + ch * 2 * LPR;
                if (col < d.ld) {
                    const int64_t o = (int64_t)row * d.ld + col;
                    const double2 e = ld2(Gr + o), y = ld2(Yl + o);
                    const double2 gq = make_double2(e.x - y.x * t, e.y - y.y * t);
                    st2(Gr + o, gq);
                    pgg += gq.x * gq.x + gq.y * gq.y;
                }
            }
        }
    }
    // f = c'x + .5*sigma*|Axb|^2 : c'x partial sums are in P_S1 (sum <C*Y, Y>), |Axb|^2 in P_AXB
    const double cx = msdp_sum_partials_block(d.P, P_S1, d.G, sh);
    __syncthreads();
    const double ss = msdp_sum_partials_block(d.P, P_AXB, MSDP_MAX_GRID, sh);
    __syncthreads();
    double pf = 0.0;
    // f_given: the dual kind's cost.  Row-sharded runs: the value enters the all-reduced sum once (rank 0)
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.row0 == 0) pf = f_given ? *f_given : cx + 0.5 * sigma * ss;
    msdp_put_partials3(d.P, P_F, pf, P_GG, pgg, -1, 0.0, sh + 8);
}

// sphere gradient finish: Gr holds 2*eS*Y, z = <eS*Y, Y> partials in P_S2: G = 2 eS Y - 2 z Y.
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_grad_finish(Dev d, int slot, double sigma) {
    __shared__ double sh[3 * MSDP_WAVES + 8];
    // Euclidean manifold (generic ManiSDP.m:153-156): G = 2*S*Y, no projection term
    const double z = (d.manifold == MANI_EUCLID) ? 0.0 : msdp_sum_partials_block(d.P, P_S2, d.G, sh);
    __syncthreads();
    const double cx = msdp_sum_partials_block(d.P, P_S1, d.G, sh);
    __syncthreads();
    const double ss = msdp_sum_partials_block(d.P, P_AXB, MSDP_MAX_GRID, sh);
    __syncthreads();
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ Yl = slot ? d.Y[1] : d.Y[0];
    double* __restrict__ Gr = slot ? d.Gr[1] : d.Gr[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    double pgg = 0.0;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 e = ld2(Gr + i), y = ld2(Yl + i);
        const double2 gq = make_double2(e.x - 2.0 * z * y.x, e.y - 2.0 * z * y.y);
        st2(Gr + i, gq);
        pgg += gq.x * gq.x + gq.y * gq.y;
    }
    double pf = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (d.row0 == 0) pf = cx + 0.5 * sigma * ss; d.ctl->z_sphere[slot] = z; }
    msdp_put_partials3(d.P, P_F, pf, P_GG, pgg, -1, 0.0, sh + 8);
}

// sphere Hess-vec finish (ManiSDP_unittrace.m:176): Hmd holds H_raw, t = <H_raw, Y> partials in P_AUX
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_hess_finish(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES + 8];
    if (!d.F[0].active) return;
    // Euclidean manifold (generic ManiSDP.m:158-162): H = 2*S*U + 4*sigma*AyU*Y as is
    const bool euc = d.manifold == MANI_EUCLID;
    const double t = euc ? 0.0 : msdp_sum_partials_block(d.P, P_AUX, d.G, sh);
    __syncthreads();
    const int cur = d.ctl->cur;
    const double z = euc ? 0.0 : d.ctl->z_sphere[cur];
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    double pd = 0.0;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 hr = ld2(d.Hmd + i), y = ld2(Yl + i), u = ld2(d.md + i);
        const double2 hq = make_double2(hr.x - t * y.x - 2.0 * z * u.x, hr.y - t * y.y - 2.0 * z * u.y);
        st2(d.Hmd + i, hq);
        pd += u.x * hq.x + u.y * hq.y;
    }
    msdp_put_partial(d.P, P_DHD, pd, sh + 8);
}

__global__ void k_cost_only(Dev d, double sigma, double* out) {
    __shared__ double sh[8];
    const double cx = msdp_sum_partials_block(d.P, P_S1, d.G, sh);
    __syncthreads();
    const double ss = msdp_sum_partials_block(d.P, P_AXB, MSDP_MAX_GRID, sh);
    if (threadIdx.x == 0) *out = cx + 0.5 * sigma * ss;
}

// ------------------------------------------------------------------ host side
// ================================================================== multiblock kind, per-block storage (round 4)
// ManiSDP_multiblock.m keeps X, S, C as cell arrays of blocks.  Rounds 2-3 embedded the direct sum into ONE dense N x N problem
// (N = sum n_i): memory and work ~ N^2 -- 3.6 GB per operand at N = 21 100, impossible for thousands of small cliques.  Here
// every dense operand (c, eS, A'(w), S) is the concatenation of its diagonal blocks, block i an n_i x nS_i row-major array
// (nS_i = roundup(n_i, 16), zero pad columns): memory and work ~ sum n_i^2.  Row r of the direct sum lives at rbase[r]; its block
// spans the rows [rlo[r], rhi[r]).  A(.) stays the SDDMM over (i, j) pairs; A'(.) walks the stored positions (CSR by position);
// the contraction is one MFMA wave per 16-row tile of a block, no split-K, straight into slab 0 of the usual epilogues.
struct BlockedDev {
    int nb, N, ntile;
    int64_t etot;                  // stored entries: sum n_i * nS_i
    const int64_t* rbase;          // N: offset of the storage row of global row r
    const int* rlo; const int* rhi; const int* rns;   // N: first / one-past-last row of r's block, its padded order
    const int* prp;                // etot + 1: CSR by stored position -> (constraint, coefficient)
    const int* prk; const double* prv;
    int nlongq; const int* longq;  // stored positions that occur in more than ADJB_LONG constraints (the entry of the monomial 1: in every 'x_i^2 = 1' row of its block): one wave each
    const int* tile_row0;          // ntile: first row of every 16-row tile (tiles never straddle blocks)
};
struct BlockOp { const double* M[2]; const double* X[2]; double scale[2]; int nmat, ld, colofs, ncols; double* out; };

// W_b = Ya_b * Yb_b' for every block, in the per-block storage (row r of block b: W[rbase[r] + (c - rlo[r])]): the Gram route of
// A(Ya Yb') for many blocks -- one 8-byte gather per nonzero of At (k_gram_apply) instead of two panel rows of ld doubles
// (k_sddmm1: 6.8 GB through the L2 per call at ld = 128 for the 100 cliques of example_bqp_sparse.m, 795 us).  One wave per
// 16-row tile, v_mfma_f64_16x16x4_f64 over the ld columns, column tile after column tile of the block.
typedef double blkg_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_block_gram(BlockedDev bd, const double* __restrict__ Ya, const double* __restrict__ Yb, int ld,
                                                    double* __restrict__ W, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= bd.ntile) return;
    const int row0 = bd.tile_row0[tile];
    const int lo = bd.rlo[row0], hi = bd.rhi[row0];
    const int ra = min(row0 + i, hi - 1);
    const double* __restrict__ arow = Ya + (int64_t)ra * ld;
    for (int c0 = lo; c0 < hi; c0 += 16) {
        const int cb = min(c0 + i, hi - 1);
        const double* __restrict__ brow = Yb + (int64_t)cb * ld;
        blkg_d4 acc = (blkg_d4){0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < ld; k0 += 4) {
            const int kk = k0 + g;
            const double av = kk < ld ? arow[kk] : 0.0;               // A operand: lane (g, i) = Ya[row0 + i][k0 + g]
            const double bv = kk < ld ? brow[kk] : 0.0;               // B operand: lane (g, i) = Yb[c0 + i][k0 + g]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        const int col = c0 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                 // D: row g + 4r, column i
            const int row = row0 + g + 4 * r;
            if (row < hi && col < hi) W[bd.rbase[row] + (col - lo)] = acc[r];
        }
    }
}

// out[q] = (base ? base[q] : 0) + scale * sum_k At[(position q), k] * vec[k]   over all stored positions
#define ADJB_LONG 16
__global__ __launch_bounds__(256) void k_adjoint_blocked(BlockedDev bd, const double* __restrict__ base, const double* __restrict__ vec,
                                                         double scale, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    // A position that occurs in hundreds of constraints (the (1, 1) entry of a moment block: every `x_i^2 = 1` row) would keep ONE
    // thread in a chain of dependent gathers for the whole launch -- 362 us per call on the 100 cliques of example_bqp_sparse.m,
    // half of a Hess-vec, whatever the order or layout of the short positions: those go to a wave each (workgroups behind the
    // first `gmain`), lanes striding over the list.
    const int gmain = (int)gridDim.x - (bd.nlongq + 3) / 4;
    if ((int)blockIdx.x >= gmain) {
        const int lane = threadIdx.x & 63;
        const int lq = ((int)blockIdx.x - gmain) * 4 + (threadIdx.x >> 6);
        if (lq >= bd.nlongq) return;
        const int64_t q = bd.longq[lq];
        const int s0 = bd.prp[q], s1 = bd.prp[q + 1];
        double acc = 0.0;
        for (int t = s0 + lane; t < s1; t += 64) acc = fma(bd.prv[t], vec[bd.prk[t]], acc);
        acc = msdp_wave_sum(acc);
        if (lane == 0) out[q] = (base ? base[q] : 0.0) + scale * acc;
        return;
    }
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < bd.etot; q += (int64_t)gmain * blockDim.x) {
        const int s0 = bd.prp[q], s1 = bd.prp[q + 1];
        if (s1 - s0 > ADJB_LONG) continue;
        double acc = 0.0;
        for (int t = s0; t < s1; ++t) acc = fma(bd.prv[t], vec[bd.prk[t]], acc);
        out[q] = (base ? base[q] : 0.0) + scale * acc;
    }
}
typedef double blk_d4 __attribute__((ext_vector_type(4)));
// out(rows of a tile, :) = sum_m scale_m * M_m(block rows, block columns) * X_m(block rows, :): one wave per tile, A fragments
// straight from the block's storage rows, B fragments straight from the panel rows (they stay in the L1 / L2 of the CU: a block's
// panel is n_i x p doubles)
template <int NT>
__global__ __launch_bounds__(256) void k_block_contract(BlockedDev bd, BlockOp op, const int* active_flag) {
    if (active_flag && !*active_flag) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= bd.ntile) return;
    const int row0 = bd.tile_row0[tile];
    const int lo = bd.rlo[row0], hi = bd.rhi[row0], ns = bd.rns[row0];
    const int64_t abase = bd.rbase[min(row0 + i, hi - 1)] + 4 * g;
    blk_d4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (blk_d4){0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < op.nmat; ++m) {
        const double* __restrict__ Mm = op.M[m] + abase;
        const double* __restrict__ Xm = op.X[m];
        const double sc = op.scale[m];
        for (int k = 0; k < ns; k += 16) {
            const double2 a01 = ld2(Mm + k), a23 = ld2(Mm + k + 2);
            const double av[4] = {a01.x * sc, a01.y * sc, a23.x * sc, a23.y * sc};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xr = lo + k + 4 * g + t;
                const bool ok = xr < hi;                          // pad columns of the block hold zeros; keep the read inside the block
                const double* xp = Xm + (int64_t)(ok ? xr : lo) * op.ld + op.colofs + i;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const double bv = (ok && 16 * nt + i < op.ncols) ? xp[16 * nt] : 0.0;
                    acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv, acc[nt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + g + 4 * r;
            if (row < hi && col < op.ncols) op.out[(int64_t)row * op.ld + op.colofs + col] = acc[nt][r];
        }
    }
}
__global__ void k_sub_diag_blocked(BlockedDev bd, double* __restrict__ S, const double* __restrict__ zrow) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bd.N; i += gridDim.x * blockDim.x) S[bd.rbase[i] + (i - bd.rlo[i])] -= zrow[i];
}

struct AffineState {
    AffineDev a{};
    int64_t nnz = 0;
    double sigma = 1.0;
    double* Cdense = nullptr;      // n x nS
    double* d_y = nullptr;
    struct DualState* dual = nullptr;   // MSDP_KIND_DUAL_UNITDIAG (below)
    BlockedDev* blk = nullptr;          // multiblock kind with per-block storage (msdp_affine_setup_blocked); host copies of the block offsets:
    std::vector<int64_t> blk_r0, blk_off; std::vector<int> blk_n, blk_ns;
    // second stream of the Hess-vec: 2*eS*U does not depend on the A(.) / A'(.) chain and runs beside it (msdp_affine_hess)
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~AffineState() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (s2) (void)hipStreamDestroy(s2);
        delete blk;
    }
};
static void msdp_dual_release(struct DualState* ds);
static std::vector<std::pair<msdp_handle, AffineState*>> g_aff;
static AffineState* astate(msdp_handle h) {
    for (auto& pr : g_aff) if (pr.first == h) return pr.second;
    return nullptr;
}
void msdp_affine_release(msdp_handle h) {
    for (size_t i = 0; i < g_aff.size(); ++i)
        if (g_aff[i].first == h) { msdp_dual_release(g_aff[i].second->dual); delete g_aff[i].second; g_aff.erase(g_aff.begin() + i); return; }
}

template <typename T>
static int up(msdp_handle h, const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    int rc = msdp_dev_alloc_bytes(h, &p, v.size() * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) HIPCHK(msdp_memcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return 0;
}

int msdp_affine_setup(msdp_handle h, const int64_t* jc, const int64_t* ir, const double* pr, const double* b,
                      const double* c) {
    Dev& d = h->d;
    const int n = d.n;
    const int64_t m = d.m;
    const int64_t nnz = jc[m];
    if (nnz > 0x7fffffff) { msdp_set_error("nnz(At) too large"); return MSDP_EINVAL; }
    AffineState* st = new AffineState();
    g_aff.push_back({h, st});
    AffineDev& a = st->a;
    a.n = n; a.nS = msdp_dense_nS(n); a.m = m;
    st->nnz = nnz;
    std::vector<int> cjc(m + 1), ci(nnz), cj(nnz);
    std::vector<double> cv(pr, pr + nnz);
    for (int64_t k = 0; k <= m; ++k) cjc[k] = (int)jc[k];
    const int64_t nn = (int64_t)n * n;
    if ((int64_t)n * msdp_dense_nS(n) > 0x7fffffffLL) { msdp_set_error("n too large for the affine kinds"); return MSDP_EUNSUPPORTED; }
    std::vector<int> rp(nn + 1, 0), cidx(nnz);
    for (int64_t t = 0; t < nnz; ++t) {
        const int64_t e = ir[t];
        if (e < 0 || e >= nn) { msdp_set_error("At row index out of range"); return MSDP_EINVAL; }
        const int i = (int)(e % n), j = (int)(e / n);      // column-major vec index (bqpmom.m:57, example_theta.m:20)
        ci[t] = i; cj[t] = j;
        cidx[t] = i * a.nS + j;
        rp[(int64_t)i * n + j + 1]++;
    }
    for (int64_t r = 0; r < nn; ++r) rp[r + 1] += rp[r];
    std::vector<int> rk(nnz);
    std::vector<double> rv(nnz);
    {
        std::vector<int> fill(rp.begin(), rp.end() - 1);
        for (int64_t k = 0; k < m; ++k)
            for (int64_t t = jc[k]; t < jc[k + 1]; ++t) {
                const int64_t r = (int64_t)ci[t] * n + cj[t];
                const int pos = fill[r]++;
                rk[pos] = (int)k; rv[pos] = pr[t];
            }
    }
    std::vector<int> it0, it1, kit(m + 1);
    for (int64_t k = 0; k < m; ++k) {
        kit[k] = (int)it0.size();
        for (int t = cjc[k]; t < cjc[k + 1]; t += SDDMM_CHUNK) { it0.push_back(t); it1.push_back(std::min(t + SDDMM_CHUNK, cjc[k + 1])); }
    }
    kit[m] = (int)it0.size();
    a.nitems = (int64_t)it0.size();
    int rc;
    if ((rc = up(h, it0, &a.it0)) || (rc = up(h, it1, &a.it1)) || (rc = up(h, kit, &a.kit))) return rc;
    {
        std::vector<int> longk;
        for (int64_t k = 0; k < m; ++k) if (kit[k + 1] - kit[k] > FIN_SHORT) longk.push_back((int)k);
        a.nlong = (int)longk.size();
        if (longk.empty()) longk.push_back(0);
        if ((rc = up(h, longk, &a.longk))) return rc;
    }
    {
        void* pv = nullptr;
        if ((rc = msdp_dev_alloc_bytes(h, &pv, (size_t)std::max<int64_t>(a.nitems, 1) * sizeof(double)))) return rc;
        a.ival = (double*)pv;
    }
    {
        // k_sddmm1: short constraints by id, the items of the long ones with their offsets per long constraint
        std::vector<int> sk, lit0, lit1, lkit;
        for (int64_t k = 0; k < m; ++k) {
            if (kit[k + 1] - kit[k] > FIN_SHORT) {
                lkit.push_back((int)lit0.size());
                for (int q = kit[k]; q < kit[k + 1]; ++q) { lit0.push_back(it0[q]); lit1.push_back(it1[q]); }
            } else sk.push_back((int)k);
        }
        lkit.push_back((int)lit0.size());
        a.nshort = (int)sk.size(); a.nlit = (int)lit0.size();
        if (sk.empty()) sk.push_back(0);
        if (lit0.empty()) { lit0.push_back(0); lit1.push_back(0); }
        if ((rc = up(h, sk, &a.sk)) || (rc = up(h, lit0, &a.lit0)) || (rc = up(h, lit1, &a.lit1)) || (rc = up(h, lkit, &a.lkit))) return rc;
        {
            std::vector<int> us0((size_t)a.nshort + a.nlit + 1, 0), us1(us0.size(), 0), uk(us0.size(), 0);
            for (int u = 0; u < a.nshort; ++u) { us0[u] = cjc[sk[u]]; us1[u] = cjc[sk[u] + 1]; uk[u] = sk[u]; }
            for (int q = 0; q < a.nlit; ++q) { us0[a.nshort + q] = lit0[q]; us1[a.nshort + q] = lit1[q]; uk[a.nshort + q] = -1 - q; }
            if ((rc = up(h, us0, &a.us0)) || (rc = up(h, us1, &a.us1)) || (rc = up(h, uk, &a.uk))) return rc;
        }
        void* pv = nullptr;
        if ((rc = msdp_dev_alloc_bytes(h, &pv, 64))) return rc;
        HIPCHK(hipMemset(pv, 0, 64));
        a.cnt = (unsigned*)pv;
    }
    if ((rc = up(h, cjc, &a.cjc)) || (rc = up(h, ci, &a.ci)) || (rc = up(h, cj, &a.cj)) || (rc = up(h, cv, &a.cv)) ||
        (rc = up(h, rp, &a.rp)) || (rc = up(h, rk, &a.rk)) || (rc = up(h, rv, &a.rv)) || (rc = up(h, cidx, &a.cidx)))
        return rc;
    {
        // tiled upper-triangle arrays for k_adjoint_tiled, only when the data is symmetric entry by entry
        bool sym = true;
        for (int i = 0; i < n && sym; ++i)
            for (int j = i + 1; j < n && sym; ++j) {
                if (c[(size_t)i * n + j] != c[(size_t)j * n + i]) { sym = false; break; }
                const int64_t r = (int64_t)i * n + j, rt = (int64_t)j * n + i;
                const int len = rp[r + 1] - rp[r];
                if (len != rp[rt + 1] - rp[rt]) { sym = false; break; }
                for (int t = 0; t < len; ++t)
                    if (rk[rp[r] + t] != rk[rp[rt] + t] || rv[rp[r] + t] != rv[rp[rt] + t]) { sym = false; break; }
            }
        a.trp = nullptr; a.trk = nullptr; a.trv = nullptr; a.tp_i = nullptr; a.tp_j = nullptr; a.ntp = 0;
        a.lpos = nullptr; a.lmir = nullptr; a.ls0 = nullptr; a.ls1 = nullptr; a.nlong_e = 0;
        a.bW = 0; a.bnlong = 0; a.bidx = nullptr; a.bval = nullptr; a.blong = nullptr; a.Wg = nullptr; a.bpk = nullptr; a.bdict = nullptr;
        a.usym = 0; a.unitems = 0; a.uit0 = a.uit1 = a.ukit = a.ulongk = a.ucidx = a.ucjc = nullptr; a.ucv = nullptr; a.unlong = 0;
        std::vector<int> ucidx_h, ucjc_h;
        std::vector<double> ucv_h;
        if (sym) {
            std::vector<int> ucidx, uit0, uit1, ukit(m + 1), ulongk, ucjc(m + 1);
            std::vector<double> ucv;
            ucidx.reserve((size_t)nnz / 2 + n); ucv.reserve((size_t)nnz / 2 + n);
            for (int64_t k = 0; k < m; ++k) {
                ukit[k] = (int)uit0.size();
                const int first = (int)ucidx.size();
                ucjc[k] = first;
                for (int t = cjc[k]; t < cjc[k + 1]; ++t) {
                    if (ci[t] > cj[t]) continue;
                    ucidx.push_back(ci[t] * a.nS + cj[t]);
                    ucv.push_back(ci[t] == cj[t] ? 0.5 * cv[t] : cv[t]);      // Wsym_ii = 2 W_ii
                }
                const int last = (int)ucidx.size();
                for (int t = first; t < last; t += SDDMM_CHUNK) { uit0.push_back(t); uit1.push_back(std::min(t + SDDMM_CHUNK, last)); }
                if ((int)uit0.size() - ukit[k] > FIN_SHORT) ulongk.push_back((int)k);
            }
            ukit[m] = (int)uit0.size();
            ucjc[m] = (int)ucidx.size();
            a.unitems = (int64_t)uit0.size();
            a.unlong = (int)ulongk.size();
            if (ulongk.empty()) ulongk.push_back(0);
            if (ucidx.empty()) { ucidx.push_back(0); ucv.push_back(0.0); }
            if (uit0.empty()) { uit0.push_back(0); uit1.push_back(0); }
            if ((rc = up(h, ucidx, &a.ucidx)) || (rc = up(h, ucv, &a.ucv)) || (rc = up(h, uit0, &a.uit0)) ||
                (rc = up(h, uit1, &a.uit1)) || (rc = up(h, ukit, &a.ukit)) || (rc = up(h, ulongk, &a.ulongk)) ||
                (rc = up(h, ucjc, &a.ucjc))) return rc;
            a.usym = 1;
            ucidx_h = ucidx; ucjc_h = ucjc; ucv_h = ucv;             // kept for the B route below
            h->dense_symmetric = true;                       // c and every A_k are symmetric: so are eS and A'(w) (msdp_densesym.hip)
        }
        const int ntile = (a.nS + ADJ_T - 1) / ADJ_T;
        if (sym && ntile < 32768) {
            std::vector<short> tpi, tpj;
            std::vector<int> trp;
            std::vector<int> trk;
            std::vector<double> trv;
            trp.reserve((size_t)ntile * (ntile + 1) / 2 * ADJ_T * ADJ_T + 1);
            trk.reserve((size_t)nnz / 2 + n + 16);
            trv.reserve((size_t)nnz / 2 + n + 16);
            // Order of the upper tiles = order of the workgroups of k_adjoint_tiled / k_adjoint_gram.  Workgroups b, b + 8, ... share an
            // XCD (round-robin dispatch, msdp_device.h), and what a tile gathers -- the Gram entries of the constraints its entries
            // occur in -- is local to its tile ROW (BQP d = 60: 1.7 MB of the 13.6-MB Gram matrix per tile row, median): the tile rows
            // are cut into 8 contiguous bands of equal tile count, band x feeds the positions x, x + 8, ...  With the plain row-major
            // order every XCD gathered from the whole matrix: 114 MB fetched by k_adjoint_gram for 22 MB of B and 14 MB of W; banded
            // 87 MB and 23.6 -> 20.4 us.  (Streaming (nt) loads of B on top: 79-85 MB but 23.5 us -- they sit in the gather's
            // dependency chain; not kept.)
            std::vector<std::pair<short, short>> tile_order;
            {
                const int64_t tot = (int64_t)ntile * (ntile + 1) / 2;
                std::vector<std::vector<std::pair<short, short>>> band(8);
                int64_t seen = 0;
                for (int bi = 0; bi < ntile; ++bi) {
                    const int cnt = ntile - bi;
                    const int x = (int)std::min<int64_t>(7, (2 * seen + cnt) * 8 / (2 * tot));
                    for (int bj = bi; bj < ntile; ++bj) band[x].push_back({(short)bi, (short)bj});
                    seen += cnt;
                }
                std::vector<size_t> head(8, 0), tail(8);
                for (int x = 0; x < 8; ++x) tail[x] = band[x].size();
                tile_order.reserve((size_t)tot);
                for (int64_t pos = 0; pos < tot; ++pos) {
                    int x = (int)(pos & 7);
                    if (head[x] < tail[x]) { tile_order.push_back(band[x][head[x]++]); continue; }
                    int lx = 0;                                       // band x is used up: the last tile of the longest remaining band
                    for (int q = 1; q < 8; ++q) if (tail[q] - head[q] > tail[lx] - head[lx]) lx = q;
                    tile_order.push_back(band[lx][--tail[lx]]);
                }
            }
            for (const auto& tb : tile_order) {
                {
                    const int bi = tb.first, bj = tb.second;
                    tpi.push_back((short)bi); tpj.push_back((short)bj);
                    for (int li = 0; li < ADJ_T; ++li)
                        for (int lj = 0; lj < ADJ_T; ++lj) {
                            trp.push_back((int)trk.size());
                            const int i = bi * ADJ_T + li, j = bj * ADJ_T + lj;
                            if (i >= n || j >= n) continue;
                            const int64_t r = (int64_t)i * n + j;
                            for (int t = rp[r]; t < rp[r + 1]; ++t) { trk.push_back(rk[t]); trv.push_back(rv[t]); }
                        }
                }
            }
            trp.push_back((int)trk.size());
            trk.push_back(0); trv.push_back(0.0);                    // padding element (see the kernel)
            a.ntp = (int)tpi.size();
            // long entries (i <= j; a diagonal tile holds both (i,j) and (j,i): keep the upper one, its mirror is stored too)
            std::vector<int> lpos, lmir, ls0, ls1;
            for (size_t tp = 0; tp < tpi.size(); ++tp)
                for (int e = 0; e < ADJ_T * ADJ_T; ++e) {
                    const size_t g = tp * (ADJ_T * ADJ_T) + e;
                    if (trp[g + 1] - trp[g] <= ADJ_LONG) continue;
                    const int i = tpi[tp] * ADJ_T + e / ADJ_T, j = tpj[tp] * ADJ_T + e % ADJ_T;
                    if (i > j) continue;
                    lpos.push_back(i * a.nS + j); lmir.push_back(j * a.nS + i);
                    ls0.push_back(trp[g]); ls1.push_back(trp[g + 1]);
                }
            a.nlong_e = (int)lpos.size();
            if (lpos.empty()) { lpos.push_back(0); lmir.push_back(0); ls0.push_back(0); ls1.push_back(0); }
            if ((rc = up(h, trp, &a.trp)) || (rc = up(h, trk, &a.trk)) || (rc = up(h, trv, &a.trv)) ||
                (rc = up(h, tpi, &a.tp_i)) || (rc = up(h, tpj, &a.tp_j)) || (rc = up(h, lpos, &a.lpos)) ||
                (rc = up(h, lmir, &a.lmir)) || (rc = up(h, ls0, &a.ls0)) || (rc = up(h, ls1, &a.ls1))) return rc;
            // ---- B route (k_adjoint_gram): B[e][e'] = sum_k a_k[e] * c_k[e'] over the upper entries, when every constraint is
            // short (a long one -- a trace row -- would fill B: its square) and At is dense in its rows (the Gram route's case)
            a.bW = 0;
            int maxcol = 0;
            for (int64_t k = 0; k < m; ++k) maxcol = std::max(maxcol, ucjc_h[k + 1] - ucjc_h[k]);
            if (maxcol > 0 && maxcol <= 8 && (int64_t)nnz * 8 >= nn) {
                const size_t TE = (size_t)ADJ_T * ADJ_T;
                std::vector<std::vector<std::pair<int, double>>> rows(tpi.size() * TE);
                std::vector<int> hist(16, 0);
                size_t nonempty = 0;
                std::vector<std::pair<int, double>> acc;
                for (size_t tp = 0; tp < tpi.size(); ++tp)
                    for (size_t e = 0; e < TE; ++e) {
                        const int i = tpi[tp] * ADJ_T + (int)(e / ADJ_T), j = tpj[tp] * ADJ_T + (int)(e % ADJ_T);
                        if (i >= n || j >= n) continue;
                        const int64_t r = (int64_t)std::min(i, j) * n + std::max(i, j);
                        acc.clear();
                        for (int t = rp[r]; t < rp[r + 1]; ++t) {
                            const int k = rk[t];
                            for (int u = ucjc_h[k]; u < ucjc_h[k + 1]; ++u) acc.push_back({ucidx_h[u], rv[t] * ucv_h[u]});
                        }
                        std::sort(acc.begin(), acc.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
                        auto& row = rows[tp * TE + e];
                        for (size_t q = 0; q < acc.size(); ++q) {
                            if (!row.empty() && row.back().first == acc[q].first) row.back().second += acc[q].second;
                            else row.push_back(acc[q]);
                        }
                        if (!row.empty()) { ++nonempty; hist[std::min<size_t>(15, row.size())]++; }
                    }
                int BW = 4;
                { size_t cum = 0; for (int w = 1; w <= 4; ++w) { cum += hist[w]; if (cum * 1000 >= nonempty * 990) { BW = w; break; } } }
                std::vector<int> bidx(tpi.size() * TE * BW, 0);
                std::vector<double> bval(tpi.size() * TE * BW, 0.0);
                std::vector<unsigned char> blong(tpi.size() * TE, 0);
                std::vector<int> blpos, blmir, bls0, bls1, blk;
                std::vector<double> blv;
                for (size_t tp = 0; tp < tpi.size(); ++tp)
                    for (size_t e = 0; e < TE; ++e) {
                        const auto& row = rows[tp * TE + e];
                        if ((int)row.size() <= BW) {
                            for (size_t q = 0; q < row.size(); ++q) {
                                bidx[(tp * BW + q) * TE + e] = row[q].first;
                                bval[(tp * BW + q) * TE + e] = row[q].second;
                            }
                            continue;
                        }
                        blong[tp * TE + e] = 1;
                        const int i = tpi[tp] * ADJ_T + (int)(e / ADJ_T), j = tpj[tp] * ADJ_T + (int)(e % ADJ_T);
                        if (i > j) continue;                         // diagonal tile: the upper copy stores both
                        blpos.push_back(i * a.nS + j); blmir.push_back(j * a.nS + i);
                        bls0.push_back((int)blk.size());
                        for (const auto& pr : row) { blk.push_back(pr.first); blv.push_back(pr.second); }
                        bls1.push_back((int)blk.size());
                    }
                a.bnlong = (int)blpos.size();
                if (blpos.empty()) { blpos.push_back(0); blmir.push_back(0); bls0.push_back(0); bls1.push_back(0); }
                if (blk.empty()) { blk.push_back(0); blv.push_back(0.0); }
                // packed form: position (24 bits) | coefficient code (8 bits) when the data allows it
                a.bpk = nullptr; a.bdict = nullptr;
                {
                    std::vector<double> dict;
                    bool ok = (int64_t)n * a.nS < (1 << 24);
                    std::vector<unsigned> bpk;
                    if (ok) {
                        bpk.resize(bidx.size());
                        for (size_t q = 0; q < bidx.size() && ok; ++q) {
                            size_t c = 0;
                            for (; c < dict.size(); ++c) if (memcmp(&dict[c], &bval[q], sizeof(double)) == 0) break;
                            if (c == dict.size()) { if (dict.size() >= 256) { ok = false; break; } dict.push_back(bval[q]); }
                            bpk[q] = (unsigned)bidx[q] | ((unsigned)c << 24);
                        }
                    }
                    if (ok) {
                        dict.resize(256, 0.0);
                        if ((rc = up(h, bpk, &a.bpk)) || (rc = up(h, dict, &a.bdict))) return rc;
                        bidx.assign(1, 0); bval.assign(1, 0.0);          // the packed arrays replace them on the device
                    }
                }
                if ((rc = up(h, bidx, &a.bidx)) || (rc = up(h, bval, &a.bval)) || (rc = up(h, blong, &a.blong)) ||
                    (rc = up(h, blpos, &a.blpos)) || (rc = up(h, blmir, &a.blmir)) || (rc = up(h, bls0, &a.bls0)) ||
                    (rc = up(h, bls1, &a.bls1)) || (rc = up(h, blk, &a.blk)) || (rc = up(h, blv, &a.blv))) return rc;
                a.bW = BW;
            }
        }
    }
    {
        // entries touched by At; the restricted adjoint is used when they are few (<= 1/8 of the matrix)
        int64_t ns = 0;
        for (int64_t r = 0; r < nn; ++r) ns += rp[r + 1] > rp[r];
        a.sup = nullptr; a.suprow = nullptr; a.nsup = 0; a.sqj = a.sqk = a.sqmore = nullptr; a.sqv = nullptr; a.rkx = nullptr;
        if (ns > 0 && ns * 8 <= nn) {
            std::vector<int> sup;
            sup.reserve((size_t)ns);
            for (int64_t r = 0; r < nn; ++r) if (rp[r + 1] > rp[r]) sup.push_back((int)r);
            std::vector<int> suprow(n + 1, 0);
            for (int r : sup) suprow[r / n + 1]++;
            for (int i = 0; i < n; ++i) suprow[i + 1] += suprow[i];
            if ((rc = up(h, sup, &a.sup)) || (rc = up(h, suprow, &a.suprow))) return rc;
            {
                std::vector<int> sqj(sup.size()), sqk(sup.size()), sqmore(sup.size());
                std::vector<double> sqv(sup.size());
                std::vector<int> longno((size_t)m, -1);
                { int ql = 0; for (int64_t k = 0; k < m; ++k) if (kit[k + 1] - kit[k] > FIN_SHORT) longno[k] = ql++; }
                auto enc = [&](int k) { return longno[k] >= 0 ? -1 - longno[k] : k; };
                for (size_t q = 0; q < sup.size(); ++q) {
                    const int64_t r = sup[q];
                    sqj[q] = (int)(r % n); sqk[q] = enc(rk[rp[r]]); sqv[q] = rv[rp[r]]; sqmore[q] = rp[r + 1] - rp[r] - 1;
                }
                std::vector<int> rkx(rk.size());
                for (size_t t = 0; t < rk.size(); ++t) rkx[t] = enc(rk[t]);
                if (rkx.empty()) rkx.push_back(0);
                if ((rc = up(h, sqj, &a.sqj)) || (rc = up(h, sqk, &a.sqk)) || (rc = up(h, sqv, &a.sqv)) || (rc = up(h, sqmore, &a.sqmore)) ||
                    (rc = up(h, rkx, &a.rkx))) return rc;
            }
            a.nsup = (int)ns;
        }
    }
    std::vector<double> bv(b, b + m);
    if ((rc = up(h, bv, &a.b))) return rc;
    void* p = nullptr;
    if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
    st->d_y = (double*)p; a.y = st->d_y;
    HIPCHK(hipMemset(st->d_y, 0, m * sizeof(double)));
    if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
    a.w = (double*)p;
    for (int s = 0; s < 2; ++s) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
        a.Axb[s] = (double*)p;
    }
    // dense C (n x nS) from the column-major vector c (symmetric)
    const size_t msz = (size_t)n * a.nS * sizeof(double);
    if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
    st->Cdense = (double*)p; d.Cd = st->Cdense;
    HIPCHK(hipMemset(st->Cdense, 0, msz));
    HIPCHK(msdp_memcpy2d(st->Cdense, (size_t)a.nS * sizeof(double), c, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n,
                       hipMemcpyHostToDevice));
    for (int s = 0; s < 2; ++s) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        d.eS[s] = (double*)p;
        // restricted adjoint: eS = C outside the entries At touches, from the start
        if (a.nsup > 0) HIPCHK(msdp_memcpy(d.eS[s], st->Cdense, msz, hipMemcpyDeviceToDevice));
        else HIPCHK(hipMemset(d.eS[s], 0, msz));
    }
    if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
    d.AyU = (double*)p;
    HIPCHK(hipMemset(d.AyU, 0, msz));
    // the Gram scratch may share AyU: W is consumed (k_gram_apply) before the adjoint rewrites AyU, and the cost /
    // line-search calls never touch AyU.  With the restricted adjoint AyU must stay zero outside the entries At
    // touches, so the Gram scratch and the dual slack of msdp_al_dual get buffers of their own.
    a.W = d.AyU;
    d.Sdual = d.AyU;
    if (a.bW > 0) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        a.Wg = (double*)p;
        HIPCHK(hipMemset(a.Wg, 0, msz));
    }
    if (a.nsup > 0) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        a.W = (double*)p;
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        d.Sdual = (double*)p;
        HIPCHK(hipMemset(d.Sdual, 0, msz));
    }
    h->h_ctl->sigma = 1.0;
    return 0;
}

// Set-up of the multiblock kind with per-block storage.  jc / ir / pr: At over the CONCATENATED vecs of the blocks (ir = e0_i + a +
// b*n_i, column-major inside block i); c likewise.  Nothing of size N^2 is built, on the host or on the device.
int msdp_affine_setup_blocked(msdp_handle h, int nb, const int64_t* block_n, const int64_t* jc, const int64_t* ir, const double* pr,
                              const double* b, const double* c) {
    Dev& d = h->d;
    const int N = d.n;
    const int64_t m = d.m, nnz = jc[m];
    if (nnz > 0x7fffffff) { msdp_set_error("nnz(At) too large"); return MSDP_EINVAL; }
    AffineState* st = new AffineState();
    g_aff.push_back({h, st});
    AffineDev& a = st->a;
    memset(&a, 0, sizeof(a));
    a.n = N; a.nS = msdp_dense_nS(N); a.m = m;
    st->nnz = nnz;
    st->blk = new BlockedDev();
    BlockedDev& bd = *st->blk;
    std::vector<int64_t> r0((size_t)nb + 1, 0), e0((size_t)nb + 1, 0), off((size_t)nb + 1, 0);
    std::vector<int> bn(nb), bns(nb);
    for (int i = 0; i < nb; ++i) {
        bn[i] = (int)block_n[i]; bns[i] = msdp_dense_nS(bn[i]);
        r0[i + 1] = r0[i] + bn[i]; e0[i + 1] = e0[i] + (int64_t)bn[i] * bn[i]; off[i + 1] = off[i] + (int64_t)bn[i] * bns[i];
    }
    const int64_t etot = off[nb];
    if (etot > 0x7fffffffLL) { msdp_set_error("multiblock: sum n_i^2 too large"); return MSDP_EUNSUPPORTED; }
    st->blk_r0 = r0; st->blk_off = off; st->blk_n = bn; st->blk_ns = bns;
    std::vector<int64_t> rbase(N);
    std::vector<int> rlo(N), rhi(N), rns(N), tile_row0;
    for (int i = 0; i < nb; ++i) {
        for (int aa = 0; aa < bn[i]; ++aa) {
            const int64_t r = r0[i] + aa;
            rbase[r] = off[i] + (int64_t)aa * bns[i]; rlo[r] = (int)r0[i]; rhi[r] = (int)r0[i + 1]; rns[r] = bns[i];
        }
        for (int t = 0; t < bn[i]; t += 16) tile_row0.push_back((int)r0[i] + t);
    }
    // nonzeros: rows (i, j) of the direct sum, stored position
    std::vector<int> cjc(m + 1), ci(nnz), cj(nnz), pos(nnz);
    std::vector<double> cv(pr, pr + nnz);
    for (int64_t k = 0; k <= m; ++k) cjc[k] = (int)jc[k];
    std::vector<int> prp(etot + 1, 0);
    for (int64_t t = 0; t < nnz; ++t) {
        const int64_t e = ir[t];
        if (e < 0 || e >= e0[nb]) { msdp_set_error("multiblock: At row index out of range"); return MSDP_EINVAL; }
        const int i = (int)(std::upper_bound(e0.begin(), e0.end(), e) - e0.begin()) - 1;
        const int64_t l = e - e0[i];
        const int aa = (int)(l % bn[i]), bb = (int)(l / bn[i]);
        ci[t] = (int)r0[i] + aa; cj[t] = (int)r0[i] + bb;
        pos[t] = (int)(off[i] + (int64_t)aa * bns[i] + bb);
        prp[pos[t] + 1]++;
    }
    for (int64_t q = 0; q < etot; ++q) prp[q + 1] += prp[q];
    std::vector<int> prk(std::max<int64_t>(nnz, 1));
    std::vector<double> prv(std::max<int64_t>(nnz, 1));
    {
        std::vector<int> fill(prp.begin(), prp.end() - 1);
        for (int64_t k = 0; k < m; ++k)
            for (int64_t t = jc[k]; t < jc[k + 1]; ++t) { const int q = fill[pos[t]]++; prk[q] = (int)k; prv[q] = pr[t]; }
    }
    // work items / units of the SDDMM (as msdp_affine_setup)
    std::vector<int> it0, it1, kit(m + 1), longk, sk, lit0, lit1, lkit;
    for (int64_t k = 0; k < m; ++k) {
        kit[k] = (int)it0.size();
        for (int t = cjc[k]; t < cjc[k + 1]; t += SDDMM_CHUNK) { it0.push_back(t); it1.push_back(std::min(t + SDDMM_CHUNK, cjc[k + 1])); }
    }
    kit[m] = (int)it0.size();
    a.nitems = (int64_t)it0.size();
    for (int64_t k = 0; k < m; ++k) {
        if (kit[k + 1] - kit[k] > FIN_SHORT) {
            longk.push_back((int)k);
            lkit.push_back((int)lit0.size());
            for (int q = kit[k]; q < kit[k + 1]; ++q) { lit0.push_back(it0[q]); lit1.push_back(it1[q]); }
        } else sk.push_back((int)k);
    }
    lkit.push_back((int)lit0.size());
    a.nlong = (int)longk.size(); a.nshort = (int)sk.size(); a.nlit = (int)lit0.size();
    std::vector<int> us0((size_t)a.nshort + a.nlit + 1, 0), us1(us0.size(), 0), uk(us0.size(), 0);
    for (int u = 0; u < a.nshort; ++u) { us0[u] = cjc[sk[u]]; us1[u] = cjc[sk[u] + 1]; uk[u] = sk[u]; }
    for (int q = 0; q < a.nlit; ++q) { us0[a.nshort + q] = lit0[q]; us1[a.nshort + q] = lit1[q]; uk[a.nshort + q] = -1 - q; }
    if (longk.empty()) longk.push_back(0);
    if (sk.empty()) sk.push_back(0);
    if (lit0.empty()) { lit0.push_back(0); lit1.push_back(0); }
    if (it0.empty()) { it0.push_back(0); it1.push_back(0); }
    if (pos.empty()) pos.push_back(0);
    std::vector<int> longq;
    for (int64_t q = 0; q < etot; ++q) if (prp[q + 1] - prp[q] > ADJB_LONG) longq.push_back((int)q);
    bd.nlongq = (int)longq.size();
    if (longq.empty()) longq.push_back(0);
    if (ci.empty()) { ci.push_back(0); cj.push_back(0); cv.push_back(0.0); }
    if (pos.empty()) pos.push_back(0);
    int rc;
    if ((rc = up(h, it0, &a.it0)) || (rc = up(h, it1, &a.it1)) || (rc = up(h, kit, &a.kit)) || (rc = up(h, longk, &a.longk)) ||
        (rc = up(h, sk, &a.sk)) || (rc = up(h, lit0, &a.lit0)) || (rc = up(h, lit1, &a.lit1)) || (rc = up(h, lkit, &a.lkit)) ||
        (rc = up(h, us0, &a.us0)) || (rc = up(h, us1, &a.us1)) || (rc = up(h, uk, &a.uk)) ||
        (rc = up(h, cjc, &a.cjc)) || (rc = up(h, ci, &a.ci)) || (rc = up(h, cj, &a.cj)) || (rc = up(h, cv, &a.cv)) ||
        (rc = up(h, rbase, &bd.rbase)) || (rc = up(h, rlo, &bd.rlo)) || (rc = up(h, rhi, &bd.rhi)) || (rc = up(h, rns, &bd.rns)) ||
        (rc = up(h, prp, &bd.prp)) || (rc = up(h, prk, &bd.prk)) || (rc = up(h, prv, &bd.prv)) || (rc = up(h, tile_row0, &bd.tile_row0)) ||
        (rc = up(h, longq, &bd.longq)))
        return rc;
    bd.nb = nb; bd.N = N; bd.ntile = (int)tile_row0.size(); bd.etot = etot;
    void* p = nullptr;
    // Gram route on the blocks (k_block_gram + k_gram_apply): the stored position of every nonzero, and room for W = Ya Yb' block by block
    if ((rc = up(h, pos, &a.cidx))) return rc;
    if ((rc = msdp_dev_alloc_bytes(h, &p, (size_t)std::max<int64_t>(etot, 1) * sizeof(double)))) return rc;
    a.W = (double*)p;
    if ((rc = msdp_dev_alloc_bytes(h, &p, (size_t)std::max<int64_t>(a.nitems, 1) * sizeof(double)))) return rc;
    a.ival = (double*)p;
    if ((rc = msdp_dev_alloc_bytes(h, &p, 64))) return rc;
    HIPCHK(hipMemset(p, 0, 64));
    a.cnt = (unsigned*)p;
    std::vector<double> bv(b, b + m);
    if ((rc = up(h, bv, &a.b))) return rc;
    if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
    st->d_y = (double*)p; a.y = st->d_y;
    HIPCHK(hipMemset(st->d_y, 0, m * sizeof(double)));
    if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
    a.w = (double*)p;
    for (int s2 = 0; s2 < 2; ++s2) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, m * sizeof(double)))) return rc;
        a.Axb[s2] = (double*)p;
    }
    // c, block by block, into the padded storage; eS, AyU start as zeros
    const size_t msz = (size_t)etot * sizeof(double);
    {
        std::vector<double> cb((size_t)etot, 0.0);
        for (int i = 0; i < nb; ++i)
            for (int bb = 0; bb < bn[i]; ++bb)
                for (int aa = 0; aa < bn[i]; ++aa) cb[(size_t)(off[i] + (int64_t)aa * bns[i] + bb)] = c[e0[i] + aa + (int64_t)bb * bn[i]];
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        st->Cdense = (double*)p; d.Cd = st->Cdense;
        HIPCHK(msdp_memcpy(st->Cdense, cb.data(), msz, hipMemcpyHostToDevice));
    }
    for (int s2 = 0; s2 < 2; ++s2) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        d.eS[s2] = (double*)p;
        HIPCHK(hipMemset(d.eS[s2], 0, msz));
    }
    if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
    d.AyU = (double*)p;
    HIPCHK(hipMemset(d.AyU, 0, msz));
    d.Sdual = d.AyU;
    h->blocked = true;
    h->dense_symmetric = false;
    // (a.W: the blocks' own Gram storage, allocated above -- launch_A takes the Gram route on it once the panel is wide enough;
    //  the N x N routes of use_gram_route / the B route never apply to this storage)
    h->h_ctl->sigma = 1.0;
    return 0;
}
// One diagonal block of the dual slack (per-block storage): rows row0 .. row0 + nbk - 1 must be exactly one block
int msdp_affine_get_block(msdp_handle h, int64_t row0, int64_t nbk, double* S) {
    AffineState* st = astate(h);
    if (!st || !st->blk) { msdp_set_error("get_block: not a handle with per-block storage"); return MSDP_ESTATE; }
    for (size_t i = 0; i + 1 < st->blk_r0.size(); ++i)
        if (st->blk_r0[i] == row0 && st->blk_n[i] == nbk) {
            HIPCHK(msdp_memcpy2d_async(S, (size_t)nbk * sizeof(double), h->d.Sdual + st->blk_off[i], (size_t)st->blk_ns[i] * sizeof(double),
                                    (size_t)nbk * sizeof(double), (size_t)nbk, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            return 0;
        }
    msdp_set_error("get_dual_slack_block: rows %lld..%lld are not one block of this handle", (long long)row0, (long long)(row0 + nbk));
    return MSDP_EINVAL;
}

// Where block (row0, n) of the per-block storage lives in d.Sdual (msdp_blockjacobi.hip)
int msdp_affine_block_source(msdp_handle h, int64_t row0, int64_t n, int64_t* off, int64_t* ld) {
    AffineState* st = astate(h);
    if (!st || !st->blk) { msdp_set_error("block_source: not a handle with per-block storage"); return MSDP_ESTATE; }
    for (size_t i = 0; i + 1 < st->blk_r0.size(); ++i)
        if (st->blk_r0[i] == row0 && st->blk_n[i] == n) { *off = st->blk_off[i]; *ld = st->blk_ns[i]; return 0; }
    msdp_set_error("block_eigs: rows %lld..%lld are not one block of this handle", (long long)row0, (long long)(row0 + n));
    return MSDP_EINVAL;
}

int msdp_affine_set_multipliers(msdp_handle h, const double* y, double sigma) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    if (!(sigma > 0)) { msdp_set_error("sigma must be positive"); return MSDP_EINVAL; }
    HIPCHK(msdp_memcpy_async(st->d_y, y, st->a.m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    st->sigma = sigma;
    h->h_ctl->sigma = sigma;
    return 0;
}

#define DISPATCH_LPR_A(KERNEL, h, GRID, ...)                                                         \
    do {                                                                                             \
        int half = (h)->d.ld / 2, lpr = 1;                                                           \
        while (lpr < half && lpr < 64) lpr <<= 1;                                                    \
        int nch = (half + lpr - 1) / lpr; if (nch < 1) nch = 1;                                      \
        dim3 grid(GRID), block(MSDP_BLOCK);                                                          \
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

int msdp_dense_hess_epilogue_obl(msdp_handle h, const double* slab, int64_t stride, int SK);   // msdp_dense.hip
int msdp_dense_gemm_slabs(msdp_handle h, int nmat);
int msdp_dense_gemm_at(msdp_handle h, hipStream_t stream, int slab_first, int slabs_reserve, int nmat, const double* const* M,
                       const double* const* X, const double* scale, const int* active_flag, const double** slab_out,
                       int64_t* stride_out, int* SK_out);
int msdp_sphere_hess_raw(msdp_handle h, const double* slab, int64_t stride, int SK);           // below
int msdp_dense_gemm_side(msdp_handle h, const double* M, const double* X, double scale, const int* active_flag, SideJob sj,
                         int* njobs_out, const double** slab_out, int64_t* stride_out, int* SK_out);   // msdp_dense.hip

// k_sddmm has no reductions, so its grid follows the number of work items, not the number of rows
static int sddmm_grid(const AffineDev& a, int ld) {
    int half = ld / 2, lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    const int64_t per_block = (int64_t)MSDP_WAVES * (64 / lpr);
    int64_t g = (a.nitems + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

// A(Ya Yb') -> item values.  SDDMM (gathers 2 p-wide rows per nonzero) or the Gram route (dense W = Ya*Yb' once,
// one double per nonzero), whichever moves fewer bytes; the option affine_route (msdp_set_option) overrides.
static bool use_gram_route(msdp_handle h, const AffineDev& a, int64_t nnz, int ld) {
    if (AffineState* stb = astate(h)) if (stb->blk) return false;                 // per-block storage: there is no N x N Gram matrix
    if (h->tune.affine_route) return h->tune.affine_route == 2;
    const double sddmm_bytes = (double)nnz * ld * 16.0;
    const double gram_bytes = 2.0 * a.n * (double)a.nS * 8.0 + (double)nnz * 20.0;
    return sddmm_bytes > 4.0 * gram_bytes && (double)a.n * a.nS * 8.0 <= 2.0e9;
}
// The constraints restricted to their entries i <= j (symmetric data): what the Gram route and the finish kernel
// that follows it iterate over.
static void upper_view(AffineDev& a) {
    a.nitems = a.unitems; a.it0 = a.uit0; a.it1 = a.uit1; a.kit = a.ukit; a.longk = a.ulongk; a.nlong = a.unlong;
    a.cidx = a.ucidx; a.cv = a.ucv; a.cjc = a.ucjc;
}
// w = A(Ya Yb') (mode 0), or additionally Axb = w - b - y/sigma into axb_out and the partial sums of Axb^2 -> P_AXB
// (mode 1: exactly MSDP_MAX_GRID workgroups, the count the consumers re-reduce).  `a` is the caller's copy: on the
// symmetric Gram route it is switched to the upper view.
static bool sharded(msdp_handle h);
// mode 2 (k_sddmm1 only, see there): *G2_out = the number of workgroups whose partial sums P_T1..P_T3 carry
static bool sddmm1_ok(msdp_handle h, const AffineDev& a, int64_t nnz) {
    return h->tune.affine_fuse && !sharded(h) && !use_gram_route(h, a, nnz, a.ld);
}
static int launch_A(msdp_handle h, AffineDev& a, int64_t nnz, const double* Ya, const double* Yb, const int* flag, int when,
                    int mode, double* axb_out, double sigma, int* G2_out = nullptr) {
    int64_t gm = (a.m + MSDP_BLOCK - 1) / MSDP_BLOCK;           // mode 0: no reduction, size the grid by m
    if (gm > 2048) gm = 2048;
    const int G = mode == 1 ? MSDP_MAX_GRID : (int)gm;
    if (AffineState* stb = astate(h)) {
        // per-block storage: the Gram route on the blocks' own storage once the panel is wide enough for the row gathers of the
        // SDDMM to cost four times the Gram matrix (the rule of use_gram_route, with sum n_i^2 in the place of n^2)
        if (stb->blk && a.W && a.cidx && mode != 2 && h->tune.affine_route != 1 &&
            ((double)nnz * a.ld * 16.0 > 4.0 * (2.0 * (double)stb->blk->etot * 8.0 + (double)nnz * 20.0) || h->tune.affine_route == 2)) {
            hipLaunchKernelGGL(k_block_gram, dim3((stb->blk->ntile + 3) / 4), dim3(256), 0, h->stream, *stb->blk, Ya, Yb, a.ld, a.W, flag, when);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_gram_apply, dim3(G), dim3(MSDP_BLOCK), 0, h->stream, a, (const double*)a.W, mode, axb_out, sigma,
                               h->d.P, flag, when);
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
    if (sddmm1_ok(h, a, nnz)) {
        int half = a.ld / 2, lpr = 1;
        while (lpr < half && lpr < 64) lpr <<= 1;
        const int64_t per_block = (int64_t)MSDP_WAVES * (64 / lpr);
        int64_t g1 = ((int64_t)a.nshort + a.nlit + per_block - 1) / per_block;
        if (g1 < 1) g1 = 1;
        if (g1 > MSDP_MAX_GRID - 1) g1 = MSDP_MAX_GRID - 1;
        if (mode == 1) g1 = MSDP_MAX_GRID - 1;
        if (mode == 2 && g1 > h->d.n_loc) g1 = std::max(1, h->d.n_loc);
        if (G2_out) *G2_out = (int)g1;
        // workgroups that hold items of long constraints: unit u belongs to workgroup (u / per_block) mod g1
        int nbl = 0;
        if (a.nlit > 0) {
            const int64_t b0 = (int64_t)a.nshort / per_block, b1 = ((int64_t)a.nshort + a.nlit - 1) / per_block;
            nbl = (int)std::min<int64_t>(g1, b1 - b0 + 1);
        }
        DISPATCH_LPR_A(k_sddmm1, h, (int)g1, a, h->d, Ya, Yb, mode, axb_out, sigma, flag, when, nbl);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (mode == 2) { msdp_set_error("launch_A: mode 2 needs the fused SDDMM"); return MSDP_ESTATE; }
    if (use_gram_route(h, a, nnz, a.ld)) {
        dim3 grid((a.nS + 63) / 64, (a.n + 63) / 64);
        const int sym = a.usym ? 1 : 0;
        if (sym) upper_view(a);
        hipLaunchKernelGGL(k_gram_mfma, grid, dim3(512), 0, h->stream, a.n, a.nS, a.ld, Ya, Yb, a.W, flag, when, sym);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_gram_apply, dim3(G), dim3(MSDP_BLOCK), 0, h->stream, a, (const double*)a.W, mode, axb_out, sigma,
                           h->d.P, flag, when);
        HIPCHK(hipGetLastError());
        return 0;
    }
    DISPATCH_LPR_A(k_sddmm, h, sddmm_grid(a, a.ld), a, Ya, Yb, flag, when);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_sddmm_finish, dim3(G), dim3(MSDP_BLOCK), 0, h->stream, a, mode, axb_out, sigma, h->d.P, flag, when);
    HIPCHK(hipGetLastError());
    return 0;
}

static int adjoint_grid(const AffineDev& a) {
    int64_t tot = (int64_t)a.n * a.nS;
    int64_t g = (tot + 255) / 256;
    if (g > 4096) g = 4096;
    return (int)g;
}

// out = base + scale * At*vec.  `restricted_ok`: `out` already holds base (or 0) outside the entries At touches.
static int launch_adjoint(msdp_handle h, const AffineDev& a, const double* base, const double* vec, double scale, double* out,
                          const int* flag, int when, bool restricted_ok) {
    if (AffineState* stb = astate(h)) if (stb->blk) {
        const int64_t g = std::min<int64_t>(4096, (stb->blk->etot + 255) / 256) + (stb->blk->nlongq + 3) / 4;
        hipLaunchKernelGGL(k_adjoint_blocked, dim3((int)g), dim3(256), 0, h->stream, *stb->blk, base, vec, scale, out, flag, when);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (restricted_ok && a.nsup > 0) {
        int g = (a.nsup + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_adjoint_support, dim3(g), dim3(256), 0, h->stream, a, base, vec, scale, out, flag, when);
    } else if (a.ntp > 0) {
        hipLaunchKernelGGL(k_adjoint_tiled, dim3(a.ntp + (a.nlong_e + 3) / 4), dim3(256), 0, h->stream, a, base, vec, scale, out, flag, when);
    } else {
        hipLaunchKernelGGL(k_adjoint_dense, dim3(adjoint_grid(a)), dim3(256), 0, h->stream, a, base, vec, scale, out, flag, when);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// slab <- scale * (A'(vec) restricted to the touched entries) * Yp   (k_support_spmm; a.nsup > 0, ld <= 512)
static int launch_support_spmm(msdp_handle h, const AffineDev& a, const double* vec, const double* Yp, double scale,
                               double* out, const int* flag, int when) {
    const int nch = (a.ld + 127) / 128;
    const dim3 g(std::min((a.n + 3) / 4, 2048)), b(256);
    switch (nch) {
        case 1: hipLaunchKernelGGL(k_support_spmm<1>, g, b, 0, h->stream, a, vec, Yp, scale, out, flag, when); break;
        case 2: hipLaunchKernelGGL(k_support_spmm<2>, g, b, 0, h->stream, a, vec, Yp, scale, out, flag, when); break;
        case 3: hipLaunchKernelGGL(k_support_spmm<3>, g, b, 0, h->stream, a, vec, Yp, scale, out, flag, when); break;
        default: hipLaunchKernelGGL(k_support_spmm<4>, g, b, 0, h->stream, a, vec, Yp, scale, out, flag, when); break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}


// The dense products of the closures: the MFMA contraction of msdp_dense.hip, or -- multiblock kind with per-block storage -- one
// wave per 16-row tile of a block, written to slab 0 (SK = 1)
int msdp_dense_ensure_slab(msdp_handle h, size_t need);
static int affine_gemm(msdp_handle h, int nmat, const double* const* M, const double* const* X, const double* scale, const int* active_flag,
                       const double** slab_out, int64_t* stride_out, int* SK_out) {
    AffineState* st = astate(h);
    if (!st || !st->blk) return msdp_dense_gemm(h, nmat, M, X, scale, active_flag, slab_out, stride_out, SK_out);
    Dev& d = h->d;
    const int64_t stride = (int64_t)d.n * d.ld;
    int rc = msdp_dense_ensure_slab(h, (size_t)2 * stride);
    if (rc) return rc;
    BlockOp op;
    memset(&op, 0, sizeof(op));
    op.nmat = nmat; op.ld = d.ld; op.out = h->slab;
    for (int m = 0; m < nmat; ++m) { op.M[m] = M[m]; op.X[m] = X[m]; op.scale[m] = scale[m]; }
    const dim3 grid((st->blk->ntile + 3) / 4), block(256);
    for (int colofs = 0; colofs < d.ld; colofs += 128) {
        op.colofs = colofs; op.ncols = std::min(128, d.ld - colofs);
        switch ((op.ncols + 15) / 16) {
            case 1: hipLaunchKernelGGL(k_block_contract<1>, grid, block, 0, h->stream, *st->blk, op, active_flag); break;
            case 2: hipLaunchKernelGGL(k_block_contract<2>, grid, block, 0, h->stream, *st->blk, op, active_flag); break;
            case 3: hipLaunchKernelGGL(k_block_contract<3>, grid, block, 0, h->stream, *st->blk, op, active_flag); break;
            case 4: hipLaunchKernelGGL(k_block_contract<4>, grid, block, 0, h->stream, *st->blk, op, active_flag); break;
            default: hipLaunchKernelGGL(k_block_contract<8>, grid, block, 0, h->stream, *st->blk, op, active_flag); break;
        }
    }
    HIPCHK(hipGetLastError());
    *slab_out = h->slab; *stride_out = stride; *SK_out = 1;
    return 0;
}
// cost + gradient state at Y[slot]:  w = A(YY'), Axb, eS, eS*Y, C*Y  (see header comment)
// ------------------------------------------------------------------ Row sharding of the affine kinds
// The rows of Y, U, G, H are split over the ranks as for the other kinds; the OPERATOR state is replicated: every rank
// holds At, the dense eS / AyU / S and computes A(Ya Yb'), Axb and the adjoint for the whole matrix itself -- the same
// kernels on the same inputs give the same bits on every rank, so no m-vector travels (for BQP d = 60 an all-reduce of
// the 9-MB A(U Y') per Hess-vec would cost more than recomputing it: 17 us).  What is shared is the dense contraction
// (each rank multiplies ITS rows of eS / AyU with the gathered panel) and every row-parallel kernel; their partial sums
// are all-reduced like those of the other kinds.  The operators read all rows of the point: yfull[slot] keeps the
// gathered copy of Y[slot] (the Hess-vec needs it next to the gathered direction).
static bool sharded(msdp_handle h) { return h->use_comm || h->nranks > 1; }   // (declared above launch_A)
// All n rows of Y[slot].  `gathered`: d.full holds them right now (the caller's all-gather) -> refresh the copy.
static int full_rows(msdp_handle h, int slot, const double* local, bool gathered, const double** out) {
    if (!sharded(h)) { *out = local; return 0; }
    if (!h->yfull[slot]) { msdp_set_error("row-sharded affine handle without gather buffers"); return MSDP_ESTATE; }
    if (!gathered) { int rc = msdp_allgather_rows(h, local); if (rc) return rc; }
    const size_t cap = (size_t)((h->d.n + h->nranks - 1) / h->nranks);
    HIPCHK(msdp_memcpy_async(h->yfull[slot], h->d.full, cap * h->nranks * (size_t)h->d.ld * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    *out = h->yfull[slot];
    return 0;
}
static const double* kept_rows(msdp_handle h, int slot) { return sharded(h) ? h->yfull[slot] : h->d.Y[slot]; }

static int dual_costgrad(msdp_handle h, AffineState* st, int slot);
static int dual_hess(msdp_handle h, AffineState* st);
static int dual_linesearch_cost(msdp_handle h, AffineState* st, const double* Yt, double* val);

int msdp_affine_costgrad(msdp_handle h, int slot) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    if (st->dual) return dual_costgrad(h, st, slot);
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    const double* Ys = d.Y[slot];                         // my rows
    const double* Yf = nullptr;                           // all rows (the caller, msdp_launch_costgrad, has gathered them)
    { int rcf = full_rows(h, slot, Ys, true, &Yf); if (rcf) return rcf; }
    const size_t roff = (size_t)d.row0 * a.nS;            // my rows of the replicated n x nS matrices
    const int* done = &d.ctl->done;
    { int rc0 = launch_A(h, a, st->nnz, Yf, Yf, done, 1, 1, a.Axb[slot], sigma); if (rc0) return rc0; }
    { int rca = launch_adjoint(h, a, d.Cd, a.Axb[slot], sigma, d.eS[slot], done, 1, true); if (rca) return rca; }
    // c'x = <C*Y, Y>
    const double* slab; int64_t stride; int SK;
    {
        const double* M[1] = {d.Cd + roff}; const double* X[1] = {Yf}; const double sc[1] = {1.0};
        int rc = affine_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK);
        if (rc) return rc;
        DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 1.0, (double*)nullptr, (double*)nullptr, P_S1);
        HIPCHK(hipGetLastError());
    }
    // eG = 2*eS*Y -> Gr[slot]; row dots (YeG = sum(Y.*eG)) and their total (2z)
    if (a.nsup > 0 && !sharded(h) && d.ld <= 512) {
        // At touches few entries: eS*Y = C*Y (the slabs just computed) + sigma * (A'(Axb) on those entries) * Y,
        // appended as one more slab -- one dense product per cost/gradient evaluation instead of two
        double* extra = const_cast<double*>(slab) + (int64_t)SK * stride;
        int rc = launch_support_spmm(h, a, (const double*)a.Axb[slot], Ys, sigma, extra, done, 1);
        if (rc) return rc;
        ++SK;
        DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 2.0, d.Gr[slot], d.eG[slot], P_S2);
        HIPCHK(hipGetLastError());
    } else {
        const double* M[1] = {d.eS[slot] + roff}; const double* X[1] = {Yf}; const double sc[1] = {1.0};
        int rc = affine_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK);
        if (rc) return rc;
        DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 2.0, d.Gr[slot], d.eG[slot], P_S2);
        HIPCHK(hipGetLastError());
    }
    { int rcr = msdp_allreduce_partials(h, P_S1, 2); if (rcr) return rcr; }      // <C*Y, Y> and the row dots over all ranks
    if (d.manifold == MANI_OBLIQUE) {
        DISPATCH_LPR_A(k_obl_grad_finish, h, d.G, d, slot, sigma, (const double*)nullptr);
    } else {
        hipLaunchKernelGGL(k_sph_grad_finish, dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, d, slot, sigma);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// Algorithmic traffic / work of one Hess-vec of the affine kinds (SURVEY.md 8d, A-operator): At is read twice (w = A(Y U'),
// A'(w): values + indices, 12 bytes per nonzero each), A'(w) is written and read once over the entries At touches (all n^2
// of them for BQP, 55 276 of 25 M for the theta-like problem), eS is read once, the three n x p panels once each.
void msdp_affine_algo_cost(msdp_handle h, double* bytes, double* flops) {
    AffineState* st = astate(h);
    *bytes = 0.0; *flops = 0.0;
    if (!st || st->dual) return;
    const AffineDev& a = st->a;
    const double n = a.n, p = h->d.p, nnz = (double)st->nnz;
    const double touched = a.nsup > 0 ? (double)a.nsup : n * n;
    *bytes = 2.0 * nnz * 12.0 + 16.0 * touched + 8.0 * n * n + 24.0 * n * p;
    *flops = 2.0 * n * n * p + 2.0 * touched * p + 2.0 * nnz * p + 4.0 * nnz;
}

int msdp_affine_hess(msdp_handle h) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    if (st->dual) return dual_hess(h, st);
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    const int cur = h->h_ctl->cur;
    const int* act = &d.F[0].active;
    // all rows of the point (kept since its cost/gradient evaluation) and of the direction (gathered by msdp_launch_hess)
    const double* Yf = kept_rows(h, cur);
    const double* Uf = sharded(h) ? (const double*)d.full : (const double*)d.md;
    const size_t roff = (size_t)d.row0 * a.nS;
    const double* slab; int64_t stride; int SK;
    int rc;
    if (h->tune.affine_overlap) {
        // A/B switch, default OFF.  Two branches of the Hess-vec are independent until the epilogue sums their slabs: 2*eS*U (one
        // dense contraction) and the chain w = A(Y U') -> A'(w) -> 4 sigma A'(w)*Y.  Here they run side by side on two streams
        // (fork / join by events; inside a captured chunk the second stream joins the capture).  Measured in round 3: slower than
        // one stream (BQP d = 60: 85 against 73 us, theta n = 5000: 83 against 74 us) -- each launch of the chain fills the chip by
        // itself, the split contraction loses its two-matrix launch, and the fork / join adds graph dependencies.
        if (!st->s2) {
            HIPCHK(hipStreamCreateWithFlags(&st->s2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
        }
        const int sk1 = msdp_dense_gemm_slabs(h, 1);
        const bool support = a.nsup > 0 && d.ld <= 512 && !sharded(h);
        const int total = support ? sk1 + 1 : 2 * sk1;
        int SKa = 0, SKb = 0;
        HIPCHK(hipEventRecord(st->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(st->s2, st->ev_fork, 0));
        {
            const double* M[1] = {d.eS[cur] + roff}; const double* X[1] = {Uf}; const double sc[1] = {2.0};
            if ((rc = msdp_dense_gemm_at(h, st->s2, 0, total, 1, M, X, sc, act, &slab, &stride, &SKa))) return rc;
        }
        HIPCHK(hipEventRecord(st->ev_join, st->s2));
        if ((rc = launch_A(h, a, st->nnz, Yf, Uf, act, 0, 0, (double*)nullptr, sigma))) return rc;
        if (support) {
            double* extra = const_cast<double*>(slab) + (int64_t)SKa * stride;
            if ((rc = launch_support_spmm(h, a, (const double*)a.w, (const double*)d.Y[cur], 4.0 * sigma, extra, act, 0))) return rc;
            SKb = 1;
        } else {
            if ((rc = launch_adjoint(h, a, (const double*)nullptr, a.w, 1.0, d.AyU, act, 0, true))) return rc;
            const double* M[1] = {d.AyU + roff}; const double* X[1] = {Yf}; const double sc[1] = {4.0 * sigma};
            const double* slab2;
            if ((rc = msdp_dense_gemm_at(h, h->stream, SKa, total, 1, M, X, sc, act, &slab2, &stride, &SKb))) return rc;
        }
        HIPCHK(hipStreamWaitEvent(h->stream, st->ev_join, 0));
        SK = SKa + SKb;
        if (d.manifold == MANI_OBLIQUE) return msdp_dense_hess_epilogue_obl(h, slab, stride, SK);
        return msdp_sphere_hess_raw(h, slab, stride, SK);
    }
    if (a.bW > 0 && h->tune.affine_broute && !sharded(h) && !(a.nsup > 0 && d.ld <= 512) && use_gram_route(h, a, st->nnz, a.ld)) {
        // B route: AyU = B * (Y U' + U Y') in one sparse pass over the upper entries (k_adjoint_gram) -- no w, At read once
        dim3 grid((a.nS + 63) / 64, (a.n + 63) / 64);
        hipLaunchKernelGGL(k_gram_mfma, grid, dim3(512), 0, h->stream, a.n, a.nS, a.ld, Yf, Uf, a.Wg, act, 0, 1);
        HIPCHK(hipGetLastError());
        const dim3 ga(a.ntp + (a.bnlong + 3) / 4), ba(256);
        const bool pk = a.bpk != nullptr;
#define AG(BW) do { if (pk) hipLaunchKernelGGL((k_adjoint_gram<BW, true>), ga, ba, 0, h->stream, a, (const double*)a.Wg, 1.0, d.AyU, act, 0); \
                    else hipLaunchKernelGGL((k_adjoint_gram<BW, false>), ga, ba, 0, h->stream, a, (const double*)a.Wg, 1.0, d.AyU, act, 0); } while (0)
        switch (a.bW) { case 1: AG(1); break; case 2: AG(2); break; case 3: AG(3); break; default: AG(4); break; }
#undef AG
        HIPCHK(hipGetLastError());
        const double* M[2] = {d.eS[cur] + roff, d.AyU + roff};
        const double* X[2] = {Uf, Yf};
        const double sc[2] = {2.0, 4.0 * sigma};
        if ((rc = affine_gemm(h, 2, M, X, sc, act, &slab, &stride, &SK))) return rc;
        if (d.manifold == MANI_OBLIQUE) return msdp_dense_hess_epilogue_obl(h, slab, stride, SK);
        return msdp_sphere_hess_raw(h, slab, stride, SK);
    }
    if (d.manifold != MANI_OBLIQUE && a.usym && d.ld <= 512 && sddmm1_ok(h, a, st->nnz)) {
        // sphere / Euclidean factor, SDDMM route, one rank: three launches + the contraction (k_sddmm1 mode 2, [adjoint,]
        // contraction, k_sph_hess_fused) instead of six
        int G2 = 0, hetero = 0;
        const bool support = a.nsup > 0;
        if (support && h->tune.affine_overlap) {
            // A/B (option affine_overlap): 2*eS*U needs neither w nor the sums -- it runs on a second stream beside k_sddmm1 and joins in
            // front of the epilogue
            if (!st->s2) {
                HIPCHK(hipStreamCreateWithFlags(&st->s2, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
            }
            const double* M[1] = {d.eS[cur]}; const double* X[1] = {d.md}; const double sc[1] = {2.0};
            HIPCHK(hipEventRecord(st->ev_fork, h->stream));
            HIPCHK(hipStreamWaitEvent(st->s2, st->ev_fork, 0));
            if ((rc = msdp_dense_gemm_at(h, st->s2, 0, 0, 1, M, X, sc, act, &slab, &stride, &SK))) return rc;
            HIPCHK(hipEventRecord(st->ev_join, st->s2));
            if ((rc = launch_A(h, a, st->nnz, Yf, Uf, act, 0, 2, (double*)nullptr, sigma, &G2))) return rc;
            HIPCHK(hipStreamWaitEvent(h->stream, st->ev_join, 0));
        } else if (support) {
            // default: the SDDMM rides in the contraction launch as a side job (msdp_dense_gemm_side) -- where the shape allows it
            rc = MSDP_EUNSUPPORTED;
            if (h->tune.affine_side && a.nlong <= MSDP_WAVES) {
                SideJob sj;
                memset(&sj, 0, sizeof(sj));
                sj.a = a; sj.Ya = Yf; sj.Yb = Uf; sj.Gr = d.Gr[cur]; sj.axc = a.Axb[cur]; sj.P = d.P; sj.sigma = sigma;
                rc = msdp_dense_gemm_side(h, d.eS[cur], d.md, 2.0, act, sj, &G2, &slab, &stride, &SK);
                if (rc == 0) hetero = 1;
                else if (rc != MSDP_EUNSUPPORTED) return rc;
            }
            if (!hetero) {
                if ((rc = launch_A(h, a, st->nnz, Yf, Uf, act, 0, 2, (double*)nullptr, sigma, &G2))) return rc;
                const double* M[1] = {d.eS[cur]}; const double* X[1] = {d.md}; const double sc[1] = {2.0};
                if ((rc = affine_gemm(h, 1, M, X, sc, act, &slab, &stride, &SK))) return rc;
            }
        } else {
            if ((rc = launch_A(h, a, st->nnz, Yf, Uf, act, 0, 2, (double*)nullptr, sigma, &G2))) return rc;
            if ((rc = launch_adjoint(h, a, (const double*)nullptr, a.w, 1.0, d.AyU, act, 0, true))) return rc;
            const double* M[2] = {d.eS[cur], d.AyU}; const double* X[2] = {Uf, Yf}; const double sc[2] = {2.0, 4.0 * sigma};
            if ((rc = affine_gemm(h, 2, M, X, sc, act, &slab, &stride, &SK))) return rc;
        }
        DISPATCH_LPR_A(k_sph_hess_fused, h, d.G, d, a, slab, stride, SK, sigma, G2, support ? 1 : 0, cur, hetero);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // w = A(Y U') ; AyU = A'(w)
    { int rc0 = launch_A(h, a, st->nnz, Yf, Uf, act, 0, 0, (double*)nullptr, sigma); if (rc0) return rc0; }
    if (a.nsup > 0 && !sharded(h) && d.ld <= 512) {
        // At touches few entries: AyU*Y is a sparse product over those entries (appended as one more slab); only
        // 2*eS*U goes through the dense contraction
        const double* M[1] = {d.eS[cur]};
        const double* X[1] = {d.md};
        const double sc[1] = {2.0};
        if ((rc = affine_gemm(h, 1, M, X, sc, act, &slab, &stride, &SK))) return rc;
        double* extra = const_cast<double*>(slab) + (int64_t)SK * stride;
        if ((rc = launch_support_spmm(h, a, (const double*)a.w, (const double*)d.Y[cur], 4.0 * sigma, extra, act, 0))) return rc;
        ++SK;
    } else {
        { int rca = launch_adjoint(h, a, (const double*)nullptr, a.w, 1.0, d.AyU, act, 0, true); if (rca) return rca; }
        const double* M[2] = {d.eS[cur] + roff, d.AyU + roff};
        const double* X[2] = {Uf, Yf};
        const double sc[2] = {2.0, 4.0 * sigma};
        if ((rc = affine_gemm(h, 2, M, X, sc, act, &slab, &stride, &SK))) return rc;
    }
    if (d.manifold == MANI_OBLIQUE) return msdp_dense_hess_epilogue_obl(h, slab, stride, SK);
    return msdp_sphere_hess_raw(h, slab, stride, SK);
}

// sphere: H_raw = sum slabs -> Hmd, partial <H_raw, Y> -> P_AUX; then the finish kernel.
template <int LPR, int NCH>
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_hess_raw(Dev d, const double* slab, int64_t slab_stride, int SK) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (!d.F[0].active) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int cur = d.ctl->cur;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    double pt = 0.0;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 acc = msdp_sum_slabs(slab, slab_stride, SK, i);
        const double2 y = ld2(Yl + i);
        st2(d.Hmd + i, acc);
        pt += acc.x * y.x + acc.y * y.y;
    }
    msdp_put_partial(d.P, P_AUX, pt, sh);
}

int msdp_sphere_hess_raw(msdp_handle h, const double* slab, int64_t stride, int SK) {
    Dev& d = h->d;
    hipLaunchKernelGGL((k_sph_hess_raw<1, 1>), dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, d, slab, stride, SK);
    HIPCHK(hipGetLastError());
    { int rcr = msdp_allreduce_partials(h, P_AUX, 1); if (rcr) return rcr; }     // <H_raw, Y> over all ranks
    hipLaunchKernelGGL(k_sph_hess_finish, dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, d);
    HIPCHK(hipGetLastError());
    return 0;
}

// co(Y) of the line search at the trial point Yt (device, n x ld): c'x + sigma/2 |A x - b - y/sigma|^2
int msdp_affine_linesearch_cost(msdp_handle h, const double* Yt, double* val) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    if (st->dual) return dual_linesearch_cost(h, st, Yt, val);
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    const int other = h->h_ctl->cur ^ 1;
    const double* Yf = nullptr;
    { int rcf = full_rows(h, other, Yt, false, &Yf); if (rcf) return rcf; }
    { int rc0 = launch_A(h, a, st->nnz, Yf, Yf, (const int*)nullptr, 0, 1, a.Axb[other], sigma); if (rc0) return rc0; }
    const double* slab; int64_t stride; int SK;
    const double* M[1] = {d.Cd + (size_t)d.row0 * a.nS}; const double* X[1] = {Yf}; const double sc[1] = {1.0};
    int rc = affine_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK);
    if (rc) return rc;
    DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Yt, slab, stride, SK, 1.0, (double*)nullptr, (double*)nullptr, P_S1);
    HIPCHK(hipGetLastError());
    if ((rc = msdp_allreduce_partials(h, P_S1, 1))) return rc;
    hipLaunchKernelGGL(k_cost_only, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, d, sigma, &d.ctl->fx_prop);
    HIPCHK(hipGetLastError());
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, &d.ctl->fx_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *val = v;
    return 0;
}

// ------------------------------------------------------------------ AL bookkeeping on the device
// The host AL loop of the affine entry points needs, once per outer iteration (ManiSDP_unitdiag.m:59-70,
// ManiSDP_unittrace.m:59-70, ManiSDP.m:58-68): obj = c'x and Axb = A x - b at the new point, then with the NEW
// multipliers eS = reshape(c - At*y), z = sum(X.*eS) and S = eS - diag(z) (unit diagonal) / S = eS - z*I
// (unit trace) / S = eS (generic).  On the host that is n^2-sized NumPy/SciPy work (X = YY', two SpMVs with
// 4.8 M nonzeros, the dense eS): 49% of the BQP d = 60 solve.  Here the same quantities come from the kernels
// of the hot path: A(YY') (SDDMM or Gram route), the dense adjoint, the MFMA contraction and a row-dot.

// rows: S[i][j] -= (i == j) * (zrow ? zrow[i] : *zscalar)
__global__ void k_sub_diag(int n, int nS, double* __restrict__ S, const double* __restrict__ zrow, const double* __restrict__ zscalar) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        S[(int64_t)i * nS + i] -= zrow ? zrow[i] : *zscalar;
}

// obj = c'x, w = A x (m doubles to the host; the caller subtracts b)
int msdp_affine_al_primal(msdp_handle h, double* obj, double* Ax_host) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const int cur = h->h_ctl->cur;
    const double* Ys = d.Y[cur];
    const double* Yf = nullptr;
    int rc = full_rows(h, cur, Ys, false, &Yf);
    if (rc) return rc;
    if ((rc = launch_A(h, a, st->nnz, Yf, Yf, (const int*)nullptr, 0, 0, (double*)nullptr, 1.0))) return rc;
    const double* slab; int64_t stride; int SK;
    const double* M[1] = {d.Cd + (size_t)d.row0 * a.nS}; const double* X[1] = {Yf}; const double sc[1] = {1.0};
    if ((rc = affine_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK))) return rc;
    DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 1.0, (double*)nullptr, (double*)nullptr, P_S1);
    HIPCHK(hipGetLastError());
    if ((rc = msdp_allreduce_partials(h, P_S1, 1))) return rc;
    if ((rc = msdp_k_sum_to_fwd(h, P_S1, &d.ctl->fx_prop))) return rc;
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, &d.ctl->fx_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(msdp_memcpy_async(Ax_host, a.w, (size_t)a.m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *obj = v;
    return 0;
}

// eS = c - At*y, z, S (left in d.Sdual for msdp_escape_eigs_dual); z -> host (n values, or 1 for the sphere, none for generic)
int msdp_affine_al_dual(msdp_handle h, const double* y_host, double* z_host) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const int cur = h->h_ctl->cur;
    const double* Ys = d.Y[cur];
    HIPCHK(msdp_memcpy_async(a.w, y_host, (size_t)a.m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // full sweep: the diagonal of Sdual was modified by k_sub_diag after the previous call
    { int rca = launch_adjoint(h, a, d.Cd, (const double*)a.w, -1.0, d.Sdual, (const int*)nullptr, 0, false); if (rca) return rca; }
    if (d.manifold == MANI_EUCLID) { HIPCHK(hipStreamSynchronize(h->stream)); return 0; }
    // t_i = <(eS*Y)_i, Y_i>  (= sum(X.*eS) per row); their total for the sphere
    const double* Yf = nullptr;
    int rc = full_rows(h, cur, Ys, false, &Yf);          // also what the escape deflates against (msdp_escape.hip)
    if (rc) return rc;
    const double* slab; int64_t stride; int SK;
    const double* M[1] = {d.Sdual + (size_t)d.row0 * a.nS}; const double* X[1] = {Yf}; const double sc[1] = {1.0};
    if ((rc = affine_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK))) return rc;
    DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 1.0, (double*)nullptr, d.W0, P_S2);
    HIPCHK(hipGetLastError());
    if (d.manifold == MANI_OBLIQUE) {
        const double* zall = d.W0;
        if (sharded(h)) {
            // z of all rows: one all-gather of the cap-long row blocks (the gather buffer is free here)
            if (!h->use_comm) { msdp_set_error("al_dual on a communicator-free shard"); return MSDP_ESTATE; }
            const size_t cap = (size_t)((d.n + h->nranks - 1) / h->nranks);
            if ((rc = msdp_allgather_vec(h, d.W0, h->full_buf, cap))) return rc;
            zall = h->full_buf;
        }
        if (st->blk) hipLaunchKernelGGL(k_sub_diag_blocked, dim3((a.n + 255) / 256), dim3(256), 0, h->stream, *st->blk, d.Sdual, zall);
        else hipLaunchKernelGGL(k_sub_diag, dim3((a.n + 255) / 256), dim3(256), 0, h->stream, a.n, a.nS, d.Sdual, zall, (const double*)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(msdp_memcpy_async(z_host, zall, (size_t)a.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        if ((rc = msdp_allreduce_partials(h, P_S2, 1))) return rc;
        if ((rc = msdp_k_sum_to_fwd(h, P_S2, &d.ctl->fx_prop))) return rc;
        hipLaunchKernelGGL(k_sub_diag, dim3((a.n + 255) / 256), dim3(256), 0, h->stream, a.n, a.nS, d.Sdual, (const double*)nullptr, (const double*)&d.ctl->fx_prop);
        HIPCHK(hipGetLastError());
        HIPCHK(msdp_memcpy_async(z_host, &d.ctl->fx_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}


// ================================================================== dual, unit diagonal
// src/dual/ManiDSDP_unitdiag.m: the variable is the dual slack S = Y'Y with diag(S) = 1 (oblique factor, as in the
// primal unit-diagonal entry point); the multipliers are the primal matrix x (n^2, dense) and the free part w.
//   cost  :174-181   S = Y'Y; sc = S(:) - c; y = iA'*sc; As = A'y - sc - x/sigma; Af = B'y - cf - w/sigma;
//                    f = b'y + sigma/2 (|As|^2 + |Af|^2)
//   grad  :183-187   X = reshape(bA - sigma*As); eG = 2*Y*X; G = eG - Y.*sum(Y.*eG)
//   hess  :189-194   yAU = reshape(A'(iA'*vec(Y'U))); eH = 2*U*X - 4*sigma*(Y*yAU) + 2*sigma*((Y*U')*Y + (Y*Y')*U)
// with iA = (diag(A*A')\A)' (:38) and bA = iA*b (:39).  The rows of A are the columns of this handle's At, so
// iA'*vec(M) is the primal kind's A(M) divided by dAAt, and A'y its adjoint: launch_A / launch_adjoint and the MFMA
// contraction are reused as they are.  With T = bA + x - sigma*C (rebuilt when the multipliers change):
//   X = T + sigma*S - sigma*A'y,      sigma*As = bA - X.
// New here: the dense Gram S = Y*Y' (k_gram_mfma, one product), the two p x p Gram matrices of the last Hessian
// term, and the element-wise kernels.
struct DualState {
    int nf = 0;                     // free variables (K.f)
    const double* dinv = nullptr;   // 1 ./ dAAt                         (m)
    const double* Ac = nullptr;     // A*c                                (m)
    const int* bjc = nullptr;       // B in CSC (m x nf)
    const int* bir = nullptr;
    const double* bpr = nullptr;
    const double* cf = nullptr;     // nf
    double* wf = nullptr;           // free multipliers w                 (nf)
    double* Af = nullptr;           // Af of the last cost evaluation     (nf)
    double* x = nullptr;            // multiplier matrix x                (n x nS)
    double* bA = nullptr;           // reshape(iA*b)                      (n x nS)
    double* T = nullptr;            // bA + x - sigma*C                   (n x nS)
    double* Sg = nullptr;           // S = Y*Y'                           (n x nS)
    double* G2[2] = {nullptr, nullptr};   // Y'*Y per slot                (ld x ld)
    double* M1 = nullptr;           // U'*Y of the current Hess-vec       (ld x ld)
    double* pp_part = nullptr;      // DUAL_PP_BLOCKS x ld x ld partials
    double* scal = nullptr;         // [0] f, [1] b'y, [2] <C,eX>, [3] |As|^2
    bool T_valid = false;
};
static void msdp_dual_release(DualState* ds) { delete ds; }
#define DUAL_PP_BLOCKS 64
#define DUAL_PP_MAXLD 128

// y = (A(S) - A c) ./ dAAt in place, partial sums of b'y -> P_S1   (grid d.G)
__global__ __launch_bounds__(MSDP_BLOCK) void k_dual_y(int64_t m, double* __restrict__ w, const double* __restrict__ dinv,
                                                       const double* __restrict__ Ac, const double* __restrict__ b, double* P,
                                                       const int* skip_flag, int skip_when) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (skip_flag && *skip_flag == skip_when) return;
    double pb = 0.0;
    for (int64_t k = blockIdx.x * (int64_t)MSDP_BLOCK + threadIdx.x; k < m; k += (int64_t)gridDim.x * MSDP_BLOCK) {
        const double y = (w[k] - Ac[k]) * dinv[k];
        w[k] = y;
        pb = fma(b[k], y, pb);
    }
    msdp_put_partial(P, P_S1, pb, sh);
}
// w .*= dinv (Hess-vec: iA'*vec(Y'U))
__global__ void k_dual_scale(int64_t m, double* __restrict__ w, const double* __restrict__ dinv, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < m; k += (int64_t)gridDim.x * blockDim.x) w[k] *= dinv[k];
}
// Af_j = B(:,j)'y - cf_j - (wf ? wf_j / sigma : 0): one workgroup per free variable
__global__ __launch_bounds__(256) void k_dual_free(const int* __restrict__ bjc, const int* __restrict__ bir, const double* __restrict__ bpr,
                                                   const double* __restrict__ y, const double* __restrict__ cf, const double* wf,
                                                   double sigma, double* __restrict__ Af, const int* skip_flag, int skip_when) {
    __shared__ double sh[MSDP_WAVES];
    if (skip_flag && *skip_flag == skip_when) return;
    const int j = blockIdx.x;
    double v = 0.0;
    for (int t = bjc[j] + threadIdx.x; t < bjc[j + 1]; t += blockDim.x) v = fma(bpr[t], y[bir[t]], v);
    v = msdp_block_sum(v, sh);
    if (threadIdx.x == 0) Af[j] = v - cf[j] - (wf ? wf[j] / sigma : 0.0);
}
// X += sigma*S, and the partial sums of |bA - X|^2 (= sigma^2 |As|^2) -> P_AXB   (MSDP_MAX_GRID workgroups)
__global__ __launch_bounds__(256) void k_dual_finish_X(int64_t tot, double* __restrict__ X, const double* __restrict__ S,
                                                       const double* __restrict__ bA, double sigma, double* P,
                                                       const int* skip_flag, int skip_when) {
    __shared__ double sh[3 * MSDP_WAVES];
    if (skip_flag && *skip_flag == skip_when) return;
    double ps = 0.0;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const double xv = fma(sigma, S[e], X[e]);
        X[e] = xv;
        const double r = bA[e] - xv;
        ps = fma(r, r, ps);
    }
    msdp_put_partial(P, P_AXB, ps, sh);
}
// f = b'y + |bA - X|^2 / (2 sigma) + sigma/2 |Af|^2   (one workgroup)
__global__ __launch_bounds__(MSDP_BLOCK) void k_dual_cost(Dev d, double sigma, const double* __restrict__ Af, int nf, double* out,
                                                          const int* skip_flag, int skip_when) {
    __shared__ double sh[8];
    if (skip_flag && *skip_flag == skip_when) return;
    const double by = msdp_sum_partials_block(d.P, P_S1, d.G, sh);
    __syncthreads();
    const double ss = msdp_sum_partials_block(d.P, P_AXB, MSDP_MAX_GRID, sh);
    if (threadIdx.x == 0) {
        double af = 0.0;
        for (int j = 0; j < nf; ++j) af = fma(Af[j], Af[j], af);
        out[0] = by + 0.5 * ss / sigma + 0.5 * sigma * af;
        out[1] = by;
    }
}
// T = bA + x - sigma*C
__global__ void k_dual_T(int64_t tot, double* __restrict__ T, const double* __restrict__ bA, const double* __restrict__ x,
                         const double* __restrict__ C, double sigma) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x)
        T[e] = bA[e] + x[e] - sigma * C[e];
}
// P = Xa' * Xb (ld x ld) from two n x ld panels: per-workgroup partials over a row range, then their sum
__global__ __launch_bounds__(256) void k_pp_gram_part(int n, int ld, const double* __restrict__ Xa, const double* __restrict__ Xb,
                                                      double* __restrict__ part, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int rows = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows, r1 = min(n, r0 + rows);
    for (int e = threadIdx.x; e < ld * ld; e += blockDim.x) {
        const int a = e / ld, b = e - a * ld;
        double acc = 0.0;
        for (int k = r0; k < r1; ++k) acc = fma(Xa[(int64_t)k * ld + a], Xb[(int64_t)k * ld + b], acc);
        part[(int64_t)blockIdx.x * ld * ld + e] = acc;
    }
}
__global__ void k_pp_gram_sum(int ld, int nblk, const double* __restrict__ part, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ld * ld; e += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int q = 0; q < nblk; ++q) acc += part[(int64_t)q * ld * ld + e];
        out[e] = acc;
    }
}
// out(i,:) = coef * (Y(i,:)*M1 + U(i,:)*G2): the 2*sigma*((Y*U')*Y + (Y*Y')*U) term of :192, one more split-K slab
__global__ void k_pp_apply(int n_loc, int ld, const double* __restrict__ Y, const double* __restrict__ U, const double* __restrict__ M1,
                           const double* __restrict__ G2, double coef, double* __restrict__ out, const int* skip_flag, int skip_when) {
    if (skip_flag && *skip_flag == skip_when) return;
    const int64_t tot = (int64_t)n_loc * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / ld; const int c = (int)(e - i * ld);
        double acc = 0.0;
        for (int a = 0; a < ld; ++a) acc = fma(Y[i * ld + a], M1[a * ld + c], fma(U[i * ld + a], G2[a * ld + c], acc));
        out[e] = coef * acc;
    }
}
// Outer step :73-81 on the rows of a workgroup: As = (C + A'y) - S (in Xd), x -= sigma*As, eX = x + bA,
// z_i = sum_j S_ij eX_ij, Xd = eX - diag(z); partial sums of <C, eX> -> P_S2 and |As|^2 -> P_S3   (grid d.G)
__global__ __launch_bounds__(MSDP_BLOCK) void k_dual_outer(Dev d, int nS, double* __restrict__ Xd, const double* __restrict__ S,
                                                           double* __restrict__ x, const double* __restrict__ bA,
                                                           const double* __restrict__ C, double sigma, double* __restrict__ z) {
    __shared__ double sh[3 * MSDP_WAVES];
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double pc = 0.0, pa = 0.0;
    for (int row = lo + wave; row < hi; row += MSDP_WAVES) {
        const int64_t o = (int64_t)row * nS;
        double zr = 0.0, exd = 0.0;
        for (int j = lane; j < d.n; j += 64) {
            const double as = Xd[o + j] - S[o + j];
            const double xn = x[o + j] - sigma * as;
            const double ex = xn + bA[o + j];
            x[o + j] = xn;
            Xd[o + j] = ex;
            zr = fma(S[o + j], ex, zr);
            pc = fma(C[o + j], ex, pc);
            pa = fma(as, as, pa);
            if (j == row) exd = ex;
        }
        zr = msdp_wave_sum(zr);
        if (lane == (row & 63)) { Xd[o + row] = exd - zr; z[row] = zr; }
    }
    msdp_put_partials3(d.P, P_S2, pc, P_S3, pa, -1, 0.0, sh);
}

static int dual_pp_gram(msdp_handle h, DualState* ds, const double* Xa, const double* Xb, double* out, const int* flag, int when) {
    const Dev& d = h->d;
    hipLaunchKernelGGL(k_pp_gram_part, dim3(DUAL_PP_BLOCKS), dim3(256), 0, h->stream, d.n, d.ld, Xa, Xb, ds->pp_part, flag, when);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_pp_gram_sum, dim3((d.ld * d.ld + 255) / 256), dim3(256), 0, h->stream, d.ld, DUAL_PP_BLOCKS,
                       (const double*)ds->pp_part, out, flag, when);
    HIPCHK(hipGetLastError());
    return 0;
}
static int dual_check(msdp_handle h, DualState* ds) {
    if (h->d.ld > DUAL_PP_MAXLD) { msdp_set_error("dual kind: factor width p = %d exceeds the supported maximum of %d", h->d.p, DUAL_PP_MAXLD); return MSDP_EUNSUPPORTED; }
    if (!ds->T_valid) { msdp_set_error("dual kind: call msdp_dual_set_penalty after msdp_dual_outer_step"); return MSDP_ESTATE; }
    return 0;
}
// steps shared by cost/grad and the line-search cost: y, Af, S, X = T + sigma*S - sigma*A'y into Xout, f -> scal[0]
static int dual_cost_state(msdp_handle h, AffineState* st, const double* Ys, double* Xout, const int* flag, int when) {
    DualState* ds = st->dual;
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    int rc;
    if ((rc = launch_A(h, a, st->nnz, Ys, Ys, flag, when, 0, (double*)nullptr, sigma))) return rc;
    hipLaunchKernelGGL(k_dual_y, dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, a.m, a.w, ds->dinv, ds->Ac, a.b, d.P, flag, when);
    HIPCHK(hipGetLastError());
    if (ds->nf > 0) {
        hipLaunchKernelGGL(k_dual_free, dim3(ds->nf), dim3(256), 0, h->stream, ds->bjc, ds->bir, ds->bpr, (const double*)a.w, ds->cf,
                           (const double*)ds->wf, sigma, ds->Af, flag, when);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_gram_mfma, dim3((a.nS + 63) / 64, (a.n + 63) / 64), dim3(512), 0, h->stream, a.n, a.nS, a.ld, Ys, Ys, ds->Sg, flag, when, 0);
    HIPCHK(hipGetLastError());
    if ((rc = launch_adjoint(h, a, (const double*)ds->T, (const double*)a.w, -sigma, Xout, flag, when, false))) return rc;
    hipLaunchKernelGGL(k_dual_finish_X, dim3(MSDP_MAX_GRID), dim3(256), 0, h->stream, (int64_t)a.n * a.nS, Xout, (const double*)ds->Sg,
                       (const double*)ds->bA, sigma, d.P, flag, when);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_dual_cost, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, d, sigma, (const double*)ds->Af, ds->nf, ds->scal, flag, when);
    HIPCHK(hipGetLastError());
    return 0;
}

static int dual_costgrad(msdp_handle h, AffineState* st, int slot) {
    DualState* ds = st->dual;
    Dev& d = h->d;
    int rc;
    if ((rc = dual_check(h, ds))) return rc;
    const double* Ys = d.Y[slot];
    const int* done = &d.ctl->done;
    if ((rc = dual_cost_state(h, st, Ys, d.eS[slot], done, 1))) return rc;
    // eG = 2*X*Y -> Gr[slot], row dots YeG
    const double* slab; int64_t stride; int SK;
    const double* M[1] = {d.eS[slot]}; const double* X[1] = {Ys}; const double sc[1] = {1.0};
    if ((rc = msdp_dense_gemm(h, 1, M, X, sc, nullptr, &slab, &stride, &SK))) return rc;
    DISPATCH_LPR_A(k_rowdot_slabs, h, d.G, d, Ys, slab, stride, SK, 2.0, d.Gr[slot], d.eG[slot], P_S2);
    HIPCHK(hipGetLastError());
    DISPATCH_LPR_A(k_obl_grad_finish, h, d.G, d, slot, st->sigma, (const double*)ds->scal);
    HIPCHK(hipGetLastError());
    return dual_pp_gram(h, ds, Ys, Ys, ds->G2[slot], done, 1);
}

static int dual_hess(msdp_handle h, AffineState* st) {
    DualState* ds = st->dual;
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    const int cur = h->h_ctl->cur;
    const int* act = &d.F[0].active;
    int rc;
    if ((rc = dual_check(h, ds))) return rc;
    if ((rc = launch_A(h, a, st->nnz, d.Y[cur], d.md, act, 0, 0, (double*)nullptr, sigma))) return rc;
    { int64_t g = (a.m + 255) / 256; if (g > 2048) g = 2048;
      hipLaunchKernelGGL(k_dual_scale, dim3((int)g), dim3(256), 0, h->stream, a.m, a.w, ds->dinv, act, 0); }
    HIPCHK(hipGetLastError());
    if ((rc = launch_adjoint(h, a, (const double*)nullptr, (const double*)a.w, 1.0, d.AyU, act, 0, false))) return rc;
    const double* slab; int64_t stride; int SK;
    const double* M[2] = {d.eS[cur], d.AyU};
    const double* X[2] = {d.md, d.Y[cur]};
    const double sc[2] = {2.0, -4.0 * sigma};
    if ((rc = msdp_dense_gemm(h, 2, M, X, sc, act, &slab, &stride, &SK))) return rc;
    if ((rc = dual_pp_gram(h, ds, d.md, d.Y[cur], ds->M1, act, 0))) return rc;
    double* extra = const_cast<double*>(slab) + (int64_t)SK * stride;
    { int64_t g = ((int64_t)d.n_loc * d.ld + 255) / 256; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(k_pp_apply, dim3((int)g), dim3(256), 0, h->stream, d.n_loc, d.ld, (const double*)d.Y[cur], (const double*)d.md,
                         (const double*)ds->M1, (const double*)ds->G2[cur], 2.0 * sigma, extra, act, 0); }
    HIPCHK(hipGetLastError());
    ++SK;
    return msdp_dense_hess_epilogue_obl(h, slab, stride, SK);
}

// co(Y) of :155-162 at the trial point Yt
static int dual_linesearch_cost(msdp_handle h, AffineState* st, const double* Yt, double* val) {
    DualState* ds = st->dual;
    Dev& d = h->d;
    int rc;
    if ((rc = dual_check(h, ds))) return rc;
    const int other = h->h_ctl->cur ^ 1;
    if ((rc = dual_cost_state(h, st, Yt, d.eS[other], (const int*)nullptr, 0))) return rc;
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, ds->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *val = v;
    return 0;
}

// Second half of msdp_create_dual_unitdiag: msdp_affine_setup has uploaded At (= A'), b and C = reshape(c).
int msdp_dual_setup(msdp_handle h, const int64_t* at_jc, const int64_t* at_ir, const double* at_pr, const double* b, const double* c,
                    const double* dAAt, int32_t nf, const int64_t* b_jc, const int64_t* b_ir, const double* b_pr, const double* cf) {
    AffineState* st = astate(h);
    if (!st) { msdp_set_error("affine state missing"); return MSDP_ESTATE; }
    Dev& d = h->d;
    const int n = d.n, nS = st->a.nS;
    const int64_t m = st->a.m;
    DualState* ds = new DualState();
    st->dual = ds;
    ds->nf = nf;
    std::vector<double> dinv((size_t)m), Ac((size_t)m, 0.0), bA((size_t)n * nS, 0.0);
    for (int64_t k = 0; k < m; ++k) {
        if (!(dAAt[k] > 0.0)) { msdp_set_error("dual kind: dAAt(%lld) = %g is not positive", (long long)k, dAAt[k]); return MSDP_EINVAL; }
        dinv[(size_t)k] = 1.0 / dAAt[k];
        double acc = 0.0;
        const double bk = b[k] * dinv[(size_t)k];
        for (int64_t t = at_jc[k]; t < at_jc[k + 1]; ++t) {
            const int64_t r = at_ir[t];                    // column-major vec index i + j*n -> row-major (i, j)
            acc += at_pr[t] * c[r];
            bA[(size_t)((r % n) * nS + r / n)] += at_pr[t] * bk;       // bA = iA*b (:39)
        }
        Ac[(size_t)k] = acc;
    }
    int rc;
    if ((rc = up(h, dinv, &ds->dinv)) || (rc = up(h, Ac, &ds->Ac))) return rc;
    std::vector<int> bjc((size_t)nf + 1, 0), bir;
    std::vector<double> bpr, cfv((size_t)std::max(nf, 1), 0.0);
    for (int j = 0; j < nf; ++j) {
        for (int64_t t = b_jc[j]; t < b_jc[j + 1]; ++t) {
            if (b_ir[t] < 0 || b_ir[t] >= m) { msdp_set_error("dual kind: row index of B out of range"); return MSDP_EINVAL; }
            bir.push_back((int)b_ir[t]); bpr.push_back(b_pr[t]);
        }
        bjc[(size_t)j + 1] = (int)bir.size();
        cfv[(size_t)j] = cf[j];
    }
    if (bir.empty()) { bir.push_back(0); bpr.push_back(0.0); }
    if ((rc = up(h, bjc, &ds->bjc)) || (rc = up(h, bir, &ds->bir)) || (rc = up(h, bpr, &ds->bpr)) || (rc = up(h, cfv, &ds->cf))) return rc;
    const size_t msz = (size_t)n * nS * sizeof(double);
    void* p = nullptr;
    double** mats[4] = {&ds->x, &ds->bA, &ds->T, &ds->Sg};
    for (int q = 0; q < 4; ++q) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, msz))) return rc;
        *mats[q] = (double*)p;
        HIPCHK(hipMemset(p, 0, msz));
    }
    HIPCHK(msdp_memcpy(ds->bA, bA.data(), msz, hipMemcpyHostToDevice));
    const size_t ppsz = (size_t)DUAL_PP_MAXLD * DUAL_PP_MAXLD * sizeof(double);
    double** pps[3] = {&ds->G2[0], &ds->G2[1], &ds->M1};
    for (int q = 0; q < 3; ++q) {
        if ((rc = msdp_dev_alloc_bytes(h, &p, ppsz))) return rc;
        *pps[q] = (double*)p;
        HIPCHK(hipMemset(p, 0, ppsz));
    }
    if ((rc = msdp_dev_alloc_bytes(h, &p, ppsz * DUAL_PP_BLOCKS))) return rc;
    ds->pp_part = (double*)p;
    const size_t nfb = (size_t)std::max(nf, 1) * sizeof(double);
    if ((rc = msdp_dev_alloc_bytes(h, &p, nfb))) return rc;
    ds->wf = (double*)p; HIPCHK(hipMemset(p, 0, nfb));
    if ((rc = msdp_dev_alloc_bytes(h, &p, nfb))) return rc;
    ds->Af = (double*)p; HIPCHK(hipMemset(p, 0, nfb));
    if ((rc = msdp_dev_alloc_bytes(h, &p, 8 * sizeof(double)))) return rc;
    ds->scal = (double*)p; HIPCHK(hipMemset(p, 0, 8 * sizeof(double)));
    // the adjoint of the dual kind always sweeps the whole matrix (X and As are dense)
    return 0;
}

// sigma and the free multipliers w for the next trustregions() call; T = bA + x - sigma*C
int msdp_dual_set_penalty_impl(msdp_handle h, double sigma, const double* wf_host) {
    AffineState* st = astate(h);
    if (!st || !st->dual) { msdp_set_error("dual_set_penalty: not a dual handle"); return MSDP_ESTATE; }
    if (!(sigma > 0)) { msdp_set_error("sigma must be positive"); return MSDP_EINVAL; }
    DualState* ds = st->dual;
    if (ds->nf > 0) {
        if (!wf_host) { msdp_set_error("dual_set_penalty: w is null"); return MSDP_EINVAL; }
        HIPCHK(msdp_memcpy_async(ds->wf, wf_host, (size_t)ds->nf * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    st->sigma = sigma;
    h->h_ctl->sigma = sigma;
    const int64_t tot = (int64_t)st->a.n * st->a.nS;
    hipLaunchKernelGGL(k_dual_T, dim3(2048), dim3(256), 0, h->stream, tot, ds->T, (const double*)ds->bA, (const double*)ds->x,
                       (const double*)h->d.Cd, sigma);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    ds->T_valid = true;
    return 0;
}

// :70-81 at the resident point: scal = {b'y, <C, eX>, |As|^2}, Af = B'y - cf (nf), z (n); x is updated on the device,
// X = eX - diag(z) is left in d.Sdual for msdp_escape_eigs_dual / msdp_get_dual_slack, y in a.w for msdp_dual_get_y.
int msdp_dual_outer_step_impl(msdp_handle h, double* scal_host, double* Af_host, double* z_host) {
    AffineState* st = astate(h);
    if (!st || !st->dual) { msdp_set_error("dual_outer_step: not a dual handle"); return MSDP_ESTATE; }
    DualState* ds = st->dual;
    Dev& d = h->d;
    AffineDev a = st->a;
    a.p = d.p; a.ld = d.ld;
    const double sigma = st->sigma;
    const double* Ys = d.Y[h->h_ctl->cur];
    int rc;
    if ((rc = launch_A(h, a, st->nnz, Ys, Ys, (const int*)nullptr, 0, 0, (double*)nullptr, sigma))) return rc;
    hipLaunchKernelGGL(k_dual_y, dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, a.m, a.w, ds->dinv, ds->Ac, a.b, d.P, (const int*)nullptr, 0);
    HIPCHK(hipGetLastError());
    if ((rc = msdp_k_sum_to_fwd(h, P_S1, ds->scal + 1))) return rc;
    if (ds->nf > 0) {
        hipLaunchKernelGGL(k_dual_free, dim3(ds->nf), dim3(256), 0, h->stream, ds->bjc, ds->bir, ds->bpr, (const double*)a.w, ds->cf,
                           (const double*)nullptr, sigma, ds->Af, (const int*)nullptr, 0);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_gram_mfma, dim3((a.nS + 63) / 64, (a.n + 63) / 64), dim3(512), 0, h->stream, a.n, a.nS, a.ld, Ys, Ys, ds->Sg,
                       (const int*)nullptr, 0, 0);
    HIPCHK(hipGetLastError());
    // (d.Sdual may alias the Gram scratch a.W: launch_A has consumed it by now)
    if ((rc = launch_adjoint(h, a, (const double*)d.Cd, (const double*)a.w, 1.0, d.Sdual, (const int*)nullptr, 0, false))) return rc;
    hipLaunchKernelGGL(k_dual_outer, dim3(d.G), dim3(MSDP_BLOCK), 0, h->stream, d, a.nS, d.Sdual, (const double*)ds->Sg, ds->x,
                       (const double*)ds->bA, (const double*)d.Cd, sigma, d.W0);
    HIPCHK(hipGetLastError());
    if ((rc = msdp_k_sum_to_fwd(h, P_S2, ds->scal + 2)) || (rc = msdp_k_sum_to_fwd(h, P_S3, ds->scal + 3))) return rc;
    HIPCHK(msdp_memcpy_async(scal_host, ds->scal + 1, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (ds->nf > 0) HIPCHK(msdp_memcpy_async(Af_host, ds->Af, (size_t)ds->nf * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(msdp_memcpy_async(z_host, d.W0, (size_t)a.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    ds->T_valid = false;
    return 0;
}

int msdp_dual_get_y_impl(msdp_handle h, double* y_host) {
    AffineState* st = astate(h);
    if (!st || !st->dual) { msdp_set_error("dual_get_y: not a dual handle"); return MSDP_ESTATE; }
    HIPCHK(msdp_memcpy_async(y_host, st->a.w, (size_t)st->a.m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
