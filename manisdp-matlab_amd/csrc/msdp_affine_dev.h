// msdp_affine_dev.h -- device view of the affine operator (At in its several layouts) shared by msdp_affine.hip and the dense
// contraction launch that carries the SDDMM of the sphere Hess-vec as a side job (msdp_dense.hip, k_dense_partial3<.., SIDE>).
#pragma once
#include "msdp_device.h"

struct AffineDev {
    int n, nS, p, ld;
    int64_t m;
    const int* cjc;        // m+1 column pointers (CSC by constraint)
    // long columns are cut into work items of <= SDDMM_CHUNK nonzeros so that one constraint (e.g. the trace
    // row of a theta problem: n nonzeros) cannot serialise a whole launch on a single lane group
    int64_t nitems;
    const int* it0;        // first nonzero of item
    const int* it1;        // one past the last nonzero of item
    const int* kit;        // m+1: items of constraint k are kit[k] .. kit[k+1]-1
    const int* longk;      // constraints with more than FIN_SHORT items
    int nlong;
    double* ival;          // partial value per item
    const int* ci;         // row i of each nonzero
    const int* cj;         // col j of each nonzero
    const double* cv;
    const int* rp;         // n*n+1 row pointers (CSR by matrix entry r = i*n + j)
    const int* cidx;       // i*nS + j of each nonzero: position in the dense Gram matrix W = Ya*Yb' (Gram route)
    // symmetric data: the same constraints over their entries i <= j only (coefficient halved on the diagonal), for the
    // Gram route on Wsym = Ya*Yb' + Yb*Ya' -- half the gathers (upper_view() swaps these in)
    int usym;
    int64_t unitems;
    const int* uit0; const int* uit1; const int* ukit; const int* ulongk; int unlong;
    const int* ucjc;       // m+1 column pointers of the upper view
    const int* ucidx; const double* ucv;
    double* W;             // n x nS scratch for the Gram route (aliases the AyU buffer)
    const int* rk;         // constraint index
    const double* rv;
    const double* b;
    const double* y;
    double* w;             // A(.) result, length m
    double* Axb[2];        // per slot
    // tiled upper-triangle copy of the CSR-by-entry arrays (symmetric data only; k_adjoint_tiled): the entries of
    // every 32 x 32 tile (bi <= bj) are stored together, row-major inside the tile
    const int* trp;        // ntp*1024 + 1 offsets into trk / trv
    const int* trk;
    const double* trv;
    const short* tp_i;     // tile pair -> (bi, bj)
    const short* tp_j;
    int ntp;               // number of tile pairs (0: data not symmetric, flat kernel)
    // entries with more than ADJ_LONG nonzeros (the (x_i, x_j) block of a BQP moment matrix: 59 each) are left out
    // by the tile workgroups and summed by one wave each in extra workgroups of the same launch
    const int* lpos;       // i*nS + j of long entry q (i <= j)
    const int* lmir;       // j*nS + i
    const int* ls0;        // its range in trk / trv
    const int* ls1;
    int nlong_e;
    // fused SDDMM (k_sddmm1): short constraints are summed whole by one lane group; the items of the long ones go through
    // ival and are summed by the workgroup that arrives last
    int nshort; const int* sk;         // short constraints (<= FIN_SHORT items)
    int nlit; const int* lit0; const int* lit1;   // items of the long constraints
    const int* lkit;                   // nlong + 1: items of long constraint q (= longk[q]) are lkit[q] .. lkit[q+1]-1
    unsigned* cnt;                     // arrival counter
    // flattened records (one round trip instead of a chain of pointer loads): unit u of k_sddmm1 -> nonzero range and constraint
    // (uk >= 0: short constraint; < 0: item -1 - uk of a long one); touched entry q of k_sph_hess_fused -> column, first
    // (coefficient, constraint) pair, number of further pairs (they follow at rp[sup[q]] + 1)
    const int* us0; const int* us1; const int* uk;
    const int* sqj; const int* sqk; const double* sqv; const int* sqmore;   // sqk < 0: long constraint number -1 - sqk (its value comes from the epilogue's own sum)
    const int* rkx;                    // rk with the long constraints encoded the same way (the `more` loop of k_sph_hess_fused)
    // B route of the Hess-vec (symmetric data, every constraint short): A'(A(M)) on the upper entries as ONE sparse matrix applied
    // to the Gram matrix, B[e][e'] = sum_k a_k[e] * c_k[e'] (k_adjoint_gram): no m-vector, no second pass over At
    int bW;                // ELL width of B (0: route not built)
    const int* bidx;       // [ntp][bW][1024]: position i'*nS + j' (i' <= j') in Wsym; padding = position 0 with coefficient 0
    const double* bval;    // [ntp][bW][1024]
    const unsigned* bpk;   // packed form of (bidx, bval) when it applies: position | code << 24, the coefficient = bdict[code] (SeDuMi
                           //   moment data has a dozen distinct coefficients: 5 bytes less per nonzero of B); else null
    const double* bdict;   // 256 coefficients
    const unsigned char* blong;   // [ntp][1024]: row longer than bW (summed by one wave each, like the long entries of the tiled adjoint)
    const int* blpos; const int* blmir; const int* bls0; const int* bls1; int bnlong;
    const int* blk; const double* blv;       // (position, coefficient) pairs of the long rows
    double* Wg;            // Gram matrix of the B route (n x nS): AyU is written while it is read
    const int* sup;        // entries r = i*n + j that occur in some constraint (nsup > 0: At touches few entries)
    const int* suprow;     // n+1: the entries of matrix row i are sup[suprow[i] .. suprow[i+1])
    int nsup;
};

#define SDDMM_CHUNK 16
#define SPB 4                   // panel rows requested together by the sparse A'(w)*Y products
#define FIN_SHORT 8           // constraints with more items than this are summed by a whole wave (k_sddmm_finish)

// Side job of the contraction launch (sphere / Euclidean Hess-vec on the SDDMM route, one rank; round 4): the work of k_sddmm1 in
// mode 2 -- w_k = <A_k, Ya Yb'> for the short constraints, the item values of the long ones (ival; the epilogue sums them), the
// partial sums of w_k (A x)_k, <U, G>, <U, Y> -- done by the first `side_rows` rows of workgroups of the contraction's grid, so
// that its chain of dependent round trips (12-20 us as a launch of its own) hides under the 40-us matrix stream instead of
// standing in front of it.  256 threads per workgroup; job = the workgroup's index among the njobs side workgroups.
struct SideJob {
    AffineDev a;
    const double* Ya; const double* Yb;      // Y (all rows), U
    const double* Gr;                        // gradient at the current point
    const double* axc;                       // Axb of the current point
    double* P;                               // partial-sum arrays (P_T1..P_T3 get njobs entries; P_T3 slot njobs = 0: the epilogue adds the long constraints)
    double sigma;
    int n_loc, ld, njobs;
};
template <int LPR, int NCH>
__device__ __forceinline__ void msdp_sddmm_side(const SideJob& sj, int job, double* sh /* >= 16 doubles */) {
    const AffineDev& a = sj.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    constexpr int CPW = 64 / LPR;
    const int sub = lane & (LPR - 1), csub = lane / LPR;
    const int64_t nunits = (int64_t)a.nshort + a.nlit;
    double p1 = 0.0, p2 = 0.0, pacc = 0.0;
    {   // <U, G>, <U, Y> over the rows of this job: requested first, they depend on nothing
        const unsigned q = (unsigned)sj.n_loc / (unsigned)sj.njobs, r = (unsigned)sj.n_loc - q * (unsigned)sj.njobs, c = (unsigned)job;
        const int lo = (int)(c * q + (c < r ? c : r)), hi = lo + (int)q + (c < r ? 1 : 0);
        const int64_t e0 = (int64_t)lo * sj.ld, e1 = (int64_t)hi * sj.ld;
        for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * blockDim.x) {
            const double2 u = ld2(sj.Yb + i), g = ld2(sj.Gr + i), y = ld2(sj.Ya + i);
            p1 += u.x * g.x + u.y * g.y;
            p2 += u.x * y.x + u.y * y.y;
        }
    }
    const int64_t ustride = (int64_t)sj.njobs * nwave * CPW;
    for (int64_t u = ((int64_t)job * nwave + wave) * CPW + csub; u < nunits; u += ustride) {
        const int s0 = a.us0[u], s1 = a.us1[u], kk = a.uk[u];
        double eb = 0.0, ey = 0.0, ea = 0.0;
        if (kk >= 0) { eb = a.b[kk]; ey = a.y[kk] / sj.sigma; ea = sj.axc[kk]; }
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
                const double* ya = sj.Ya + (int64_t)ii[u2] * sj.ld + 2 * sub;
                const double* yb = sj.Yb + (int64_t)jj[u2] * sj.ld + 2 * sub;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (2 * sub + ch * 2 * LPR < sj.ld) {
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
            if (kk >= 0) { a.w[kk] = acc; pacc += acc * (ea + eb + ey); }
            else a.ival[-1 - kk] = acc;                      // consumed by the NEXT launch (the epilogue): a plain store will do
        }
    }
    p1 = msdp_wave_sum(p1); p2 = msdp_wave_sum(p2); pacc = msdp_wave_sum(pacc);
    if (lane == 0) { sh[wave] = p1; sh[4 + wave] = p2; sh[8 + wave] = pacc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int i = 0; i < nwave; ++i) s += sh[threadIdx.x * 4 + i];
        sj.P[(P_T1 + threadIdx.x) * MSDP_MAX_GRID + job] = s;
    }
    if (job == 0 && threadIdx.x == 3) sj.P[P_T3 * MSDP_MAX_GRID + sj.njobs] = 0.0;
}
