// msdp_blockjacobi.hip -- eig(S_i) of every diagonal block of the dual slack, on the device, in one launch.
//
// ManiSDP_multiblock.m:78-88 takes [vS{i}, dS{i}] = eig(S{i}) of every block in every outer iteration (all eigenvalues: dinf needs the
// smallest and the largest, the escape the number of negative ones; eigenvectors: the first `delta` columns, :129-147).  With
// many small blocks that host loop is the solve: example_bqp_sparse.m's chain of 100 cliques spends 48 of its 69 s in 17 600
// LAPACK calls of order 211 (profiles/r4_multiblock_times.log).  Here one workgroup per block runs a cyclic two-sided Jacobi
// iteration on a copy of the block (A <- J'AJ, V <- VJ, rotations of one round in parallel: round-robin pairing, m/2 disjoint
// pairs per round, m - 1 rounds per sweep), the matrix in global memory (n_i <= 256: it lives in the L2 of the workgroup's XCD),
// until off(A)^2 <= 1e-30 |A|_F^2; eigenvalues sorted ascending, the first k eigenvectors returned.  Jacobi's eigenvalues are
// accurate to |A| eps, its eigenvectors orthogonal to rounding; vectors of a multiple eigenvalue are one orthonormal basis of the
// eigenspace, as with LAPACK, and their signs are whatever the rotations leave.
#include "msdp_device.h"
#include <vector>

int msdp_affine_block_source(msdp_handle h, int64_t row0, int64_t n, int64_t* off, int64_t* ld);   // msdp_affine.hip (per-block storage)
int msdp_dense_nS(int n);

#define JAC_MAXN 256
#define JAC_THREADS 1024

struct JacArgs {
    int nb, k;
    const int64_t* soff; const int64_t* sld;      // block b: S_b(i, j) = S[soff[b] + i * sld[b] + j]
    const int* n; const int64_t* woff;            // order; offset of the block's n x n workspaces
    const int64_t* r0;                            // first row of the block in the output arrays
    const double* S;
    double* A; double* V;                         // workspaces (sum n_b^2 doubles each)
    double* w; double* vec;                       // outputs: w[r0 + rank], vec[(r0 + i) * k + rank] for rank < k
    int* sweeps;                                  // per block: sweeps used (diagnostic; -1: not converged in JAC_MAXSWEEP)
};
#define JAC_MAXSWEEP 40

__global__ __launch_bounds__(JAC_THREADS) void k_block_jacobi(JacArgs a) {
    __shared__ double cc[JAC_MAXN / 2], ss[JAC_MAXN / 2];
    __shared__ int pp[JAC_MAXN / 2], qq[JAC_MAXN / 2];
    __shared__ double red[2 * (JAC_THREADS / 64)];
    __shared__ double dg[JAC_MAXN];
    __shared__ int rk[JAC_MAXN];
    __shared__ int done;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n[b];
    const int m = n + (n & 1), half = m >> 1;
    const int64_t so = a.soff[b], sl = a.sld[b];
    double* __restrict__ A = a.A + a.woff[b];
    double* __restrict__ V = a.V + a.woff[b];
    for (int e = tid; e < n * n; e += JAC_THREADS) {
        const int i = e / n, j = e - i * n;
        A[e] = 0.5 * (a.S[so + (int64_t)i * sl + j] + a.S[so + (int64_t)j * sl + i]);      // ManiSDP_multiblock.m:86 symmetrises too
        V[e] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < JAC_MAXSWEEP; ++sweep) {
        // off(A)^2 and |A|_F^2
        double off = 0.0, tot = 0.0;
        for (int e = tid; e < n * n; e += JAC_THREADS) {
            const int i = e / n, j = e - i * n;
            const double v = A[e] * A[e];
            tot += v;
            if (i != j) off += v;
        }
        off = msdp_wave_sum(off); tot = msdp_wave_sum(tot);
        if (lane == 0) { red[wave] = off; red[JAC_THREADS / 64 + wave] = tot; }
        __syncthreads();
        if (tid == 0) {
            double o2 = 0.0, t2 = 0.0;
            for (int q = 0; q < JAC_THREADS / 64; ++q) { o2 += red[q]; t2 += red[JAC_THREADS / 64 + q]; }
            done = (o2 <= 1e-30 * t2) ? 1 : 0;
        }
        __syncthreads();
        if (done) break;
        for (int r = 0; r < m - 1; ++r) {
            if (tid < half) {
                // round-robin pairing of m players: player m - 1 stays, the others rotate
                int p, q;
                if (tid == 0) { p = m - 1; q = r; }
                else { p = (r + tid) % (m - 1); q = (r - tid + (m - 1)) % (m - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < n) {                                              // (q == n: the dummy player of an odd order)
                    const double apq = A[p * n + q];
                    if (apq != 0.0) {
                        const double app = A[p * n + p], aqq = A[q * n + q];
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                } else q = p;                                              // identity on (p, p)
                pp[tid] = p; qq[tid] = q; cc[tid] = c; ss[tid] = s;
            }
            __syncthreads();
            // columns p, q of A and V (threads of a wave share a row: its entries are one contiguous stretch).  Four items per trip,
            // all their loads before the first store: the compiler cannot tell the stores of one item from the loads of the next
            // (same array) and would otherwise run them one after the other, a memory round trip each
            for (int e0 = tid; e0 < n * half; e0 += 4 * JAC_THREADS) {
                double ap[4], aq[4], vp[4], vq[4], c[4], s[4];
                int ip[4], iq[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * JAC_THREADS;
                    on[u] = e < n * half;
                    const int ec = on[u] ? e : tid;
                    const int i = ec / half, k2 = ec - i * half;
                    const int p = pp[k2], q = qq[k2];
                    on[u] = on[u] && p != q;
                    c[u] = cc[k2]; s[u] = ss[k2];
                    ip[u] = i * n + p; iq[u] = i * n + q;
                    ap[u] = A[ip[u]]; aq[u] = A[iq[u]]; vp[u] = V[ip[u]]; vq[u] = V[iq[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (on[u]) {
                        A[ip[u]] = c[u] * ap[u] - s[u] * aq[u]; A[iq[u]] = s[u] * ap[u] + c[u] * aq[u];
                        V[ip[u]] = c[u] * vp[u] - s[u] * vq[u]; V[iq[u]] = s[u] * vp[u] + c[u] * vq[u];
                    }
                }
            }
            __syncthreads();
            // rows p, q of A
            for (int e0 = tid; e0 < half * n; e0 += 4 * JAC_THREADS) {
                double ap[4], aq[4], c[4], s[4];
                int ip[4], iq[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * JAC_THREADS;
                    on[u] = e < half * n;
                    const int ec = on[u] ? e : tid;
                    const int k2 = ec / n, j = ec - k2 * n;
                    const int p = pp[k2], q = qq[k2];
                    on[u] = on[u] && p != q;
                    c[u] = cc[k2]; s[u] = ss[k2];
                    ip[u] = p * n + j; iq[u] = q * n + j;
                    ap[u] = A[ip[u]]; aq[u] = A[iq[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (on[u]) { A[ip[u]] = c[u] * ap[u] - s[u] * aq[u]; A[iq[u]] = s[u] * ap[u] + c[u] * aq[u]; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) a.sweeps[b] = sweep < JAC_MAXSWEEP ? sweep : -1;
    // eigenvalues = diagonal; rank by counting (ties by index); the first k eigenvectors
    if (tid < n) dg[tid] = A[tid * n + tid];
    __syncthreads();
    if (tid < n) {
        const double v = dg[tid];
        int rnk = 0;
        for (int j = 0; j < n; ++j) rnk += (dg[j] < v || (dg[j] == v && j < tid)) ? 1 : 0;
        rk[tid] = rnk;
        a.w[a.r0[b] + rnk] = v;
    }
    __syncthreads();
    const int k = a.k < n ? a.k : n;
    for (int e = tid; e < n * n; e += JAC_THREADS) {
        const int i = e / n, j = e - i * n;
        const int rnk = rk[j];
        if (rnk < k) a.vec[(a.r0[b] + i) * a.k + rnk] = V[e];
    }
    if (k < a.k) for (int e = tid; e < n * (a.k - k); e += JAC_THREADS) a.vec[(a.r0[b] + e / (a.k - k)) * a.k + k + e % (a.k - k)] = 0.0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same eigenproblem the way LAPACK's dsyevx goes about it, one workgroup per block: Householder tridiagonalisation (the
// reflectors stay in the strict lower triangle of the work matrix), ALL eigenvalues by bisection on the Sturm count (one thread
// per eigenvalue), the eigenvectors of the k smallest by inverse iteration on the tridiagonal matrix (LU with partial pivoting,
// modified Gram-Schmidt inside clusters: near an optimum the blocks of S are rank deficient and the k smallest eigenvalues ARE a
// cluster at zero), back-transformation by the reflectors (one wave per vector).  75 MB of traffic per block of order 211 where
// the Jacobi iteration above moves 5 GB through the same single CU: 277 -> ~2 ms per block.
#define TRI_ITERS 5
struct TriArgs {
    JacArgs j;
    double* ws;                                   // per block: 5 * k * JAC_MAXN doubles (tridiagonal LU of the inverse iteration)
};
__device__ __forceinline__ double tri_block_sum(double v, double* red, int tid) {
    v = msdp_wave_sum(v);
    __syncthreads();                                               // red is free again
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < JAC_THREADS / 64; ++q) s += red[q];
    return s;
}
__global__ __launch_bounds__(JAC_THREADS) void k_block_tridiag(TriArgs ta) {
    const JacArgs& a = ta.j;
    __shared__ double vv[JAC_MAXN], pw[JAC_MAXN], dd[JAC_MAXN], ee[JAC_MAXN], tt[JAC_MAXN], wv[JAC_MAXN];
    __shared__ double red[JAC_THREADS / 64];
    __shared__ double Z[JAC_MAXN * 8 + 8];                          // the k <= 8 vectors, [i * KZ + c]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n[b];
    const int64_t so = a.soff[b], sl = a.sld[b];
    double* __restrict__ A = a.A + a.woff[b];
    const int KZ = 8;
    const int k = min(min(a.k, n), KZ);
    for (int e = tid; e < n * n; e += JAC_THREADS) {
        const int i = e / n, j = e - i * n;
        A[e] = 0.5 * (a.S[so + (int64_t)i * sl + j] + a.S[so + (int64_t)j * sl + i]);
    }
    if (tid < JAC_MAXN) { dd[tid] = 0.0; ee[tid] = 0.0; tt[tid] = 0.0; }
    __syncthreads();
    // ---- tridiagonalisation: A <- H_k A H_k, H_k = I - tau v v', v = (1, A[k+2:, k]) on rows / columns k+1 .. n-1
    for (int kk = 0; kk + 2 < n; ++kk) {
        const int m = n - kk - 1;
        double part = 0.0;
        for (int i = tid; i < m; i += JAC_THREADS) {
            const double x = A[(kk + 1 + i) * n + kk];
            vv[i] = x;
            if (i > 0) part += x * x;
        }
        const double tail = tri_block_sum(part, red, tid);
        const double x0 = vv[0];
        if (tail == 0.0) {                                          // nothing below the subdiagonal: H = I
            if (tid == 0) { dd[kk] = A[kk * n + kk]; ee[kk] = x0; tt[kk] = 0.0; }
            __syncthreads();
            continue;
        }
        const double alpha = (x0 >= 0.0 ? -1.0 : 1.0) * sqrt(x0 * x0 + tail);
        const double beta = x0 - alpha, tau = -beta / alpha;
        __syncthreads();
        for (int i = tid; i < m; i += JAC_THREADS) vv[i] = (i == 0) ? 1.0 : vv[i] / beta;
        __syncthreads();
        // p = tau * A22 v: one wave per row
        for (int i = wave; i < m; i += JAC_THREADS / 64) {
            const double* row = A + (kk + 1 + i) * n + kk + 1;
            double acc = 0.0;
            for (int j = lane; j < m; j += 64) acc = fma(row[j], vv[j], acc);
            acc = msdp_wave_sum(acc);
            if (lane == 0) pw[i] = tau * acc;
        }
        __syncthreads();
        double pv = 0.0;
        for (int i = tid; i < m; i += JAC_THREADS) pv += pw[i] * vv[i];
        const double pdotv = tri_block_sum(pv, red, tid);
        for (int i = tid; i < m; i += JAC_THREADS) pw[i] -= 0.5 * tau * pdotv * vv[i];            // w
        __syncthreads();
        for (int e = tid; e < m * m; e += JAC_THREADS) {
            const int i = e / m, j = e - i * m;
            A[(kk + 1 + i) * n + kk + 1 + j] -= vv[i] * pw[j] + pw[i] * vv[j];
        }
        for (int i = tid + 1; i < m; i += JAC_THREADS) A[(kk + 1 + i) * n + kk] = vv[i];          // the reflector (v_0 = 1 implied)
        if (tid == 0) { dd[kk] = A[kk * n + kk]; ee[kk] = alpha; tt[kk] = tau; }
        __syncthreads();
    }
    if (tid == 0) {
        if (n >= 2) { dd[n - 2] = A[(n - 2) * n + n - 2]; ee[n - 2] = A[(n - 1) * n + n - 2]; }
        dd[n - 1] = A[(n - 1) * n + n - 1];
    }
    __syncthreads();
    // ---- all eigenvalues: bisection on the Sturm count, thread j -> eigenvalue j (ascending)
    double glo = 0.0, ghi = 0.0, pivmin = 0.0;
    {
        double lo = 1e300, hi = -1e300, emax = 0.0;
        for (int i = 0; i < n; ++i) {
            const double r = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i + 1 < n ? fabs(ee[i]) : 0.0);
            lo = fmin(lo, dd[i] - r); hi = fmax(hi, dd[i] + r);
            if (i + 1 < n) emax = fmax(emax, ee[i] * ee[i]);
        }
        const double span = fmax(hi - lo, 1e-300);
        glo = lo - 1e-12 * span - 1e-300; ghi = hi + 1e-12 * span + 1e-300;
        pivmin = fmax(1e-292, 2.2250738585072014e-308 * fmax(1.0, emax));
    }
    if (tid < n) {
        double lo = glo, hi = ghi;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (mid == lo || mid == hi) break;
            int cnt = 0;
            double q = dd[0] - mid;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0;
            for (int i = 1; i < n; ++i) {
                q = dd[i] - mid - ee[i - 1] * ee[i - 1] / q;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0;
            }
            if (cnt > tid) hi = mid; else lo = mid;
        }
        wv[tid] = 0.5 * (lo + hi);
    }
    __syncthreads();
    if (tid < n) a.w[a.r0[b] + tid] = wv[tid];
    if (k == 0) { if (tid == 0) a.sweeps[b] = 0; return; }
    // ---- eigenvectors of the k smallest: inverse iteration on T, lane c < k solves (T - w_c I) z = y_c (LU with partial pivoting,
    // dgttrf / dgtts2), thread 0 orthogonalises inside clusters and normalises
    double scale = fmax(fabs(wv[0]), fabs(wv[n - 1]));
    if (!(scale > 0.0)) scale = 1.0;
    for (int e = tid; e < n * KZ; e += JAC_THREADS) {
        unsigned hsh = (unsigned)e * 2654435761u + 12345u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        Z[e] = (double)(hsh & 0xffffff) / 16777216.0 - 0.5;
    }
    __syncthreads();
    double* ws = ta.ws + (size_t)b * 5 * KZ * JAC_MAXN;
    for (int it = 0; it < TRI_ITERS; ++it) {
        if (tid < k) {
            const int c = tid;
            double* dl = ws + (size_t)(c * 5 + 0) * JAC_MAXN; double* d = ws + (size_t)(c * 5 + 1) * JAC_MAXN;
            double* du = ws + (size_t)(c * 5 + 2) * JAC_MAXN; double* du2 = ws + (size_t)(c * 5 + 3) * JAC_MAXN;
            double* x = ws + (size_t)(c * 5 + 4) * JAC_MAXN;
            const double lam = wv[c];
            const double tiny = 2.220446049250313e-16 * scale;
            for (int i = 0; i < n; ++i) { d[i] = dd[i] - lam; x[i] = Z[i * KZ + c]; if (i + 1 < n) { dl[i] = ee[i]; du[i] = ee[i]; } du2[i] = 0.0; }
            // factor and forward-solve in one sweep
            for (int i = 0; i + 1 < n; ++i) {
                if (fabs(d[i]) >= fabs(dl[i])) {
                    if (d[i] == 0.0) d[i] = tiny;
                    const double f = dl[i] / d[i];
                    d[i + 1] -= f * du[i];
                    x[i + 1] -= f * x[i];
                } else {                                            // interchange rows i and i + 1
                    const double f = d[i] / dl[i];
                    d[i] = dl[i];
                    const double t1 = d[i + 1];
                    d[i + 1] = du[i] - f * t1;
                    if (i + 2 < n) { du2[i] = du[i + 1]; du[i + 1] = -f * du[i + 1]; }
                    du[i] = t1;
                    const double t2 = x[i]; x[i] = x[i + 1]; x[i + 1] = t2 - f * x[i];
                }
            }
            if (fabs(d[n - 1]) < tiny) d[n - 1] = (d[n - 1] < 0.0 ? -tiny : tiny);
            // back substitution with U (diagonals d, du, du2)
            x[n - 1] /= d[n - 1];
            if (n > 1) x[n - 2] = (x[n - 2] - du[n - 2] * x[n - 1]) / d[n - 2];
            for (int i = n - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
            double nrm = 0.0;
            for (int i = 0; i < n; ++i) nrm = fmax(nrm, fabs(x[i]));
            if (!(nrm > 0.0)) nrm = 1.0;
            for (int i = 0; i < n; ++i) Z[i * KZ + c] = x[i] / nrm;
        }
        __syncthreads();
        if (wave == 0) {                                            // modified Gram-Schmidt inside clusters, in eigenvalue order
            for (int c = 0; c < k; ++c) {
                for (int c2 = 0; c2 < c; ++c2) {
                    if (fabs(wv[c] - wv[c2]) > 1e-3 * scale) continue;
                    double dot = 0.0;
                    for (int i = lane; i < n; i += 64) dot += Z[i * KZ + c] * Z[i * KZ + c2];
                    dot = msdp_wave_sum(dot);
                    for (int i = lane; i < n; i += 64) Z[i * KZ + c] -= dot * Z[i * KZ + c2];
                }
                double nn = 0.0;
                for (int i = lane; i < n; i += 64) nn += Z[i * KZ + c] * Z[i * KZ + c];
                nn = sqrt(msdp_wave_sum(nn));
                if (!(nn > 0.0)) nn = 1.0;
                for (int i = lane; i < n; i += 64) Z[i * KZ + c] /= nn;
            }
        }
        __syncthreads();
    }
    // ---- back-transformation z = H_0 H_1 ... H_{n-3} y: wave c applies the reflectors to its own vector, last one first
    if (wave < k) {
        const int c = wave;
        for (int kk = n - 3; kk >= 0; --kk) {
            const double tau = tt[kk];
            if (tau == 0.0) continue;
            const int m = n - kk - 1;
            double s = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double v = (i == 0) ? 1.0 : A[(kk + 1 + i) * n + kk];
                s += v * Z[(kk + 1 + i) * KZ + c];
            }
            s = msdp_wave_sum(s) * tau;
            for (int i = lane; i < m; i += 64) {
                const double v = (i == 0) ? 1.0 : A[(kk + 1 + i) * n + kk];
                Z[(kk + 1 + i) * KZ + c] -= s * v;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < n * a.k; e += JAC_THREADS) {
        const int i = e / a.k, c = e - i * a.k;
        a.vec[(a.r0[b] + i) * a.k + c] = (c < k) ? Z[i * KZ + c] : 0.0;
    }
    if (tid == 0) a.sweeps[b] = 0;
}

// Workspace of msdp_block_eigs, kept on the handle (ten hipMalloc / hipFree pairs per outer iteration, each hipFree a
// device synchronisation; an upload that failed half-way leaked the earlier allocations): one block, grown when a call needs more,
// released with the handle.
void msdp_block_eigs_release(msdp_handle h) {
    if (h->blk_ws) (void)hipFree(h->blk_ws);
    h->blk_ws = nullptr; h->blk_ws_cap = 0;
}
extern "C" int msdp_block_eigs(msdp_handle h, int32_t nb, const int64_t* row0, const int64_t* nblk, int32_t k, int32_t method, double* w, double* V) {
    if (!h) { msdp_set_error("null handle"); return MSDP_EINVAL; }
    if (nb < 1 || !row0 || !nblk || !w || (k > 0 && !V) || k < 0 || k > 64 || method < 0 || method > 2) { msdp_set_error("block_eigs: bad argument"); return MSDP_EINVAL; }
    // method 0: tridiagonalisation + bisection + inverse iteration (k <= 8), 1: Jacobi, 2: tridiagonalisation (error when k > 8)
    const bool tri = method == 2 || (method == 0 && k <= 8);
    if (method == 2 && k > 8) { msdp_set_error("block_eigs: the tridiagonal method returns at most 8 eigenvectors"); return MSDP_EUNSUPPORTED; }
    if (h->d.costkind != COST_AFFINE || !h->dual_valid) { msdp_set_error("block_eigs: call msdp_al_dual first"); return MSDP_ESTATE; }
    const int N = h->d.n, nS = msdp_dense_nS(N);
    std::vector<int64_t> soff(nb), sld(nb), woff(nb), r0(nb);
    std::vector<int> nn(nb);
    int64_t tot = 0, rows = 0;
    for (int b = 0; b < nb; ++b) {
        if (nblk[b] < 1 || nblk[b] > JAC_MAXN) { msdp_set_error("block_eigs: block orders up to %d (block %d has %lld)", JAC_MAXN, b, (long long)nblk[b]); return MSDP_EUNSUPPORTED; }
        if (row0[b] < 0 || row0[b] + nblk[b] > N) { msdp_set_error("block_eigs: block %d outside the matrix", b); return MSDP_EINVAL; }
        if (h->blocked) {
            int rc = msdp_affine_block_source(h, row0[b], nblk[b], &soff[b], &sld[b]);
            if (rc) return rc;
        } else { soff[b] = row0[b] * nS + row0[b]; sld[b] = nS; }
        nn[b] = (int)nblk[b]; woff[b] = tot; r0[b] = rows;
        tot += nblk[b] * nblk[b]; rows += nblk[b];
    }
    const int kk = k > 0 ? k : 1;
    JacArgs a;
    a.nb = nb; a.k = kk; a.S = h->d.Sdual;
    // one workspace, carved into 256-byte aligned pieces
    size_t need = 0;
    auto piece = [&](size_t bytes) { const size_t o = need; need += (bytes + 255) / 256 * 256; return o; };
    const size_t o_soff = piece(nb * sizeof(int64_t)), o_sld = piece(nb * sizeof(int64_t)), o_woff = piece(nb * sizeof(int64_t)), o_r0 = piece(nb * sizeof(int64_t));
    const size_t o_n = piece(nb * sizeof(int)), o_sw = piece(nb * sizeof(int));
    const size_t o_A = piece((size_t)tot * sizeof(double)), o_V = tri ? 0 : piece((size_t)tot * sizeof(double));
    const size_t o_ws = tri ? piece((size_t)nb * 5 * 8 * JAC_MAXN * sizeof(double)) : 0;
    const size_t o_w = piece((size_t)rows * sizeof(double)), o_vec = piece((size_t)rows * kk * sizeof(double));
    if (h->blk_ws_cap < need) {
        msdp_block_eigs_release(h);
        if (hipMalloc(&h->blk_ws, need) != hipSuccess) { (void)hipGetLastError(); h->blk_ws = nullptr; msdp_set_error("block_eigs: device allocation of %zu bytes failed", need); return MSDP_ENOMEM; }
        h->blk_ws_cap = need;
    }
    char* base = (char*)h->blk_ws;
    int64_t *d_soff = (int64_t*)(base + o_soff), *d_sld = (int64_t*)(base + o_sld), *d_woff = (int64_t*)(base + o_woff), *d_r0 = (int64_t*)(base + o_r0);
    int* d_n = (int*)(base + o_n); int* d_sw = (int*)(base + o_sw);
    double *d_A = (double*)(base + o_A), *d_V = tri ? nullptr : (double*)(base + o_V), *d_w = (double*)(base + o_w), *d_vec = (double*)(base + o_vec);
    double* d_ws = tri ? (double*)(base + o_ws) : nullptr;
    HIPCHK(msdp_memcpy_async(d_soff, soff.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(msdp_memcpy_async(d_sld, sld.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(msdp_memcpy_async(d_woff, woff.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(msdp_memcpy_async(d_r0, r0.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(msdp_memcpy_async(d_n, nn.data(), nb * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));                 // (the index vectors are locals of this call)
    a.soff = d_soff; a.sld = d_sld; a.n = d_n; a.woff = d_woff; a.r0 = d_r0; a.A = d_A; a.V = d_V; a.w = d_w; a.vec = d_vec; a.sweeps = d_sw;
    if (tri) {
        TriArgs ta; ta.j = a; ta.ws = d_ws;
        hipLaunchKernelGGL(k_block_tridiag, dim3(nb), dim3(JAC_THREADS), 0, h->stream, ta);
    } else hipLaunchKernelGGL(k_block_jacobi, dim3(nb), dim3(JAC_THREADS), 0, h->stream, a);
    hipError_t e = hipGetLastError();
    std::vector<int> sw(nb);
    if (e == hipSuccess) e = msdp_memcpy_async(w, d_w, rows * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && k > 0) e = msdp_memcpy_async(V, d_vec, rows * k * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = msdp_memcpy_async(sw.data(), d_sw, nb * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { msdp_set_error("block_eigs: %s", hipGetErrorString(e)); return MSDP_EHIP; }
    for (int b = 0; b < nb; ++b)
        if (sw[b] < 0) { msdp_set_error("block_eigs: Jacobi iteration of block %d did not converge in %d sweeps", b, JAC_MAXSWEEP); return MSDP_ESTATE; }
    return 0;
}
