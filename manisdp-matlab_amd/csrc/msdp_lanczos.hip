// msdp_lanczos.hip -- persistent kernel for the deflated three-term Lanczos recurrence of the
// saddle-escape step (msdp_escape.hip; replaces eig(full(S)) of ManiSDP_onlyunitdiag.m:50).
//
// A Lanczos step on S = C - diag(z) is tiny at n = 20000 (one length-n SpMV, a handful of
// length-n dot products / axpys) and certifying lambda_min >= -1e-8 needs ~10^4..10^5 of them
// per call: as seven launches per step (33-58 us per step measured) the eigen-escape was 85% of
// the G81 wall-clock.  Here ONE launch runs the steps [m0, m1): every workgroup keeps its rows of
// v_{j-1}, v_j, w in registers and its rows of the deflation basis Q in LDS; per step
//   1. w = S*v_j                      (neighbour entries gathered from the exchange buffer, sc1)
//   2. grid reduction of [Q'w ; Q'v_j ; v_j'w]  (one sync: alpha_j and the deflation coefficients of the UPDATED
//      vector, Q'(w - alpha v_j - beta v_{j-1}) = Q'w - alpha Q'v_j - beta Q'v_{j-1}, by linearity)
//   3. w -= alpha_j v_j + beta_j v_{j-1} + Q Q'(...);  exchange buffer <- w;  grid reduction of |w|^2
//   4. beta_{j+1} = |w|, v_{j+1} = w / beta_{j+1}  -> column j+1 of the stored Lanczos basis
// i.e. two grid synchronisations per step.  The synchronisation is the slot scheme of
// msdp_persist.hip (agent-coherent sc1 stores, sentinel polling, three rotating generations,
// bounded spin); slots are workgroup-major so that lane c sums value c over the workgroups in a
// fixed order -- every workgroup obtains bit-identical scalars.
#include "msdp_device.h"
#include <math.h>
#include <cstdlib>

#define LZ_PB 512
#define LZ_PWAVES (LZ_PB / 64)
#define LZ_NV 128                         // values per workgroup slot: Q'w, Q'v (<= 63 columns each) + alpha
#define LZ_GMAX 256
#define LZ_GEN 3
#define LZ_SENT 0xFFF8DEADBEEF0001ULL
#define LZ_SPIN_LIMIT (1 << 22)
#define LZ_CPOL_SC1 16
#define LZ_RMAX 2                         // rows per thread
#define LZ_KREG 6                         // (col, val) pairs of a row kept in registers

typedef unsigned int lz_v2u __attribute__((ext_vector_type(2)));

size_t msdp_lanczos_slot_bytes() { return (size_t)LZ_GEN * LZ_GMAX * LZ_NV * sizeof(unsigned long long); }

struct LzArgs {
    int n, G, nq, m0, m1, RW;             // RW: row capacity of a workgroup (LDS stride of a Q column)
    int qglobal;                          // 1: the deflation columns do not fit the LDS and are read from Q itself (stride n)
    const int* rp; const int* ci; const double* cv; const double* z;
    const double* Q;                      // nq columns, stride n
    double* V;                            // Lanczos basis, column j at V + j*n
    double* X;                            // exchange buffer (n)
    double* dalpha; double* dbeta;
    unsigned long long* slots;            // [LZ_GEN][LZ_GMAX][LZ_NV]
    int* err;
};

__device__ __forceinline__ double lz_ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const lz_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, LZ_CPOL_SC1);
    return __longlong_as_double(((long long)v.y << 32) | (long long)v.x);
}
__device__ __forceinline__ unsigned long long lz_ld_sc1_u64(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const lz_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, LZ_CPOL_SC1);
    return ((unsigned long long)v.y << 32) | (unsigned long long)v.x;
}
__device__ __forceinline__ void lz_st_sc1_u64(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned long long b) {
    lz_v2u v;
    v.x = (unsigned)(b & 0xffffffffULL); v.y = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte_off, 0, LZ_CPOL_SC1);
}

__global__ void k_lz_reset(unsigned long long* slots, int* err) {
    const int tot = LZ_GEN * LZ_GMAX * LZ_NV;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) slots[i] = LZ_SENT;
    if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0;
}

// Grid reduction of nv (<= LZ_NV) values.  In: vals[c] (LDS, this workgroup's partial of value c).  Out: tot[c]
// (LDS, grid total, identical bits in every workgroup).  All 8 waves poll: wave k owns the workgroups
// [k*G/8, (k+1)*G/8) and lane c (and c+64) sums value c over them in index order, one batch of loads = one
// round trip; the eight partial sums are then added in wave order.  part: LDS [LZ_PWAVES][LZ_NV].
__device__ __forceinline__ bool lz_sync(__amdgpu_buffer_rsrc_t rs, unsigned gen, int G, int nv, const double* vals,
                                        double* tot, double* part, double* flag, int* err) {
    if (threadIdx.x == 0) *flag = 0.0;
    __syncthreads();                                     // vals complete
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned gbase = (gen % LZ_GEN) * (unsigned)(LZ_GMAX * LZ_NV * 8);
    if (wave == 0) {
        // the reset stores of the previous sync must have been performed before this post can be seen
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < LZ_NV / 64; ++ps) {
            const int c = lane + 64 * ps;
            if (c < nv) lz_st_sc1_u64(rs, gbase + ((unsigned)blockIdx.x * LZ_NV + c) * 8u, (unsigned long long)__double_as_longlong(vals[c]));
        }
    }
    const int gpw = G / LZ_PWAVES, g_lo = wave * gpw;
    {
        // lane c serves the values c and c + 64 in the SAME batch of loads: one round trip per poll whatever nv is
        constexpr int NP = LZ_NV / 64;
        double s[NP];
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) s[ps] = 0.0;
            for (int g0 = 0; g0 < gpw; g0 += 8) {
                unsigned long long b[NP][8];
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int c = lane + 64 * ps;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int g = g_lo + g0 + u;
                        b[ps][u] = (c < nv && g0 + u < gpw) ? lz_ld_sc1_u64(rs, gbase + ((unsigned)g * LZ_NV + c) * 8u) : 0ULL;
                    }
                }
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int c = lane + 64 * ps;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (c < nv && g0 + u < gpw) {
                            ok = ok && (b[ps][u] != LZ_SENT);
                            s[ps] += __longlong_as_double((long long)b[ps][u]);
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");               // the loads must be re-issued on every poll
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > LZ_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) { *flag = 1.0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                break;
            }
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int c = lane + 64 * ps;
            if (c < nv) part[wave * LZ_NV + c] = s[ps];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < LZ_PWAVES; ++k) s += part[k * LZ_NV + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    if (wave == 0) {
        // every workgroup has posted this generation, hence finished reading the previous one: reset my slot of it
        const unsigned pbase = ((gen + LZ_GEN - 1) % LZ_GEN) * (unsigned)(LZ_GMAX * LZ_NV * 8);
#pragma unroll
        for (int ps = 0; ps < LZ_NV / 64; ++ps)
            lz_st_sc1_u64(rs, pbase + ((unsigned)blockIdx.x * LZ_NV + lane + 64 * ps) * 8u, LZ_SENT);
    }
    __syncthreads();
    return *flag == 0.0;
}

// partial (over my rows) of <col_c, x> for c in [c0, c0+64): thread (c, seg) sums an eighth of the rows
__device__ __forceinline__ double lz_coldot(const double* cols, size_t RW, int nrow, int c, int ncol, const double* x) {
    const int seg = threadIdx.x & 7;
    double acc = 0.0;
    if (c < ncol) {
        const double* qc = cols + (size_t)c * RW;
        for (int t = seg; t < nrow; t += 8) acc = fma(qc[t], x[t], acc);
    }
    return msdp_group_sum<8>(acc);
}

// QG: the deflation columns are read in place (a.qglobal); a template parameter so that the LDS form keeps its ds_read accesses
template <bool QG>
__global__ __launch_bounds__(LZ_PB) void k_lanczos_persist(LzArgs a) {
    extern __shared__ double lds[];
    const int nq = a.nq;
    double* Qs = lds;                                   // [nq][RW] deflation columns
    double* vs = Qs + (QG ? (size_t)0 : (size_t)nq * a.RW);   // [RW] v_j
    double* ws = vs + a.RW;                             // [RW] w
    double* vals = ws + a.RW;                           // [LZ_NV]
    double* tot = vals + LZ_NV;                         // [LZ_NV]
    double* part = tot + LZ_NV;                         // [LZ_PWAVES][LZ_NV]
    double* hvp = part + LZ_PWAVES * LZ_NV;             // [64]  Q'v_{j-1}
    double* flag = hvp + 64;                            // [8]
    // reduced values: [0, nq) = Q'w, [nq, 2nq) = Q'v_j, 2nq = v_j'w
    const int nv = 2 * nq + 1;
    const unsigned q = (unsigned)a.n / (unsigned)a.G, rem = (unsigned)a.n - q * (unsigned)a.G;
    const unsigned b = blockIdx.x;
    const int lo = (int)(b * q + (b < rem ? b : rem));
    const int hi = lo + (int)q + (b < rem ? 1 : 0);
    const int nrow = hi - lo;
    __amdgpu_buffer_rsrc_t rs_slots = __builtin_amdgcn_make_buffer_rsrc(a.slots, 0, (unsigned)(LZ_GEN * LZ_GMAX * LZ_NV * 8), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.X, 0, (unsigned)a.n * 8u, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_v0 = __builtin_amdgcn_make_buffer_rsrc(a.V + (size_t)a.m0 * a.n, 0, (unsigned)a.n * 8u, 0x00020000);

    // deflation columns: the workgroup's rows in LDS, or -- where (nq + 2) rows-per-workgroup doubles exceed it (n > ~117 000
    // with 40 columns) -- read in place: two passes over this workgroup's rows of Q per step, 2 n nq 8 bytes per step over the
    // grid, which the Infinity Cache holds (70 MB at n = 160 000, nq = 55)
    const double* __restrict__ Qc = QG ? a.Q + lo : (const double*)Qs;
    const size_t qst = QG ? (size_t)a.n : (size_t)a.RW;
    if (!QG)
        for (int c = 0; c < nq; ++c)
            for (int t = threadIdx.x; t < nrow; t += LZ_PB) Qs[(size_t)c * a.RW + t] = a.Q[(size_t)c * a.n + lo + t];
    double v[LZ_RMAX], vp[LZ_RMAX], w[LZ_RMAX], zr[LZ_RMAX];
    int rc[LZ_RMAX][LZ_KREG], cnt[LZ_RMAX], kbeg[LZ_RMAX];
    double rvv[LZ_RMAX][LZ_KREG];
#pragma unroll
    for (int r = 0; r < LZ_RMAX; ++r) {
        const int t = threadIdx.x + r * LZ_PB;
        const bool ok = t < nrow;
        v[r] = ok ? a.V[(size_t)a.m0 * a.n + lo + t] : 0.0;
        // the first LZ_KREG (col, val) pairs of the row stay in registers for the whole launch (S is static):
        // removes two dependent loads (rowptr -> col/val) from every step's S*v chain
        cnt[r] = 0;
        if (ok) {
            const int s0 = a.rp[lo + t], s1 = a.rp[lo + t + 1];
            cnt[r] = s1 - s0;
            kbeg[r] = s0;
#pragma unroll
            for (int u = 0; u < LZ_KREG; ++u) {
                const bool in = s0 + u < s1;
                rc[r][u] = in ? a.ci[s0 + u] : lo + t;
                rvv[r][u] = in ? a.cv[s0 + u] : 0.0;
            }
        } else {
            kbeg[r] = 0;
#pragma unroll
            for (int u = 0; u < LZ_KREG; ++u) { rc[r][u] = lo; rvv[r][u] = 0.0; }
        }
        vp[r] = (ok && a.m0 > 0) ? a.V[(size_t)(a.m0 - 1) * a.n + lo + t] : 0.0;
        zr[r] = ok ? a.z[lo + t] : 0.0;
        w[r] = 0.0;
        if (ok) ws[t] = vp[r];
    }
    double beta = a.m0 > 0 ? a.dbeta[a.m0] : 0.0;
    double sc = 1.0;                                    // gather source holds v_j / sc
    bool first = true;
    unsigned gen = 0;
    const int cth = threadIdx.x >> 3;                   // value index served by this thread's 8-lane group
    __syncthreads();
    // Q'v_{m0-1} (zero for m0 = 0): one extra reduction per launch
    if (nq > 0) {
        const double d0 = lz_coldot(Qc, qst, nrow, cth, nq, ws);
        if ((threadIdx.x & 7) == 0 && cth < LZ_NV) vals[cth] = d0;
        if (!lz_sync(rs_slots, gen++, a.G, nq, vals, tot, part, flag, a.err)) return;
        if ((int)threadIdx.x < 64) hvp[threadIdx.x] = ((int)threadIdx.x < nq) ? tot[threadIdx.x] : 0.0;
        __syncthreads();
    }
    for (int m = a.m0; m < a.m1; ++m) {
        // ---- w = S*v_j
#pragma unroll
        for (int r = 0; r < LZ_RMAX; ++r) {
            const int t = threadIdx.x + r * LZ_PB;
            if (t < nrow) {
                const int row = lo + t;
                double acc = 0.0;
                double xr[LZ_KREG];
#pragma unroll
                for (int u = 0; u < LZ_KREG; ++u) {
                    const unsigned off = (unsigned)rc[r][u] * 8u;
                    xr[u] = first ? lz_ld_sc1(rs_v0, off) : lz_ld_sc1(rs_x, off);
                }
#pragma unroll
                for (int u = 0; u < LZ_KREG; ++u) acc = fma(rvv[r][u], xr[u], acc);
                for (int k = kbeg[r] + LZ_KREG; k < kbeg[r] + cnt[r]; ++k) {       // rows longer than LZ_KREG entries
                    const unsigned off = (unsigned)a.ci[k] * 8u;
                    const double x = first ? lz_ld_sc1(rs_v0, off) : lz_ld_sc1(rs_x, off);
                    acc = fma(a.cv[k], x, acc);
                }
                (void)row;
                w[r] = sc * acc - zr[r] * v[r];
                ws[t] = w[r];
                vs[t] = v[r];
            }
        }
        // ---- partials of [Q'w ; Q'v ; v'w].  v'w: every thread owns its rows' products (wave sums + 8 LDS words);
        // Q'w and Q'v: thread (c, seg) walks an eighth of the rows once for both (one pass over the Q column).
        {
            double al = 0.0;
#pragma unroll
            for (int r = 0; r < LZ_RMAX; ++r) if ((int)threadIdx.x + r * LZ_PB < nrow) al = fma(w[r], v[r], al);
            al = msdp_wave_sum(al);
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = al;       // part[] is free between reductions
        }
        __syncthreads();
        {
            const int seg = threadIdx.x & 7;
            double hw = 0.0, hv = 0.0;
            if (cth < nq) {
                const double* qc = Qc + (size_t)cth * qst;
                for (int t = seg; t < nrow; t += 8) { const double qv = qc[t]; hw = fma(qv, ws[t], hw); hv = fma(qv, vs[t], hv); }
            }
            hw = msdp_group_sum<8>(hw);
            hv = msdp_group_sum<8>(hv);
            if (seg == 0 && cth < nq) { vals[cth] = hw; vals[nq + cth] = hv; }
            if (threadIdx.x == 0) {
                double s2 = 0.0;
                for (int i = 0; i < LZ_PWAVES; ++i) s2 += part[i];
                vals[2 * nq] = s2;
            }
        }
        if (!lz_sync(rs_slots, gen++, a.G, nv, vals, tot, part, flag, a.err)) return;
        const double alpha = tot[2 * nq];
        // deflation coefficients Q'(w - alpha v - beta v_prev), once per workgroup (vals[] is free until the next post)
        if ((int)threadIdx.x < nq) vals[threadIdx.x] = tot[threadIdx.x] - alpha * tot[nq + threadIdx.x] - beta * hvp[threadIdx.x];
        __syncthreads();
        // ---- w -= alpha v + beta v_prev + Q Q'(w - alpha v - beta v_prev);  |w|^2
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < LZ_RMAX; ++r) {
            const int t = threadIdx.x + r * LZ_PB;
            if (t < nrow) {
                double x = w[r] - alpha * v[r] - beta * vp[r];
                double dq = 0.0;
                for (int c = 0; c < nq; ++c) dq = fma(vals[c], Qc[(size_t)c * qst + t], dq);
                x -= dq;
                w[r] = x;
                nn = fma(x, x, nn);
                lz_st_sc1_u64(rs_x, (unsigned)(lo + t) * 8u, (unsigned long long)__double_as_longlong(x));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my rows of w are performed before my workgroup posts
        nn = msdp_wave_sum(nn);
        __syncthreads();                                           // everyone has used tot[] / hvp[] of the first reduction
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = nn;    // ws is free again
        if ((int)threadIdx.x < nq) hvp[threadIdx.x] = tot[nq + threadIdx.x];      // Q'v_j becomes Q'v_{j-1}
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int i = 0; i < LZ_PWAVES; ++i) s += ws[i];
            vals[0] = s;
        }
        if (!lz_sync(rs_slots, gen++, a.G, 1, vals, tot, part, flag, a.err)) return;
        const double n2 = tot[0];
        const double bnew = sqrt(n2 > 0.0 ? n2 : 0.0);
        const double inv = bnew > 0.0 ? 1.0 / bnew : 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.dalpha[m] = alpha; a.dbeta[m + 1] = bnew; }
        // ---- v_{j+1} = w / beta_{j+1}
#pragma unroll
        for (int r = 0; r < LZ_RMAX; ++r) {
            const int t = threadIdx.x + r * LZ_PB;
            if (t < nrow) {
                vp[r] = v[r];
                v[r] = w[r] * inv;
                a.V[(size_t)(m + 1) * a.n + lo + t] = v[r];
            }
        }
        sc = inv;
        beta = bnew;
        first = false;
        __syncthreads();                                           // tot[0] consumed before the next vals/tot traffic
    }
}

// Undeflated runs (nq = 0: the first escape call of a solve and the independent lambda_min check that precedes
// "Optimality is reached!" -- 33 000 steps on G81) with ONE grid synchronisation per step.  Before the reduction a
// workgroup knows w' = S*v_j - beta_j*v_{j-1} on its rows; it publishes those rows AND its rows of v_j, and reduces
// v_j'w', |w'|^2 and |v_j|^2 together.  After the reduction alpha = v_j'w' / |v_j|^2 and
// beta_{j+1}^2 = |w' - alpha*v_j|^2 = |w'|^2 - alpha^2 |v_j|^2 exactly, whatever the norm of v_j is (with the norm
// ASSUMED to be 1 a deviation eps of |v_j|^2 comes back as (alpha/beta)^2 * eps in |v_{j+1}|^2 -- a factor of four per
// step for a spectrum in [0, lambda_max]: measured, theta = -0.62 on G81).  alpha^2 and beta^2 are of the same order
// for a Lanczos process, so the difference keeps its digits; when it does not (beta^2 < 1e-8*|w'|^2: breakdown) the
// step falls back to a second reduction of the exact norm.  A neighbour
// entry of the NEW vector is formed where it is needed from the two published values,
// v_{j+1}[c] = (w'[c] - alpha*v_j[c]) / beta_{j+1}, with the same operations the owner of row c uses: bit-identical,
// no recurrence through S (a recurrence S*v_{j+1} = (S*w' - alpha*S*v_j)/beta would amplify rounding errors by
// alpha/beta per step).  The exchange buffers alternate with the parity of j: the rows a workgroup overwrites were last
// read before the previous synchronisation.
__device__ __forceinline__ double lz_next(double wv, double vv, double alpha, double inv) { return fma(-alpha, vv, wv) * inv; }

__global__ __launch_bounds__(LZ_PB) void k_lanczos_plain(LzArgs a) {
    __shared__ double vals[LZ_NV], tot[LZ_NV], part[LZ_PWAVES * LZ_NV], red[3 * LZ_PWAVES], flag[8];
    const unsigned q = (unsigned)a.n / (unsigned)a.G, rem = (unsigned)a.n - q * (unsigned)a.G;
    const unsigned b = blockIdx.x;
    const int lo = (int)(b * q + (b < rem ? b : rem));
    const int hi = lo + (int)q + (b < rem ? 1 : 0);
    const int nrow = hi - lo;
    __amdgpu_buffer_rsrc_t rs_slots = __builtin_amdgcn_make_buffer_rsrc(a.slots, 0, (unsigned)(LZ_GEN * LZ_GMAX * LZ_NV * 8), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.X, 0, (unsigned)a.n * 32u, 0x00020000);   // [w' even | w' odd | v even | v odd]
    __amdgpu_buffer_rsrc_t rs_v0 = __builtin_amdgcn_make_buffer_rsrc(a.V + (size_t)a.m0 * a.n, 0, (unsigned)a.n * 8u, 0x00020000);
    const unsigned nb = (unsigned)a.n * 8u;
    double v[LZ_RMAX], vp[LZ_RMAX], w[LZ_RMAX], zr[LZ_RMAX], xn[LZ_RMAX][LZ_KREG];
    int rc[LZ_RMAX][LZ_KREG], cnt[LZ_RMAX], kbeg[LZ_RMAX];
    double rvv[LZ_RMAX][LZ_KREG];
#pragma unroll
    for (int r = 0; r < LZ_RMAX; ++r) {
        const int t = threadIdx.x + r * LZ_PB;
        const bool ok = t < nrow;
        v[r] = ok ? a.V[(size_t)a.m0 * a.n + lo + t] : 0.0;
        vp[r] = (ok && a.m0 > 0) ? a.V[(size_t)(a.m0 - 1) * a.n + lo + t] : 0.0;
        zr[r] = ok ? a.z[lo + t] : 0.0;
        w[r] = 0.0;
        cnt[r] = 0; kbeg[r] = 0;
        const int s0 = ok ? a.rp[lo + t] : 0, s1 = ok ? a.rp[lo + t + 1] : 0;
        cnt[r] = s1 - s0; kbeg[r] = s0;
#pragma unroll
        for (int u = 0; u < LZ_KREG; ++u) {
            const bool in = s0 + u < s1;
            rc[r][u] = in ? a.ci[s0 + u] : lo;
            rvv[r][u] = in ? a.cv[s0 + u] : 0.0;
            xn[r][u] = lz_ld_sc1(rs_v0, (unsigned)rc[r][u] * 8u);            // neighbour entries of v_{m0} (previous launch / host)
        }
    }
    double beta = a.m0 > 0 ? a.dbeta[a.m0] : 0.0;
    double alpha_prev = 0.0, inv_prev = 0.0;          // the scalars that turn the published (w', v) of step m-1 into v_m
    unsigned gen = 0;
    for (int m = a.m0; m < a.m1; ++m) {
        const unsigned par = (unsigned)(m & 1), ppar = par ^ 1u;
        const bool first = m == a.m0;
        // ---- w' = S*v_m - beta_m*v_{m-1} on my rows; publish w' and v_m; partials of alpha and |w'|^2
        double al = 0.0, ww = 0.0, vv = 0.0;
#pragma unroll
        for (int r = 0; r < LZ_RMAX; ++r) {
            const int t = threadIdx.x + r * LZ_PB;
            if (t < nrow) {
                double acc = 0.0;
                vv = fma(v[r], v[r], vv);
#pragma unroll
                for (int u = 0; u < LZ_KREG; ++u) acc = fma(rvv[r][u], xn[r][u], acc);
                for (int k = kbeg[r] + LZ_KREG; k < kbeg[r] + cnt[r]; ++k) {       // rows longer than LZ_KREG entries
                    const unsigned off = (unsigned)a.ci[k] * 8u;
                    const double x = first ? lz_ld_sc1(rs_v0, off)
                                           : lz_next(lz_ld_sc1(rs_x, ppar * nb + off), lz_ld_sc1(rs_x, (2u + ppar) * nb + off), alpha_prev, inv_prev);
                    acc = fma(a.cv[k], x, acc);
                }
                const double wv = acc - zr[r] * v[r] - beta * vp[r];
                w[r] = wv;
                al = fma(wv, v[r], al);
                ww = fma(wv, wv, ww);
                lz_st_sc1_u64(rs_x, par * nb + (unsigned)(lo + t) * 8u, (unsigned long long)__double_as_longlong(wv));
                lz_st_sc1_u64(rs_x, (2u + par) * nb + (unsigned)(lo + t) * 8u, (unsigned long long)__double_as_longlong(v[r]));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my rows are performed before my workgroup posts
        al = msdp_wave_sum(al);
        ww = msdp_wave_sum(ww);
        vv = msdp_wave_sum(vv);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = al; red[LZ_PWAVES + (threadIdx.x >> 6)] = ww; red[2 * LZ_PWAVES + (threadIdx.x >> 6)] = vv; }
        __syncthreads();
        if (threadIdx.x < 3) {
            double s2 = 0.0;
            for (int i = 0; i < LZ_PWAVES; ++i) s2 += red[threadIdx.x * LZ_PWAVES + i];
            vals[threadIdx.x] = s2;
        }
        if (!lz_sync(rs_slots, gen++, a.G, 3, vals, tot, part, flag, a.err)) return;
        const double w2 = tot[1], v2 = tot[2];
        const double alpha = v2 > 0.0 ? tot[0] / v2 : 0.0;
        double n2 = w2 - alpha * alpha * v2;
        __syncthreads();                                           // tot[] read by everyone before the next reduction reuses it
        if (!(n2 >= 1e-8 * w2)) {
            // (near) breakdown: the difference has lost its digits -- reduce the exact norm (uniform branch: every
            // workgroup holds the same bits of alpha and |w'|^2)
            double nn = 0.0;
#pragma unroll
            for (int r = 0; r < LZ_RMAX; ++r)
                if ((int)threadIdx.x + r * LZ_PB < nrow) { const double x = fma(-alpha, v[r], w[r]); nn = fma(x, x, nn); }
            nn = msdp_wave_sum(nn);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nn;
            __syncthreads();
            if (threadIdx.x == 0) {
                double s2 = 0.0;
                for (int i = 0; i < LZ_PWAVES; ++i) s2 += red[i];
                vals[0] = s2;
            }
            if (!lz_sync(rs_slots, gen++, a.G, 1, vals, tot, part, flag, a.err)) return;
            n2 = tot[0];
            __syncthreads();
        }
        const double bnew = sqrt(n2 > 0.0 ? n2 : 0.0);
        const double inv = bnew > 0.0 ? 1.0 / bnew : 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.dalpha[m] = alpha; a.dbeta[m + 1] = bnew; }
        // ---- v_{m+1} on my rows and, for the next product, at my rows' neighbours
        const bool more = m + 1 < a.m1;
#pragma unroll
        for (int r = 0; r < LZ_RMAX; ++r) {
            const int t = threadIdx.x + r * LZ_PB;
            if (t < nrow) {
                const double vn = lz_next(w[r], v[r], alpha, inv);
                vp[r] = v[r];
                v[r] = vn;
                a.V[(size_t)(m + 1) * a.n + lo + t] = vn;
                if (more) {
                    double xw[LZ_KREG], xv[LZ_KREG];
#pragma unroll
                    for (int u = 0; u < LZ_KREG; ++u) {
                        const unsigned off = (unsigned)rc[r][u] * 8u;
                        xw[u] = lz_ld_sc1(rs_x, par * nb + off);
                        xv[u] = lz_ld_sc1(rs_x, (2u + par) * nb + off);
                    }
#pragma unroll
                    for (int u = 0; u < LZ_KREG; ++u) xn[r][u] = lz_next(xw[u], xv[u], alpha, inv);
                }
            }
        }
        alpha_prev = alpha; inv_prev = inv;
        beta = bnew;
    }
}

// ---------------------------------------------------------------- host side
static int lz_grid(int n, int nq, int* RW_out, size_t* lds_out, int* qglobal_out = nullptr, int force_qglobal = 0) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = v;
        else cus = 0;
        (void)hipGetLastError();
    }
    int gmax = (cus / 8) * 8;
    if (gmax > LZ_GMAX) gmax = LZ_GMAX;
    // The fewest workgroups with one row per thread (undeflated runs: a step is synchronisation latency plus one dependent gather: fewer
    // slots to poll is faster as long as no thread walks two rows -- G81 undeflated: 3.6 us per step on 40 workgroups,
    // 4.2 on 64, 4.9 on 32), more only where the rows or the LDS copy of Q demand it; two rows per thread as a last resort
    // deflated runs: the per-step LDS work on the rows of Q grows with the rows per workgroup; measured on G81 with
    // nq = 28..43: 64 workgroups 29.5 / 38.1 ms per run, 40: 32.5 / 41.5, 96: 34.5 / 44.6
    const int gfirst = nq > 0 ? 64 : 8;
    if (qglobal_out) *qglobal_out = 0;
    for (int rows_per_thread = 1; rows_per_thread <= LZ_RMAX; ++rows_per_thread)
        for (int G = gfirst; G <= gmax; G += 8) {
            const int rw = (n + G - 1) / G;
            if (rw > rows_per_thread * LZ_PB) continue;
            const size_t lds = ((size_t)(nq + 2) * rw + 2 * LZ_NV + LZ_PWAVES * LZ_NV + 64 + 8) * sizeof(double);
            if (lds > 150 * 1024) continue;
            *RW_out = rw;
            // option lanczos_qglobal (tests): the same grid with the columns read in place
            if (force_qglobal && nq > 0 && qglobal_out) { *qglobal_out = 1; *lds_out = ((size_t)2 * rw + 2 * LZ_NV + LZ_PWAVES * LZ_NV + 64 + 8) * sizeof(double); }
            else *lds_out = lds;
            return G;
        }
    // the deflation columns do not fit the LDS at any grid: read them in place, on all workgroups (their rows of Q stream
    // from the caches twice per step: the more workgroups, the more bandwidth)
    if (nq > 0 && qglobal_out && gmax >= 8) {
        const int rw = (n + gmax - 1) / gmax;
        const size_t lds = ((size_t)2 * rw + 2 * LZ_NV + LZ_PWAVES * LZ_NV + 64 + 8) * sizeof(double);
        if (rw <= LZ_RMAX * LZ_PB && lds <= 150 * 1024) { *RW_out = rw; *lds_out = lds; *qglobal_out = 1; return gmax; }
    }
    return 0;
}

// 1 when the persistent kernel can run the recurrence for this problem (sparse C, single rank, everything fits)
// full_csr: the caller holds ALL rows of C (the replicated escape of a row-sharded handle, msdp_escape.hip): the recurrence
// is a single-GPU computation on every rank then and the kernels apply as they are.  Not with the in-process ranks: N handles
// of one GPU would each need their workgroups co-resident.
int msdp_lanczos_persist_ok(msdp_handle h, int nq, const int* full_csr) {
    if (!h->tune.persist || h->persist_failed || h->d.costkind != COST_SPARSE) return 0;
    if (full_csr ? (h->lgroup != nullptr) : (h->nranks != 1 || h->use_comm || !h->d.rowptr)) return 0;
    if (2 * nq + 1 > LZ_NV || h->d.n < 64) return 0;
    int rw; size_t lds;
    int qg;
    const int G = lz_grid(h->d.n, nq, &rw, &lds, &qg, h->tune.lanczos_qglobal);
    if (G <= 0) return 0;
    static int attr_ok = -1;
    if (attr_ok < 0) {
        attr_ok = (hipFuncSetAttribute((const void*)k_lanczos_persist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                   hipFuncSetAttribute((const void*)k_lanczos_persist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess) ? 1 : 0;
        (void)hipGetLastError();
    }
    return attr_ok;
}

// Steps [m0, m1) of the recurrence; V[:, m0] (and V[:, m0-1], dbeta[m0] when m0 > 0) must be in place.
int msdp_lanczos_persist_run(msdp_handle h, const double* z, const double* Q, int nq, double* V, double* X, double* dalpha,
                             double* dbeta, unsigned long long* slots, int* err, int m0, int m1,
                             const int* rp, const int* ci, const double* cv) {
    LzArgs a;
    a.n = h->d.n; a.nq = nq; a.m0 = m0; a.m1 = m1;
    size_t lds;
    a.G = lz_grid(a.n, nq, &a.RW, &lds, &a.qglobal, h->tune.lanczos_qglobal);
    if (a.G <= 0) { msdp_set_error("persistent Lanczos: not eligible"); return MSDP_ESTATE; }
    a.rp = rp; a.ci = ci; a.cv = cv; a.z = z;             // all n rows of C (the handle's own CSR, or the replicated copy)
    a.Q = Q; a.V = V; a.X = X; a.dalpha = dalpha; a.dbeta = dbeta; a.slots = slots; a.err = err;
    hipLaunchKernelGGL(k_lz_reset, dim3(64), dim3(256), 0, h->stream, slots, err);
    HIPCHK(hipGetLastError());
    if (nq == 0 && h->tune.lanczos_onesync) {                 // X holds 4 n doubles (msdp_escape.hip)
        hipLaunchKernelGGL(k_lanczos_plain, dim3(a.G), dim3(LZ_PB), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (a.qglobal) hipLaunchKernelGGL(k_lanczos_persist<true>, dim3(a.G), dim3(LZ_PB), lds, h->stream, a);
    else hipLaunchKernelGGL(k_lanczos_persist<false>, dim3(a.G), dim3(LZ_PB), lds, h->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
