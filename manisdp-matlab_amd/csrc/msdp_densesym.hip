// msdp_densesym.hip -- the dense tall-skinny contraction for a SYMMETRIC matrix, reading the upper triangle only.
//
//   onlyunitdiag, dense C :  eH = C*U                       (ManiSDP_onlyunitdiag.m:128, dense symmetric C)
//   unitdiag / unittrace  :  eH = 2*eS*U + 4*sigma*AyU*Y    (ManiSDP_unitdiag.m:169, ManiSDP_unittrace.m:174; eS, AyU symmetric
//                                                            whenever the SeDuMi data is)
//
// k_dense_partial3 (msdp_dense.hip) streams all n x n entries: 8 n^2 bytes per product, the term that bounds the kernel up to
// p ~ 39.  Here a 16 x 16 tile T = S[I, J] above the diagonal is fetched ONCE and feeds both
//       out[I] += T  * X[J]        (direct:      the wave's stationary accumulators, as in k_dense_partial3)
//       out[J] += T' * X[I]        (transposed:  X[I] is stationary in registers; T' is the fragment turned through LDS)
// with v_mfma_f64_16x16x4_f64.  Work item = (matrix, block of RB rows, slice of the columns to the right of the block's
// first row); the RB x RB block on the diagonal is read whole (both triangles: 2-7 % of the bytes) and treated direct-only, so no
// tile ever needs a triangular mask.  The transposed contributions of the 8 waves of a workgroup to the same 16 rows of
// out[J] are added through LDS in wave order and written once per (row block, J tile) -- "T slabs": slab rb holds rows beyond
// its row block; the direct accumulators are written once per item ("D slabs").  k_sym_fold then adds, for every row, its T
// slabs in row-block order and its D slabs in slice order into ONE slab, which the unchanged epilogues of msdp_dense.hip /
// msdp_affine.hip read with SK = 1.  Nothing is atomic, every sum has a fixed order: results are bit-reproducible run to run.
// Traffic at n = 20000, p = 32: 1.6 GB of matrix + 0.23 GB of T slabs written and read + 0.07 GB of D slabs, against 3.2 GB.
#include "msdp_device.h"
#include <algorithm>
#include <cstring>
#include <vector>

typedef double sym_d4 __attribute__((ext_vector_type(4)));

int msdp_dev_alloc_bytes(msdp_handle h, void** out, size_t bytes);
int msdp_dense_ensure_slab(msdp_handle h, size_t need);       // msdp_dense.hip
int msdp_dense_nS(int n);

struct SymItem { int m, rb, k0, k1, dslot, pad0, pad1, pad2; };

struct SymOp {
    const double* M[2];          // n x nS row-major, symmetric
    const double* X[2];          // n x ld panels
    double scale[2];
    int n, nS, ld, ldl;
    int RB, nrb;
    int tslab0[2], dslab0;       // slab indices: T slab of (m, rb) = tslab0[m] + rb; D slab q = dslab0 + q
    double* slab;
    int64_t stride;
    const SymItem* items;
};

#define SYM_KT 16
#define SYM_TS 20                // row stride of the wave-private 16 x 16 transposition tile (doubles)

template <int NT> struct SymCfg {
    static constexpr int NC = 16 * NT;                       // panel columns of the launch (<= 32)
    static constexpr int RS = NC + 4;                        // row stride of a wave's 16 x NC block in the reduction buffer
    static constexpr int RSLOT = (16 * RS > 16 * SYM_TS) ? 16 * RS : 16 * SYM_TS;   // the transposition tile lives in the same slot
};
static int sym_ldl(int NT) { int ldl = 16 * NT; while ((ldl & 7) != 4) ldl += 2; return ldl; }
static size_t sym_lds_bytes(int NT, int waves, bool db) {
    const int RS = 16 * NT + 4;
    const int rslot = std::max(16 * RS, 16 * SYM_TS);
    return ((size_t)2 * SYM_KT * sym_ldl(NT) + (size_t)(db ? 2 : 1) * waves * rslot + 128) * sizeof(double);
}

// DB: the reduction buffer is double-buffered and a step needs ONE workgroup barrier instead of two (the reduction of step s runs
// beside the products of step s + 1 of the faster waves) -- for the shapes that leave one workgroup per CU anyway
// (__launch_bounds__' second figure is waves per SIMD: NT x RT = 4 holds 166 registers per lane -- three waves per SIMD, i.e. ONE 8-wave
// workgroup per CU, or one of 12 waves: round 6's shape.  Two 8-wave workgroups per CU need 128: with the fragments requested at the top
// of their step instead of a step ahead the kernel fits them with 92 bytes of scratch and runs 806 us where this one takes 596; two
// 6-wave workgroups per CU: 746 us; three of 4 waves (RB = 128: three times the T slabs of RB = 384): 630-635 us.)
template <int NT, int RT, int SYM_WAVES, bool DB>
__global__ __launch_bounds__(SYM_WAVES * 64, (SYM_WAVES == 16 ? 4 : (NT * RT >= 4 ? 3 : 4))) void k_dense_sym(SymOp op, const int* active_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (active_flag && !*active_flag) return;
    typedef SymCfg<NT> Cfg;
    constexpr int KT = SYM_KT, NC = Cfg::NC, RS = Cfg::RS, RSLOT = Cfg::RSLOT, TS = SYM_TS, NTHR = SYM_WAVES * 64;
    const SymItem it = op.items[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int n = op.n, nS = op.nS, ld = op.ld, ldl = op.ldl;
    const double* __restrict__ Mm = it.m ? op.M[1] : op.M[0];
    const double* __restrict__ Xm = it.m ? op.X[1] : op.X[0];
    const double sc = it.m ? op.scale[1] : op.scale[0];
    const int rbase = it.rb * op.RB + wave * (16 * RT);
    const int kdiag_end = (it.rb + 1) * op.RB;                // columns before this: the block on the diagonal, direct only
    double* stage = lds;                                      // 2 x KT x ldl panel tiles
    double* red = lds + 2 * KT * ldl;                         // (DB ? 2 : 1) x SYM_WAVES x RSLOT
    double* dummy = red + (DB ? 2 : 1) * SYM_WAVES * RSLOT;   // 128 doubles: store target of the threads that stage nothing
    for (int e = threadIdx.x; e < 2 * KT * ldl; e += NTHR) stage[e] = 0.0;

    // X[I] as the B operand of the transposed product: lane (g, i) holds X[I0 + 4g + t][16 nt + i]
    double UI[RT][4][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = rbase + 16 * rt + 4 * g + t, col = 16 * nt + i;
                const bool ok = row < n && col < ld;
                const double v = Xm[ok ? (int64_t)row * ld + col : 0];
                UI[rt][t][nt] = ok ? v : 0.0;
            }
    sym_d4 accD[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accD[rt][nt] = (sym_d4){0.0, 0.0, 0.0, 0.0};

    // staging: thread -> (panel row sr, column pair c2); one pass covers the KT rows
    constexpr int HP = NC / 2;
    const int c2 = threadIdx.x & (HP - 1), sr = threadIdx.x / HP;
    const bool sact = sr < KT && 2 * c2 < ld;
    double* const stg_dst0 = sact ? &stage[sr * ldl + 2 * c2] : &dummy[2 * lane];
    double* const stg_dst1 = sact ? &stage[KT * ldl + sr * ldl + 2 * c2] : &dummy[2 * lane];
    const double* arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = Mm + (int64_t)min(rbase + 16 * rt + i, n - 1) * nS + 4 * g;

    auto load_stg = [&](int k0) -> double2 {
        const int kk = k0 + sr;
        const bool ok = sact && k0 < it.k1 && kk < n;
        const double2 v = ld2(Xm + (ok ? (int64_t)kk * ld + 2 * c2 : 0));
        return ok ? v : make_double2(0.0, 0.0);
    };
    auto load_F = [&](int k0, double2 (&F)[RT][2]) {          // branch-free: steps beyond the slice read its first tile
        const int kc = k0 < it.k1 ? k0 : it.k0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { F[rt][0] = ld2(arow[rt] + kc); F[rt][1] = ld2(arow[rt] + kc + 2); }
    };
    double* tslab = op.slab + (int64_t)(op.tslab0[it.m ? 1 : 0] + it.rb) * op.stride;

    // products of one step: direct into accD, transposed into this wave's slot of `redb` (which also serves as the
    // wave-private tile the fragment is turned through); the staged panel tile of the NEXT step is stored at the end
    auto compute = [&](const double* bt, double* stg_dst, int k0, const double2 (&F)[RT][2], const double2 stg_next, double* redb) -> bool {
        const bool valid = k0 < it.k1;
        const double s = valid ? sc : 0.0;
        const bool tr = valid && k0 >= kdiag_end;             // workgroup-uniform
        double* myred = redb + wave * RSLOT;
        sym_d4 accT[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accT[nt] = (sym_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const double av[4] = {F[rt][0].x * s, F[rt][0].y * s, F[rt][1].x * s, F[rt][1].y * s};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double* brow = &bt[(4 * g + t) * ldl + i];
                double bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = brow[16 * nt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accD[rt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[nt], accD[rt][nt], 0, 0, 0);
            }
            if (tr) {
                // turn the fragment: lane (g, i) holds T[i][4g + t]; the transposed product needs T[4g + t][i]
                *reinterpret_cast<double2*>(&myred[i * TS + 4 * g]) = make_double2(av[0], av[1]);
                *reinterpret_cast<double2*>(&myred[i * TS + 4 * g + 2]) = make_double2(av[2], av[3]);
                __builtin_amdgcn_wave_barrier();
                double ft[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) ft[t] = myred[(4 * g + t) * TS + i];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) accT[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ft[t], UI[rt][t][nt], accT[nt], 0, 0, 0);
            }
        }
        if (tr) {
            // C/D layout: accT[nt][r] = (row g + 4r, column 16 nt + i) of this wave's contribution to out[k0 .. k0+15]
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) myred[(g + 4 * r) * RS + 16 * nt + i] = accT[nt][r];
        }
        *reinterpret_cast<double2*>(stg_dst) = stg_next;       // the tile of step k0 + KT; its buffer was last read in step k0 - KT
        return tr;
    };
    // the 8 (16) waves' contributions to out[k0 .. k0+15], added in wave order, written once
    auto reduce = [&](int k0, const double* redb) {
        for (int e = threadIdx.x; e < 16 * NC; e += NTHR) {
            const int j = e / NC, c = e - j * NC;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < SYM_WAVES; ++w) v += redb[w * RSLOT + j * RS + c];
            if (k0 + j < n && c < ld) tslab[(int64_t)(k0 + j) * ld + c] = v;
        }
    };

    double* red1 = DB ? red + SYM_WAVES * RSLOT : red;
    __syncthreads();                                          // zero fill done
    double2 F0[RT][2], F1[RT][2];
    load_F(it.k0, F0);
    { const double2 v = load_stg(it.k0); *reinterpret_cast<double2*>(stg_dst0) = v; }
    if (DB) __syncthreads();                                  // first tile staged
    // Round 6, what this loop was measured against at n = 20000, p = 32 (tools/densesym_shapes.py; whole Hess-vec, twelve waves 565-577 us
    // by box): without the transposed products 390 us, without the reductions 543, without the barriers 550 -- the direct half alone
    // streams its 1.6 GB at 4.8 TB/s (what the full kernel reaches on 3.2 GB) and the transposed half adds its 163 us of matrix
    // instructions nearly in full: the waves of the ONE workgroup a CU holds wait for their fragments together and multiply together.
    // Tried and not kept: the reduction of step s inside step s + 1, behind the first direct products (580 us); the fragments of three
    // steps in flight instead of one (eight waves, 215 registers: 606 against 607 us), requested as 256 contiguous bytes per row (629);
    // the barrier of step s in the MIDDLE of step s + 1 (behind its direct products; the panel tile staged two steps ahead in a third LDS
    // tile): by the cycle counter the first wave of a SIMD waits 3 700 of 9 500 cycles at the end-of-step barrier, with the barrier moved it
    // waits as long in the middle -- 575 against 567-577 us; the direct products with t outermost (a B fragment read once for both row
    // tiles, four accumulators rotating): no change.  The 48 direct matrix instructions of a SIMD's three waves take 4 400-4 800 cycles
    // (64 each at peak), the 48 transposed ones 3 000-4 000.
    for (int k0 = it.k0; k0 < it.k1; k0 += 2 * KT) {
        if (!DB) __syncthreads();                             // tile k0 staged; the reduction of the previous step has read `red`
        double2 sn = load_stg(k0 + KT);
        load_F(k0 + KT, F1);
        bool tr = compute(stage, stg_dst1, k0, F0, sn, red);
        __syncthreads();
        if (tr) reduce(k0, red);
        if (!DB) __syncthreads();
        sn = load_stg(k0 + 2 * KT);
        load_F(k0 + 2 * KT, F0);
        tr = compute(stage + KT * ldl, stg_dst0, k0 + KT, F1, sn, red1);
        __syncthreads();
        if (tr) reduce(k0 + KT, red1);
    }
    double* dsl = op.slab + (int64_t)(op.dslab0 + it.dslot) * op.stride;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 16 * nt + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + 16 * rt + g + 4 * r;
                if (row < n && col < ld) dsl[(int64_t)row * ld + col] = accD[rt][nt][r];
            }
        }
}

// out[row] = sum of the T slabs of the row blocks before row's own (matrix 0, then matrix 1), then the D slabs of its
// row block -- every sum in index order.  Workgroup b takes the rows [frow[b], frow[b+1]): ranges of equal COST (the last
// rows sum nrb slabs, the first ones a handful).
struct SymFold {
    const double* slab; int64_t stride;
    int ld, nmat, RB;
    int tslab0[2], dslab0;
    const int* qd; const int* frow;
    double* out;
};
__global__ __launch_bounds__(256) void k_sym_fold(SymFold f, const int* active_flag) {
    if (active_flag && !*active_flag) return;
    const int r0 = f.frow[blockIdx.x], r1 = f.frow[blockIdx.x + 1];
    const int half = f.ld >> 1;
    for (int e = threadIdx.x; e < (r1 - r0) * half; e += 256) {
        const int rl = e / half, cp = e - rl * half;
        const int row = r0 + rl;
        const int64_t o = (int64_t)row * f.ld + 2 * cp;
        const int rbk = row / f.RB;
        double2 acc = msdp_sum_slabs(f.slab + (int64_t)f.tslab0[0] * f.stride, f.stride, rbk, o);
        if (f.nmat == 2) {
            const double2 t = msdp_sum_slabs(f.slab + (int64_t)f.tslab0[1] * f.stride, f.stride, rbk, o);
            acc.x += t.x; acc.y += t.y;
        }
        const double2 dd = msdp_sum_slabs(f.slab + (int64_t)f.dslab0 * f.stride, f.stride, f.qd[rbk], o);
        acc.x += dd.x; acc.y += dd.y;
        st2(f.out + o, acc);
    }
}

// ------------------------------------------------------------------ host side
struct SymPlan {
    int NT = 0, nmat = 0, RT = 0, WV = 0, RB = 0, nrb = 0, n = 0;
    int nitems = 0, QDmax = 0, nslabs = 0, Gf = 0;
    int tslab0[2] = {0, 0}, dslab0 = 0;
    int len_opt = 0, res_opt = 0;
    SymItem* d_items = nullptr;
    int* d_qd = nullptr;
    int* d_frow = nullptr;
};
struct SymPlans { SymPlan p[2][2]; };      // [NT - 1][nmat - 1]

typedef void (*sym_fn_t)(SymOp, const int*);
static sym_fn_t sym_fn(int NT, int RT, int WV, bool db) {
    if (WV == 12) {
        if (db) return NT == 1 ? k_dense_sym<1, 2, 12, true> : k_dense_sym<2, 2, 12, true>;
        return NT == 1 ? k_dense_sym<1, 2, 12, false> : k_dense_sym<2, 2, 12, false>;
    }
    if (db) {
        if (WV == 16) return NT == 1 ? k_dense_sym<1, 1, 16, true> : k_dense_sym<2, 1, 16, true>;
        if (NT == 1) return RT == 2 ? k_dense_sym<1, 2, 8, true> : k_dense_sym<1, 1, 8, true>;
        return RT == 2 ? k_dense_sym<2, 2, 8, true> : k_dense_sym<2, 1, 8, true>;
    }
    if (WV == 16) return NT == 1 ? k_dense_sym<1, 1, 16, false> : k_dense_sym<2, 1, 16, false>;
    if (NT == 1) return RT == 2 ? k_dense_sym<1, 2, 8, false> : k_dense_sym<1, 1, 8, false>;
    return RT == 2 ? k_dense_sym<2, 2, 8, false> : k_dense_sym<2, 1, 8, false>;
}
// shape of a workgroup: dense_sym_rt = 1: 8 waves x 16 rows (RB = 128); 2: 8 waves x 32 rows (RB = 256); 3: 16 waves x 16 rows (RB = 256);
// round 6: 4: 12 waves x 32 rows (RB = 384: three waves per SIMD -- what 166 registers per lane allow -- in ONE workgroup per CU)
static void sym_shape(msdp_handle h, int NT, int* RT, int* WV) {
    int mode = h->tune.dense_sym_rt;
    // measured (tools/archive/densesym_probe.py, n = 20000: p = 16 551 us full / 414 / 384 / 444 for the shapes 1 / 2 / 3; p = 32: 613 /
    // 650 / 594 / 607; n = 10000, p = 32: 177 / 182 / 173 / 166)
    // (round 6, p = 32: n = 10000 164.6 / 167.3 / 153.4 us for the shapes 2 / 3 / 4, n = 20000 595.7 / 619.6 / 568.4, n = 40000 2401 / 2500 / 2236;
    // p = 16 keeps shape 2: 388.8 against 401.6 at n = 20000; tools/densesym_shapes.py)
    if (!mode) mode = NT == 1 ? 2 : (h->d.n >= 9000 ? 4 : 3);
    *RT = (mode == 2 || mode == 4) ? 2 : 1;
    *WV = mode == 3 ? 16 : (mode == 4 ? 12 : 8);
}

int msdp_densesym_eligible(msdp_handle h, int nmat) {
    const Dev& d = h->d;
    if (!h->tune.dense_sym || !h->dense_symmetric) return 0;
    if (h->nranks != 1 || h->use_comm || d.n_loc != d.n || d.row0 != 0) return 0;          // all rows on this rank
    if (d.blk_lo && h->tune.block_skip) return 0;
    if (nmat < 1 || nmat > 2 || d.ld > 32 || d.ld < 2) return 0;
    if (h->tune.dense_sym == 1 && d.n < h->tune.dense_sym_min) return 0;
    return 1;
}

static int sym_build(msdp_handle h, SymPlan& P, int NT, int nmat) {
    const Dev& d = h->d;
    const int n = d.n, nS = msdp_dense_nS(n);
    P.NT = NT; P.nmat = nmat; P.n = n;
    sym_shape(h, NT, &P.RT, &P.WV);
    P.RB = P.WV * 16 * P.RT;
    P.nrb = (n + P.RB - 1) / P.RB;
    // slice length L (steps of 16 columns, even): the fewest rounds of resident workgroups, then the shortest slices.  (Round 6 tried slices
    // of EQUAL length per row block, the target length chosen by simulating the longest-first schedule on the resident workgroups: 501 ->
    // 510 items and 146 -> 140 steps on the busiest CU for n = 20000 in the model; measured 565 -> 560 us there, 153 -> 163 us at n = 10000 and
    // 595 -> 611 for the 8 x 32-row shape -- an item costs more than the model's three steps, D slabs and fold work included.  Not kept.)
    // workgroups resident at a time: one per CU for 12 / 16 waves, two for 8 (dense_sym_res overrides: the 8 x 32-row shape at p > 16 holds
    // 166 registers per lane, i.e. ONE workgroup per CU)
    const int resident = h->tune.dense_sym_res > 0 ? h->tune.dense_sym_res : ((P.WV >= 12 || NT * P.RT >= 4) ? 256 : 512);
    std::vector<int> steps_rb(P.nrb);
    for (int rb = 0; rb < P.nrb; ++rb) { steps_rb[rb] = (nS - rb * P.RB) / 16; }
    int bestL = 0; double bestcost = 1e300;
    const int Lmin = 8;
    for (int L = Lmin; L <= std::max(Lmin, (nS / 16 + 1) & ~1); L += 2) {
        int64_t items = 0;
        for (int rb = 0; rb < P.nrb; ++rb) items += (int64_t)nmat * ((steps_rb[rb] + L - 1) / L);
        const double rounds = (double)((items + resident - 1) / resident);
        const double cost = rounds * (L + 3.0);                // + 3: the fixed work of an item (X[I] load, D slab, pipeline fill)
        if (cost < bestcost * 0.999) { bestcost = cost; bestL = L; }
        if (items <= resident / 2) break;
    }
    const int L = h->tune.dense_sym_len > 0 ? std::max(2, h->tune.dense_sym_len & ~1) : bestL;
    std::vector<SymItem> items;
    std::vector<int> qd(P.nrb, 0);
    for (int rb = 0; rb < P.nrb; ++rb)
        for (int m = 0; m < nmat; ++m)
            for (int s0 = 0; s0 < steps_rb[rb]; s0 += L) {
                SymItem it; memset(&it, 0, sizeof(it));
                it.m = m; it.rb = rb;
                it.k0 = rb * P.RB + 16 * s0;
                it.k1 = rb * P.RB + 16 * std::min(steps_rb[rb], s0 + L);
                it.dslot = qd[rb]++;
                items.push_back(it);
            }
    std::stable_sort(items.begin(), items.end(), [](const SymItem& a, const SymItem& b) { return (a.k1 - a.k0) > (b.k1 - b.k0); });
    P.nitems = (int)items.size();
    P.QDmax = *std::max_element(qd.begin(), qd.end());
    P.tslab0[0] = 2; P.tslab0[1] = 2 + P.nrb; P.dslab0 = 2 + nmat * P.nrb;
    P.nslabs = P.dslab0 + P.QDmax;
    // fold ranges of equal cost
    std::vector<int64_t> cost_rb(P.nrb);
    int64_t total = 0;
    for (int rb = 0; rb < P.nrb; ++rb) {
        const int rows = std::min(P.RB, n - rb * P.RB);
        cost_rb[rb] = (int64_t)(nmat * rb + qd[rb] + 1);
        total += cost_rb[rb] * rows;
    }
    int Gf = (int)std::min<int64_t>(2048, std::max<int64_t>(32, total * (d.ld / 2) / (256 * 24)));
    Gf = std::min(Gf, n);
    std::vector<int> frow(Gf + 1, n);
    {
        int b = 0; int64_t acc = 0;
        frow[0] = 0;
        for (int row = 0; row < n && b + 1 < Gf; ++row) {
            acc += cost_rb[row / P.RB];
            if (acc * Gf >= total * (int64_t)(b + 1)) { frow[++b] = row + 1; }
        }
        for (int q = b + 1; q <= Gf; ++q) frow[q] = n;
    }
    P.Gf = Gf;
    void* p = nullptr;
    int rc;
    if ((rc = msdp_dev_alloc_bytes(h, &p, items.size() * sizeof(SymItem)))) return rc;
    P.d_items = (SymItem*)p;
    HIPCHK(msdp_memcpy(P.d_items, items.data(), items.size() * sizeof(SymItem), hipMemcpyHostToDevice));
    if ((rc = msdp_dev_alloc_bytes(h, &p, qd.size() * sizeof(int)))) return rc;
    P.d_qd = (int*)p;
    HIPCHK(msdp_memcpy(P.d_qd, qd.data(), qd.size() * sizeof(int), hipMemcpyHostToDevice));
    if ((rc = msdp_dev_alloc_bytes(h, &p, frow.size() * sizeof(int)))) return rc;
    P.d_frow = (int*)p;
    HIPCHK(msdp_memcpy(P.d_frow, frow.data(), frow.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

static int sym_plan(msdp_handle h, int nmat, SymPlan** out) {
    if (!h->symplans) h->symplans = new SymPlans();
    SymPlans* S = (SymPlans*)h->symplans;
    const int NT = (h->d.ld + 15) / 16;
    SymPlan& P = S->p[NT - 1][nmat - 1];
    int want_rt, want_wv;
    sym_shape(h, NT, &want_rt, &want_wv);
    if (P.NT != NT || P.n != h->d.n || P.RT != want_rt || P.WV != want_wv || P.len_opt != h->tune.dense_sym_len || P.res_opt != h->tune.dense_sym_res) {
        int rc = sym_build(h, P, NT, nmat);
        if (rc) return rc;
        P.len_opt = h->tune.dense_sym_len;
        P.res_opt = h->tune.dense_sym_res;
    }
    *out = &P;
    return 0;
}
void msdp_densesym_release(msdp_handle h) {
    delete (SymPlans*)h->symplans;
    h->symplans = nullptr;
}

// Before any graph capture: build the plans of 1..nmat matrices at the current width and size the slab buffer.
int msdp_densesym_reserve(msdp_handle h, int nmat, size_t* slabs_out) {
    size_t need = 0;
    for (int q = 1; q <= nmat; ++q) {
        SymPlan* P;
        int rc = sym_plan(h, q, &P);
        if (rc) return rc;
        need = std::max(need, (size_t)P->nslabs);
    }
    *slabs_out = need;
    return 0;
}

// slab 0 <- sum_m scale[m] * M[m] * X[m]  (two launches: k_dense_sym, k_sym_fold); the caller's epilogue reads it with SK = 1
int msdp_densesym_gemm(msdp_handle h, hipStream_t stream, int nmat, const double* const* M, const double* const* X,
                       const double* scale, const int* active_flag) {
    const Dev& d = h->d;
    SymPlan* P;
    int rc = sym_plan(h, nmat, &P);
    if (rc) return rc;
    const int64_t stride = (int64_t)d.n * d.ld;
    if ((rc = msdp_dense_ensure_slab(h, (size_t)P->nslabs * stride))) return rc;
    SymOp op;
    memset(&op, 0, sizeof(op));
    for (int m = 0; m < nmat; ++m) { op.M[m] = M[m]; op.X[m] = X[m]; op.scale[m] = scale[m]; }
    if (nmat == 1) { op.M[1] = M[0]; op.X[1] = X[0]; op.scale[1] = 0.0; }
    op.n = d.n; op.nS = msdp_dense_nS(d.n); op.ld = d.ld; op.ldl = sym_ldl(P->NT);
    op.RB = P->RB; op.nrb = P->nrb;
    op.tslab0[0] = P->tslab0[0]; op.tslab0[1] = P->tslab0[1]; op.dslab0 = P->dslab0;
    op.slab = h->slab; op.stride = stride;
    op.items = P->d_items;
    // one barrier per step (double-buffered reduction) where the shape leaves one workgroup per CU anyway; dense_sym_db: 1 never, 2 always
    const bool db = h->tune.dense_sym_db == 2 || (h->tune.dense_sym_db == 0 && (P->WV >= 12 || P->NT * P->RT >= 4 || P->NT == 1));
    sym_fn_t fn = sym_fn(P->NT, P->RT, P->WV, db);
    if (!fn) { msdp_set_error("symmetric contraction: no kernel instance"); return MSDP_ESTATE; }
    const size_t ldsb = sym_lds_bytes(P->NT, P->WV, db);
    if (ldsb > 65536) HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(fn, dim3(P->nitems), dim3(P->WV * 64), ldsb, stream, op, active_flag);
    HIPCHK(hipGetLastError());
    SymFold f;
    memset(&f, 0, sizeof(f));
    f.slab = h->slab; f.stride = stride; f.ld = d.ld; f.nmat = nmat;
    f.RB = P->RB;
    f.tslab0[0] = P->tslab0[0]; f.tslab0[1] = P->tslab0[1]; f.dslab0 = P->dslab0;
    f.qd = P->d_qd; f.frow = P->d_frow; f.out = h->slab;
    hipLaunchKernelGGL(k_sym_fold, dim3(P->Gf), dim3(256), 0, stream, f, active_flag);
    HIPCHK(hipGetLastError());
    return 0;
}
