// msdp_window.hip -- S*U for vectors far beyond the L2s with the gathered rows staged ONCE per workgroup in LDS
// (ManiSDP_onlyunitdiag.m:127-130: eH = U*C; H = eH - Y.*sum(Y.*eH) - U.*eG; north_star's "LDS-staged p-wide panels").
//
// Why: k_hess_ell_obl gathers, for every row, the rows of U its row of C references straight from the L2 -- on a grid graph five
// 16-byte-per-lane row gathers per row, 1.9 KB per row through the L2 for 0.84 KB from HBM (n = 10^6, p = 32: 212 us = 0.49 of
// HBM, traffic already 1.00 x the algorithmic bytes: profiles/r4_pmc_hess_n1e6_p32.json).  Here the rows are cut into PATCHES of B
// rows that are close in the graph (breadth-first growth from the lowest unassigned row: on a 2-D grid a patch is a diamond-like
// tile whose halo is ~ 25 % of its rows, where a block of B consecutive rows has a halo of 2B).  A workgroup loads the window of a
// patch -- its own rows and the halo rows, each ONCE, whole 16-byte-per-lane rows, coalesced -- into LDS, and forms every row
// product from LDS with patch-local column indices.  Per row through the L2: (1 + halo / B) rows of U + the rows of Y and H + 60
// bytes of (index, value) pairs: 0.9 KB instead of 1.9.  The patches of the workgroups that run together on an XCD are consecutive
// (the traversal of msdp_sweep_rows), so a halo row is in that XCD's L2 when the neighbouring patch asks for it.
// Same fma order per row as spmm_row's ELL form: the result is bit-identical to k_hess_ell_obl (tests/test_gpu_onlyunitdiag.py).
#include "msdp_device.h"
#include <algorithm>
#include <deque>
#include <vector>

struct WinPlan {
    int npatch, EW, wmax;
    int64_t own_total;
    const int* poff;      // [npatch + 1]  offsets into wrows
    const int* pown;      // [npatch]      own rows of the patch = the first pown[p] rows of its window
    const int* ooff;      // [npatch + 1]  offsets of the patch's own rows into the (index, value) slices
    const int* wrows;     // window rows, GLOBAL row numbers (gather source d.full / the local vector at row - row0)
    const int* lidx;      // [EW][own_total]  patch-local position of the column
    const double* lval;   // [EW][own_total]
};

struct WinCacheEntry { int lpr = 0; int ld_max = 0; WinPlan plan{}; std::vector<void*> dev; bool failed = false; };
struct WinCache { WinCacheEntry e[8]; };

// H = proj-fused (C*md - Y.*rowdot(Y, C*md) - md.*eG) with the rows of md the patch touches staged in LDS; partial <md, Hmd>.
// Software pipeline over the patches of a workgroup: the window of the NEXT patch is requested into registers (NW row loads per
// lane group, all in flight) before the products of the current patch are formed from LDS buffer `cur`, and goes into the other
// buffer behind them -- the first version (load, barrier, compute, barrier, one row per trip of either loop) spent 35 us per patch
// in chains of dependent round trips: 306 us at n = 10^6, p = 32 against 208 us for the direct gathers.
#define WIN_NW 5                                               // window rows per lane group: wmax <= WIN_NW * MSDP_WAVES * 64 / LPR
// TWO (round 5, second form): two workgroups per CU, each with ONE window buffer (the next window waits in registers until the
// products of the current one are done: two barriers per patch, and the other resident workgroup's products fill the gaps); the
// launch has 2 x d.G workgroups and k_win_fold adds the partial sums of workgroup b + G to those of b (the consumers read d.G slots).
template <int LPR, int EWC, bool TWO>                         // EWC: the stored ELL width (5 or 8)
__global__ __launch_bounds__(MSDP_BLOCK) void k_hess_win_obl(Dev d, WinPlan w) {
    extern __shared__ double2 win[];                       // (TWO ? 1 : 2) x [wmax][LPR]
    __shared__ double sh[3 * MSDP_WAVES];
    if (!d.F[0].active) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    constexpr int RSTEP = MSDP_WAVES * RPW;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const bool colok = 2 * sub < d.ld;
    const int csub = colok ? 2 * sub : 0;
    const int cur = d.ctl->cur;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ eG = cur ? d.eG[1] : d.eG[0];
    const double* __restrict__ Uf = d.full;
    double* __restrict__ H = d.Hmd;
    const double2 zz = make_double2(0.0, 0.0);
    double pd = 0.0;
    // XCD x (workgroups b = x mod 8 under round-robin dispatch) walks the patches [x P / 8, (x + 1) P / 8); its G / 8 resident
    // workgroups take consecutive patches and advance together
    const int X = blockIdx.x & 7, s = blockIdx.x >> 3, S = (int)gridDim.x >> 3;
    const int p0 = (int)((int64_t)w.npatch * X / 8), p1 = (int)((int64_t)w.npatch * (X + 1) / 8);
    const int slot0 = wave * RPW + rsub;
    // (plain locals for what the lambdas below touch: a by-reference capture of a kernel-argument struct puts the struct into scratch)
    const int* __restrict__ poff = w.poff; const int* __restrict__ wrows = w.wrows;
    const int ld = d.ld, wmax = w.wmax;
    const bool nt = (d.sweep & 2) != 0;                    // streaming accesses for what the launch touches once (Y, the output), as in the direct kernel
    double2 nx[WIN_NW];
    // the rows of patch P this lane group stages: requested by GET_WINDOW, stored by PUT_WINDOW (macros, not lambdas: an array that
    // two lambdas capture by reference stays in scratch memory)
#define GET_WINDOW(P) do { \
        const int w0_ = poff[P], W_ = poff[(P) + 1] - w0_; \
        int rows_[WIN_NW]; \
        _Pragma("unroll") for (int q = 0; q < WIN_NW; ++q) { const int i_ = slot0 + q * RSTEP; rows_[q] = wrows[w0_ + (i_ < W_ ? i_ : 0)]; } \
        _Pragma("unroll") for (int q = 0; q < WIN_NW; ++q) nx[q] = ld2(Uf + (int64_t)rows_[q] * ld + csub); \
    } while (0)
#define PUT_WINDOW(BUF) do { \
        double2* dst_ = win + (size_t)(BUF) * wmax * LPR; \
        _Pragma("unroll") for (int q = 0; q < WIN_NW; ++q) { const int i_ = slot0 + q * RSTEP; double2 t_ = nx[q]; if (!colok) { t_.x = 0.0; t_.y = 0.0; } if (i_ < wmax) dst_[i_ * LPR + sub] = t_; } \
    } while (0)
    int p = p0 + s, buf = 0;
    if (p < p1) { GET_WINDOW(p); PUT_WINDOW(0); }
    __syncthreads();
    for (; p < p1; p += S) {
        const int pn = p + S;
        if (pn < p1) GET_WINDOW(pn);                           // in flight during the products of patch p
        const double2* wb = win + (size_t)buf * w.wmax * LPR;
        const int w0 = w.poff[p], nown = w.pown[p], o0 = w.ooff[p];
        constexpr int UN = 2;
        for (int i0 = wave * RPW; i0 < nown; i0 += UN * RSTEP) {
            int ic[UN], row[UN];
            bool rok[UN];
            int c[UN][EWC];
            double v[UN][EWC];
            double2 y[UN];
            double eg[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * RSTEP + rsub;
                rok[u] = i < nown;
                ic[u] = rok[u] ? i : 0;
                row[u] = w.wrows[w0 + ic[u]] - d.row0;     // local row
#pragma unroll
                for (int k = 0; k < EWC; ++k) {
                    c[u][k] = w.lidx[(int64_t)k * w.own_total + o0 + ic[u]];
                    v[u][k] = w.lval[(int64_t)k * w.own_total + o0 + ic[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) { y[u] = nt ? ld2_nt(Yl + (int64_t)row[u] * d.ld + csub) : ld2(Yl + (int64_t)row[u] * d.ld + csub); eg[u] = eG[row[u]]; }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                double2 acc = zz;
#pragma unroll
                for (int k = 0; k < EWC; ++k) {
                    const double2 x = wb[c[u][k] * LPR + sub];
                    acc.x = fma(v[u][k], x.x, acc.x);
                    acc.y = fma(v[u][k], x.y, acc.y);
                }
                if (!colok) { acc = zz; y[u] = zz; }
                const double2 uu = wb[ic[u] * LPR + sub];
                const double dot = msdp_group_sum<LPR>(acc.x * y[u].x + acc.y * y[u].y);
                if (rok[u] && colok) {
                    double2 hq;
                    hq.x = acc.x - y[u].x * dot - uu.x * eg[u];
                    hq.y = acc.y - y[u].y * dot - uu.y * eg[u];
                    if (nt) st2_nt(H + (int64_t)row[u] * d.ld + 2 * sub, hq); else st2(H + (int64_t)row[u] * d.ld + 2 * sub, hq);
                    pd += uu.x * hq.x + uu.y * hq.y;
                }
            }
        }
        if (TWO) {
            __syncthreads();                                   // everybody is done with the window
            if (pn < p1) PUT_WINDOW(0);
            __syncthreads();
        } else {
            if (pn < p1) PUT_WINDOW(buf ^ 1);                  // (nobody reads that buffer: its patch was finished one barrier ago)
            __syncthreads();
            buf ^= 1;
        }
    }
    msdp_put_partial(d.P, P_DHD, pd, sh);
}
#undef GET_WINDOW
#undef PUT_WINDOW
// the partial sums of the second round of workgroups onto those of the first (deterministic: one fixed pair per slot)
__global__ void k_win_fold(Dev d, int G) {
    if (!d.F[0].active) return;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < G) d.P[(size_t)P_DHD * MSDP_MAX_GRID + b] += d.P[(size_t)P_DHD * MSDP_MAX_GRID + b + G];
}

// ------------------------------------------------------------------ host: patches
// Breadth-first patches of at most B rows over the LOCAL rows (h_rowptr / h_colind hold ALL rows of C with global numbers; columns
// outside the rank's rows are halo by definition); a patch
// whose window exceeds wmax rows is cut in two (its rows are in breadth-first order: both halves stay connected-ish).
static void win_build_patches(msdp_handle h, int B, int wmax, std::vector<std::vector<int>>& own, std::vector<std::vector<int>>& halo) {
    const int nl = h->d.n_loc, r0 = h->d.row0;
    const std::vector<int>& rp = h->h_rowptr; const std::vector<int>& ci = h->h_colind;
    std::vector<int> asg((size_t)nl, -1);
    int nxt = 0;
    std::vector<std::vector<int>> raw;
    while (true) {
        while (nxt < nl && asg[nxt] >= 0) ++nxt;
        if (nxt >= nl) break;
        const int p = (int)raw.size();
        std::vector<int> o;
        std::deque<int> q;
        while ((int)o.size() < B) {
            if (q.empty()) {
                while (nxt < nl && asg[nxt] >= 0) ++nxt;
                if (nxt >= nl) break;
                asg[nxt] = p; o.push_back(nxt); q.push_back(nxt);
                continue;
            }
            const int r = q.front(); q.pop_front();
            for (int k = rp[r + r0]; k < rp[r + r0 + 1] && (int)o.size() < B; ++k) {
                const int c = ci[k] - r0;
                if (c >= 0 && c < nl && asg[c] < 0) { asg[c] = p; o.push_back(c); q.push_back(c); }
            }
        }
        raw.push_back(std::move(o));
    }
    // halo of a patch = the distinct columns (global numbers) of its rows that are not rows of the patch
    std::vector<int> mark((size_t)h->d.n, -1);
    std::deque<std::vector<int>> work(raw.begin(), raw.end());
    int stamp = 0;
    while (!work.empty()) {
        std::vector<int> o = std::move(work.front()); work.pop_front();
        ++stamp;
        for (int r : o) mark[r + r0] = stamp;
        std::vector<int> hl;
        for (int r : o)
            for (int k = rp[r + r0]; k < rp[r + r0 + 1]; ++k) { const int c = ci[k]; if (mark[c] != stamp) { mark[c] = stamp; hl.push_back(c); } }
        if ((int)(o.size() + hl.size()) > wmax && o.size() > 1) {
            std::vector<int> a(o.begin(), o.begin() + o.size() / 2), b(o.begin() + o.size() / 2, o.end());
            work.push_front(std::move(b)); work.push_front(std::move(a));
            continue;
        }
        std::sort(hl.begin(), hl.end());
        std::sort(o.begin(), o.end());                         // consecutive lane groups take ascending rows: runs of consecutive rows stay together in memory
        own.push_back(std::move(o)); halo.push_back(std::move(hl));
    }
}

template <class T>
static int win_upload(WinCacheEntry& e, const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return MSDP_ENOMEM; }
    e.dev.push_back(p);
    if (!v.empty() && msdp_memcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return MSDP_EHIP; }
    *out = (const T*)p;
    return 0;
}

static const size_t WIN_LDS_BYTES = 144 * 1024;            // one 1024-thread workgroup per CU: the two window buffers may take most of the 160 KB

static bool win_two(msdp_handle h) { return h->tune.window == 3 && 2 * h->d.G <= MSDP_MAX_GRID; }
static int win_get_plan(msdp_handle h, int lpr, WinCacheEntry** out) {
    if (!h->win) h->win = new WinCache();
    WinCacheEntry* e = nullptr;
    const int key = lpr + (win_two(h) ? 1000 : 0);
    for (auto& x : h->win->e) if (x.lpr == key) { e = &x; break; }
    if (!e) for (auto& x : h->win->e) if (x.lpr == 0) { e = &x; break; }
    if (!e) return MSDP_EUNSUPPORTED;
    *out = e;
    if (e->lpr == key) return e->failed ? MSDP_EUNSUPPORTED : 0;
    e->lpr = key;
    // two buffers of wmax rows in one workgroup per CU, or one buffer in each of two; a lane group stages at most WIN_NW rows of a window
    int wmax = (int)(std::min(WIN_LDS_BYTES, (size_t)h->tune.window_lds * 1024) / 2 / ((size_t)lpr * sizeof(double2)));
    wmax = std::min(wmax, WIN_NW * MSDP_WAVES * (64 / lpr));
    // target patch size: on a 2-D grid the halo of a breadth-first patch of B rows is ~ 2.5 sqrt(B) + a few rows
    const int B = std::max(32, (int)(wmax * 0.78) / 16 * 16);
    std::vector<std::vector<int>> own, halo;
    win_build_patches(h, B, wmax, own, halo);
    const int EW = h->d.ellW, r0 = h->d.row0;
    const int np = (int)own.size();
    std::vector<int> poff((size_t)np + 1, 0), pown((size_t)np, 0), ooff((size_t)np + 1, 0), wrows;
    int64_t own_total = 0, win_total = 0;
    for (int p = 0; p < np; ++p) { own_total += (int64_t)own[p].size(); win_total += (int64_t)(own[p].size() + halo[p].size()); }
    // a graph without locality (every row drags its own halo in) gains nothing from the staging: keep the direct gathers
    if (win_total > 3 * own_total) { e->failed = true; return MSDP_EUNSUPPORTED; }
    wrows.reserve((size_t)win_total);
    std::vector<int> lidx((size_t)EW * own_total, 0);
    std::vector<double> lval((size_t)EW * own_total, 0.0);
    std::vector<int> pos((size_t)h->d.n, -1);
    const std::vector<int>& rp = h->h_rowptr; const std::vector<int>& ci = h->h_colind; const std::vector<double>& cv = h->h_cval;
    int64_t o0 = 0;
    for (int p = 0; p < np; ++p) {
        poff[p] = (int)wrows.size(); pown[p] = (int)own[p].size(); ooff[p] = (int)o0;
        int i = 0;
        for (int r : own[p]) { pos[r + r0] = i++; wrows.push_back(r + r0); }
        for (int c : halo[p]) { pos[c] = i++; wrows.push_back(c); }
        for (size_t q = 0; q < own[p].size(); ++q) {
            const int r = own[p][q];
            int k = 0;
            for (int t = rp[r + r0]; t < rp[r + r0 + 1] && k < EW; ++t, ++k) { lidx[(size_t)k * own_total + o0 + q] = pos[ci[t]]; lval[(size_t)k * own_total + o0 + q] = cv[t]; }
            for (; k < EW; ++k) { lidx[(size_t)k * own_total + o0 + q] = (int)q; lval[(size_t)k * own_total + o0 + q] = 0.0; }   // ELL padding: (row, 0.0)
        }
        o0 += (int64_t)own[p].size();
    }
    poff[np] = (int)wrows.size(); ooff[np] = (int)o0;
    WinPlan& pl = e->plan;
    pl.npatch = np; pl.EW = EW; pl.wmax = wmax; pl.own_total = own_total;
    int rc = 0;
    if ((rc = win_upload(*e, poff, &pl.poff)) || (rc = win_upload(*e, pown, &pl.pown)) || (rc = win_upload(*e, ooff, &pl.ooff)) ||
        (rc = win_upload(*e, wrows, &pl.wrows)) || (rc = win_upload(*e, lidx, &pl.lidx)) || (rc = win_upload(*e, lval, &pl.lval))) {
        e->failed = true;
        msdp_set_error("windowed Hess-vec: plan upload failed");
        return rc;
    }
    if (h->tune.timing) fprintf(stderr, "[msdp window] lpr %d: %d patches of <= %d rows, windows %.2f x the rows (wmax %d)\n", lpr, np, B, (double)win_total / (double)own_total, wmax);
    return 0;
}

void msdp_window_release(msdp_handle h) {
    if (!h->win) return;
    for (auto& x : h->win->e) for (void* p : x.dev) (void)hipFree(p);
    delete h->win;
    h->win = nullptr;
}

// 1: the staged Hess-vec serves this handle at its current width (sparse C with ELL rows, oblique, p <= 64, and a plan whose windows
// stay below three times the rows); the plan is built at the first call per lane count
int msdp_window_eligible(msdp_handle h) {
    const Dev& d = h->d;
    if (!h->tune.window || d.costkind != COST_SPARSE || d.manifold != MANI_OBLIQUE || d.rowfree || d.ellW < 1 || d.ellW > MSDP_ELL_MAXW || h->h_rowptr.empty()) return 0;
    // automatic: from 3 * 2^22 vector entries on (where the streaming accesses start) and rows of more than 16 doubles -- measured at
    // n = 10^6: p = 32 197 us against 208 us for the direct gathers, p = 16 101.5 against 100.1 (no gain: rows of one 128-byte line)
    if (h->tune.window == 1 && (!(d.sweep & 2) || d.ld <= 16)) return 0;
    if (d.ld > 64) return 0;
    int lpr = 8;
    while (2 * lpr < d.ld) lpr <<= 1;
    WinCacheEntry* e = nullptr;
    return win_get_plan(h, lpr, &e) == 0 ? 1 : 0;
}

int msdp_window_hess(msdp_handle h) {
    const Dev& d = h->d;
    int lpr = 8;
    while (2 * lpr < d.ld) lpr <<= 1;
    WinCacheEntry* e = nullptr;
    int rc = win_get_plan(h, lpr, &e);
    if (rc) return rc;
    const bool two = win_two(h);
    const size_t lds = (size_t)(two ? 1 : 2) * e->plan.wmax * lpr * sizeof(double2);
    dim3 grid(two ? 2 * d.G : d.G), block(MSDP_BLOCK);
    typedef void (*fn_t)(Dev, WinPlan);
    const bool e5 = e->plan.EW == 5;
    fn_t fn;
    if (two) fn = lpr == 8 ? (e5 ? k_hess_win_obl<8, 5, true> : k_hess_win_obl<8, 8, true>) : (lpr == 16 ? (e5 ? k_hess_win_obl<16, 5, true> : k_hess_win_obl<16, 8, true>) : (e5 ? k_hess_win_obl<32, 5, true> : k_hess_win_obl<32, 8, true>));
    else fn = lpr == 8 ? (e5 ? k_hess_win_obl<8, 5, false> : k_hess_win_obl<8, 8, false>) : (lpr == 16 ? (e5 ? k_hess_win_obl<16, 5, false> : k_hess_win_obl<16, 8, false>) : (e5 ? k_hess_win_obl<32, 5, false> : k_hess_win_obl<32, 8, false>));
    static bool attr_set[12] = {false};                    // (not a stream operation: once per process and kernel, outside any graph capture)
    const int ai = ((lpr == 8 ? 0 : (lpr == 16 ? 1 : 2)) * 2 + (e5 ? 0 : 1)) * 2 + (two ? 1 : 0);
    if (!attr_set[ai]) { HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIN_LDS_BYTES)); attr_set[ai] = true; }
    hipLaunchKernelGGL(fn, grid, block, lds, h->stream, d, e->plan);
    if (two) hipLaunchKernelGGL(k_win_fold, dim3((d.G + 255) / 256), dim3(256), 0, h->stream, d, d.G);
    HIPCHK(hipGetLastError());
    return 0;
}
