// msdp_api.hip -- C ABI of libmanisdp_hip.so (include/manisdp_hip.h): handle life
// cycle, resident-point I/O, the host side of the device-resident RTR/tCG driver,
// the fine-grained parity entry points, RCCL sharding and measurement hooks.
#include "msdp_common.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local char g_err[1024] = "";
void msdp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* msdp_last_error(void) { return g_err; }
extern "C" const char* msdp_version(void) { return "manisdp_hip 0.1.0 (gfx950)"; }

// kernels.hip wrappers
int msdp_k_pack(msdp_handle h, const double* src, double* dst, int n, int p, int ld, bool colmajor);
int msdp_k_unpack(msdp_handle h, const double* src, double* dst, int n, int p, int ld, bool colmajor);
int msdp_k_proj_obl(msdp_handle h, const double* Y, const double* U, double* V);
int msdp_k_retr_obl(msdp_handle h, const double* Y, const double* U, double* Z, double alpha);
int msdp_k_set_active(msdp_handle h, int active);
int msdp_k_sum_to(msdp_handle h, int which, double* out);
int msdp_sphere_proj(msdp_handle h, const double* Y, const double* U, double* V);
int msdp_sphere_retr(msdp_handle h, const double* Y, const double* U, double* Z, double alpha);
int msdp_affine_setup(msdp_handle h, const int64_t* jc, const int64_t* ir, const double* pr,
                      const double* b, const double* c);
int msdp_affine_set_multipliers(msdp_handle h, const double* y, double sigma);
int msdp_affine_linesearch_cost(msdp_handle h, const double* Yt, double* val);
int msdp_dense_setup(msdp_handle h, const double* C);
int msdp_dense_reserve(msdp_handle h, int nmat);
int msdp_dense_setup_synthetic(msdp_handle h, uint64_t seed);
void msdp_affine_release(msdp_handle h);
int msdp_affine_setup_blocked(msdp_handle h, int nb, const int64_t* block_n, const int64_t* jc, const int64_t* ir, const double* pr,
                              const double* b, const double* c);        // msdp_affine.hip: multiblock kind, per-block storage
int msdp_affine_get_block(msdp_handle h, int64_t row0, int64_t nbk, double* S);
void msdp_densesym_release(msdp_handle h);                  // msdp_densesym.hip
void msdp_window_release(msdp_handle h);                    // msdp_window.hip
void msdp_block_eigs_release(msdp_handle h);                // msdp_blockjacobi.hip
int msdp_window_eligible(msdp_handle h);
int msdp_escape_impl(msdp_handle h, int k, double tol, int maxit, double* lam, double* V, double* lmax, int* iters,
                     const double* Mdev);
int msdp_dense_nS(int n);
void msdp_blockeig_release(msdp_handle h);
void msdp_affine_algo_cost(msdp_handle h, double* bytes, double* flops);

#define CHECK_H(h)                                          \
    if (!(h)) { msdp_set_error("null handle"); return MSDP_EINVAL; }

template <typename T>
static int dev_alloc(msdp_handle h, T** out, size_t count) {
    void* p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        msdp_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return MSDP_ENOMEM;
    }
    h->allocs.push_back(p);
    *out = (T*)p;
    return 0;
}
// Uncached (MTYPE UC) device memory for the words that workgroups on different XCDs exchange inside one launch:
// sc1 accesses to it skip the L2 look-up on both ends (tools/microbench_sync.hip: grid reduction 1.99 -> 1.24 us;
// 12.8 -> 10.0 us per tCG trip).  Falls back to plain hipMalloc where the flag is not supported.
// Uncached blocks come from a per-process pool and go back to it, never to the driver, while the process lives
// (msdp_release_cache frees the pool).  Round 3: uncached memory that was allocated and hipFree'd per handle left LATER handles
// of the process with corrupted buffers (a fresh handle's eG read back as garbage in two runs out of three of
// tests/test_gpu_blockeig.py once the block eigen-solver added a 20-MB uncached allocation per handle; the same tests pass with
// plain memory, and with this pool) -- memory whose caching attribute changes between owners is not safe to recycle here.
#include <mutex>
#include <map>
// Round 4: the pool is a set of ARENAS with a coalescing first-fit sub-allocator instead of one driver block per request.  A
// long-lived host (MATLAB) that cycles handles of varying sizes re-uses the same arenas -- freed blocks merge with their
// neighbours, so the pool grows to the high-water mark of what was live together, not with the number of distinct sizes
// (the per-request pool matched sizes within 2 x only and grew without bound).  Arenas go back to the driver only when NO
// uncached block of the process is live any more: then, beyond MSDP_UC_POOL_CAP bytes, largest first (msdp_destroy of the last
// handle), or all of them (msdp_release_cache) -- uncached pages never change owner while a handle that could be handed
// them lives.  tests/test_gpu_edge_cases.py::test_handle_churn_keeps_results_and_pool_bounded.
struct UcArena { char* base; size_t bytes; int dev; std::map<size_t, size_t> freemap; size_t live; };   // freemap: offset -> size
static std::mutex g_uc_mutex;
static std::vector<UcArena> g_uc_arenas;
static std::map<void*, std::pair<int, size_t>> g_uc_live;     // block -> (arena index, size)
static const size_t UC_ALIGN = 256, UC_ARENA_MIN = (size_t)32 << 20;
static size_t g_uc_cap = (size_t)1 << 30;                     // pool bytes kept when nothing is live (MSDP_UC_POOL_CAP, bytes)
static int g_uc_release = 0;                                  // MSDP_UC_RELEASE=1: arenas may go back to the driver (see msdp_uc_free)
static size_t uc_pool_bytes_locked() { size_t t = 0; for (auto& a : g_uc_arenas) t += a.bytes; return t; }
static int g_uc_direct = 0;                                   // MSDP_UC_POOL=0: one driver block per request, hipFree'd at once (the round-3
                                                              //   arrangement that corrupted later handles; kept for tools/uc_pool_stress.py only)
// Round 5: the exchange memory is FINE-GRAINED device memory (hipDeviceMallocFinegrained), not uncached (hipDeviceMallocUncached) any more.
// tools/uc_pool_stress.py, 300 handles per mode: uncached blocks that went back to the driver corrupt whoever receives their pages next --
// the same three handles wrong whether the block was hipFree'd as it was (mode 0), hipMemset + synchronised first (4), or rewritten line
// by line with cached stores and an L2 write-back / invalidate (5): a formerly-uncached page keeps something of its memory type that no
// access from user space clears.  Fine-grained blocks freed the same way: 0 of 300 wrong (mode 6), and the persistent trip is FASTER on
// them (G81, p = 32: 6.41 against 6.55 us; 146 100 against 142 500 Hess-vec/s per trustregions() call, profiles/r5_finegrained_vs_uncached.log).
// MSDP_UC_MEM=uncached restores the old memory type (then the arenas never go back to the driver, as in round 4).
static unsigned g_uc_flags = hipDeviceMallocFinegrained;     // MSDP_UC_MEM=uncached: hipDeviceMallocUncached; MSDP_UC_POOL=6 / 7: fine-grained (direct / arenas)
// Round 5 probes (tools/uc_pool_stress.py): what has to happen to a formerly-uncached block before hipFree for its pages to be safe in
// somebody else's hands?  4: hipMemset of the whole block + hipDeviceSynchronize; 5: every 128-byte line written by a kernel with plain
// (cached) stores, then an L2 write-back + invalidate by every wave (buffer_wbl2 sc1 / buffer_inv sc1), then hipDeviceSynchronize.
__global__ void k_uc_scrub(unsigned long long* p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0ULL;
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
}
void* msdp_uc_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    bytes = (bytes + UC_ALIGN - 1) / UC_ALIGN * UC_ALIGN;
    int dev = -1;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_uc_mutex);
    static bool env_read = false;
    if (!env_read) {
        env_read = true;
        const char* e = getenv("MSDP_UC_POOL_CAP"); if (e && *e) g_uc_cap = (size_t)strtoull(e, nullptr, 10);
        e = getenv("MSDP_UC_MEM"); if (e && !strcmp(e, "uncached")) g_uc_flags = hipDeviceMallocUncached;
        e = getenv("MSDP_UC_POOL"); if (e && *e >= '0' && *e <= '5' && *e != '1') g_uc_flags = hipDeviceMallocUncached;   // the probes of the old memory type
        e = getenv("MSDP_UC_RELEASE"); if (e && *e == '1') g_uc_release = 1; else if (e && *e == '0') g_uc_release = 0;
        else g_uc_release = g_uc_flags == hipDeviceMallocFinegrained ? 1 : 0;   // fine-grained pages are safe in anybody's hands
        // probes of tools/uc_pool_stress.py: 0 = direct (hipFree at destroy), 2 = direct + hipDeviceSynchronize before every free,
        // 3 = direct, uncached blocks never freed
        // round 5: 4 / 5 = direct, the block scrubbed before hipFree (see k_uc_scrub); 6 = direct, fine-grained instead of uncached memory;
        // 7 = the arenas, of fine-grained memory
        e = getenv("MSDP_UC_POOL");
        if (e && *e >= '0' && *e <= '7' && *e != '1') g_uc_direct = *e == '0' ? 1 : (*e == '7' ? 0 : *e - '0');
        if (e && (*e == '6' || *e == '7')) g_uc_flags = hipDeviceMallocFinegrained;
    }
    if (g_uc_direct) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, g_uc_flags) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (g_uc_direct == 3) g_uc_live[p] = {-1, bytes};     // registered with no arena: msdp_uc_free keeps it for ever
        if (g_uc_direct == 4 || g_uc_direct == 5) g_uc_live[p] = {-2, bytes};   // registered for its size: scrubbed in msdp_uc_free, hipFree'd by the caller
        return p;                                             // else not registered: msdp_uc_free returns false and the caller hipFree's it
    }
    for (int pass = 0; pass < 2; ++pass) {
        // best fit over the free ranges of this device's arenas
        int ba = -1; size_t boff = 0, bsz = (size_t)-1;
        for (size_t ai = 0; ai < g_uc_arenas.size(); ++ai) {
            UcArena& a = g_uc_arenas[ai];
            if (a.dev != dev) continue;
            for (auto& fr : a.freemap)
                if (fr.second >= bytes && fr.second < bsz) { ba = (int)ai; boff = fr.first; bsz = fr.second; }
        }
        if (ba >= 0) {
            UcArena& a = g_uc_arenas[ba];
            a.freemap.erase(boff);
            if (bsz > bytes) a.freemap[boff + bytes] = bsz - bytes;
            a.live += bytes;
            void* p = a.base + boff;
            g_uc_live[p] = {ba, bytes};
            return p;
        }
        if (pass == 1) break;
        void* p = nullptr;
        const size_t ab = std::max(bytes, UC_ARENA_MIN);
        if (hipExtMallocWithFlags(&p, ab, g_uc_flags) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        UcArena a; a.base = (char*)p; a.bytes = ab; a.dev = dev; a.live = 0; a.freemap[0] = ab;
        g_uc_arenas.push_back(a);
    }
    return nullptr;
}
static void uc_trim_locked(size_t keep) {                     // only ever called with no live block anywhere
    std::sort(g_uc_arenas.begin(), g_uc_arenas.end(), [](const UcArena& x, const UcArena& y) { return x.bytes > y.bytes; });
    while (!g_uc_arenas.empty() && uc_pool_bytes_locked() > keep) { (void)hipFree(g_uc_arenas.front().base); g_uc_arenas.erase(g_uc_arenas.begin()); }
}
bool msdp_uc_free(void* p) {                                  // true: p was an uncached block (now back in its arena)
    if (!p) return false;
    std::lock_guard<std::mutex> lk(g_uc_mutex);
    auto it = g_uc_live.find(p);
    if (it == g_uc_live.end()) { if (g_uc_direct == 2) (void)hipDeviceSynchronize(); return false; }
    if (it->second.first == -2) {                             // probe modes 4 / 5: scrub, then the caller hipFree's
        const size_t sz = it->second.second;
        (void)hipDeviceSynchronize();
        if (g_uc_direct == 4) (void)hipMemset(p, 0, sz);
        else hipLaunchKernelGGL(k_uc_scrub, dim3(256), dim3(256), 0, 0, (unsigned long long*)p, sz / 8);
        (void)hipDeviceSynchronize();
        g_uc_live.erase(it);
        return false;
    }
    if (it->second.first < 0) return true;                    // probe mode 3: leaked on purpose
    UcArena& a = g_uc_arenas[it->second.first];
    size_t off = (size_t)((char*)p - a.base), sz = it->second.second;
    a.live -= sz;
    auto nx = a.freemap.lower_bound(off);
    if (nx != a.freemap.end() && off + sz == nx->first) { sz += nx->second; nx = a.freemap.erase(nx); }
    if (nx != a.freemap.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == off) { off = pv->first; sz += pv->second; a.freemap.erase(pv); } }
    a.freemap[off] = sz;
    g_uc_live.erase(it);
    // Round 5: arenas of UNCACHED memory are never handed back to the driver while the process lives -- such pages corrupt whoever
    // receives them next (torch, a MATLAB gpuArray in the same process included), and no scrub of tools/uc_pool_stress.py is clean.
    // Arenas of fine-grained memory (the default now) go back beyond MSDP_UC_POOL_CAP when nothing is live and on msdp_release_cache.
    if (g_uc_release && g_uc_live.empty() && uc_pool_bytes_locked() > g_uc_cap) uc_trim_locked(g_uc_cap);   // indices are free to change: nothing is live
    return true;
}
void msdp_uc_release_pool() {
    std::lock_guard<std::mutex> lk(g_uc_mutex);
    if (!g_uc_release || !g_uc_live.empty()) return;          // a live handle owns uncached blocks: its arenas stay
    uc_trim_locked(0);
}
// Pool statistics: bytes the arenas hold, bytes handed out, number of arenas (tests)
extern "C" int msdp_debug_pool_stats(int64_t* pool_bytes, int64_t* live_bytes, int64_t* arenas) {
    std::lock_guard<std::mutex> lk(g_uc_mutex);
    size_t live = 0;
    for (auto& a : g_uc_arenas) live += a.live;
    if (pool_bytes) *pool_bytes = (int64_t)uc_pool_bytes_locked();
    if (live_bytes) *live_bytes = (int64_t)live;
    if (arenas) *arenas = (int64_t)g_uc_arenas.size();
    return 0;
}
extern "C" int msdp_debug_mem_info(int64_t* free_bytes, int64_t* total_bytes) {
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("hipMemGetInfo failed"); return MSDP_EHIP; }
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}
template <typename T>
static int dev_alloc_uncached(msdp_handle h, T** out, size_t count) {
    if (count == 0) count = 1;
    void* p = msdp_uc_alloc(count * sizeof(T));
    if (!p) return dev_alloc<T>(h, out, count);
    h->allocs.push_back(p);
    *out = (T*)p;
    return 0;
}
int msdp_dev_alloc_bytes(msdp_handle h, void** out, size_t bytes) {
    char* p = nullptr;
    int rc = dev_alloc<char>(h, &p, bytes);
    *out = p;
    return rc;
}
static void dev_free(msdp_handle h, void* p) {
    if (!p) return;
    for (size_t i = 0; i < h->allocs.size(); ++i)
        if (h->allocs[i] == p) { h->allocs.erase(h->allocs.begin() + i); break; }
    if (!msdp_uc_free(p)) (void)hipFree(p);
}

// The stream and the four pinned control blocks of a handle come from a small cache of the process (round 6): hipStreamCreate 2.8 ms,
// hipStreamDestroy 3.9 - 4.5 ms and the hipHostMalloc / hipHostFree pairs were 8 of the 163 ms of a G81 solve to KKT 1e-8, paid by every
// handle a host opens (rocprofv3 --hip-trace, tools/hip_api_totals.py).  A kit goes back when its handle is destroyed (the stream
// synchronised), at most HOST_KIT_MAX per process are kept, msdp_release_cache frees them.
struct HostKit { int dev; hipStream_t stream; Ctl* h_ctl; Frame* h_frame; volatile int* h_flags; volatile unsigned long long* h_status; };
static std::mutex g_kit_mutex;
static std::vector<HostKit> g_kits;
static const size_t HOST_KIT_MAX = 8;
static void host_kit_free(HostKit& k) {
    if (k.h_ctl) (void)hipHostFree(k.h_ctl);
    if (k.h_frame) (void)hipHostFree(k.h_frame);
    if (k.h_status) (void)hipHostFree((void*)k.h_status);
    if (k.h_flags) (void)hipHostFree((void*)k.h_flags);
    if (k.stream) (void)hipStreamDestroy(k.stream);
}
static bool host_kit_take(msdp_handle h) {
    int dev = -1;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_kit_mutex);
    for (size_t i = 0; i < g_kits.size(); ++i) {
        if (g_kits[i].dev != dev) continue;
        HostKit k = g_kits[i];
        g_kits.erase(g_kits.begin() + i);
        h->stream = k.stream; h->h_ctl = k.h_ctl; h->h_frame = k.h_frame; h->h_flags = k.h_flags; h->h_status = k.h_status;
        memset(h->h_ctl, 0, sizeof(Ctl)); memset(h->h_frame, 0, 2 * sizeof(Frame)); memset((void*)h->h_flags, 0, 64); memset((void*)h->h_status, 0, 64);
        return true;
    }
    return false;
}
static void host_kit_give(msdp_handle h) {
    HostKit k = {-1, h->stream, h->h_ctl, h->h_frame, h->h_flags, h->h_status};
    h->stream = nullptr; h->h_ctl = nullptr; h->h_frame = nullptr; h->h_flags = nullptr; h->h_status = nullptr;
    (void)hipGetDevice(&k.dev);
    const bool whole = k.stream && k.h_ctl && k.h_frame && k.h_flags && k.h_status;
    if (whole && hipStreamSynchronize(k.stream) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_kit_mutex);
        if (g_kits.size() < HOST_KIT_MAX) { g_kits.push_back(k); return; }
    }
    (void)hipGetLastError();
    host_kit_free(k);
}
void msdp_host_kits_release() {                               // msdp_release_cache
    std::lock_guard<std::mutex> lk(g_kit_mutex);
    for (auto& k : g_kits) host_kit_free(k);
    g_kits.clear();
}

static bool boundary_colmajor(msdp_handle h) { return h->kind == MSDP_KIND_UNITTRACE || h->kind == MSDP_KIND_GENERIC; }

static int rows_capacity(msdp_handle h) {
    // equal per-rank row count so the all-gather is one uniform RCCL call
    return (h->d.n + h->nranks - 1) / h->nranks;
}

static void choose_grid(msdp_handle h) {
    Dev& d = h->d;
    int half = d.ld / 2, lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    const int rows_per_step = MSDP_WAVES * (64 / lpr);
    int want = (rows_capacity(h) + rows_per_step - 1) / rows_per_step;
    // At most one workgroup per CU: beyond 256 some CUs get a second 1024-thread workgroup and the launch waits for it
    // (measured on G81 p=32: G=320 -> 27.4 us per tCG trip, G=256 -> 24.1 us).  Round 3 (tools/archive/grid_probe.py, option grid):
    // 512 workgroups -- two full rounds -- lose as well, at every size: n = 40 000, p = 40: 50.6 us per trip against 43.1 with
    // 256; n = 80 000, p = 40: 79.0 / 71.3; n = 250 000, p = 64: 390 / 378; n = 10^6, p = 32: 770 / 766 (the stand-alone S*U
    // kernel alone gains 8 % from 512 at n = 10^6 and loses 8 % at n = 40 000).
    const int gmax = MSDP_MAX_GRID;
    int G = ((want + 7) / 8) * 8;
    if (G < 8) G = 8;
    if (G > 256) G = 256;
    if (h->tune.grid > 0) G = std::min((gmax / 8) * 8, std::max(8, ((h->tune.grid + 7) / 8) * 8));    // A/B switch
    d.G = G;
    d.sweep = (h->tune.sweep >= 2 || (h->tune.sweep == 1 && (int64_t)rows_capacity(h) * d.ld >= ((int64_t)1 << 21))) ? 1 : 0;
    // bit 1: streaming (nt) accesses for the operands a gather launch touches once -- from 3 * 2^22 vector entries on (96 MB:
    // n = 250 000 at p = 32 loses 14 % with them, p = 64 and n = 10^6 at p = 16 gain 17 %), or with sweep = 3; bits 4-7: 64-row
    // steps per workgroup and window of the stand-alone Hess-vec, minus one
    if (d.sweep && (h->tune.sweep == 3 || (h->tune.sweep == 1 && (int64_t)rows_capacity(h) * d.ld >= ((int64_t)3 << 22)))) d.sweep |= 2;
    if (d.sweep && h->tune.sweep_k > 1) d.sweep |= (std::min(h->tune.sweep_k, 16) - 1) << 4;
}

// (Re)allocate every n_loc x ld vector for factor widths up to pcap.
int msdp_alloc_vectors(msdp_handle h, int pcap) {
    Dev& d = h->d;
    const int ldcap = ((pcap + 1) / 2) * 2;
    const size_t rows = (size_t)rows_capacity(h);
    const size_t cnt = rows * (size_t)ldcap;
    double** vecs[] = {&d.Y[0], &d.Y[1], &d.Gr[0], &d.Gr[1], &d.eta[0], &d.eta[1], &d.Heta[0], &d.Heta[1],
                       &d.r, &d.r2, &d.md, &d.md2, &d.Hmd, &d.W0, &d.W1};
    if (h->full_buf) dev_free(h, h->full_buf);
    h->full_buf = nullptr;
    d.full = nullptr;
    {
        // ONE allocation for the fifteen factor-sized vectors (15 hipMalloc calls were 15 of the 19 ms the first set_point of
        // a G81 solve took -- 7 % of the 0.22-s solve, tools/archive/g81_host_profile.py); each vector starts on a 256-byte boundary
        const size_t nvec = sizeof(vecs) / sizeof(vecs[0]);
        const size_t stride = (cnt + 31) / 32 * 32;
        if (h->vec_pool) dev_free(h, h->vec_pool);
        h->vec_pool = nullptr;
        for (double** v : vecs) *v = nullptr;
        int rc = dev_alloc<double>(h, &h->vec_pool, stride * nvec);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->vec_pool, 0, stride * nvec * sizeof(double), h->stream));
        size_t i = 0;
        for (double** v : vecs) *v = h->vec_pool + stride * (i++);
    }
    if (d.mdx) dev_free(h, d.mdx);
    d.mdx = nullptr;
    {
        // regions of one vector each: the EARLY trips of the persistent tCG alternate between the first two (msdp_persist.hip), the
        // one-reduction trips too and use the next two for their direct exchanges; the fused launch of msdp_pipe.h exchanges the
        // proposal's rows and the gradient rows of the two point slots through three more (round 6) -- where the persistent kernels can
        // apply at all (rows per rank within their reach)
        const size_t xcnt = (rows_capacity(h) <= 65536 ? 7 : 4) * cnt;
        int rc = dev_alloc_uncached<double>(h, &d.mdx, xcnt);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(d.mdx, 0, xcnt * sizeof(double), h->stream));
    }
    if (h->use_comm || h->nranks > 1) {
        int rc = dev_alloc<double>(h, &h->full_buf, cnt * (size_t)h->nranks);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->full_buf, 0, cnt * h->nranks * sizeof(double), h->stream));
        d.full = h->full_buf;
        for (int s2 = 0; s2 < 2; ++s2) {
            if (h->yfull[s2]) { dev_free(h, h->yfull[s2]); h->yfull[s2] = nullptr; }
            if (d.costkind != COST_AFFINE) continue;
            int rc2 = dev_alloc<double>(h, &h->yfull[s2], cnt * (size_t)h->nranks);
            if (rc2) return rc2;
            HIPCHK(hipMemsetAsync(h->yfull[s2], 0, cnt * h->nranks * sizeof(double), h->stream));
        }
    }
    h->pcap = pcap;
    h->ldcap = ldcap;
    return 0;
}

static int alloc_common(msdp_handle h) {
    Dev& d = h->d;
    int rc;
    // msdp_comm_init / msdp_debug_shard call this a second time (the row split changed): release the first set
    if (d.ctl) { dev_free(h, d.ctl); d.ctl = nullptr; }
    if (d.F) { dev_free(h, d.F); d.F = nullptr; }
    if (d.P) { dev_free(h, d.P); d.P = nullptr; }
    if (h->psync_slots) { dev_free(h, h->psync_slots); h->psync_slots = nullptr; }
    if (h->psync_err) { dev_free(h, h->psync_err); h->psync_err = nullptr; }
    if ((rc = dev_alloc<Ctl>(h, &d.ctl, 1))) return rc;
    if ((rc = dev_alloc<Frame>(h, &d.F, 2))) return rc;
    // behind the partial-sum arrays: the sums of the sharded one-all-reduce trip (msdp_trip1.hip) and its arrival counter
    const size_t p_doubles = (size_t)MSDP_NPART * MSDP_MAX_GRID + 4 + 4 * MSDP_XS_MAX_RANKS + 2;
    if ((rc = dev_alloc<double>(h, &d.P, p_doubles))) return rc;
    HIPCHK(hipMemset(d.ctl, 0, sizeof(Ctl)));
    HIPCHK(hipMemset(d.F, 0, 2 * sizeof(Frame)));
    HIPCHK(hipMemset(d.P, 0, p_doubles * sizeof(double)));
    d.xs = d.P + (size_t)MSDP_NPART * MSDP_MAX_GRID;
    d.xs_all = d.xs + 4;
    d.xcount = reinterpret_cast<unsigned*>(d.xs_all + 4 * MSDP_XS_MAX_RANKS);
    d.xn = h->nranks;
    {
        char* ps = nullptr;
        if ((rc = dev_alloc_uncached<char>(h, &ps, msdp_psync_bytes()))) return rc;
        h->psync_slots = (unsigned long long*)ps;
        if ((rc = dev_alloc<int>(h, &h->psync_err, 1))) return rc;
        HIPCHK(hipMemset(h->psync_err, 0, sizeof(int)));
    }
    const size_t rows = (size_t)rows_capacity(h);
    for (int s = 0; s < 2; ++s) {
        if (d.eG[s]) { dev_free(h, d.eG[s]); d.eG[s] = nullptr; }
        if ((rc = dev_alloc<double>(h, &d.eG[s], rows))) return rc;
        HIPCHK(hipMemset(d.eG[s], 0, rows * sizeof(double)));
    }
    return 0;
}

static int new_handle(int kind, int64_t n, msdp_handle* out) {
    if (!out) { msdp_set_error("out handle pointer is null"); return MSDP_EINVAL; }
    if (n <= 0 || n > 0x7fffffff) { msdp_set_error("matrix order n = %lld out of range", (long long)n); return MSDP_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        msdp_set_error("no HIP device visible: libmanisdp_hip has no CPU fallback");
        return MSDP_EHIP;
    }
    msdp_handle h = new msdp_handle_s();
    h->kind = kind;
    h->d.n = (int)n;
    h->d.n_loc = (int)n;
    h->d.row0 = 0;
    h->d.manifold = (kind == MSDP_KIND_UNITTRACE) ? MANI_SPHERE : (kind == MSDP_KIND_GENERIC ? MANI_EUCLID : MANI_OBLIQUE);
    {
        // the documented environment switches, read once per handle (msdp_set_option changes them afterwards)
        auto on = [](const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; };
        if (on("MSDP_NO_PERSIST")) h->tune.persist = 0;
        if (on("MSDP_NO_FUSED_RTR")) h->tune.fused_rtr = 0;
        if (on("MSDP_NO_PERSIST_PIPE")) h->tune.persist_pipe = 0;
        if (on("MSDP_NO_GRAPH")) h->tune.graph = 0;
        if (on("MSDP_TIMING")) h->tune.timing = 1;
        if (on("MSDP_ESC_DEBUG")) h->tune.esc_debug = 1;
        h->d.persist_ep = h->tune.persist_ep;
        if (const char* e = getenv("MSDP_AFFINE_ROUTE")) h->tune.affine_route = !strcmp(e, "gram") ? 2 : (!strcmp(e, "sddmm") ? 1 : 0);
    }
    hipError_t e = hipSuccess;
    if (!host_kit_take(h)) {
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_ctl, sizeof(Ctl), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_frame, 2 * sizeof(Frame), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_flags, 64, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_status, 64, hipHostMallocMapped);
    }
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_flag[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_flag[1], hipEventDisableTiming);
    if (e == hipSuccess) {
        *h->h_status = 0;
        void* dp = nullptr;
        e = hipHostGetDevicePointer(&dp, (void*)h->h_status, 0);
        h->d.status = (unsigned long long*)dp;
    }
    if (e != hipSuccess) {
        msdp_set_error("stream/event/pinned setup failed: %s", hipGetErrorString(e));
        delete h;
        return MSDP_EHIP;
    }
    *out = h;
    return 0;
}

// Upload the CSR rows [row0, row0+n_loc) of the host copy.
static int upload_sparse_rows(msdp_handle h) {
    Dev& d = h->d;
    const int r0 = d.row0, r1 = d.row0 + d.n_loc;
    const int base = h->h_rowptr[r0];
    const int64_t nnz = h->h_rowptr[r1] - base;
    std::vector<int> rp((size_t)rows_capacity(h) + 1);
    for (int i = 0; i <= d.n_loc; ++i) rp[i] = h->h_rowptr[r0 + i] - base;
    for (size_t i = d.n_loc + 1; i < rp.size(); ++i) rp[i] = rp[d.n_loc];
    if (h->d_rowptr) dev_free(h, h->d_rowptr);
    if (h->d_colind) dev_free(h, h->d_colind);
    if (h->d_cval) dev_free(h, h->d_cval);
    int rc;
    if ((rc = dev_alloc<int>(h, &h->d_rowptr, rp.size()))) return rc;
    if ((rc = dev_alloc<int>(h, &h->d_colind, (size_t)nnz))) return rc;
    if ((rc = dev_alloc<double>(h, &h->d_cval, (size_t)nnz))) return rc;
    HIPCHK(msdp_memcpy(h->d_rowptr, rp.data(), rp.size() * sizeof(int), hipMemcpyHostToDevice));
    if (nnz) {
        HIPCHK(msdp_memcpy(h->d_colind, h->h_colind.data() + base, nnz * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(msdp_memcpy(h->d_cval, h->h_cval.data() + base, nnz * sizeof(double), hipMemcpyHostToDevice));
    }
    d.rowptr = h->d_rowptr; d.colind = h->d_colind; d.cval = h->d_cval; d.nnz = nnz;
    // ELL copy when every row is short (fixed-degree graphs such as G81: W = 5)
    int W = 0;
    for (int i = 0; i < d.n_loc; ++i) W = std::max(W, rp[i + 1] - rp[i]);
    d.ellW = 0; d.ellc = nullptr; d.ellv = nullptr;
    if (W >= 1 && W <= 8) {
        // stored width 5 or 8 (the persistent tCG kernel is instantiated for these and loads every slice
        // without a branch); the padding entries are (own row, 0.0)
        W = W <= 5 ? 5 : 8;
        const size_t cap = (size_t)rows_capacity(h);
        std::vector<int> ec((size_t)W * cap);
        std::vector<double> ev((size_t)W * cap, 0.0);
        for (int w = 0; w < W; ++w)
            for (size_t i = 0; i < cap; ++i) ec[(size_t)w * cap + i] = (int)std::min<size_t>(i, d.n_loc ? d.n_loc - 1 : 0) + d.row0;
        for (int i = 0; i < d.n_loc; ++i)
            for (int t = rp[i]; t < rp[i + 1]; ++t) {
                const int w = t - rp[i];
                ec[(size_t)w * cap + i] = h->h_colind[base + t];
                ev[(size_t)w * cap + i] = h->h_cval[base + t];
            }
        if (h->d_ellc) dev_free(h, h->d_ellc);
        if (h->d_ellv) dev_free(h, h->d_ellv);
        if ((rc = dev_alloc<int>(h, &h->d_ellc, ec.size()))) return rc;
        if ((rc = dev_alloc<double>(h, &h->d_ellv, ev.size()))) return rc;
        HIPCHK(msdp_memcpy(h->d_ellc, ec.data(), ec.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(msdp_memcpy(h->d_ellv, ev.data(), ev.size() * sizeof(double), hipMemcpyHostToDevice));
        d.ellW = W; d.ell_stride = (int64_t)cap; d.ellc = h->d_ellc; d.ellv = h->d_ellv;
    }
    return 0;
}

extern "C" int msdp_set_device(int32_t device) {
    HIPCHK(hipSetDevice(device));
    return 0;
}
extern "C" int msdp_device_count(int32_t* count) {
    if (!count) return MSDP_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return 0;
}

extern "C" void msdp_rtr_default_opts(msdp_rtr_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->maxiter = 1000; o->maxinner = 100; o->mininner = 1;
    o->tolgradnorm = 1e-6; o->kappa = 0.1; o->theta = 1.0; o->rho_prime = 0.1;
    o->rho_regularization = 1e3; o->Delta_bar = -1.0; o->Delta0 = -1.0;
}

extern "C" int msdp_create_onlyunitdiag_csc(int64_t n, const int64_t* jc, const int64_t* ir,
                                            const double* pr, int32_t pcap, msdp_handle* out) {
    if (!jc || (!ir && jc[n] > 0) || (!pr && jc[n] > 0)) { msdp_set_error("null sparse arrays"); return MSDP_EINVAL; }
    msdp_handle h = nullptr;
    int rc = new_handle(MSDP_KIND_ONLYUNITDIAG, n, &h);
    if (rc) return rc;
    const int64_t nnz = jc[n];
    if (nnz > 0x7fffffff) { msdp_set_error("nnz(C) too large"); msdp_destroy(h); return MSDP_EINVAL; }
    h->h_rowptr.resize(n + 1);
    h->h_colind.resize(nnz);
    h->h_cval.assign(pr, pr + nnz);
    for (int64_t i = 0; i <= n; ++i) h->h_rowptr[i] = (int)jc[i];
    for (int64_t k = 0; k < nnz; ++k) {
        if (ir[k] < 0 || ir[k] >= n) { msdp_set_error("row index out of range"); msdp_destroy(h); return MSDP_EINVAL; }
        h->h_colind[k] = (int)ir[k];
    }
    h->d.costkind = COST_SPARSE;
    if ((rc = alloc_common(h)) || (rc = upload_sparse_rows(h)) || (rc = msdp_alloc_vectors(h, pcap > 0 ? pcap : 32))) {
        msdp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int msdp_create_onlyunitdiag_dense(int64_t n, const double* C, int32_t pcap, msdp_handle* out) {
    if (!C) { msdp_set_error("null C"); return MSDP_EINVAL; }
    msdp_handle h = nullptr;
    int rc = new_handle(MSDP_KIND_ONLYUNITDIAG, n, &h);
    if (rc) return rc;
    h->d.costkind = COST_DENSE;
    if ((rc = alloc_common(h)) || (rc = msdp_dense_setup(h, C)) || (rc = msdp_alloc_vectors(h, pcap > 0 ? pcap : 32))) {
        msdp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int msdp_create_onlyunitdiag_dense_synthetic(int64_t n, uint64_t seed, int32_t nranks, int32_t rank,
                                                       int32_t pcap, msdp_handle* out) {
    if (nranks < 1 || rank < 0 || rank >= nranks) { msdp_set_error("bad shard (%d of %d)", rank, nranks); return MSDP_EINVAL; }
    msdp_handle h = nullptr;
    int rc = new_handle(MSDP_KIND_ONLYUNITDIAG, n, &h);
    if (rc) return rc;
    h->d.costkind = COST_DENSE;
    h->nranks = nranks;
    h->rank = rank;
    h->presharded = true;
    const int cap = rows_capacity(h);
    h->d.row0 = std::min<int64_t>(n, (int64_t)rank * cap);
    h->d.n_loc = (int)std::min<int64_t>(cap, n - h->d.row0);
    if ((rc = alloc_common(h)) || (rc = msdp_dense_setup_synthetic(h, seed)) ||
        (rc = msdp_alloc_vectors(h, pcap > 0 ? pcap : 32))) {
        msdp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

// Test-only: fill the gather buffer of a pre-sharded handle that has NO communicator (a single process
// standing in for rank r of N) with all n rows of a host matrix, so that the non-square shard kernels can be
// checked against a full-size reference on one GPU.  With a communicator the all-gather does this.
extern "C" int msdp_debug_set_full_rows(msdp_handle h, const double* rows_host) {
    CHECK_H(h);
    if (!h->presharded || h->use_comm || !h->d.p) { msdp_set_error("debug_set_full_rows: pre-sharded, communicator-free handle with a point"); return MSDP_ESTATE; }
    Dev& d = h->d;
    const size_t cnt = (size_t)d.n * d.p;
    double* stage = nullptr;
    if (hipMalloc((void**)&stage, cnt * sizeof(double)) != hipSuccess) { msdp_set_error("staging alloc failed"); return MSDP_ENOMEM; }
    hipError_t e = msdp_memcpy_async(stage, rows_host, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream);
    int rc = 0;
    if (e != hipSuccess) { msdp_set_error("H2D failed"); rc = MSDP_EHIP; }
    if (!rc) rc = msdp_k_pack(h, stage, h->full_buf, d.n, d.p, d.ld, false);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    return rc;
}

extern "C" int msdp_create_affine(int32_t kind, int64_t n, int64_t m, const int64_t* at_jc, const int64_t* at_ir,
                                  const double* at_pr, const double* b, const double* c, int32_t pcap,
                                  msdp_handle* out) {
    if (kind != MSDP_KIND_UNITDIAG && kind != MSDP_KIND_UNITTRACE && kind != MSDP_KIND_GENERIC) { msdp_set_error("bad kind %d", kind); return MSDP_EINVAL; }
    if (!at_jc || !b || !c || m <= 0) { msdp_set_error("null/empty affine data"); return MSDP_EINVAL; }
    msdp_handle h = nullptr;
    int rc = new_handle(kind, n, &h);
    if (rc) return rc;
    h->d.costkind = COST_AFFINE;
    h->d.m = m;
    if ((rc = alloc_common(h)) || (rc = msdp_affine_setup(h, at_jc, at_ir, at_pr, b, c)) ||
        (rc = msdp_alloc_vectors(h, pcap > 0 ? pcap : 32))) {
        msdp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int msdp_create_multiblock(int32_t nb, const int64_t* block_n, int32_t nob, int64_t m, const int64_t* at_jc,
                                      const int64_t* at_ir, const double* at_pr, const double* b, const double* c,
                                      int32_t pcap, msdp_handle* out) {
    if (nb < 1 || !block_n || nob < 0 || nob > nb) { msdp_set_error("multiblock: bad block description"); return MSDP_EINVAL; }
    if (!at_jc || !b || !c || m <= 0) { msdp_set_error("null/empty affine data"); return MSDP_EINVAL; }
    // offsets of the blocks inside the direct sum (rows) and inside the concatenated vec (entries)
    std::vector<int64_t> r0((size_t)nb + 1, 0), e0((size_t)nb + 1, 0);
    for (int i = 0; i < nb; ++i) {
        if (block_n[i] < 1) { msdp_set_error("multiblock: block %d has order %lld", i, (long long)block_n[i]); return MSDP_EINVAL; }
        r0[i + 1] = r0[i] + block_n[i];
        e0[i + 1] = e0[i] + block_n[i] * block_n[i];
    }
    const int64_t N = r0[nb], E = e0[nb], nnz = at_jc[m];
    // Per-block storage (round 4): every dense operand is the concatenation of its diagonal blocks, memory and work ~ sum n_i^2.
    // Default from 16 blocks or N >= 4096 on; MSDP_MULTIBLOCK_BLOCKED = 0 / 1 forces the embedded N x N form / this one (tests).
    bool blocked = nb >= 16 || N >= 4096;
    if (const char* e = getenv("MSDP_MULTIBLOCK_BLOCKED")) { if (*e == '0') blocked = false; else if (*e == '1') blocked = true; }
    if (blocked) {
        if (N > 0x3fffffff) { msdp_set_error("multiblock: total order too large"); return MSDP_EUNSUPPORTED; }
        msdp_handle hb = nullptr;
        int rcb = new_handle(MSDP_KIND_UNITDIAG, N, &hb);
        if (rcb) return rcb;
        hb->d.costkind = COST_AFFINE;
        hb->d.m = m;
        if ((rcb = alloc_common(hb)) || (rcb = msdp_affine_setup_blocked(hb, nb, block_n, at_jc, at_ir, at_pr, b, c)) ||
            (rcb = msdp_alloc_vectors(hb, pcap > 0 ? pcap : 32))) { msdp_destroy(hb); return rcb; }
        hb->kind = MSDP_KIND_MULTIBLOCK;
        std::vector<unsigned char> rfb((size_t)N, 0);
        bool anyb = false;
        for (int i = nob; i < nb; ++i)
            for (int64_t a = r0[i]; a < r0[i + 1]; ++a) { rfb[(size_t)a] = 1; anyb = true; }
        if (anyb) {
            unsigned char* drf = nullptr;
            if ((rcb = dev_alloc<unsigned char>(hb, &drf, (size_t)N))) { msdp_destroy(hb); return rcb; }
            if (msdp_memcpy(drf, rfb.data(), (size_t)N, hipMemcpyHostToDevice) != hipSuccess) { msdp_set_error("multiblock: upload failed"); msdp_destroy(hb); return MSDP_EHIP; }
            hb->d.rowfree = drf;
        }
        *out = hb;
        return 0;
    }
    if (N > 46000) { msdp_set_error("multiblock: total order %lld too large for the embedded dense representation", (long long)N); return MSDP_EUNSUPPORTED; }
    // embed: entry (a, b) of block i -> entry (r0_i + a, r0_i + b) of the N x N direct sum (column-major vec index)
    auto embed = [&](int64_t e, int64_t* g) -> bool {
        if (e < 0 || e >= E) return false;
        const int i = (int)(std::upper_bound(e0.begin(), e0.end(), e) - e0.begin()) - 1;
        const int64_t l = e - e0[i], a = l % block_n[i], bb = l / block_n[i];
        *g = (r0[i] + a) + (r0[i] + bb) * N;
        return true;
    };
    std::vector<int64_t> ir((size_t)nnz);
    for (int64_t t = 0; t < nnz; ++t)
        if (!embed(at_ir[t], &ir[(size_t)t])) { msdp_set_error("multiblock: At row index out of range"); return MSDP_EINVAL; }
    std::vector<double> cN((size_t)N * N, 0.0);
    for (int i = 0; i < nb; ++i)
        for (int64_t bb = 0; bb < block_n[i]; ++bb)
            for (int64_t a = 0; a < block_n[i]; ++a)
                cN[(size_t)((r0[i] + a) + (r0[i] + bb) * N)] = c[e0[i] + a + bb * block_n[i]];
    msdp_handle h = nullptr;
    int rc = msdp_create_affine(MSDP_KIND_UNITDIAG, N, m, at_jc, ir.data(), at_pr, b, cN.data(), pcap, &h);
    if (rc) return rc;
    h->kind = MSDP_KIND_MULTIBLOCK;
    std::vector<unsigned char> rf((size_t)N, 0);
    bool any = false;
    for (int i = nob; i < nb; ++i)
        for (int64_t a = r0[i]; a < r0[i + 1]; ++a) { rf[(size_t)a] = 1; any = true; }
    if (any) {
        unsigned char* drf = nullptr;
        if ((rc = dev_alloc<unsigned char>(h, &drf, (size_t)N))) { msdp_destroy(h); return rc; }
        if (msdp_memcpy(drf, rf.data(), (size_t)N, hipMemcpyHostToDevice) != hipSuccess) { msdp_set_error("multiblock: upload failed"); msdp_destroy(h); return MSDP_EHIP; }
        h->d.rowfree = drf;
    }
    if (nb > 1) {                                          // block ranges per row: the dense contraction skips the zero off-diagonal blocks
        std::vector<int> lo((size_t)N), hi((size_t)N);
        for (int i = 0; i < nb; ++i)
            for (int64_t a = r0[i]; a < r0[i + 1]; ++a) { lo[(size_t)a] = (int)r0[i]; hi[(size_t)a] = (int)r0[i + 1]; }
        int *dlo = nullptr, *dhi = nullptr;
        if ((rc = dev_alloc<int>(h, &dlo, (size_t)N)) || (rc = dev_alloc<int>(h, &dhi, (size_t)N))) { msdp_destroy(h); return rc; }
        if (msdp_memcpy(dlo, lo.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            msdp_memcpy(dhi, hi.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { msdp_set_error("multiblock: upload failed"); msdp_destroy(h); return MSDP_EHIP; }
        h->d.blk_lo = dlo; h->d.blk_hi = dhi;
    }
    *out = h;
    return 0;
}

void msdp_escape_workspace_park(double* ptr, size_t cap_doubles);     // msdp_escape.hip
int msdp_dual_setup(msdp_handle h, const int64_t* at_jc, const int64_t* at_ir, const double* at_pr, const double* b, const double* c,
                    const double* dAAt, int32_t nf, const int64_t* b_jc, const int64_t* b_ir, const double* b_pr, const double* cf);
int msdp_dual_set_penalty_impl(msdp_handle h, double sigma, const double* wf_host);
int msdp_dual_outer_step_impl(msdp_handle h, double* scal_host, double* Af_host, double* z_host);
int msdp_dual_get_y_impl(msdp_handle h, double* y_host);

extern "C" int msdp_create_dual_unitdiag(int64_t n, int64_t m, const int64_t* at_jc, const int64_t* at_ir, const double* at_pr,
                                         const double* dAAt, const double* b, const double* c, int32_t nf, const int64_t* b_jc,
                                         const int64_t* b_ir, const double* b_pr, const double* cf, int32_t pcap, msdp_handle* out) {
    if (!at_jc || !at_ir || !at_pr || !dAAt || !b || !c || m <= 0) { msdp_set_error("dual_unitdiag: null/empty data"); return MSDP_EINVAL; }
    if (nf < 0 || (nf > 0 && (!b_jc || !b_ir || !b_pr || !cf))) { msdp_set_error("dual_unitdiag: bad free part"); return MSDP_EINVAL; }
    msdp_handle h = nullptr;
    int rc = msdp_create_affine(MSDP_KIND_UNITDIAG, n, m, at_jc, at_ir, at_pr, b, c, pcap, &h);
    if (rc) return rc;
    h->kind = MSDP_KIND_DUAL_UNITDIAG;
    if ((rc = msdp_dual_setup(h, at_jc, at_ir, at_pr, b, c, dAAt, nf, b_jc, b_ir, b_pr, cf))) { msdp_destroy(h); return rc; }
    *out = h;
    return 0;
}

extern "C" int msdp_dual_set_penalty(msdp_handle h, double sigma, const double* w) {
    CHECK_H(h);
    if (h->kind != MSDP_KIND_DUAL_UNITDIAG) { msdp_set_error("dual_set_penalty: not a dual handle"); return MSDP_ESTATE; }
    h->state_valid = false;
    h->gradnorm_valid = false;
    return msdp_dual_set_penalty_impl(h, sigma, w);
}

extern "C" int msdp_dual_outer_step(msdp_handle h, double* scal, double* Af, double* z) {
    CHECK_H(h);
    if (h->kind != MSDP_KIND_DUAL_UNITDIAG) { msdp_set_error("dual_outer_step: not a dual handle"); return MSDP_ESTATE; }
    if (!scal || !z) { msdp_set_error("dual_outer_step: null argument"); return MSDP_EINVAL; }
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    h->state_valid = false;
    int rc = msdp_dual_outer_step_impl(h, scal, Af, z);
    h->dual_valid = (rc == 0);
    return rc;
}

extern "C" int msdp_dual_get_y(msdp_handle h, double* y) {
    CHECK_H(h);
    if (h->kind != MSDP_KIND_DUAL_UNITDIAG || !h->dual_valid || !y) { msdp_set_error("dual_get_y: call msdp_dual_outer_step first"); return MSDP_ESTATE; }
    return msdp_dual_get_y_impl(h, y);
}

static void local_leave(msdp_handle h);       // in-process communicator stand-in, below
static void halo_release(msdp_handle h);

extern "C" int msdp_destroy(msdp_handle h) {
    if (!h) return 0;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) (void)ncclCommDestroy((ncclComm_t)h->comm);
    for (void* p : h->allocs) if (!msdp_uc_free(p)) (void)hipFree(p);
    for (int s2 = 0; s2 < 2; ++s2) if (h->ev_flag[s2]) (void)hipEventDestroy(h->ev_flag[s2]);
    for (int s2 = 0; s2 < 2; ++s2) if (h->chunk_execs[s2]) (void)hipGraphExecDestroy(h->chunk_execs[s2]);
    msdp_affine_release(h);
    msdp_densesym_release(h);
    msdp_window_release(h);
    msdp_block_eigs_release(h);
    halo_release(h);
    if (h->xr_paddr) (void)hipFree(h->xr_paddr);
    local_leave(h);
    if (h->lc_tmp) (void)hipFree(h->lc_tmp);
    if (h->esc_rp) (void)hipFree(h->esc_rp);
    if (h->esc_ci) (void)hipFree(h->esc_ci);
    if (h->esc_cv) (void)hipFree(h->esc_cv);
    if (h->esc_z) (void)hipFree(h->esc_z);
    msdp_escape_workspace_park(h->esc_mem, h->esc_cap);      // esc_prev lives inside it; kept for the next handle of the process
    if (h->lz_slots && !msdp_uc_free(h->lz_slots)) (void)hipFree(h->lz_slots);
    msdp_blockeig_release(h);
    if (h->esc_top) (void)hipFree(h->esc_top);
    if (h->xr_ev) (void)hipEventDestroy(h->xr_ev);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    host_kit_give(h);                                         // the stream and the pinned control blocks: to the next handle of the process
    delete h;
    return 0;
}

extern "C" int msdp_set_multipliers(msdp_handle h, const double* y, double sigma) {
    CHECK_H(h);
    if (h->d.costkind != COST_AFFINE) { msdp_set_error("set_multipliers: handle has no affine constraints"); return MSDP_ESTATE; }
    if (h->kind == MSDP_KIND_DUAL_UNITDIAG) { msdp_set_error("set_multipliers: dual handles take msdp_dual_set_penalty"); return MSDP_ESTATE; }
    h->state_valid = false;
    h->chunk_len = 0;      // sigma is baked into the captured launches: force a re-capture
    return msdp_affine_set_multipliers(h, y, sigma);
}

// ------------------------------------------------------------------ point I/O
static int host_cur(msdp_handle h) { return h->h_ctl->cur; }

// Stage a boundary-layout host matrix (local rows) into a device vector.
static int upload_rows(msdp_handle h, const double* host, double* dst) {
    Dev& d = h->d;
    const size_t cnt = (size_t)d.n_loc * d.p;
    double* stage = nullptr;
    hipError_t e = hipMalloc((void**)&stage, (cnt ? cnt : 1) * sizeof(double));
    if (e != hipSuccess) { msdp_set_error("staging alloc failed"); return MSDP_ENOMEM; }
    int rc = 0;
    if (boundary_colmajor(h)) {
        // n x p column-major; each rank reads its row block of every column
        if (h->nranks == 1) {
            e = msdp_memcpy_async(stage, host, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream);
        } else {
            e = msdp_memcpy2d_async(stage, (size_t)d.n_loc * sizeof(double), host + d.row0, (size_t)d.n * sizeof(double),
                                 (size_t)d.n_loc * sizeof(double), d.p, hipMemcpyHostToDevice, h->stream);
        }
    } else {
        e = msdp_memcpy_async(stage, host + (size_t)d.row0 * d.p, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream);
    }
    if (e != hipSuccess) { msdp_set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = MSDP_EHIP; }
    if (!rc) rc = msdp_k_pack(h, stage, dst, d.n_loc, d.p, d.ld, boundary_colmajor(h));
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    return rc;
}

static int download_rows(msdp_handle h, const double* src, double* host) {
    Dev& d = h->d;
    const size_t cnt = (size_t)d.n_loc * d.p;
    double* stage = nullptr;
    hipError_t e = hipMalloc((void**)&stage, (cnt ? cnt : 1) * sizeof(double));
    if (e != hipSuccess) { msdp_set_error("staging alloc failed"); return MSDP_ENOMEM; }
    int rc = msdp_k_unpack(h, src, stage, d.n_loc, d.p, d.ld, boundary_colmajor(h));
    if (!rc) {
        if (boundary_colmajor(h) && h->nranks > 1)
            e = msdp_memcpy2d_async(host + d.row0, (size_t)d.n * sizeof(double), stage, (size_t)d.n_loc * sizeof(double),
                                 (size_t)d.n_loc * sizeof(double), d.p, hipMemcpyDeviceToHost, h->stream);
        else
            e = msdp_memcpy_async(host + (boundary_colmajor(h) ? 0 : (size_t)d.row0 * d.p), stage, cnt * sizeof(double),
                               hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) { msdp_set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = MSDP_EHIP; }
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    return rc;
}

extern "C" int msdp_set_point(msdp_handle h, int32_t p, const double* Y) {
    CHECK_H(h);
    if (p < 1 || !Y) { msdp_set_error("set_point: p = %d, Y = %p", p, (const void*)Y); return MSDP_EINVAL; }
    if (p > 1024) { msdp_set_error("factor width p = %d exceeds the supported maximum of 1024", p); return MSDP_EUNSUPPORTED; }
    Dev& d = h->d;
    if (p > h->pcap) {
        int rc = msdp_alloc_vectors(h, p + 16);
        if (rc) return rc;
    }
    d.p = p;
    d.ld = ((p + 1) / 2) * 2;
    if (!h->use_comm && h->nranks == 1) d.full = d.md;   // overwritten per launch by allgather_rows
    choose_grid(h);
    if (d.costkind != COST_SPARSE) {
        int rc = msdp_dense_reserve(h, d.costkind == COST_AFFINE ? 2 : 1);
        if (rc) return rc;
    }
    h->h_ctl->cur = 0;
    // zero the slot so pad columns and pad rows are exactly zero
    HIPCHK(hipMemsetAsync(d.Y[0], 0, (size_t)rows_capacity(h) * h->ldcap * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(d.Y[1], 0, (size_t)rows_capacity(h) * h->ldcap * sizeof(double), h->stream));
    int rc = upload_rows(h, Y, d.Y[0]);
    if (rc) return rc;
    h->have_point = true;
    h->state_valid = false;
    h->gradnorm_valid = false;
    return 0;
}

int msdp_allreduce_array(msdp_handle h, double* buf, size_t count);                             // below (local stand-in only)
int msdp_k_fgram(msdp_handle h, const double* Y, double* part, int nblk, double* out);          // msdp_kernels.hip
int msdp_k_frotate(msdp_handle h, int cap, int r, int ldn, const double* Y, const double* Q, double* Yn);
int msdp_k_fappend(msdp_handle h, int cap, int k, int ldn, const double* Y, const double* V, double alpha, int normalize, double* Yn);

// The resident point has been rewritten into slot `slot` with width p: make it the current one
static int adopt_point(msdp_handle h, int slot, int p) {
    Dev& d = h->d;
    d.p = p;
    d.ld = ((p + 1) / 2) * 2;
    if (!h->use_comm && h->nranks == 1) d.full = d.md;
    choose_grid(h);
    if (d.costkind != COST_SPARSE) {
        int rc = msdp_dense_reserve(h, d.costkind == COST_AFFINE ? 2 : 1);
        if (rc) return rc;
    }
    h->h_ctl->cur = slot;
    h->state_valid = false;
    h->gradnorm_valid = false;
    return 0;
}

extern "C" int msdp_factor_gram(msdp_handle h, double* G) {
    CHECK_H(h);
    if (!h->have_point || !G) { msdp_set_error("factor_gram: no resident point / null out"); return MSDP_ESTATE; }
    Dev& d = h->d;
    const int ld = d.ld, p = d.p;
    int nblk = (int)std::min<int64_t>(64, std::max<int64_t>(1, (int64_t)(1 << 22) / ((int64_t)ld * ld)));
    double* buf = nullptr;
    if (hipMalloc((void**)&buf, ((size_t)nblk + 1) * ld * ld * sizeof(double)) != hipSuccess) { msdp_set_error("factor_gram: scratch alloc failed"); return MSDP_ENOMEM; }
    double* out = buf + (size_t)nblk * ld * ld;
    int rc = msdp_k_fgram(h, d.Y[host_cur(h)], buf, nblk, out);
    if (!rc && h->use_comm && h->lgroup) rc = msdp_allreduce_array(h, out, (size_t)ld * ld);
    else if (!rc && h->use_comm) {
        ncclResult_t r = ncclAllReduce(out, out, (size_t)ld * ld, ncclDouble, ncclSum, (ncclComm_t)h->comm, h->stream);
        if (r != ncclSuccess) { msdp_set_error("ncclAllReduce failed: %s", ncclGetErrorString(r)); rc = MSDP_ECOMM; }
    }
    if (!rc && msdp_memcpy2d_async(G, (size_t)p * sizeof(double), out, (size_t)ld * sizeof(double), (size_t)p * sizeof(double), p,
                                hipMemcpyDeviceToHost, h->stream) != hipSuccess) { msdp_set_error("factor_gram: D2H failed"); rc = MSDP_EHIP; }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(buf);
    return rc;
}

extern "C" int msdp_factor_rotate(msdp_handle h, int32_t r, const double* Q) {
    CHECK_H(h);
    if (!h->have_point || !Q) { msdp_set_error("factor_rotate: no resident point / null Q"); return MSDP_ESTATE; }
    Dev& d = h->d;
    if (r < 1 || r > d.p) { msdp_set_error("factor_rotate: r = %d outside 1..p = %d", r, d.p); return MSDP_EINVAL; }
    const int cur = host_cur(h), ldn = ((r + 1) / 2) * 2;
    double* qd = nullptr;
    if (hipMalloc((void**)&qd, (size_t)d.p * r * sizeof(double)) != hipSuccess) { msdp_set_error("factor_rotate: scratch alloc failed"); return MSDP_ENOMEM; }
    int rc = 0;
    if (msdp_memcpy_async(qd, Q, (size_t)d.p * r * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) { msdp_set_error("factor_rotate: H2D failed"); rc = MSDP_EHIP; }
    if (!rc) rc = msdp_k_frotate(h, rows_capacity(h), r, ldn, d.Y[cur], qd, d.Y[cur ^ 1]);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(qd);
    if (rc) return rc;
    return adopt_point(h, cur ^ 1, r);
}

extern "C" int msdp_factor_append(msdp_handle h, int32_t k, const double* V, double alpha, int32_t normalize) {
    CHECK_H(h);
    if (!h->have_point || !V) { msdp_set_error("factor_append: no resident point / null V"); return MSDP_ESTATE; }
    Dev& d = h->d;
    if (k < 1) { msdp_set_error("factor_append: k = %d", k); return MSDP_EINVAL; }
    if (d.p + k > h->pcap) { msdp_set_error("factor_append: width %d exceeds the allocated capacity %d (use msdp_set_point)", d.p + k, h->pcap); return MSDP_EUNSUPPORTED; }
    if (d.manifold != MANI_OBLIQUE && normalize) { msdp_set_error("factor_append: row normalisation is the oblique kinds'"); return MSDP_EUNSUPPORTED; }
    const int cur = host_cur(h), pn = d.p + k, ldn = ((pn + 1) / 2) * 2;
    double* vd = nullptr;
    if (hipMalloc((void**)&vd, (size_t)std::max(d.n_loc, 1) * k * sizeof(double)) != hipSuccess) { msdp_set_error("factor_append: scratch alloc failed"); return MSDP_ENOMEM; }
    int rc = 0;
    // my rows of every column of the n x k column-major V
    if (msdp_memcpy2d_async(vd, (size_t)d.n_loc * sizeof(double), V + d.row0, (size_t)d.n * sizeof(double), (size_t)d.n_loc * sizeof(double), k,
                         hipMemcpyHostToDevice, h->stream) != hipSuccess) { msdp_set_error("factor_append: H2D failed"); rc = MSDP_EHIP; }
    if (!rc) rc = msdp_k_fappend(h, rows_capacity(h), k, ldn, d.Y[cur], vd, alpha, normalize, d.Y[cur ^ 1]);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(vd);
    if (rc) return rc;
    return adopt_point(h, cur ^ 1, pn);
}

// Device-side copy of the resident point and back: lets a caller restart from the same point without another PCIe
// upload (bench.py: the start point of every timed step is already in HBM).
extern "C" int msdp_point_snapshot(msdp_handle h) {
    CHECK_H(h);
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    const size_t cnt = (size_t)rows_capacity(h) * h->ldcap;
    if (h->snap_cap < cnt) {
        if (h->snap) dev_free(h, h->snap);
        h->snap = nullptr; h->snap_cap = 0;
        int rc = dev_alloc<double>(h, &h->snap, cnt);
        if (rc) return rc;
        h->snap_cap = cnt;
    }
    HIPCHK(msdp_memcpy_async(h->snap, h->d.Y[host_cur(h)], cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->snap_p = h->d.p;
    return 0;
}
extern "C" int msdp_point_restore(msdp_handle h) {
    CHECK_H(h);
    if (!h->snap || h->snap_p != h->d.p || !h->have_point) { msdp_set_error("point_restore: no snapshot of the current width"); return MSDP_ESTATE; }
    const size_t cnt = (size_t)rows_capacity(h) * h->ldcap;
    h->h_ctl->cur = 0;
    HIPCHK(msdp_memcpy_async(h->d.Y[0], h->snap, cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->state_valid = false;
    h->gradnorm_valid = false;
    return 0;
}

extern "C" int msdp_get_point(msdp_handle h, double* Y) {
    CHECK_H(h);
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    return download_rows(h, h->d.Y[host_cur(h)], Y);
}

// Test-only: eta and Heta as the LAST tCG solve of msdp_rtr left them (tCG.m:95: [eta, Heta, ...] = tCG(...)); meaningful after
// a call with maxiter = 1 on the paths that hand the step over through global memory (chunked path, persistent kernel with the
// option fused_rtr = 0).  tCG keeps Heta = Hess(eta) by linearity (tCG.m:192-220); tests/test_gpu_onlyunitdiag.py bounds the
// deviation of the persistent kernel, whose Hess-vecs are assembled as C*r_new + beta*C*mdelta_old (msdp_persist.hip, TWOSYNC).
extern "C" int msdp_debug_get_tcg_step(msdp_handle h, double* eta, double* Heta) {
    CHECK_H(h);
    if (!h->have_point || !eta || !Heta) { msdp_set_error("debug_get_tcg_step: no resident point / null out"); return MSDP_ESTATE; }
    Frame f;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(msdp_memcpy(&f, &h->d.F[0], sizeof(Frame), hipMemcpyDeviceToHost));
    const int ix = f.eta_idx ? 1 : 0;
    int rc = download_rows(h, h->d.eta[ix], eta);
    if (!rc) rc = download_rows(h, h->d.Heta[ix], Heta);
    return rc;
}

// Every row of the resident point on every rank (one all-gather, then the download): the host loops of the row-sharded
// affine kinds run replicated on all ranks and need identical inputs for their rank / escape decisions.
extern "C" int msdp_get_point_all(msdp_handle h, double* Y) {
    CHECK_H(h);
    if (!h->have_point || !Y) { msdp_set_error("get_point_all: no resident point / null out"); return MSDP_ESTATE; }
    if (h->nranks == 1 && !h->use_comm) return download_rows(h, h->d.Y[host_cur(h)], Y);
    if (!h->use_comm) { msdp_set_error("get_point_all: needs a communicator"); return MSDP_ESTATE; }
    Dev& d = h->d;
    int rc = msdp_allgather_rows(h, d.Y[host_cur(h)]);
    if (rc) return rc;
    const size_t cnt = (size_t)d.n * d.p;
    double* stage = nullptr;
    if (hipMalloc((void**)&stage, (cnt ? cnt : 1) * sizeof(double)) != hipSuccess) { msdp_set_error("staging alloc failed"); return MSDP_ENOMEM; }
    rc = msdp_k_unpack(h, h->full_buf, stage, d.n, d.p, d.ld, boundary_colmajor(h));
    if (!rc && msdp_memcpy_async(Y, stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { msdp_set_error("D2H copy failed"); rc = MSDP_EHIP; }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    return rc;
}

extern "C" int msdp_get_p(msdp_handle h, int32_t* p) {
    CHECK_H(h);
    if (!p) return MSDP_EINVAL;
    *p = h->d.p;
    return 0;
}

extern "C" int msdp_set_option(msdp_handle h, const char* name, int32_t value) {
    CHECK_H(h);
    if (!name) { msdp_set_error("set_option: null name"); return MSDP_EINVAL; }
    Tuning& t = h->tune;
    if (!strcmp(name, "persist")) t.persist = value != 0;
    else if (!strcmp(name, "fused_rtr")) t.fused_rtr = value != 0;
    else if (!strcmp(name, "graph")) { t.graph = value != 0; h->chunk_len = 0; }
    else if (!strcmp(name, "affine_route")) { if (value < 0 || value > 2) { msdp_set_error("affine_route: 0 auto, 1 sddmm, 2 gram"); return MSDP_EINVAL; } t.affine_route = value; h->chunk_len = 0; h->state_valid = false; }
    else if (!strcmp(name, "timing")) t.timing = value != 0;
    else if (!strcmp(name, "esc_debug")) t.esc_debug = value != 0;
    else if (!strcmp(name, "escape_deflate")) t.escape_deflate = value != 0;
    else if (!strcmp(name, "escape_warm")) t.escape_warm = value != 0;
    else if (!strcmp(name, "escape_start_y")) t.escape_start_y = value != 0;
    else if (!strcmp(name, "xpersist")) t.xpersist = value != 0;
    else if (!strcmp(name, "persist_refresh")) t.persist_refresh = value > 0 ? value : 0;
    else if (!strcmp(name, "xtail")) t.xtail = value ? 1 : 0;
    else if (!strcmp(name, "xr_twolevel")) t.xr_twolevel = value ? 1 : 0;
    else if (!strcmp(name, "window")) { t.window = value < 0 ? 0 : (value > 3 ? 3 : value); h->chunk_len = 0; }
    else if (!strcmp(name, "window_lds")) { t.window_lds = value < 16 ? 16 : (value > 144 ? 144 : value); h->chunk_len = 0; msdp_window_release(h); }
    else if (!strcmp(name, "persist_early")) t.persist_early = value > 0 ? value : 0;
    else if (!strcmp(name, "persist_pipe")) t.persist_pipe = value > 0 ? 1 : 0;
    else if (!strcmp(name, "pipe_refresh")) t.pipe_refresh = value > 0 ? (value < 2 ? 2 : value) : 0;   // (1 would store the refresh rows of trip j + 1 into the regions a slower workgroup still gathers those of trip j from: msdp_pipe.h)
    else if (!strcmp(name, "pipe_local")) t.pipe_local = value > 0 ? 1 : 0;
    else if (!strcmp(name, "persist_goff")) t.persist_goff = value ? 1 : 0;
    else if (!strcmp(name, "persist_ep")) { t.persist_ep = value ? 1 : 0; h->d.persist_ep = t.persist_ep; h->persist_sig_fn = nullptr; }
    else if (!strcmp(name, "persist_slots")) { t.persist_slots = (value == 3 || value == 4) ? value : 0; h->d.persist_slots = t.persist_slots; }
    else if (!strcmp(name, "psync_backoff")) t.psync_backoff = value > 0 ? value : 0;
    else if (!strcmp(name, "psync8_backoff")) t.psync8_backoff = value > 0 ? (value > 255 ? 255 : value) : 0;
    else if (!strcmp(name, "affine_overlap")) { t.affine_overlap = value != 0; h->chunk_len = 0; }
    else if (!strcmp(name, "trip1")) { t.trip1 = value < 0 ? 0 : (value > 2 ? 2 : value); h->chunk_len = 0; }
    else if (!strcmp(name, "dense_sk")) t.dense_sk = value < 0 ? 0 : value;
    else if (!strcmp(name, "sweep_k")) { t.sweep_k = value < 1 ? 1 : value; choose_grid(h); h->chunk_len = 0; }
    else if (!strcmp(name, "sweep")) { t.sweep = value < 0 ? 0 : (value > 3 ? 3 : value); choose_grid(h); h->chunk_len = 0; }
    else if (!strcmp(name, "trip2")) { t.trip2 = value < 0 ? 0 : (value > 2 ? 2 : value); h->chunk_len = 0; }
    else if (!strcmp(name, "escape_method")) { if (value < 0 || value > 2) { msdp_set_error("escape_method: 0 auto, 1 lanczos, 2 block"); return MSDP_EINVAL; } t.escape_method = value; }
    else if (!strcmp(name, "be_width")) { if (value != 0 && value != 32 && value != 64 && value != 128) { msdp_set_error("be_width: 0, 32, 64 or 128"); return MSDP_EINVAL; } t.be_width = value; }
    else if (!strcmp(name, "be_degree")) t.be_degree = value > 0 ? value : 0;
    else if (!strcmp(name, "be_grid")) t.be_grid = value > 0 ? (value > MSDP_MAX_GRID ? MSDP_MAX_GRID : value) : 0;
    else if (!strcmp(name, "be_lpr")) { if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) { msdp_set_error("be_lpr: 0, 8, 16, 32 or 64"); return MSDP_EINVAL; } t.be_lpr = value; }
    else if (!strcmp(name, "lanczos_onesync")) t.lanczos_onesync = value != 0;
    else if (!strcmp(name, "lanczos_qglobal")) t.lanczos_qglobal = value != 0;
    else if (!strcmp(name, "block_skip")) t.block_skip = value != 0;
    else if (!strcmp(name, "halo_exchange")) { t.halo_exchange = value != 0; h->state_valid = false; }
    else if (!strcmp(name, "dense_pack")) { t.dense_pack = value != 0; h->chunk_len = 0; }
    else if (!strcmp(name, "affine_fuse")) { t.affine_fuse = value != 0; h->chunk_len = 0; h->state_valid = false; }
    else if (!strcmp(name, "affine_side")) { t.affine_side = value != 0; h->chunk_len = 0; }
    else if (!strcmp(name, "affine_broute")) { t.affine_broute = value != 0; h->chunk_len = 0; }
    else if (!strcmp(name, "dense_sym")) { t.dense_sym = value < 0 ? 0 : (value > 2 ? 2 : value); h->chunk_len = 0; if (h->have_point && h->d.costkind != COST_SPARSE && !h->blocked) { int rc = msdp_dense_reserve(h, h->d.costkind == COST_AFFINE ? 2 : 1); if (rc) return rc; } }
    else if (!strcmp(name, "dense_sym_min")) { t.dense_sym_min = value > 0 ? value : 0; h->chunk_len = 0; if (h->have_point && h->d.costkind != COST_SPARSE && !h->blocked) { int rc = msdp_dense_reserve(h, h->d.costkind == COST_AFFINE ? 2 : 1); if (rc) return rc; } }
    else if (!strcmp(name, "dense_sym_res")) { t.dense_sym_res = value > 0 ? value : 0; h->chunk_len = 0; if (h->have_point && h->d.costkind != COST_SPARSE && !h->blocked) { int rc = msdp_dense_reserve(h, h->d.costkind == COST_AFFINE ? 2 : 1); if (rc) return rc; } }
    else if (!strcmp(name, "dense_sym_rt")) { t.dense_sym_rt = (value >= 1 && value <= 4) ? value : 0; h->chunk_len = 0; if (h->have_point && h->d.costkind != COST_SPARSE && !h->blocked) { int rc = msdp_dense_reserve(h, h->d.costkind == COST_AFFINE ? 2 : 1); if (rc) return rc; } }
    else if (!strcmp(name, "dense_sym_db")) { t.dense_sym_db = (value >= 0 && value <= 2) ? value : 0; h->chunk_len = 0; }
    else if (!strcmp(name, "dense_sym_len")) { t.dense_sym_len = value > 0 ? value : 0; h->chunk_len = 0; if (h->have_point && h->d.costkind != COST_SPARSE && !h->blocked) { int rc = msdp_dense_reserve(h, h->d.costkind == COST_AFFINE ? 2 : 1); if (rc) return rc; } }
    else if (!strcmp(name, "debug_fail_persist")) t.fail_persist = value != 0;
    else if (!strcmp(name, "debug_xr_skip")) t.fail_xr = value != 0;
    else if (!strcmp(name, "debug_fail_block")) t.fail_block = value != 0;
    else if (!strcmp(name, "grid")) { t.grid = value > 0 ? value : 0; choose_grid(h); h->chunk_len = 0; }
    else { msdp_set_error("set_option: unknown option '%s'", name); return MSDP_EINVAL; }
    return 0;
}

extern "C" int msdp_get_kind(msdp_handle h, int32_t* kind) {
    CHECK_H(h);
    if (!kind) return MSDP_EINVAL;
    *kind = h->kind;
    return 0;
}

// send / receive lists of the halo exchange ("Halo exchange" below)
struct Halo {
    int N = 0;
    std::vector<int> send_cnt, send_off, recv_cnt, recv_off;     // per peer, in rows
    int send_rows = 0, recv_rows = 0;
    int* send_idx = nullptr;       // device: local row index of every row to send (peer-major)
    int* recv_idx = nullptr;       // device: global row index of every row to receive (peer-major)
    double* sendbuf = nullptr;     // device: send_rows x ldcap
    double* recvbuf = nullptr;     // device: recv_rows x ldcap
    int ldcap = 0;
    // cross-rank persistent kernels, round 5 ("push" exchange): every member's exchange buffer = [its own rows (cap)] [a slot for
    // every foreign row its rows of C reference, in the order of recv_idx].  A member stores its rows into its own buffer AND into the
    // halo slots of the members that reference them, so every gather is a local load with buffer-local indices:
    int* xr_colind = nullptr;      // device: the local CSR column indices remapped to buffer positions (c - row0, or cap + halo slot)
    int* xr_ellc = nullptr;        // device: the same for the ELL copy ([w][cap])
    int* xr_pq = nullptr;          // device: [2][n_loc] member that needs local row i (-1: none), up to two per row
    int* xr_pidx = nullptr;        // device: [2][n_loc] its buffer position there (cap_of_that_member + slot)
    bool xr_ok = false;            // false: some row is referenced by more than two other members -> lock-step trips
};

// ------------------------------------------------------------------ in-process stand-in for the communicator
// N handles of ONE process on ONE GPU, each driven by its own host thread, stand in for N ranks: the three collectives the
// library uses (all-reduce of a device array, all-gather of equal slabs) are carried out with a host barrier and device
// copies / a summation kernel between the handles' buffers.  Everything else -- the row partition, the row offsets into the
// replicated operator state, the lock-step tCG driver, the order and number of collective calls on every rank -- is the
// code of the RCCL run, so one GPU can execute the N-rank paths (tests/test_gpu_local_ranks.py).  Sums run in rank order on
// every member: all members obtain the same bits, as with ncclAllReduce.
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#define LOCAL_MAX_RANKS 8
// Round 5: the same group, with its members in DIFFERENT PROCESSES (msdp_comm_init_ipc) -- ranks on one GPU, or one rank per GPU of a
// node with peer access.  What the in-process members share through their common address space travels through two shared blocks here:
// a POSIX shared-memory segment for the host side (barrier, votes, halo list sizes, the IPC handle) and ONE device allocation of rank 0
// that every member maps through hipIpcOpenMemHandle (the ARENA: the slot regions and exchange buffer of the cross-rank persistent tCG,
// and one staging slab per rank for the collectives: a member copies its contribution into its slab, the others read it there).  The
// collectives, the lock-step driver and the cross-rank persistent tCG above them are the code of the in-process group, call for call.
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#define IPC_MAGIC 0x4d53445049504331ULL   // "MSDPIPC1": written by rank 0 when the cleared segment is ready
struct IpcShared {
    std::atomic<unsigned long long> magic;
    std::atomic<int> arrived; std::atomic<unsigned long long> gen; std::atomic<int> broken;
    std::atomic<int> attached; std::atomic<int> arena_ready;
    hipIpcMemHandle_t arena;
    unsigned long long arena_bytes, stage_bytes, rows_doubles, slot_bytes;
    hipIpcMemHandle_t rows_handle[LOCAL_MAX_RANKS];          // every member's exchange buffer (its own allocation, on its own device)
    hipIpcMemHandle_t blk_handle[LOCAL_MAX_RANKS];           // every member's two-level synchronisation block (msdp_psync.h psync2; round 6)
    char devid[LOCAL_MAX_RANKS][32];                         // the PCI bus id of every member's device
    int vote[LOCAL_MAX_RANKS];
    int plan[LOCAL_MAX_RANKS][4];
    int halo_off[LOCAL_MAX_RANKS][LOCAL_MAX_RANKS], halo_cnt[LOCAL_MAX_RANKS][LOCAL_MAX_RANKS];
};
struct LocalGroup {
    // members in other processes (msdp_comm_init_ipc): the host segment, the arena as this process maps it, this member's rank
    bool ipc = false;
    IpcShared* shm = nullptr;
    std::string shm_name;
    char* arena = nullptr;
    char* stage = nullptr;         // arena + slots + exchange buffer: n slabs of stage_bytes
    size_t stage_bytes = 0;
    int my_rank = 0;
    int ipc_ew = 0;                // the ELL width the members agreed on for the running call (0: CSR form)
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    bool broken = false;
    const double* ptr[LOCAL_MAX_RANKS] = {nullptr};
    const Halo* halo[LOCAL_MAX_RANKS] = {nullptr};
    int members = 0;
    int vote[LOCAL_MAX_RANKS] = {0};
    // cross-rank persistent tCG (msdp_persist.hip XR): what the members' launches share -- two slot regions of the grid
    // synchronisation, the error word (fine-grained device memory); the exchange buffers: xr_rows below
    unsigned long long* xr_slots = nullptr;
    int* xr_err = nullptr;
    // the combined launch (member 0 issues it for everybody): every member's Dev and plan, the events that order it behind the
    // members' streams and the members' streams behind it
    Dev xr_dev[LOCAL_MAX_RANKS];
    int xr_plan[3 * LOCAL_MAX_RANKS] = {0};
    hipEvent_t xr_ready[LOCAL_MAX_RANKS] = {nullptr};
    hipEvent_t xr_done = nullptr;
    // the members' exchange buffers of the cross-rank kernels: xr_rows[q] = member q's rows (in-process: all in this address space;
    // process group: this process's own allocation for q = my_rank, the IPC mappings of the others'), xr_rows_doubles each
    double* xr_rows[LOCAL_MAX_RANKS] = {nullptr};
    size_t xr_rows_doubles = 0;
    int xr_halo_max = 0;          // halo slots behind a member's rows in every buffer (the largest halo of the group)
    // two-level reductions (process group): the members' blocks as this process maps them (mine: my own allocation), the device copy of
    // that table
    unsigned long long* xr2_blk[LOCAL_MAX_RANKS] = {nullptr};
    unsigned long long** xr2_table = nullptr;
};
static std::mutex g_groups_mutex;
static std::map<int, LocalGroup*> g_groups;
// false: a member did not arrive within the time limit (it failed or never made the matching call) -- the group is broken and
// every later collective fails at once instead of hanging the process.  The limit is 120 s unless MSDP_LOCAL_BARRIER_TIMEOUT
// (seconds) says otherwise: eight replicated 60000-step verification runs sharing one loaded GPU are legitimately slow.
static double local_barrier_timeout() {
    static const double t = [] {
        const char* e = getenv("MSDP_LOCAL_BARRIER_TIMEOUT");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 120.0;
    }();
    return t;
}
// a member that fails between two barriers marks the group broken at once, so that its peers do not wait out the time limit
static void local_break(LocalGroup* g) {
    if (!g) return;
    if (g->ipc) { g->shm->broken.store(1); return; }
    std::lock_guard<std::mutex> lk(g->m);
    g->broken = true;
    g->cv.notify_all();
}
static bool local_barrier(LocalGroup* g) {
    if (g->ipc) {
        // sense-reversing barrier on the shared segment; polite polling (a collective on this path lasts tens of microseconds at least)
        IpcShared* sh = g->shm;
        if (sh->broken.load()) return false;
        const unsigned long long my = sh->gen.load();
        if (sh->arrived.fetch_add(1) + 1 == g->n) { sh->arrived.store(0); sh->gen.fetch_add(1); return true; }
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (sh->gen.load() == my) {
            if (sh->broken.load()) return false;
            if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > local_barrier_timeout()) { sh->broken.store(1); return false; }
        }
        return true;
    }
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const unsigned long long my = g->gen;
    if (++g->arrived == g->n) { g->arrived = 0; ++g->gen; g->cv.notify_all(); return true; }
    if (!g->cv.wait_for(lk, std::chrono::duration<double>(local_barrier_timeout()), [&] { return g->gen != my || g->broken; }) || g->broken) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}
#define LOCAL_BARRIER(g) do { if (!local_barrier(g)) { msdp_set_error("in-process communicator: a member did not reach the collective (group broken)"); return MSDP_ECOMM; } } while (0)
// minimum of one int per member (an agreement: every member takes the branch only if all of them can)
static int local_vote_min(msdp_handle h, int v, int* out) {
    LocalGroup* g = h->lgroup;
    if (g->ipc) g->shm->vote[h->rank] = v;
    else { std::lock_guard<std::mutex> lk(g->m); g->vote[h->rank] = v; }
    LOCAL_BARRIER(g);
    int m = v;
    for (int r = 0; r < g->n; ++r) m = std::min(m, g->ipc ? g->shm->vote[r] : g->vote[r]);
    LOCAL_BARRIER(g);                                      // nobody overwrites its vote before everyone has read it
    *out = m;
    return 0;
}
int msdp_xpersist_eligible(msdp_handle h, int nranks);                          // msdp_persist.hip
size_t msdp_xr2_block_bytes();
size_t msdp_xr2_err_offset();
int msdp_xr2_reset(hipStream_t stream, unsigned long long* blk);
size_t msdp_xpersist_slot_bytes();
int msdp_xpersist_reset(hipStream_t stream, unsigned long long* slots, int* err);
// The shared block of the group: allocated by member 0 the first time (and again when the factor outgrows the exchange buffer)
static int xr_ensure_shared(msdp_handle h) {
    LocalGroup* g = h->lgroup;
    size_t need = (g->ipc ? 4 : 1) * ((size_t)rows_capacity(h) + (size_t)(g->ipc ? g->xr_halo_max : h->xr_halo_rows)) * (size_t)std::max(h->ldcap, 64);   // per member: its rows + halo slots (process group: four regions)
    if (g->ipc) {                                            // the buffers were cut at msdp_comm_init_ipc
        if (g->xr_rows_doubles < need) { msdp_set_error("cross-rank persistent tCG: the factor outgrew the exchange buffers of this communicator (ld %d)", h->ldcap); return MSDP_ENOMEM; }
        return 0;
    }
    // the members agree on the LARGEST halo and the largest need (member 0's alone decided, a member with a wider
    // factor failed)
    {
        int m = 0;
        int rcv = local_vote_min(h, -h->xr_halo_rows, &m);
        if (rcv) return rcv;
        const int hmax = -m;
        need = ((size_t)rows_capacity(h) + (size_t)hmax) * (size_t)std::max(h->ldcap, 64);
        if ((rcv = local_vote_min(h, -(int)((need + 1023) / 1024), &m))) return rcv;
        need = (size_t)(-m) * 1024;
        if (h->rank == 0) g->xr_halo_max = hmax;
    }
    int rc = 0;
    if (h->rank == 0 && (!g->xr_slots || g->xr_rows_doubles < need)) {
        if (!g->xr_slots) {
            g->xr_slots = (unsigned long long*)msdp_uc_alloc(msdp_xpersist_slot_bytes() + 256);
            if (g->xr_slots) {
                g->xr_err = (int*)((char*)g->xr_slots + msdp_xpersist_slot_bytes());
                if (hipMemset(g->xr_err, 0, 256) != hipSuccess) rc = MSDP_EHIP;
                if (hipEventCreateWithFlags(&g->xr_done, hipEventDisableTiming) != hipSuccess) rc = MSDP_EHIP;
            }
        }
        g->xr_rows_doubles = 0;
        for (int q = 0; q < g->n; ++q) {
            if (g->xr_rows[q]) { if (!msdp_uc_free(g->xr_rows[q])) (void)hipFree(g->xr_rows[q]); g->xr_rows[q] = nullptr; }
            g->xr_rows[q] = (double*)msdp_uc_alloc(need * sizeof(double));
            if (!g->xr_rows[q]) { rc = MSDP_ENOMEM; break; }
            if (hipMemset(g->xr_rows[q], 0, need * sizeof(double)) != hipSuccess) rc = MSDP_EHIP;
        }
        if (!rc) g->xr_rows_doubles = need;
        if (!g->xr_slots) rc = MSDP_ENOMEM;
    }
    LOCAL_BARRIER(g);
    if (!g->xr_slots || g->xr_rows_doubles < need) { msdp_set_error("cross-rank persistent tCG: shared buffers unavailable"); return rc ? rc : MSDP_ENOMEM; }
    return 0;
}
int msdp_xpersist_member(msdp_handle h, int nranks, int rank, double* const* rows, int halo_rows, Dev* out, int* plan3);        // msdp_persist.hip
int msdp_launch_tcg_xpersist_all(hipStream_t stream, int nranks, const Dev* devs, const int* plans, unsigned long long* slots, int* err);
// Start of a trustregions() call on the cross-rank path: member 0 clears both slot regions and the error word; nobody goes on before
static int xr_begin(msdp_handle h, bool* use) {
    LocalGroup* g = h->lgroup;
    *use = false;
    int rc = xr_ensure_shared(h);
    if (rc) return rc;
    if (!h->xr_ev) HIPCHK(hipEventCreateWithFlags(&h->xr_ev, hipEventDisableTiming));
    *use = true;
    if (g->ipc) {
        // the plan of the call: lanes per row and row slots must agree, a differing ELL width sends everybody to the CSR form
        Dev dv; int pl[3];
        if ((rc = msdp_xpersist_member(h, h->nranks, h->rank, g->xr_rows, g->xr_halo_max, &dv, pl))) { local_break(g); return rc; }
        for (int q = 0; q < 3; ++q) g->shm->plan[h->rank][q] = pl[q];
        LOCAL_BARRIER(g);
        g->ipc_ew = pl[1];
        for (int r = 0; r < g->n; ++r) {
            if (g->shm->plan[r][0] != pl[0] || g->shm->plan[r][2] != pl[2]) { msdp_set_error("cross-rank persistent tCG: the members' plans differ"); local_break(g); return MSDP_ESTATE; }
            if (g->shm->plan[r][1] != pl[1]) g->ipc_ew = 0;
        }
    }
    if (h->rank == 0) {
        if ((rc = msdp_xpersist_reset(h->stream, g->xr_slots, g->xr_err))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (g->ipc && h->xr2_blk) {
        // two-level form: every member clears ITS block (slots, member lines, error word); nobody posts before everybody has
        if ((rc = msdp_xr2_reset(h->stream, h->xr2_blk))) { local_break(g); return rc; }
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    LOCAL_BARRIER(g);
    return 0;
}
static int* xr2_err(msdp_handle h) { return reinterpret_cast<int*>(reinterpret_cast<char*>(h->xr2_blk) + msdp_xr2_err_offset()); }
// One tCG for all members: each hands its Dev and plan to the group and marks its stream; member 0 makes its stream wait for
// the others', launches the combined kernel and marks its end; the others' streams wait for that mark.
int msdp_launch_tcg_xpersist_one(hipStream_t stream, const Dev& dv, const int* plan, unsigned long long* slots, int* err);   // msdp_persist.hip
static int xr_launch(msdp_handle h) {
    LocalGroup* g = h->lgroup;
    int rc;
    if (g->ipc) {
        // members in different processes: every member launches ITS workgroups itself (separate processes have separate hardware
        // queues; the launches meet in the first grid synchronisation, a bounded spin turns a member that never comes into MSDP_ECOMM).
        // The members agreed on the plan in xr_begin: no host exchange per launch.
        Dev dv; int pl[3];
        if ((rc = msdp_xpersist_member(h, h->nranks, h->rank, g->xr_rows, g->xr_halo_max, &dv, pl))) { local_break(g); return rc; }
        pl[1] = g->ipc_ew;
        if (h->tune.fail_xr) { h->tune.fail_xr = 0; dv.xr_gtot += 8; dv.xr2_skip = 8; }
        if ((rc = msdp_launch_tcg_xpersist_one(h->stream, dv, pl, g->xr_slots, dv.xr2_on ? xr2_err(h) : g->xr_err))) { local_break(g); return rc; }
        return 0;
    }
    {
        Dev dv; int pl[3];
        if ((rc = msdp_xpersist_member(h, h->nranks, h->rank, g->xr_rows, g->xr_halo_max, &dv, pl))) { local_break(g); return rc; }
        HIPCHK(hipEventRecord(h->xr_ev, h->stream));
        std::lock_guard<std::mutex> lk(g->m);
        g->xr_dev[h->rank] = dv;
        for (int q = 0; q < 3; ++q) g->xr_plan[3 * h->rank + q] = pl[q];
        g->xr_ready[h->rank] = h->xr_ev;
    }
    LOCAL_BARRIER(g);
    if (h->rank == 0) {
        for (int q = 1; q < g->n; ++q) HIPCHK(hipStreamWaitEvent(h->stream, g->xr_ready[q], 0));
        if (h->tune.fail_xr) {                                 // test hook: the workgroups wait for eight more than exist -> bounded spin -> error word
            h->tune.fail_xr = 0;
            for (int q = 0; q < g->n; ++q) g->xr_dev[q].xr_gtot += 8;
        }
        if ((rc = msdp_launch_tcg_xpersist_all(h->stream, g->n, g->xr_dev, g->xr_plan, g->xr_slots, g->xr_err))) { local_break(g); return rc; }
        HIPCHK(hipEventRecord(g->xr_done, h->stream));
    }
    LOCAL_BARRIER(g);
    if (h->rank != 0) HIPCHK(hipStreamWaitEvent(h->stream, g->xr_done, 0));
    return 0;
}
int msdp_launch_tr_tail_xr(hipStream_t stream, const Dev& dv, unsigned long long* slots, int* err);     // msdp_trtail.hip
static int xr_tail(msdp_handle h) {
    LocalGroup* g = h->lgroup;
    Dev dv; int pl[3];
    int rc = msdp_xpersist_member(h, h->nranks, h->rank, g->xr_rows, g->xr_halo_max, &dv, pl);
    if (!rc) rc = msdp_launch_tr_tail_xr(h->stream, dv, g->xr_slots, dv.xr2_on ? xr2_err(h) : g->xr_err);
    if (rc) local_break(g);
    return rc;
}
static int xr_check(msdp_handle h) {
    int e = 0;
    HIPCHK(msdp_memcpy(&e, h->lgroup->xr_err, sizeof(int), hipMemcpyDeviceToHost));
    if (!e && h->xr2_blk) HIPCHK(msdp_memcpy(&e, xr2_err(h), sizeof(int), hipMemcpyDeviceToHost));    // (the two-level form's word lives in the member's own block)
    if (e) {
        msdp_set_error("cross-rank persistent tCG: a grid synchronisation timed out (a member's launch did not arrive or the workgroups were not co-resident)");
        local_break(h->lgroup);                              // the other members' host-side collectives fail at once instead of waiting for this one
        return MSDP_ECOMM;
    }
    return 0;
}
struct LocalPtrs { const double* p[LOCAL_MAX_RANKS]; };
__global__ void k_local_sum(LocalPtrs src, int n, size_t count, double* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int r = 0; r < n; ++r) acc += src.p[r][i];
        out[i] = acc;
    }
}
// A member's contribution to a collective: in one process the others read it where it lies; across processes it is copied into the
// member's staging slab of the arena first (the caller has synchronised its stream: the source is complete)
static int group_publish(msdp_handle h, const double* buf, size_t count) {
    LocalGroup* g = h->lgroup;
    if (!g->ipc) { std::lock_guard<std::mutex> lk(g->m); g->ptr[h->rank] = buf; return 0; }
    if (count * sizeof(double) > g->stage_bytes) { msdp_set_error("inter-process communicator: a contribution of %zu bytes exceeds the staging slab (%zu)", count * sizeof(double), g->stage_bytes); local_break(g); return MSDP_ENOMEM; }
    HIPCHK(msdp_memcpy_async(g->stage + (size_t)h->rank * g->stage_bytes, buf, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
static const double* group_peer(LocalGroup* g, int r) { return g->ipc ? (const double*)(g->stage + (size_t)r * g->stage_bytes) : g->ptr[r]; }
static int local_allreduce(msdp_handle h, double* buf, size_t count) {
    LocalGroup* g = h->lgroup;
    if (h->lc_tmp_cap < count) {
        if (h->lc_tmp) (void)hipFree(h->lc_tmp);
        h->lc_tmp = nullptr; h->lc_tmp_cap = 0;
        if (hipMalloc((void**)&h->lc_tmp, count * sizeof(double)) != hipSuccess) { msdp_set_error("local all-reduce: scratch allocation failed"); return MSDP_ENOMEM; }
        h->lc_tmp_cap = count;
    }
    HIPCHK(hipStreamSynchronize(h->stream));               // my contribution is complete
    { int rcp = group_publish(h, buf, count); if (rcp) return rcp; }
    LOCAL_BARRIER(g);
    LocalPtrs src;
    for (int r = 0; r < LOCAL_MAX_RANKS; ++r) src.p[r] = r < g->n ? group_peer(g, r) : nullptr;
    int blocks = (int)std::min<size_t>(1024, (count + 255) / 256);
    hipLaunchKernelGGL(k_local_sum, dim3(blocks), dim3(256), 0, h->stream, src, g->n, count, h->lc_tmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    LOCAL_BARRIER(g);                                      // every member has read every contribution
    HIPCHK(msdp_memcpy_async(buf, h->lc_tmp, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}
static int local_allgather(msdp_handle h, const double* local, double* all, size_t count_per_rank) {
    LocalGroup* g = h->lgroup;
    HIPCHK(hipStreamSynchronize(h->stream));
    { int rcp = group_publish(h, local, count_per_rank); if (rcp) return rcp; }
    LOCAL_BARRIER(g);
    for (int r = 0; r < g->n; ++r)
        HIPCHK(msdp_memcpy_async(all + (size_t)r * count_per_rank, group_peer(g, r), count_per_rank * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    LOCAL_BARRIER(g);                                      // nobody overwrites its slab before everyone has copied it
    return 0;
}
// halo rows between in-process members: every member publishes its packed send buffer and its per-peer offsets
static int local_halo(msdp_handle h, Halo* ha, int ld) {
    LocalGroup* g = h->lgroup;
    HIPCHK(hipStreamSynchronize(h->stream));               // my send buffer is packed
    if (g->ipc) {
        for (int q = 0; q < g->n; ++q) { g->shm->halo_cnt[h->rank][q] = ha->send_cnt[q]; g->shm->halo_off[h->rank][q] = ha->send_off[q]; }
        int rcp = group_publish(h, ha->sendbuf, (size_t)ha->send_rows * ld); if (rcp) return rcp;
    } else { std::lock_guard<std::mutex> lk(g->m); g->ptr[h->rank] = ha->sendbuf; g->halo[h->rank] = ha; }
    LOCAL_BARRIER(g);
    for (int q = 0; q < g->n; ++q) {
        if (q == h->rank || ha->recv_cnt[q] == 0) continue;
        const int q_cnt = g->ipc ? g->shm->halo_cnt[q][h->rank] : g->halo[q]->send_cnt[h->rank];
        const int q_off = g->ipc ? g->shm->halo_off[q][h->rank] : g->halo[q]->send_off[h->rank];
        if (q_cnt != ha->recv_cnt[q]) {
            msdp_set_error("halo exchange: rank %d sends %d rows, rank %d expects %d", q, q_cnt, h->rank, ha->recv_cnt[q]);
            local_break(g);                                    // the peers learn at once, not after the barrier's time limit
            return MSDP_ECOMM;
        }
        HIPCHK(msdp_memcpy_async(ha->recvbuf + (size_t)ha->recv_off[q] * ld, group_peer(g, q) + (size_t)q_off * ld,
                              (size_t)ha->recv_cnt[q] * ld * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    LOCAL_BARRIER(g);
    return 0;
}
static void local_leave(msdp_handle h) {
    if (!h->lgroup) return;
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    LocalGroup* g = h->lgroup;
    h->lgroup = nullptr;
    if (g->ipc) {
        // the arena belongs to rank 0 (the mappings of the others keep its memory alive until they close them)
        if (g->arena) { if (g->my_rank == 0) (void)hipFree(g->arena); else (void)hipIpcCloseMemHandle(g->arena); }
        for (int q = 0; q < g->n; ++q) if (g->xr_rows[q]) { if (q == g->my_rank) (void)hipFree(g->xr_rows[q]); else (void)hipIpcCloseMemHandle(g->xr_rows[q]); }
        for (int q = 0; q < g->n; ++q) if (g->xr2_blk[q]) { if (q == g->my_rank) (void)hipFree(g->xr2_blk[q]); else (void)hipIpcCloseMemHandle(g->xr2_blk[q]); }
        if (g->xr2_table) (void)hipFree(g->xr2_table);
        h->xr2_blk = nullptr; h->xr2_peers = nullptr; h->lgroup_is_ipc = false;
        if (g->shm) (void)munmap((void*)g->shm, sizeof(IpcShared));
        if (g->my_rank == 0 && !g->shm_name.empty()) (void)shm_unlink(g->shm_name.c_str());
        delete g;
        return;
    }
    if (--g->members == 0) {
        for (auto it = g_groups.begin(); it != g_groups.end(); ++it) if (it->second == g) { g_groups.erase(it); break; }
        if (g->xr_slots && !msdp_uc_free(g->xr_slots)) (void)hipFree(g->xr_slots);
        for (int q = 0; q < LOCAL_MAX_RANKS; ++q) if (g->xr_rows[q] && !msdp_uc_free(g->xr_rows[q])) (void)hipFree(g->xr_rows[q]);
        if (g->xr_done) (void)hipEventDestroy(g->xr_done);
        delete g;
    }
}

int msdp_allgather_rows(msdp_handle h, const double* local_rows);
// ------------------------------------------------------------------ Halo exchange (sparse C)
// The all-gather of the n x p direction moves (N-1)/N * n*p*8 bytes into every rank before every S*U -- for the G81 family
// on eight ranks 35 MB per trip where the rows of C a rank owns reference 400 rows of other ranks (100 KB).  With the option
// "halo_exchange" a rank packs, for every peer, the rows that peer's rows of C reference, one grouped ncclSend / ncclRecv
// per peer moves them, and an unpack kernel scatters the received rows to their global positions in the gather buffer, which
// the S*U kernels read exactly as after an all-gather.  Every rank holds the whole CSR structure on the host, so all send /
// receive lists are computed locally and agree by construction.  Only for sparse C and only for the two exchanges in front
// of the cost/gradient and Hess-vec kernels; msdp_get_point_all and the replicated escape keep the all-gather (they need
// every row).
__global__ void k_halo_pack(int rows, int ld, const int* __restrict__ idx, const double* __restrict__ local, double* __restrict__ buf) {
    const int64_t tot = (int64_t)rows * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / ld; const int c = (int)(e - k * ld);
        buf[e] = local[(int64_t)idx[k] * ld + c];
    }
}
__global__ void k_halo_unpack(int rows, int ld, const int* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ full) {
    const int64_t tot = (int64_t)rows * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / ld; const int c = (int)(e - k * ld);
        full[(int64_t)idx[k] * ld + c] = buf[e];
    }
}
static void halo_release(msdp_handle h) {
    Halo* ha = h->halo;
    if (!ha) return;
    if (ha->send_idx) (void)hipFree(ha->send_idx);
    if (ha->recv_idx) (void)hipFree(ha->recv_idx);
    if (ha->sendbuf) (void)hipFree(ha->sendbuf);
    if (ha->recvbuf) (void)hipFree(ha->recvbuf);
    if (ha->xr_colind) (void)hipFree(ha->xr_colind);
    if (ha->xr_ellc) (void)hipFree(ha->xr_ellc);
    if (ha->xr_pq) (void)hipFree(ha->xr_pq);
    if (ha->xr_pidx) (void)hipFree(ha->xr_pidx);
    delete ha;
    h->halo = nullptr;
    h->d.xr_colind = h->d.xr_ellc = h->d.xr_pq = h->d.xr_pidx = nullptr; h->xr_ok = false; h->xr_halo_rows = 0;
    h->xr_paddr_pq = nullptr; h->d.xr_paddr = nullptr;   // (the push addresses are rebuilt from the next partition's lists)
}
// Lists for the current partition (called by comm_partition for sparse C); buffers follow the vectors' capacity
static int halo_setup(msdp_handle h) {
    halo_release(h);
    if (h->d.costkind != COST_SPARSE || h->h_rowptr.empty() || h->nranks < 2) return 0;
    const int N = h->nranks, n = h->d.n, cap = rows_capacity(h), me = h->rank;
    Halo* ha = new Halo();
    ha->N = N;
    ha->send_cnt.assign(N, 0); ha->send_off.assign(N + 1, 0); ha->recv_cnt.assign(N, 0); ha->recv_off.assign(N + 1, 0);
    // needs[q]: rows outside q's range that q's rows of C reference (sorted, unique)
    std::vector<std::vector<int>> send_rows(N);          // what I send to q (local indices), in the order q will unpack
    std::vector<int> recv_rows;
    std::vector<char> mark((size_t)n, 0);
    const int my0 = std::min(n, me * cap), my1 = std::min(n, my0 + cap), nloc = my1 - my0;
    std::vector<int> pq((size_t)2 * std::max(nloc, 1), -1), pidx((size_t)2 * std::max(nloc, 1), 0);
    bool push_ok = true;
    for (int q = 0; q < N; ++q) {
        const int q0 = std::min(n, q * cap), q1 = std::min(n, q0 + cap);
        std::vector<int> need;
        for (int i = q0; i < q1; ++i)
            for (int t = h->h_rowptr[i]; t < h->h_rowptr[i + 1]; ++t) {
                const int c = h->h_colind[t];
                if ((c < q0 || c >= q1) && !mark[c]) { mark[c] = 1; need.push_back(c); }
            }
        std::sort(need.begin(), need.end());
        for (int c : need) mark[c] = 0;
        if (q == me) {
            recv_rows = need;                            // sorted by global row = grouped by owning peer
            for (int c : need) ha->recv_cnt[c / cap]++;
        } else {
            const int m0 = std::min(n, me * cap), m1 = std::min(n, m0 + cap);
            for (int c : need) if (c >= m0 && c < m1) send_rows[q].push_back(c - m0);
            ha->send_cnt[q] = (int)send_rows[q].size();
            // push exchange: my row c sits at position cap + (index of c in q's sorted need list) of q's buffer
            for (size_t k = 0; k < need.size(); ++k) {
                const int c = need[k];
                if (c < m0 || c >= m1) continue;
                const int i = c - m0;
                if (pq[i] < 0) { pq[i] = q; pidx[i] = cap + (int)k; }
                else if (pq[(size_t)nloc + i] < 0) { pq[(size_t)nloc + i] = q; pidx[(size_t)nloc + i] = cap + (int)k; }
                else push_ok = false;
            }
        }
    }
    std::vector<int> sidx;
    for (int q = 0; q < N; ++q) { ha->send_off[q + 1] = ha->send_off[q] + ha->send_cnt[q]; ha->recv_off[q + 1] = ha->recv_off[q] + ha->recv_cnt[q]; sidx.insert(sidx.end(), send_rows[q].begin(), send_rows[q].end()); }
    ha->send_rows = ha->send_off[N]; ha->recv_rows = ha->recv_off[N];
    ha->ldcap = h->ldcap > 0 ? h->ldcap : ((h->pcap + 1) / 2) * 2;
    auto upi = [&](const std::vector<int>& v, int** out) -> int {
        if (hipMalloc((void**)out, (v.size() ? v.size() : 1) * sizeof(int)) != hipSuccess) return MSDP_ENOMEM;
        if (!v.empty() && msdp_memcpy(*out, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return MSDP_EHIP;
        return 0;
    };
    int rc = upi(sidx, &ha->send_idx);
    if (!rc) rc = upi(recv_rows, &ha->recv_idx);
    // push exchange: buffer-local column indices of my rows (CSR and ELL copies), push targets
    if (!rc) {
        std::vector<int> gmap((size_t)n, -1);
        for (int c = my0; c < my1; ++c) gmap[c] = c - my0;
        for (size_t k = 0; k < recv_rows.size(); ++k) gmap[recv_rows[k]] = cap + (int)k;
        const int base = nloc > 0 ? h->h_rowptr[my0] : 0, nnzl = nloc > 0 ? h->h_rowptr[my1] - base : 0;
        std::vector<int> xcol((size_t)std::max(nnzl, 1), 0);
        for (int t = 0; t < nnzl; ++t) xcol[t] = gmap[h->h_colind[base + t]];
        rc = upi(xcol, &ha->xr_colind);
        if (!rc && h->d.ellW > 0) {
            const int W = h->d.ellW;
            std::vector<int> ec((size_t)W * cap);
            for (int w = 0; w < W; ++w) for (int i = 0; i < cap; ++i) ec[(size_t)w * cap + i] = std::min(i, nloc > 0 ? nloc - 1 : 0);    // padding: (own row, 0.0)
            for (int i = 0; i < nloc; ++i)
                for (int t = h->h_rowptr[my0 + i]; t < h->h_rowptr[my0 + i + 1] && t - h->h_rowptr[my0 + i] < W; ++t)
                    ec[(size_t)(t - h->h_rowptr[my0 + i]) * cap + i] = gmap[h->h_colind[t]];
            rc = upi(ec, &ha->xr_ellc);
        }
        if (!rc) rc = upi(pq, &ha->xr_pq);
        if (!rc) rc = upi(pidx, &ha->xr_pidx);
        ha->xr_ok = push_ok && !rc;
        h->d.xr_colind = ha->xr_colind; h->d.xr_ellc = ha->xr_ellc; h->d.xr_pq = ha->xr_pq; h->d.xr_pidx = ha->xr_pidx;
        h->xr_ok = ha->xr_ok; h->xr_halo_rows = (int)recv_rows.size();
    }
    if (!rc && hipMalloc((void**)&ha->sendbuf, (size_t)std::max(ha->send_rows, 1) * ha->ldcap * sizeof(double)) != hipSuccess) rc = MSDP_ENOMEM;
    if (!rc && hipMalloc((void**)&ha->recvbuf, (size_t)std::max(ha->recv_rows, 1) * ha->ldcap * sizeof(double)) != hipSuccess) rc = MSDP_ENOMEM;
    h->halo = ha;
    if (rc) { msdp_set_error("halo exchange: set-up allocation failed"); halo_release(h); }
    return rc;
}
static int local_halo(msdp_handle h, Halo* ha, int ld);      // in-process stand-in, below the LocalGroup definition
// rows of `local` the other ranks reference -> their gather buffers; mine + what I reference -> my gather buffer
static int halo_exchange(msdp_handle h, const double* local_rows, bool with_sums = false) {
    Halo* ha = h->halo;
    const int ld = h->d.ld;
    ++h->coll_calls;
    if (ld > ha->ldcap) {                                  // the vectors were re-allocated for a wider factor: follow
        const int cap = h->ldcap;
        if (ha->sendbuf) (void)hipFree(ha->sendbuf);
        if (ha->recvbuf) (void)hipFree(ha->recvbuf);
        ha->sendbuf = ha->recvbuf = nullptr;
        if (hipMalloc((void**)&ha->sendbuf, (size_t)std::max(ha->send_rows, 1) * cap * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&ha->recvbuf, (size_t)std::max(ha->recv_rows, 1) * cap * sizeof(double)) != hipSuccess) { msdp_set_error("halo exchange: buffer allocation failed"); return MSDP_ENOMEM; }
        ha->ldcap = cap;
    }
    h->d.full = h->full_buf;
    const size_t own = (size_t)rows_capacity(h) * ld;
    HIPCHK(msdp_memcpy_async(h->full_buf + (size_t)h->rank * own, local_rows, own * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (ha->send_rows > 0) {
        const int64_t tot = (int64_t)ha->send_rows * ld;
        hipLaunchKernelGGL(k_halo_pack, dim3((int)std::min<int64_t>(1024, (tot + 255) / 256)), dim3(256), 0, h->stream, ha->send_rows, ld,
                           (const int*)ha->send_idx, local_rows, ha->sendbuf);
        HIPCHK(hipGetLastError());
    }
    if (h->lgroup) { int rc = local_halo(h, ha, ld); if (rc) return rc; }
    else {
        // with_sums (msdp_trip1.hip): every pair of ranks also swaps its four sums in the same group
        if (with_sums) HIPCHK(msdp_memcpy_async(h->d.xs_all + 4 * (size_t)h->rank, h->d.xs, 4 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        ncclResult_t r = ncclGroupStart();
        for (int q = 0; q < ha->N && r == ncclSuccess; ++q) {
            if (q == h->rank) continue;
            if (ha->send_cnt[q] > 0) r = ncclSend(ha->sendbuf + (size_t)ha->send_off[q] * ld, (size_t)ha->send_cnt[q] * ld, ncclDouble, q, (ncclComm_t)h->comm, h->stream);
            if (r == ncclSuccess && ha->recv_cnt[q] > 0) r = ncclRecv(ha->recvbuf + (size_t)ha->recv_off[q] * ld, (size_t)ha->recv_cnt[q] * ld, ncclDouble, q, (ncclComm_t)h->comm, h->stream);
            if (with_sums && r == ncclSuccess) r = ncclSend(h->d.xs, 4, ncclDouble, q, (ncclComm_t)h->comm, h->stream);
            if (with_sums && r == ncclSuccess) r = ncclRecv(h->d.xs_all + 4 * (size_t)q, 4, ncclDouble, q, (ncclComm_t)h->comm, h->stream);
        }
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) { msdp_set_error("halo exchange: ncclSend/ncclRecv failed: %s", ncclGetErrorString(r != ncclSuccess ? r : r2)); return MSDP_ECOMM; }
    }
    if (ha->recv_rows > 0) {
        const int64_t tot = (int64_t)ha->recv_rows * ld;
        hipLaunchKernelGGL(k_halo_unpack, dim3((int)std::min<int64_t>(1024, (tot + 255) / 256)), dim3(256), 0, h->stream, ha->recv_rows, ld,
                           (const int*)ha->recv_idx, (const double*)ha->recvbuf, h->full_buf);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
extern "C" int msdp_debug_p2p_self(msdp_handle h, int64_t count, const double* in_host, double* out_host) {
    if (!h || !h->use_comm || h->lgroup || !h->comm || count <= 0) { msdp_set_error("debug_p2p_self: needs an RCCL communicator"); return MSDP_ESTATE; }
    double *a = nullptr, *b = nullptr;
    if (hipMalloc((void**)&a, count * sizeof(double)) != hipSuccess || hipMalloc((void**)&b, count * sizeof(double)) != hipSuccess) {
        if (a) (void)hipFree(a);
        msdp_set_error("debug_p2p_self: allocation failed"); return MSDP_ENOMEM;
    }
    int rc = 0;
    if (msdp_memcpy_async(a, in_host, count * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = MSDP_EHIP;
    if (!rc) {
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclSend(a, (size_t)count, ncclDouble, h->rank, (ncclComm_t)h->comm, h->stream);
        if (r == ncclSuccess) r = ncclRecv(b, (size_t)count, ncclDouble, h->rank, (ncclComm_t)h->comm, h->stream);
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) { msdp_set_error("debug_p2p_self: %s", ncclGetErrorString(r != ncclSuccess ? r : r2)); rc = MSDP_ECOMM; }
    }
    if (!rc && msdp_memcpy_async(out_host, b, count * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = MSDP_EHIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess && !rc) rc = MSDP_EHIP;
    (void)hipFree(a); (void)hipFree(b);
    return rc;
}

int msdp_exchange_rows(msdp_handle h, const double* local_rows) {
    if (h->use_comm && h->halo && h->tune.halo_exchange && h->nranks > 1) return halo_exchange(h, local_rows);
    return msdp_allgather_rows(h, local_rows);
}

// The exchange of msdp_trip1.hip: the rows as above AND d.xs (4 doubles) of every rank into d.xs_all, in ONE grouped
// collective call -- RCCL fuses the operations between ncclGroupStart / ncclGroupEnd into one launch.
int msdp_exchange_rows_sums(msdp_handle h, const double* local_rows) {
    Dev& d = h->d;
    if (!h->use_comm) {                                   // one rank, no communicator: the kernels read the rows and the sums in place
        if (h->nranks != 1) { msdp_set_error("exchange_rows_sums: no communicator"); return MSDP_ESTATE; }
        d.full = const_cast<double*>(local_rows);
        d.xs_all = d.xs;
        return 0;
    }
    const bool halo = h->halo && h->tune.halo_exchange && h->nranks > 1;
    if (h->lgroup) {
        // in-process stand-in: the two parts one after the other, counted as the one call they are under RCCL
        int rc = halo ? halo_exchange(h, local_rows) : msdp_allgather_rows(h, local_rows);      // (counted there)
        if (!rc) rc = local_allgather(h, d.xs, d.xs_all, 4);
        return rc;
    }
    if (halo) return halo_exchange(h, local_rows, true);
    ++h->coll_calls;
    const size_t cnt = (size_t)rows_capacity(h) * d.ld;
    d.full = h->full_buf;
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclAllGather(local_rows, h->full_buf, cnt, ncclDouble, (ncclComm_t)h->comm, h->stream);
    if (r == ncclSuccess) r = ncclAllGather(d.xs, d.xs_all, 4, ncclDouble, (ncclComm_t)h->comm, h->stream);
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) { msdp_set_error("grouped ncclAllGather failed: %s", ncclGetErrorString(r != ncclSuccess ? r : r2)); return MSDP_ECOMM; }
    return 0;
}

// ------------------------------------------------------------------ collectives
int msdp_allreduce_partials(msdp_handle h, int first, int count) {
    if (!h->use_comm) return 0;
    ++h->coll_calls;
    if (h->lgroup) return local_allreduce(h, h->d.P + (size_t)first * MSDP_MAX_GRID, (size_t)count * MSDP_MAX_GRID);
    double* buf = h->d.P + (size_t)first * MSDP_MAX_GRID;
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count * MSDP_MAX_GRID, ncclDouble, ncclSum,
                                   (ncclComm_t)h->comm, h->stream);
    if (r != ncclSuccess) { msdp_set_error("ncclAllReduce failed: %s", ncclGetErrorString(r)); return MSDP_ECOMM; }
    return 0;
}

// Make all rows of a row-sharded vector visible to the gather kernels: one RCCL
// all-gather of the thin n x p factor (each rank sends its slab to its 7 peers,
// one message per xGMI link).  With one rank the local buffer is used directly.
int msdp_allgather_rows(msdp_handle h, const double* local_rows) {
    if (!h->use_comm) {
        if (h->nranks > 1) {        // a lone process standing in for one shard (tests / per-shard measurement)
            const size_t cnt1 = (size_t)rows_capacity(h) * h->d.ld;
            HIPCHK(msdp_memcpy_async(h->full_buf + (size_t)h->rank * cnt1, local_rows, cnt1 * sizeof(double),
                                  hipMemcpyDeviceToDevice, h->stream));
            h->d.full = h->full_buf;
            return 0;
        }
        h->d.full = const_cast<double*>(local_rows);
        return 0;
    }
    const size_t cnt = (size_t)rows_capacity(h) * h->d.ld;
    // slabs are packed with the CURRENT ld so the full buffer is n_pad x ld row-major
    h->d.full = h->full_buf;
    ++h->coll_calls;
    if (h->lgroup) return local_allgather(h, local_rows, h->full_buf, cnt);
    ncclResult_t r = ncclAllGather(local_rows, h->full_buf, cnt, ncclDouble, (ncclComm_t)h->comm, h->stream);
    if (r != ncclSuccess) { msdp_set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return MSDP_ECOMM; }
    return 0;
}

int msdp_allreduce_array(msdp_handle h, double* buf, size_t count) {
    if (!h->lgroup) { msdp_set_error("allreduce_array: local group only"); return MSDP_ESTATE; }
    return local_allreduce(h, buf, count);
}

// count_per_rank doubles from every rank, in rank order
int msdp_allgather_vec(msdp_handle h, const double* local, double* all, size_t count_per_rank) {
    if (!h->use_comm) { msdp_set_error("allgather_vec: no communicator"); return MSDP_ESTATE; }
    ++h->coll_calls;
    if (h->lgroup) return local_allgather(h, local, all, count_per_rank);
    ncclResult_t r = ncclAllGather(local, all, count_per_rank, ncclDouble, (ncclComm_t)h->comm, h->stream);
    if (r != ncclSuccess) { msdp_set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return MSDP_ECOMM; }
    return 0;
}

extern "C" int msdp_comm_unique_id(void* id128) {
    if (!id128) return MSDP_EINVAL;
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { msdp_set_error("ncclGetUniqueId failed: %s", ncclGetErrorString(r)); return MSDP_ECOMM; }
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    memcpy(id128, &id, 128);
    return 0;
}

static int comm_partition(msdp_handle h, int32_t nranks, int32_t rank);

// Test / diagnostic: member `rank` of the in-process group `group_id` of `nranks` handles (one process, one GPU, one host
// thread per handle).  Same partition, same code paths as msdp_comm_init; the collectives are the local stand-ins above.
extern "C" int msdp_comm_init_local(msdp_handle h, int32_t nranks, int32_t rank, int32_t group_id) {
    CHECK_H(h);
    if (nranks < 1 || nranks > LOCAL_MAX_RANKS || rank < 0 || rank >= nranks) { msdp_set_error("comm_init_local: bad arguments"); return MSDP_EINVAL; }
    if (h->have_point || h->use_comm) { msdp_set_error("comm_init_local must precede set_point / comm_init"); return MSDP_ESTATE; }
    if (h->presharded && (nranks != h->nranks || rank != h->rank)) { msdp_set_error("comm_init_local: shard was created as rank %d of %d", h->rank, h->nranks); return MSDP_EINVAL; }
    if (h->kind == MSDP_KIND_MULTIBLOCK || h->kind == MSDP_KIND_DUAL_UNITDIAG) { msdp_set_error("row sharding is not implemented for the multiblock and dual kinds"); return MSDP_EUNSUPPORTED; }
    {
        std::lock_guard<std::mutex> lk(g_groups_mutex);
        LocalGroup*& g = g_groups[group_id];
        if (!g) { g = new LocalGroup(); g->n = nranks; }
        if (g->n != nranks || g->members >= nranks) { msdp_set_error("comm_init_local: group %d has %d of %d members", group_id, g->members, g->n); return MSDP_EINVAL; }
        ++g->members;
        h->lgroup = g;
    }
    return comm_partition(h, nranks, rank);
}

// Members in different processes (one per GPU of a node, or several on one GPU): `name` identifies the group (a POSIX shared-memory
// name, e.g. "/msdp_<pid of the launcher>_<counter>"; every member passes the same one).  Rank 0 allocates the arena and exports it,
// the others map it; with ranks on different devices the mapping goes over peer access (hipIpcMemLazyEnablePeerAccess).
static int comm_init_ipc_attach(msdp_handle h, int32_t nranks, int32_t rank, unsigned long long my_ino);
extern "C" int msdp_comm_init_ipc(msdp_handle h, int32_t nranks, int32_t rank, const char* name) {
    CHECK_H(h);
    if (nranks < 1 || nranks > LOCAL_MAX_RANKS || rank < 0 || rank >= nranks || !name || name[0] != '/') { msdp_set_error("comm_init_ipc: bad arguments (the name starts with '/')"); return MSDP_EINVAL; }
    if (h->have_point || h->use_comm) { msdp_set_error("comm_init_ipc must precede set_point / comm_init"); return MSDP_ESTATE; }
    if (h->presharded && (nranks != h->nranks || rank != h->rank)) { msdp_set_error("comm_init_ipc: shard was created as rank %d of %d", h->rank, h->nranks); return MSDP_EINVAL; }
    if (h->kind == MSDP_KIND_MULTIBLOCK || h->kind == MSDP_KIND_DUAL_UNITDIAG) { msdp_set_error("row sharding is not implemented for the multiblock and dual kinds"); return MSDP_EUNSUPPORTED; }
    // The host segment: rank 0 removes whatever a crashed run left under this name, creates the segment anew (O_EXCL),
    // clears it and writes the magic word LAST; the others open it without O_CREAT, wait for the magic word and check that the name
    // still leads to the segment they mapped (a stale one that rank 0 has replaced meanwhile does not).
    void* mp = MAP_FAILED;
    unsigned long long my_ino = 0;
    if (rank == 0) {
        (void)shm_unlink(name);
        const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { msdp_set_error("comm_init_ipc: shm_open(%s, O_EXCL) failed", name); return MSDP_ECOMM; }
        if (ftruncate(fd, (off_t)sizeof(IpcShared)) != 0) { (void)close(fd); (void)shm_unlink(name); msdp_set_error("comm_init_ipc: ftruncate failed"); return MSDP_ECOMM; }
        mp = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        (void)close(fd);
        if (mp == MAP_FAILED) { (void)shm_unlink(name); msdp_set_error("comm_init_ipc: mmap failed"); return MSDP_ECOMM; }
        memset(mp, 0, sizeof(IpcShared));
        ((IpcShared*)mp)->magic.store(IPC_MAGIC);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > local_barrier_timeout();
            const int fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcShared)) {
                void* q = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                (void)close(fd);
                if (q != MAP_FAILED) {
                    if (((IpcShared*)q)->magic.load() == IPC_MAGIC) {
                        // the name still leads here?  (rank 0 unlinks a stale segment before it creates the group's)
                        const int fd2 = shm_open(name, O_RDWR, 0600);
                        struct stat st2;
                        const bool same = fd2 >= 0 && fstat(fd2, &st2) == 0 && st2.st_ino == st.st_ino;
                        if (fd2 >= 0) (void)close(fd2);
                        if (same) { mp = q; my_ino = (unsigned long long)st.st_ino; break; }
                    }
                    (void)munmap(q, sizeof(IpcShared));
                }
            } else if (fd >= 0) (void)close(fd);
            if (late) { msdp_set_error("comm_init_ipc: rank 0 did not create the segment %s", name); return MSDP_ECOMM; }
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
    }
    LocalGroup* g = new LocalGroup();
    g->ipc = true; g->n = nranks; g->members = 1; g->my_rank = rank; g->shm = (IpcShared*)mp; g->shm_name = name;
    h->lgroup = g;
    int rc = comm_init_ipc_attach(h, nranks, rank, my_ino);
    if (rc) {
        // (a member that fails here tells the others, and leaves nothing behind: mapping, group record and -- rank 0 -- the name)
        local_break(g);
        local_leave(h);
    }
    return rc;
}
static int comm_init_ipc_attach(msdp_handle h, int32_t nranks, int32_t rank, unsigned long long my_ino) {
    LocalGroup* g = h->lgroup;
    const char* name = g->shm_name.c_str();
    int rc = comm_partition(h, nranks, rank);
    if (rc) return rc;
    IpcShared* sh = g->shm;
    const size_t slot_bytes = msdp_xpersist_slot_bytes() + 256;
    const size_t ldx = (size_t)std::max(h->ldcap, 64);
    // every member's own exchange buffer: its rows + a slot for every foreign row it references (the largest halo of the group: the
    // members vote below, through the segment)
    sh->vote[rank] = h->xr_halo_rows;
    const size_t stage = std::max<size_t>(((size_t)rows_capacity(h) * ldx * sizeof(double) + 255) / 256 * 256, (size_t)1 << 16);
    const size_t total = slot_bytes + (size_t)nranks * stage;
    if (rank == 0) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: arena allocation of %zu bytes failed", total); local_break(g); return MSDP_ENOMEM; }
        HIPCHK(hipMemset(p, 0, total));
        HIPCHK(hipDeviceSynchronize());
        hipIpcMemHandle_t hd;
        if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); msdp_set_error("comm_init_ipc: hipIpcGetMemHandle failed"); local_break(g); return MSDP_ECOMM; }
        g->arena = (char*)p;
        sh->arena = hd; sh->arena_bytes = total; sh->stage_bytes = stage; sh->slot_bytes = slot_bytes;
        sh->arena_ready.store(1);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!sh->arena_ready.load()) {
            if (sh->broken.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > local_barrier_timeout()) { msdp_set_error("comm_init_ipc: rank 0 did not publish the arena"); return MSDP_ECOMM; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (sh->arena_bytes != total || sh->stage_bytes != stage) { msdp_set_error("comm_init_ipc: the members disagree about the problem size"); local_break(g); return MSDP_EINVAL; }
        void* p = nullptr;
        hipIpcMemHandle_t hd = sh->arena;
        if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: hipIpcOpenMemHandle failed"); local_break(g); return MSDP_ECOMM; }
        g->arena = (char*)p;
    }
    g->xr_slots = (unsigned long long*)g->arena;
    g->xr_err = (int*)(g->arena + msdp_xpersist_slot_bytes());
    g->stage = g->arena + slot_bytes;
    g->stage_bytes = stage;
    sh->attached.fetch_add(1);
    LOCAL_BARRIER(g);                                        // everybody has mapped the arena; the halo sizes are in the segment
    if (rank != 0) {
        // (once more behind the first barrier: a segment of an earlier run that passed every wait above on stale values)
        const int fd2 = shm_open(name, O_RDWR, 0600);
        struct stat st2;
        const bool same = fd2 >= 0 && fstat(fd2, &st2) == 0 && (unsigned long long)st2.st_ino == my_ino;
        if (fd2 >= 0) (void)close(fd2);
        if (!same) { msdp_set_error("comm_init_ipc: attached to a stale segment %s", name); return MSDP_ECOMM; }
    }
    size_t hmax = 0;
    for (int q = 0; q < nranks; ++q) hmax = std::max<size_t>(hmax, (size_t)sh->vote[q]);
    // (four regions of rows + halo slots each: the one-reduction trip of msdp_pipe.h publishes H md alternately in two and its refresh
    // rows in two more; the two-reduction trip and the TR tail use the first)
    const size_t rows_doubles = 4 * ((size_t)rows_capacity(h) + hmax) * ldx;
    g->xr_halo_max = (int)hmax;
    // the exchange buffer of MY rows (+ my halo slots): my own allocation (on my device), exported; then the others', mapped
    {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, rows_doubles * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: exchange buffer allocation failed"); local_break(g); return MSDP_ENOMEM; }
        HIPCHK(hipMemset(p, 0, rows_doubles * sizeof(double)));
        HIPCHK(hipDeviceSynchronize());
        hipIpcMemHandle_t hd;
        if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); msdp_set_error("comm_init_ipc: hipIpcGetMemHandle (exchange buffer) failed"); local_break(g); return MSDP_ECOMM; }
        g->xr_rows[rank] = (double*)p;
        sh->rows_handle[rank] = hd;
    }
    LOCAL_BARRIER(g);                                        // everybody has exported its buffer
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        void* p = nullptr;
        hipIpcMemHandle_t hd = sh->rows_handle[q];
        if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: hipIpcOpenMemHandle (exchange buffer of rank %d) failed", q); local_break(g); return MSDP_ECOMM; }
        g->xr_rows[q] = (double*)p;
    }
    g->xr_rows_doubles = rows_doubles;
    // Round 6, the two-level reductions of msdp_psync.h (psync2): every member's own block -- its local slot regions, the member lines the
    // others' leaders push into, its error word -- fine-grained memory of ITS device, exported like the exchange buffer; and where the
    // members sit: the PCI bus ids tell how many share a device (their workgroups must be resident together) and whether any two own
    // different ones (then the reductions are two-level and the pushed rows cross devices)
    {
        void* p = nullptr;
        const size_t bb = msdp_xr2_block_bytes();
        if (hipExtMallocWithFlags(&p, bb, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: synchronisation block allocation failed"); return MSDP_ENOMEM; }
        g->xr2_blk[rank] = (unsigned long long*)p;
        int rc2 = msdp_xr2_reset(h->stream, g->xr2_blk[rank]);
        if (rc2) return rc2;
        HIPCHK(hipStreamSynchronize(h->stream));
        hipIpcMemHandle_t hd;
        if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: hipIpcGetMemHandle (synchronisation block) failed"); return MSDP_ECOMM; }
        sh->blk_handle[rank] = hd;
        int dev = 0;
        char bus[32] = {0};
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof(bus), "device-%d", dev); }
        memcpy(sh->devid[rank], bus, sizeof(bus));
    }
    LOCAL_BARRIER(g);                                        // every block is exported, every device id is in the segment
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        void* p = nullptr;
        hipIpcMemHandle_t hd = sh->blk_handle[q];
        if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: hipIpcOpenMemHandle (synchronisation block of rank %d) failed", q); return MSDP_ECOMM; }
        g->xr2_blk[q] = (unsigned long long*)p;
    }
    {
        if (hipMalloc((void**)&g->xr2_table, LOCAL_MAX_RANKS * sizeof(unsigned long long*)) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("comm_init_ipc: out of device memory"); return MSDP_ENOMEM; }
        unsigned long long* tab[LOCAL_MAX_RANKS];
        for (int q = 0; q < LOCAL_MAX_RANKS; ++q) tab[q] = g->xr2_blk[q < nranks ? q : rank];
        HIPCHK(msdp_memcpy(g->xr2_table, tab, sizeof(tab), hipMemcpyHostToDevice));
        int share = 1;
        bool multi = false;
        for (int q = 0; q < nranks; ++q) {
            int cnt = 0;
            for (int t = 0; t < nranks; ++t) cnt += strncmp(sh->devid[q], sh->devid[t], sizeof(sh->devid[q])) == 0 ? 1 : 0;
            share = std::max(share, cnt);
            if (strncmp(sh->devid[q], sh->devid[rank], sizeof(sh->devid[q])) != 0) multi = true;
        }
        // (`multi` as every member sees it: true on all of them as soon as two devices are involved)
        bool any_multi = false;
        for (int q = 0; q < nranks; ++q) for (int t = 0; t < nranks; ++t) if (strncmp(sh->devid[q], sh->devid[t], sizeof(sh->devid[q])) != 0) any_multi = true;
        (void)multi;
        h->xr2_blk = g->xr2_blk[rank]; h->xr2_peers = g->xr2_table; h->xr2_share = share; h->xr2_multi = any_multi; h->lgroup_is_ipc = true;
    }
    LOCAL_BARRIER(g);
    return 0;
}

extern "C" int msdp_comm_init(msdp_handle h, int32_t nranks, int32_t rank, const void* id128) {
    CHECK_H(h);
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128) { msdp_set_error("bad comm arguments"); return MSDP_EINVAL; }
    if (h->have_point) { msdp_set_error("comm_init must precede set_point"); return MSDP_ESTATE; }
    if (h->presharded && (nranks != h->nranks || rank != h->rank)) { msdp_set_error("comm_init: shard was created as rank %d of %d", h->rank, h->nranks); return MSDP_EINVAL; }
    if (h->kind == MSDP_KIND_MULTIBLOCK || h->kind == MSDP_KIND_DUAL_UNITDIAG) {
        msdp_set_error("row sharding is not implemented for the multiblock and dual kinds");
        return MSDP_EUNSUPPORTED;
    }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) { msdp_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r)); return MSDP_ECOMM; }
    h->comm = comm;
    return comm_partition(h, nranks, rank);
}

// The row partition of a communicator of `nranks` members (RCCL or the in-process stand-in)
static int comm_partition(msdp_handle h, int32_t nranks, int32_t rank) {
    h->nranks = nranks;
    h->rank = rank;
    h->use_comm = true;      // also with nranks == 1: a size-1 communicator exercises the same RCCL calls
    const int cap = rows_capacity(h);
    h->d.row0 = rank * cap;
    int r1 = h->d.row0 + cap;
    if (r1 > h->d.n) r1 = h->d.n;
    h->d.n_loc = r1 > h->d.row0 ? r1 - h->d.row0 : 0;
    int rc = alloc_common(h);
    if (rc) return rc;
    if (h->d.costkind == COST_SPARSE && (rc = upload_sparse_rows(h))) return rc;
    if (h->d.costkind == COST_DENSE && !h->presharded) {
        msdp_set_error("dense C must be created per shard (msdp_create_onlyunitdiag_dense_synthetic)");
        return MSDP_EUNSUPPORTED;
    }
    if ((rc = msdp_alloc_vectors(h, h->pcap))) return rc;
    return halo_setup(h);
}

extern "C" int msdp_debug_shard(msdp_handle h, int32_t nranks, int32_t rank) {
    CHECK_H(h);
    if (nranks < 1 || rank < 0 || rank >= nranks) { msdp_set_error("bad shard (%d of %d)", rank, nranks); return MSDP_EINVAL; }
    if (h->have_point || h->use_comm) { msdp_set_error("debug_shard must precede set_point / comm_init"); return MSDP_ESTATE; }
    if (h->d.costkind == COST_DENSE || h->kind == MSDP_KIND_MULTIBLOCK || h->kind == MSDP_KIND_DUAL_UNITDIAG) {
        msdp_set_error("debug_shard: sparse-C and affine (unitdiag / unittrace / generic) handles only");
        return MSDP_EUNSUPPORTED;
    }
    h->nranks = nranks;
    h->rank = rank;
    h->presharded = true;                  // lets msdp_debug_set_full_rows stand in for the all-gather
    const int cap = rows_capacity(h);
    h->d.row0 = rank * cap;
    int r1 = h->d.row0 + cap;
    if (r1 > h->d.n) r1 = h->d.n;
    h->d.n_loc = r1 > h->d.row0 ? r1 - h->d.row0 : 0;
    int rc = alloc_common(h);
    if (rc) return rc;
    if (h->d.costkind == COST_SPARSE && (rc = upload_sparse_rows(h))) return rc;
    return msdp_alloc_vectors(h, h->pcap);
}

extern "C" int msdp_local_rows(msdp_handle h, int64_t* row0, int64_t* row1) {
    CHECK_H(h);
    if (row0) *row0 = h->d.row0;
    if (row1) *row1 = h->d.row0 + h->d.n_loc;
    return 0;
}

extern "C" int msdp_debug_last_rtr_device_ms(msdp_handle h, double* ms) {
    CHECK_H(h);
    if (!ms) return MSDP_EINVAL;
    *ms = h->last_rtr_device_ms;
    return 0;
}

extern "C" int msdp_tcg_path(msdp_handle h, int32_t* path) {
    CHECK_H(h);
    if (!path) return MSDP_EINVAL;
    if (!h->have_point) { msdp_set_error("tcg_path: no resident point"); return MSDP_ESTATE; }
    *path = msdp_persist_eligible(h) ? 1 : ((h->use_comm && h->lgroup && h->xpersist_last) ? 2 : 0);   // 2: the last call ran the cross-rank persistent tCG
    return 0;
}

// ------------------------------------------------------------------ RTR driver
static int push_ctl(msdp_handle h) {
    HIPCHK(msdp_memcpy_async(h->d.ctl, h->h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, h->stream));
    return 0;
}
static int pull_ctl(msdp_handle h) {
    HIPCHK(msdp_memcpy_async(h->h_ctl, h->d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

static void fill_ctl(msdp_handle h, const msdp_rtr_opts* o) {
    Ctl* c = h->h_ctl;
    const int cur = c->cur;
    const double sigma = c->sigma;
    const double z0 = c->z_sphere[0], z1 = c->z_sphere[1];
    memset(c, 0, sizeof(Ctl));
    c->cur = cur; c->sigma = sigma; c->z_sphere[0] = z0; c->z_sphere[1] = z1;
    c->maxiter = o->maxiter; c->maxinner = o->maxinner; c->mininner = o->mininner;
    c->tolgradnorm = o->tolgradnorm; c->kappa = o->kappa; c->theta = o->theta;
    c->rho_prime = o->rho_prime; c->rho_reg = o->rho_regularization;
    c->persist_refresh = h->tune.persist_refresh;
    c->persist_early = h->tune.persist_early;
    c->pipe_refresh = h->tune.pipe_refresh == 1 ? 2 : h->tune.pipe_refresh;   // never 1 (see msdp_set_option)
    c->pipe_local = h->tune.pipe_local;
    c->persist_goff = h->tune.persist_goff;
    c->psync_backoff = h->tune.psync_backoff;
    c->psync8_backoff = h->tune.psync8_backoff;
    // trustregions.m:363-372; typicaldist: pi*sqrt(n) (ManiSDP_onlyunitdiag.m:137) or pi (spherefactory.m:111)
    // ... or sqrt(n*p) (euclideanfactory.m:57)
    const double typical = (h->d.manifold == MANI_OBLIQUE) ? M_PI * sqrt((double)h->d.n)
                           : (h->d.manifold == MANI_EUCLID ? sqrt((double)h->d.n * (double)h->d.p) : M_PI);
    c->Delta_bar = (o->Delta_bar > 0) ? o->Delta_bar : typical;
    c->Delta0 = (o->Delta0 > 0) ? o->Delta0 : c->Delta_bar / 8.0;
}

#define TCG_CHUNK 8           // tCG trips per enqueued chunk (one hipGraph of 3 x 8 kernel nodes)
static bool use_graphs(msdp_handle h) { return h->tune.graph && !h->use_comm; }

// Start of a tCG (tCG.m:102-157).  Two-launch trips (msdp_trip2.hip): the Hess-vec of trip j+1 rides in the launch that closes
// trip j, so the first one is issued here, behind the initialisation.
static int tcg_begin(msdp_handle h) {
    if (msdp_trip1_ok(h)) {
        // sharded trip with one all-reduce (msdp_trip1.hip): the first product is a direct one on the gradient rows
        int rc;
        h->d.xn = h->nranks;
        if (!h->use_comm) h->d.xs_all = h->d.xs;
        h->trip1_count = 0;
        if ((rc = msdp_launch_trip1_init(h))) return rc;
        if ((rc = msdp_exchange_rows(h, h->d.md))) return rc;
        if ((rc = msdp_launch_trip1_head(h, true))) return rc;
        return msdp_allreduce_partials(h, P_DHD, 1);
    }
    if (msdp_trip2_ok(h)) {
        int rc = msdp_launch_trip2_init(h);
        return rc ? rc : msdp_launch_trip2_head(h);
    }
    return msdp_launch_tcg_init(h);
}
static int enqueue_trips(msdp_handle h, int cnt) {
    int rc;
    if (msdp_trip1_ok(h)) {
        const int refresh = h->tune.persist_refresh;
        for (int t = 0; t < cnt; ++t) {
            if ((rc = msdp_launch_trip1_upd(h))) return rc;                         // tCG.m:166-241
            // eta and r ping-pong: trip t (counted from 0) of a running tCG writes r' into r2 when t is even (after the end of
            // a tCG the launches are no-ops and the buffer does not matter)
            const double* rnew = (h->trip1_count & 1) ? h->d.r : h->d.r2;
            if ((rc = msdp_exchange_rows_sums(h, rnew))) return rc;                 // rows of r' + every rank's three sums
            if ((rc = msdp_launch_trip1_head(h, false))) return rc;                 // tCG.m:227-287, tCG.m:163 by linearity
            ++h->trip1_count;
            // every refresh-th trip multiplies directly once more (inside a graph capture the count is not the replay's: there
            // launch_chunk appends the refresh behind the graph -- one rank, no collective in between)
            if (!h->trip1_capture && refresh > 0 && (h->trip1_count % refresh) == 0) {
                if ((rc = msdp_exchange_rows(h, h->d.md))) return rc;
                if ((rc = msdp_launch_trip1_head(h, true))) return rc;
            }
            if ((rc = msdp_allreduce_partials(h, P_DHD, 1))) return rc;             // <mdelta, H mdelta> over all ranks (tCG.m:166)
        }
        return 0;
    }
    if (msdp_trip2_ok(h)) {
        for (int t = 0; t < cnt; ++t) {
            if ((rc = msdp_launch_trip2_upd(h))) return rc;    // tCG.m:166-241
            if ((rc = msdp_launch_trip2_head(h))) return rc;   // tCG.m:227-287, then tCG.m:163 of the next trip
        }
        return 0;
    }
    for (int t = 0; t < cnt; ++t) {
        if ((rc = msdp_launch_hess(h))) return rc;        // tCG.m:163
        if ((rc = msdp_launch_upd1(h))) return rc;        // tCG.m:166-241
        if ((rc = msdp_launch_upd2(h))) return rc;        // tCG.m:249-287
    }
    return 0;
}

// One hipGraph of CH tCG trips (3*CH kernel nodes).  All kernel arguments are the Dev
// struct by value and all run-time state lives in device memory, so the same executable
// graph is replayed for every chunk until the Dev struct changes (new p / reallocation).
// Kernels of a finished tCG exit at their first instruction, so replaying a whole chunk
// past the end of the solve is safe.
static int ensure_chunk_graph(msdp_handle h, int CH) {
    h->d.full = h->d.md;
    // The affine kinds bake the current slot's pointers (eS[cur], Y[cur]) into the launches on the host, so
    // they keep one executable graph per slot; the other kinds read `cur` on the device.
    const int slot = (h->d.costkind == COST_AFFINE) ? h->h_ctl->cur : 0;
    if (h->chunk_len != CH || memcmp(&h->chunk_sig, &h->d, sizeof(Dev)) != 0) {
        for (int s = 0; s < 2; ++s)
            if (h->chunk_execs[s]) { (void)hipGraphExecDestroy(h->chunk_execs[s]); h->chunk_execs[s] = nullptr; }
        h->chunk_sig = h->d;
        h->chunk_len = CH;
    }
    if (!h->chunk_execs[slot]) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        const int count_keep = h->trip1_count;
        h->trip1_capture = true;
        int rc = enqueue_trips(h, CH);
        h->trip1_capture = false;
        h->trip1_count = count_keep;
        hipError_t e = hipStreamEndCapture(h->stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) { msdp_set_error("graph capture failed: %s", hipGetErrorString(e)); return MSDP_EHIP; }
        e = hipGraphInstantiate(&h->chunk_execs[slot], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { msdp_set_error("graph instantiate failed: %s", hipGetErrorString(e)); h->chunk_execs[slot] = nullptr; return MSDP_EHIP; }
    }
    h->chunk_exec = h->chunk_execs[slot];
    return 0;
}

static int launch_chunk(msdp_handle h, int CH, bool graph) {
    if (graph) {
        HIPCHK(hipGraphLaunch(h->chunk_exec, h->stream));
        if (msdp_trip1_ok(h)) {
            // msdp_trip1.hip on one rank: the refresh schedule of the linear products, behind every (refresh / CH)-th replay
            const int refresh = h->tune.persist_refresh, before = h->trip1_count;
            h->trip1_count += CH;
            if (refresh > 0 && h->trip1_count / refresh != before / refresh) {
                int rc = msdp_exchange_rows(h, h->d.md);
                if (!rc) rc = msdp_launch_trip1_head(h, true);
                if (rc) return rc;
            }
        }
        return 0;
    }
    return enqueue_trips(h, CH);
}

// tCG of the current TR iteration when the rows are sharded over a communicator.  Every rank must issue the SAME
// sequence of collectives, so how many chunks are enqueued may depend only on device state that is identical on all
// ranks: the `tcg_running` flag, which every rank computes from the same all-reduced sums.  The flag after each chunk
// is copied to a pinned word behind the chunk (an event marks the copy); the host stays ONE chunk ahead of the device
// -- chunk i+1 is already enqueued when the flag of chunk i is read -- so the stream never drains while the host
// decides, and at most one chunk of no-op trips (whose collectives still run) follows the end of a tCG.
static int run_tcg_lockstep(msdp_handle h, int maxinner) {
    const int CH = TCG_CHUNK;
    const int nchunks = (maxinner + CH - 1) / CH;
    int rc;
    h->d.status = nullptr;                                         // no host-mapped progress word on this path
    if ((rc = tcg_begin(h))) return rc;                            // trustregions.m:484-496
    int enq = 0;
    auto push_chunk = [&]() -> int {
        int r2 = enqueue_trips(h, CH);
        if (r2) return r2;
        HIPCHK(msdp_memcpy_async((void*)&h->h_flags[enq & 1], &h->d.ctl->tcg_running, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->ev_flag[enq & 1], h->stream));
        ++enq;
        return 0;
    };
    if ((rc = push_chunk())) return rc;
    if (nchunks > 1 && (rc = push_chunk())) return rc;
    for (int i = 0; i < enq; ++i) {
        HIPCHK(hipEventSynchronize(h->ev_flag[i & 1]));
        if (!h->h_flags[i & 1]) break;                             // finished inside chunk i; what follows is a no-op
        if (enq < nchunks && (rc = push_chunk())) return rc;       // slot (i & 1) is free again: chunk i + 2 takes it
    }
    return 0;
}

// Run the tCG inner loop of the current TR iteration: chunks of CH trips are enqueued one
// ahead of the device (so the graph-launch latency is hidden) while the host polls the
// host-mapped progress word the lead thread of k_tcg_upd2 publishes every trip.
static int run_tcg(msdp_handle h, int maxinner, int k, bool* done_out = nullptr) {
    const int CH = TCG_CHUNK;
    const bool graph = use_graphs(h);
    int rc;
    if (graph && (rc = ensure_chunk_graph(h, CH))) return rc;
    if ((rc = tcg_begin(h))) return rc;                           // trustregions.m:484-496
    int enq = 0;
    if ((rc = launch_chunk(h, CH, graph))) return rc;
    enq = 1;
    if (enq * CH < maxinner) { if ((rc = launch_chunk(h, CH, graph))) return rc; enq = 2; }
    const unsigned long long want = (unsigned long long)(unsigned)(k + 1);
    const auto t0 = std::chrono::steady_clock::now();
    auto last_query = t0;
    long spins = 0;
    for (;;) {
        const unsigned long long s = *h->h_status;
        if ((s >> 32) == want) {
            const int active = (int)(s & 1ULL);
            const int jraw = (int)((s & 0xffffffffULL) >> 1);
            if (done_out && (jraw & 0x40000000)) *done_out = true;
            const int j = jraw & 0x3fffffff;
            if (!active) break;
            if (enq * CH < maxinner && j >= (enq - 1) * CH) {
                if ((rc = launch_chunk(h, CH, graph))) return rc;
                ++enq;
                continue;
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(10));     // polite polling (a chunk of 8 trips lasts ~0.2 ms)
        if ((++spins & 0xff) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - last_query).count() > 1.0) {
            last_query = std::chrono::steady_clock::now();      // hipStreamQuery is not a cheap poll (can block for tens of ms): safety net only
            if (hipStreamQuery(h->stream) == hipSuccess) {
                // everything enqueued has run: the final status must be visible now
                const unsigned long long s2 = *h->h_status;
                if ((s2 >> 32) == want && !(s2 & 1ULL)) {
                    if (done_out && (((s2 & 0xffffffffULL) >> 1) & 0x40000000)) *done_out = true;
                    break;
                }
                if (enq * CH >= maxinner || (s2 >> 32) != want) {
                    msdp_set_error("tCG progress word inconsistent (status %llx, TR iteration %d)", s2, k);
                    return MSDP_EHIP;
                }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                msdp_set_error("tCG made no progress for 120 s");
                return MSDP_EHIP;
            }
        }
    }
    return 0;
}

static void restore_status_ptr(msdp_handle h) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, (void*)h->h_status, 0) == hipSuccess) h->d.status = (unsigned long long*)dp;
}

// Did a persistent launch give up on a grid synchronisation?  (Its bounded spins turn a would-be hang -- the
// workgroups of the launch not all resident because something else occupies CUs -- into this flag.)
static int persist_timed_out(msdp_handle h, bool* out) {
    int perr = 0;
    HIPCHK(msdp_memcpy(&perr, h->psync_err, sizeof(int), hipMemcpyDeviceToHost));
    *out = perr != 0;
    return 0;
}

// ctl and the persistent kernels' error word with ONE host synchronisation (the word lands in a pinned slot of h_flags)
static int pull_ctl_and_err(msdp_handle h, bool* timed_out) {
    HIPCHK(msdp_memcpy_async((void*)&h->h_flags[8], h->psync_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    int rc = pull_ctl(h);
    if (rc) return rc;
    *timed_out = h->h_flags[8] != 0;
    return 0;
}

// The body of msdp_rtr.  *timed_out: a persistent launch reported a grid-synchronisation time-out (the resident
// point is then in an undefined state; the caller restores the start point and calls again, which takes the
// chunked path because h->persist_failed is set).
static int rtr_core(msdp_handle h, const msdp_rtr_opts* opts, bool* timed_out) {
    int rc;
    *timed_out = false;
    fill_ctl(h, opts);
    *h->h_status = 0;
    if ((rc = push_ctl(h))) return rc;
    int cur = h->h_ctl->cur;
    if ((rc = msdp_launch_costgrad(h, cur))) return rc;          // trustregions.m:405
    if ((rc = msdp_launch_rtr_begin(h))) return rc;
    const bool timing = h->tune.timing != 0;
    double t_tcg = 0.0, t_rest = 0.0, t_enq_sum = 0.0, t_enq_max = 0.0;
    const bool async_tr = h->d.costkind == COST_SPARSE && !h->use_comm;
    const bool persist = async_tr && msdp_persist_eligible(h);
    const bool fused = persist && !h->tune.fail_persist && msdp_persist_fused_ok(h);
    // the fused launch reads ctl on the device (a solve that is already done is a no-op there): the host needs the state of
    // the start point only on the other paths -- one host round trip less per call (20-100 us, host to host)
    if (!fused && (rc = pull_ctl(h))) return rc;
    if (!fused) HIPCHK(hipEventRecord(h->ev0, h->stream));       // (msdp_debug_last_rtr_device_ms: closed in msdp_rtr)
    h->last_rtr_fused = fused;
    if (persist && h->tune.fail_persist) {                       // test hook: behave as if the launch had timed out
        h->tune.fail_persist = 0;
        *timed_out = true;
        return 0;
    }
    if (fused) {
        // Fused path: the whole trustregions() loop (every tCG, retraction, cost/gradient at the proposal and the
        // accept/reject logic) runs in ONE launch; the host only waits for it (msdp_persist.hip, FUSE = true).
        const auto ta = std::chrono::steady_clock::now();
        h->d.status = nullptr;                                            // no progress word needed
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        rc = msdp_launch_rtr_fused(h);
        restore_status_ptr(h);
        if (rc) return rc;
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        if ((rc = pull_ctl_and_err(h, timed_out))) return rc;
        {   // (the stream is idle: the events are complete)
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_rtr_device_ms = (double)ms; else (void)hipGetLastError();
        }
        if (*timed_out) return 0;
        t_tcg = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
    } else if (persist) {
        // Persistent path: one launch runs the whole tCG of a TR iteration with the working set on chip
        // (msdp_persist.hip), a second one the rest of the iteration (msdp_trtail.hip: retraction, cost and gradient at
        // the proposal, accept/reject; each clears the other's synchronisation slots).  The host stays one TR iteration
        // ahead of the device: iteration i+1 is enqueued as soon as the kernel of iteration i publishes that it has
        // started; a finished solve (ctl->done) turns everything still enqueued into no-ops and is reported through
        // the same progress word.
        bool first_iter = true;
        auto enqueue_iter = [&]() -> int {
            int r2;
            if ((r2 = msdp_launch_tcg_persist(h, first_iter ? 1 : 0))) return r2;   // trustregions.m:484-496 + tCG.m
            first_iter = false;
            return msdp_launch_tr_tail(h);                                // :540-729
        };
        // (round 5: up to AHEAD iterations beyond the last one known to have started -- with ONE the device waited for the host
        // whenever an iteration was shorter than the host's polling sleep + two launches: 78 us per TR iteration whatever its tCG
        // (tools/fused_overhead_probe.py).  What is enqueued behind a finished solve returns at once: at most AHEAD - 1 pairs of
        // empty launches per call.)
        const int AHEAD = 3;
        int enq = 0, started = 0;
        bool done = false;
        const auto ta = std::chrono::steady_clock::now();
        auto last_query = ta;
        while (!done) {
            while (enq < opts->maxiter && enq < started + AHEAD) {
                const auto te = std::chrono::steady_clock::now();
                if ((rc = enqueue_iter())) return rc;
                if (timing) {
                    const double de = std::chrono::duration<double>(std::chrono::steady_clock::now() - te).count();
                    t_enq_sum += de; if (de > t_enq_max) t_enq_max = de;
                }
                ++enq;
                last_query = std::chrono::steady_clock::now();
            }
            if (started >= enq) break;                           // every iteration of the budget has started (or maxiter = 0)
            long spins = 0;
            for (;;) {
                const unsigned long long s = *h->h_status;
                const int it = (int)(s >> 32);
                if (it > started && it <= enq) {
                    started = it;
                    if (((s & 0xffffffffULL) >> 1) & 0x40000000) done = true;
                    break;
                }
                // a TR iteration lasts 0.02-2 ms: poll politely (a hard spin burns a full core; under a container CPU quota that
                // got this thread throttled for tens of ms at a time, seen as 60 ms holes in the kernel trace of the G81 solve)
                std::this_thread::sleep_for(std::chrono::microseconds(20));
                if ((++spins & 0xff) == 0) {
                    // hipStreamQuery is NOT a cheap poll (every call makes the runtime touch the queue; called every
                    // few microseconds it stalled the stream for tens of ms, seen as gaps in the kernel trace): it is
                    // only the safety net against a lost progress word; on a stream that is running a long kernel one call was
                    // measured to block for ~40 ms, so ask only after 2 s without any progress
                    const auto now = std::chrono::steady_clock::now();
                    if (std::chrono::duration<double>(now - last_query).count() > 2.0) {
                        last_query = now;
                        if (hipStreamQuery(h->stream) == hipSuccess) {
                            const unsigned long long s2 = *h->h_status;
                            const int it2 = (int)(s2 >> 32);
                            if (it2 > started && it2 <= enq) { started = it2; if (((s2 & 0xffffffffULL) >> 1) & 0x40000000) done = true; break; }
                            // everything enqueued has run and the word never arrived: a launch that gave up on a grid
                            // synchronisation exits without publishing
                            if ((rc = persist_timed_out(h, timed_out))) return rc;
                            if (*timed_out) return 0;
                            msdp_set_error("persistent tCG: progress word inconsistent (status %llx, expected iteration %d..%d)", s2, started + 1, enq);
                            return MSDP_EHIP;
                        }
                    }
                    if (std::chrono::duration<double>(now - ta).count() > 300.0) {
                        msdp_set_error("persistent tCG made no progress for 300 s");
                        return MSDP_EHIP;
                    }
                }
            }
        }
        if ((rc = pull_ctl(h))) return rc;
        if ((rc = persist_timed_out(h, timed_out))) return rc;
        if (*timed_out) return 0;
        t_tcg = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
    } else if (async_tr) {
        // No host sync between TR iterations: the proposal slot is resolved on the device, the next
        // iteration's tcg_init + first chunks are enqueued right behind k_rtr_decide, and k_tcg_init
        // publishes `done` through the progress word (a finished solve turns everything enqueued into no-ops).
        int k = 0;
        bool done = false;
        while (k < opts->maxiter) {
            const auto ta = std::chrono::steady_clock::now();
            if ((rc = run_tcg(h, opts->maxinner, k, &done))) return rc;
            const auto tb = std::chrono::steady_clock::now();
            t_tcg += std::chrono::duration<double>(tb - ta).count();
            if (done) break;
            if ((rc = msdp_launch_retract(h))) return rc;             // :540
            if ((rc = msdp_launch_costgrad(h, 3))) return rc;         // :544 (proposal slot, device-resolved)
            if ((rc = msdp_launch_rtr_decide(h))) return rc;          // :548-729
            ++k;
        }
        if ((rc = pull_ctl(h))) return rc;
    } else {
        // One host synchronisation per TR iteration (dense / affine kinds bake the slot into their launches; with a
        // communicator the tCG runs in lock-step, see run_tcg_lockstep)
        // in-process ranks, sparse C: ONE persistent tCG spans the ranks' launches (msdp_persist.hip XR) -- no collective per trip;
        // every member must be able to (a vote), otherwise all of them take the lock-step chunks
        bool xp = false;
        if (h->use_comm && h->lgroup && h->nranks > 1 && h->d.costkind == COST_SPARSE) {
            int agreed = 0;
            if ((rc = local_vote_min(h, msdp_xpersist_eligible(h, h->nranks), &agreed))) return rc;
            xp = agreed != 0;
            if (xp && (rc = xr_begin(h, &xp))) return rc;
        }
        h->xpersist_last = xp;
        while (!h->h_ctl->done) {                                     // trustregions.m:441
            cur = h->h_ctl->cur;
            const auto ta = std::chrono::steady_clock::now();
            if (xp) {
                h->d.status = nullptr;
                rc = xr_launch(h);
                restore_status_ptr(h);
            }
            else if (h->use_comm) rc = run_tcg_lockstep(h, opts->maxinner);
            else rc = run_tcg(h, opts->maxinner, h->h_ctl->k);        // :495
            if (rc) return rc;
            const auto tb = std::chrono::steady_clock::now();
            if (xp && h->lgroup->ipc && h->tune.xtail) {
                // members in different processes: the rest of the iteration is ONE launch per member too (k_tr_tail_obl<.., XR>) -- the
                // proposal rows through the group's exchange buffer, barrier and reduction over its slots, no collective
                if ((rc = xr_tail(h))) return rc;
                // ... and the decision stays on the device: three more iterations are enqueued before the host looks (both kernels
                // return at once when the solve is done, on every member alike), one host synchronisation per FOUR iterations
                for (int ahead = 0; ahead < 3 && !rc; ++ahead) {
                    h->d.status = nullptr;
                    rc = xr_launch(h);
                    restore_status_ptr(h);
                    if (!rc) rc = xr_tail(h);
                }
                if (rc) return rc;
            } else {
                if ((rc = msdp_launch_retract(h))) return rc;             // :540
                if ((rc = msdp_launch_costgrad(h, cur ^ 1))) return rc;   // :544
                if ((rc = msdp_launch_rtr_decide(h))) return rc;          // :548-729
            }
            if ((rc = pull_ctl(h))) return rc;
            if (xp && (rc = xr_check(h))) return rc;
            const auto tc = std::chrono::steady_clock::now();
            t_tcg += std::chrono::duration<double>(tb - ta).count();
            t_rest += std::chrono::duration<double>(tc - tb).count();
        }
    }
    if (timing) {
        fprintf(stderr, "[msdp_rtr] enqueue total %.3f ms, slowest %.3f ms\n", t_enq_sum * 1e3, t_enq_max * 1e3);
        fprintf(stderr, "[msdp_rtr] p=%d ld=%d G=%d path=%d k=%d hessvecs=%d acc=%d rej=%d  tCG phase %.3f ms  (retract+cost+decide+sync) %.3f ms\n",
                h->d.p, h->d.ld, h->d.G, persist ? 1 : 0, h->h_ctl->k, h->h_ctl->hessvecs,
                h->h_ctl->accepted, h->h_ctl->rejected, t_tcg * 1e3, t_rest * 1e3);
    }
    return 0;
}

extern "C" int msdp_rtr(msdp_handle h, const msdp_rtr_opts* opts, msdp_rtr_stats* stats) {
    CHECK_H(h);
    if (!opts) { msdp_set_error("rtr: null options"); return MSDP_EINVAL; }
    if (!h->have_point) { msdp_set_error("rtr: no resident point (call msdp_set_point)"); return MSDP_ESTATE; }
    if (opts->rho_prime >= 0.25) { msdp_set_error("options.rho_prime must be strictly smaller than 1/4"); return MSDP_EINVAL; }
    if (opts->maxinner < 1 || opts->maxiter < 0) { msdp_set_error("rtr: maxinner >= 1 and maxiter >= 0 required"); return MSDP_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    h->last_opts = *opts;
    (void)msdp_window_eligible(h);                                 // (builds the patch plan of the LDS-staged S*U outside any graph capture)
    // The persistent kernels assume that all their workgroups are resident together.  When the GPU is shared (a
    // second handle solving on another stream, another process) that can fail; the launch then gives up after a
    // bounded spin.  Keep a copy of the start point so that the call can be repeated on the chunked path.
    const int cur0 = h->h_ctl->cur;
    const size_t cnt = (size_t)rows_capacity(h) * h->ldcap;
    const bool guard = h->d.costkind == COST_SPARSE && !h->use_comm && msdp_persist_eligible(h);
    if (guard) {
        if (h->rtr_start_cap < cnt) {
            if (h->rtr_start) dev_free(h, h->rtr_start);
            h->rtr_start = nullptr; h->rtr_start_cap = 0;
            int rc0 = dev_alloc<double>(h, &h->rtr_start, cnt);
            if (rc0) return rc0;
            h->rtr_start_cap = cnt;
        }
        HIPCHK(msdp_memcpy_async(h->rtr_start, h->d.Y[cur0], cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    bool timed_out = false;
    int rc = rtr_core(h, opts, &timed_out);
    if (rc) return rc;
    if (timed_out) {
        if (!guard) { msdp_set_error("persistent tCG: grid synchronisation timed out"); return MSDP_EHIP; }
        fprintf(stderr, "libmanisdp_hip: a persistent tCG launch could not synchronise its workgroups (GPU shared with another "
                        "launch?); this handle continues on the chunked path\n");
        h->persist_failed = true;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemset(h->psync_err, 0, sizeof(int)));
        h->h_ctl->cur = cur0;
        HIPCHK(msdp_memcpy_async(h->d.Y[cur0], h->rtr_start, cnt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        restore_status_ptr(h);
        h->chunk_len = 0;                                          // re-capture the chunk graph against the current Dev
        rc = rtr_core(h, opts, &timed_out);
        if (rc) return rc;
        if (timed_out) { msdp_set_error("persistent tCG: time-out on the chunked path (internal error)"); return MSDP_EHIP; }
    }
    h->state_valid = true;
    h->gradnorm_valid = true;
    if (!h->last_rtr_fused) {
        // (the other paths: the stream time of everything the call enqueued behind the evaluation of its start point, host gaps included)
        float ms = 0.f;
        if (hipEventRecord(h->ev1, h->stream) == hipSuccess && hipEventSynchronize(h->ev1) == hipSuccess &&
            hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_rtr_device_ms = (double)ms; else (void)hipGetLastError();
    }
    if (stats) {
        const Ctl* c = h->h_ctl;
        memset(stats, 0, sizeof(*stats));
        stats->cost = c->fx; stats->gradnorm = c->norm_grad; stats->Delta = c->Delta;
        stats->iters = c->k; stats->hessvecs = c->hessvecs; stats->accepted = c->accepted;
        stats->rejected = c->rejected; stats->cost_evals = c->cost_evals;
        stats->last_stop_inner = c->last_stop_inner;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

extern "C" int msdp_rtr_host(msdp_handle h, int32_t p, double* Y, const msdp_rtr_opts* opts, msdp_rtr_stats* stats) {
    int rc = msdp_set_point(h, p, Y);
    if (rc) return rc;
    if ((rc = msdp_rtr(h, opts, stats))) return rc;
    return msdp_get_point(h, Y);
}

// ------------------------------------------------------------------ fine-grained ops
static int ensure_state(msdp_handle h) {
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    (void)msdp_window_eligible(h);                                 // (patch plan of the LDS-staged S*U: built here, outside any graph capture)
    if (h->state_valid) return 0;
    h->h_ctl->done = 0;
    h->h_ctl->bench_mode = 0;
    int rc = push_ctl(h);
    if (rc) return rc;
    if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
    h->state_valid = true;
    return 0;
}

extern "C" int msdp_cost(msdp_handle h, double* f) {
    CHECK_H(h);
    if (!f) return MSDP_EINVAL;
    int rc = ensure_state(h);
    if (rc) return rc;
    // re-run the reduction of the stored partials only if they are still those of the
    // resident point; simplest is to recompute the cost
    h->h_ctl->done = 0;
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
    if ((rc = msdp_k_sum_to(h, P_F, &h->d.ctl->fx))) return rc;
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, &h->d.ctl->fx, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *f = v;
    return 0;
}

extern "C" int msdp_rgrad(msdp_handle h, double* G) {
    CHECK_H(h);
    int rc = ensure_state(h);
    if (rc) return rc;
    return download_rows(h, h->d.Gr[host_cur(h)], G);
}

extern "C" int msdp_hessvec(msdp_handle h, const double* U, double* H) {
    CHECK_H(h);
    int rc = ensure_state(h);
    if (rc) return rc;
    if ((rc = upload_rows(h, U, h->d.md))) return rc;
    if ((rc = msdp_k_set_active(h, 1))) return rc;
    if ((rc = msdp_launch_hess(h))) return rc;
    if ((rc = msdp_k_set_active(h, 0))) return rc;
    return download_rows(h, h->d.Hmd, H);
}

extern "C" int msdp_proj(msdp_handle h, const double* U, double* V) {
    CHECK_H(h);
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    int rc = upload_rows(h, U, h->d.W0);
    if (rc) return rc;
    if (h->d.manifold == MANI_OBLIQUE) rc = msdp_k_proj_obl(h, h->d.Y[host_cur(h)], h->d.W0, h->d.W1);
    else rc = msdp_sphere_proj(h, h->d.Y[host_cur(h)], h->d.W0, h->d.W1);
    if (rc) return rc;
    return download_rows(h, h->d.W1, V);
}

extern "C" int msdp_retr(msdp_handle h, const double* U, double* Z) {
    CHECK_H(h);
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    int rc = upload_rows(h, U, h->d.W0);
    if (rc) return rc;
    if (h->d.manifold == MANI_OBLIQUE) rc = msdp_k_retr_obl(h, h->d.Y[host_cur(h)], h->d.W0, h->d.W1, 1.0);
    else rc = msdp_sphere_retr(h, h->d.Y[host_cur(h)], h->d.W0, h->d.W1, 1.0);
    if (rc) return rc;
    return download_rows(h, h->d.W1, Z);
}

extern "C" int msdp_get_z(msdp_handle h, double* z) {
    CHECK_H(h);
    if (h->kind != MSDP_KIND_ONLYUNITDIAG) { msdp_set_error("get_z: only for onlyunitdiag handles"); return MSDP_EUNSUPPORTED; }
    int rc = ensure_state(h);
    if (rc) return rc;
    HIPCHK(msdp_memcpy_async(z + h->d.row0, h->d.eG[host_cur(h)], (size_t)h->d.n_loc * sizeof(double),
                          hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// z of ALL rows on every rank of a row-sharded handle (one all-gather): input of the replicated host loop
extern "C" int msdp_get_z_all(msdp_handle h, double* z) {
    CHECK_H(h);
    if (h->kind != MSDP_KIND_ONLYUNITDIAG || !z) { msdp_set_error("get_z_all: onlyunitdiag handles / null out"); return MSDP_EUNSUPPORTED; }
    if (!h->use_comm || h->nranks == 1) return msdp_get_z(h, z);
    int rc = ensure_state(h);
    if (rc) return rc;
    const size_t cap = (size_t)rows_capacity(h);
    if ((rc = msdp_allgather_vec(h, h->d.eG[host_cur(h)], h->full_buf, cap))) return rc;     // the gather buffer is free here
    HIPCHK(msdp_memcpy_async(z, h->full_buf, (size_t)h->d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// co() of the line search at retr(Y + alpha*U): for onlyunitdiag co = sum((Y*C).*Y) = 2 f.
extern "C" int msdp_linesearch_cost(msdp_handle h, const double* U, double alpha, double* val) {
    CHECK_H(h);
    if (!h->have_point || !val) { msdp_set_error("linesearch_cost: no point / null out"); return MSDP_ESTATE; }
    int rc;
    const int cur = host_cur(h);
    Dev& d = h->d;
    if (U && alpha != 0.0) {
        if ((rc = upload_rows(h, U, d.W0))) return rc;
        if (d.manifold == MANI_OBLIQUE) rc = msdp_k_retr_obl(h, d.Y[cur], d.W0, d.Y[cur ^ 1], alpha);
        else rc = msdp_sphere_retr(h, d.Y[cur], d.W0, d.Y[cur ^ 1], alpha);
        if (rc) return rc;
    } else {
        HIPCHK(msdp_memcpy_async(d.Y[cur ^ 1], d.Y[cur], (size_t)rows_capacity(h) * d.ld * sizeof(double),
                              hipMemcpyDeviceToDevice, h->stream));
    }
    if (d.costkind == COST_AFFINE) return msdp_affine_linesearch_cost(h, d.Y[cur ^ 1], val);
    h->h_ctl->done = 0;
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_launch_costgrad(h, cur ^ 1))) return rc;
    if ((rc = msdp_k_sum_to(h, P_F, &d.ctl->fx_prop))) return rc;
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, &d.ctl->fx_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *val = 2.0 * v;
    return 0;
}

// Adopt the retraction of Y + alpha*U as the new resident point (result of line_search).
extern "C" int msdp_linesearch_accept(msdp_handle h) {
    CHECK_H(h);
    h->h_ctl->cur ^= 1;
    h->state_valid = false;
    h->gradnorm_valid = false;
    return 0;
}

extern "C" int msdp_escape_eigs(msdp_handle h, int32_t k, double tol, int32_t maxit, double* lam_min, double* V,
                                double* lam_max, int32_t* iters) {
    CHECK_H(h);
    int rc = ensure_state(h);
    if (rc) return rc;
    if (!h->gradnorm_valid) {
        // |S*Y|_F and f at the resident point (decides whether span(Y) may be deflated)
        h->h_ctl->done = 0;
        if ((rc = push_ctl(h))) return rc;
        if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
        if ((rc = msdp_k_sum_to(h, P_GG, &h->d.ctl->gg_prop))) return rc;
        if ((rc = msdp_k_sum_to(h, P_F, &h->d.ctl->fx_prop))) return rc;
        double v[2] = {0.0, 0.0};
        HIPCHK(msdp_memcpy_async(&v[0], &h->d.ctl->gg_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(msdp_memcpy_async(&v[1], &h->d.ctl->fx_prop, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->h_ctl->norm_grad = sqrt(v[0] > 0 ? v[0] : 0.0);
        h->h_ctl->fx = v[1];
        h->gradnorm_valid = true;
    }
    return msdp_escape_impl(h, k, tol, maxit, lam_min, V, lam_max, iters, nullptr);
}

// Same for an explicit dense symmetric S handed over by the AL loop of the affine kinds (S = C - A'y - diag(z) or
// - z*I is formed on the host exactly as in ManiSDP_unitdiag.m:65-67 / ManiSDP_unittrace.m:65-67): replaces the
// O(n^3) eig(S) of :68 by Lanczos runs whose S*v is a device GEMV.  span(Y) is deflated when the last RTR call
// ended with a small gradient (grad = 2*S*Y for these problems).
extern "C" int msdp_escape_eigs_matrix(msdp_handle h, const double* S, int32_t k, double tol, int32_t maxit,
                                       double* lam_min, double* V, double* lam_max, int32_t* iters) {
    CHECK_H(h);
    if (!S || !lam_min || !V) { msdp_set_error("escape_eigs_matrix: null argument"); return MSDP_EINVAL; }
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    const int n = h->d.n, nS = msdp_dense_nS(n);
    double* M = nullptr;
    if (hipMalloc((void**)&M, (size_t)n * nS * sizeof(double)) != hipSuccess) { msdp_set_error("escape_eigs_matrix: allocation failed"); return MSDP_ENOMEM; }
    hipError_t e = hipMemsetAsync(M, 0, (size_t)n * nS * sizeof(double), h->stream);
    if (e == hipSuccess)
        e = msdp_memcpy2d_async(M, (size_t)nS * sizeof(double), S, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n,
                             hipMemcpyHostToDevice, h->stream);
    int rc = 0;
    if (e != hipSuccess) { msdp_set_error("escape_eigs_matrix: upload failed: %s", hipGetErrorString(e)); rc = MSDP_EHIP; }
    if (!rc) rc = msdp_escape_impl(h, k, tol, maxit, lam_min, V, lam_max, iters, M);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(M);
    return rc;
}

int msdp_affine_al_primal(msdp_handle h, double* obj, double* Ax_host);       // msdp_affine.hip
int msdp_affine_al_dual(msdp_handle h, const double* y_host, double* z_host);

extern "C" int msdp_al_primal(msdp_handle h, double* obj, double* Ax) {
    CHECK_H(h);
    if (!obj || !Ax) { msdp_set_error("al_primal: null argument"); return MSDP_EINVAL; }
    if (h->d.costkind != COST_AFFINE) { msdp_set_error("al_primal: affine handles only"); return MSDP_EUNSUPPORTED; }
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    h->state_valid = false;                    // the scratch vectors / partial sums of the resident state are reused
    return msdp_affine_al_primal(h, obj, Ax);
}

extern "C" int msdp_al_dual(msdp_handle h, const double* y, double* z) {
    CHECK_H(h);
    if (!y || (!z && h->kind != MSDP_KIND_GENERIC)) { msdp_set_error("al_dual: null argument"); return MSDP_EINVAL; }
    if (h->d.costkind != COST_AFFINE) { msdp_set_error("al_dual: affine handles only"); return MSDP_EUNSUPPORTED; }
    if (!h->have_point) { msdp_set_error("no resident point"); return MSDP_ESTATE; }
    h->state_valid = false;
    int rc = msdp_affine_al_dual(h, y, z);
    h->dual_valid = (rc == 0);
    return rc;
}

extern "C" int msdp_escape_eigs_dual(msdp_handle h, int32_t k, double tol, int32_t maxit, double* lam_min, double* V,
                                     double* lam_max, int32_t* iters) {
    CHECK_H(h);
    if (!lam_min || !V) { msdp_set_error("escape_eigs_dual: null argument"); return MSDP_EINVAL; }
    if (h->d.costkind != COST_AFFINE || !h->dual_valid) { msdp_set_error("escape_eigs_dual: call msdp_al_dual first"); return MSDP_ESTATE; }
    // per-block storage: d.Sdual holds sum n_i * nS_i doubles, not an n x nS matrix -- the eigen-pairs come block by block
    if (h->blocked) { msdp_set_error("escape_eigs_dual: this multiblock handle stores its blocks only (msdp_block_eigs)"); return MSDP_EUNSUPPORTED; }
    int rc = msdp_escape_impl(h, k, tol, maxit, lam_min, V, lam_max, iters, h->d.Sdual);
    (void)hipStreamSynchronize(h->stream);
    return rc;
}

extern "C" int msdp_escape_info(msdp_handle h, int32_t* nvalid, int32_t* converged, double* residual) {
    CHECK_H(h);
    if (nvalid) *nvalid = h->esc_nvalid;
    if (converged) *converged = h->esc_converged;
    if (residual) *residual = h->esc_maxres;
    return 0;
}

// Test / measurement only: average time of one collective call of the given kind on the handle's stream (reps back to back).
//   0 exchange of the direction rows, 1 all-reduce of one partial-sum array, 2 exchange + sums (msdp_trip1.hip), 3 all-reduce of
//   three arrays, 4 rows then sums as two ungrouped all-gathers
extern "C" int msdp_debug_time_collective(msdp_handle h, int32_t which, int32_t reps, double* avg_us) {
    CHECK_H(h);
    if (!avg_us || reps < 1 || which < 0 || which > 4) return MSDP_EINVAL;
    if (!h->use_comm) { msdp_set_error("debug_time_collective: no communicator"); return MSDP_ESTATE; }
    int rc = 0;
    auto one = [&]() -> int {
        switch (which) {
            case 0: return msdp_exchange_rows(h, h->d.md);
            case 1: return msdp_allreduce_partials(h, P_DHD, 1);
            case 2: return msdp_exchange_rows_sums(h, h->d.md);
            case 3: return msdp_allreduce_partials(h, P_S1, 3);
            default: { int r = msdp_exchange_rows(h, h->d.md); return r ? r : msdp_allgather_vec(h, h->d.xs, h->d.xs_all, 4); }
        }
    };
    const long long keep = h->coll_calls;
    for (int i = 0; i < 3 && !rc; ++i) rc = one();
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps && !rc; ++i) rc = one();
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    h->coll_calls = keep;
    if (rc) return rc;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_us = 1e3 * (double)ms / reps;
    return 0;
}

extern "C" int msdp_debug_collective_calls(msdp_handle h, int64_t* calls) {
    CHECK_H(h);
    if (!calls) return MSDP_EINVAL;
    *calls = h->coll_calls;
    return 0;
}

extern "C" int msdp_escape_method(msdp_handle h, int32_t* method) {
    CHECK_H(h);
    if (!method) return MSDP_EINVAL;
    *method = h->esc_method_last;
    return 0;
}

extern "C" int msdp_escape_lower_bound(msdp_handle h, double* lam_lower) {
    CHECK_H(h);
    if (!lam_lower) return MSDP_EINVAL;
    *lam_lower = h->esc_lower;
    return 0;
}

extern "C" int msdp_get_dual_slack(msdp_handle h, double* S) {
    CHECK_H(h);
    if (!S) { msdp_set_error("get_dual_slack: null argument"); return MSDP_EINVAL; }
    if (h->d.costkind != COST_AFFINE || !h->dual_valid) { msdp_set_error("get_dual_slack: call msdp_al_dual first"); return MSDP_ESTATE; }
    if (h->blocked) { msdp_set_error("get_dual_slack: this multiblock handle stores its blocks only (msdp_get_dual_slack_block)"); return MSDP_EUNSUPPORTED; }
    const int n = h->d.n, nS = msdp_dense_nS(n);
    HIPCHK(msdp_memcpy2d_async(S, (size_t)n * sizeof(double), h->d.Sdual, (size_t)nS * sizeof(double), (size_t)n * sizeof(double), n,
                            hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int msdp_get_dual_slack_block(msdp_handle h, int64_t row0, int64_t nb, double* S) {
    CHECK_H(h);
    if (!S) { msdp_set_error("get_dual_slack_block: null argument"); return MSDP_EINVAL; }
    if (h->d.costkind != COST_AFFINE || !h->dual_valid) { msdp_set_error("get_dual_slack_block: call msdp_al_dual first"); return MSDP_ESTATE; }
    const int n = h->d.n, nS = msdp_dense_nS(n);
    if (row0 < 0 || nb < 1 || row0 + nb > n) { msdp_set_error("get_dual_slack_block: rows %lld..%lld outside 0..%d", (long long)row0, (long long)(row0 + nb), n); return MSDP_EINVAL; }
    if (h->blocked) return msdp_affine_get_block(h, row0, nb, S);
    HIPCHK(msdp_memcpy2d_async(S, (size_t)nb * sizeof(double), h->d.Sdual + (size_t)row0 * nS + row0, (size_t)nS * sizeof(double),
                            (size_t)nb * sizeof(double), (size_t)nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ measurement
static void algo_cost(msdp_handle h, double* bytes, double* flops) {
    const Dev& d = h->d;
    const double n = d.n_loc, p = d.p;
    if (d.costkind == COST_SPARSE) {
        // nnz*(8+4) + (n+1)*4 + 3*8*n*p + 8*n ; 2*nnz*p + 5*n*p
        *bytes = (double)d.nnz * 12.0 + (n + 1) * 4.0 + 24.0 * n * p + 8.0 * n;
        *flops = 2.0 * (double)d.nnz * p + 5.0 * n * p;
    } else if (d.costkind == COST_DENSE) {
        *bytes = 8.0 * n * (double)d.n + 24.0 * n * p;
        *flops = 2.0 * n * (double)d.n * p;
    } else {
        msdp_affine_algo_cost(h, bytes, flops);
    }
}

extern "C" int msdp_bench_hessvec(msdp_handle h, int32_t reps, double* avg_ms, double* algo_bytes, double* algo_flops) {
    CHECK_H(h);
    if (reps < 1 || !avg_ms) return MSDP_EINVAL;
    int rc = ensure_state(h);
    if (rc) return rc;
    // direction: the Riemannian gradient at the resident point
    HIPCHK(msdp_memcpy_async(h->d.md, h->d.Gr[host_cur(h)], (size_t)rows_capacity(h) * h->d.ld * sizeof(double),
                          hipMemcpyDeviceToDevice, h->stream));
    if ((rc = msdp_k_set_active(h, 1))) return rc;
    for (int i = 0; i < 3; ++i) if ((rc = msdp_launch_hess(h))) return rc;
    // replay a graph of 50 back-to-back launches so the host launch path is not what is timed
    const int per = 50;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    const bool graph = use_graphs(h);
    if (graph) {
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < per && !rc; ++i) rc = msdp_launch_hess(h);
        hipError_t e = hipStreamEndCapture(h->stream, &g);
        if (rc) return rc;
        if (e != hipSuccess) { msdp_set_error("graph capture failed: %s", hipGetErrorString(e)); return MSDP_EHIP; }
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPCHK(hipGraphLaunch(ge, h->stream));
    }
    const int nrep = (reps + per - 1) / per;
    reps = nrep * per;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < nrep; ++i) {
        if (graph) { HIPCHK(hipGraphLaunch(ge, h->stream)); }
        else for (int t = 0; t < per; ++t) if ((rc = msdp_launch_hess(h))) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = (double)ms / reps;
    if ((rc = msdp_k_set_active(h, 0))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    double b, f;
    algo_cost(h, &b, &f);
    if (algo_bytes) *algo_bytes = b;
    if (algo_flops) *algo_flops = f;
    return 0;
}

// Time ONE kernel of the tCG trip in isolation (graph of 50 back-to-back launches):
// which = 0 hess, 1 upd1, 2 upd2.  Exits are disabled (bench mode).
extern "C" int msdp_bench_kernel(msdp_handle h, int32_t which, int32_t reps, double* avg_ms) {
    CHECK_H(h);
    if (reps < 1 || !avg_ms || which < 0 || which > 2) return MSDP_EINVAL;
    int rc = ensure_state(h);
    if (rc) return rc;
    msdp_rtr_opts o;
    msdp_rtr_default_opts(&o);
    o.maxinner = 0x7ffffff0; o.maxiter = 1;
    fill_ctl(h, &o);
    h->h_ctl->bench_mode = 1;
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
    if ((rc = msdp_launch_rtr_begin(h))) return rc;
    if ((rc = msdp_launch_tcg_init(h))) return rc;
    for (int i = 0; i < 2; ++i)
        if ((rc = msdp_launch_hess(h)) || (rc = msdp_launch_upd1(h)) || (rc = msdp_launch_upd2(h))) return rc;
    if ((rc = msdp_launch_hess(h))) return rc;
    if (which == 2 && (rc = msdp_launch_upd1(h))) return rc;      // upd2 reads frame 1
    const int per = 50;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < per && !rc; ++i)
        rc = which == 0 ? msdp_launch_hess(h) : (which == 1 ? msdp_launch_upd1(h) : msdp_launch_upd2(h));
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc) return rc;
    if (e != hipSuccess) { msdp_set_error("graph capture failed: %s", hipGetErrorString(e)); return MSDP_EHIP; }
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphLaunch(ge, h->stream));
    const int nrep = (reps + per - 1) / per;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < nrep; ++i) HIPCHK(hipGraphLaunch(ge, h->stream));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = (double)ms / (nrep * per);
    h->h_ctl->bench_mode = 0;
    h->h_ctl->done = 0;
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_k_set_active(h, 0))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->state_valid = false;
    return 0;
}

int msdp_persist_trace_dims(msdp_handle h, int* G, int* nj, int* j0);        // msdp_persist.hip
// Measurement only: where the time of a persistent tCG trip goes.  Runs msdp_bench_tcg_trip(reps) on the traced instance of the
// persistent kernel and returns thread 0's s_memtime stamps of 7 phase boundaries (see msdp_persist.hip, TSTAMP) for every
// workgroup and the trips j0 .. j0 + nj - 1: out[((g * nj + t) * 8 + phase)], cap >= G * nj * 8 entries; dims = {G, nj, j0}.
extern "C" int msdp_bench_tcg_trip(msdp_handle h, int32_t reps, double* avg_ms);
extern "C" int msdp_debug_persist_trace(msdp_handle h, int32_t reps, uint64_t* out, int64_t cap, int32_t* dims, double* avg_ms) {
    CHECK_H(h);
    if (!out || !dims || !avg_ms) return MSDP_EINVAL;
    if (!msdp_persist_eligible(h)) { msdp_set_error("persist_trace: the persistent kernel does not apply to this handle"); return MSDP_EUNSUPPORTED; }
    int G = 0, nj = 0, j0 = 0;
    msdp_persist_trace_dims(h, &G, &nj, &j0);
    dims[0] = G; dims[1] = nj; dims[2] = j0;
    const bool fused = reps <= 0;                                  // the TR iterations of one trustregions() call in the fused launch
    const size_t cnt = (size_t)G * nj * 8 * (fused ? 2 : 1);       // (fused: + the trips of one TR iteration, msdp_pipe.h MSDP_TRACE_KSEL)
    if (fused) { j0 = 0; dims[2] = 0; }
    if (cap < (int64_t)cnt || (!fused && reps < j0 + nj)) { msdp_set_error("persist_trace: cap >= %zu entries and reps >= %d needed", cnt, j0 + nj); return MSDP_EINVAL; }
    if (!h->trace_buf) {
        void* p = nullptr;
        int rc = msdp_dev_alloc_bytes(h, &p, (size_t)2 * MSDP_MAX_GRID * nj * 8 * sizeof(unsigned long long));
        if (rc) return rc;
        h->trace_buf = (unsigned long long*)p;
    }
    HIPCHK(hipMemset(h->trace_buf, 0, cnt * sizeof(unsigned long long)));
    h->d.trace = h->trace_buf;
    int rc;
    if (fused) {
        // one call with the options of the handle's last msdp_rtr (the reference's inner-solver defaults before any): avg_ms = its time
        msdp_rtr_opts o = h->last_opts;
        if (o.maxinner < 1) { msdp_rtr_default_opts(&o); o.maxiter = 40; o.maxinner = 100; }
        if (!msdp_persist_fused_ok(h)) { h->d.trace = nullptr; msdp_set_error("persist_trace: the fused launch does not apply to this handle"); return MSDP_EUNSUPPORTED; }
        msdp_rtr_stats st;
        rc = msdp_rtr(h, &o, &st);
        *avg_ms = st.seconds * 1e3;
    } else rc = msdp_bench_tcg_trip(h, reps, avg_ms);
    h->d.trace = nullptr;
    if (rc) return rc;
    HIPCHK(msdp_memcpy(out, h->trace_buf, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int msdp_bench_tcg_trip(msdp_handle h, int32_t reps, double* avg_ms) {
    CHECK_H(h);
    if (reps < 1 || !avg_ms) return MSDP_EINVAL;
    int rc = ensure_state(h);
    if (rc) return rc;
    msdp_rtr_opts o;
    msdp_rtr_default_opts(&o);
    o.maxinner = 0x7ffffff0; o.maxiter = 1;
    fill_ctl(h, &o);
    h->h_ctl->bench_mode = 1;
    if (msdp_persist_eligible(h)) {
        // persistent kernel: `reps` trips with the exits disabled in one launch (run twice, time the second)
        h->h_ctl->maxinner = reps;
        if ((rc = push_ctl(h))) return rc;
        if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
        if ((rc = msdp_launch_rtr_begin(h))) return rc;
        h->d.status = nullptr;
        rc = msdp_launch_tcg_persist(h);
        if (!rc) {
            hipError_t e1 = hipEventRecord(h->ev0, h->stream);
            rc = msdp_launch_tcg_persist(h);
            hipError_t e2 = hipEventRecord(h->ev1, h->stream);
            hipError_t e3 = hipEventSynchronize(h->ev1);
            float ms = 0.f;
            hipError_t e4 = hipEventElapsedTime(&ms, h->ev0, h->ev1);
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) { msdp_set_error("bench events failed"); rc = MSDP_EHIP; }
            *avg_ms = (double)ms / reps;
        }
        {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, (void*)h->h_status, 0) == hipSuccess) h->d.status = (unsigned long long*)dp;
        }
        h->h_ctl->bench_mode = 0;
        h->h_ctl->done = 0;
        int rc2 = push_ctl(h);
        HIPCHK(hipStreamSynchronize(h->stream));
        h->state_valid = false;
        int perr = 0;
        HIPCHK(msdp_memcpy(&perr, h->psync_err, sizeof(int), hipMemcpyDeviceToHost));
        if (perr) { msdp_set_error("persistent tCG: grid synchronisation timed out"); return MSDP_EHIP; }
        return rc ? rc : rc2;
    }
    if (h->use_comm && h->lgroup && h->nranks > 1 && h->d.costkind == COST_SPARSE) {
        // in-process ranks: the cross-rank persistent tCG when every member can run it (`reps` trips, exits disabled, one launch per
        // member; run twice, time the second) -- every member calls this function together
        int agreed = 0;
        if ((rc = local_vote_min(h, msdp_xpersist_eligible(h, h->nranks), &agreed))) return rc;
        bool xuse = false;
        if (agreed && (rc = xr_begin(h, &xuse))) return rc;
        if (xuse) {
            h->h_ctl->maxinner = reps;
            if ((rc = push_ctl(h))) return rc;
            if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
            if ((rc = msdp_launch_rtr_begin(h))) return rc;
            h->d.status = nullptr;
            float ms = 0.f;
            for (int pass = 0; pass < 2 && !rc; ++pass) {
                if (pass && (rc = xr_begin(h, &xuse))) break;
                HIPCHK(hipEventRecord(h->ev0, h->stream));
                rc = xr_launch(h);
                HIPCHK(hipEventRecord(h->ev1, h->stream));
                HIPCHK(hipEventSynchronize(h->ev1));
                HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
                if (!rc) rc = xr_check(h);
                LOCAL_BARRIER(h->lgroup);
            }
            *avg_ms = (double)ms / reps;
            restore_status_ptr(h);
            h->h_ctl->bench_mode = 0;
            h->h_ctl->done = 0;
            int rc2 = push_ctl(h);
            HIPCHK(hipStreamSynchronize(h->stream));
            h->state_valid = false;
            h->xpersist_last = true;
            return rc ? rc : rc2;
        }
    }
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_launch_costgrad(h, host_cur(h)))) return rc;
    if ((rc = msdp_launch_rtr_begin(h))) return rc;
    h->h_ctl->done = 0;
    if ((rc = tcg_begin(h))) return rc;
    if ((rc = enqueue_trips(h, 2))) return rc;
    const int CH = TCG_CHUNK;
    const bool graph = use_graphs(h);
    if (graph && (rc = ensure_chunk_graph(h, CH))) return rc;
    const int nchunks = (reps + CH - 1) / CH;
    reps = nchunks * CH;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < nchunks; ++i) if ((rc = launch_chunk(h, CH, graph))) return rc;
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = (double)ms / reps;
    h->h_ctl->bench_mode = 0;
    h->h_ctl->done = 0;
    if ((rc = push_ctl(h))) return rc;
    if ((rc = msdp_k_set_active(h, 0))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->state_valid = false;
    return 0;
}
