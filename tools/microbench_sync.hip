// Grid-synchronisation latency microbenchmark (MI355X, 256 workgroups x 512 threads, one per CU).
// variant 0: sentinel slots, wave 0 polls all G slots (msdp_persist.hip scheme)
// variant 1: one atomic counter (fetch_add + poll)
// variant 2: slots replicated 8x, workgroup b polls replica b & 7 (fewer pollers per line)
// variant 3: variant 0 with an s_sleep between polls
// xcd mode (argv[1] = xcd): variant 0 with only the workgroups blockIdx % 8 == 0 of an 8x larger launch taking part, i.e. all
//   participants on ONE XCD (round-robin dispatch) behind one L2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define SENT 0xFFF8DEADBEEF0001ULL
#define MAXG 512
__device__ __forceinline__ void load4(const unsigned long long* p, unsigned long long (&b)[4]) {
    asm volatile(
        "global_load_dwordx2 %0, %4, off sc1\n\t"
        "global_load_dwordx2 %1, %4, off offset:512 sc1\n\t"
        "global_load_dwordx2 %2, %4, off offset:1024 sc1\n\t"
        "global_load_dwordx2 %3, %4, off offset:1536 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(p) : "memory");
}
template <int VAR, int REPL, int STRIDE = 1>
__global__ __launch_bounds__(512) void k(unsigned long long* slots, unsigned long long* cnt, int N, double* out) {
    __shared__ double sh[2];
    if (STRIDE > 1 && blockIdx.x % STRIDE) return;
    const int G = gridDim.x / STRIDE, lane = threadIdx.x & 63, bid = blockIdx.x / STRIDE;
    double acc = 0.0;
    for (int gen = 0; gen < N; ++gen) {
        __syncthreads();
        if (threadIdx.x < 64) {
            if (VAR == 4) {
                // 8 counters (one 64-B line apart), workgroup b adds to counter b & 7, everybody polls all 8
                if (lane == 0) __hip_atomic_fetch_add(cnt + 8 * (blockIdx.x & 7), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long want = (unsigned long long)(gen + 1) * (G / 8);
                for (;;) {
                    unsigned long long v = want;
                    if (lane < 8) v = __hip_atomic_load(cnt + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(v < want) == 0ULL) break;
                }
                if (lane == 0) sh[0] = 1.0;
            } else if (VAR == 1) {
                if (lane == 0) {
                    __hip_atomic_fetch_add(cnt, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long want = (unsigned long long)(gen + 1) * G;
                    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {}
                    sh[0] = 1.0;
                }
            } else {
                const int REP = (VAR == 2) ? REPL : 1;
                unsigned long long* base = slots + (size_t)(gen % 3) * REP * MAXG;
                const double mine = 1.0 + bid;
                if (VAR == 2) { if (lane < REPL) __hip_atomic_store(base + lane * MAXG + bid, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else if (lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(base + bid, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                const unsigned long long* p0 = base + ((VAR == 2) ? (bid % REPL) * MAXG : 0) + lane;
                double s;
                for (;;) {
                    unsigned long long b[4];
                    load4(p0, b);
                    bool ok = true; s = 0.0;
                    for (int q = 0; q < 4; ++q) if (lane + 64 * q < G) { ok = ok && b[q] != SENT; s += __longlong_as_double((long long)b[q]); }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
                    if (VAR == 3) __builtin_amdgcn_s_sleep(2);
                }
                if (lane == 0) sh[0] = s;
                unsigned long long* prev = slots + (size_t)((gen + 2) % 3) * REP * MAXG;
                if (VAR == 2) { if (lane < REPL) __hip_atomic_store(prev + lane * MAXG + bid, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else if (lane == 0) __hip_atomic_store(prev + bid, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        acc += sh[0];
    }
    if (threadIdx.x == 0) out[bid] = acc;
}
__global__ void fill(unsigned long long* s, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s[i] = SENT; }
static int g_alloc_mode = 0;   // 0 hipMalloc, 1 fine-grained, 2 uncached
template <int VAR, int REPL = 8, int STRIDE = 1> void run(int G, int N) {
    unsigned long long* slots; unsigned long long* cnt; double* out;
    const size_t ns = 3 * 64 * MAXG;
    if (g_alloc_mode == 0) hipMalloc(&slots, ns * 8); else hipExtMallocWithFlags((void**)&slots, ns * 8, g_alloc_mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached); hipMalloc(&cnt, 1024); hipMalloc(&out, MAXG * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        fill<<<64, 256>>>(slots, ns); hipMemset(cnt, 0, 1024);
        hipEventRecord(e0);
        k<VAR, REPL, STRIDE><<<G * STRIDE, 512>>>(slots, cnt, N, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("alloc %d variant %d repl %d stride %d G=%d: %.3f us per sync\n", g_alloc_mode, VAR, REPL, STRIDE, G, ms * 1e3 / N);
    }
}
int main(int argc, char** argv) {
    const int N = 2000;
    if (argc > 1) {          // single-XCD participants against the same count spread over all XCDs
        for (int mode : {0, 2}) { g_alloc_mode = mode; for (int G : {16, 20, 32, 40}) { run<0, 8, 8>(G, N); run<0, 8, 1>(G, N); run<2, 8, 1>(G, N); } }
        return 0;
    }
    for (int mode : {0, 1, 2}) { g_alloc_mode = mode; for (int G : {256, 64}) { run<0>(G, N); run<2, 8>(G, N); run<4>(G, N); } }
    return 0;
}
