// ipc_probe.hip -- round 5: can two PROCESSES on one GPU run one grid-synchronised computation through IPC-shared device memory?
//   (a) hipIpcGetMemHandle / hipIpcOpenMemHandle on plain and on uncached (hipDeviceMallocUncached) allocations,
//   (b) do the launches of two processes co-run (128 workgroups x 512 threads + 100 KB of LDS each: one per CU),
//   (c) what a barrier across both launches costs (counters in the shared block, sc1 / agent-scope accesses, bounded spins).
// usage: ipc_probe owner <mode> <file> <iters>   |   ipc_probe peer <file> <iters>        mode 0 hipMalloc, 1 uncached
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <chrono>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(_e), __LINE__); exit(3); } } while (0)

// region: [0..7] barrier counters 64 B apart (8 x 8 words), [64] error flag, [65] arrivals of launches, [128..] per-process stamps
__global__ __launch_bounds__(512) void k_barriers(unsigned long long* sh, int role, int G, int Gtot, int iters, unsigned long long* out) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = 0.0;
    const int bid = role * G + blockIdx.x;
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8 && threadIdx.x == 0) t0 = wall_clock64();
        __syncthreads();
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (lane == 0) __hip_atomic_fetch_add(sh + 8 * (bid & 7), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long want = (unsigned long long)(it + 1) * (unsigned)(Gtot / 8);
            long spins = 0;
            for (;;) {
                unsigned long long v = want;
                if (lane < 8) v = __hip_atomic_load(sh + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v < want) == 0ULL) break;
                if (++spins > (it == 0 ? (1L << 24) : (1L << 21)) || ((spins & 1023) == 0 && __hip_atomic_load(sh + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (lane == 0) { fail = 1; __hip_atomic_store(sh + 64, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    break;
                }
            }
        }
        __syncthreads();
        if (fail) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 0xDEADULL; return; }
    }
    if (threadIdx.x == 0) t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = 1; out[1] = t1 - t0; }
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const bool owner = !strcmp(argv[1], "owner");
    const int mode = owner ? atoi(argv[2]) : 0;
    const char* file = owner ? argv[3] : argv[2];
    const int iters = atoi(owner ? argv[4] : argv[3]);
    const size_t bytes = 1 << 20;
    unsigned long long* sh = nullptr;
    if (owner) {
        if (mode == 1) CK(hipExtMallocWithFlags((void**)&sh, bytes, hipDeviceMallocUncached));
        else if (mode == 2) CK(hipExtMallocWithFlags((void**)&sh, bytes, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void**)&sh, bytes));
        CK(hipMemset(sh, 0, bytes));
        CK(hipDeviceSynchronize());
        hipIpcMemHandle_t hd;
        hipError_t e = hipIpcGetMemHandle(&hd, sh);
        if (e != hipSuccess) { printf("owner: hipIpcGetMemHandle (mode %d) FAILED: %s\n", mode, hipGetErrorString(e)); FILE* f = fopen(file, "wb"); fputc('X', f); fclose(f); return 4; }
        char tmp[512]; snprintf(tmp, sizeof tmp, "%s.tmp", file);
        FILE* f = fopen(tmp, "wb"); fwrite(&hd, sizeof hd, 1, f); fclose(f); rename(tmp, file);
        printf("owner: mode %d exported\n", mode);
    } else {
        hipIpcMemHandle_t hd;
        for (int w = 0; w < 600; ++w) { if (access(file, R_OK) == 0) break; usleep(100000); }
        FILE* f = fopen(file, "rb"); if (!f) { printf("peer: no handle file\n"); return 4; }
        if (fread(&hd, sizeof hd, 1, f) != 1) { printf("peer: owner could not export\n"); return 4; }
        fclose(f);
        hipError_t e = hipIpcOpenMemHandle((void**)&sh, hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { printf("peer: hipIpcOpenMemHandle FAILED: %s\n", hipGetErrorString(e)); return 4; }
        printf("peer: opened\n");
    }
    fflush(stdout);
    unsigned long long* out = nullptr;
    CK(hipMalloc((void**)&out, 64)); CK(hipMemset(out, 0, 64));
    const int G = 128, lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void*)k_barriers, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const auto ta = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_barriers, dim3(G), dim3(512), lds, 0, sh, owner ? 0 : 1, G, 2 * G, iters, out);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
    unsigned long long res[2] = {0, 0};
    CK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
    if (res[0] == 1) printf("%s: %d barriers across two launches of 128 workgroups: %.3f us each (wall_clock64 100 MHz), launch wall %.3f s\n", owner ? "owner" : "peer", iters - 8, (double)res[1] * 10.0 / 1000.0 / (iters - 8), sec);
    else printf("%s: barrier TIMED OUT (the two launches did not run together) after %.2f s\n", owner ? "owner" : "peer", sec);
    if (!owner) (void)hipIpcCloseMemHandle(sh);
    return res[0] == 1 ? 0 : 5;
}
