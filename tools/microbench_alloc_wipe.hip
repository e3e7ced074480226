// What does the first use of memory cost that the process freed a moment ago?  (round 6: the second G81 solve of a process waits
// 17 - 29 ms in the first hipStreamSynchronize after msdp_alloc_vectors, with the device idle.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/build/microbench_alloc_wipe tools/microbench_alloc_wipe.hip && tools/build/microbench_alloc_wipe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t MB = 1 << 20;
    for (int rep = 0; rep < 4; ++rep) {
        for (size_t freed_mb : {0, 64, 256, 512, 1024}) {
            void* big = nullptr;
            if (freed_mb) {
                hipMalloc(&big, freed_mb * MB);
                hipMemsetAsync(big, 1, freed_mb * MB, s);
                hipStreamSynchronize(s);
                double t0 = now();
                hipFree(big);
                double t1 = now();
                printf("rep %d: hipFree of %4zu MB %.3f ms; ", rep, freed_mb, 1e3 * (t1 - t0));
            } else printf("rep %d: nothing freed;           ", rep);
            void* p = nullptr;
            double t0 = now();
            hipMalloc(&p, 128 * MB);
            double t1 = now();
            hipMemsetAsync(p, 0, 128 * MB, s);
            double t2 = now();
            hipStreamSynchronize(s);
            double t3 = now();
            hipMemsetAsync(p, 0, 128 * MB, s);
            hipStreamSynchronize(s);
            double t4 = now();
            printf("hipMalloc 128 MB %.3f ms, memset enqueue %.3f, first sync %.3f ms, second memset+sync %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2),
                   1e3 * (t4 - t3));
            hipFree(p);
        }
    }
    return 0;
}
