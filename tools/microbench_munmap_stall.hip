// Does unmapping host memory the GPU never saw stall the next GPU operation of the process?  (round 6: the second G81 solve of a process
// waits 20 - 35 ms in its first host <-> device copy after NumPy has freed an 8-MB array, tools/kkt_stall_bisect.py.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/build/microbench_munmap_stall tools/microbench_munmap_stall.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(double* p) { p[threadIdx.x] += 1.0; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;             // bit 0: fine-grained device allocation live; bit 1: mapped pinned host block live
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* d = nullptr; hipMalloc(&d, 64 << 20);
    char* pin = nullptr; hipHostMalloc((void**)&pin, 16 << 20);
    void* fg = nullptr; if (mode & 1) hipExtMallocWithFlags(&fg, 64 << 20, hipDeviceMallocFinegrained);
    void* hm = nullptr; if (mode & 2) hipHostMalloc(&hm, 4096, hipHostMallocMapped);
    auto probe = [&](const char* what) {
        double t0 = now();
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        double t1 = now();
        hipMemcpyAsync(d, pin, 8 << 20, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t2 = now();
        printf("%-44s kernel + sync %.3f ms, 8-MB pinned copy + sync %.3f ms\n", what, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
    };
    probe("cold"); probe("warm"); probe("warm");
    for (size_t mb : {1, 8, 64}) {
        char lab[128];
        void* p = mmap(nullptr, mb << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        munmap(p, mb << 20);
        snprintf(lab, sizeof lab, "mmap + munmap %zu MB, untouched", mb); probe(lab); probe("  again");
        p = mmap(nullptr, mb << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        memset(p, 1, mb << 20);
        snprintf(lab, sizeof lab, "mmap + touch %zu MB (still mapped)", mb); probe(lab);
        munmap(p, mb << 20);
        snprintf(lab, sizeof lab, "munmap of the touched %zu MB", mb); probe(lab); probe("  again");
        p = malloc(mb << 20); memset(p, 1, mb << 20); free(p);
        snprintf(lab, sizeof lab, "malloc + touch + free %zu MB", mb); probe(lab); probe("  again");
    }
    return 0;
}
