// Host <-> device copies of CALLER memory (round 6).
//
// No pointer the caller hands over ever reaches the HIP runtime: every host <-> device copy of msdp_*.hip goes through
// msdp_memcpy* below, and those move unpinned host memory through a pinned staging buffer of the process.  Why: the runtime
// registers the pages of a pageable range it is asked to copy to or from -- of ANY size -- and when the owner of those pages gives
// them back to the kernel later (a NumPy array or a MATLAB mxArray is freed: the allocator trims its heap or unmaps the block),
// the NEXT host <-> device operation of the process stalls for 20 - 35 ms with the device idle.  Measured on the G81 solve to KKT
// 1e-8 (n = 20 000, p0 = 40; tools/kkt_stall_variants.py, the second solve of a process, six runs each):
//     every copy handed to the runtime as it is (MSDP_XFER_DIRECT_MAX=100000000000)   0.171 - 0.182 s, every run stalls
//     copies above 16 KB staged (MSDP_XFER_DIRECT_MAX=16384)                          0.160 - 0.162 s, two runs of six stall (0.179, 0.194)
//     every copy of unpinned memory staged (the default, 0)                           0.159 - 0.163 s, none stalls
// and 0.159 s when the caller keeps every array of the first solve alive, staged or not -- the stall belongs to the freed pages, not to
// the copy (tools/kkt_stall_bisect.py: a small second handle's upload takes 35 ms instead of 0.15 right after NumPy has freed an
// 8-MB array the library never saw, once the first solve's result arrays are gone).  A plain HIP program that maps, touches and
// unmaps host memory beside its copies shows nothing (tools/microbench_munmap_stall.hip): it takes a registered range.
// A memcpy into pinned memory costs 0.15 ms per megabyte and overlaps with the DMA of the chunk before.
//
// Semantics: msdp_memcpy is hipMemcpy; msdp_memcpy_async is hipMemcpyAsync except that a copy which touches unpinned host memory
// has COMPLETED when the call returns (the runtime's own pageable path gives no weaker guarantee that callers here rely on).
// Device <-> device copies and copies whose host side is already pinned (hipHostMalloc: the handle's own control blocks) pass
// straight through.
#include "msdp_common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static size_t XFER_DIRECT_MAX = 0;                            // MSDP_XFER_DIRECT_MAX (bytes): copies up to this size go to the runtime as they are (A/B runs)
static const size_t XFER_HALF = (size_t)16 << 20;            // two halves: the memcpy of one chunk runs beside the DMA of the other
static std::mutex g_xfer_mutex;
static char* g_xfer_buf = nullptr;                            // 2 * XFER_HALF bytes of pinned memory, allocated on first use
static hipEvent_t g_xfer_ev[2] = {nullptr, nullptr};
static int g_xfer_dev = -1;

void msdp_xfer_release() {                         // msdp_release_cache
    std::lock_guard<std::mutex> lk(g_xfer_mutex);
    if (g_xfer_buf) (void)hipHostFree(g_xfer_buf);
    g_xfer_buf = nullptr;
    for (auto& e : g_xfer_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    g_xfer_dev = -1;
}

static bool xfer_ready_locked() {
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (g_xfer_buf && dev == g_xfer_dev) return true;
    if (g_xfer_buf) { (void)hipHostFree(g_xfer_buf); g_xfer_buf = nullptr; }
    for (auto& e : g_xfer_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (hipHostMalloc((void**)&g_xfer_buf, 2 * XFER_HALF) != hipSuccess) { (void)hipGetLastError(); g_xfer_buf = nullptr; return false; }
    for (auto& e : g_xfer_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(g_xfer_buf); g_xfer_buf = nullptr; return false; }
    g_xfer_dev = dev;
    return true;
}

static bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // memory the runtime does not know
    return a.type == hipMemoryTypeHost;
}

// rows x width bytes; `hp` / `dp`: pitches of the host and of the device side (a 1-D copy is one row)
static hipError_t staged(void* dev, size_t dp, void* host, size_t hp, size_t width, size_t rows, bool to_device, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_xfer_mutex);
    if (!xfer_ready_locked()) return hipErrorOutOfMemory;
    hipError_t e = hipSuccess;
    // a chunk: a run of whole rows, or a piece of one row when a row exceeds half the buffer
    struct Chunk { size_t row, off, nrows, nbytes; };
    std::vector<Chunk> chunks;
    if (width > XFER_HALF) {
        for (size_t r = 0; r < rows; ++r)
            for (size_t off = 0; off < width; off += XFER_HALF) chunks.push_back({r, off, 1, std::min(XFER_HALF, width - off)});
    } else {
        const size_t nr = std::max<size_t>(1, XFER_HALF / width);
        for (size_t r = 0; r < rows; r += nr) chunks.push_back({r, 0, std::min(nr, rows - r), width});
    }
    auto dev_at = [&](const Chunk& c) { return (char*)dev + c.row * dp + c.off; };
    auto host_at = [&](const Chunk& c) { return (char*)host + c.row * hp + c.off; };
    auto copy_dev = [&](const Chunk& c, char* pin) -> hipError_t {     // pinned <-> device for one chunk, on s
        const hipMemcpyKind k = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        if (c.nrows == 1 || dp == c.nbytes)
            return to_device ? hipMemcpyAsync(dev_at(c), pin, c.nbytes * c.nrows, k, s) : hipMemcpyAsync(pin, dev_at(c), c.nbytes * c.nrows, k, s);
        return to_device ? hipMemcpy2DAsync(dev_at(c), dp, pin, c.nbytes, c.nbytes, c.nrows, k, s)
                         : hipMemcpy2DAsync(pin, c.nbytes, dev_at(c), dp, c.nbytes, c.nrows, k, s);
    };
    auto copy_host = [&](const Chunk& c, char* pin) {                  // pinned <-> caller memory for one chunk
        if (c.nrows == 1 || hp == c.nbytes) {
            if (to_device) memcpy(pin, host_at(c), c.nbytes * c.nrows); else memcpy(host_at(c), pin, c.nbytes * c.nrows);
            return;
        }
        for (size_t r = 0; r < c.nrows; ++r) {
            if (to_device) memcpy(pin + r * c.nbytes, host_at(c) + r * hp, c.nbytes); else memcpy(host_at(c) + r * hp, pin + r * c.nbytes, c.nbytes);
        }
    };
    const int nc = (int)chunks.size();
    for (int i = 0; i < nc; ++i) {
        const Chunk& c = chunks[i];
        char* pin = g_xfer_buf + (size_t)(i & 1) * XFER_HALF;
        if (to_device) {
            if (i >= 2 && (e = hipEventSynchronize(g_xfer_ev[i & 1])) != hipSuccess) return e;   // the DMA that last read this half
            copy_host(c, pin);
            if ((e = copy_dev(c, pin)) != hipSuccess) return e;
            if ((e = hipEventRecord(g_xfer_ev[i & 1], s)) != hipSuccess) return e;
        } else {
            if ((e = copy_dev(c, pin)) != hipSuccess) return e;
            if ((e = hipEventRecord(g_xfer_ev[i & 1], s)) != hipSuccess) return e;
            if (i >= 1) {                                      // drain the chunk before while this one is in flight
                if ((e = hipEventSynchronize(g_xfer_ev[(i - 1) & 1])) != hipSuccess) return e;
                copy_host(chunks[i - 1], g_xfer_buf + (size_t)((i - 1) & 1) * XFER_HALF);
            }
        }
    }
    if (!to_device && nc > 0) {
        if ((e = hipEventSynchronize(g_xfer_ev[(nc - 1) & 1])) != hipSuccess) return e;
        copy_host(chunks[nc - 1], g_xfer_buf + (size_t)((nc - 1) & 1) * XFER_HALF);
        return hipSuccess;
    }
    return hipStreamSynchronize(s);                            // the staging halves are free again when the lock goes
}

static bool wants_staging(const void* host, size_t bytes, hipMemcpyKind kind) {
    static const bool env_read = [] { const char* e = getenv("MSDP_XFER_DIRECT_MAX"); if (e && *e) XFER_DIRECT_MAX = (size_t)strtoull(e, nullptr, 10); return true; }();
    (void)env_read;
    if (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost) return false;
    if (bytes <= XFER_DIRECT_MAX) return false;
    return !host_is_pinned(host);
}

hipError_t msdp_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    const bool h2d = kind == hipMemcpyHostToDevice;
    if (!wants_staging(h2d ? src : dst, bytes, kind)) return hipMemcpyAsync(dst, src, bytes, kind, s);
    return h2d ? staged(dst, bytes, const_cast<void*>(src), bytes, bytes, 1, true, s) : staged(const_cast<void*>(src), bytes, dst, bytes, bytes, 1, false, s);
}
hipError_t msdp_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    const bool h2d = kind == hipMemcpyHostToDevice;
    if (!wants_staging(h2d ? src : dst, bytes, kind)) return hipMemcpy(dst, src, bytes, kind);
    // hipMemcpy orders itself behind the null stream: so does this
    return h2d ? staged(dst, bytes, const_cast<void*>(src), bytes, bytes, 1, true, nullptr) : staged(const_cast<void*>(src), bytes, dst, bytes, bytes, 1, false, nullptr);
}
hipError_t msdp_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
    const bool h2d = kind == hipMemcpyHostToDevice;
    if (!wants_staging(h2d ? src : dst, width * height, kind)) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
    if (width == 0 || height == 0) return hipSuccess;
    return h2d ? staged(dst, dpitch, const_cast<void*>(src), spitch, width, height, true, s) : staged(const_cast<void*>(src), spitch, dst, dpitch, width, height, false, s);
}
hipError_t msdp_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind) {
    const bool h2d = kind == hipMemcpyHostToDevice;
    if (!wants_staging(h2d ? src : dst, width * height, kind)) return hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
    if (width == 0 || height == 0) return hipSuccess;
    return h2d ? staged(dst, dpitch, const_cast<void*>(src), spitch, width, height, true, nullptr) : staged(const_cast<void*>(src), spitch, dst, dpitch, width, height, false, nullptr);
}
