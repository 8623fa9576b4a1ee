// ptmi_stage.h -- host <-> device transfers for the boundary's host-buffer entry points.
//
// The reference hands `runN` ordinary (pageable) host arrays and gets FRESH ones back (app/Main.hs:190-191,
// `A.toVectors` at :350), so the compat entry ptmi_render1 moves 56 B per pixel over PCIe per call.  Measured on
// the MI355X box (tools/measure_host_copies.py, 1080p = 58 MB each way):
//   * plain hipMemcpyAsync on pageable pages pins the caller's pages in place: 2.6 ms per call when the pages are
//     resident and the process's mappings are quiet, but 25-30 ms per call -- whatever the size -- as soon as the
//     caller frees / reallocates the arrays it passed before (fresh output arrays per call: exactly the
//     reference's pattern), because every change of a mapping that was pinned costs a driver-side re-validation;
//   * the Stager never lets the driver see the caller's pages: a pinned ring, worker threads that copy 1-MB
//     pieces between the caller's pages and the ring, one DMA per 8-MB chunk issued in order on the context's
//     stream (2.9 ms per 1080p call, steady, against 2.6 ms for the plain copies at their best).  Pages the caller has never touched are populated first, in parallel (madvise
//     MADV_POPULATE_WRITE / _READ; touching as the fallback), instead of faulting one by one inside memcpy.
// Bytes and order of device operations are those of the plain copies.  PTMI_STAGE_THREADS=0 switches it off.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace ptmi {

struct CopySpan {
    void *dev;
    void *host;       // source for to_device (not written), destination for to_host
    size_t bytes;
};

class Stager {
public:
    Stager();
    ~Stager();
    Stager(const Stager &) = delete;
    Stager &operator=(const Stager &) = delete;

    // Worker threads in use (0 = the engine is off and callers use plain copies).
    int threads() const { return n_threads_; }
    // Smallest transfer worth the hand-over to the workers.
    static constexpr size_t kMinBytes = 1u << 20;

    // Host -> device.  Returns once every chunk's DMA has been enqueued on `stream` (the caller's pages have been
    // read by then; the DMAs complete in stream order before anything enqueued afterwards).
    hipError_t to_device(const CopySpan *spans, int n, hipStream_t stream);
    // Device -> host.  Returns once the caller's buffers hold the data.
    hipError_t to_host(const CopySpan *spans, int n, hipStream_t stream);

private:
    enum Kind { kCopy, kPopulateRead, kPopulateWrite };
    struct Job { Kind kind; void *dst; const void *src; size_t bytes; std::atomic<int> *counter; };
    struct Chunk { size_t span, offset, bytes, ring; };

    hipError_t prepare(const CopySpan *spans, int n, bool host_is_destination, std::vector<Chunk> &chunks);
    void populate_missing(const CopySpan *spans, int n, bool writable);
    void submit_copy(void *dst, const void *src, size_t bytes, std::atomic<int> *counter);
    void worker();
    static void wait_zero(std::atomic<int> &counter);

    int n_threads_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> jobs_;
    bool stop_ = false;

    char *ring_ = nullptr;                 // pinned
    size_t ring_bytes_ = 0;
    std::vector<hipEvent_t> events_;
    hipEvent_t ring_free_ = nullptr;       // recorded after the last DMA that READS the ring (to_device)
    bool ring_busy_ = false;
    std::vector<std::atomic<int>> pending_;   // per chunk: pieces still to copy
    std::atomic<int> populate_pending_{0};
};

}  // namespace ptmi
