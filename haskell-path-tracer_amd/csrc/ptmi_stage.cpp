// ptmi_stage.cpp -- see ptmi_stage.h.
#include "ptmi_stage.h"

#include <sys/mman.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

namespace ptmi {

namespace {

// one DMA / one worker memcpy; PTMI_STAGE_CHUNK_KB and PTMI_STAGE_PIECE_KB override (measurements)
size_t env_kb(const char *name, size_t fallback_kb)
{
    const char *e = std::getenv(name);
    const long v = e ? std::atol(e) : 0;
    return (size_t)(v >= 64 && v <= (1 << 20) ? v : (long)fallback_kb) << 10;
}
// Measured at 1080p (7 planes of 8.3 MB each way): 2-MB chunks 3.9 ms per render1, 8-MB chunks 2.9 ms (every DMA and
// its event cost ~15 us of host time); 4, 8 or 16 workers make no difference.
const size_t kChunkBytes = env_kb("PTMI_STAGE_CHUNK_KB", 8192);
const size_t kPieceBytes = env_kb("PTMI_STAGE_PIECE_KB", 1024);
constexpr size_t kRingAlign = 256;
constexpr size_t kMinPopulateBytes = 256u << 10;
constexpr int kSamplesPerSpan = 16;

int configured_threads()
{
    if (const char *e = std::getenv("PTMI_STAGE_THREADS")) {
        const int v = std::atoi(e);
        return v < 0 ? 0 : (v > 64 ? 64 : v);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw < 4) return 0;                       // nothing to gain over the runtime's own staging
    const unsigned half = hw / 2;
    return (int)(half < 8 ? half : 8);
}

size_t page_size()
{
    static const size_t ps = [] { const long v = sysconf(_SC_PAGESIZE); return v > 0 ? (size_t)v : (size_t)4096; }();
    return ps;
}

// true if one of a few evenly spaced pages of [p, p + bytes) is not resident
bool looks_unpopulated(const char *p, size_t bytes)
{
    const size_t ps = page_size();
    const uintptr_t first = (uintptr_t)p & ~(uintptr_t)(ps - 1);
    const size_t pages = ((uintptr_t)p + bytes - first + ps - 1) / ps;
    const size_t step = pages > (size_t)kSamplesPerSpan ? pages / kSamplesPerSpan : 1;
    unsigned char vec = 0;
    for (size_t k = 0; k < pages; k += step) {
        if (mincore((void *)(first + k * ps), ps, &vec) != 0) return false;     // cannot tell (e.g. unmapped): let the copy find out
        if (!(vec & 1)) return true;
    }
    if (mincore((void *)(first + (pages - 1) * ps), ps, &vec) == 0 && !(vec & 1)) return true;
    return false;
}

void populate(char *begin, size_t bytes, bool writable)
{
    const size_t ps = page_size();
    char *first = (char *)((uintptr_t)begin & ~(uintptr_t)(ps - 1));
    char *end = (char *)(((uintptr_t)begin + bytes + ps - 1) & ~(uintptr_t)(ps - 1));
    if (madvise(first, (size_t)(end - first), writable ? MADV_POPULATE_WRITE : MADV_POPULATE_READ) == 0) return;
    // Older kernel, or a mapping madvise refuses: touch one byte per page inside the caller's range (a destination is
    // about to be overwritten anyway; writing a byte's own value back keeps even that invisible).
    for (size_t off = 0; off < bytes; off += ps) {
        volatile char *v = begin + off;
        const char x = *v;
        if (writable) *v = x;
    }
    if (bytes) {
        volatile char *v = begin + bytes - 1;
        const char x = *v;
        if (writable) *v = x;
    }
}

}  // namespace

Stager::Stager() : n_threads_(configured_threads()) {}

Stager::~Stager()
{
    {
        std::lock_guard<std::mutex> lock(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
    if (ring_busy_ && ring_free_) (void)hipEventSynchronize(ring_free_);
    if (ring_) (void)hipHostFree(ring_);
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
    if (ring_free_) (void)hipEventDestroy(ring_free_);
}

void Stager::worker()
{
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> lock(mu_);
            cv_.wait(lock, [&] { return stop_ || !jobs_.empty(); });
            if (jobs_.empty()) return;          // stop_ and nothing left
            job = jobs_.front();
            jobs_.pop_front();
        }
        if (job.kind == kCopy) std::memcpy(job.dst, job.src, job.bytes);
        else populate(static_cast<char *>(job.dst), job.bytes, job.kind == kPopulateWrite);
        job.counter->fetch_sub(1, std::memory_order_acq_rel);
    }
}

// One chunk's page copy, cut into pieces so that several workers share it (the last chunk's copy is the tail of the
// whole transfer).  `counter` must already hold the number of pieces.
void Stager::submit_copy(void *dst, const void *src, size_t bytes, std::atomic<int> *counter)
{
    {
        std::lock_guard<std::mutex> lock(mu_);
        for (size_t off = 0; off < bytes; off += kPieceBytes) {
            const size_t len = bytes - off < kPieceBytes ? bytes - off : kPieceBytes;
            jobs_.push_back(Job{kCopy, static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len, counter});
        }
    }
    cv_.notify_all();
}

void Stager::wait_zero(std::atomic<int> &counter)
{
    for (int spin = 0; counter.load(std::memory_order_acquire) > 0; ++spin) {
        if (spin < 200) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

void Stager::populate_missing(const CopySpan *spans, int n, bool writable)
{
    std::vector<Job> todo;
    for (int s = 0; s < n; ++s) {
        char *p = static_cast<char *>(spans[s].host);
        if (!p || spans[s].bytes < kMinPopulateBytes || !looks_unpopulated(p, spans[s].bytes)) continue;
        for (size_t off = 0; off < spans[s].bytes; off += kChunkBytes) {
            const size_t len = spans[s].bytes - off < kChunkBytes ? spans[s].bytes - off : kChunkBytes;
            todo.push_back(Job{writable ? kPopulateWrite : kPopulateRead, p + off, nullptr, len, &populate_pending_});
        }
    }
    if (todo.empty()) return;
    populate_pending_.store((int)todo.size(), std::memory_order_release);
    {
        std::lock_guard<std::mutex> lock(mu_);
        for (const Job &j : todo) jobs_.push_back(j);
    }
    cv_.notify_all();
    wait_zero(populate_pending_);
}

hipError_t Stager::prepare(const CopySpan *spans, int n, bool host_is_destination, std::vector<Chunk> &chunks)
{
    if (workers_.empty()) {
        for (int i = 0; i < n_threads_; ++i) workers_.emplace_back([this] { worker(); });
    }
    if (!ring_free_) {
        const hipError_t e = hipEventCreateWithFlags(&ring_free_, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    if (ring_busy_) {                           // an earlier to_device may still be reading the ring
        const hipError_t e = hipEventSynchronize(ring_free_);
        if (e != hipSuccess) return e;
        ring_busy_ = false;
    }
    size_t need = 0;
    chunks.clear();
    for (int s = 0; s < n; ++s) {
        for (size_t off = 0; off < spans[s].bytes; off += kChunkBytes) {
            const size_t len = spans[s].bytes - off < kChunkBytes ? spans[s].bytes - off : kChunkBytes;
            chunks.push_back(Chunk{(size_t)s, off, len, need});
            need += (len + kRingAlign - 1) / kRingAlign * kRingAlign;
        }
    }
    if (need > ring_bytes_) {
        if (ring_) { (void)hipHostFree(ring_); ring_ = nullptr; ring_bytes_ = 0; }
        void *p = nullptr;
        const hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        ring_ = static_cast<char *>(p);
        ring_bytes_ = need;
    }
    if (pending_.size() < chunks.size()) {
        std::vector<std::atomic<int>> bigger(chunks.size());
        pending_.swap(bigger);
    }
    for (size_t k = 0; k < chunks.size(); ++k)
        pending_[k].store((int)((chunks[k].bytes + kPieceBytes - 1) / kPieceBytes), std::memory_order_relaxed);
    populate_missing(spans, n, host_is_destination);
    return hipSuccess;
}

hipError_t Stager::to_device(const CopySpan *spans, int n, hipStream_t stream)
{
    std::vector<Chunk> chunks;
    hipError_t e = prepare(spans, n, false, chunks);
    if (e != hipSuccess) return e;
    for (size_t k = 0; k < chunks.size(); ++k) {
        const Chunk &c = chunks[k];
        submit_copy(ring_ + c.ring, static_cast<const char *>(spans[c.span].host) + c.offset, c.bytes, &pending_[k]);
    }
    for (size_t k = 0; k < chunks.size(); ++k) {
        const Chunk &c = chunks[k];
        wait_zero(pending_[k]);                 // also on the error path: no worker may outlive the call
        if (e == hipSuccess)
            e = hipMemcpyAsync(static_cast<char *>(spans[c.span].dev) + c.offset, ring_ + c.ring, c.bytes,
                               hipMemcpyHostToDevice, stream);
    }
    if (e == hipSuccess) e = hipEventRecord(ring_free_, stream);
    if (e == hipSuccess) ring_busy_ = true;
    else (void)hipStreamSynchronize(stream);    // whatever was enqueued must stop reading the ring before it is reused
    return e;
}

hipError_t Stager::to_host(const CopySpan *spans, int n, hipStream_t stream)
{
    std::vector<Chunk> chunks;
    hipError_t e = prepare(spans, n, true, chunks);
    if (e != hipSuccess) return e;
    while (events_.size() < chunks.size()) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        events_.push_back(ev);
    }
    for (size_t k = 0; k < chunks.size() && e == hipSuccess; ++k) {
        const Chunk &c = chunks[k];
        e = hipMemcpyAsync(ring_ + c.ring, static_cast<const char *>(spans[c.span].dev) + c.offset, c.bytes,
                           hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(events_[k], stream);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream);
        return e;
    }
    size_t submitted = 0;
    for (; submitted < chunks.size(); ++submitted) {
        const Chunk &c = chunks[submitted];
        e = hipEventSynchronize(events_[submitted]);
        if (e != hipSuccess) break;
        submit_copy(static_cast<char *>(spans[c.span].host) + c.offset, ring_ + c.ring, c.bytes, &pending_[submitted]);
    }
    for (size_t k = 0; k < submitted; ++k) wait_zero(pending_[k]);
    if (e != hipSuccess) (void)hipStreamSynchronize(stream);
    return e;
}

}  // namespace ptmi
