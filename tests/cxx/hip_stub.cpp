// hip_stub.cpp -- a TEST-ONLY stand-in for the part of the HIP runtime libptmi's HOST side calls (tests/test_host_sanitized.py preloads it
// into a SUBPROCESS, in front of libamdhip64).  It exists so that the host logic of the C ABI -- contexts, options, the chained closure's
// tokens / slots / evictions, partitions, groups, the stream form's bookkeeping, every error path -- can run in this container, which has
// no GPU, under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool, CPU ones are).
//
// What it is: "device memory" is host memory (calloc: a fresh block reads as zeros), copies are memcpy, memsets are memset, streams and
// events are small heap objects and everything is synchronous -- or, after hipstub_set_deferred(1), as asynchronous as the real thing: see Op.  KERNELS DO NOT RUN: hipLaunchKernel counts the launch under the kernel's
// name and returns success.  Whatever a real kernel would have written stays as it was (zeros in a fresh block) -- except the few counter
// words a test places into a read-back with hipstub_poke, to send the host down the paths that depend on what a kernel counted.
//
// What a test on it validates: memory safety and error handling of the host code (a copy past the end of a block is a heap overflow ASan
// sees; a block freed twice or used after hipFree likewise; a block never freed shows in hipstub_live_blocks), and the SEQUENCE of runtime
// calls.  What it does NOT validate: any rendered value, any kernel, any timing, the HIP runtime itself.  The product never loads it: it is
// reachable through LD_PRELOAD in the test's child process only, and libptmi has no CPU path (ptmi_create fails without a device).
//
// Fault injection: hipstub_fail(kind, k) makes the k-th call (1-based, counted from now) of one kind fail once (hipstub_fail_run: and the
// `more` calls of the kind after it) -- 0 hipMalloc
// (hipErrorOutOfMemory), 1 hipMemcpy / hipMemcpyAsync, 2 hipLaunchKernel, 3 hipStreamSynchronize, 4 hipHostMalloc, 5 event / stream creation
// (hipErrorUnknown) -- so that a test can walk every failure point of a scenario and check that the context is still destroyable, nothing
// leaks and nothing is touched after being freed.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <vector>

extern "C" void __sanitizer_print_stack_trace(void);

namespace {

constexpr int kStreamMagic = 0x57e4a3, kEventMagic = 0xe7e47;
struct Stream;
struct Event { int magic; std::chrono::steady_clock::time_point t; Stream *recorded_on = nullptr; bool pending = false; };
// DEFERRED mode (hipstub_set_deferred): what a real stream does LATER is done later here too -- at the next synchronisation that covers it --
// so that host code which reads a result, reuses a pinned buffer or returns a borrowed one before it has synchronised meets stale bytes
// (the tests compare round trips of copies) or, if the memory has gone in between, AddressSanitizer.  As on the real runtime a copy
// from or to PAGEABLE host memory is finished with that memory when the call returns (staged on the way in, synchronous on the way out);
// only pinned memory (hipHostMalloc), device-to-device copies, fills and kernels are truly asynchronous.
struct Op {
    enum Type { Copy, Fill, Fill32, Record, Wait, Nop } type;
    void *dst; const void *src; size_t bytes; int value; Event *event;
    std::vector<char> staged;                  // a pageable source, taken at call time
};
struct Stream { int magic; std::deque<Op> queue; };
bool g_deferred = false;
std::recursive_mutex g_queue_mu;               // one lock over every queue: worker threads enqueue while another thread synchronises

// Made on first use and never destroyed: a hipcc-compiled object registers its kernels from a static constructor, which may run before
// this library's own (the library preloaded into a program that links libptmi directly) and unregisters them from a destructor after it.
struct State {
    std::mutex mu;
    std::map<void *, size_t> blocks;            // live "device" blocks
    std::map<void *, long> block_seq;           // ... and the how-manieth hipMalloc of the process each was
    std::map<void *, size_t> host_blocks;       // live pinned host blocks
    std::map<const void *, std::string> kernels;        // host stub address -> device name
    std::map<std::string, long> launches;
    struct Poke { std::string kernel; long nth; size_t offset; unsigned int value; bool armed; };
    std::vector<Poke> pokes;
    Stream null_stream{kStreamMagic, {}};
    std::set<Stream *> streams;                 // live streams (deferred mode: hipFree and hipMemcpy synchronise with all of them / the NULL one)
};
State &S() { static State *state = new State; return *state; }
#define g_mu (S().mu)
#define g_blocks (S().blocks)
#define g_block_seq (S().block_seq)
#define g_host_blocks (S().host_blocks)
#define g_kernels (S().kernels)
#define g_launches (S().launches)
long g_mallocs = 0;
long g_streams = 0, g_events = 0;
long g_calls[6] = {0}, g_fail_at[6] = {0}, g_fail_more[6] = {0};
thread_local int g_device = 0;             // (the current device is the thread's too)
int g_cus = 8;                                 // a small device: launch grids and per-CU tables stay small
size_t g_total = 64ull << 30;
// the sticky error is the calling THREAD's, as in the real runtime
thread_local hipError_t g_last = hipSuccess;
thread_local long g_seq = 0, g_last_seq = 0, g_launch_seq = 0;      // the thread's call numbers: of the call that set g_last, of its latest launch
std::atomic<long> g_stale{0};                  // hipGetLastError calls that handed out an error OLDER than the thread's latest (successful) launch


struct Config { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local Config t_config{};

bool fails(int kind)
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_calls[kind];
    if (g_fail_at[kind] > 0 && --g_fail_at[kind] == 0) {
        if (std::getenv("HIPSTUB_TRACE")) { std::fprintf(stderr, "HIPSTUB: injected failure of kind %d here:\n", kind); __sanitizer_print_stack_trace(); }
        return true;
    }
    if (g_fail_at[kind] == 0 && g_fail_more[kind] > 0) { --g_fail_more[kind]; return true; }      // ... and the calls after it
    return false;
}

hipError_t set(hipError_t e) { if (e != hipSuccess) { g_last = e; g_last_seq = ++g_seq; } return e; }

bool stream_ok(hipStream_t s) { return s == nullptr || reinterpret_cast<Stream *>(s)->magic == kStreamMagic; }   // (ASan sees a freed or wild one)
Stream *stream_of(hipStream_t s) { return s ? reinterpret_cast<Stream *>(s) : &S().null_stream; }

void apply_pokes(void *dst, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_mu);
    std::vector<State::Poke> &pokes = S().pokes;
    for (size_t i = 0; i < pokes.size();) {
        if (pokes[i].armed && pokes[i].offset + sizeof(unsigned int) <= bytes) {
            std::memcpy(static_cast<char *>(dst) + pokes[i].offset, &pokes[i].value, sizeof(unsigned int));
            pokes.erase(pokes.begin() + (long)i);
        } else ++i;
    }
}

void flush(Stream *st, Event *upto = nullptr);
void run(Op &op)
{
    switch (op.type) {
    case Op::Copy: std::memmove(op.dst, op.staged.empty() ? op.src : op.staged.data(), op.bytes); break;     // (instrumented: memory that has gone in between is ASan's to report)
    case Op::Fill: std::memset(op.dst, op.value, op.bytes); break;
    case Op::Fill32: for (size_t i = 0; i < op.bytes; ++i) static_cast<int *>(op.dst)[i] = op.value; break;
    case Op::Record: op.event->t = std::chrono::steady_clock::now(); op.event->pending = false; break;
    case Op::Wait: if (op.event->pending && op.event->recorded_on) flush(op.event->recorded_on, op.event); break;
    case Op::Nop: break;
    }
}
// everything `st` has queued, in order -- or up to and including the record of `upto`
void flush(Stream *st, Event *upto)
{
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    while (!st->queue.empty()) {
        Op op = std::move(st->queue.front());
        st->queue.pop_front();
        run(op);
        if (upto && op.type == Op::Record && op.event == upto) return;
    }
}
void flush_all()
{
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    flush(&S().null_stream);
    for (Stream *st : S().streams) flush(st);
}
bool is_pinned(const void *p)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_host_blocks.upper_bound(const_cast<void *>(p));
    if (it == g_host_blocks.begin()) return false;
    --it;
    return static_cast<const char *>(p) < static_cast<const char *>(it->first) + it->second;
}

}  // namespace

extern "C" {

// ---- the test's handles ------------------------------------------------------------------------------------------------------------
int hipstub_is_the_stub(void) { return 1; }
void hipstub_fail(int kind, long k) { std::lock_guard<std::mutex> lock(g_mu); if (kind >= 0 && kind < 6) { g_fail_at[kind] = k; g_fail_more[kind] = 0; } }
// the k-th call of the kind from now fails AND the `more` calls of the kind after it (a device that stays broken); hipstub_fail(kind, 0) ends it
void hipstub_fail_run(int kind, long k, long more) { std::lock_guard<std::mutex> lock(g_mu); if (kind >= 0 && kind < 6) { g_fail_at[kind] = k; g_fail_more[kind] = more; } }
long hipstub_calls(int kind) { std::lock_guard<std::mutex> lock(g_mu); return kind >= 0 && kind < 6 ? g_calls[kind] : -1; }
long hipstub_live_blocks(void) { std::lock_guard<std::mutex> lock(g_mu); return (long)g_blocks.size(); }
long hipstub_live_host_blocks(void) { std::lock_guard<std::mutex> lock(g_mu); return (long)g_host_blocks.size(); }
long hipstub_live_streams(void) { std::lock_guard<std::mutex> lock(g_mu); return g_streams; }
long hipstub_live_events(void) { std::lock_guard<std::mutex> lock(g_mu); return g_events; }
unsigned long long hipstub_live_bytes(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    unsigned long long total = 0;
    for (const auto &b : g_blocks) total += b.second;
    return total;
}
long hipstub_launches(const char *name_part)          // launches of kernels whose (mangled) name contains name_part; "" = all
{
    std::lock_guard<std::mutex> lock(g_mu);
    long total = 0;
    for (const auto &k : g_launches)
        if (!name_part || !*name_part || k.first.find(name_part) != std::string::npos) total += k.second;
    return total;
}
// Kernels do not run, so the counters the host reads back after a launch are zeros.  A test can say what they "were": after the nth launch
// (from now) of a kernel whose name contains `kernel_part`, the first device-to-host copy large enough gets `value` at byte `offset`.
void hipstub_poke(const char *kernel_part, long nth, unsigned long long offset, unsigned int value)
{
    std::lock_guard<std::mutex> lock(g_mu);
    S().pokes.push_back(State::Poke{kernel_part ? kernel_part : "", nth, (size_t)offset, value, false});
}
void hipstub_clear_pokes(void) { std::lock_guard<std::mutex> lock(g_mu); S().pokes.clear(); }
void hipstub_set_deferred(int on) { flush_all(); g_deferred = on != 0; }
long hipstub_queued(void)                         // operations waiting for a synchronisation (deferred mode)
{
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    long n = (long)S().null_stream.queue.size();
    for (Stream *st : S().streams) n += (long)st->queue.size();
    return n;
}
void hipstub_print_live(void)                     // which hipMalloc calls of the process made the blocks that are still alive
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (const auto &b : g_blocks) std::fprintf(stderr, "HIPSTUB: live block %p, %zu bytes, hipMalloc number %ld\n", b.first, b.second, g_block_seq[b.first]);
}
long hipstub_stale_errors(void) { return g_stale.load(); }
int hipstub_clear_error(void) { const int e = (int)g_last; g_last = hipSuccess; return e; }     // (the test's own: not counted)
void hipstub_set_device_size(int cus, unsigned long long total_bytes) { std::lock_guard<std::mutex> lock(g_mu); g_cus = cus; g_total = total_bytes; }

// ---- registration of the code objects (what a hipcc-compiled host object does at load time) ---------------------------------------
void **__hipRegisterFatBinary(const void *) { static void *handle = nullptr; return &handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_function, char *, const char *device_name, unsigned int, void *, void *, void *, void *, int *)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_kernels[host_function] = device_name ? device_name : "?";
}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    t_config = Config{grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    *grid = t_config.grid; *block = t_config.block; *shmem = t_config.shmem; *stream = t_config.stream;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *function, dim3 grid, dim3 block, void **args, size_t, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (fails(2)) return set(hipErrorLaunchFailure);
    if (!args) return set(hipErrorInvalidValue);
    // a launch the hardware would refuse is refused here too
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || (unsigned long long)block.x * block.y * block.z > 1024ull)
        return set(hipErrorInvalidConfiguration);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_kernels.find(function);
    const std::string name = it == g_kernels.end() ? std::string("<unregistered>") : it->second;
    ++g_launches[name];
    for (State::Poke &p : S().pokes)
        if (!p.armed && name.find(p.kernel) != std::string::npos && --p.nth == 0) p.armed = true;
    g_launch_seq = ++g_seq;
    return hipSuccess;
}

// ---- devices -----------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return set(hipErrorInvalidDevice); g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t attr, int device)
{
    if (device != 0) return set(hipErrorInvalidDevice);
    switch (attr) {
    case hipDeviceAttributeMultiprocessorCount: *value = g_cus; break;
    case hipDeviceAttributeWarpSize: *value = 64; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *value = 160 * 1024; break;
    default: *value = 0; break;
    }
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *total_b = g_total; *free_b = g_total / 2; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
// The real runtime's rule: the last ERROR any call of the thread returned, kept until somebody asks.  A caller that launches a kernel and
// then asks is therefore handed an older call's error if nobody asked in between -- counted, so that a test can demand it never happens.
hipError_t hipGetLastError(void)
{
    const hipError_t e = g_last;
    if (e != hipSuccess && g_last_seq < g_launch_seq) {
        ++g_stale;
        if (std::getenv("HIPSTUB_TRACE")) { std::fprintf(stderr, "HIPSTUB: an older call's error is handed out after a launch, here:\n"); __sanitizer_print_stack_trace(); }
    }
    g_last = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error (HIP stand-in)";
    case hipErrorOutOfMemory: return "out of memory (HIP stand-in)";
    case hipErrorLaunchFailure: return "launch failure (HIP stand-in)";
    default: return "error (HIP stand-in)";
    }
}

// ---- memory ------------------------------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t bytes)
{
    if (!p) return set(hipErrorInvalidValue);
    if (fails(0)) { *p = nullptr; return set(hipErrorOutOfMemory); }
    if (bytes > g_total) { *p = nullptr; return set(hipErrorOutOfMemory); }      // more than the "device" has (hipstub_set_device_size)
    void *block = std::calloc(bytes ? bytes : 1, 1);
    if (!block) { *p = nullptr; return set(hipErrorOutOfMemory); }
    std::lock_guard<std::mutex> lock(g_mu);
    g_blocks[block] = bytes;
    g_block_seq[block] = ++g_mallocs;
    if (const char *t = std::getenv("HIPSTUB_TRACE_MALLOC"))
        if (std::atol(t) == g_mallocs) { std::fprintf(stderr, "HIPSTUB: hipMalloc number %ld (%zu bytes) here:\n", g_mallocs, bytes); __sanitizer_print_stack_trace(); }
    *p = block;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    if (g_deferred) flush_all();                             // (the real hipFree waits for the device)
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end()) {
            std::fprintf(stderr, "HIPSTUB: hipFree of %p, which is not a live device block (freed twice, or not from hipMalloc)\n", p);
            std::abort();
        }
        g_blocks.erase(it);
        g_block_seq.erase(p);
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int)
{
    if (!p) return set(hipErrorInvalidValue);
    if (fails(4)) { *p = nullptr; return set(hipErrorOutOfMemory); }
    void *block = std::calloc(bytes ? bytes : 1, 1);
    if (!block) { *p = nullptr; return set(hipErrorOutOfMemory); }
    std::lock_guard<std::mutex> lock(g_mu);
    g_host_blocks[block] = bytes;
    *p = block;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    if (!p) return hipSuccess;
    if (g_deferred) flush_all();
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_host_blocks.find(p);
        if (it == g_host_blocks.end()) {
            std::fprintf(stderr, "HIPSTUB: hipHostFree of %p, which is not a live pinned block\n", p);
            std::abort();
        }
        g_host_blocks.erase(it);
    }
    std::free(p);
    return hipSuccess;
}
namespace {
hipError_t copy_now(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    std::memmove(dst, src, bytes);                           // (instrumented: a span past either block's end is ASan's to report)
    if (kind == hipMemcpyDeviceToHost) apply_pokes(dst, bytes);      // what a kernel "would have counted": into the first read-back that holds the words
    return hipSuccess;
}
}  // namespace
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (fails(1)) return set(hipErrorUnknown);
    if (bytes && (!dst || !src)) return set(hipErrorInvalidValue);
    if (g_deferred) flush(&S().null_stream);                 // (synchronous, on the NULL stream; the library's streams are non-blocking ones)
    return copy_now(dst, src, bytes, kind);
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (fails(1)) return set(hipErrorUnknown);
    if (bytes && (!dst || !src)) return set(hipErrorInvalidValue);
    if (!g_deferred) return copy_now(dst, src, bytes, kind);
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    Stream *st = stream_of(stream);
    if (kind == hipMemcpyDeviceToHost && !is_pinned(dst)) {  // to pageable memory: in stream order, and done when the call returns
        flush(st);
        return copy_now(dst, src, bytes, kind);
    }
    Op op{Op::Copy, dst, src, bytes, 0, nullptr, {}};
    if (kind == hipMemcpyHostToDevice && !is_pinned(src)) op.staged.assign(static_cast<const char *>(src), static_cast<const char *>(src) + bytes);     // from pageable memory: taken now
    st->queue.push_back(std::move(op));
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (bytes && !dst) return set(hipErrorInvalidValue);
    if (!g_deferred) { std::memset(dst, value, bytes); return hipSuccess; }
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    stream_of(stream)->queue.push_back(Op{Op::Fill, dst, nullptr, bytes, value, nullptr, {}});
    return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (count && !dst) return set(hipErrorInvalidValue);
    if (!g_deferred) {
        int *p = static_cast<int *>(dst);
        for (size_t i = 0; i < count; ++i) p[i] = value;
        return hipSuccess;
    }
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    stream_of(stream)->queue.push_back(Op{Op::Fill32, dst, nullptr, count, value, nullptr, {}});
    return hipSuccess;
}

// ---- streams and events ------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    if (fails(5)) { *s = nullptr; return set(hipErrorUnknown); }
    Stream *st = new Stream{kStreamMagic, {}};
    *s = reinterpret_cast<hipStream_t>(st);
    { std::lock_guard<std::recursive_mutex> qlock(g_queue_mu); S().streams.insert(st); }
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_streams;
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned int flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s || !stream_ok(s)) return set(hipErrorInvalidHandle);
    Stream *p = reinterpret_cast<Stream *>(s);
    { std::lock_guard<std::recursive_mutex> qlock(g_queue_mu); flush(p); S().streams.erase(p); }      // (the real call lets the queued work finish)
    p->magic = 0;
    delete p;
    std::lock_guard<std::mutex> lock(g_mu);
    --g_streams;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (!stream_ok(s)) return set(hipErrorInvalidHandle);
    if (fails(3)) return set(hipErrorUnknown);
    flush(stream_of(s));
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int)
{
    if (fails(5)) { *e = nullptr; return set(hipErrorUnknown); }
    *e = reinterpret_cast<hipEvent_t>(new Event{kEventMagic, std::chrono::steady_clock::now(), nullptr, false});
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_events;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    Event *p = reinterpret_cast<Event *>(e);
    if (!p || p->magic != kEventMagic) return set(hipErrorInvalidHandle);
    {   // (the real call defers the destruction until the record has happened; a wait for an event that has happened is no wait)
        std::lock_guard<std::recursive_mutex> qlock(g_queue_mu);
        if (p->pending && p->recorded_on) flush(p->recorded_on, p);
        auto forget = [&](Stream *st) { for (Op &op : st->queue) if (op.type == Op::Wait && op.event == p) op.type = Op::Nop; };
        forget(&S().null_stream);
        for (Stream *st : S().streams) forget(st);
    }
    p->magic = 0;
    delete p;
    std::lock_guard<std::mutex> lock(g_mu);
    --g_events;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    Event *p = reinterpret_cast<Event *>(e);
    if (!p || p->magic != kEventMagic || !stream_ok(s)) return set(hipErrorInvalidHandle);
    if (!g_deferred) { p->t = std::chrono::steady_clock::now(); return hipSuccess; }
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    if (p->pending && p->recorded_on) flush(p->recorded_on, p);      // (recorded again before the first record happened: keep it simple)
    p->recorded_on = stream_of(s); p->pending = true;
    p->recorded_on->queue.push_back(Op{Op::Record, nullptr, nullptr, 0, 0, p, {}});
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    Event *p = reinterpret_cast<Event *>(e);
    if (!p || p->magic != kEventMagic) return set(hipErrorInvalidHandle);
    if (p->pending && p->recorded_on) flush(p->recorded_on, p);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    Event *pa = reinterpret_cast<Event *>(a), *pb = reinterpret_cast<Event *>(b);
    if (!pa || !pb || pa->magic != kEventMagic || pb->magic != kEventMagic) return set(hipErrorInvalidHandle);
    if (pa->pending || pb->pending) return hipErrorNotReady;
    *ms = std::chrono::duration<float, std::milli>(pb->t - pa->t).count();
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
    Event *p = reinterpret_cast<Event *>(e);
    if (!p || p->magic != kEventMagic || !stream_ok(s)) return set(hipErrorInvalidHandle);
    if (!g_deferred) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(g_queue_mu);
    if (p->pending) stream_of(s)->queue.push_back(Op{Op::Wait, nullptr, nullptr, 0, 0, p, {}});       // (an event that has happened already is no wait)
    return hipSuccess;
}

}  // extern "C"
