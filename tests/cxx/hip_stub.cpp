// hip_stub.cpp -- a TEST-ONLY stand-in for the part of the HIP runtime libptmi's HOST side calls (tests/test_host_sanitized.py preloads it
// into a SUBPROCESS, in front of libamdhip64).  It exists so that the host logic of the C ABI -- contexts, options, the chained closure's
// tokens / slots / evictions, partitions, groups, the stream form's bookkeeping, every error path -- can run in this container, which has
// no GPU, under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool, CPU ones are).
//
// What it is: "device memory" is host memory (calloc: a fresh block reads as zeros), copies are memcpy, memsets are memset, streams and
// events are small heap objects and everything is synchronous.  KERNELS DO NOT RUN: hipLaunchKernel counts the launch under the kernel's
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
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" void __sanitizer_print_stack_trace(void);

namespace {

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

struct Stream { int magic; };
struct Event { int magic; std::chrono::steady_clock::time_point t; };
constexpr int kStreamMagic = 0x57e4a3, kEventMagic = 0xe7e47;

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
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (fails(1)) return set(hipErrorUnknown);
    if (bytes && (!dst || !src)) return set(hipErrorInvalidValue);
    std::memmove(dst, src, bytes);                           // (instrumented: a span past either block's end is ASan's to report)
    if (kind == hipMemcpyDeviceToHost) {                     // what a kernel "would have counted": the test's words, into the first read-back that holds them
        std::lock_guard<std::mutex> lock(g_mu);
        std::vector<State::Poke> &pokes = S().pokes;
        for (size_t i = 0; i < pokes.size();) {
            if (pokes[i].armed && pokes[i].offset + sizeof(unsigned int) <= bytes) {
                std::memcpy(static_cast<char *>(dst) + pokes[i].offset, &pokes[i].value, sizeof(unsigned int));
                pokes.erase(pokes.begin() + (long)i);
            } else ++i;
        }
    }
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    return hipMemcpy(dst, src, bytes, kind);
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (bytes && !dst) return set(hipErrorInvalidValue);
    std::memset(dst, value, bytes);
    return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t stream)
{
    if (!stream_ok(stream)) return set(hipErrorInvalidHandle);
    if (count && !dst) return set(hipErrorInvalidValue);
    int *p = static_cast<int *>(dst);
    for (size_t i = 0; i < count; ++i) p[i] = value;
    return hipSuccess;
}

// ---- streams and events ------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    if (fails(5)) { *s = nullptr; return set(hipErrorUnknown); }
    *s = reinterpret_cast<hipStream_t>(new Stream{kStreamMagic});
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_streams;
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned int flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s || !stream_ok(s)) return set(hipErrorInvalidHandle);
    Stream *p = reinterpret_cast<Stream *>(s);
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
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int)
{
    if (fails(5)) { *e = nullptr; return set(hipErrorUnknown); }
    *e = reinterpret_cast<hipEvent_t>(new Event{kEventMagic, std::chrono::steady_clock::now()});
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_events;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    Event *p = reinterpret_cast<Event *>(e);
    if (!p || p->magic != kEventMagic) return set(hipErrorInvalidHandle);
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
    p->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    Event *p = reinterpret_cast<Event *>(e);
    return (!p || p->magic != kEventMagic) ? set(hipErrorInvalidHandle) : hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    Event *pa = reinterpret_cast<Event *>(a), *pb = reinterpret_cast<Event *>(b);
    if (!pa || !pb || pa->magic != kEventMagic || pb->magic != kEventMagic) return set(hipErrorInvalidHandle);
    *ms = std::chrono::duration<float, std::milli>(pb->t - pa->t).count();
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
    Event *p = reinterpret_cast<Event *>(e);
    return (!p || p->magic != kEventMagic || !stream_ok(s)) ? set(hipErrorInvalidHandle) : hipSuccess;
}

}  // extern "C"
