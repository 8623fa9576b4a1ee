// ptmi_group.cpp -- the multi-device entry of the C ABI (include/ptmi.h, "groups"): ONE host process -- the reference's
// application is one Haskell process (app/Main.hs:188-191) -- drives the GPUs of a node through a group of per-device
// contexts.  The image is cut into row stripes dealt round-robin to the members (ptmi_set_partition); every member keeps
// its colour + RNG planes resident and renders without any exchange (`render Inline` is a per-pixel map,
// src/Scene/Trace.hs:193-200); seeds come from the global pixel index, so the stitched result is the single-device image.
// The one exchange is the read-out of the three colour planes (what graphicsLoop reads, app/Main.hs:350):
//   * to the HOST (ptmi_group_download_color): every member's planes come down through its own staged copy path, all
//     members at once (one host thread each), and the stripes are stitched into the caller's [H][W] planes;
//   * to a ROOT DEVICE (ptmi_group_gather_color): RCCL over xGMI -- ncclCommInitAll for the group's devices, one grouped
//     ncclSend per peer / ncclRecv per peer on the root (every peer has its own link to the root), then a stitch kernel on
//     the root's stream.  librccl is loaded on first use (dlopen): single-device users never pay for it.
#include "../../include/ptmi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ptmi_kernels.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    bool load()
    {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { error = std::string("dlopen librccl: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(handle, n); if (!p) error = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};

Rccl g_rccl;
std::mutex g_rccl_mu;

}  // namespace

struct ptmi_group {
    std::mutex mu;
    std::vector<int> devices;
    std::vector<ptmi_ctx *> members;
    int stripe_rows = 8;
    int width = 0, height = 0;
    std::string err;
    // device gather
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> comm_streams;       // one per member, on its device
    std::vector<float *> send_snap;              // per member: [3][local_rows][W] snapshot of its colour planes (contiguous)
    float *recv_block = nullptr;                 // on the root: every member's snapshot, back to back
    int recv_root = -1;
    size_t recv_floats = 0;
    std::vector<float> host_scratch;             // host download: members' planes before stitching
};

namespace {

// (the message is the calling thread's, as ptmi_last_error's: csrc/ptmi_api.cpp)
thread_local std::string t_group_error;
thread_local const ptmi_group *t_group_error_of = nullptr;

int gfail(ptmi_group *g, int code, const std::string &msg)
{
    if (g) { g->err = msg; t_group_error = msg; t_group_error_of = g; }
    return code;
}

int member_fail(ptmi_group *g, int i, int rc)
{
    return gfail(g, rc, "member " + std::to_string(i) + " (device " + std::to_string(g->devices[(size_t)i]) + "): " + ptmi_last_error(g->members[(size_t)i]));
}

#define GROUP_HIP(g, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return gfail((g), e_ == hipErrorOutOfMemory ? PTMI_ENOMEM : PTMI_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
    } while (0)
#define GROUP_NCCL(g, call)                                                                      \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) { (void)hipGetLastError(); return gfail((g), PTMI_EHIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } \
    } while (0)

void release_gather(ptmi_group *g)
{
    for (size_t i = 0; i < g->send_snap.size(); ++i)
        if (g->send_snap[i]) { (void)hipSetDevice(g->devices[i]); (void)hipFree(g->send_snap[i]); }
    g->send_snap.clear();
    if (g->recv_block) { (void)hipSetDevice(g->devices[(size_t)g->recv_root]); (void)hipFree(g->recv_block); g->recv_block = nullptr; }
    g->recv_root = -1; g->recv_floats = 0;
}

}  // namespace

extern "C" {

/* rows a part holds / the image row of one of them: the arithmetic of ptmi_set_partition, usable without a device */
int ptmi_partition_rows(int height, int stripe_rows, int n_parts, int part)
{
    if (height <= 0 || stripe_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return PTMI_EINVAL;
    const long long cycle = (long long)stripe_rows * n_parts;
    long long rows = (height / cycle) * stripe_rows;
    long long rem = height % cycle - (long long)part * stripe_rows;
    if (rem > stripe_rows) rem = stripe_rows;
    if (rem > 0) rows += rem;
    return (int)rows;
}

int ptmi_partition_global_row(int height, int stripe_rows, int n_parts, int part, int local_row)
{
    const int rows = ptmi_partition_rows(height, stripe_rows, n_parts, part);
    if (rows < 0) return rows;
    if (local_row < 0 || local_row >= rows) return PTMI_EINVAL;
    return ((local_row / stripe_rows) * n_parts + part) * stripe_rows + local_row % stripe_rows;
}

int ptmi_group_create(ptmi_group **out, const int *devices, int n_devices, int stripe_rows)
{
    if (!out) return PTMI_EINVAL;
    *out = nullptr;
    if (!devices || n_devices <= 0 || stripe_rows < 0) return PTMI_EINVAL;
    ptmi_group *g = new (std::nothrow) ptmi_group;
    if (!g) return PTMI_ENOMEM;
    g->stripe_rows = stripe_rows > 0 ? stripe_rows : 8;
    for (int i = 0; i < n_devices; ++i) {
        ptmi_ctx *c = nullptr;
        const int rc = ptmi_create(&c, devices[i]);
        if (rc != PTMI_OK) { ptmi_group_destroy(g); return rc; }      /* ptmi_last_error(NULL) holds the message */
        g->devices.push_back(devices[i]);
        g->members.push_back(c);
        if (n_devices > 1) {
            const int prc = ptmi_set_partition(c, g->stripe_rows, n_devices, i);
            if (prc != PTMI_OK) { ptmi_group_destroy(g); return prc; }
        }
    }
    *out = g;
    return PTMI_OK;
}

void ptmi_group_destroy(ptmi_group *g)
{
    if (!g) return;
    release_gather(g);
    for (size_t i = 0; i < g->comm_streams.size(); ++i)
        if (g->comm_streams[i]) { (void)hipSetDevice(g->devices[i]); (void)hipStreamDestroy(g->comm_streams[i]); }
    for (ncclComm_t c : g->comms) if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (ptmi_ctx *c : g->members) ptmi_destroy(c);
    (void)hipGetLastError();                                 // (errors ignored above are not left in the runtime's sticky slot)
    delete g;
}

int ptmi_group_size(const ptmi_group *g) { return g ? (int)g->members.size() : PTMI_EINVAL; }

ptmi_ctx *ptmi_group_member(ptmi_group *g, int i)
{
    if (!g || i < 0 || i >= (int)g->members.size()) return nullptr;
    return g->members[(size_t)i];
}

const char *ptmi_group_last_error(const ptmi_group *g)
{
    if (!g) return "";
    if (t_group_error_of == g) return t_group_error.c_str();
    thread_local std::string copy;
    std::lock_guard<std::mutex> lock(const_cast<ptmi_group *>(g)->mu);
    copy = g->err;
    return copy.c_str();
}

int ptmi_group_set_scene(ptmi_group *g, const ptmi_sphere *spheres, int n_spheres, const ptmi_plane *planes, int n_planes)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_set_scene(g->members[i], spheres, n_spheres, planes, n_planes)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_resize(ptmi_group *g, int width, int height)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    release_gather(g);
    g->width = g->height = 0;                                // unsized until EVERY member stands at the new size: a failure half-way leaves members of
    for (size_t i = 0; i < g->members.size(); ++i)           // two sizes (and one of none), and the read-outs below go by the group's
        if (int rc = ptmi_resize(g->members[i], width, height)) return member_fail(g, (int)i, rc);
    g->width = width; g->height = height;
    return PTMI_OK;
}

namespace {
// The rows member i holds -- which must be the rows the GROUP's size deals it (a member resized on its own, or left unsized by a failure,
// would have the read-outs copy by one size into buffers of another).
int member_rows(ptmi_group *g, int i, int *rows)
{
    int width = 0, height = 0;
    ptmi::context_size(g->members[(size_t)i], &width, &height);
    *rows = ptmi_local_rows(g->members[(size_t)i]);
    if (width != g->width || height != g->height || *rows < 0 || *rows != ptmi_partition_rows(g->height, g->stripe_rows, (int)g->members.size(), i))
        return gfail(g, PTMI_ESTATE, "member " + std::to_string(i) + " is not sized as the group is (ptmi_group_resize sizes the members)");
    return PTMI_OK;
}
}  // namespace

int ptmi_group_init_output(ptmi_group *g, uint64_t seed0)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_init_output(g->members[i], seed0)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_reseed(ptmi_group *g, uint64_t seed0)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_reseed(g->members[i], seed0)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_render(ptmi_group *g, const ptmi_camera *camera, int algorithm, int bounce_limit, int n_spp)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    const size_t n = g->members.size();
    // The per-pixel kernels are launched asynchronously: every device is busy before the first call returns.  The stream
    // form of Streams reads stream lengths back while it runs (the host plays `awhile`), so its members get a thread each.
    bool blocking = false;                                   // (any member: options and variants can be set per member too)
    if (algorithm == PTMI_STREAMS && n > 1)
        for (size_t i = 0; i < n; ++i) blocking |= ptmi_render_blocks(g->members[i], algorithm) == 1;
    if (blocking) {
        std::vector<int> rcs(n, PTMI_OK);
        std::vector<std::thread> threads;
        for (size_t i = 0; i < n; ++i)
            threads.emplace_back([&, i]() { rcs[i] = ptmi_render(g->members[i], camera, algorithm, bounce_limit, n_spp); });
        for (std::thread &t : threads) t.join();
        for (size_t i = 0; i < n; ++i) if (rcs[i] != PTMI_OK) return member_fail(g, (int)i, rcs[i]);
        return PTMI_OK;
    }
    for (size_t i = 0; i < n; ++i)
        if (int rc = ptmi_render(g->members[i], camera, algorithm, bounce_limit, n_spp)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_set_option(ptmi_group *g, int option, int64_t value)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_set_option(g->members[i], option, value)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_set_variant(ptmi_group *g, int variant)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_set_variant(g->members[i], variant)) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_synchronize(ptmi_group *g)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = ptmi_synchronize(g->members[i])) return member_fail(g, (int)i, rc);
    return PTMI_OK;
}

int ptmi_group_download_color(ptmi_group *g, float *r, float *gp, float *b)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    if (g->width <= 0) return gfail(g, PTMI_ESTATE, "ptmi_group_resize has not been called");
    if (!r || !gp || !b) return gfail(g, PTMI_EINVAL, "a plane pointer is NULL");
    const int n = (int)g->members.size();
    const size_t w = (size_t)g->width;
    if (n == 1) {
        int rows = 0;
        if (int rc = member_rows(g, 0, &rows)) return rc;
        if (int rc = ptmi_download_color(g->members[0], r, gp, b)) return member_fail(g, 0, rc);
        return PTMI_OK;
    }
    // every member's planes into its slice of the scratch, all members at once; then the stripes into place
    std::vector<size_t> offset((size_t)n + 1, 0);
    std::vector<int> rows_of((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (int rc = member_rows(g, i, &rows_of[(size_t)i])) return rc;
        offset[(size_t)i + 1] = offset[(size_t)i] + 3 * (size_t)rows_of[(size_t)i] * w;
    }
    g->host_scratch.resize(offset[(size_t)n]);
    std::vector<int> rcs((size_t)n, PTMI_OK);
    std::vector<std::thread> threads;
    float *out[3] = {r, gp, b};
    for (int i = 0; i < n; ++i) {
        threads.emplace_back([&, i]() {
            ptmi_ctx *c = g->members[(size_t)i];
            const int rows = rows_of[(size_t)i];
            float *base = g->host_scratch.data() + offset[(size_t)i];
            const size_t plane = (size_t)rows * w;
            rcs[(size_t)i] = ptmi_download_color(c, base, base + plane, base + 2 * plane);
            if (rcs[(size_t)i] != PTMI_OK) return;
            const int s = g->stripe_rows;
            for (int k = 0; k < 3; ++k)
                for (int lr = 0; lr < rows; lr += s) {                 /* one stripe: contiguous on both sides */
                    const int take = rows - lr < s ? rows - lr : s;
                    const int gr = ((lr / s) * n + i) * s;
                    std::memcpy(out[k] + (size_t)gr * w, base + (size_t)k * plane + (size_t)lr * w, (size_t)take * w * sizeof(float));
                }
        });
    }
    for (std::thread &t : threads) t.join();
    for (int i = 0; i < n; ++i) if (rcs[(size_t)i] != PTMI_OK) return member_fail(g, i, rcs[(size_t)i]);
    return PTMI_OK;
}

int ptmi_group_gather_color(ptmi_group *g, int root, float *r_dev, float *g_dev, float *b_dev)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    const int n = (int)g->members.size();
    if (root < 0 || root >= n) return gfail(g, PTMI_EINVAL, "root is not a member index");
    if (g->width <= 0) return gfail(g, PTMI_ESTATE, "ptmi_group_resize has not been called");
    if (!r_dev || !g_dev || !b_dev) return gfail(g, PTMI_EINVAL, "a plane pointer is NULL");
    const size_t w = (size_t)g->width;
    struct DeviceGuard {                                     // the caller's current HIP device is the caller's: put it back
        int dev = -1;
        DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
        ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    } restore_device;
    const bool force_rccl = [] { const char *e = getenv("PTMI_GROUP_FORCE_RCCL"); return e && e[0] == '1'; }();
    const bool use_rccl = n > 1 || force_rccl;
    if (use_rccl && g->comms.empty()) {
        std::lock_guard<std::mutex> rl(g_rccl_mu);
        if (!g_rccl.load()) return gfail(g, PTMI_EHIP, g_rccl.error);
        g->comms.assign((size_t)n, nullptr);
        GROUP_NCCL(g, g_rccl.CommInitAll(g->comms.data(), n, g->devices.data()));
    }
    if (g->comm_streams.empty()) {
        g->comm_streams.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i) {
            GROUP_HIP(g, hipSetDevice(g->devices[(size_t)i]));
            GROUP_HIP(g, hipStreamCreateWithFlags(&g->comm_streams[(size_t)i], hipStreamNonBlocking));
        }
    }
    std::vector<size_t> floats((size_t)n), offset((size_t)n + 1, 0);
    std::vector<int> rows_of((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (int rc = member_rows(g, i, &rows_of[(size_t)i])) return rc;
        floats[(size_t)i] = 3 * (size_t)rows_of[(size_t)i] * w;
        offset[(size_t)i + 1] = offset[(size_t)i] + floats[(size_t)i];
    }
    if (g->send_snap.empty() || g->recv_root != root || g->recv_floats != offset[(size_t)n]) {
        release_gather(g);
        g->send_snap.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i) {
            GROUP_HIP(g, hipSetDevice(g->devices[(size_t)i]));
            GROUP_HIP(g, hipMalloc(&g->send_snap[(size_t)i], (floats[(size_t)i] ? floats[(size_t)i] : 1) * sizeof(float)));
        }
        GROUP_HIP(g, hipSetDevice(g->devices[(size_t)root]));
        GROUP_HIP(g, hipMalloc(&g->recv_block, (offset[(size_t)n] ? offset[(size_t)n] : 1) * sizeof(float)));
        g->recv_root = root; g->recv_floats = offset[(size_t)n];
    }
    // 1. every member: wait for its render stream, snapshot its three colour planes contiguously (device-to-device)
    for (int i = 0; i < n; ++i) {
        GROUP_HIP(g, hipSetDevice(g->devices[(size_t)i]));
        if (int rc = ptmi_snapshot_color(g->members[(size_t)i], g->send_snap[(size_t)i], g->comm_streams[(size_t)i])) return member_fail(g, i, rc);
    }
    // 2. the exchange: peers send, the root receives every peer's snapshot (its own is copied on the device)
    float *recv = g->recv_block;
    if (use_rccl) {
        GROUP_NCCL(g, g_rccl.GroupStart());
        ncclResult_t inside = ncclSuccess;                   // an error between GroupStart and GroupEnd must not leave the group open
        const char *what = "";
        for (int i = 0; i < n && inside == ncclSuccess; ++i) {
            if (i == root && !force_rccl) continue;
            if (floats[(size_t)i] == 0) continue;
            inside = g_rccl.Send(g->send_snap[(size_t)i], floats[(size_t)i], ncclFloat32, root, g->comms[(size_t)i], g->comm_streams[(size_t)i]);
            what = "ncclSend";
            if (inside != ncclSuccess) break;
            inside = g_rccl.Recv(recv + offset[(size_t)i], floats[(size_t)i], ncclFloat32, i, g->comms[(size_t)root], g->comm_streams[(size_t)root]);
            what = "ncclRecv";
        }
        const ncclResult_t ended = g_rccl.GroupEnd();
        if (inside != ncclSuccess || ended != ncclSuccess) (void)hipGetLastError();      // (a runtime error under RCCL's: reported here, not left behind)
        if (inside != ncclSuccess) return gfail(g, PTMI_EHIP, std::string(what) + ": " + g_rccl.GetErrorString(inside));
        if (ended != ncclSuccess) return gfail(g, PTMI_EHIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ended));
    }
    GROUP_HIP(g, hipSetDevice(g->devices[(size_t)root]));
    if (!(use_rccl && force_rccl) && floats[(size_t)root])
        GROUP_HIP(g, hipMemcpyAsync(recv + offset[(size_t)root], g->send_snap[(size_t)root], floats[(size_t)root] * sizeof(float),
                                    hipMemcpyDeviceToDevice, g->comm_streams[(size_t)root]));
    // 3. stitch the stripes into the caller's [H][W] planes on the root
    for (int i = 0; i < n; ++i) {
        const int rows = rows_of[(size_t)i];
        if (rows <= 0) continue;
        GROUP_HIP(g, ptmi::launch_stitch(recv + offset[(size_t)i], rows, g->width, g->stripe_rows, n, i, r_dev, g_dev, b_dev, g->comm_streams[(size_t)root]));
    }
    // the other members' sends must have left before their snapshots are reused
    for (int i = 0; i < n; ++i) {
        GROUP_HIP(g, hipSetDevice(g->devices[(size_t)i]));
        GROUP_HIP(g, hipStreamSynchronize(g->comm_streams[(size_t)i]));
    }
    return PTMI_OK;
}

int ptmi_group_get_stats(ptmi_group *g, ptmi_stats *out)
{
    if (!g) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(g->mu);
    if (!out) return gfail(g, PTMI_EINVAL, "out is NULL");
    std::memset(out, 0, sizeof *out);
    for (size_t i = 0; i < g->members.size(); ++i) {
        ptmi_stats s;
        if (int rc = ptmi_get_stats(g->members[i], &s)) return member_fail(g, (int)i, rc);
        out->live_bounces += s.live_bounces; out->nominal_bounces += s.nominal_bounces; out->samples += s.samples;
        out->last_render_ms = s.last_render_ms > out->last_render_ms ? s.last_render_ms : out->last_render_ms;
        out->stream_iterations = s.stream_iterations > out->stream_iterations ? s.stream_iterations : out->stream_iterations;
        out->stream_rays_dropped += s.stream_rays_dropped; out->stream_rays_truncated += s.stream_rays_truncated;
        out->stream_rays_spilled += s.stream_rays_spilled; out->stream_rays_overflowed += s.stream_rays_overflowed;
    }
    return PTMI_OK;
}

}  // extern "C"
