// ptmi_api.cpp -- the C ABI of include/ptmi.h: context, device planes, scene packing, launches.
// Compiled with hipcc together with the kernel units (ptmi_*.hip) into libptmi.so.  There is no CPU
// fallback here: without a HIP device ptmi_create fails with PTMI_ENODEVICE.
#include "../../include/ptmi.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ptmi_kernels.h"
#include "ptmi_stage.h"

using namespace ptmi;

// render Inline with contracted arithmetic: the second object made from ptmi_inline.hip (see its last lines)
extern "C" int ptmi_contracted_launch_inline(const void *args, int variant, void *stream);

struct ptmi_ctx {
    std::mutex mu;
    int device = 0;
    std::string err;

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    int width = 0, height = 0;
    int stripe_rows = 0, n_parts = 1, part = 0;   // stripe_rows == 0: one part holds everything
    int rows_local = 0;

    Planes owned{};          // seven planes carved from owned_block
    void *owned_block = nullptr;
    Planes bound{};
    bool use_bound = false;

    float4 *d_scene = nullptr;
    int n_spheres = 0, n_planes = 0;

    unsigned long long *d_live = nullptr;
    unsigned int *d_work = nullptr;
    unsigned int *d_iters = nullptr;
    uint64_t nominal = 0, samples = 0;

    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_snap = nullptr;
    // The stream form's tail (render_streams_wavefront): the end of the dispatch order is rendered by the per-pixel kernel on a
    // stream of its own, beside the persistent launch.  d_tail_start: where that end begins (written by the order kernel).
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned int *d_tail_start = nullptr;
    int opt_tail_permille = -1;                // PTMI_OPT_STREAM_TAIL: thousandths of the recorded cost the tail may hold (0 = no tail; -1 = automatic)
    bool ev_valid = false;
    int variant = 0;
    Stager stager;                          // pinned ring + worker threads for host-buffer entry points (ptmi_stage.h)

    // cost-ordered dispatch of the tiled kernels: what every quad of tiles cost in the last launch with this key, and
    // the order (most expensive first) later launches with the same key use.  order_state = launches made with this key
    unsigned int *d_quad_cost = nullptr, *d_quad_order = nullptr, *d_quad_class = nullptr;
    unsigned int quad_capacity = 0;
    int order_state = 0;
    unsigned int *d_chunk_done = nullptr;      // sample chunks of the tiled Inline kernel: one word per tile workgroup
    unsigned int chunk_capacity = 0;
    struct OrderKey { ptmi_camera cam; uint64_t scene_version; int dims[8]; } order_key{};
    uint64_t scene_version = 0;

    // scratch for ptmi_render1 / point queries
    void *scratch = nullptr;
    size_t scratch_bytes = 0;

    // wavefront Streams (scenes with the GLASS extension): two ray streams + {next length, dropped}
    bool has_glass = false;
    void *queue_block = nullptr;
    size_t queue_capacity = 0;
    void *hit_block = nullptr;       // stream form: the start hits of the pixels, in regions (HitList)
    size_t hit_capacity = 0;
    unsigned int *d_hit_counts = nullptr;   // ... records per region
    unsigned long long *d_hit_missed = nullptr;   // ... and the pixels of every region that have none
    unsigned int hit_regions = 0;
    // The start-hit list is a function of (camera, scene, shape, partition, dispatch order) only -- every sample of a pixel
    // shoots the same primary ray, in every call -- so it is kept until one of them changes.
    struct HitKey { ptmi_camera cam; uint64_t scene_version, order_generation; int dims[8]; unsigned int region_slots; int cap_allows_split; const void *planes_r; } hit_key{};
    bool hit_list_valid = false;
    uint64_t order_generation = 0;          // bumped whenever the dispatch order (d_quad_order, or its use) changes
    uint64_t hit_split_pixels = 0;          // pixels whose glass primary hit the list replaced by its children's hits
    void *d_snapshots = nullptr;     // stream form, split kernel: the seed every item starts from
    size_t snapshot_bytes = 0;
    int cus = 0;                     // compute units of the device (persistent grids)
    size_t device_memory = (size_t)64 << 30;   // bytes of the device (budget of the stream form's seed snapshots)
    void *tree_stack = nullptr;      // tree walk: the lanes' first waiting children (RenderArgs.tree_stack)
    size_t tree_stack_bytes = 0;
    unsigned int *d_region_done = nullptr;   // stream form, ordered passes: items published per region
    unsigned int region_done_words = 0;
    int opt_ordered_passes = 0;      // PTMI_OPT_ORDERED_PASSES: 0 = automatic, 1 = off, k = k passes
    int opt_pass_handoff = 0;        // PTMI_OPT_PASS_HANDOFF: 0 = release / acquire once per (region, pass); 1 = the fence-free write-through hand-off
    int *d_pass_first = nullptr;     // stream form, split kernel: the samples of every pass (ItemArgs.pass_first), kMaxStreamPasses + 1 entries
    std::vector<int> pass_first_host;   // ... what the device block holds
    unsigned int *d_qcount = nullptr;
    uint64_t rays_dropped = 0;
    uint64_t rays_truncated = 0;
    uint64_t rays_spilled = 0;       // stream form: children that found the wave's ring full and went through HBM
    uint64_t rays_overflowed = 0;    // ... and its spill queue too: traced by an overflow level
    void *spill_block = nullptr;     // stream form: the waves' spill queues
    size_t spill_capacity = 0;
    uint64_t live_host = 0;        // live rays counted on the host (wavefront path)
    unsigned long long *d_stream_counters = nullptr;   // kScWords device counters of the per-pixel Streams kernels

    // options of render Streams (ptmi_set_option)
    int opt_seed_rule = PTMI_SEED_AUTO;              // resolved per scene: effective_seed_rule()
    int opt_step_cap = kStreamStepCapDefault;
    int opt_capacity = 4;
    int grown_capacity = 0;                          // stream form with GLASS: rays per pixel the overflow streams have been GROWN to after a call would have dropped children (0: never)
    void *colour_backup = nullptr;                   // ... the three colour planes as they were before the call's launch (the call is redone if children were dropped)
    size_t colour_backup_bytes = 0;
    int opt_form = PTMI_FORM_AUTO;
    int opt_batch = 0;
    int opt_spp_chunks = 0;                    // 0 = automatic
    int opt_arithmetic = PTMI_ARITH_EXACT;
    int opt_glass_batch = 0;                   // PTMI_OPT_GLASS_BATCH: 0 = automatic, 1 = off, k = GLASS hits wait until k are pending in their wave
    int opt_graded = 1;                        // PTMI_OPT_STREAM_GRADED: the split kernel's passes shrink towards the end of the launch
    int opt_snapshot_mb = 0;                   // PTMI_OPT_SNAPSHOT_BUDGET_MB: 0 = an eighth of the device's memory
    int opt_pass_groups = 0;                  // PTMI_OPT_STREAM_PASS_GROUPS: 0 = automatic, 1 = off, k = the last k passes are handed out region by region

    // The chained closure (ptmi_render1_chained): the RenderResults the caller holds tokens for.  A state's seven planes are one device
    // block (carve), or -- once it had to make room -- one host block of the same layout.  Blocks of released states wait in chain_free.
    struct ChainState {
        uint64_t token = 0;
        int width = 0, height = 0;
        void *block = nullptr;                 // device
        std::unique_ptr<char[]> host;          // evicted: planes_bytes(n) bytes, carve's layout
    };
    std::vector<ChainState> chain;             // in token order (oldest first)
    std::vector<std::pair<size_t, void *>> chain_free;   // (bytes, device block)
    uint64_t chain_serial = 0;                 // this context's number in the process: the upper bits of its tokens
    uint64_t chain_counter = 0;
    int opt_chain_slots = 0;                   // PTMI_OPT_CHAIN_SLOTS: 0 = automatic
    ptmi_chain_stats chain_stats{};
};

namespace {

thread_local std::string g_create_error;
// The message of a failed call is the calling THREAD's (several threads may share a context -- the application has three, app/Main.hs:178-180
// -- and two of them may fail at once: a pointer into the context's own string would be freed under the first reader by the second failure).
thread_local std::string t_error;               // this thread's last failure on a context ...
thread_local const ptmi_ctx *t_error_of = nullptr;   // ... and which

int fail(ptmi_ctx *c, int code, const std::string &msg)
{
    if (c) { c->err = msg; t_error = msg; t_error_of = c; }      // (c->err: under c->mu, which every caller with a context holds)
    else g_create_error = msg;
    return code;
}

// A runtime error leaves through this library's return code -- and not, a second time, through the runtime's sticky slot (hipGetLastError
// keeps the last failure of the thread until somebody asks: the caller's next launch check, or another library's, would find it there).
#define PTMI_HIP(c, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError();                                                       \
            return fail((c), e_ == hipErrorOutOfMemory ? PTMI_ENOMEM : PTMI_EHIP,           \
                        std::string(#call) + ": " + hipGetErrorString(e_));                \
        }                                                                                  \
    } while (0)

// PTMI_SEED_AUTO: `combine new old` (the seed of the result) wherever the reference defines the outcome -- no ray-splitting
// material -- and the accumulator's seed with GLASS, where several results of one step would race for it
int effective_seed_rule(const ptmi_ctx *c)
{
    if (c->opt_seed_rule != PTMI_SEED_AUTO) return c->opt_seed_rule;
    return c->has_glass ? PTMI_SEED_KEEP_ACCUMULATOR : PTMI_SEED_FROM_RESULT;
}

// PTMI_OPT_STREAMS_FORM resolved: does `render Streams` run in its stream ("wavefront") form?  AUTO: only where it pays and costs nothing that
// was promised -- a scene with GLASS (no reference semantics, no defined addition order in Accelerate's permute either) on ONE PART of a
// partitioned image at >= 256 samples per call: the multi-GPU job is as fast as its slowest part, and that part is 5 % faster in the stream
// form (profiles/r05_c5_part.json: part 6 of C5 29.1 against 30.7 ms; imbalance 1.02 against 1.03-1.14).  n_spp < 0: "for some sample count".
bool uses_stream_form(const ptmi_ctx *c, int algorithm, int n_parts, int n_spp)
{
    if (algorithm != PTMI_STREAMS) return false;
    if (c->opt_form == PTMI_FORM_STREAM || c->variant == 9) return true;
    if (c->opt_form == PTMI_FORM_PIXEL) return false;
    return c->has_glass && n_parts > 1 && (n_spp < 0 || n_spp >= 256);
}

int effective_stripe(const ptmi_ctx *c) { return c->stripe_rows > 0 ? c->stripe_rows : (c->height > 0 ? c->height : 1); }

int rows_of_part(int height, int stripe, int n_parts, int part)
{
    const long long cycle = (long long)stripe * n_parts;
    long long rows = (height / cycle) * stripe;
    long long rem = height % cycle - (long long)part * stripe;
    if (rem > stripe) rem = stripe;
    if (rem > 0) rows += rem;
    return (int)rows;
}

Planes carve(void *block, size_t n)
{
    Planes p;
    char *b = static_cast<char *>(block);
    const size_t plane = ((n * 4 + 255) / 256) * 256;
    p.r = reinterpret_cast<float *>(b);
    p.g = reinterpret_cast<float *>(b + plane);
    p.b = reinterpret_cast<float *>(b + 2 * plane);
    p.sa = reinterpret_cast<uint32_t *>(b + 3 * plane);
    p.sb = reinterpret_cast<uint32_t *>(b + 4 * plane);
    p.sc = reinterpret_cast<uint32_t *>(b + 5 * plane);
    p.sctr = reinterpret_cast<uint32_t *>(b + 6 * plane);
    return p;
}
size_t planes_bytes(size_t n) { return 7 * (((n * 4 + 255) / 256) * 256); }
// (two ints multiply to 2^62: seven planes of that would wrap a size_t.  2^40 pixels are 28 TB of planes -- beyond any device, within size_t)
constexpr unsigned long long kMaxPixels = 1ull << 40;
bool too_many_pixels(int width, int height) { return (unsigned long long)width * (unsigned long long)height > kMaxPixels; }

Planes &active(ptmi_ctx *c) { return c->use_bound ? c->bound : c->owned; }

int ensure_scratch(ptmi_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return PTMI_OK;
    if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    PTMI_HIP(c, hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return PTMI_OK;
}

// Host-buffer transfers of the boundary (stream-ordered on c->stream).  Transfers of a megabyte or more go through
// the context's Stager (ptmi_stage.h: parallel page copies into a pinned ring, one DMA per 8-MB chunk -- the driver
// never pins the caller's pages); small ones, or all of them when the engine is off (PTMI_STAGE_THREADS=0), are
// plain async copies.  copy_to_host leaves the stream drained in both cases.
hipError_t copy_to_device(ptmi_ctx *c, const CopySpan *spans, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += spans[i].bytes;
    if (c->stager.threads() > 0 && total >= Stager::kMinBytes) return c->stager.to_device(spans, n, c->stream);
    for (int i = 0; i < n; ++i) {
        if (!spans[i].bytes) continue;
        const hipError_t e = hipMemcpyAsync(spans[i].dev, spans[i].host, spans[i].bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t copy_to_host(ptmi_ctx *c, const CopySpan *spans, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += spans[i].bytes;
    if (c->stager.threads() > 0 && total >= Stager::kMinBytes) return c->stager.to_host(spans, n, c->stream);
    for (int i = 0; i < n; ++i) {
        if (!spans[i].bytes) continue;
        const hipError_t e = hipMemcpyAsync(spans[i].host, spans[i].dev, spans[i].bytes, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(c->stream);
}

// primaryRays' per-launch values (src/Scene/Trace.hs:205-242), evaluated on the host with
// the same arithmetic definitions as the device code (ptmi_core.h); tan is the host libm's.
PrimaryUniforms make_uniforms(const ptmi_camera &cam, int width, int height)
{
    PrimaryUniforms u;
    const float c_fov = (float)cam.fov;
    const float screen_angle = (c_fov * kPi / 180.0f) / 2.0f;
    const float screen_distance = 1.0f / tanf(screen_angle);
    const float screen_half_width = tanf(screen_angle) * screen_distance;
    const V3 c_pos = mk(cam.position[0], cam.position[1], cam.position[2]);
    const V3 c_rot = mk(cam.rotation[0], cam.rotation[1], cam.rotation[2]);
    const V3 c_dir = rotate(angles_to_quaternion(c_rot), mk(0.0f, 0.0f, -1.0f));   // Util.hs:48-50, :96-97
    const float screen_aspect = (float)width / (float)height;                        // Util.hs:192-193
    const V3 center = c_pos + scale_r(c_dir, screen_distance);
    const V3 center_offset = center - c_pos;
    const V3 right = div_r(normalize(cross(center_offset, mk(0.0f, 1.0f, 0.0f))), screen_half_width);
    const V3 top = div_r(cross(c_dir, right), screen_aspect);
    u.pos = c_pos; u.center = center; u.right = right; u.top = top;
    u.size_x = (float)width; u.size_y = (float)(-height);                            // Util.hs:198-200
    return u;
}

void pack_scene(const ptmi_sphere *sph, int ns, const ptmi_plane *pl, int np, std::vector<float4> &out)
{
    out.assign((size_t)ns + 2 * (size_t)np + 2 * ((size_t)ns + np), float4{0, 0, 0, 0});
    size_t k = 0;
    for (int i = 0; i < ns; ++i)
        out[k++] = float4{sph[i].position[0], sph[i].position[1], sph[i].position[2], sph[i].radius * sph[i].radius};
    for (int j = 0; j < np; ++j) {
        out[k++] = float4{pl[j].position[0], pl[j].position[1], pl[j].position[2], 0.0f};
        out[k++] = float4{pl[j].direction[0], pl[j].direction[1], pl[j].direction[2], 0.0f};
    }
    auto mat = [&](const float *color, float illum, int32_t tag, float p) {
        out[k++] = float4{color[0], color[1], color[2], illum};
        out[k++] = float4{u2f((uint32_t)tag), p, p / kPi, 0.5f * (1.0f - p)};
    };
    for (int i = 0; i < ns; ++i) mat(sph[i].color, sph[i].illuminance, sph[i].brdf_tag, sph[i].brdf_param);
    for (int j = 0; j < np; ++j) mat(pl[j].color, pl[j].illuminance, pl[j].brdf_tag, pl[j].brdf_param);
}

constexpr size_t kLiveBytes = (size_t)kStatShards * kStatStride * sizeof(unsigned long long);     // sharded statistics (ptmi_kernels.h)
constexpr int kGlassBatchDefault = 1;      // PTMI_OPT_GLASS_BATCH = 0 (automatic): parking off -- see the measurements in DESIGN.md 5.5
constexpr size_t kItersBytes = (size_t)kStatShards * 2 * kStatStride * sizeof(unsigned int);

RayQueue carve_queue(void *block, size_t capacity, int which)
{
    RayQueue q;
    q.base = static_cast<uint32_t *>(block) + (size_t)which * kRayQueueWords * capacity;
    q.capacity = (unsigned int)capacity;
    return q;
}

// The schedule of the cost-ordered dispatch: `launches` = launches made so far with one (camera, scene, shape, limit, algorithm).
// The order is REBUILT from the recorded costs before launch 1, 2, 4, 8, ... -- and never again once the limit is reached: the state stops
// there, it is a power of two itself, and a rebuild bumps the order's generation, which would make the stream form re-run its primary kernel
// on every later call of a standing camera.  A launch RECORDS its costs only if the NEXT launch rebuilds -- launch 0, 1, 3, 7, 15, ... (the
// sums stay far from 2^32) -- since round 5: a recorded cost is an atomic per wave, or per item in the stream form, on words that all XCDs share
// (32 bytes of write traffic each, tools/traffic_terms.py), and between two rebuilds nobody reads them; every rebuild still sees one more
// launch than the one before it, and a standing camera's steady state records nothing.
int order_schedule(int launches, int stream_form, int *rebuild, int *record)
{
    const int limit = stream_form ? (1 << 11) : (1 << 20);
    const bool below = launches >= 0 && launches < limit;
    if (rebuild) *rebuild = (below && launches > 0 && (launches & (launches - 1)) == 0) ? 1 : 0;
    if (record) *record = (below && launches + 1 < limit && ((launches + 1) & launches) == 0) ? 1 : 0;
    return below ? launches + 1 : limit;
}

// The passes of the stream form's split kernel: which samples of its pixels pass p renders (first[p] .. first[p + 1]).
// A pass is one item per start hit, taken by whatever lane is free, and the launch ends as its LAST items do: with uniform
// passes of 16 samples the waves of a 1080p / 64-spp glass launch ended anywhere between 75 % and 98 % of it (a wave holds 64
// items of very different weight when the tickets run out) -- an eighth of the chip's wave-time idle.  So the passes are GRADED
// (guided self-scheduling): at most `per_item` samples -- the size that gives a lane its ~16 items -- and never more than a
// fraction of what is still to come, chosen so that an item five times the mean weight, taken when its pass begins, is over
// before the remaining passes are: the last passes are a few samples (not fewer than a floor: below), and the end of the launch is as long as their trees.
// Which sample belongs to which pass changes no seed (snapshots) and no ray; only the order of a pixel's float additions,
// which the stream form with GLASS does not define anyway.  Returns the number of passes (<= kMaxStreamPasses).
constexpr int kMaxStreamPasses = 64;
constexpr int kMaxStreamRaysPerPixel = 64;   // PTMI_OPT_STREAM_CAPACITY's upper end, and how far a call grows its overflow streams before it counts drops
int stream_schedule(int n_spp, unsigned long long n_px, unsigned long long lanes, int batch, bool graded, int first[kMaxStreamPasses + 1])
{
    first[0] = 0;
    if (n_spp <= 0 || n_px == 0 || lanes == 0) { first[1] = n_spp > 0 ? n_spp : 0; return 1; }
    int per_item = batch;
    if (per_item <= 0) {                                      // a lane should see ~16 items
        const double items_wanted = 16.0 * (double)lanes;
        int passes = (int)(items_wanted / (double)n_px + 0.999);
        if (passes < 1) passes = 1;
        if (passes > kMaxStreamPasses) passes = kMaxStreamPasses;
        per_item = (n_spp + passes - 1) / passes;
        if (per_item < 8) per_item = n_spp < 8 ? n_spp : 8;
    }
    if (per_item > n_spp) per_item = n_spp;
    if ((n_spp + per_item - 1) / per_item > kMaxStreamPasses) per_item = (n_spp + kMaxStreamPasses - 1) / kMaxStreamPasses;
    int passes = 0, done = 0;
    if (!graded) {
        while (done < n_spp) { done = done + per_item < n_spp ? done + per_item : n_spp; first[++passes] = done; }
        return passes;
    }
    const double f = ((double)n_px / (double)lanes) / 5.0;     // items per lane and pass / the weight of a heavy item
    double frac = f / (1.0 + f);
    frac = frac < 0.2 ? 0.2 : (frac > 0.5 ? 0.5 : frac);
    // ... and never BELOW a floor: a pass costs every start hit a ticket's share, a refill (record + snapshot into the lane's LDS column) and an
    // item end (three float atomics), which a single sample does not repay -- glass 1080p / 64 spp with the passes ending 4, 2, 1, 1: 7.70-7.77 ms;
    // ending 6, 6, 4: 7.60-7.61 (C5's 512 spp: no difference either way).  What the floor leaves over joins the pass before it where the item
    // cap allows; the sizes are handed out longest first.
    const int floor_size = per_item >= 16 ? 4 : (per_item >= 8 ? 2 : 1);
    int sizes[kMaxStreamPasses];
    while (done < n_spp) {
        const int left = n_spp - done;
        int size = (int)((double)left * frac + 0.999999);
        if (size > per_item) size = per_item;
        if (size < floor_size) size = floor_size;
        const int passes_left = kMaxStreamPasses - passes;     // never more than kMaxStreamPasses passes: a pass takes at least its share of what is left
        const int share = (left + passes_left - 1) / passes_left;
        if (size < share) size = share;
        if (size > left) size = left;
        if (left - size < floor_size && (left <= per_item || passes_left == 1)) size = left;      // (no pass below the floor at the very end)
        done += size; sizes[passes++] = size;
    }
    std::sort(sizes, sizes + passes, [](int x, int y) { return x > y; });
    for (int k = 0, at = 0; k < passes; ++k) { at += sizes[k]; first[k + 1] = at; }
    return passes;
}

// In which order the split kernel hands out its tickets (ItemArgs.group_first; PTMI_OPT_STREAM_PASS_GROUPS).  Pass by pass -- every region of the
// start-hit list in pass 0, then every region in pass 1 ... -- a region's 64-byte records, its snapshots' lines and the colour lines of its pixels
// come from HBM once per PASS: 1 482 MB of fetches per 1080p / 64-spp call of the glass scene (six passes), against 190 MB of records.  In GROUPS
// of consecutive passes, each group region by region, the items of a start hit that belong to one group are taken within microseconds of each
// other from one ticket queue by waves behind one L2, and only the first reads HBM (tools/traffic_terms.py, profiles/r05_traffic_terms.json:
// 1 482 -> 959 MB in pairs, 757 in threes, 546 as one group, the launch's time within +- 0.5 %).  What a group must not do is undo the GRADING:
// as ONE group the launch ends with the cheapest regions' items of EVERY pass, the long ones included -- with few regions per wave (a C5 part:
// 2.6 per pass) that is + 2 % and a quarter more spilled children.  Hence, automatically: a group is a run of passes of EQUAL size (16, 16, 16 | 8 |
// 4, 4 at 1080p / 64 spp; 74 x 5 | 50 | 32 | 21 | 14 | 9 | 6, 6 | 4 on a C5 part) -- within it every item is as long as every other, so the order
// of its tickets cannot lengthen the end of the launch.
// option 0 = automatic, 1 = every pass on its own, k in [2, 64] = the last k passes as one group, 100 + g = groups of g passes all the way (what
// does not divide goes first, pass by pass).  Returns the number of groups; table[0 .. groups] = the first pass of every group, and `passes`.
int pass_group_table(int option, const int *first, int passes, int table[kMaxStreamPasses + 1])
{
    int groups = 0;
    table[0] = 0;
    auto close = [&](int next_first) { table[++groups] = next_first; };
    if (passes < 2 || option == 1) {
        for (int p = 1; p <= passes; ++p) close(p);
    } else if (option == 0) {
        for (int p = 1; p <= passes; ++p)
            if (p == passes || first[p + 1] - first[p] != first[p] - first[p - 1]) close(p);
    } else if (option >= 100) {
        int g = option - 100 < passes ? option - 100 : passes;
        if (g < 1) g = 1;
        for (int p = 1; p <= passes % g; ++p) close(p);
        for (int p = passes % g + g; p <= passes; p += g) close(p);
    } else {
        const int g = option < passes ? option : passes;
        for (int p = 1; p <= passes - g; ++p) close(p);
        close(passes);
    }
    return groups;
}

// `render Streams` as a stream ("wavefront" form, ptmi_stream_*.hip).  Every sample of a pixel shoots the same primary ray: its
// hit is evaluated once per call into the start-hit list (regions in dispatch order), then ONE persistent launch renders all
// samples of the call:
//   * scenes whose rays never split (default batch): streams_pixels_kernel -- a lane owns a pixel for the launch, no atomics,
//     bit-identical to the per-pixel kernel under both seed rules; nothing is read back, the call is asynchronous;
//   * GLASS (or PTMI_OPT_STREAM_BATCH > 0): streams_split_kernel -- items of (start hit, sample range), children through the
//     waves' LDS rings.  The predicate `null state` (Trace.hs:166-170) is the overflow stream's length: read back ONCE, after
//     the launch; only if children really travelled through HBM does the host play `awhile`, one launch per overflow level.
int render_streams_wavefront(ptmi_ctx *c, RenderArgs &a, int n_spp, const ptmi_camera &camera)
{
    const size_t n = (size_t)a.rows_local * a.width;
    if (n == 0 || n_spp <= 0) return PTMI_OK;
    if (n > 0x3ffffff0ull) return fail(c, PTMI_ELIMIT, "image too large for the stream form of Streams");   // 32-bit byte offsets into the planes
    const bool ordered = !c->has_glass && (c->opt_batch == 0 || a.seed_from_result);
    const unsigned int n_regions = streams_regions(a.width, a.rows_local);
    const unsigned int region_slots = ordered ? 64u : 128u;    // a glass primary hit contributes up to two start hits
    const size_t hit_slots = (size_t)n_regions * region_slots;
    if (hit_slots > 0xfffffff0ull) return fail(c, PTMI_ELIMIT, "image too large for the stream form of Streams");
    if (hit_slots > c->hit_capacity || n_regions > c->hit_regions) {
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        if (c->hit_block) { (void)hipFree(c->hit_block); c->hit_block = nullptr; c->hit_capacity = 0; }
        if (c->d_hit_counts) { (void)hipFree(c->d_hit_counts); c->d_hit_counts = nullptr; c->hit_regions = 0; }
        if (c->d_hit_missed) { (void)hipFree(c->d_hit_missed); c->d_hit_missed = nullptr; }
        c->hit_list_valid = false;
        PTMI_HIP(c, hipMalloc(&c->hit_block, (size_t)(kHitListWords + 1) * hit_slots * 4));      // the records, and behind them one key word per slot
        c->hit_capacity = hit_slots;
        PTMI_HIP(c, hipMalloc(&c->d_hit_counts, (size_t)n_regions * sizeof(unsigned int)));
        PTMI_HIP(c, hipMalloc(&c->d_hit_missed, (size_t)n_regions * sizeof(unsigned long long)));
        c->hit_regions = n_regions;
    }
    if (!c->d_qcount) PTMI_HIP(c, hipMalloc(&c->d_qcount, (size_t)kLvWords * sizeof(unsigned int)));
    HitList hits;
    hits.base = static_cast<uint32_t *>(c->hit_block);
    hits.slot_key = hits.base + (size_t)kHitListWords * c->hit_capacity;
    hits.counts = c->d_hit_counts;
    hits.missed = c->d_hit_missed;
    hits.region_slots = region_slots;
    hits.n_regions = n_regions;
    // the statistics accumulate on the device over the whole call; cursors start from zero
    PTMI_HIP(c, hipMemsetAsync(c->d_qcount, 0, (size_t)kLvWords * sizeof(unsigned int), c->stream));
    auto cursor_of = [&](int level) { return (size_t)(kLvCursor + kLvPerLevel * (level % kLvMaxLevels)) * kCounterStride; };
    ptmi_ctx::HitKey key{};
    key.cam = camera; key.scene_version = c->scene_version; key.order_generation = a.quad_order ? c->order_generation : 0;
    const int key_dims[8] = {a.width, a.height, a.rows_local, a.stripe_rows, a.n_parts, a.part, a.quad_order ? 1 : 0, 0};
    std::memcpy(key.dims, key_dims, sizeof key_dims);
    key.region_slots = region_slots; key.cap_allows_split = a.stream_step_cap >= 3 ? 1 : 0; key.planes_r = nullptr;
    const bool list_kept = c->hit_list_valid && std::memcmp(&key, &c->hit_key, sizeof key) == 0;
    // A list built by THIS call counts as kept by later ones only if this call gets to its end: what the host keeps beside it (the split pixels'
    // count, read back below) is only then the list's.  Any error return on the way leaves the list invalid.
    struct ListGuard { ptmi_ctx *c; bool fresh; bool done = false; ~ListGuard() { if (fresh && !done) c->hit_list_valid = false; } } guard{c, !list_kept};
    if (!list_kept) {
        c->hit_list_valid = false;
        PTMI_HIP(c, launch_streams_primary(a, hits, c->d_qcount, c->stream));
        c->hit_key = key; c->hit_list_valid = true;
    }
    ItemArgs it{};
    it.hits = hits;
    it.n_positions = n_regions / 4u;
    it.passes = 1; it.group_first = nullptr; it.groups = 0;
    it.n_slots = (unsigned int)hit_slots;
    it.stats = c->d_qcount;
    auto tickets_of = [&](int launch) { return c->d_qcount + (size_t)(kLvTickets + 8 * launch) * kCounterStride; };
    const int cus = c->cus > 0 ? c->cus : 256;
    if (ordered) {
        // One lane renders a pixel's samples in order, so an item is a serial chain and the end of a launch is as long as its last
        // items: with three items per lane (1080p) a quarter of the wave-time of the launch lay after the first wave had ended.
        // The samples are therefore cut into ordered passes INSIDE the one launch (streams_pixels_kernel): a pixel's next pass is
        // handed out once its previous one has been published.
        const unsigned int grid_full = (unsigned int)(cus * 4 * streams_pixels_waves());
        const unsigned long long lanes = 64ull * grid_full;
        int passes = 1;
        if (c->opt_ordered_passes > 0) passes = c->opt_ordered_passes;    // (1 = off: one pass, no hand-off between waves inside the launch)
        else if (c->opt_batch > 0) passes = (n_spp + c->opt_batch - 1) / c->opt_batch;     // PTMI_OPT_STREAM_BATCH: samples per item
        else if (c->opt_pass_handoff == 0 && n < 3ull * lanes && n_spp >= 256) passes = n_spp / 64 < 8 ? n_spp / 64 : 8;
        // (automatic only with the FENCED hand-off -- release / acquire at agent scope once per region and pass, what the memory model
        // promises; the fence-free hand-off of rounds 3-5, PTMI_OPT_PASS_HANDOFF = 1, is measured valid, not promised, and never chosen
        // without the caller's explicit PTMI_OPT_ORDERED_PASSES)
        // (few, LONG items per lane: items of >= 64 samples, at most 8 passes.  The kernel of the ordered passes is 3 % slower per trip
        // than the one-pass kernel, and an item costs its refill and its seven stores.  1080p, S16, ms with 1 / 2 / 4 / 8 / 16 / 32 passes:
        // 64 spp 4.46 / 4.62 / 4.63 / 4.84 / 4.83 / 4.86; 256 spp 17.49 / 17.24 / 16.99 / 17.11 / 17.70 / 18.96; 1024 spp 69.7 / 68.1 / 66.0 /
        // 64.9 / 65.3 / 66.8.  With the per-pixel tail below, which only a one-pass launch has, the whole 1080p image -- 4.5 pixels per lane --
        // is better off in one pass: 256 spp 16.60 against 17.19 ms with four passes, 512 spp 32.86 against 33.48, 1024 spp equal; one of 8
        // parts of a 4K image -- 2.3 pixels per lane -- is not: 1024 spp 38.98 against 36.10 with eight passes.  Hence 3 pixels per lane.)
        if (passes > 64) passes = 64;
        const int most = n_spp / streams_min_pass_samples();   // a pass holds at least that many samples
        if (passes > most) passes = most;
        if (passes < 1) passes = 1;
        it.passes = passes;                                // (ordered passes wait for each other: every pass on its own, no group table)
        it.fenced = c->opt_pass_handoff == 0 ? 1 : 0;
        if (passes > 1 && n_regions >= (1u << 26)) return fail(c, PTMI_ELIMIT, "image too large for ordered passes (2^26 regions)");
        it.chunk_cursor = tickets_of(0);
        // THE TAIL.  A lane renders a pixel's whole sample chain, so the persistent launch ends as its last items do: its waves end between
        // 70 and 100 % of it.  The cheapest quads of the dispatch order -- the order kernel marks where they begin, on the device -- are
        // therefore left to the per-pixel chain kernel, launched beside the persistent kernel on a low-priority stream: its waves (one
        // tile each) take the slots the persistent waves leave as they end.  Every pixel is rendered by exactly one of the two kernels,
        // by the same arithmetic: no result depends on where the boundary lies.
        // (Only while a pixel's samples are ONE item: where they are cut into ordered passes the end of the launch is short already, and a
        // tail wave would render its tile's many samples in one piece -- 1080p / 256 spp: 16.99 ms with four passes, 17.86 with a tail beside them.)
        const unsigned int *tail = (passes == 1 && c->opt_tail_permille != 0 && a.quad_order && c->d_tail_start && quad_positions(a.width, a.rows_local) > 0) ? c->d_tail_start : nullptr;
        if (tail && !c->tail_stream) {
            int least = 0, greatest = 0;
            PTMI_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            PTMI_HIP(c, hipStreamCreateWithPriority(&c->tail_stream, hipStreamNonBlocking, least));
            PTMI_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            PTMI_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        it.tail_start = tail;
        PTMI_HIP(c, launch_streams_advance_missed(a, hits, n_spp, tail, c->stream));
        PTMI_HIP(c, hipMemsetAsync(c->d_iters, 0, kItersBytes, c->stream));
        if (tail) {
            PTMI_HIP(c, hipEventRecord(c->ev_fork, c->stream));
            PTMI_HIP(c, hipStreamWaitEvent(c->tail_stream, c->ev_fork, 0));
        }

        if (passes > 1) {
            if (n_regions > c->region_done_words) {
                PTMI_HIP(c, hipStreamSynchronize(c->stream));
                if (c->d_region_done) { (void)hipFree(c->d_region_done); c->d_region_done = nullptr; c->region_done_words = 0; }
                PTMI_HIP(c, hipMalloc(&c->d_region_done, (size_t)n_regions * sizeof(unsigned int)));
                c->region_done_words = n_regions;
            }
            PTMI_HIP(c, hipMemsetAsync(c->d_region_done, 0, (size_t)n_regions * sizeof(unsigned int), c->stream));
            it.region_done = c->d_region_done;
        }
        unsigned int grid = grid_full;
        const unsigned long long tickets = (unsigned long long)n_regions * (unsigned long long)passes;
        if (grid > tickets) grid = (unsigned int)(tickets < 1 ? 1 : tickets);
        PTMI_HIP(c, launch_streams_pixels(a, it, grid, c->stream));
        if (tail) {
            PTMI_HIP(c, launch_render_streams_tail(a, tail, c->tail_stream));
            PTMI_HIP(c, hipEventRecord(c->ev_join, c->tail_stream));
            PTMI_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
        guard.done = true;
        return PTMI_OK;
    }

    // ---- rays may split, or the samples of a pixel run as unordered items
    if (quad_positions(a.width, a.rows_local) > (1u << 21)) a.quad_cost = nullptr;      // (the item record keeps the quad in 21 bits: no costs beyond 2^29 pixels)
    unsigned int grid = (unsigned int)(cus * 4 * streams_split_waves());
    // the passes: graded items, long first and short ones last (stream_schedule above); PTMI_OPT_STREAM_BATCH caps an item's samples
    int first[kMaxStreamPasses + 1];
    int passes = stream_schedule(n_spp, n, 64ull * grid, c->opt_batch, c->opt_graded != 0, first);
    {   // the seed snapshots are passes x record slots x 16 bytes: within an eighth of the device's memory, merging the LAST passes if need be
        const size_t budget = c->opt_snapshot_mb > 0 ? (size_t)c->opt_snapshot_mb << 20 : c->device_memory / 8;
        while (passes > 1 && (size_t)passes * hit_slots * sizeof(uint4) > budget) { --passes; first[passes] = n_spp; }
        if ((size_t)passes * hit_slots * sizeof(uint4) > budget) return fail(c, PTMI_ELIMIT, "seed snapshots of the stream form would exceed their budget (PTMI_OPT_SNAPSHOT_BUDGET_MB; default: an eighth of the device's memory)");
    }
    const unsigned long long n_tickets = (unsigned long long)n_regions * (unsigned long long)passes;      // a ticket = a region of the start-hit list in one pass
    if (n_tickets > 0x7fffffffull) return fail(c, PTMI_ELIMIT, "too many items for the stream form of Streams");
    if (grid > n_tickets) grid = (unsigned int)n_tickets;
    const size_t snap_bytes = (size_t)passes * hit_slots * sizeof(uint4);
    if (snap_bytes > c->snapshot_bytes) {
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_snapshots) { (void)hipFree(c->d_snapshots); c->d_snapshots = nullptr; c->snapshot_bytes = 0; }
        PTMI_HIP(c, hipMalloc(&c->d_snapshots, snap_bytes));
        c->snapshot_bytes = snap_bytes;
    }
    // the overflow streams: rays per pixel (PTMI_OPT_STREAM_CAPACITY, or what an earlier call grew them to) x pixels, at least every wave's static block
    const unsigned int first_block = streams_first_block();
    const unsigned int level_grid_max = (unsigned int)(cus * 4 * 6);
    const size_t floor_slots = (size_t)(grid > level_grid_max ? grid : level_grid_max) * first_block + 256;
    int cap_rays = c->grown_capacity > c->opt_capacity ? c->grown_capacity : c->opt_capacity;
    auto size_streams = [&](int rays_per_pixel) -> int {
        size_t need = n * (size_t)rays_per_pixel;
        if (need < floor_slots) need = floor_slots;
        if (need > 0xfffffff0ull) return fail(c, PTMI_ELIMIT, "image too large for the stream form of Streams");
        if (need > c->queue_capacity) {
            PTMI_HIP(c, hipStreamSynchronize(c->stream));
            if (c->queue_block) { (void)hipFree(c->queue_block); c->queue_block = nullptr; c->queue_capacity = 0; }
            PTMI_HIP(c, hipMalloc(&c->queue_block, 2 * (size_t)kRayQueueWords * need * 4));
            c->queue_capacity = need;
        }
        return PTMI_OK;
    };
    if (int rc = size_streams(cap_rays)) return rc;
    size_t capacity = c->queue_capacity;
    RayQueue q[2] = {carve_queue(c->queue_block, capacity, 0), carve_queue(c->queue_block, capacity, 1)};
    const size_t spill_records = (size_t)grid * streams_spill_records();
    if (spill_records > c->spill_capacity) {
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        if (c->spill_block) { (void)hipFree(c->spill_block); c->spill_block = nullptr; c->spill_capacity = 0; }
        PTMI_HIP(c, hipMalloc(&c->spill_block, (size_t)kRayQueueWords * spill_records * 4));
        c->spill_capacity = spill_records;
    }
    std::vector<unsigned int> base((size_t)kLvMaxLevels, 0u);   // per level (mod kLvMaxLevels): where its reserved blocks start

    int group_table[kMaxStreamPasses + 1];
    const int groups = pass_group_table(c->opt_pass_groups, first, passes, group_table);
    {   // the pass table and the group table on the device (one block: kMaxStreamPasses + 1 entries each): rewritten only when they change
        std::vector<int> table(first, first + passes + 1);
        table.resize((size_t)kMaxStreamPasses + 1, 0);
        table.insert(table.end(), group_table, group_table + groups + 1);
        if (!c->d_pass_first) PTMI_HIP(c, hipMalloc(&c->d_pass_first, 2 * (size_t)(kMaxStreamPasses + 1) * sizeof(int)));
        if (table != c->pass_first_host) {
            PTMI_HIP(c, hipStreamSynchronize(c->stream));      // (an earlier launch may still be reading the old tables)
            PTMI_HIP(c, hipMemcpy(c->d_pass_first, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
            c->pass_first_host = table;
        }
    }
    PTMI_HIP(c, launch_streams_seeds(a.planes, hits, static_cast<uint4 *>(c->d_snapshots), (long long)n, passes, c->d_pass_first, n_spp, c->stream));
    it.passes = passes; it.pass_first = c->d_pass_first;
    it.group_first = groups < passes ? c->d_pass_first + (kMaxStreamPasses + 1) : nullptr;     // (every pass on its own needs no table)
    it.groups = groups;
    // GLASS hits wait in their lanes until that many are pending in the wave (measured: DESIGN.md 5.5); nothing to wait for without GLASS
    it.glass_batch = !c->has_glass ? 0 : (c->opt_glass_batch > 0 ? c->opt_glass_batch : kGlassBatchDefault);
    it.chunk_cursor = tickets_of(0);
    it.seed_snapshots = static_cast<const uint4 *>(c->d_snapshots);
    it.spill = carve_queue(c->spill_block, c->spill_capacity, 0);
    it.out_count = c->d_qcount + cursor_of(0);
    it.out_base = grid * first_block;
    it.emitted = it.out_count + 2 * kCounterStride;
    it.may_emit = c->has_glass ? 1 : 0;

    std::vector<unsigned int> raw((size_t)kLvWords);
    auto emitted_of = [&](int level) {                        // children the level stored: the sum of its shards (read back into `raw`)
        unsigned long long total = 0;
        for (int k = 0; k < kLvEmitShards; ++k) total += raw[cursor_of(level) + (size_t)(2 + k) * kCounterStride];
        return total;
    };
    auto read_counters = [&](int levels) -> int {           // the statistics and the lines of the first `levels` levels
        int lines = kLvCursor + kLvPerLevel * levels;
        if (lines > kLvTickets) lines = kLvTickets;
        PTMI_HIP(c, hipMemcpyAsync(raw.data(), c->d_qcount, (size_t)lines * kCounterStride * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        return PTMI_OK;
    };
    // `expand` (Trace.hs:284-293) makes its vectors as long as the step needs; the overflow streams here have a capacity.  A call that WOULD drop
    // children is therefore redone with longer streams: the three colour planes -- all a launch changes that the next attempt reads; the seeds moved
    // before the launch, the snapshots stand -- are copied aside first (25 MB at 1080p: ~ 10 us in stream order), and when the counters say that
    // children found no room the planes are put back, the streams doubled (up to kMaxStreamRaysPerPixel rays per pixel, or to what the device
    // still has) and the launch and its levels run again.  The capacity a call reached is kept for later calls.  Only with GLASS: nothing else emits.
    const size_t plane_bytes = n * sizeof(float);
    // (only while a redo is possible: with the streams at kMaxStreamRaysPerPixel already the drops would stand, and nothing is copied aside.
    // The recorded quad costs of the launch are part of what a redo must take back: the failed attempt's items added theirs.)
    const size_t cost_bytes = a.quad_cost ? (size_t)quad_positions(a.width, a.rows_local) * sizeof(unsigned int) : 0;
    const bool can_redo = c->has_glass && cap_rays < kMaxStreamRaysPerPixel;
    if (can_redo) {
        if (3 * plane_bytes + cost_bytes > c->colour_backup_bytes) {
            PTMI_HIP(c, hipStreamSynchronize(c->stream));
            if (c->colour_backup) { (void)hipFree(c->colour_backup); c->colour_backup = nullptr; c->colour_backup_bytes = 0; }
            PTMI_HIP(c, hipMalloc(&c->colour_backup, 3 * plane_bytes + cost_bytes));
            c->colour_backup_bytes = 3 * plane_bytes + cost_bytes;
        }
        char *bk = static_cast<char *>(c->colour_backup);
        PTMI_HIP(c, hipMemcpyAsync(bk, a.planes.r, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        PTMI_HIP(c, hipMemcpyAsync(bk + plane_bytes, a.planes.g, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        PTMI_HIP(c, hipMemcpyAsync(bk + 2 * plane_bytes, a.planes.b, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        if (cost_bytes) PTMI_HIP(c, hipMemcpyAsync(bk + 3 * plane_bytes, a.quad_cost, cost_bytes, hipMemcpyDeviceToDevice, c->stream));
    } else if (c->colour_backup && !c->has_glass) {          // the scene lost its GLASS: the three planes' worth of memory goes back
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->colour_backup); c->colour_backup = nullptr; c->colour_backup_bytes = 0;
    }
    unsigned long long overflowed = 0;
    for (int attempt = 0;; ++attempt) {
        it.out = q[0];
        base[0] = it.out_base;
        PTMI_HIP(c, launch_streams_split(a, it, grid, c->stream));
        // stream_iterations lives in the per-pixel kernels' sharded counter: shard 0 carries this form's figure, copied on the device
        if (attempt == 0) PTMI_HIP(c, hipMemsetAsync(c->d_iters, 0, kItersBytes, c->stream));
        if (int rc = read_counters(1)) return rc;
        overflowed = 0;
        // `null state` (Trace.hs:166-170): the loop goes on while the last level left children in its overflow stream (a level
        // cuts the rays the step cap forbids as it reads them, and counts them)
        for (int level = 0; c->has_glass && emitted_of(level) > 0u;) {
            overflowed += emitted_of(level);
            const size_t cursor = (size_t)raw[cursor_of(level)] + base[(size_t)(level % kLvMaxLevels)];
            const size_t items = cursor < capacity ? cursor : capacity;
            ++level;
            LevelArgs lv{};
            lv.in = q[(level + 1) & 1]; lv.out = q[level & 1];
            lv.stats = c->d_qcount;
            lv.in_count = c->d_qcount + cursor_of(level - 1);
            lv.in_base = base[(size_t)((level - 1) % kLvMaxLevels)];
            lv.out_count = c->d_qcount + cursor_of(level);
            lv.emitted = lv.out_count + 2 * kCounterStride;
            lv.may_emit = 1;
            const size_t chunks = (items + 63) / 64;
            const unsigned int lgrid = (unsigned int)(chunks < 1 ? 1 : (chunks > level_grid_max ? level_grid_max : chunks));
            lv.out_base = lgrid * first_block;
            base[(size_t)(level % kLvMaxLevels)] = lv.out_base;
            // the counter words of this level: long idle when they come round again
            PTMI_HIP(c, hipMemsetAsync(lv.out_count, 0, (size_t)kLvPerLevel * kCounterStride * sizeof(unsigned int), c->stream));
            PTMI_HIP(c, launch_streams_level(a, lv, lgrid, c->stream));
            if (int rc = read_counters(level + 1)) return rc;
        }
        const unsigned int dropped = raw[(size_t)kLvDropped * kCounterStride];
        if (!can_redo || dropped == 0u || cap_rays >= kMaxStreamRaysPerPixel) break;
        // ---- children were dropped: longer streams, the planes put back, once more.  "Longer" must be true: the block may be larger than
        // rays-per-pixel x pixels says (the floor of the waves' static blocks, a block left over from a larger image, a lowered
        // PTMI_OPT_STREAM_CAPACITY), and a redo into streams of the same length drops the same children again -- so the capacity doubles until
        // it asks for more than is there; if even kMaxStreamRaysPerPixel does not, the drops stand, counted.
        int grown = cap_rays;
        size_t need = 0;
        do {
            grown = grown * 2 < kMaxStreamRaysPerPixel ? grown * 2 : kMaxStreamRaysPerPixel;
            need = n * (size_t)grown;
            if (need < floor_slots) need = floor_slots;
        } while (need <= capacity && grown < kMaxStreamRaysPerPixel);
        if (need <= capacity) { c->grown_capacity = grown; break; }
        if (need > 0xfffffff0ull) break;
        if (need > c->queue_capacity) {
            void *bigger = nullptr;
            if (hipMalloc(&bigger, 2 * (size_t)kRayQueueWords * need * 4) != hipSuccess) { (void)hipGetLastError(); break; }    // the device has no more: the drops stand, counted
            (void)hipFree(c->queue_block);                   // (the stream is idle: read_counters synchronised it)
            c->queue_block = bigger; c->queue_capacity = need;
        }
        cap_rays = grown; c->grown_capacity = grown;
        capacity = c->queue_capacity;
        q[0] = carve_queue(c->queue_block, capacity, 0); q[1] = carve_queue(c->queue_block, capacity, 1);
        char *bk = static_cast<char *>(c->colour_backup);
        PTMI_HIP(c, hipMemcpyAsync(a.planes.r, bk, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        PTMI_HIP(c, hipMemcpyAsync(a.planes.g, bk + plane_bytes, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        PTMI_HIP(c, hipMemcpyAsync(a.planes.b, bk + 2 * plane_bytes, plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        if (cost_bytes) PTMI_HIP(c, hipMemcpyAsync(a.quad_cost, bk + 3 * plane_bytes, cost_bytes, hipMemcpyDeviceToDevice, c->stream));
        // every counter of the call starts over -- but for the split pixels' count, which the primary kernel wrote and which is not run again
        const unsigned int split_pixels = raw[(size_t)kLvSplitPixels * kCounterStride];
        PTMI_HIP(c, hipMemsetAsync(c->d_qcount, 0, (size_t)kLvWords * sizeof(unsigned int), c->stream));
        PTMI_HIP(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_qcount + (size_t)kLvSplitPixels * kCounterStride), (int)split_pixels, 1, c->stream));
        std::fill(base.begin(), base.end(), 0u);
    }
    c->rays_overflowed += overflowed;
    for (int k = 0; k < kLvLiveShards; ++k) c->live_host += raw[(size_t)(kLvLive + k) * kCounterStride];
    // the two children of every glass primary hit whose split is cached in the start list: counted here, per sample
    if (!list_kept) c->hit_split_pixels = raw[(size_t)kLvSplitPixels * kCounterStride];
    c->live_host += 2ull * c->hit_split_pixels * (uint64_t)n_spp;
    c->rays_dropped += raw[(size_t)kLvDropped * kCounterStride];
    c->rays_truncated += raw[(size_t)kLvCut * kCounterStride];
    c->rays_spilled += raw[(size_t)kLvSpilled * kCounterStride];
    unsigned int longest = raw[(size_t)kLvDeepest * kCounterStride];            // stream_iterations: the deepest step of this call
    if (c->hit_split_pixels && longest < 2) longest = 2;                        // the children of the cached glass primary hits: traceStep 2
    if (longest == 0) longest = 1;                                              // every primary ray missed: one traceStep all the same
    PTMI_HIP(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_iters), (int)longest, 1, c->stream));
    guard.done = true;
    return PTMI_OK;
}

int launch_render(ptmi_ctx *c, const Planes &planes, const ptmi_camera *camera, int algorithm,
                  int bounce_limit, int n_spp, int width, int height, int rows_local,
                  int stripe_rows, int n_parts, int part, const int64_t *sx, const int64_t *sy)
{
    RenderArgs a{};
    a.cam = make_uniforms(*camera, width, height);
    a.scene.packed = c->d_scene; a.scene.n_spheres = c->n_spheres; a.scene.n_planes = c->n_planes;
    a.planes = planes;
    a.screen_x = sx; a.screen_y = sy;
    a.width = width; a.height = height; a.rows_local = rows_local;
    a.stripe_rows = stripe_rows; a.n_parts = n_parts; a.part = part;
    a.bounce_limit = bounce_limit; a.n_spp = n_spp;
    a.cus = c->cus;
    a.live_counter = c->d_live; a.work_counter = c->d_work; a.stream_iterations = c->d_iters;
    a.stream_step_cap = c->opt_step_cap; a.seed_from_result = effective_seed_rule(c) == PTMI_SEED_FROM_RESULT;
    a.stream_counters = c->d_stream_counters;
    const bool stream_form = uses_stream_form(c, algorithm, n_parts, n_spp);
    // Cost-ordered dispatch (ptmi_device.h: lane_pixel): launches with one (camera, scene, shape, limit, algorithm)
    // record what every quad of tiles costs; later launches with the same key dispatch the most expensive quads first
    // (sorted on the device).  Everything is enqueued on the stream; results do not depend on it.
    const bool per_pixel_kernel = !stream_form;
    int next_order_state = c->order_state;
    // (the stream form orders its start-hit list the same way, under a key of its own; its items record their costs)
#ifdef PTMI_STREAM_NO_ORDER
    const bool stream_order = false;
#else
    const bool stream_order = quad_positions(width, rows_local) > 0 && !sx;
#endif
    if (stream_form ? stream_order : uses_quad_order(a, algorithm == PTMI_INLINE, c->variant)) {
        const unsigned int n_quads = quad_positions(width, rows_local);
        if (n_quads > c->quad_capacity) {
            if (c->d_quad_cost) { PTMI_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_quad_cost); (void)hipFree(c->d_quad_order); (void)hipFree(c->d_quad_class); }
            c->d_quad_cost = c->d_quad_order = c->d_quad_class = nullptr; c->quad_capacity = 0; c->order_state = 0; ++c->order_generation;
            PTMI_HIP(c, hipMalloc(&c->d_quad_cost, n_quads * sizeof(unsigned int)));
            PTMI_HIP(c, hipMalloc(&c->d_quad_order, n_quads * sizeof(unsigned int)));
            PTMI_HIP(c, hipMalloc(&c->d_quad_class, n_quads * sizeof(unsigned int)));
            c->quad_capacity = n_quads;
        }
        ptmi_ctx::OrderKey key{};
        key.cam = *camera; key.scene_version = c->scene_version;
        const int dims[8] = {width, height, rows_local, stripe_rows, n_parts, part, bounce_limit, stream_form ? 2 : algorithm};
        std::memcpy(key.dims, dims, sizeof dims);
        if (std::memcmp(&key, &c->order_key, sizeof key) != 0) { c->order_key = key; c->order_state = 0; ++c->order_generation; }
        // order_state counts the launches made with this key.  The launches before a rebuild add their costs (a 1-spp launch says little
        // on its own: the compat entry renders one sample per call); the order is rebuilt before launch 1, 2, 4, 8, ...
        const int launches = c->order_state;
        int rebuild = 0, record = 0;
        const int state_after = order_schedule(launches, stream_form ? 1 : 0, &rebuild, &record);
        if (!c->d_tail_start) PTMI_HIP(c, hipMalloc(&c->d_tail_start, 64));
        if (launches == 0) {
            PTMI_HIP(c, hipMemsetAsync(c->d_quad_cost, 0, n_quads * sizeof(unsigned int), c->stream));
            PTMI_HIP(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_tail_start), (int)n_quads, 1, c->stream));   // no tail yet
        } else if (rebuild) {
            // (the stream form: the order kernel also says where the order's cheap end begins -- render_streams_wavefront.  How much of the
            // recorded cost that end may hold: 15 % where a lane sees four pixels or more, 40 % where it sees fewer -- 1280x720 / 64 spp: 2.34 ms
            // with 15 %, 2.16 with 40 %, chain kernel 1.96; 1080p: 4.24 / 4.21.)
            const unsigned long long lanes_full = 64ull * (unsigned long long)(c->cus > 0 ? c->cus : 256) * 4ull * (unsigned long long)streams_pixels_waves();
            const unsigned int tail_permille = c->opt_tail_permille >= 0 ? (unsigned int)c->opt_tail_permille
                                             : ((unsigned long long)width * (unsigned long long)rows_local < 4ull * lanes_full ? 400u : 150u);
            PTMI_HIP(c, launch_quad_order(c->d_quad_cost, c->d_quad_order, c->d_quad_class, n_quads, stream_form ? c->d_tail_start : nullptr,
                                          tail_permille, c->stream));
            ++c->order_generation;
        }
        if (launches > 0) a.quad_order = c->d_quad_order;
        if (record) a.quad_cost = c->d_quad_cost;
        next_order_state = state_after;                        // (every launch counts, whether or not it records)
    }
    if (per_pixel_kernel) {
        // one word per tile workgroup for the sample chunks of the tiled per-pixel kernels (ptmi_device.h: enter_sample_chunk)
        const unsigned int need = ((unsigned int)(((width + 7) / 8) * ((rows_local + 7) / 8)) + 31u) & ~31u;
        if (need > c->chunk_capacity) {
            if (c->d_chunk_done) { PTMI_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_chunk_done); c->d_chunk_done = nullptr; c->chunk_capacity = 0; }
            PTMI_HIP(c, hipMalloc(&c->d_chunk_done, ((size_t)need + 32) * sizeof(unsigned int)));     // + the ticket counter's line
            c->chunk_capacity = need;
        }
        a.chunk_done = c->d_chunk_done; a.chunk_capacity = c->chunk_capacity;
        a.spp_chunks = sx ? 1 : c->opt_spp_chunks;
    }
    if (c->timing) { PTMI_HIP(c, hipEventRecord(c->ev0, c->stream)); }
    if (algorithm == PTMI_INLINE && c->opt_arithmetic == PTMI_ARITH_CONTRACTED) {
        PTMI_HIP(c, (hipError_t)ptmi_contracted_launch_inline(&a, c->variant == 9 ? 0 : c->variant, c->stream));
    } else if (algorithm == PTMI_INLINE) {
        PTMI_HIP(c, launch_render_inline(a, c->variant, c->stream));
    } else if (stream_form) {                              // rays travel through streams in HBM (PTMI_OPT_STREAMS_FORM; variant 9)
        if (int rc = render_streams_wavefront(c, a, n_spp, *camera)) return rc;
    } else if (c->has_glass) {                             // rays may split: the per-pixel tree walk
        // the first waiting children of every lane as 64-byte records in global memory (ptmi_streams_tree.hip): 16 KB per tile
        const size_t want = (size_t)tree_workgroups(width, rows_local) * kTreeFastLevels * 64 * 64;
        if (want > c->tree_stack_bytes) {
            PTMI_HIP(c, hipStreamSynchronize(c->stream));
            if (c->tree_stack) { (void)hipFree(c->tree_stack); c->tree_stack = nullptr; c->tree_stack_bytes = 0; }
            if (hipMalloc(&c->tree_stack, want) != hipSuccess) {
                (void)hipGetLastError(); c->tree_stack = nullptr;
                return fail(c, PTMI_ENOMEM, "no device memory for the tree walk's records of waiting children (16 KB per 8x8 tile)");
            }
            c->tree_stack_bytes = want;
        }
        a.tree_stack = static_cast<float4 *>(c->tree_stack);
        PTMI_HIP(c, launch_render_streams_tree(a, c->variant, c->stream));
    } else {
        PTMI_HIP(c, launch_render_streams(a, c->variant, c->stream));
    }
    if (c->timing) { PTMI_HIP(c, hipEventRecord(c->ev1, c->stream)); c->ev_valid = true; }
    c->order_state = next_order_state;
    const uint64_t px = (uint64_t)rows_local * (uint64_t)width;
    c->samples += px * (uint64_t)(n_spp > 0 ? n_spp : 0);
    if (algorithm == PTMI_INLINE)                          // Streams ignores the limit (Trace.hs:166-170): no nominal count
        c->nominal += px * (uint64_t)(n_spp > 0 ? n_spp : 0) * (uint64_t)(bounce_limit > 0 ? bounce_limit : 0);
    return PTMI_OK;
}

int check_render_args(ptmi_ctx *c, const ptmi_camera *camera, int algorithm, int bounce_limit, int n_spp)
{
    if (!camera) return fail(c, PTMI_EINVAL, "camera is NULL");
    if (algorithm != PTMI_INLINE && algorithm != PTMI_STREAMS) return fail(c, PTMI_EINVAL, "unknown algorithm");
    if (bounce_limit < 0) return fail(c, PTMI_EINVAL, "bounce_limit < 0");
    if (n_spp < 0) return fail(c, PTMI_EINVAL, "n_spp < 0");
    if (!c->d_scene) return fail(c, PTMI_ESTATE, "ptmi_set_scene has not been called");
    // "features that require diverging rays like light refraction" need the stream algorithm (Trace.hs:56-67)
    if (algorithm == PTMI_INLINE && c->has_glass)
        return fail(c, PTMI_EINVAL, "the scene holds a GLASS material: render Inline cannot split rays, use PTMI_STREAMS");
    if (algorithm == PTMI_STREAMS && c->has_glass && c->opt_seed_rule == PTMI_SEED_FROM_RESULT)
        return fail(c, PTMI_EINVAL, "PTMI_SEED_FROM_RESULT is undefined when rays split (GLASS): several results race for one pixel's seed");
    return PTMI_OK;
}


// ---- the chained closure: states under tokens (include/ptmi.h, "the closure, chained") ----------------------------------------
// Token = (the context's serial number << 40) | a counter that starts at 1: never 0, never reused, another context's never matches.
uint64_t chain_new_token(ptmi_ctx *c) { return (c->chain_serial << 40) | (++c->chain_counter & ((1ull << 40) - 1)); }

int chain_find(ptmi_ctx *c, uint64_t token)
{
    if (token == 0) return -1;
    for (size_t i = c->chain.size(); i-- > 0;)              // (the newest states are the ones asked for)
        if (c->chain[i].token == token) return (int)i;
    return -1;
}

int chain_slots(const ptmi_ctx *c, size_t bytes)
{
    if (c->opt_chain_slots > 0) return c->opt_chain_slots;
    const size_t fit = (c->device_memory / 16) / (bytes ? bytes : 1);
    return fit < 3 ? 3 : (fit > 64 ? 64 : (int)fit);
}

// A device block of `bytes` for a new state.  In this order: a block a released state left behind; a fresh allocation while fewer than
// PTMI_OPT_CHAIN_SLOTS blocks exist; the block of the OLDEST state still on the device (but `keep`), whose planes first move to host
// memory the library owns -- a state is never lost, only served from further away.
int chain_take_block(ptmi_ctx *c, size_t bytes, uint64_t keep, void **out)
{
    for (size_t i = 0; i < c->chain_free.size(); ++i)
        if (c->chain_free[i].first == bytes) { *out = c->chain_free[i].second; c->chain_free.erase(c->chain_free.begin() + (long)i); return PTMI_OK; }
    if (!c->chain_free.empty()) {                            // blocks of another image size: of no use any more
        PTMI_HIP(c, hipStreamSynchronize(c->stream));
        for (auto &fb : c->chain_free) (void)hipFree(fb.second);
        c->chain_free.clear();
    }
    size_t on_device = 0;
    for (const ptmi_ctx::ChainState &st : c->chain) on_device += st.block ? 1 : 0;
    if ((int)on_device < chain_slots(c, bytes)) {
        if (hipMalloc(out, bytes) == hipSuccess) return PTMI_OK;
        (void)hipGetLastError();                             // the device is full before the budget is: make room instead
    }
    for (ptmi_ctx::ChainState &st : c->chain) {
        if (!st.block || st.token == keep) continue;
        const size_t st_bytes = planes_bytes((size_t)st.width * st.height);
        st.host.reset(new (std::nothrow) char[st_bytes]);
        if (!st.host) return fail(c, PTMI_ENOMEM, "no host memory for a state of the chained closure that has to leave the device");
        const CopySpan whole{st.block, st.host.get(), st_bytes};
        PTMI_HIP(c, copy_to_host(c, &whole, 1));             // (drains the stream: nothing still reads the block)
        ++c->chain_stats.evictions;
        void *block = st.block;
        st.block = nullptr;
        if (st_bytes == bytes) { *out = block; return PTMI_OK; }
        (void)hipFree(block);
        PTMI_HIP(c, hipMalloc(out, bytes));
        return PTMI_OK;
    }
    PTMI_HIP(c, hipMalloc(out, bytes));                       // nothing to move out (the budget is below what one call needs): allocate all the same
    return PTMI_OK;
}

void chain_drop(ptmi_ctx *c, int idx)
{
    ptmi_ctx::ChainState &st = c->chain[(size_t)idx];
    if (st.block) c->chain_free.emplace_back(planes_bytes((size_t)st.width * st.height), st.block);     // stream-ordered reuse: no wait
    c->chain.erase(c->chain.begin() + idx);
}

// The state a chained call WRITES: token_in's planes in a block of its own (device-to-device copy, or in place when the caller gives the
// input up), or -- token_in not held -- the caller's host planes uploaded (`planes_in`: seven pointers, or three with colour_only).  On
// success *out_idx is a fresh entry at the end of c->chain, on the device, under a new token.
int chain_begin(ptmi_ctx *c, int width, int height, uint64_t token_in, int flags, const void *const *planes_in, bool colour_only, int *out_idx)
{
    if (width <= 0 || height <= 0) return fail(c, PTMI_EINVAL, "width and height must be positive");
    if (too_many_pixels(width, height)) return fail(c, PTMI_ELIMIT, "image too large");
    if (flags & ~PTMI_CHAIN_CONSUME) return fail(c, PTMI_EINVAL, "unknown flag");
    const size_t n = (size_t)width * height, bytes = planes_bytes(n);
    int in = chain_find(c, token_in);
    if (in >= 0 && (c->chain[(size_t)in].width != width || c->chain[(size_t)in].height != height))
        return fail(c, PTMI_EINVAL, "the token names a state of another image size");
    ptmi_ctx::ChainState fresh;
    fresh.width = width; fresh.height = height;
    if (in < 0) {                                            // the copy path: the RenderResult comes from the host
        const int n_planes = colour_only ? 3 : 7;
        for (int i = 0; i < n_planes; ++i)
            if (!planes_in || !planes_in[i])
                return fail(c, PTMI_ESTALE, token_in ? "the token names no state this context holds, and no host planes were given to take its place"
                                                     : "neither a token nor host planes were given");
        if (int rc = chain_take_block(c, bytes, 0, &fresh.block)) return rc;
        const Planes p = carve(fresh.block, n);
        void *dev[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
        CopySpan spans[7];
        for (int i = 0; i < n_planes; ++i) spans[i] = CopySpan{dev[i], const_cast<void *>(planes_in[i]), n * 4};
        hipError_t e = copy_to_device(c, spans, n_planes);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // the caller's planes are borrowed for the call only
        if (e != hipSuccess) { c->chain_free.emplace_back(bytes, fresh.block); PTMI_HIP(c, e); }
        ++c->chain_stats.renders_uploaded;
    } else if (flags & PTMI_CHAIN_CONSUME) {                 // the caller gives the input up: its block becomes the output
        ptmi_ctx::ChainState &src = c->chain[(size_t)in];
        if (!src.block) {                                    // (it had moved to the host: back first)
            if (int rc = chain_take_block(c, bytes, src.token, &src.block)) return rc;
            const CopySpan whole{src.block, src.host.get(), bytes};
            hipError_t e = copy_to_device(c, &whole, 1);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { c->chain_free.emplace_back(bytes, src.block); src.block = nullptr; PTMI_HIP(c, e); }
            src.host.reset();
        }
        fresh.block = src.block;
        src.block = nullptr;
        c->chain.erase(c->chain.begin() + in);
        ++c->chain_stats.renders_chained; ++c->chain_stats.renders_in_place;
    } else {                                                 // the input stands: the sample is rendered into a copy
        if (int rc = chain_take_block(c, bytes, token_in, &fresh.block)) return rc;
        in = chain_find(c, token_in);                        // (the vector did not move, but be sure)
        const ptmi_ctx::ChainState &src = c->chain[(size_t)in];
        hipError_t e;
        if (src.block) e = hipMemcpyAsync(fresh.block, src.block, bytes, hipMemcpyDeviceToDevice, c->stream);
        else {
            const CopySpan whole{fresh.block, src.host.get(), bytes};
            e = copy_to_device(c, &whole, 1);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (e != hipSuccess) { c->chain_free.emplace_back(bytes, fresh.block); PTMI_HIP(c, e); }
        ++c->chain_stats.renders_chained;
    }
    fresh.token = chain_new_token(c);
    c->chain.push_back(std::move(fresh));
    *out_idx = (int)c->chain.size() - 1;
    return PTMI_OK;
}

// (a call that fails after chain_begin must not leave a half-made state under a token nobody was told)
void chain_abandon(ptmi_ctx *c, int idx) { chain_drop(c, idx); }

int chain_download(ptmi_ctx *c, const ptmi_ctx::ChainState &st, void *const dst[7])
{
    const size_t n = (size_t)st.width * st.height;
    bool any = false;
    for (int i = 0; i < 7; ++i) any |= dst[i] != nullptr;
    if (!any) return PTMI_OK;
    ++c->chain_stats.fetches;
    if (st.block) {
        const Planes p = carve(st.block, n);
        const void *src[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
        CopySpan spans[7];
        int k = 0;
        for (int i = 0; i < 7; ++i)
            if (dst[i]) spans[k++] = CopySpan{const_cast<void *>(src[i]), dst[i], n * 4};
        PTMI_HIP(c, copy_to_host(c, spans, k));              // stream-ordered behind the renders; returns with the planes filled
        return PTMI_OK;
    }
    const Planes p = carve(st.host.get(), n);
    const void *src[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
    for (int i = 0; i < 7; ++i)
        if (dst[i]) std::memcpy(dst[i], src[i], n * 4);
    return PTMI_OK;
}

}  // namespace

void ptmi::context_size(const ptmi_ctx *cc, int *width, int *height)
{
    ptmi_ctx *c = const_cast<ptmi_ctx *>(cc);
    std::lock_guard<std::mutex> lock(c->mu);
    *width = c->width; *height = c->height;
}

extern "C" {

int ptmi_version(void) { return PTMI_VERSION; }

const char *ptmi_strerror(int code)
{
    switch (code) {
    case PTMI_OK: return "ok";
    case PTMI_EINVAL: return "invalid argument";
    case PTMI_ENODEVICE: return "no usable HIP device";
    case PTMI_EHIP: return "HIP runtime error";
    case PTMI_ENOMEM: return "out of memory";
    case PTMI_ESTATE: return "call order violated";
    case PTMI_ELIMIT: return "scene exceeds PTMI_MAX_PRIMITIVES";
    case PTMI_ESTALE: return "the token names no state this context holds";
    default: return "unknown error";
    }
}

const char *ptmi_last_error(const ptmi_ctx *ctx)
{
    if (!ctx) return g_create_error.c_str();
    if (t_error_of == ctx) return t_error.c_str();             // the caller's own failure: nobody else writes this string
    thread_local std::string copy;                             // a failure another thread saw (a group's member threads): copied under the lock
    std::lock_guard<std::mutex> lock(const_cast<ptmi_ctx *>(ctx)->mu);
    copy = ctx->err;
    return copy.c_str();
}

int ptmi_create(ptmi_ctx **out, int device)
{
    if (!out) return fail(nullptr, PTMI_EINVAL, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, PTMI_ENODEVICE, std::string("hipGetDeviceCount: ") +
                    (e != hipSuccess ? hipGetErrorString(e) : "0 devices") + " (libptmi has no CPU path)");
    if (device < 0 || device >= count) return fail(nullptr, PTMI_ENODEVICE, "device index out of range");
    ptmi_ctx *c = new (std::nothrow) ptmi_ctx;
    if (!c) return fail(nullptr, PTMI_ENOMEM, "host allocation failed");
    c->device = device;
    static std::atomic<uint64_t> contexts_made{0};
    c->chain_serial = ++contexts_made;
    auto bail = [&](hipError_t err, const char *what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        ptmi_destroy(c);                                     // (also takes the error out of the runtime's sticky slot)
        return PTMI_EHIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return bail(e, "hipDeviceGetAttribute");
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) c->device_memory = total_b;
    }
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->stream = c->own_stream;
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_snap, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&c->d_live, kLiveBytes)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(&c->d_work, kWorkWords * sizeof(unsigned int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(&c->d_iters, kItersBytes)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(&c->d_stream_counters, kScWords * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemsetAsync(c->d_stream_counters, 0, kScWords * sizeof(unsigned long long), c->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    // stream-ordered fills: the context's stream is non-blocking, so a NULL-stream hipMemset would race with it
    if ((e = hipMemsetAsync(c->d_live, 0, kLiveBytes, c->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(c->d_work, 0, kWorkWords * sizeof(unsigned int), c->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(c->d_iters, 0, kItersBytes, c->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = c;
    return PTMI_OK;
}

void ptmi_destroy(ptmi_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->owned_block) (void)hipFree(c->owned_block);
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->d_live) (void)hipFree(c->d_live);
    if (c->d_work) (void)hipFree(c->d_work);
    if (c->d_pass_first) (void)hipFree(c->d_pass_first);
    if (c->d_iters) (void)hipFree(c->d_iters);
    if (c->d_stream_counters) (void)hipFree(c->d_stream_counters);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->queue_block) (void)hipFree(c->queue_block);
    if (c->colour_backup) (void)hipFree(c->colour_backup);
    if (c->hit_block) (void)hipFree(c->hit_block);
    if (c->d_hit_counts) (void)hipFree(c->d_hit_counts);
    if (c->d_hit_missed) (void)hipFree(c->d_hit_missed);
    if (c->d_snapshots) (void)hipFree(c->d_snapshots);
    if (c->spill_block) (void)hipFree(c->spill_block);
    if (c->tree_stack) (void)hipFree(c->tree_stack);
    if (c->d_qcount) (void)hipFree(c->d_qcount);
    if (c->d_quad_cost) (void)hipFree(c->d_quad_cost);
    if (c->d_quad_order) (void)hipFree(c->d_quad_order);
    if (c->d_quad_class) (void)hipFree(c->d_quad_class);
    if (c->d_tail_start) (void)hipFree(c->d_tail_start);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->tail_stream) { (void)hipStreamSynchronize(c->tail_stream); (void)hipStreamDestroy(c->tail_stream); }
    if (c->d_chunk_done) (void)hipFree(c->d_chunk_done);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_snap) (void)hipEventDestroy(c->ev_snap);
    if (c->d_region_done) (void)hipFree(c->d_region_done);
    for (ptmi_ctx::ChainState &st : c->chain) if (st.block) (void)hipFree(st.block);
    for (auto &fb : c->chain_free) (void)hipFree(fb.second);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    (void)hipGetLastError();                                 // whatever failed above was ignored on purpose: it is nobody else's to find
    delete c;
}

int ptmi_set_scene(ptmi_ctx *c, const ptmi_sphere *spheres, int n_spheres, const ptmi_plane *planes, int n_planes)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (n_spheres < 0 || n_planes < 0 || (n_spheres > 0 && !spheres) || (n_planes > 0 && !planes))
        return fail(c, PTMI_EINVAL, "bad scene arguments");
    // expMinWith _ [] = error "Invalid call to 'expMinWith'"   (src/Util.hs:172)
    if (n_spheres + n_planes == 0) return fail(c, PTMI_EINVAL, "empty scene (expMinWith on an empty list)");
    if (n_spheres + n_planes > PTMI_MAX_PRIMITIVES) return fail(c, PTMI_ELIMIT, "too many primitives");
    for (int i = 0; i < n_spheres; ++i)
        if (spheres[i].brdf_tag < PTMI_MATTE || spheres[i].brdf_tag > PTMI_GLASS)
            return fail(c, PTMI_EINVAL, "sphere with unknown brdf_tag");
    for (int j = 0; j < n_planes; ++j)
        if (planes[j].brdf_tag < PTMI_MATTE || planes[j].brdf_tag > PTMI_GLASS)
            return fail(c, PTMI_EINVAL, "plane with unknown brdf_tag");
    PTMI_HIP(c, hipSetDevice(c->device));
    std::vector<float4> packed;
    pack_scene(spheres, n_spheres, planes, n_planes, packed);
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    // the new scene stands complete before the old one goes: a failure here leaves the context with the scene (and the counts) it had
    void *fresh = nullptr;
    PTMI_HIP(c, hipMalloc(&fresh, packed.size() * sizeof(float4)));
    hipError_t e = hipMemcpyAsync(fresh, packed.data(), packed.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // `packed` dies at return
    if (e != hipSuccess) { (void)hipFree(fresh); PTMI_HIP(c, e); }
    if (c->d_scene) (void)hipFree(c->d_scene);
    c->d_scene = static_cast<float4 *>(fresh);
    c->n_spheres = n_spheres; c->n_planes = n_planes;
    ++c->scene_version;
    c->has_glass = false;
    for (int i = 0; i < n_spheres; ++i) c->has_glass |= spheres[i].brdf_tag == PTMI_GLASS;
    for (int j = 0; j < n_planes; ++j) c->has_glass |= planes[j].brdf_tag == PTMI_GLASS;
    return PTMI_OK;
}

int ptmi_set_partition(ptmi_ctx *c, int stripe_rows, int n_parts, int part)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (stripe_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts)
        return fail(c, PTMI_EINVAL, "bad partition");
    if (c->owned_block) return fail(c, PTMI_ESTATE, "ptmi_set_partition must precede ptmi_resize");
    c->stripe_rows = stripe_rows; c->n_parts = n_parts; c->part = part;
    return PTMI_OK;
}

int ptmi_resize(ptmi_ctx *c, int width, int height)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (width <= 0 || height <= 0) return fail(c, PTMI_EINVAL, "width and height must be positive");
    if (too_many_pixels(width, height)) return fail(c, PTMI_ELIMIT, "image too large");
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    if (c->owned_block) { (void)hipFree(c->owned_block); c->owned_block = nullptr; }
    if (c->colour_backup) { (void)hipFree(c->colour_backup); c->colour_backup = nullptr; c->colour_backup_bytes = 0; }   // (sized by the old image)
    // The old planes are gone (first: the new ones may need their room).  Until the new ones stand the context is UNSIZED -- a failure below
    // must not leave it pointing into the block just freed: every call that needs planes then answers PTMI_ESTATE.
    c->owned = Planes{};
    c->width = c->height = c->rows_local = 0;
    c->use_bound = false;
    const int rows_local = rows_of_part(height, effective_stripe(c), c->n_parts, c->part);
    const size_t n = (size_t)rows_local * (size_t)width;
    const size_t bytes = planes_bytes(n > 0 ? n : 1);
    void *block = nullptr;
    PTMI_HIP(c, hipMalloc(&block, bytes));
    hipError_t e = hipMemsetAsync(block, 0, bytes, c->stream);         // ordered before anything launched on the stream
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(block); PTMI_HIP(c, e); }
    c->owned_block = block;
    c->owned = carve(block, n > 0 ? n : 1);
    c->width = width; c->height = height; c->rows_local = rows_local;
    return PTMI_OK;
}

int ptmi_local_rows(const ptmi_ctx *cc)
{
    if (!cc) return PTMI_EINVAL;
    ptmi_ctx *c = const_cast<ptmi_ctx *>(cc);
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return PTMI_ESTATE;
    return c->rows_local;
}

int ptmi_global_row(const ptmi_ctx *cc, int local_row)
{
    if (!cc) return PTMI_EINVAL;
    ptmi_ctx *c = const_cast<ptmi_ctx *>(cc);
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return PTMI_ESTATE;
    if (local_row < 0 || local_row >= c->rows_local) return PTMI_EINVAL;
    const int s = effective_stripe(c);
    return ((local_row / s) * c->n_parts + c->part) * s + local_row % s;
}

int ptmi_bind_planes(ptmi_ctx *c, float *r, float *g, float *b, uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    const int n_null = !r + !g + !b + !sa + !sb + !sc + !sctr;
    if (n_null == 7) { c->use_bound = false; return PTMI_OK; }
    if (n_null != 0) return fail(c, PTMI_EINVAL, "bind either all seven planes or none");
    c->bound = Planes{r, g, b, sa, sb, sc, sctr};
    c->use_bound = true;
    return PTMI_OK;
}

int ptmi_get_planes(ptmi_ctx *c, float **r, float **g, float **b, uint32_t **sa, uint32_t **sb, uint32_t **sc, uint32_t **sctr)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    const Planes &p = active(c);
    if (r) *r = p.r; if (g) *g = p.g; if (b) *b = p.b;
    if (sa) *sa = p.sa; if (sb) *sb = p.sb; if (sc) *sc = p.sc; if (sctr) *sctr = p.sctr;
    return PTMI_OK;
}

int ptmi_set_stream(ptmi_ctx *c, void *hip_stream)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->ev_valid = false;
    return PTMI_OK;
}

int ptmi_set_timing(ptmi_ctx *c, int enabled)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    c->timing = enabled != 0;
    c->ev_valid = false;
    return PTMI_OK;
}

int ptmi_set_variant(ptmi_ctx *c, int variant)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (variant < 0 || variant > 18) return fail(c, PTMI_EINVAL, "unknown variant");
    if (!variant_available(variant)) return fail(c, PTMI_EINVAL, "this variant is an ablation kernel: build libptmi with -DPTMI_ABLATIONS");
    c->variant = variant;
    return PTMI_OK;
}

int ptmi_order_schedule(int launches, int stream_form, int *rebuild, int *record)
{
    return order_schedule(launches, stream_form, rebuild, record);
}

int ptmi_stream_schedule(int n_spp, uint64_t n_pixels, uint64_t lanes, int batch, int graded, int32_t *first, int capacity)
{
    if (n_spp < 0 || !first || capacity < 2) return PTMI_EINVAL;
    int table[kMaxStreamPasses + 1];
    const int passes = stream_schedule(n_spp, n_pixels, lanes, batch, graded != 0, table);
    if (passes + 1 > capacity) return PTMI_ELIMIT;
    for (int k = 0; k <= passes; ++k) first[k] = table[k];
    return passes;
}

int ptmi_stream_tickets(int option, const int32_t *first, int passes, int queue_regions, int32_t *pass_out, int32_t *region_out, int capacity)
{
    if (passes < 1 || passes > kMaxStreamPasses || queue_regions < 1 || !first || !pass_out || !region_out) return PTMI_EINVAL;
    if (option < 0 || option > 164 || (option > 64 && option < 102)) return PTMI_EINVAL;
    for (int p = 0; p < passes; ++p)
        if (first[p + 1] <= first[p]) return PTMI_EINVAL;
    int table[kMaxStreamPasses + 1], sizes[kMaxStreamPasses + 1];
    for (int p = 0; p <= passes; ++p) sizes[p] = first[p];
    const int groups = pass_group_table(option, sizes, passes, table);
    const long long tickets = (long long)passes * queue_regions;
    if (tickets > capacity) return PTMI_ELIMIT;
    for (long long j = 0; j < tickets; ++j) {
        unsigned int pass = 0, k = 0;
        decode_ticket((unsigned int)j, (unsigned int)queue_regions, groups < passes ? table : nullptr, groups, pass, k);
        pass_out[j] = (int32_t)pass; region_out[j] = (int32_t)k;
    }
    return (int)tickets;
}

int ptmi_set_option(ptmi_ctx *c, int option, int64_t value)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    switch (option) {
    case PTMI_OPT_STREAMS_SEED_RULE:
        if (value != PTMI_SEED_KEEP_ACCUMULATOR && value != PTMI_SEED_FROM_RESULT && value != PTMI_SEED_AUTO) return fail(c, PTMI_EINVAL, "unknown seed rule");
        c->opt_seed_rule = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_STEP_CAP:
        if (value < 1 || value > (1 << 30)) return fail(c, PTMI_EINVAL, "step cap must be in [1, 2^30]");
        c->opt_step_cap = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_CAPACITY:
        if (value < 1 || value > 64) return fail(c, PTMI_EINVAL, "stream capacity must be in [1, 64] rays per pixel-sample");
        c->opt_capacity = (int)value; c->grown_capacity = 0;
        if (c->queue_block) {                                // "starts over from the value given": the streams are carved anew by the next call
            PTMI_HIP(c, hipSetDevice(c->device));
            PTMI_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->queue_block); c->queue_block = nullptr; c->queue_capacity = 0;
        }
        return PTMI_OK;
    case PTMI_OPT_STREAMS_FORM:
        if (value != PTMI_FORM_AUTO && value != PTMI_FORM_STREAM && value != PTMI_FORM_PIXEL) return fail(c, PTMI_EINVAL, "unknown Streams form");
        c->opt_form = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_BATCH:
        if (value < 0 || value > 64) return fail(c, PTMI_EINVAL, "stream batch must be in [0, 64] samples");
        c->opt_batch = (int)value; return PTMI_OK;
    case PTMI_OPT_SPP_CHUNKS:
        if (value < 0 || value > 64) return fail(c, PTMI_EINVAL, "sample chunks must be in [0, 64]");
        c->opt_spp_chunks = (int)value; return PTMI_OK;
    case PTMI_OPT_ARITHMETIC:
        if (value != PTMI_ARITH_EXACT && value != PTMI_ARITH_CONTRACTED) return fail(c, PTMI_EINVAL, "unknown arithmetic mode");
        c->opt_arithmetic = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_TAIL:
        if (value < -1 || value > 1000) return fail(c, PTMI_EINVAL, "stream tail must be -1 (automatic) or thousandths in [0, 1000]");
        c->opt_tail_permille = (int)value; return PTMI_OK;
    case PTMI_OPT_ORDERED_PASSES:
        if (value < 0 || value > 64) return fail(c, PTMI_EINVAL, "ordered passes must be 0 (automatic), 1 (off) or k in [2, 64]");
        c->opt_ordered_passes = (int)value; return PTMI_OK;
    case PTMI_OPT_GLASS_BATCH:
        if (value < 0 || value > 64) return fail(c, PTMI_EINVAL, "glass batch must be 0 (automatic), 1 (off) or k in [2, 64] lanes");
        c->opt_glass_batch = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_GRADED:
        if (value != 0 && value != 1) return fail(c, PTMI_EINVAL, "graded passes are on (1) or off (0)");
        c->opt_graded = (int)value; return PTMI_OK;
    case PTMI_OPT_SNAPSHOT_BUDGET_MB:
        if (value < 0 || value > (1 << 20)) return fail(c, PTMI_EINVAL, "snapshot budget must be 0 (automatic) or megabytes in [1, 2^20]");
        c->opt_snapshot_mb = (int)value; return PTMI_OK;
    case PTMI_OPT_STREAM_PASS_GROUPS:
        if (value < 0 || value > 164 || (value > 64 && value < 102)) return fail(c, PTMI_EINVAL, "pass groups must be 0 (automatic), 1 (every pass on its own), k in [2, 64] or 100 + g, g in [2, 64]");
        c->opt_pass_groups = (int)value; return PTMI_OK;
    case PTMI_OPT_CHAIN_SLOTS:
        if (value != 0 && (value < 2 || value > 4096)) return fail(c, PTMI_EINVAL, "chain slots must be 0 (automatic) or in [2, 4096]");
        c->opt_chain_slots = (int)value; return PTMI_OK;
    case PTMI_OPT_PASS_HANDOFF:
        if (value != PTMI_HANDOFF_FENCED && value != PTMI_HANDOFF_FENCE_FREE) return fail(c, PTMI_EINVAL, "unknown pass hand-off");
        c->opt_pass_handoff = (int)value; return PTMI_OK;
    default: return fail(c, PTMI_EINVAL, "unknown option");
    }
}

int ptmi_get_option(ptmi_ctx *c, int option, int64_t *value)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!value) return fail(c, PTMI_EINVAL, "value is NULL");
    switch (option) {
    case PTMI_OPT_STREAMS_SEED_RULE: *value = c->opt_seed_rule; return PTMI_OK;
    case PTMI_OPT_STREAM_STEP_CAP:   *value = c->opt_step_cap; return PTMI_OK;
    case PTMI_OPT_STREAM_CAPACITY:   *value = c->grown_capacity > c->opt_capacity ? c->grown_capacity : c->opt_capacity; return PTMI_OK;
    case PTMI_OPT_STREAMS_FORM:      *value = c->opt_form; return PTMI_OK;
    case PTMI_OPT_STREAM_BATCH:      *value = c->opt_batch; return PTMI_OK;
    case PTMI_OPT_SPP_CHUNKS: *value = c->opt_spp_chunks; return PTMI_OK;
    case PTMI_OPT_ARITHMETIC: *value = c->opt_arithmetic; return PTMI_OK;
    case PTMI_OPT_STREAM_TAIL: *value = c->opt_tail_permille; return PTMI_OK;
    case PTMI_OPT_ORDERED_PASSES: *value = c->opt_ordered_passes; return PTMI_OK;
    case PTMI_OPT_GLASS_BATCH: *value = c->opt_glass_batch; return PTMI_OK;
    case PTMI_OPT_STREAM_GRADED: *value = c->opt_graded; return PTMI_OK;
    case PTMI_OPT_STREAM_PASS_GROUPS: *value = c->opt_pass_groups; return PTMI_OK;
    case PTMI_OPT_SNAPSHOT_BUDGET_MB: *value = c->opt_snapshot_mb; return PTMI_OK;
    case PTMI_OPT_CHAIN_SLOTS: *value = c->opt_chain_slots; return PTMI_OK;
    case PTMI_OPT_PASS_HANDOFF: *value = c->opt_pass_handoff; return PTMI_OK;
    default: return fail(c, PTMI_EINVAL, "unknown option");
    }
}

static int seed_common(ptmi_ctx *c, uint64_t seed0, bool clear)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, launch_seed(active(c), c->width, c->rows_local, effective_stripe(c), c->n_parts, c->part,
                            seed0, clear, c->stream));
    return PTMI_OK;
}

int ptmi_init_output(ptmi_ctx *c, uint64_t seed0) { return seed_common(c, seed0, true); }
int ptmi_reseed(ptmi_ctx *c, uint64_t seed0) { return seed_common(c, seed0, false); }

int ptmi_create_with(ptmi_ctx *c, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    if (!w0 || !w1 || !w2) return fail(c, PTMI_EINVAL, "word planes are NULL");
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->rows_local * c->width;
    if (int rc = ensure_scratch(c, 3 * n * 4)) return rc;
    uint32_t *d = static_cast<uint32_t *>(c->scratch);
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    const CopySpan words[3] = {{d, const_cast<uint32_t *>(w0), n * 4}, {d + n, const_cast<uint32_t *>(w1), n * 4},
                               {d + 2 * n, const_cast<uint32_t *>(w2), n * 4}};
    PTMI_HIP(c, copy_to_device(c, words, 3));
    PTMI_HIP(c, launch_create_with(active(c), d, d + n, d + 2 * n, (int64_t)n, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return PTMI_OK;
}

int ptmi_upload_state(ptmi_ctx *c, const float *r, const float *g, const float *b,
                      const uint32_t *sa, const uint32_t *sb, const uint32_t *sc, const uint32_t *sctr)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    const size_t bytes = (size_t)c->rows_local * c->width * 4;
    Planes &p = active(c);
    const void *src[7] = {r, g, b, sa, sb, sc, sctr};
    void *dst[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
    CopySpan spans[7];
    int n_spans = 0;
    for (int i = 0; i < 7; ++i)
        if (src[i] && bytes) spans[n_spans++] = CopySpan{dst[i], const_cast<void *>(src[i]), bytes};
    PTMI_HIP(c, copy_to_device(c, spans, n_spans));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return PTMI_OK;
}

int ptmi_download_state(ptmi_ctx *c, float *r, float *g, float *b,
                        uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    const size_t bytes = (size_t)c->rows_local * c->width * 4;
    Planes &p = active(c);
    void *dst[7] = {r, g, b, sa, sb, sc, sctr};
    const void *src[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
    CopySpan spans[7];
    int n_spans = 0;
    for (int i = 0; i < 7; ++i)
        if (dst[i] && bytes) spans[n_spans++] = CopySpan{const_cast<void *>(src[i]), dst[i], bytes};
    PTMI_HIP(c, copy_to_host(c, spans, n_spans));
    return PTMI_OK;
}

int ptmi_download_color(ptmi_ctx *c, float *r, float *g, float *b)
{
    return ptmi_download_state(c, r, g, b, nullptr, nullptr, nullptr, nullptr);
}

int ptmi_render(ptmi_ctx *c, const ptmi_camera *camera, int algorithm, int bounce_limit, int n_spp)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (int rc = check_render_args(c, camera, algorithm, bounce_limit, n_spp)) return rc;
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    PTMI_HIP(c, hipSetDevice(c->device));
    return launch_render(c, active(c), camera, algorithm, bounce_limit, n_spp, c->width, c->height,
                         c->rows_local, effective_stripe(c), c->n_parts, c->part, nullptr, nullptr);
}

/* 1 when ptmi_render with this algorithm returns only after device work it waits for (the stream form with a ray-splitting
 * scene or unordered items reads its overflow counters back), 0 when it only enqueues.  ptmi_group_render gives such members a
 * host thread each. */
int ptmi_render_blocks(ptmi_ctx *c, int algorithm)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    const bool stream_form = uses_stream_form(c, algorithm, c->n_parts, -1);        // (for some sample count: the automatic choice looks at it)
    const bool ordered = !c->has_glass && (c->opt_batch == 0 || effective_seed_rule(c) == PTMI_SEED_FROM_RESULT);
    return stream_form && !ordered ? 1 : 0;
}

int ptmi_synchronize(ptmi_ctx *c)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return PTMI_OK;
}

int ptmi_render1(ptmi_ctx *c, const ptmi_camera *camera, int algorithm, int bounce_limit,
                 int width, int height, const int64_t *screen_x, const int64_t *screen_y,
                 const float *r_in, const float *g_in, const float *b_in,
                 const uint32_t *sa_in, const uint32_t *sb_in, const uint32_t *sc_in, const uint32_t *sctr_in,
                 float *r_out, float *g_out, float *b_out,
                 uint32_t *sa_out, uint32_t *sb_out, uint32_t *sc_out, uint32_t *sctr_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (int rc = check_render_args(c, camera, algorithm, bounce_limit, 1)) return rc;
    if (width <= 0 || height <= 0) return fail(c, PTMI_EINVAL, "width and height must be positive");
    if (too_many_pixels(width, height)) return fail(c, PTMI_ELIMIT, "image too large");
    if (!r_in || !g_in || !b_in || !sa_in || !sb_in || !sc_in || !sctr_in ||
        !r_out || !g_out || !b_out || !sa_out || !sb_out || !sc_out || !sctr_out)
        return fail(c, PTMI_EINVAL, "a plane pointer is NULL");
    if ((screen_x == nullptr) != (screen_y == nullptr)) return fail(c, PTMI_EINVAL, "give both screen planes or neither");
    if (screen_x && algorithm == PTMI_STREAMS) return fail(c, PTMI_EINVAL, "Streams takes the implicit screen only");
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)width * height;
    const size_t pb = planes_bytes(n);
    const size_t sb = screen_x ? 2 * n * sizeof(int64_t) : 0;
    if (int rc = ensure_scratch(c, pb + sb)) return rc;
    Planes p = carve(c->scratch, n);
    int64_t *dsx = screen_x ? reinterpret_cast<int64_t *>(static_cast<char *>(c->scratch) + pb) : nullptr;
    int64_t *dsy = screen_x ? dsx + n : nullptr;
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    const void *src[7] = {r_in, g_in, b_in, sa_in, sb_in, sc_in, sctr_in};
    void *dev[7] = {p.r, p.g, p.b, p.sa, p.sb, p.sc, p.sctr};
    void *dst[7] = {r_out, g_out, b_out, sa_out, sb_out, sc_out, sctr_out};
    CopySpan in[9];
    for (int i = 0; i < 7; ++i) in[i] = CopySpan{dev[i], const_cast<void *>(src[i]), n * 4};
    if (screen_x) {
        in[7] = CopySpan{dsx, const_cast<int64_t *>(screen_x), n * sizeof(int64_t)};
        in[8] = CopySpan{dsy, const_cast<int64_t *>(screen_y), n * sizeof(int64_t)};
    }
    PTMI_HIP(c, copy_to_device(c, in, screen_x ? 9 : 7));
    if (int rc = launch_render(c, p, camera, algorithm, bounce_limit, 1, width, height, height, height, 1, 0, dsx, dsy)) return rc;
    CopySpan out[7];
    for (int i = 0; i < 7; ++i) out[i] = CopySpan{dev[i], dst[i], n * 4};
    PTMI_HIP(c, copy_to_host(c, out, 7));           // stream-ordered behind the kernel; returns with the planes filled
    return PTMI_OK;
}

int ptmi_render1_chained(ptmi_ctx *c, const ptmi_camera *camera, int algorithm, int bounce_limit, int width, int height,
                         uint64_t token_in, int flags,
                         const float *r_in, const float *g_in, const float *b_in,
                         const uint32_t *sa_in, const uint32_t *sb_in, const uint32_t *sc_in, const uint32_t *sctr_in,
                         uint64_t *token_out,
                         float *r_out, float *g_out, float *b_out,
                         uint32_t *sa_out, uint32_t *sb_out, uint32_t *sc_out, uint32_t *sctr_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!token_out) return fail(c, PTMI_EINVAL, "token_out is NULL");
    *token_out = 0;
    if (int rc = check_render_args(c, camera, algorithm, bounce_limit, 1)) return rc;
    PTMI_HIP(c, hipSetDevice(c->device));
    const void *const planes_in[7] = {r_in, g_in, b_in, sa_in, sb_in, sc_in, sctr_in};
    int idx = -1;
    if (int rc = chain_begin(c, width, height, token_in, flags, planes_in, false, &idx)) return rc;
    const Planes p = carve(c->chain[(size_t)idx].block, (size_t)width * height);
    if (int rc = launch_render(c, p, camera, algorithm, bounce_limit, 1, width, height, height, height, 1, 0, nullptr, nullptr)) { chain_abandon(c, idx); return rc; }
    void *const dst[7] = {r_out, g_out, b_out, sa_out, sb_out, sc_out, sctr_out};
    if (int rc = chain_download(c, c->chain[(size_t)idx], dst)) { chain_abandon(c, idx); return rc; }
    *token_out = c->chain[(size_t)idx].token;
    return PTMI_OK;
}

int ptmi_chain_init_output(ptmi_ctx *c, int width, int height, uint64_t seed0, uint64_t *token_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!token_out) return fail(c, PTMI_EINVAL, "token_out is NULL");
    *token_out = 0;
    if (width <= 0 || height <= 0) return fail(c, PTMI_EINVAL, "width and height must be positive");
    if (too_many_pixels(width, height)) return fail(c, PTMI_ELIMIT, "image too large");
    PTMI_HIP(c, hipSetDevice(c->device));
    ptmi_ctx::ChainState fresh;
    fresh.width = width; fresh.height = height;
    const size_t n = (size_t)width * height;
    if (int rc = chain_take_block(c, planes_bytes(n), 0, &fresh.block)) return rc;
    const hipError_t e = launch_seed(carve(fresh.block, n), width, height, height, 1, 0, seed0, true, c->stream);
    if (e != hipSuccess) { c->chain_free.emplace_back(planes_bytes(n), fresh.block); PTMI_HIP(c, e); }
    fresh.token = chain_new_token(c);
    *token_out = fresh.token;
    c->chain.push_back(std::move(fresh));
    return PTMI_OK;
}

int ptmi_chain_reseed(ptmi_ctx *c, uint64_t seed0, int width, int height, uint64_t token_in, int flags,
                      const float *r_in, const float *g_in, const float *b_in, uint64_t *token_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!token_out) return fail(c, PTMI_EINVAL, "token_out is NULL");
    *token_out = 0;
    PTMI_HIP(c, hipSetDevice(c->device));
    const void *const planes_in[7] = {r_in, g_in, b_in, nullptr, nullptr, nullptr, nullptr};
    int idx = -1;
    if (int rc = chain_begin(c, width, height, token_in, flags, planes_in, true, &idx)) return rc;
    const hipError_t e = launch_seed(carve(c->chain[(size_t)idx].block, (size_t)width * height), width, height, height, 1, 0, seed0, false, c->stream);
    if (e != hipSuccess) { chain_abandon(c, idx); PTMI_HIP(c, e); }
    *token_out = c->chain[(size_t)idx].token;
    return PTMI_OK;
}

int ptmi_chain_fetch(ptmi_ctx *c, uint64_t token, float *r, float *g, float *b, uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    const int idx = chain_find(c, token);
    if (idx < 0) return fail(c, PTMI_ESTALE, "the token names no state this context holds");
    PTMI_HIP(c, hipSetDevice(c->device));
    void *const dst[7] = {r, g, b, sa, sb, sc, sctr};
    return chain_download(c, c->chain[(size_t)idx], dst);
}

int ptmi_chain_release(ptmi_ctx *c, uint64_t token)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    const int idx = chain_find(c, token);
    if (idx >= 0) chain_drop(c, idx);
    return PTMI_OK;
}

int ptmi_chain_info(ptmi_ctx *c, ptmi_chain_stats *out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!out) return fail(c, PTMI_EINVAL, "out is NULL");
    *out = c->chain_stats;
    out->states_on_device = out->states_on_host = 0;
    for (const ptmi_ctx::ChainState &st : c->chain) { if (st.block) ++out->states_on_device; else ++out->states_on_host; }
    out->width = c->chain.empty() ? 0 : (uint32_t)c->chain.back().width;
    out->height = c->chain.empty() ? 0 : (uint32_t)c->chain.back().height;
    out->device_slots = (uint32_t)chain_slots(c, planes_bytes((size_t)(out->width ? out->width : 1) * (out->height ? out->height : 1)));
    return PTMI_OK;
}

int ptmi_present(ptmi_ctx *c, int iterations, float *rgb32f_out, uint8_t *rgba8_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    if (iterations <= 0) return fail(c, PTMI_EINVAL, "iterations must be positive");
    if (!rgb32f_out && !rgba8_out) return PTMI_OK;
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->rows_local * c->width;
    if (n == 0) return PTMI_OK;
    const size_t rgb_bytes = ((n * 12 + 255) / 256) * 256;
    if (int rc = ensure_scratch(c, rgb_bytes + n * 4)) return rc;
    float *d_rgb = static_cast<float *>(c->scratch);
    uint32_t *d_rgba = reinterpret_cast<uint32_t *>(static_cast<char *>(c->scratch) + rgb_bytes);
    PTMI_HIP(c, launch_present(active(c), (long long)n, iterations, rgb32f_out ? d_rgb : nullptr,
                               rgba8_out ? d_rgba : nullptr, c->stream));
    CopySpan out[2];
    int n_out = 0;
    if (rgb32f_out) out[n_out++] = CopySpan{d_rgb, rgb32f_out, n * 12};
    if (rgba8_out) out[n_out++] = CopySpan{d_rgba, rgba8_out, n * 4};
    PTMI_HIP(c, copy_to_host(c, out, n_out));
    return PTMI_OK;
}

int ptmi_snapshot_color(ptmi_ctx *c, float *dst_device, void *hip_stream)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->width <= 0) return fail(c, PTMI_ESTATE, "ptmi_resize has not been called");
    if (!dst_device) return fail(c, PTMI_EINVAL, "dst_device is NULL");
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->rows_local * c->width * 4;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    if (s != c->stream) {                                      // order the copy behind the renders already issued
        PTMI_HIP(c, hipEventRecord(c->ev_snap, c->stream));
        PTMI_HIP(c, hipStreamWaitEvent(s, c->ev_snap, 0));
    }
    Planes &p = active(c);
    const float *src[3] = {p.r, p.g, p.b};
    for (int k = 0; k < 3 && bytes; ++k)
        PTMI_HIP(c, hipMemcpyAsync(reinterpret_cast<char *>(dst_device) + (size_t)k * bytes, src[k], bytes, hipMemcpyDeviceToDevice, s));
    return PTMI_OK;
}

int ptmi_get_stats(ptmi_ctx *c, ptmi_stats *out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!out) return fail(c, PTMI_EINVAL, "out is NULL");
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long live = 0, sc[kScWords] = {0, 0, 0, 0}; unsigned int iters = 0;
    std::vector<unsigned long long> live_shards((size_t)kStatShards * kStatStride);     // the sharded statistics: sum / maximum over the shards
    std::vector<unsigned int> iter_shards((size_t)kStatShards * 2 * kStatStride);
    PTMI_HIP(c, hipMemcpyAsync(live_shards.data(), c->d_live, kLiveBytes, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipMemcpyAsync(sc, c->d_stream_counters, sizeof sc, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipMemcpyAsync(iter_shards.data(), c->d_iters, kItersBytes, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < kStatShards; ++k) {
        live += live_shards[(size_t)k * kStatStride];
        const unsigned int it = iter_shards[(size_t)k * 2 * kStatStride];
        iters = it > iters ? it : iters;
    }
    out->live_bounces = live + c->live_host; out->nominal_bounces = c->nominal; out->samples = c->samples;
    out->stream_iterations = iters;
    out->stream_rays_dropped = c->rays_dropped + sc[kScDropped];
    out->stream_rays_truncated = c->rays_truncated + sc[kScTruncated];
    out->stream_rays_spilled = c->rays_spilled;
    out->stream_rays_overflowed = c->rays_overflowed;
    out->last_render_ms = 0.0f;
    if (c->timing && c->ev_valid) PTMI_HIP(c, hipEventElapsedTime(&out->last_render_ms, c->ev0, c->ev1));
    return PTMI_OK;
}

int ptmi_debug_counters_n(ptmi_ctx *c, uint32_t *out, int capacity)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!out) return fail(c, PTMI_EINVAL, "out is NULL");
    if (capacity < 0) return fail(c, PTMI_EINVAL, "capacity is negative");
    const int words = capacity < kWorkWords ? capacity : kWorkWords;
    PTMI_HIP(c, hipSetDevice(c->device));
    if (words > 0) PTMI_HIP(c, hipMemcpyAsync(out, c->d_work, (size_t)words * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return words;
}

int ptmi_debug_counters(ptmi_ctx *c, uint32_t out[64])
{
    const int rc = ptmi_debug_counters_n(c, out, 64);
    return rc < 0 ? rc : PTMI_OK;
}

int ptmi_reset_stats(ptmi_ctx *c)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    PTMI_HIP(c, hipSetDevice(c->device));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    PTMI_HIP(c, hipMemsetAsync(c->d_live, 0, kLiveBytes, c->stream));
    PTMI_HIP(c, hipMemsetAsync(c->d_iters, 0, kItersBytes, c->stream));
    PTMI_HIP(c, hipMemsetAsync(c->d_work, 0, kWorkWords * sizeof(unsigned int), c->stream));
    PTMI_HIP(c, hipMemsetAsync(c->d_stream_counters, 0, kScWords * sizeof(unsigned long long), c->stream));
    c->nominal = 0; c->samples = 0; c->rays_dropped = 0; c->rays_truncated = 0; c->rays_spilled = 0; c->rays_overflowed = 0; c->live_host = 0;
    return PTMI_OK;
}

static int eval_prims(ptmi_ctx *c, const void *prims, int words, const float *rays, int n,
                      int32_t *is_just, float *t, float *normalp, bool sphere)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (n < 0 || (n > 0 && (!prims || !rays || !is_just || !t))) return fail(c, PTMI_EINVAL, "bad point-query arguments");
    if (n == 0) return PTMI_OK;
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t nb = (size_t)n;
    const size_t bytes = nb * (words * 4 + 6 * 4 + 4 + 4 + 6 * 4);
    if (int rc = ensure_scratch(c, bytes)) return rc;
    float *d_prims = static_cast<float *>(c->scratch);
    float *d_rays = d_prims + nb * words;
    int32_t *d_just = reinterpret_cast<int32_t *>(d_rays + nb * 6);
    float *d_t = reinterpret_cast<float *>(d_just + nb);
    float *d_np = d_t + nb;
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    PTMI_HIP(c, hipMemcpyAsync(d_prims, prims, nb * words * 4, hipMemcpyHostToDevice, c->stream));
    PTMI_HIP(c, hipMemcpyAsync(d_rays, rays, nb * 6 * 4, hipMemcpyHostToDevice, c->stream));
    if (sphere) PTMI_HIP(c, launch_eval_sphere(d_prims, d_rays, n, d_just, d_t, d_np, c->stream));
    else        PTMI_HIP(c, launch_eval_plane(d_prims, d_rays, n, d_just, d_t, d_np, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    PTMI_HIP(c, hipMemcpyAsync(is_just, d_just, nb * 4, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipMemcpyAsync(t, d_t, nb * 4, hipMemcpyDeviceToHost, c->stream));
    if (normalp) PTMI_HIP(c, hipMemcpyAsync(normalp, d_np, nb * 24, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return PTMI_OK;
}

int ptmi_eval_distance_to_sphere(ptmi_ctx *c, const ptmi_sphere *spheres, const float *rays, int n,
                                 int32_t *is_just, float *t, float *hit_normalp)
{
    return eval_prims(c, spheres, 10, rays, n, is_just, t, hit_normalp, true);
}

int ptmi_eval_distance_to_plane(ptmi_ctx *c, const ptmi_plane *planes, const float *rays, int n,
                                int32_t *is_just, float *t, float *hit_normalp)
{
    return eval_prims(c, planes, 12, rays, n, is_just, t, hit_normalp, false);
}

int ptmi_eval_sincos(ptmi_ctx *c, const float *x, int n, float *sin_out, float *cos_out)
{
    if (!c) return PTMI_EINVAL;
    std::lock_guard<std::mutex> lock(c->mu);
    if (n < 0 || (n > 0 && (!x || !sin_out || !cos_out))) return fail(c, PTMI_EINVAL, "bad sincos arguments");
    if (n == 0) return PTMI_OK;
    PTMI_HIP(c, hipSetDevice(c->device));
    const size_t nb = (size_t)n;
    if (int rc = ensure_scratch(c, nb * 12)) return rc;
    float *dx = static_cast<float *>(c->scratch), *ds = dx + nb, *dc = ds + nb;
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    PTMI_HIP(c, hipMemcpyAsync(dx, x, nb * 4, hipMemcpyHostToDevice, c->stream));
    PTMI_HIP(c, launch_eval_sincos(dx, n, ds, dc, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    PTMI_HIP(c, hipMemcpyAsync(sin_out, ds, nb * 4, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipMemcpyAsync(cos_out, dc, nb * 4, hipMemcpyDeviceToHost, c->stream));
    PTMI_HIP(c, hipStreamSynchronize(c->stream));
    return PTMI_OK;
}

}  // extern "C"
