// ptmi_kernels.h -- launch interface between the C ABI (ptmi_api.cpp) and the gfx950 kernels (ptmi_*.hip, one unit per kernel family).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <type_traits>
#include <utility>

#include "ptmi_core.h"

struct ptmi_ctx;

namespace ptmi {

// (between the C ABI's two units: the image size a context stands at, 0 x 0 while it is unsized -- ptmi_group.cpp checks its members with it)
void context_size(const ptmi_ctx *ctx, int *width, int *height);

// Scene as staged into LDS, one float4 stream (see pack_scene in ptmi_api.cpp):
//   [0, ns)                 sphere geometry   (cx, cy, cz, r*r)
//   [ns, ns + 2 np)         plane geometry    (px, py, pz, 0) (nx, ny, nz, 0)
//   [geom, geom + 2 (ns+np)) material         (cr, cg, cb, illuminance) (tag bits, p, p/pi, (1-p)/2)
struct SceneView {
    const float4 *packed;      // device memory
    int n_spheres, n_planes;
    __host__ __device__ int geom_f4() const { return n_spheres + 2 * n_planes; }
    __host__ __device__ int total_f4() const { return geom_f4() + 2 * (n_spheres + n_planes); }
};

struct Planes {
    float *r, *g, *b;
    uint32_t *sa, *sb, *sc, *sctr;
};

struct RenderArgs {
    PrimaryUniforms cam;
    SceneView scene;
    Planes planes;
    const int64_t *screen_x, *screen_y;   // optional explicit Matrix (V2 Int); NULL = implicit
    int width, height;                    // whole image (screenWidth/Height)
    int rows_local;                       // rows held by this context
    int stripe_rows, n_parts, part;       // row-stripe partition
    int bounce_limit, n_spp;
    // device statistics, SHARDED: kStatShards words kStatStride words apart, a wave adds to shard (workgroup & (kStatShards - 1)).
    // One word serves ~90 atomics per microsecond: 7 500 waves of a 1-spp launch ending together took 83 us to count themselves.
    unsigned long long *live_counter;     // += live bounces (the host sums the shards)
    unsigned int *work_counter;           // device counter for dynamic pixel hand-out (variants)
    unsigned int *stream_iterations;      // Streams: steps taken by the last sample (max over waves)
    // Cost-ordered dispatch of the tiled kernels (see lane_pixel in ptmi_device.h).  A "quad" is a run of four
    // x-adjacent 8x8 tiles.  quad_order: the quad each dispatch position works on (NULL = image order).
    // quad_cost: where each wave adds the loop trips it paid (NULL = do not record).
    const unsigned int *quad_order;
    unsigned int *quad_cost;
    // Sample chunks of the tiled render Inline kernel (see render_inline_kernel): spp_chunks = 0 lets the launcher choose,
    // 1 switches them off, k >= 2 forces k copies; chunk_done: one word per tile workgroup (capacity words) and, behind them,
    // the ticket counter of the launch (chunk_done[chunk_capacity]), device memory.
    int cus;                              // compute units of the context's device
    int spp_chunks;
    unsigned int *chunk_done;
    unsigned int chunk_capacity;
    // render Streams only
    int stream_step_cap;                  // traceSteps per ray lineage before the safety cap cuts it (the reference has none)
    int seed_from_result;                 // PTMI_SEED_FROM_RESULT: a hit's ray seed replaces the pixel's (assumption A5)
    unsigned long long *stream_counters;  // device: [kScTruncated] rays cut by the cap, [kScDropped] children that found no room
    // tree walk: the first kTreeFastLevels waiting children of every lane, [tile workgroup][level][lane] records of four float4
    // (tree_stack_tiles workgroups); required by render_streams_tree_kernel
    float4 *tree_stack;
    // render_streams_kernel as the TAIL of the stream form (streams_pixels_kernel's launch): the workgroups start at dispatch position
    // *first_position (a device word: where the stream form's part of the dispatch order ends) and those past the grid do nothing;
    // costs are recorded in the stream form's unit (shaded hits per quad).  NULL = the whole grid, as ever.
    const unsigned int *first_position;
};
enum { kScTruncated = 0, kScDropped = 1, kScWords = 4 };
constexpr int kWorkWords = 256;            // RenderArgs.work_counter: the hand-out counter in [0], the diagnostic builds' counters behind it (ptmi_diag.h; ptmi_debug_counters)
constexpr int kTreeStackDepth = 16;        // per-pixel tree walk: pending children a lane can hold (render_streams_tree_kernel)
#ifndef PTMI_TREE_FAST_LEVELS
#define PTMI_TREE_FAST_LEVELS 4
#endif
constexpr int kTreeFastLevels = PTMI_TREE_FAST_LEVELS;   // ... of which the first few are 64-byte records in global memory (RenderArgs.tree_stack), the rest scratch

// Ray stream of the stream form of Streams: `capacity` records (type RayState, Trace.hs:46) of 16 words = one 64-byte line:
//   [origin xyz, direction x] [direction yz, throughput xy] [throughput z, local pixel index, seed a, seed b]
//   [seed c, seed counter, step index (traceSteps taken by the ray's ancestors = the awhile iteration it belongs to), -]
// A record is written and read as four 16-byte accesses with one address computation (as fifteen word planes every emitted
// child cost fifteen stores with an address each).  One base pointer + the capacity: kernel arguments live in scalar registers.
constexpr int kStatShards = 256, kStatStride = 16;    // 128 B apart (u64) -- u32 shards use the same byte distance (stride 32)

struct RayQueue {
    uint32_t *base;
    unsigned int capacity;  // records
    PTMI_HD float4 *record(unsigned int i) const { return reinterpret_cast<float4 *>(base) + (size_t)i * 4; }
    PTMI_HD uint32_t *pixel_word(unsigned int i) const { return base + (size_t)i * 16 + 9; }     // kHole marks an unused slot
};
constexpr int kRayQueueWords = 16;
constexpr int kCounterStride = 32;          // device counters sit 128 B apart (one per cache line)
constexpr int kStreamStepCapDefault = 1 << 16;   // traceSteps per ray lineage; the reference has no bound (Trace.hs:166-170) -- this only guarantees termination
// Counters of the stream form, each kCounterStride words (one cache line) apart; sums are sharded by workgroup, because one
// word serves only ~90 atomics per microsecond and the 6 144 waves of a level end together:
//   [kLvLive, +kLvLiveShards) children emitted | kLvCut rays cut by the step cap | kLvDropped children that found the output
//   stream full | kLvDeepest deepest step + 1 | kLvSpilled children that went through HBM (spill queues, overflow stream) | kLvSplitPixels pixels whose (glass) primary hit
//   was replaced by its children's hits | per level l, kLvPerLevel lines from kLvCursor + kLvPerLevel l:
//     + 0: the reservation cursor of the stream level l WRITES, counted from the end of its waves' static blocks (grid * first
//          block: LevelArgs.out_base / in_base; that sum = the item count, holes included, of the stream level l + 1 reads)
//     + 1: level l's chunk hand-out | + 2 .. + 2 + kLvEmitShards: the children level l stored (shards)
constexpr int kLvLiveShards = 64, kLvEmitShards = 8, kLvPerLevel = 2 + kLvEmitShards;
constexpr int kLvLive = 0, kLvCut = kLvLiveShards, kLvDropped = kLvCut + 1, kLvDeepest = kLvCut + 2, kLvSpilled = kLvCut + 3, kLvSplitPixels = kLvCut + 4,
              kLvCursor = kLvCut + 5, kLvMaxLevels = 64;
// ... and, behind the per-level lines, the ticket counters of the item kernels: eight (one per XCD queue) per launch of a call
constexpr int kLvTickets = kLvCursor + kLvPerLevel * kLvMaxLevels, kLvMaxLaunches = 32;
constexpr int kLvWords = (kLvTickets + 8 * kLvMaxLaunches) * kCounterStride;
// The hits the samples of the held pixels START from, written once per render call (streams_primary_kernel), in REGIONS: one
// region per 64-pixel tile of the image (a wave of the primary kernel: an 8x8 pixel tile, or 64 consecutive pixels of a row
// for images too small for tiles), in the order the tiles are dispatched (most expensive quad of tiles first once costs
// are known) -- no atomics, the order is deterministic.  A region holds its tile's records compacted to the front (ballot +
// popcount prefix in the wave) and `counts[region]` says how many: usually one per pixel whose primary ray hits, none for
// a pixel whose ray misses.  For a GLASS primary hit -- whose two children are the same two rays in every sample, a glass
// hit draws nothing that changes a direction -- the first hit of each child instead (0, 1 or 2 records), with the
// throughput, the step index and the number of raw draws its ray's seed is ahead of the sample's: regions then have 128 slots.
struct HitList {             // records of 16 words: [position xyz, normal x] [normal yz, direction xy] [direction z, throughput xyz] [primitive, local pixel index, meta, quad]
    uint32_t *base;
    uint32_t *slot_key;      // per record slot: the record's pixel (30 bits) | the code of its raw draws (0, 3, 4 -> 0, 1, 2) << 30 -- all that
                             // streams_slot_seeds_kernel needs of a record: 4 bytes per slot and call instead of the record's 64-byte line
    unsigned int *counts;    // records per region
    unsigned long long *missed;   // per region: the lanes (pixels of the tile) that have no start hit
    unsigned int region_slots;   // 64, or 128 when a primary hit can split
    unsigned int n_regions;
    PTMI_HD float4 *record(unsigned int i) const { return reinterpret_cast<float4 *>(base) + (size_t)i * 4; }
};
constexpr int kHitListWords = 16;
// Overflow levels of the stream form (streams_level_kernel): children that found the wave's ring full travel through HBM.
struct LevelArgs {
    RayQueue in, out;
    const unsigned int *in_count;   // device: the producer's reservation cursor
    unsigned int in_base;           // ... which counts from here (the producer's out_base)
    unsigned int *out_count;        // device: this level's reservation cursor, zero at launch
    unsigned int out_base;          // grid * (first block size): where reserved blocks start
    unsigned int *emitted;          // device: children this level stored in `out`: kLvEmitShards words, kCounterStride apart
    unsigned int *stats;            // device: base of the counter block
    int may_emit;                   // 0: the scene has no ray-splitting material -- nothing is ever written to `out`
};
// The item kernels of the stream form (streams_pixels_kernel, streams_split_kernel): persistent waves take the regions of the
// start-hit list as chunks, by ticket, from eight queues (one per XCD; ChunkCursor in ptmi_stream_form.h).
struct ItemArgs {
    HitList hits;
    unsigned int n_positions;       // groups of four regions (dispatch positions): hits.n_regions / 4
    const unsigned int *tail_start; // streams_pixels_kernel: a device word -- the positions from there on are left to the per-pixel kernel (NULL: none)
    int passes;                     // tickets run over the chunks this many times: a pixel's samples in that many items
    const int *group_first;         // device, groups + 1 entries (NULL: every pass on its own): the passes are handed out in GROUPS of consecutive passes
    int groups;                     // [group_first[g], group_first[g + 1]), a group region by region -- region r in every pass of the group, then region r + 1: a
                                    // region's items of one group follow each other through one ticket queue, and all but the first find its records and
                                    // colour lines in that XCD's L2 (decode_ticket).  A group of one pass is that pass, pass by pass as ever
    unsigned int *region_done;      // streams_pixels_kernel, passes > 1: per region, the items published so far (zero at launch)
    int fenced;                     // ... 1: release / acquire at agent scope, once per (region, pass); 0: the fence-free write-through hand-off (PTMI_OPT_PASS_HANDOFF)
    unsigned int *chunk_cursor;     // device: the launch's eight ticket counters, kCounterStride words apart, zero at launch
    // streams_split_kernel only
    const int *pass_first;          // device, passes + 1 entries: pass p renders samples [pass_first[p], pass_first[p + 1]) of its pixels -- long items first, short ones
                                    // at the end of the launch, which is as long as its last items (stream_schedule in ptmi_api.cpp)
    const uint4 *seed_snapshots;    // [passes][n_slots]: the seed the item (record slot, pass) starts from (streams_slot_seeds_kernel)
    int glass_batch;                // > 1: GLASS hits wait in their lanes until that many are pending in the wave (PTMI_OPT_GLASS_BATCH)
    unsigned int n_slots;           // slots of the start-hit list: hits.n_regions * hits.region_slots
    RayQueue spill;                 // the waves' own spill queues: streams_spill_records() records each, gridDim of them
    RayQueue out;                   // overflow stream: children that found ring and spill queue full
    unsigned int *out_count;        // device: reservation cursor of `out`, zero at launch
    unsigned int out_base;
    unsigned int *emitted;
    unsigned int *stats;
    int may_emit;
};

// Ticket j of a queue of n regions -> (pass, the region's index k in the queue), j < n * passes.  Without a group table: pass by pass.  With one:
// group g owns tickets [n group_first[g], n group_first[g + 1]) and hands them out region by region.  Host and device: the kernels' next_chunk and
// ptmi_stream_tickets (the order as a pure function, tested without a GPU).  (The scan is wave-uniform scalar work, once per ticket -- an item is
// hundreds to thousands of loop trips; a graded schedule has a handful of groups.)
PTMI_HD void decode_ticket(unsigned int j, unsigned int n, const int *group_first, int groups, unsigned int &pass, unsigned int &k)
{
    if (!group_first) {
        pass = j / n;
        k = j - pass * n;
        return;
    }
    int g = 0;
    while (g + 1 < groups && j >= n * (unsigned int)group_first[g + 1]) ++g;
    const unsigned int first = (unsigned int)group_first[g], size = (unsigned int)group_first[g + 1] - first, rem = j - n * first;
    k = rem / size;
    pass = first + (rem - k * size);
}

hipError_t launch_streams_pixels(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream);
hipError_t launch_streams_split(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream);
int streams_pixels_waves();           // waves per SIMD the item kernels are built for (persistent grids)
int streams_split_waves();
int streams_min_pass_samples();       // ordered passes: a pass must hold at least this many samples
unsigned int streams_spill_records();   // records of a wave's spill queue in HBM
unsigned int streams_regions(int width, int rows_local);    // regions of the start-hit list
hipError_t launch_streams_seeds(Planes p, HitList hits, uint4 *snapshots, long long n, int passes, const int *pass_first, int draws, hipStream_t stream);   // pass_first: device, passes + 1 entries
hipError_t launch_streams_level(const RenderArgs &a, const LevelArgs &lv, unsigned int grid, hipStream_t stream);
hipError_t launch_streams_primary(const RenderArgs &a, HitList hits, unsigned int *counters, hipStream_t stream);
// updateSeeds for the pixels without start hits (the ordered item kernel advances the others itself)
hipError_t launch_streams_advance_missed(const RenderArgs &a, HitList hits, int draws, const unsigned int *tail_start, hipStream_t stream);
hipError_t launch_render_streams_tail(const RenderArgs &a, const unsigned int *first_position, hipStream_t stream);
unsigned int streams_first_block();   // output slots every wave of a level owns from the start

hipError_t launch_render_inline(const RenderArgs &a, int variant, hipStream_t stream);
bool variant_available(int variant);     // ablation variants exist only in builds with -DPTMI_ABLATIONS
hipError_t launch_render_inline_ablation(const RenderArgs &a, int variant, bool big_scene, hipStream_t stream);   // ptmi_inline_ablations.hip (-DPTMI_ABLATIONS)
hipError_t launch_render_streams(const RenderArgs &a, int variant, hipStream_t stream);
hipError_t launch_render_streams_tree(const RenderArgs &a, int variant, hipStream_t stream);   // scenes with GLASS: per-pixel tree walk
unsigned int tree_workgroups(int width, int rows_local);   // workgroups per copy of its grid (RenderArgs.tree_stack holds kTreeFastLevels x 64 records of 64 B for each)
// 8x8 tiles leave lanes idle on the right and bottom edges; rows of 64 leave them idle at the end only
inline bool tiles_pay_dims(int width, int rows_local) { return width >= 64 && rows_local >= 16; }
inline bool tiles_pay(const RenderArgs &a) { return tiles_pay_dims(a.width, a.rows_local); }
// entries of quad_order / quad_cost (0 = tiles not used): the tile grid padded to a multiple of 32 (lane_pixel), four tiles to a quad
inline unsigned int quad_positions(int width, int rows_local)
{
    if (!tiles_pay_dims(width, rows_local)) return 0;
    const unsigned int tiles = (unsigned int)(((width + 7) / 8) * ((rows_local + 7) / 8));
    return ((tiles + 31u) & ~31u) / 4u;
}
hipError_t launch_quad_order(const unsigned int *cost, unsigned int *order, unsigned int *cls, unsigned int n, unsigned int *tail_start,
                             unsigned int tail_permille, hipStream_t stream);
bool uses_quad_order(const RenderArgs &a, int algorithm_inline, int variant);
hipError_t launch_seed(Planes p, int width, int rows_local, int stripe_rows, int n_parts, int part,
                       uint64_t seed0, bool clear_color, hipStream_t stream);
hipError_t launch_create_with(Planes p, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2,
                              int64_t n, hipStream_t stream);
hipError_t launch_eval_sphere(const float *spheres10, const float *rays, int n,
                              int32_t *is_just, float *t, float *normalp, hipStream_t stream);
hipError_t launch_eval_plane(const float *planes12, const float *rays, int n,
                             int32_t *is_just, float *t, float *normalp, hipStream_t stream);
// group read-out: member `part`'s snapshot [3][rows][W] (stripes contiguous) into the whole image's planes [H][W]
hipError_t launch_stitch(const float *src, int rows, int width, int stripe_rows, int n_parts, int part,
                         float *r, float *g, float *b, hipStream_t stream);
hipError_t launch_present(Planes p, long long n, int iterations, float *rgb, uint32_t *rgba, hipStream_t stream);
hipError_t launch_eval_sincos(const float *x, int n, float *s, float *c, hipStream_t stream);

// A launch that reports ITS OWN status.  `kernel<<<...>>>(...)` drops hipLaunchKernel's result, and hipGetLastError() afterwards hands out the
// thread's STICKY error -- the last failure of any runtime call, this library's or another's (an allocation the caller recovered from, an error
// ptmi_destroy ignored), kept until somebody asks: a launch that went out would be reported as failed.  Arguments are converted to the
// kernel's parameter types as a call would convert them.
template <typename Tuple, size_t... I>
inline hipError_t launch_packed(const void *kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, Tuple &values, std::index_sequence<I...>)
{
    void *pointers[] = {static_cast<void *>(&std::get<I>(values))..., nullptr};
    return hipLaunchKernel(kernel, grid, block, pointers, lds_bytes, stream);
}
template <typename... P, typename... A>
inline hipError_t launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, const A &...args)
{
    static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
    std::tuple<std::remove_cv_t<P>...> values(static_cast<P>(args)...);
    return launch_packed(reinterpret_cast<const void *>(kernel), grid, block, lds_bytes, stream, values, std::index_sequence_for<P...>{});
}

}  // namespace ptmi
