// ptmi_stream_form.h -- what the kernels of the stream ("wavefront") form of render Streams share: the start-hit list's pixel
// mapping, the chunk cursor over its regions (eight ticket queues, one per XCD), ray records, agent-scope accesses.
#pragma once

#include "ptmi_device.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// render Streams as a stream ("wavefront" form), the kernels.
//
//   streams_primary_kernel   every sample of a pixel shoots the same primary ray (Trace.hs:244-262), so its checkHit + hit are
//                            evaluated ONCE per render call; the hits the samples start from go into the start-hit list, in
//                            regions of one 64-pixel tile each, compacted inside the wave by ballot + popcount prefix (HitList);
//                            pixels whose primary ray misses never enter a stream.
//   streams_pixels_kernel    scenes whose rays never split (and PTMI_OPT_STREAM_BATCH = 0): one launch, persistent waves take
//                            the regions as chunks (the first gridDim statically, later ones by ticket), a lane takes a start
//                            hit and renders ALL samples of its pixel from it, colour and seed in registers -- read once,
//                            written once, no atomics, additions in sample order: bit-identical to the per-pixel kernel and
//                            the oracle under both seed rules.  `expand` (Trace.hs:284-289) with numNewRays in {0, 1} is "the
//                            child is the lane's next ray"; a lane whose pixel is done REFILLS from the wave's chunk (ballot of
//                            the idle lanes + popcount prefix), so the rounds stay dense although pixels differ in cost.
//   streams_split_kernel     scenes with a ray-splitting material (the build-defined GLASS), or samples cut into unordered
//                            items (PTMI_OPT_STREAM_BATCH): the same persistent shape; an item is (start hit, a range of the
//                            pixel's samples).  At a GLASS hit the reflection stays in the lane and the refraction goes into the
//                            wave's CHILD RING in LDS -- `expand` as wave-level compaction: ballot of the emitting lanes,
//                            popcount prefix for the slot -- from which lanes that have no ray take their next one (ballot +
//                            prefix again) before they start their item's next sample.  Nearly every child is traced by the
//                            wave that emitted it, in the same launch; only when the ring is full does a child travel through
//                            the overflow stream in HBM (blocks of slots reserved per wave, one atomic per 256 children).
//                            `permute (+)` (Trace.hs:179-184): a lane's own lineages add into its LDS accumulator, flushed with
//                            one float atomic per colour word per item; rays taken from the ring add with float atomics (exact
//                            zeros skipped).  The order of a pixel's additions is undefined, as in Accelerate's permute.
//   streams_level_kernel     the overflow levels: one launch per level of what is left in the HBM stream (usually nothing).
//   streams_slot_seeds_kernel / streams_advance_seeds_kernel (split kernel only)   the seed every item starts from, per record slot of the
//                            start-hit list and per pass, recorded BEFORE updateSeed (Trace.hs:190-191) moves every pixel's seed on by the
//                            call's samples.
// Every ray carries its step index (the `awhile` iteration it belongs to), so the safety cap cuts the same rays as in
// the other forms; cut rays, dropped children (overflow stream full) and emitted children are counted.
// ---------------------------------------------------------------------------------------
constexpr unsigned int kHole = 0xffffffffu;                  // pixel word of an unused output slot
constexpr unsigned int kFirstBlock = 64, kNextBlock = 256;   // output slots a wave owns at start / reserves per atomic

// 16 bytes past the L1 (global_load_dwordx4 ... nt): for records this wave wrote itself a few trips ago
__device__ __forceinline__ float4 load_past_l1(const float4 *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return float4{v.x, v.y, v.z, v.w};
}
// Element `byte_off / 4` of a plane through a 32-bit byte offset (the stream form holds < 2^30 pixels): the access becomes
// scalar base + 32-bit vector offset instead of a 64-bit address pair per plane.
template <typename T> __device__ __forceinline__ T &plane_at(T *base, uint32_t byte_off)
{
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}
// agent-scope relaxed accesses (global_load / global_store ... sc1): the load passes the CU's L1 by, the store is written through
__device__ __forceinline__ float load_agent(const float *p) { return u2f(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ uint32_t load_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(float *p, float v) { __hip_atomic_store(reinterpret_cast<uint32_t *>(p), f2u(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The start-hit list.  One workgroup = one quad of four x-adjacent 8x8 tiles (TILES) or 256 consecutive pixels, one wave
// = one region.  Dispatch position p works on quad quad_order[p] (most expensive first, once costs are known), and its
// regions are 4 p .. 4 p + 3: the list is in dispatch order, which is the order the item kernels hand the chunks out.
// A glass primary hit is replaced by the first hits of its two children when that changes nothing observable: the
// step cap cannot cut the children (>= 3) and the glass hit itself emits nothing (its emittance would have to be added
// once per sample).  advance_missed: a pixel without start hits gets its updateSeeds here (streams_pixels_kernel does
// the others' itself).  `counters`: the stream form's counter block (kLvSplitPixels, kLvDeepest).
// which pixel a lane of the primary kernel (and of the kernel that advances the missed pixels' seeds) looks at
template <bool TILES>
__device__ __forceinline__ bool primary_pixel(const RenderArgs &a, unsigned int &quad, unsigned int &region, long long &pixel)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int position = blockIdx.x;
    quad = (TILES && a.quad_order) ? a.quad_order[position] : position;
    region = position * 4u + (unsigned int)wave;
    if (TILES) {
        const unsigned int tile = quad * 4u + (unsigned int)wave;
        const int tiles_x = (a.width + 7) / 8;
        const int tx = (int)(tile % (unsigned)tiles_x), ty = (int)(tile / (unsigned)tiles_x);
        const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
        pixel = (long long)y * a.width + x;
        return x < a.width && y < a.rows_local;
    }
    pixel = (long long)region * 64 + lane;
    return pixel < (long long)a.rows_local * a.width;
}

__device__ __forceinline__ void queue_store(const RayQueue &q, unsigned int i, V3 o, V3 d, V3 t, uint32_t pixel, Sfc32 s, uint32_t depth)
{
    float4 *r = q.record(i);
    r[0] = float4{o.x, o.y, o.z, d.x};
    r[1] = float4{d.y, d.z, t.x, t.y};
    r[2] = float4{t.z, u2f(pixel), u2f(s.a), u2f(s.b)};
    r[3] = float4{u2f(s.c), u2f(s.counter), u2f(depth), 0.0f};
}

// The chunk cursor of the item kernels.  A chunk is one region of the start-hit list (its records, up to 64 or 128); the regions come in groups
// of four, one group per dispatch POSITION (the four tiles of a quad, most expensive quad first).  The positions are dealt to
// eight queues, position p to queue p mod 8 -- one queue per XCD -- so that the tiles of a quad, whose pixels share cache lines
// of the planes, are worked on behind ONE L2 (dealt to any XCD, every line of the planes was fetched four times).  A queue is a
// ticket counter: ticket j stands for a (pass, region) of the queue's n regions -- pass (j div n), region (j mod n), passes outermost, or,
// with a group table, groups of passes outermost and each group region by region (decode_ticket, ptmi_kernels.h) -- positions in dispatch order.  A wave takes tickets -- one returning atomic each; an item is tens to thousands of loop trips
// -- from the queue of the XCD it runs on (HW_REG_XCC_ID; which wave works on which chunk changes no result) and, when that one
// is exhausted, from the other XCDs' queues.  Eight counters instead of one: a single word serves ~90 atomics per microsecond
// and thousands of waves start together.  Regions without records (tiles whose primary rays all miss) are skipped.
struct ChunkCursor {
    unsigned int taken, len, first, pass;    // of the chunk in hand: records handed out, records, first slot, pass
    unsigned int region;                     // ... its region
    unsigned int home, tries;                // the wave's XCD; queues found exhausted (8: nothing is left)
    unsigned int n_positions;                // dispatch positions the kernel works on (the first ones of the order)
    bool ready;                              // (ordered passes) the chunk's previous pass has been published and acquired
};
__device__ __forceinline__ unsigned int xcc_id()
{
    return (unsigned int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;     // HW_REG_XCC_ID, bits [3:0]
}
__device__ __forceinline__ bool chunks_left(const ChunkCursor &c) { return c.tries < 8u; }
template <bool GROUPS>      // GROUPS: the kernel may be handed a group table (the split kernel).  The pixels kernel's ordered passes never are, and the scan it
                            // would never run still cost it 1 % as code in its loop (4.34 -> 4.30 ms on C2 through Streams, round 5): compiled out there
__device__ __forceinline__ void next_chunk(ChunkCursor &c, const ItemArgs &it)
{
    // A chunk is a WHOLE region (64 slots, or 128 where a glass primary hit contributes two records).  Until round 4 a 128-slot region was two
    // chunks of 64 with a ticket each: for every tile without split pixels -- nearly all -- the second ticket found nothing, and a ticket is a
    // returning atomic followed by a load of the region's record count that needs its answer: two memory latencies in a row with nothing for the
    // wave to do, paid twice per useful chunk.  Glass 1080p 7.91 -> 7.71 ms, C5 part 29.2 -> 28.5.  (Going further -- one ticket per dispatch
    // POSITION, four regions and one 16-byte load of their counts -- is four times fewer tickets and much worse: a wave then owns up to 512
    // items of one pass, heavy quads pile up in single waves, 8.05 ms and 36.4 with 250 000 rays in the overflow stream; the pixels kernel,
    // whose items are whole pixels, went from 4.24 to 4.82 ms.)
    c.taken = 0; c.len = 0; c.ready = false;
    while (c.tries < 8u) {
        const unsigned int q = (c.home + c.tries) & 7u;
        // positions in queue q: p = 8 s + q < n_positions
        const unsigned int n_pos = c.n_positions > q ? (c.n_positions - q - 1u) / 8u + 1u : 0u;
        const unsigned int n = n_pos * 4u;
        unsigned int j = 0;
        if ((threadIdx.x & 63) == 0) j = atomicAdd(it.chunk_cursor + (size_t)q * kCounterStride, 1u);
        j = (unsigned int)__builtin_amdgcn_readfirstlane((int)j);
        if (n == 0u || j / n >= (unsigned int)it.passes) { ++c.tries; continue; }
        unsigned int k;
        decode_ticket(j, n, GROUPS ? it.group_first : nullptr, it.groups, c.pass, k);
        const unsigned int s_pos = k >> 2, r = k & 3u;
        c.region = (s_pos * 8u + q) * 4u + r;
        c.first = c.region * it.hits.region_slots;
        c.len = it.hits.counts[c.region];
        if (c.len) return;
    }
}

// what an item cost, for the dispatch order of later launches with the same key: the loop trips the lane spent on it (or the
// hits it shaded, one per trip)
__device__ __forceinline__ void record_item_cost(const RenderArgs &a, unsigned int quad, unsigned int trips)
{
    if (a.quad_cost) atomicAdd(a.quad_cost + quad, trips);
}

}  // namespace

}  // namespace ptmi
