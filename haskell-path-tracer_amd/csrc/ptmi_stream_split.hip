// ptmi_stream_split.hip -- the stream form of render Streams for scenes whose rays SPLIT (the build-defined GLASS extension) or
// whose samples are cut into unordered items: persistent waves, a child ring in LDS per wave, spill queues and overflow levels.
#include "ptmi_stream_form.h"

namespace ptmi {

namespace {

#ifndef PTMI_LEVEL_WAVES
#define PTMI_LEVEL_WAVES 6
#endif
#ifndef PTMI_REFILL_BATCH
#define PTMI_REFILL_BATCH 8
#endif
constexpr unsigned int kRefillBatch = PTMI_REFILL_BATCH;     // overflow levels: idle lanes a wave waits for before it runs the refill block

// ---------------------------------------------------------------------------------------
// streams_split_kernel: items of (start hit, sample range), a child ring per wave.  Loop shape
// [finish dead hits][refill items][next ray: ring, else the wave's spill queue, else the item's next sample][shade][expand][trace].
// Where the children of a wave wait, in this order: the RING in LDS (kRing records: nearly all of them, for one or two
// trips); when the ring is full, the wave's own SPILL QUEUE in HBM (kSpill 64-byte records that only this wave writes and
// reads: same-wave program order, no cross-wave visibility question); when that is full too, the overflow stream that
// a later launch reads (streams_level_kernel) -- which a render call practically never needs.
// ---------------------------------------------------------------------------------------
#ifndef PTMI_SPLIT_WAVES
#define PTMI_SPLIT_WAVES 6
#endif
#ifndef PTMI_RING
#define PTMI_RING 16
#endif
constexpr unsigned int kRing = PTMI_RING;                     // records of a wave's child ring (a power of two, <= 64)
#ifndef PTMI_SPILL
#define PTMI_SPILL 4096
#endif
// records of a wave's spill queue in HBM (a power of two; 256 KB per wave, 1.6 GB for the 6 144 waves of a launch).  A wave that
// works through the inside of a glass sphere emits up to 64 children per trip and places 30: with 256 records 0.03 % of the glass
// scene's children went on to the overflow stream -- four more launches and read-backs per call, 8.76 ms; 1 024: 47 rays, 8.61;
// 4 096: none, 8.49.
constexpr unsigned int kSpill = PTMI_SPILL;
#ifndef PTMI_ITEM_BATCH
#define PTMI_ITEM_BATCH 1
#endif
constexpr unsigned int kItemBatch = PTMI_ITEM_BATCH;          // lanes without an item a wave waits for before it runs the refill block (its loads stall the whole wave)
template <bool LDS_SCENE, bool TILES>
__global__ void __launch_bounds__(kRenderBlock, PTMI_SPLIT_WAVES) streams_split_kernel(const RenderArgs a, const ItemArgs it)
{
    // the lane's own item: 0-2 position of the start hit, 3-5 normal, 6-8 incoming direction, 9-11 throughput,
    // 12 primitive (10 bits) | step index of the start hit (0 or 1) << 10 | the pixel's quad << 11 (where the item's cost goes), 13 pixel,
    // 14-17 the seed the ray of its next sample carries.  (Eighteen words and no more: with the ring and a 16-primitive scene a workgroup needs
    // 6 368 bytes of LDS, five of the 1 280-byte granules gfx950 allocates, and 24 workgroups -- six waves per SIMD -- fit a CU; a nineteenth
    // word is a sixth granule and 21 workgroups: measured, the launch ran 4 % longer with an eighth of its waves starting when the others ended.)
    __shared__ float item_rec[18][kRenderBlock];
    __shared__ uint32_t ring[15][kRing];                      // children waiting for a lane: RayQueue's record, word by word (15 words)
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
        stage_glass_constants(lds_scene, a.scene);
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned int w = blockIdx.x;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    float *mine = &item_rec[0][threadIdx.x];
    auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
    auto get = [&](int k) { return mine[k * kRenderBlock]; };

    ChunkCursor cur; cur.home = xcc_id(); cur.tries = 0; cur.n_positions = it.n_positions;
    next_chunk<true>(cur, it);
    unsigned int ring_head = 0, ring_n = 0;                   // wave-uniform
    unsigned int spill_head = 0, spill_n = 0;                 // wave-uniform: the wave's spill queue, records [w kSpill, (w + 1) kSpill) of it.spill
    unsigned int blk = w * kFirstBlock, blk_end = blk + kFirstBlock;   // wave-uniform: the overflow block being filled

    bool busy = false, foreign = false, has_ray = false, pending = false;   // own item in hand / the ray came from the ring / a ray to trace / a hit to shade
    V3 o = mk(0, 0, 0), d = o, throughput = o, normal = o;    // o: the ray's origin, or the position of the pending hit
    V3 own_acc = o;                                           // what the lane's own lineages have added for its item's pixel
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    uint32_t pixel = 0, depth = 0;                            // depth: step index of the lane's current ray
    int idx = 0, samples_left = 0;
    unsigned int deepest = 0, item_trips = 0;                 // item_trips: loop trips since the lane took its item (its cost, for later launches' dispatch order)
    unsigned int live_w = 0, cut_w = 0, spilled_w = 0;        // wave-uniform statistics

    // computeResult + permute (+) (Trace.hs:179-184, :318-323); adding an exact zero changes nothing
    auto add_colour = [&](V3 c) __attribute__((always_inline)) {
        if (foreign) {
            if (diag::kTrafficSkip & 2u) return;
            if (c.x != 0.0f) atomicAdd(&plane_at(a.planes.r, pixel << 2), c.x);
            if (c.y != 0.0f) atomicAdd(&plane_at(a.planes.g, pixel << 2), c.y);
            if (c.z != 0.0f) atomicAdd(&plane_at(a.planes.b, pixel << 2), c.z);
        } else {
            own_acc = own_acc + c;
        }
    };

    diag::SplitProbe probe; probe.begin(a.work_counter);                   // (diagnostic builds: ptmi_diag.h)
    for (;;) {
        probe.stamp(7);
        probe.trip(pending && near_zero(throughput), busy);
        // ---- a hit whose ray arrived with near-zero throughput (numNewRays = 0, Trace.hs:329-331) adds its emittance and nothing
        // else of it survives: the lineage ends here
        if (pending && near_zero(throughput)) {
            const float4 ma = M[2 * idx];
            add_colour(scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
            pending = false;
        }
        probe.stamp(0);
        // ---- refill: lanes without an item take the next ones of the wave's chunk (whatever ray they are tracing meanwhile)
        const unsigned long long empty = __ballot(!busy);
        if ((unsigned int)__builtin_popcountll(empty) >= kItemBatch && chunks_left(cur)) {     // wave-uniform
            probe.refill();
            const unsigned int want = (unsigned int)__builtin_popcountll(empty), avail = cur.len - cur.taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = rank_in(empty);
            if (!busy && rank < take) {
                // initialState (Trace.hs:158-162) one step on: the cached start hit, and the seed the ray of the item's first sample carries -- the
                // pixel's seed after pass_first[pass] updateSeeds and, for a child of a cached glass primary hit, the 3 or 4 raw draws its ancestors
                // made (streams_slot_seeds_kernel); two INDEPENDENT loads, both addressed by the slot.  (Indexed by the pixel, the snapshot waited
                // for the record: two memory latencies in a row in three trips of four -- 8.09 -> 7.93 ms on the glass scene.)
                const unsigned int slot = cur.first + cur.taken + rank;
                const float4 *r = it.hits.record(slot);
                const uint4 snap = it.seed_snapshots[(size_t)cur.pass * it.n_slots + slot];
                const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                Sfc32 s0; s0.a = snap.x; s0.b = snap.y; s0.c = snap.z; s0.counter = snap.w;
                put(0, r0.x); put(1, r0.y); put(2, r0.z); put(3, r0.w); put(4, r1.x); put(5, r1.y);
                put(6, r1.z); put(7, r1.w); put(8, r2.x); put(9, r2.y); put(10, r2.z); put(11, r2.w);
                put(12, u2f(f2u(r3.x) | ((f2u(r3.z) & 1u) << 10) | (f2u(r3.w) << 11))); put(13, r3.y);
                put(14, u2f(s0.a)); put(15, u2f(s0.b)); put(16, u2f(s0.c)); put(17, u2f(s0.counter));
                // an item's cost (its loop trips) is recorded for later launches' dispatch order by the items of pass 0 ONLY -- the bit above the
                // trip count says "do not record".  One atomic per item was 17.8 M atomics per 1080p call on 8 100 words that all eight XCDs
                // share: 577 MB of the split kernel's 688 MB of HBM writes (tools/traffic_terms.py, round 5); the longest pass ranks the quads as well.
                item_trips = cur.pass == 0u ? 0u : 0x80000000u;
                samples_left = it.pass_first[cur.pass + 1u] - it.pass_first[cur.pass];      // the samples of this pass (ItemArgs.pass_first)
                busy = true;
            }
            cur.taken += take;
            if (cur.taken >= cur.len) next_chunk<true>(cur, it);
            // (Tried and dropped, round 4: touching the records of the NEXT refill a trip early -- 8 or 16 lanes load one word each: 8.21 / 8.25
            // ms against 8.09; taking the next chunk one chunk early to touch all of its records and then their seed snapshots: 8.16; taking only
            // its TICKET and record count early, so that the switch to the next chunk waits for nothing: 8.01 against 7.93, C5 part 30.0 against
            // 29.1 -- a wave that sits on two chunks takes from the queues earlier than it works.  Extra loads cost more than the latency they hide.
            // Later in the round, on the final kernel: this block moved behind the "next ray" block with its eighteen words loaded straight into the lanes'
            // LDS columns by global_load_lds_dword (tools/lds_dma_probe.hip), so that nothing waits for them: + 1 to 2 %.  HISTORY.md, "Round 4".)
        }
        probe.tickets(chunks_left(cur), busy, samples_left, spill_n, ring_n);
        probe.stamp(1);
        // ---- the next ray of every lane that holds neither a ray nor a hit: a child from the wave's ring first ...
        const bool free_lane = !pending && !has_ray;
        const unsigned long long free_m = __ballot(free_lane);
        probe.next_ray(free_m, ring_n, free_lane && busy && samples_left > 0, free_lane && busy && samples_left <= 0);
        if (ring_n == 0u && spill_n && free_m) {              // wave-uniform, rare: the ring is empty -- up to kRing records of the wave's spill queue move into it
            // the records were written by this wave, at least a trip ago: once its stores have been acknowledged (they have: the
            // wait is free) they are in the L2, and loads that bypass the L1 see them.  (Spill -> ring -> lane rather than spill -> lane: the
            // lanes' ray state then has ONE source besides their own item, and the compiler keeps one copy of it.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int n = spill_n < kRing ? spill_n : kRing;
            if ((unsigned int)lane < n) {
                const float4 *r = it.spill.record(w * kSpill + ((spill_head + (unsigned int)lane) & (kSpill - 1u)));
                const float4 r0 = load_past_l1(r), r1 = load_past_l1(r + 1), r2 = load_past_l1(r + 2), r3 = load_past_l1(r + 3);
                const unsigned int slot = (ring_head + (unsigned int)lane) & (kRing - 1u);
                ring[0][slot] = f2u(r0.x); ring[1][slot] = f2u(r0.y); ring[2][slot] = f2u(r0.z);
                ring[3][slot] = f2u(r0.w); ring[4][slot] = f2u(r1.x); ring[5][slot] = f2u(r1.y);
                ring[6][slot] = f2u(r1.z); ring[7][slot] = f2u(r1.w); ring[8][slot] = f2u(r2.x);
                ring[9][slot] = f2u(r2.y);
                ring[10][slot] = f2u(r2.z); ring[11][slot] = f2u(r2.w); ring[12][slot] = f2u(r3.x); ring[13][slot] = f2u(r3.y);
                ring[14][slot] = f2u(r3.z);
            }
            spill_head = (spill_head + n) & (kSpill - 1u); spill_n -= n; ring_n = n;
        }
        bool took = false;
        {
            const unsigned int want = (unsigned int)__builtin_popcountll(free_m);
            const unsigned int take = want < ring_n ? want : ring_n;       // (0 when the ring is empty: the block below then runs for no lane)
            const unsigned int rank = rank_in(free_m);
            if (free_lane && rank < take) {
                const unsigned int slot = (ring_head + rank) & (kRing - 1u);
                o = mk(u2f(ring[0][slot]), u2f(ring[1][slot]), u2f(ring[2][slot]));
                d = mk(u2f(ring[3][slot]), u2f(ring[4][slot]), u2f(ring[5][slot]));
                throughput = mk(u2f(ring[6][slot]), u2f(ring[7][slot]), u2f(ring[8][slot]));
                pixel = ring[9][slot];
                seed.a = ring[10][slot]; seed.b = ring[11][slot]; seed.c = ring[12][slot]; seed.counter = ring[13][slot];
                depth = ring[14][slot];
                has_ray = true; foreign = true; took = true;
            }
            ring_head = (ring_head + take) & (kRing - 1u); ring_n -= take;
        }
        // ---- ... else the next sample of its own item (its start hit and that sample's seed are in the lane's LDS column); an
        // item without samples left is over: its colour goes to the planes, one atomic per word
        if (free_lane && !took && busy) {
            if (samples_left > 0) {
                --samples_left;
                o = mk(get(0), get(1), get(2));
                normal = mk(get(3), get(4), get(5));
                d = mk(get(6), get(7), get(8));
                throughput = mk(get(9), get(10), get(11));
                const uint32_t pm = f2u(get(12));
                idx = (int)(pm & 0x3ffu);
                pixel = f2u(get(13));
                Sfc32 ss; ss.a = f2u(get(14)); ss.b = f2u(get(15)); ss.c = f2u(get(16)); ss.counter = f2u(get(17));
                seed = ss;                                     // (already past the draws its ray's ancestors made)
                (void)sfc32_next(ss);                          // updateSeed (Trace.hs:190-191): the next sample starts one draw further
                put(14, u2f(ss.a)); put(15, u2f(ss.b)); put(16, u2f(ss.c)); put(17, u2f(ss.counter));
                depth = (pm >> 10) & 1u;
                deepest = deepest > 1u ? deepest : 1u;        // the primary ray's traceStep
                pending = true; foreign = false;              // (a start hit of a dead ray -- a reflection of weight ~0 -- waits for the next trip's first block)
            } else {
                const uint32_t px = f2u(get(13));
                if (!(diag::kTrafficSkip & 1u)) {
                    if (own_acc.x != 0.0f) atomicAdd(&plane_at(a.planes.r, px << 2), own_acc.x);
                    if (own_acc.y != 0.0f) atomicAdd(&plane_at(a.planes.g, px << 2), own_acc.y);
                    if (own_acc.z != 0.0f) atomicAdd(&plane_at(a.planes.b, px << 2), own_acc.z);
                }
                own_acc = mk(0.0f, 0.0f, 0.0f);
                if (TILES && !(item_trips >> 31) && !(diag::kTrafficSkip & 4u)) record_item_cost(a, f2u(get(12)) >> 11, item_trips);     // (the quad travels with the start hit: no division here, where two lanes of 64 are active)
                busy = false;
            }
        }
        // no lane holds a ray or a hit: every lane was free, so ring and spill queue are empty (64 free lanes would have taken
        // from them) and no lane holds an item with samples left.  (With chunks left nothing below has a lane to run for and the
        // next trip refills: a `continue` here would be a second back edge, and cost the loop its register allocation.)
        if (!__any(has_ray || pending) && !chunks_left(cur) && !__any(busy) && ring_n == 0 && spill_n == 0) break;
        item_trips += busy ? 1u : 0u;
        probe.stamp(2);

        // ---- shade round, for the hits of rays that are alive (a dead one just fetched waits for the next trip's first block).
        // GLASS is an ARM of the one shade, not a second shade: genVec's three draws come before the match for every material
        // (Trace.hs:394-405), ia = dir . n and the mirror direction are what Glossy computes too, and the reflection child is what the
        // common tail makes of `next = reflection` and the factor R instead of brdf * prob -- p + next ^* epsilon, throughput * (color ^* f),
        // the seed after the three draws.  What only GLASS needs -- Schlick's R, the refraction direction, the SECOND child -- is the small
        // divergent block in the middle (it used to be the whole of glass_children, wave-wide for three lanes in nearly every trip).
        // PARKING (it.glass_batch > 1): a GLASS hit waits in its lane until that many of the wave's lanes hold one -- or the wave has
        // nothing else to shade or trace -- so that the block and the expand behind it run for >= glass_batch lanes at a time.
        const float4 mb = M[2 * idx + 1];
        const bool alive = pending && !near_zero(throughput);
        const bool glass_hit = alive && f2u(mb.x) == 2u;
        bool shade_now = alive;
        if (it.glass_batch > 1) {                                 // wave-uniform
            const unsigned long long gm = __ballot(glass_hit);
            const bool hold = gm != 0ull && (unsigned int)__builtin_popcountll(gm) < (unsigned int)it.glass_batch && __any((alive && !glass_hit) || has_ray);
            if (hold) { shade_now = alive && !glass_hit; probe.parked((unsigned int)__builtin_popcountll(gm)); }
        }
        const bool glass = shade_now && glass_hit;
        live_w += (unsigned int)__builtin_popcountll(__ballot(shade_now));    // one child per shaded hit ...
        probe.shade(shade_now, glass);
        // (the shade comes in three pieces -- [draws, mirror direction][the GLASS block with its expand][rotation and the child in the lane] -- so
        // that the second child's thirteen words live only inside the middle piece, not across the three sin / cos pairs of the last one)
        const float4 ma = M[2 * idx];
        const V3 color = mk(ma.x, ma.y, ma.z);
        V3 rv = mk(0.0f, 0.0f, 0.0f), reflection = rv;
        float ia = 0.0f, glass_R = 0.0f;
        if (shade_now) {
            rv.x = gen_component(seed); rv.y = gen_component(seed); rv.z = gen_component(seed);      // genVec (Util.hs:114-118)
            ia = dot(d, normal);
            reflection = d - scale_l(2.0f * ia, normal);
        }
        probe.stamp(3);
        // ---- the GLASS block and expand: the refraction child (extension; spec = the oracle's glass_children), then compaction of the emitted
        // children into the wave's ring; what the ring cannot hold goes to the wave's spill queue, what that cannot hold to the overflow stream
        const unsigned long long kids = it.may_emit ? __ballot(glass) : 0ull;
        if (kids) {                                           // wave-uniform
            const bool emits = glass;
            V3 ko = o, kd = o, kt = o; Sfc32 ks = seed;
            if (glass) {
                probe.glass_block(diag::lanes(true));
                glass_R = glass_refraction_child(color, glass_constants_of<LDS_SCENE>(mb), o, normal, d, ia, reflection, throughput, seed, ko, kd, kt, ks);
            }
            const unsigned int cnt = (unsigned int)__builtin_popcountll(kids), rank = rank_in(kids);
            live_w += cnt;                                    // ... and a second one per GLASS hit
            const unsigned int room_ring = kRing - ring_n;
            const unsigned int to_ring = cnt < room_ring ? cnt : room_ring;
            if (emits && rank < to_ring) {
                const unsigned int slot = (ring_head + ring_n + rank) & (kRing - 1u);
                ring[0][slot] = f2u(ko.x); ring[1][slot] = f2u(ko.y); ring[2][slot] = f2u(ko.z);
                ring[3][slot] = f2u(kd.x); ring[4][slot] = f2u(kd.y); ring[5][slot] = f2u(kd.z);
                ring[6][slot] = f2u(kt.x); ring[7][slot] = f2u(kt.y); ring[8][slot] = f2u(kt.z);
                ring[9][slot] = pixel;
                ring[10][slot] = ks.a; ring[11][slot] = ks.b; ring[12][slot] = ks.c; ring[13][slot] = ks.counter;
                ring[14][slot] = depth + 1u;                  // the child's step index (its parent's is raised in the last piece of the shade)
            }
            ring_n += to_ring;
            if (cnt > to_ring) {                              // the ring is full (rare)
                const unsigned int rest = cnt - to_ring;
                const unsigned int room_spill = kSpill - spill_n;
                const unsigned int to_spill = rest < room_spill ? rest : room_spill;
                if (emits && rank >= to_ring && rank - to_ring < to_spill)
                    queue_store(it.spill, w * kSpill + ((spill_head + spill_n + (rank - to_ring)) & (kSpill - 1u)), ko, kd, kt, pixel, ks, depth + 1u);
                spill_n += to_spill; spilled_w += to_spill;
                if (rest > to_spill) {                        // the wave's spill queue is full too: the overflow stream (a later launch reads it)
                    const unsigned int cnt2 = rest - to_spill, first2 = to_ring + to_spill;
                    const bool spills = emits && rank >= first2;
                    const unsigned int rank2 = rank - first2;  // (meaningful where `spills`)
                    const unsigned int room = blk_end - blk;
                    unsigned int slot = blk + rank2;
                    if (cnt2 > room) {                        // the block is full: one atomic reserves the next for the whole wave
                        unsigned int fresh = 0;
                        if (lane == 0) fresh = it.out_base + atomicAdd(it.out_count, kNextBlock);
                        fresh = (unsigned int)__builtin_amdgcn_readfirstlane((int)fresh);
                        if (rank2 >= room) slot = fresh + (rank2 - room);
                        blk = fresh + (cnt2 - room); blk_end = fresh + kNextBlock;
                    } else {
                        blk += cnt2;
                    }
                    const unsigned int lost = (unsigned int)__builtin_popcountll(__ballot(spills && slot >= it.out.capacity));
                    if (spills && slot < it.out.capacity) queue_store(it.out, slot, ko, kd, kt, pixel, ks, depth + 1u);   // depth: the child's step index
                    if (lane == 0) {                          // practically never: counted where it happens, not in registers carried round the loop
                        if (cnt2 > lost) {
                            atomicAdd(it.emitted + (size_t)(w & (unsigned int)(kLvEmitShards - 1)) * kCounterStride, cnt2 - lost);
                            atomicAdd(it.stats + kLvSpilled * kCounterStride, cnt2 - lost);
                        }
                        if (lost) atomicAdd(it.stats + kLvDropped * kCounterStride, lost);
                    }
                }
            }
        }
        probe.stamp(4);
        // ---- the rest of the shade: Matte: rotate (anglesToQuaternion $ pi *^ rv) iNormal | Glossy: rotate (anglesToQuaternion $ (1 - p) *^ rv) reflection
        if (shade_now) {
            const bool matte = f2u(mb.x) == 0u;
            const V3 axis = matte ? normal : reflection;
            const float hk = matte ? 0.5f * kPi : mb.w;
            const V3 rotated = rotate(quaternion_from_half_angles(hk * rv.x, hk * rv.y, hk * rv.z), axis);
            const float nd = dot(rotated, axis);
            const float brdf = matte ? mb.z * nd : __builtin_fmaxf(0.0f, nd);
            constexpr float next_ray_prob = 1.0f / (kPi * 2.0f);
            const float factor = glass ? glass_R : brdf * next_ray_prob;
            const V3 next = mk(glass ? reflection.x : rotated.x, glass ? reflection.y : rotated.y, glass ? reflection.z : rotated.z);   // (component selects: a select between two structs went through scratch memory)
            // computeResult for EVERY hit (Trace.hs:318-323), then the child in the lane (0 + e * t: an exact zero either way is skipped or changes nothing)
            add_colour(mk(0.0f, 0.0f, 0.0f) + (scale_r(color, ma.w) * throughput));
            o = o + scale_r(next, kEpsilon);
            d = next;
            throughput = throughput * scale_r(color, factor);
            ++depth; pending = false; has_ray = true;          // the child: next traceStep, same lane
        }
        probe.stamp(5);
        // ---- trace round: one traceStep (Trace.hs:272-294) for every lane that holds a ray
        cut_w += (unsigned int)__builtin_popcountll(__ballot(has_ray && depth >= step_cap));
        probe.trace(has_ray);
        if (has_ray) {
            if (depth >= step_cap) {                          // the safety cap (the reference has none): the ray exists, but is never traced
                has_ray = false;
            } else {
                deepest = depth + 1u > deepest ? depth + 1u : deepest;
                const HitSel h = check_hit<LDS_SCENE && kStagedWalk>(S, ns, np, o, d);
                has_ray = false;
                if (h.just) {
                    hit_record(S, ns, h.idx, o, d, h.t, o, normal);
                    idx = h.idx;
                    pending = true;
                }
            }
        }
        probe.stamp(6);
    }
    probe.flush(a.work_counter, cur.home);
    // what is left of this wave's last overflow block: holes
    if (it.may_emit) {
        const unsigned int end = blk_end < it.out.capacity ? blk_end : it.out.capacity;
        for (unsigned int i = blk + (unsigned int)lane; i < end; i += 64u) *it.out.pixel_word(i) = kHole;
    }
    // statistics: one set of atomics per wave, on counters sharded by workgroup
    unsigned int deep = deepest;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(deep, off, 64); deep = other > deep ? other : deep; }
    if (lane == 0) {
        unsigned int *st = it.stats;
        if (live_w) atomicAdd(st + (kLvLive + (w & (unsigned int)(kLvLiveShards - 1))) * kCounterStride, live_w);
        // a maximum: most waves find it already there (a plain load first; the atomic only when it would raise the word)
        if (deep > __hip_atomic_load(st + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + kLvDeepest * kCounterStride, deep);
        if (cut_w) atomicAdd(st + kLvCut * kCounterStride, cut_w);
        if (spilled_w) atomicAdd(st + kLvSpilled * kCounterStride, spilled_w);
    }
}

// ---------------------------------------------------------------------------------------
// streams_level_kernel: one overflow level.  Input: the stream the previous level (or the split kernel) wrote -- ray
// states and holes; persistent waves take its 64-record chunks in a static stride; a lane follows its ray's lineage (at a
// GLASS hit the reflection stays, the refraction goes to the output stream); lanes whose lineage has ended refill from
// the wave's chunk in batches.  Colours through float atomics.
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE>
__global__ void __launch_bounds__(kRenderBlock, PTMI_LEVEL_WAVES) streams_level_kernel(const RenderArgs a, const LevelArgs lv)
{
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
        stage_glass_constants(lds_scene, a.scene);
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned int G = gridDim.x, w = blockIdx.x;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;

    unsigned int n_in = *lv.in_count + lv.in_base;            // the producer's cursor (real items and holes)
    n_in = n_in < lv.in.capacity ? n_in : lv.in.capacity;
    const unsigned int n_chunks = (n_in + 63u) / 64u;
    unsigned int chunk = w, taken = 0;                        // wave-uniform cursor: chunk index, items of it already handed out
    unsigned int blk = w * kFirstBlock, blk_end = blk + kFirstBlock;   // wave-uniform: the output block being filled
    unsigned int chunk_len = 0, chunk_first = 0;
    auto open = [&]() __attribute__((always_inline)) {
        if (chunk >= n_chunks) { chunk_len = 0; return; }
        chunk_first = chunk * 64u;
        chunk_len = n_in - chunk_first < 64u ? n_in - chunk_first : 64u;
    };
    open();

    bool has_ray = false, pending = false;                    // a ray to trace / a hit to shade
    V3 o = mk(0, 0, 0), d = o, throughput = o, normal = o;
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    uint32_t pixel = 0, depth = 0;
    int idx = 0;
    unsigned int deepest = 0;
    unsigned int live_w = 0, cut_w = 0, dropped_w = 0, stored_w = 0;

    auto add_colour = [&](V3 c) __attribute__((always_inline)) {
        if (c.x != 0.0f) atomicAdd(a.planes.r + pixel, c.x);
        if (c.y != 0.0f) atomicAdd(a.planes.g + pixel, c.y);
        if (c.z != 0.0f) atomicAdd(a.planes.b + pixel, c.z);
    };

    for (;;) {
        if (pending && near_zero(throughput)) {
            const float4 ma = M[2 * idx];
            add_colour(scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
            pending = false;
        }
        const unsigned long long idle = __ballot(!has_ray && !pending);
        if (chunk < n_chunks && ((unsigned int)__builtin_popcountll(idle) >= kRefillBatch || (idle && !~idle))) {   // wave-uniform
            const unsigned int want = (unsigned int)__builtin_popcountll(idle), avail = chunk_len - taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = (unsigned int)__builtin_popcountll(idle & below);
            if (!has_ray && !pending && rank < take) {
                const float4 *r = lv.in.record(chunk_first + taken + rank);
                const float4 r2 = r[2];
                pixel = f2u(r2.y);
                if (pixel != kHole) {
                    const float4 r0 = r[0], r1 = r[1], r3 = r[3];
                    o = mk(r0.x, r0.y, r0.z);
                    d = mk(r0.w, r1.x, r1.y);
                    throughput = mk(r1.z, r1.w, r2.x);
                    seed.a = f2u(r2.z); seed.b = f2u(r2.w); seed.c = f2u(r3.x); seed.counter = f2u(r3.y);
                    depth = f2u(r3.z); has_ray = true;
                }
            }
            taken += take;
            if (taken >= chunk_len) { chunk += G; taken = 0; open(); }
        }
        if (!__any(has_ray || pending) && chunk >= n_chunks) break;     // (a chunk of holes: nothing below runs, the next trip looks at the next one)
        bool emits = false;
        V3 ko = o, kd = o, kt = o; Sfc32 ks = seed;
        const bool alive = pending && !near_zero(throughput);
        live_w += (unsigned int)__builtin_popcountll(__ballot(alive));
        if (alive) {
            const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
            V3 contribution;
            if (f2u(mb.x) == 2u) {
                contribution = scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput;
                V3 co[2], cd[2], ct[2]; Sfc32 cs[2];
                glass_children(mk(ma.x, ma.y, ma.z), glass_constants_of<LDS_SCENE>(mb), o, normal, d, throughput, seed, co, cd, ct, cs);
                o = co[0]; d = cd[0]; throughput = ct[0]; seed = cs[0];
                ko = co[1]; kd = cd[1]; kt = ct[1]; ks = cs[1];
                emits = true;
            } else {
                contribution = mk(0.0f, 0.0f, 0.0f);
                shade(M, idx, o, normal, o, d, throughput, contribution, seed);
            }
            add_colour(contribution);
            ++depth; pending = false; has_ray = true;
        }
        const unsigned long long kids = lv.may_emit ? __ballot(emits) : 0ull;
        if (kids) {
            const unsigned int cnt = (unsigned int)__builtin_popcountll(kids), rank = (unsigned int)__builtin_popcountll(kids & below);
            live_w += cnt;
            const unsigned int room = blk_end - blk;
            unsigned int slot = blk + rank;
            if (cnt > room) {
                unsigned int fresh = 0;
                if (lane == 0) fresh = lv.out_base + atomicAdd(lv.out_count, kNextBlock);
                fresh = (unsigned int)__builtin_amdgcn_readfirstlane((int)fresh);
                if (rank >= room) slot = fresh + (rank - room);
                blk = fresh + (cnt - room); blk_end = fresh + kNextBlock;
            } else {
                blk += cnt;
            }
            const unsigned int lost = (unsigned int)__builtin_popcountll(__ballot(emits && slot >= lv.out.capacity));
            stored_w += cnt - lost; dropped_w += lost;
            if (emits && slot < lv.out.capacity) queue_store(lv.out, slot, ko, kd, kt, pixel, ks, depth);
        }
        cut_w += (unsigned int)__builtin_popcountll(__ballot(has_ray && depth >= step_cap));
        if (has_ray) {
            if (depth >= step_cap) {
                has_ray = false;
            } else {
                deepest = depth + 1u > deepest ? depth + 1u : deepest;
                const HitSel h = check_hit<LDS_SCENE && kStagedWalk>(S, ns, np, o, d);
                has_ray = false;
                if (h.just) {
                    hit_record(S, ns, h.idx, o, d, h.t, o, normal);
                    idx = h.idx;
                    pending = true;
                }
            }
        }
    }
    if (lv.may_emit) {
        const unsigned int end = blk_end < lv.out.capacity ? blk_end : lv.out.capacity;
        for (unsigned int i = blk + (unsigned int)lane; i < end; i += 64u) *lv.out.pixel_word(i) = kHole;
    }
    unsigned int deep = deepest;
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(deep, off, 64); deep = other > deep ? other : deep; }
    if (lane == 0) {
        unsigned int *st = lv.stats;
        if (live_w) atomicAdd(st + (kLvLive + (w & (unsigned int)(kLvLiveShards - 1))) * kCounterStride, live_w);
        if (stored_w) atomicAdd(lv.emitted + (size_t)(w & (unsigned int)(kLvEmitShards - 1)) * kCounterStride, stored_w);
        if (deep > __hip_atomic_load(st + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st + kLvDeepest * kCounterStride, deep);
        if (cut_w) atomicAdd(st + kLvCut * kCounterStride, cut_w);
        if (dropped_w) atomicAdd(st + kLvDropped * kCounterStride, dropped_w);
    }
}

}  // namespace

hipError_t launch_streams_level(const RenderArgs &a, const LevelArgs &lv, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    if (lds > kMaxSceneLds) return launch(streams_level_kernel<false>, g, b, 0, stream, a, lv);      // a scene too big for LDS at this occupancy: scalar loads
    else                    return launch(streams_level_kernel<true>, g, b, lds, stream, a, lv);
}

hipError_t launch_streams_split(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    const bool tiles = tiles_pay(a);
    if (lds > kMaxSceneLds) {
        if (tiles) return launch(streams_split_kernel<false, true>, g, b, 0, stream, a, it);
        else       return launch(streams_split_kernel<false, false>, g, b, 0, stream, a, it);
    } else {
        if (tiles) return launch(streams_split_kernel<true, true>, g, b, lds, stream, a, it);
        else       return launch(streams_split_kernel<true, false>, g, b, lds, stream, a, it);
    }
}

int streams_split_waves() { return PTMI_SPLIT_WAVES; }
unsigned int streams_spill_records() { return kSpill; }
unsigned int streams_first_block() { return kFirstBlock; }

}  // namespace ptmi
