// ptmi_stream_pixels.hip -- the stream form of render Streams for scenes whose rays never split: persistent waves, items by
// ticket, lanes that refill by ballot + prefix, optional ordered passes handed from lane to lane (one release per region and pass).
#include "ptmi_stream_form.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// streams_pixels_kernel: the stream form for scenes whose rays never split.  The loop is render_streams_kernel's
// [finish dead rays][next sample][shade][trace] with a [refill] block in front: a lane whose pixel is done stores its
// seven words and becomes idle; idle lanes take the next start hits of the wave's chunk.
// ---------------------------------------------------------------------------------------
#ifndef PTMI_PIXELS_WAVES
#define PTMI_PIXELS_WAVES 7
#endif
constexpr int kMinPassSamples = 1;                           // the fewest samples an ordered pass may hold (a lane publishes an item before it takes the next)
constexpr unsigned int kChunkSlots = 64;                     // ordered passes, fenced hand-off: chunks a wave may have in flight (one counter each in LDS)
template <bool LDS_SCENE, bool PASSES>
__global__ void __launch_bounds__(kRenderBlock, PTMI_PIXELS_WAVES) streams_pixels_kernel(const RenderArgs a, const ItemArgs it)
{
    // the lane's item: 0-2 position of the start hit, 3-5 axis and 6 half-angle scale of its bounce, 7 primitive, 8 quad,
    // 9 the lane's count of shaded hits when the item began, 10 the samples the item renders, 11 its region
    __shared__ float item_const[12][kRenderBlock];
    __shared__ unsigned int chunk_left[kChunkSlots], chunk_total[kChunkSlots];      // (PASSES only: the other instantiations never touch them)
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    // ORDERED PASSES (it.passes > 1).  A pixel's samples are a serial chain, and with few items per lane the end of the launch is as
    // long as the last items.  The samples are therefore cut into passes: an item renders one pass's samples of its pixel, and the
    // pixel's seven words travel through the planes to whichever lane -- of any wave, on any XCD -- takes its next pass.  What orders
    // them: an item of pass p is handed out only when region_done[its region] says that every item of the region's pass p - 1 has
    // been PUBLISHED.  The L2s of the eight XCDs are not coherent with each other and a CU's L1 is never refreshed.  Two hand-offs:
    //   * FENCED (it.fenced, the default since round 6): the architecturally promised form -- stores, the storing wave's vmcnt(0), an
    //     agent-scope RELEASE, the counter; on the taking side the poll, then an agent-scope ACQUIRE, then the loads.  A release is a
    //     chip resource (a write-back of the XCD's L2: 7 168 waves releasing every 16 trips DOUBLED a launch in round 3), so it is paid
    //     once per (region, pass), not per item: every item of a chunk is taken by the ONE wave that drew its ticket, so that wave
    //     knows when the chunk's last item ends -- a counter per chunk in flight in LDS, decremented by the lane that ends an item -- and
    //     that lane alone releases and adds the chunk's whole count.  130 000 releases per launch of one of 8 parts of a 4K image at
    //     1024 spp instead of 8.3 million: 3.6 per microsecond on the chip.  (The words are still stored write-through, so the release
    //     finds the L2 clean of them.)
    //   * FENCE-FREE (PTMI_OPT_PASS_HANDOFF = 1, rounds 3-5's form): the seven words are stored WRITE-THROUGH (sc1), the storing wave
    //     waits for its stores (vmcnt(0)) before its lanes add to the region's counter, and the taking lanes read counter and words with
    //     sc1 loads (MI355X_MICROARCH.md, "Valid forms").  Measured valid on gfx950 over 5 billion hand-offs, not promised by the
    //     memory model: the caller's explicit choice only.
    // Nothing here depends on which XCD or CU a wave runs on.
    // (PASSES is a template parameter: carried as run-time branches the blocks below cost the one-pass kernel 3.8 % -- 4.57 -> 4.75 ms on
    // S16 -- in scalar registers spilled and instructions per trip)
    const int passes = PASSES ? it.passes : 1;
    const bool fenced = PASSES && it.fenced != 0;            // wave-uniform (a kernel argument)
    if (PASSES) { if (threadIdx.x < (int)kChunkSlots) chunk_left[threadIdx.x] = 0u; }      // (one wave per workgroup: LDS operations of a wave are in order)
    unsigned int chunk_serial = 0, cur_slot = 0;
    float *mine = &item_const[0][threadIdx.x];
    auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
    auto get = [&](int k) { return mine[k * kRenderBlock]; };
    // PTMI_SEED_FROM_RESULT (combine new old): the seed the ray of the sample's last hit carried.  (Four registers, copied at every
    // hit; four LDS words written at every hit and four selects per sample cost 3 % more, stepping the lane's seed back over the
    // hit's draws at the sample's end -- sfc32_prev -- 2.5 %.)
    Sfc32 hit_seed; hit_seed.a = hit_seed.b = hit_seed.c = hit_seed.counter = 0;

    ChunkCursor cur; cur.home = xcc_id(); cur.tries = 0;
    // (the positions from *tail_start on -- the cheapest quads -- are the per-pixel kernel's, whose waves fill the slots this launch's
    // waves leave as they end: ptmi_api.cpp)
    cur.n_positions = it.n_positions;
    if (it.tail_start) { const unsigned int t = *it.tail_start; cur.n_positions = t < it.n_positions ? t : it.n_positions; }
    // The wave's next chunk.  Fenced hand-off: the chunk gets a slot for its count of unfinished items; if the slot's previous tenant -- the
    // chunk drawn kChunkSlots chunks ago -- still has an item running in some lane (one very long item), no ticket is drawn now and the
    // refill tries again next trip: the busy lanes go on, their items end, the slot frees (no lane busy = no slot taken: the loop's exit
    // condition is never starved).
    auto take_chunk = [&]() {
        if (fenced) {
            const unsigned int slot = (chunk_serial + 1u) & (kChunkSlots - 1u);
            const unsigned int tenant = (unsigned int)__builtin_amdgcn_readfirstlane((int)chunk_left[slot]);
            if (tenant != 0u) { cur.taken = 0; cur.len = 0; return; }
            next_chunk<false>(cur, it);
            if (cur.len) {
                ++chunk_serial; cur_slot = slot;
                if (lane == 0) { chunk_left[slot] = cur.len; chunk_total[slot] = cur.len; }
            }
        } else {
            next_chunk<false>(cur, it);
        }
    };
    take_chunk();
    diag::TailProbe probe; probe.begin();                     // (diagnostic builds: ptmi_diag.h)

    bool busy = false, pending = false, has_ray = false, over = false, unpublished = false;
    V3 acc = mk(0.0f, 0.0f, 0.0f), pos = acc, normal = acc, d = acc, throughput = acc;
    Sfc32 pixel_seed; pixel_seed.a = pixel_seed.b = pixel_seed.c = pixel_seed.counter = 0;
    Sfc32 seed = pixel_seed;
    int idx = 0, s = 0;
    uint32_t pixel4 = 0;                                      // byte offset of the lane's pixel in a plane
    unsigned int steps = 0, longest = 0, live = 0;
    for (;;) {
        // ---- publish (ordered passes): the items whose words this wave stored in its last trip (a trip ago: the wait is free)
        if (PASSES && __any(unpublished)) {                   // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores have completed before the counters move
            if (fenced) {                                     // a chunk of this wave is complete: ONE release for all its items' words
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler may drop the wait behind buffer_wbl2 when it believes the scoreboard empty)
                if (unpublished) { const unsigned int at = f2u(get(11)); atomicAdd(it.region_done + (at & 0x3ffffffu), chunk_total[at >> 26]); unpublished = false; }
            } else if (unpublished) { atomicAdd(it.region_done + (f2u(get(11)) & 0x3ffffffu), 1u); unpublished = false; }
        }
        // ---- refill: idle lanes take the next start hits of the wave's chunk (at once: an item is a pixel's whole sample chain)
        const unsigned long long idle = __ballot(!busy);
        bool open = idle && chunks_left(cur);
        if (PASSES && open && cur.pass > 0u && !cur.ready) {           // wave-uniform: has the region's previous pass been published?
            unsigned int done = 0;
            if (lane == 0) done = __hip_atomic_load(it.region_done + cur.region, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = (unsigned int)__builtin_amdgcn_readfirstlane((int)done);
            // (a compiler barrier, no instruction: the sc1 loads of the region's planes below are relaxed atomics to other addresses, and nothing
            // but this keeps a future compiler from hoisting them above the poll they depend on -- the mirror of the storing side's asm memory
            // clobber.  The hardware orders them: same wave, loads issued after the poll's value has returned.)
            asm volatile("" ::: "memory");
            if (done >= cur.pass * cur.len) {
                cur.ready = true;
                if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE poll, ONE acquire, then this wave's loads of the region's words
            } else { open = false; if (!__any(busy)) __builtin_amdgcn_s_sleep(8); }
        }
        if (open) {                                           // wave-uniform
            probe.refill_begin();
            const unsigned int want = (unsigned int)__builtin_popcountll(idle), avail = cur.len - cur.taken;
            const unsigned int take = want < avail ? want : avail;
            const unsigned int rank = rank_in(idle);
            if (!busy && rank < take) {
                const float4 *r = it.hits.record(cur.first + cur.taken + rank);
                const float4 r0 = r[0], r1 = r[1], r3 = r[3];   // [position, axis x] [axis yz, half-angle scale, -] ... [primitive, pixel, -, quad]
                pixel4 = f2u(r3.y) << 2;
                put(0, r0.x); put(1, r0.y); put(2, r0.z);
                put(3, r0.w); put(4, r1.x); put(5, r1.y); put(6, r1.z);
                put(7, r3.x); put(8, r3.w); put(9, u2f(live));
                // the samples of this pass: n_spp over the passes, the first (n_spp mod passes) passes one more
                put(10, u2f((uint32_t)(a.n_spp / passes + ((int)cur.pass < a.n_spp % passes ? 1 : 0)))); put(11, u2f(cur.region | (cur_slot << 26)));   // (regions: < 2^26, checked by the host)
                if (PASSES) {                                  // another wave's stores of a moment ago: sc1 loads
                    acc = mk(load_agent(&plane_at(a.planes.r, pixel4)), load_agent(&plane_at(a.planes.g, pixel4)), load_agent(&plane_at(a.planes.b, pixel4)));
                    pixel_seed.a = load_agent(&plane_at(a.planes.sa, pixel4)); pixel_seed.b = load_agent(&plane_at(a.planes.sb, pixel4));
                    pixel_seed.c = load_agent(&plane_at(a.planes.sc, pixel4)); pixel_seed.counter = load_agent(&plane_at(a.planes.sctr, pixel4));
                } else {
                    acc = mk(plane_at(a.planes.r, pixel4), plane_at(a.planes.g, pixel4), plane_at(a.planes.b, pixel4));
                    pixel_seed.a = plane_at(a.planes.sa, pixel4); pixel_seed.b = plane_at(a.planes.sb, pixel4);
                    pixel_seed.c = plane_at(a.planes.sc, pixel4); pixel_seed.counter = plane_at(a.planes.sctr, pixel4);
                }
                s = -1; busy = true; over = true; pending = false; has_ray = false;
            }
            cur.taken += take;
            if (cur.taken >= cur.len) take_chunk();
            probe.refill_end(take);
        }
        if (!__any(busy) && !chunks_left(cur)) break;      // (no lane busy, chunks left: nothing below has a lane to run for; the next trip refills)
        probe.trip(busy);
        float4 mb = M[2 * idx + 1];
        V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
        if (pending) {
            // A ray whose throughput is already near zero dies at this hit (numNewRays, Trace.hs:329-331): the hit still
            // adds its emittance (computeResult runs for every intersection) and nothing else of it survives.
            if (near_zero(throughput)) {
                const float4 ma = M[2 * idx];
                acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
                hit_seed = seed;
                ++steps;
                pending = false; over = true;
            } else {
                bounce_axis(mb, normal, d, axis, hk);
            }
        }
        probe.ending_begin(over, s, mine + 10 * kRenderBlock);
        if (over) {
            if (s >= 0) {                                      // a sample has been rendered
                if (a.seed_from_result && steps > 0u) {        // combine new old: the seed the sample's last hit carried
                    pixel_seed = hit_seed;
                }
                (void)random_float(pixel_seed);                // updateSeed
            }
            ++s; steps = 0;
            over = false;
            if (s < (int)f2u(get(10))) {                       // the pixel's next sample of this pass, from its cached start hit
                longest = longest > 1u ? longest : 1u;         // the primary ray's traceStep
                seed = pixel_seed;
                throughput = mk(1.0f, 1.0f, 1.0f);
                pos = mk(get(0), get(1), get(2)); idx = (int)f2u(get(7));
                mb = M[2 * idx + 1];
                axis = mk(get(3), get(4), get(5)); hk = get(6);
                pending = true;
            } else {                                           // the pixel is done: its seven words, once
                if (PASSES) {                                  // write-through: the pixel's next pass may run behind another L2
                    store_agent(&plane_at(a.planes.r, pixel4), acc.x); store_agent(&plane_at(a.planes.g, pixel4), acc.y); store_agent(&plane_at(a.planes.b, pixel4), acc.z);
                    store_agent(&plane_at(a.planes.sa, pixel4), pixel_seed.a); store_agent(&plane_at(a.planes.sb, pixel4), pixel_seed.b);
                    store_agent(&plane_at(a.planes.sc, pixel4), pixel_seed.c); store_agent(&plane_at(a.planes.sctr, pixel4), pixel_seed.counter);
                    // published at the top of the next trip: every item (fence-free), or the chunk by the lane that ends its LAST item (fenced)
                    unpublished = fenced ? atomicSub(&chunk_left[f2u(get(11)) >> 26], 1u) == 1u : true;
                } else {
                    plane_at(a.planes.r, pixel4) = acc.x; plane_at(a.planes.g, pixel4) = acc.y; plane_at(a.planes.b, pixel4) = acc.z;
                    plane_at(a.planes.sa, pixel4) = pixel_seed.a; plane_at(a.planes.sb, pixel4) = pixel_seed.b;
                    plane_at(a.planes.sc, pixel4) = pixel_seed.c; plane_at(a.planes.sctr, pixel4) = pixel_seed.counter;
                }
                record_item_cost(a, f2u(get(8)), live - f2u(get(9)));       // its shaded hits stand for the loop trips it took
                busy = false;
            }
        }
        probe.ending_end();
        if (pending) {                                         // alive (a fresh sample starts with throughput 1)
            const bool capped = steps + 1u >= step_cap;
            hit_seed = seed;
            // results: colour += emittance * throughput for EVERY hit; then the new ray (shade, with the axis in hand)
            V3 next; float brdf;
            next_about_axis(mb, axis, hk, seed, next, brdf);
            apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, acc);
            ++steps; ++live;                                   // the child exists even if the cap then cuts it
            pending = false;
            if (capped) {                                      // rare: only when the safety cap bites
                over = true;
                const unsigned long long cm = __ballot(1);
                if (lane == (int)__builtin_ctzll(cm)) atomicAdd(a.stream_counters + kScTruncated, (unsigned long long)__builtin_popcountll(cm));
            } else has_ray = true;
        }
        if (has_ray) {
            longest = steps + 1u > longest ? steps + 1u : longest;     // the traceStep this ray belongs to
            const HitSel h = check_hit<LDS_SCENE && kStagedWalk>(S, ns, np, pos, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                idx = h.idx;
                pending = true;
            } else {
                over = true;
            }
        }
    }
    if (PASSES && __any(unpublished)) {                      // (nobody waits for the last pass; a wave that ends earlier owes its items)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (unpublished) { const unsigned int at = f2u(get(11)); atomicAdd(it.region_done + (at & 0x3ffffffu), fenced ? chunk_total[at >> 26] : 1u); }
    }
    probe.flush(a.work_counter);
    // statistics: the per-pixel kernels' sharded counters
    for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
    const unsigned long long live_total = wave_sum(live);
    if (lane == 0) {
        if (a.live_counter && live_total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, live_total);
        if (a.stream_iterations && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
}

}  // namespace

hipError_t launch_streams_pixels(const RenderArgs &a, const ItemArgs &it, unsigned int grid, hipStream_t stream)
{
    if (grid == 0) return hipSuccess;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 g(grid), b(kRenderBlock);
    if (it.passes > 1) {
        if (lds > kMaxSceneLds) return launch(streams_pixels_kernel<false, true>, g, b, 0, stream, a, it);
        else                    return launch(streams_pixels_kernel<true, true>, g, b, lds, stream, a, it);
    } else {
        if (lds > kMaxSceneLds) return launch(streams_pixels_kernel<false, false>, g, b, 0, stream, a, it);
        else                    return launch(streams_pixels_kernel<true, false>, g, b, lds, stream, a, it);
    }
}

int streams_pixels_waves() { return PTMI_PIXELS_WAVES; }
int streams_min_pass_samples() { return kMinPassSamples; }

}  // namespace ptmi
