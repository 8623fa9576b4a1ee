// ptmi_inline_ablations.hip -- the ablation forms of render Inline (ptmi_set_variant 1-3, 6, 10-12, 18): round 1's
// loop, regenerate-only, lock step, the pooled second shade round, the persistent hand-out.  Same per-pixel arithmetic as
// render_inline_kernel (ptmi_inline.hip), bit-identical planes (tests/test_gpu_render_parity.py); they exist to be MEASURED
// against it and are compiled only with -DPTMI_ABLATIONS into libptmi_ablations.so (tests, tools/measure_extra.py).
#ifdef PTMI_ABLATIONS
#include "ptmi_device.h"

namespace ptmi {

namespace {

#ifndef PTMI_FETCH_BATCH
#define PTMI_FETCH_BATCH 1
#endif
constexpr int kFetchBatch = PTMI_FETCH_BATCH;   // lanes that must be idle before the wave fetches pixels (persistent kernel)
constexpr int kChunk = 64;      // pixels a wave takes from the global counter per atomic (persistent kernel)

// ---------------------------------------------------------------------------------------
// render Inline, other loop shapes (MODE):
//   kCachedR1    round 1's default: primary hit cached, two shade rounds per trace round, every shade in full (no frozen-shade shortcut)
//   kRegenerate  lanes start their next sample as soon as a path ends, one shade per trace, no cached primary hit
//   kLockstep    all lanes of the wave run sample s together (what a per-sample launch would do)
// ---------------------------------------------------------------------------------------
enum { kRegenerate = 1, kLockstep = 2, kCachedR1 = 3 };
template <bool LDS_SCENE, int MODE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, MODE == kCachedR1 ? 6 : 4) render_inline_modes_kernel(const RenderArgs a)
{
    __shared__ float pixel_const[MODE == kCachedR1 ? 19 : 1][kRenderBlock];   // round 1's loop: per-lane restart record, 19 words
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    // SAMPLE CHUNKS (a.spp_chunks > 1; tiled kernels only).  A pixel's samples are one serial chain, so a launch has as
    // many waves as the image has tiles, each as long as n_spp; with few tiles and many samples -- one of 8 parts of a
    // 4K image at 1024 spp: 16 200 waves for 6 144 slots -- the last round of waves runs on a partly empty chip and
    // costs 15 %.  The grid is therefore spp_chunks copies of the tile grid: copy c of a tile renders samples
    // [c S, (c+1) S) of its pixels, after copy c-1 has stored the planes and published done[tile] = c.  A workgroup's place
    // in that chain is a ticket it draws when it starts (enter_sample_chunk), so the producer of what it waits for has
    // started before it, whatever order the hardware dispatches workgroups in (in practice the producer finished a whole
    // round earlier: the wait falls through).  The planes travel through
    // memory between copies: release / acquire at agent scope (L2 write-back, L1 invalidate); copies of one tile run on
    // the same XCD (the grid of a copy is a multiple of 32).  Results do not depend on the chunking (sample-split invariance).
    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }

        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);

        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        Sfc32 seed;
        seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
        seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];

        const int limit = a.bounce_limit, n_spp = n_spp_chunk;

        if (limit <= 0) {
            // iterate 0: every sample returns (0, seed); new + old
            if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;
        } else if (MODE == kCachedR1) {
            // Round 1's default: the primary hit is evaluated once per pixel and every sample starts from that record; a sample
            // costs k shades and k-1 traces (k = its live bounces), every shade in full (no frozen-shade shortcut).
            // Loop shape [shade][shade again for lanes whose sample just ended][trace].
            const HitSel h0 = check_hit(S, ns, np, origin, primary);
            if (!h0.just) {
                if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;     // every sample: result 0, seed untouched
            } else {
                // The per-pixel constants a sample restarts from (primary hit record + primary direction, 10
                // words) are read once per sample: they live in a lane-private LDS column instead of VGPRs,
                // which is what lets the kernel fit 72 VGPRs = 7 waves per SIMD (16 words: 28 waves fit the LDS of a CU).
                float *mine = &pixel_const[0][threadIdx.x];
                auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
                auto get = [&](int k) { return mine[k * kRenderBlock]; };
                V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
                hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
                put(0, pos.x); put(1, pos.y); put(2, pos.z);
                put(3, normal.x); put(4, normal.y); put(5, normal.z);
                put(6, primary.x); put(7, primary.y); put(8, primary.z);
                put(9, acc.x); put(10, acc.y); put(11, acc.z);        // the accumulator is touched once per sample: LDS too
                const int idx0 = h0.idx;
                {   // what every first shade of this pixel uses (shade_first)
                    const float4 ma0 = M[2 * idx0], mb0 = M[2 * idx0 + 1];
                    V3 axis; float hk;
                    bounce_axis(mb0, normal, primary, axis, hk);
                    const V3 first_result = mk(0.0f, 0.0f, 0.0f) + (scale_r(mk(ma0.x, ma0.y, ma0.z), ma0.w) * mk(1.0f, 1.0f, 1.0f));
                    put(12, axis.x); put(13, axis.y); put(14, axis.z); put(15, hk);
                    if (MODE == kCachedR1) { put(16, first_result.x); put(17, first_result.y); put(18, first_result.z); }
                }
                int s = 0, it = 0, idx = idx0;
                V3 d = primary;                                       // the ray that produced the hit / the next ray
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                bool pending = n_spp > 0, has_ray = false;
                auto restart = [&]() {                                // next sample of this pixel
                    // \(new, seed') (old, _) -> (new + old, seed')
                    put(9, result.x + get(9)); put(10, result.y + get(10)); put(11, result.z + get(11));
                    ++s; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    pos = mk(get(0), get(1), get(2)); normal = mk(get(3), get(4), get(5));
                    d = mk(get(6), get(7), get(8)); idx = idx0;
                    pending = s < n_spp;
                };
                diag::PhaseProbe phase;
                while (pending) {
                    ++trips;
                    phase.trip();
                    // round A: whatever hit is pending
                    if (pending && !has_ray) {
                        phase.round_a(true);
                        float4 mb = M[2 * idx + 1];
                        V3 axis; float hk;
                        bounce_axis(mb, normal, d, axis, hk);
                        V3 next; float brdf;
                        next_about_axis(mb, axis, hk, seed, next, brdf);
                        apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
                        ++it; ++live;
                        // the next prepareRay would freeze the path (Trace.hs:364-365)
                        if (it >= limit || near_zero(throughput)) restart();
                        else { pending = false; has_ray = true; }
                    }
                    phase.end_a();
                    // round B: only lanes that restarted in round A get here with a pending hit, and that hit is the cached primary hit
                    // with result 0 and throughput 1: the specialised first shade
                    if (pending && !has_ray) {
                        phase.round_b(true);
                        shade_first<false>(M, idx0, pos, mk(get(12), get(13), get(14)), get(15), mk(get(16), get(17), get(18)),
                                    pos, d, throughput, result, seed);
                        ++it; ++live;
                        if (it >= limit || near_zero(throughput)) restart();
                        else { pending = false; has_ray = true; }
                    }
                    phase.end_b(); phase.round_c(has_ray);
                    if (has_ray) {
                        const HitSel h = check_hit(S, ns, np, pos, d, diag::sphere_counters(a.work_counter));
                        has_ray = false;
                        if (h.just) {
                            hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                            idx = h.idx;
                            pending = true;
                        } else {
                            restart();
                        }
                    }
                    phase.end_c();
                }
                acc = mk(get(9), get(10), get(11));
                phase.flush(a.work_counter);
            }
        } else if (MODE == kRegenerate) {
            int s = 0, it = 0;
            V3 o = origin, d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
            while (s < n_spp) {
                const HitSel h = check_hit(S, ns, np, o, d);
                bool end = true;
                if (h.just) {
                    V3 hit_pos, normal;
                    hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                    shade(M, h.idx, hit_pos, normal, o, d, throughput, result, seed);
                    ++it; ++live;
                    // the next prepareRay would freeze the path (Trace.hs:364-365)
                    end = (it >= limit) || near_zero(throughput);
                }
                if (end) {
                    acc = result + acc;                      // \(new, seed') (old, _) -> (new + old, seed')
                    ++s; it = 0;
                    o = origin; d = primary;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                }
            }
        } else {
            for (int s = 0; s < n_spp; ++s) {
                V3 o = origin, d = primary;
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                for (int it = 0; it < limit; ++it) {
                    if (near_zero(throughput)) break;
                    const HitSel h = check_hit(S, ns, np, o, d);
                    if (!h.just) break;
                    V3 hit_pos, normal;
                    hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                    shade(M, h.idx, hit_pos, normal, o, d, throughput, result, seed);
                    ++live;
                }
                acc = result + acc;
            }
        }

        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
        a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
    }
    leave_sample_chunk<TILE_W>(a, wg, chunk);

    if (TILE_W > 0) record_cost(a, quad, trips);
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
}

// ---------------------------------------------------------------------------------------
// render Inline, pooled second shade round.  Same loop [shade A][shade B][trace C] and the same arithmetic as
// render_inline_kernel, but the B round -- lanes whose sample ended in A and whose next sample starts from the cached
// primary hit; only ~49 % of a wave's lanes -- is shared by the W waves of a workgroup: a restarting lane
// posts (seed, owner) into an LDS pool (ballot + prefix inside the wave, one LDS atomic per wave for the base),
// the pool's items are shaded densely by as many waves as it takes (the others skip the round), and the
// owner picks up (seed', next, b) and finishes computeRay itself.  An item's inputs besides the seed are the
// owner's restart record, which already lives in LDS.  A pixel's arithmetic does not depend on which lane
// executes it, so the planes stay bit-identical.
// Requires bounce_limit >= 1 and n_spp >= 1 (the launcher routes the degenerate cases elsewhere).
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE, int W>
__global__ void __launch_bounds__(64 * W, 6) render_inline_pooled_kernel(const RenderArgs a)
{
    constexpr int kThreads = 64 * W;
    __shared__ float pixel_const[13][kThreads];             // restart record: hit pos, normal, primary dir, acc, primitive index
    __shared__ unsigned int pool[8][kThreads];              // in: seed[4], owner | out: seed'[4], b, next[3]
    __shared__ unsigned int pool_count[2];                  // items | (waves with work << 16), alternating per trip
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) pool_count[tid] = 0;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = tid; i < total; i += kThreads) lds_scene[i] = a.scene.packed[i];
    }
    __syncthreads();
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    const long long n_local = (long long)a.rows_local * a.width;
    const long long pixel = (long long)blockIdx.x * kThreads + tid;
    const bool valid = pixel < n_local;
    const int limit = a.bounce_limit, n_spp = a.n_spp;
    unsigned int live = 0;

    float *mine = &pixel_const[0][tid];
    auto put = [&](int k, float v) { mine[k * kThreads] = v; };
    auto get = [&](int k) { return mine[k * kThreads]; };

    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    V3 pos = mk(0, 0, 0), normal = pos, d = pos;
    V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
    int s = 0, it = 0, idx = 0, idx0 = 0;
    bool pending = false, has_ray = false;

    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }
        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);
        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
        seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];
        const HitSel h0 = check_hit(S, ns, np, origin, primary);
        if (!h0.just) {
            acc = mk(0.0f, 0.0f, 0.0f) + acc;                // every sample: result 0, seed untouched
        } else {
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            put(0, pos.x); put(1, pos.y); put(2, pos.z);
            put(3, normal.x); put(4, normal.y); put(5, normal.z);
            put(6, primary.x); put(7, primary.y); put(8, primary.z);
            put(12, u2f((uint32_t)h0.idx));
            idx0 = idx = h0.idx;
            d = primary;
            pending = true;
        }
        put(9, acc.x); put(10, acc.y); put(11, acc.z);
    }
    auto restart = [&]() {                                    // next sample of this pixel
        // \(new, seed') (old, _) -> (new + old, seed')
        put(9, result.x + get(9)); put(10, result.y + get(10)); put(11, result.z + get(11));
        ++s; it = 0;
        throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
        pos = mk(get(0), get(1), get(2)); normal = mk(get(3), get(4), get(5));
        d = mk(get(6), get(7), get(8)); idx = idx0;
        pending = s < n_spp;
    };

#ifdef PTMI_POOL_STATS
    unsigned int st_trips = 0, st_alive = 0, st_batches = 0, st_items = 0;
    unsigned long long cyc[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int k) { const unsigned long long t = __builtin_amdgcn_s_memtime(); cyc[k] += t - t_prev; t_prev = t; };
#else
    auto stamp = [](int) {};
#endif
    for (unsigned int trip = 0;; ++trip) {
        // ---- round A: in place
        if (pending && !has_ray) {
            shade(M, idx, pos, normal, pos, d, throughput, result, seed);
            ++it; ++live;
            // the next prepareRay would freeze the path (Trace.hs:364-365)
            if (it >= limit || near_zero(throughput)) restart();
            else { pending = false; has_ray = true; }
        }
        stamp(0);
        // ---- round B: restarted lanes post their item
        const bool need_b = pending && !has_ray;
        const unsigned long long mask = __ballot(need_b);
        const bool wave_alive = __any(pending || has_ray);
        unsigned int base = 0;
        if (lane == 0 && wave_alive)
            base = atomicAdd(&pool_count[trip & 1], (unsigned int)__builtin_popcountll(mask) | (1u << 16));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base) & 0xffffu;
        const int slot = (int)base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (need_b) {
            pool[0][slot] = seed.a; pool[1][slot] = seed.b; pool[2][slot] = seed.c; pool[3][slot] = seed.counter;
            pool[4][slot] = (unsigned int)tid;
        }
        __syncthreads();
        stamp(1);
        const unsigned int count = pool_count[trip & 1];
        if ((count >> 16) == 0) break;                       // no wave of the workgroup has work left
        if (tid == 0) pool_count[(trip & 1) ^ 1] = 0;        // last read before this barrier, next written after the one below
        const int n_items = (int)(count & 0xffffu);
        // the waves take 64-item batches in an order that rotates with the trip, so the extra work moves over the SIMDs
        for (int b0 = (int)((wave + trip) % W) * 64; b0 < n_items; b0 += kThreads) {
            const int item = b0 + lane;
#ifdef PTMI_POOL_STATS
            ++st_batches; st_items += (unsigned int)__builtin_popcountll(__ballot(item < n_items));
#endif
            if (item < n_items) {
                Sfc32 sd; sd.a = pool[0][item]; sd.b = pool[1][item]; sd.c = pool[2][item]; sd.counter = pool[3][item];
                const float *rec = &pixel_const[0][pool[4][item]];
                const V3 n_i = mk(rec[3 * kThreads], rec[4 * kThreads], rec[5 * kThreads]);
                const V3 d_i = mk(rec[6 * kThreads], rec[7 * kThreads], rec[8 * kThreads]);
                const int idx_i = (int)f2u(rec[12 * kThreads]);
                V3 next; float brdf;
                next_direction(M, idx_i, n_i, d_i, sd, next, brdf);
                pool[0][item] = sd.a; pool[1][item] = sd.b; pool[2][item] = sd.c; pool[3][item] = sd.counter;
                pool[4][item] = f2u(brdf);
                pool[5][item] = f2u(next.x); pool[6][item] = f2u(next.y); pool[7][item] = f2u(next.z);
            }
        }
        stamp(2);
        __syncthreads();
        if (need_b) {
            seed.a = pool[0][slot]; seed.b = pool[1][slot]; seed.c = pool[2][slot]; seed.counter = pool[3][slot];
            const float brdf = u2f(pool[4][slot]);
            const V3 next = mk(u2f(pool[5][slot]), u2f(pool[6][slot]), u2f(pool[7][slot]));
            apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
            ++it; ++live;
            if (it >= limit || near_zero(throughput)) restart();
            else { pending = false; has_ray = true; }
        }
        stamp(3);
#ifdef PTMI_POOL_STATS
        ++st_trips; st_alive += wave_alive ? 1u : 0u;
#endif
        // ---- round C: trace
        if (has_ray) {
            const HitSel h = check_hit(S, ns, np, pos, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                idx = h.idx;
                pending = true;
            } else {
                restart();
            }
        }
        stamp(4);
    }

    if (valid) {
        a.planes.r[pixel] = get(9); a.planes.g[pixel] = get(10); a.planes.b[pixel] = get(11);
        a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
        a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
    }
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if (lane == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
#ifdef PTMI_POOL_STATS
    // diagnostic build only: per wave [1] trips, [2] trips with own work, [3] B batches executed, [4] B items executed,
    // [8..17] cycles in: round A | post + barrier | B batches | barrier + pick-up | trace
    if (lane == 0) {
        atomicAdd(a.work_counter + 1, st_trips); atomicAdd(a.work_counter + 2, st_alive);
        atomicAdd(a.work_counter + 3, st_batches); atomicAdd(a.work_counter + 4, st_items);
        for (int k = 0; k < 5; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(a.work_counter + 8 + 2 * k), cyc[k]);
    }
#endif
}

// ---------------------------------------------------------------------------------------
// render Inline, persistent form (default).  Same per-pixel arithmetic as render_inline_kernel, but a lane
// that finishes its pixel (all n_spp samples) takes the next unprocessed pixel from a global counter
// instead of idling until the slowest of the wave's 64 pixels is done: the number of trace rounds a
// pixel needs is a sum over its samples and varies by +-25 % inside a wave at 64 spp (measured:
// 47.5 of 64 lanes active per VALU instruction with the static mapping).  Pixels are handed out with
// one atomic per wave: ballot of the lanes that want one, popcount prefix for the rank, the lowest
// wanting lane adds the count -- so lanes that ask together get consecutive pixels (at start: 64
// consecutive pixels per wave, fully coalesced plane reads).  A fetched pixel's primary ray joins the
// wave's next trace round; its hit is cached for the pixel's remaining samples.
// Requires bounce_limit >= 1 and n_spp >= 1 (the launcher routes the degenerate cases elsewhere).
// ---------------------------------------------------------------------------------------
template <bool LDS_SCENE>
__global__ void __launch_bounds__(kRenderBlock) render_inline_persistent_kernel(const RenderArgs a)
{
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    const unsigned long long n_local = (unsigned long long)a.rows_local * (unsigned long long)a.width;
    const int limit = a.bounce_limit, n_spp = a.n_spp;
    const V3 origin = a.cam.pos;
    const int lane = threadIdx.x & 63;
    unsigned int live = 0;

    unsigned long long pixel = 0;
    V3 acc = mk(0, 0, 0), primary = mk(0, 0, 0), p0 = mk(0, 0, 0), n0 = mk(0, 0, 0);
    V3 hit_pos = p0, normal = n0, o = origin, d = primary;
    V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
    Sfc32 seed; seed.a = seed.b = seed.c = seed.counter = 0;
    int s = 0, it = 0, idx = 0, idx0 = 0;
    bool pending = false, has_ray = false, is_primary = false, finished = false, exhausted = false;
    unsigned long long pool_next = 0, pool_end = 0;          // wave-uniform
    bool queue_empty = false;                                // wave-uniform

    diag::PhaseProbe phase;
    for (;;) {
        phase.trip();
        // ---- shade: twice, so that a lane whose sample ends in the first round starts the next in the second
        for (int round = 0; round < 2; ++round) {
            if (pending && !has_ray) {
                phase.round_a(round == 0); phase.round_b(round != 0);
                shade(M, idx, hit_pos, normal, o, d, throughput, result, seed);
                ++it; ++live;
                // the next prepareRay would freeze the path (Trace.hs:364-365)
                if (it >= limit || near_zero(throughput)) {
                    acc = result + acc;                       // \(new, seed') (old, _) -> (new + old, seed')
                    ++s; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    hit_pos = p0; normal = n0; idx = idx0; d = primary;
                    pending = s < n_spp;
                    finished = !pending;
                } else {
                    pending = false; has_ray = true;
                }
            }
        }
        // ---- retire finished pixels (one store site), then hand out new ones
        if (finished) {
            a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
            a.planes.sa[pixel] = seed.a; a.planes.sb[pixel] = seed.b;
            a.planes.sc[pixel] = seed.c; a.planes.sctr[pixel] = seed.counter;
            finished = false;
        }
        const bool want = !pending && !has_ray && !exhausted;
        const unsigned long long want_mask = __ballot(want);
        // The hand-out block (index arithmetic, seven plane loads, primary ray set-up) runs for the whole wave
        // whenever it runs, so lanes wait until kFetchBatch of them want a pixel -- or nothing else is in flight.
        if (want_mask && (__builtin_popcountll(want_mask) >= kFetchBatch || !__any(has_ray || pending))) {   // wave-uniform
            // The wave owns a pool [pool_next, pool_end) of consecutive pixels, refilled kChunk at a time with
            // ONE atomic on the global counter (a single counter word serves only ~90 requests/us on this chip:
            // one atomic per fetched pixel made the kernel 3x slower).  Wanting lanes take pool entries by rank.
            const unsigned int n_want = (unsigned int)__builtin_popcountll(want_mask);
            const unsigned int rank = (unsigned int)__builtin_popcountll(want_mask & ((1ull << lane) - 1ull));
            unsigned long long mine = ~0ull;
            unsigned int avail = (unsigned int)(pool_end - pool_next);
            unsigned int take = n_want < avail ? n_want : avail;
            if (want && rank < take) mine = pool_next + rank;
            pool_next += take;
            if (take < n_want && !queue_empty) {
                const int leader = (int)__builtin_ctzll(want_mask);
                unsigned int base = 0;
                if (lane == leader) base = atomicAdd(a.work_counter, (unsigned int)kChunk);
                base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
                if ((unsigned long long)base >= n_local) {
                    queue_empty = true;
                } else {
                    pool_next = base;
                    pool_end = (unsigned long long)base + kChunk < n_local ? (unsigned long long)base + kChunk : n_local;
                    avail = (unsigned int)(pool_end - pool_next);
                    const unsigned int more = n_want - take < avail ? n_want - take : avail;
                    if (want && rank >= take && rank - take < more) mine = pool_next + (rank - take);
                    pool_next += more;
                }
            }
            if (want) {
                if (mine != ~0ull) {
                    pixel = mine;
                    const int local_row = (int)(pixel / (unsigned long long)a.width);
                    const int col = (int)(pixel - (unsigned long long)local_row * (unsigned long long)a.width);
                    int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
                    if (a.screen_x) { px = a.screen_x[pixel]; py = a.screen_y[pixel]; }
                    primary = primary_direction(a.cam, px, py);
                    acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
                    seed.a = a.planes.sa[pixel]; seed.b = a.planes.sb[pixel];
                    seed.c = a.planes.sc[pixel]; seed.counter = a.planes.sctr[pixel];
                    o = origin; d = primary;
                    s = 0; it = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                    has_ray = true; is_primary = true;
                } else if (queue_empty) {
                    exhausted = true;
                }
            }
        }
        if (!__any(has_ray || pending)) break;               // nothing left in flight in this wave
        phase.round_c(has_ray);
        // ---- trace: every lane that has a ray (next bounce, or the primary ray of a fresh pixel)
        if (has_ray) {
            const HitSel h = check_hit(S, ns, np, o, d);
            has_ray = false;
            if (h.just) {
                hit_record(S, ns, h.idx, o, d, h.t, hit_pos, normal);
                idx = h.idx;
                pending = true;
                if (is_primary) { p0 = hit_pos; n0 = normal; idx0 = idx; is_primary = false; }
            } else if (is_primary) {
                acc = mk(0.0f, 0.0f, 0.0f) + acc;            // every sample: result 0, seed untouched
                is_primary = false; finished = true;
            } else {
                acc = result + acc;
                ++s; it = 0;
                throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                hit_pos = p0; normal = n0; idx = idx0; d = primary;
                pending = s < n_spp;
                finished = !pending;
            }
        }
    }

    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if (lane == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
    phase.flush_lanes_only(a.work_counter);
}


}  // namespace

hipError_t launch_render_inline_ablation(const RenderArgs &a, int variant, bool big_scene, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    if (variant == 1 || variant == 6) {
        // persistent grid; more workgroups than fit would only start late and find the queue empty: cap at 8 per CU
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        const int max_blocks = (cus > 0 ? cus : 256) * 8 * (256 / kRenderBlock);
        const unsigned int blocks = grid.x < (unsigned int)max_blocks ? grid.x : (unsigned int)max_blocks;
        e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
        if (variant == 1) return launch(render_inline_persistent_kernel<true>, dim3(blocks), block, lds, stream, a);
        else              return launch(render_inline_persistent_kernel<false>, dim3(blocks), block, 0, stream, a);
    }
    if (variant == 18) {                                      // round 1's loop (no frozen-shade shortcut), 8x8 tiles, LDS scene
        return launch(render_inline_modes_kernel<true, kCachedR1, 8>, dim3(tile_grid(a, 8)), block, lds, stream, a);
    }
    if (variant >= 10 && variant <= 12) {                    // pooled second shade round, W = 2 / 4 / 8 waves per workgroup
        const int w = variant == 10 ? 2 : variant == 11 ? 4 : 8;
        const dim3 pgrid(blocks_for(n_local, 64 * w)), pblock(64 * w);
        if (big_scene) {                                       // a scene too big to stage per workgroup: scalar loads
            if (w == 2)      return launch(render_inline_pooled_kernel<false, 2>, pgrid, pblock, 0, stream, a);
            else if (w == 4) return launch(render_inline_pooled_kernel<false, 4>, pgrid, pblock, 0, stream, a);
            else             return launch(render_inline_pooled_kernel<false, 8>, pgrid, pblock, 0, stream, a);
        } else {
            if (w == 2)      return launch(render_inline_pooled_kernel<true, 2>, pgrid, pblock, lds, stream, a);
            else if (w == 4) return launch(render_inline_pooled_kernel<true, 4>, pgrid, pblock, lds, stream, a);
            else             return launch(render_inline_pooled_kernel<true, 8>, pgrid, pblock, lds, stream, a);
        }
    }
    switch (variant) {
    case 2:  if (big_scene) return launch(render_inline_modes_kernel<false, kLockstep>, grid, block, 0, stream, a);
             else           return launch(render_inline_modes_kernel<true, kLockstep>, grid, block, lds, stream, a);
    case 3:  if (big_scene) return launch(render_inline_modes_kernel<false, kRegenerate>, grid, block, 0, stream, a);
             else           return launch(render_inline_modes_kernel<true, kRegenerate>, grid, block, lds, stream, a);
    default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace ptmi
#endif  // PTMI_ABLATIONS
