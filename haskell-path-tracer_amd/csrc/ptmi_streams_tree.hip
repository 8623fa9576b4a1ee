// ptmi_streams_tree.hip -- render Streams for scenes whose rays SPLIT (the build-defined GLASS extension), per-pixel form: the
// tree walk (the default with GLASS).  The stream form of the same algorithm is ptmi_stream_split.hip.
#include "ptmi_device.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// render Streams for scenes whose rays SPLIT (the build-defined GLASS extension), per-pixel form: the tree walk.
// A lane owns a pixel and walks each sample's ray TREE depth first: at a GLASS hit the reflection child continues in
// the lane and the refraction child waits on a lane-private stack (scratch memory); when a lineage ends the lane pops
// the most recent waiting child, and only when the stack is empty does it go on with the sample's next start hit or
// the pixel's next sample.  Compared with the stream form below: no ray ever travels through HBM queues, a colour word
// has ONE adder (no atomics, and the order of a pixel's additions is defined: depth first, reflection before
// refraction -- oracle: ora_render_streams_tree, bit-exact), and the waves are dispatched by recorded cost, exactly as in
// render_streams_kernel.  The set of rays traced is the stream algorithm's (same children, same seeds, same step
// indices); a child that finds kTreeStackDepth children waiting in its lane is dropped and counted.
//
// THE START RECORD.  Every sample of a pixel shoots the same primary ray (Trace.hs:244-262), and a GLASS hit involves no
// random draw that changes a direction (glass_children only ADVANCES the seed): if the primary hit is glass, its two
// children are the same two rays in every sample too.  So what is evaluated once per pixel and kept in a lane-private
// LDS column is not only the primary hit but, for a glass primary hit, the first hit of EACH child -- the hits a sample
// starts from (0, 1 or 2 of them; a child that misses contributes nothing).  A sample then adds the glass hit's
// emittance (the same value every time), counts its two children, and works through its start hits in tree order, each
// with the seed its ray would carry: the sample's seed advanced by 3 (reflection) or 4 (refraction) raw draws.  Two
// traces and one glass evaluation per sample disappear for such pixels; everything downstream -- including start hits
// that are glass themselves -- takes the general path.  (With a step cap below 3 the children could be cut: no caching.)
// ---------------------------------------------------------------------------------------
#ifndef PTMI_TREE_WAVES
#define PTMI_TREE_WAVES 6        // 80 VGPRs (three values spilled around the shade) and a start record of 2 x 10 words: 8.39 -> 7.96 ms on the glass scene
#endif
template <bool LDS_SCENE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, PTMI_TREE_WAVES) render_streams_tree_kernel(const RenderArgs a)
{
    // two start hits per lane: position (3), incoming direction (3), throughput (3), primitive | steps << 16 | draws << 24
    // (The start hits' normals are recomputed at every sample start -- normal_at, the second half of hit_record.  Kept in LDS they save
    // 2.4 % at 5 waves per SIMD (8.39 -> 8.19 ms), but six waves' columns then no longer fit a CU, and the sixth wave is worth 5 %.)
#ifndef PTMI_TREE_NORMAL_LDS
#define PTMI_TREE_NORMAL_LDS 0
#endif
    constexpr int kEntry = PTMI_TREE_NORMAL_LDS ? 13 : 10;   // words per start hit
    __shared__ uint32_t start_rec[2 * kEntry][kRenderBlock];

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

    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);            // sample chunks, as in render_inline_kernel
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0, longest = 0, cut = 0, dropped = 0;
    const unsigned int step_cap = (unsigned int)a.stream_step_cap;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        const int64_t px = col, py = global_row(local_row, a.stripe_rows, a.n_parts, a.part);
        const V3 origin = a.cam.pos;
        const V3 primary = primary_direction(a.cam, px, py);
        V3 acc = mk(a.planes.r[pixel], a.planes.g[pixel], a.planes.b[pixel]);
        Sfc32 pixel_seed;
        pixel_seed.a = a.planes.sa[pixel]; pixel_seed.b = a.planes.sb[pixel];
        pixel_seed.c = a.planes.sc[pixel]; pixel_seed.counter = a.planes.sctr[pixel];
        const int n_spp = n_spp_chunk;

        const HitSel h0 = check_hit(S, ns, np, origin, primary);   // same primary ray for every sample
        if (!h0.just) {
            for (int s = 0; s < n_spp; ++s) (void)random_float(pixel_seed);     // updateSeed only
        } else {
            uint32_t *rec = &start_rec[0][threadIdx.x];
            auto put_entry = [&](int e, V3 p, V3 nrm, V3 dir, V3 t, int prim, unsigned int steps_done, unsigned int draws) {
                uint32_t *q = rec + (size_t)e * kEntry * kRenderBlock;
                q[0] = f2u(p.x); q[kRenderBlock] = f2u(p.y); q[2 * kRenderBlock] = f2u(p.z);
                q[3 * kRenderBlock] = f2u(dir.x); q[4 * kRenderBlock] = f2u(dir.y); q[5 * kRenderBlock] = f2u(dir.z);
                q[6 * kRenderBlock] = f2u(t.x); q[7 * kRenderBlock] = f2u(t.y); q[8 * kRenderBlock] = f2u(t.z);
                q[9 * kRenderBlock] = (uint32_t)prim | (steps_done << 16) | (draws << 24);
                if (PTMI_TREE_NORMAL_LDS) { q[10 * kRenderBlock] = f2u(nrm.x); q[11 * kRenderBlock] = f2u(nrm.y); q[12 * kRenderBlock] = f2u(nrm.z); }
            };
            V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            int n_entries = 0;
            bool first_is_reflection = false;
            V3 emit0 = mk(0.0f, 0.0f, 0.0f);
            const float4 ma0 = M[2 * h0.idx], mb0 = M[2 * h0.idx + 1];
#ifdef PTMI_TREE_NO_PREFIX
            const bool prefix = false;
#else
            const bool prefix = f2u(mb0.x) == 2u && step_cap >= 3u;   // a glass primary hit whose children cannot be cut
#endif
            if (prefix) {
                emit0 = scale_r(mk(ma0.x, ma0.y, ma0.z), ma0.w) * mk(1.0f, 1.0f, 1.0f);     // computeResult of the primary hit
                V3 ko[2], kd[2], kt[2]; Sfc32 ks[2];
                glass_children(mk(ma0.x, ma0.y, ma0.z), glass_constants_of<LDS_SCENE>(mb0), pos, normal, primary, mk(1.0f, 1.0f, 1.0f), pixel_seed, ko, kd, kt, ks);
                for (int k = 0; k < 2; ++k) {
                    const V3 ro = k == 0 ? ko[0] : ko[1], rd = k == 0 ? kd[0] : kd[1], rt = k == 0 ? kt[0] : kt[1];
                    const HitSel h = check_hit<LDS_SCENE && kStagedWalk>(S, ns, np, ro, rd);
                    if (h.just) {
                        V3 hp, hn;
                        hit_record(S, ns, h.idx, ro, rd, h.t, hp, hn);
                        if (n_entries == 0) first_is_reflection = k == 0;
                        put_entry(n_entries++, hp, hn, rd, rt, h.idx, 1u, (unsigned int)k);
                    }
                }
                // THE LEAD SEED.  A sample's reflection child carries the sample's seed advanced by 3 raw draws, its refraction
                // child by 4, and updateSeed moves the pixel's seed on by one: so from here to the end of the pixel `pixel_seed`
                // holds the pixel's seed advanced by 3 -- the reflection's seed as it stands, the refraction's one step further --
                // and is stepped back three times before it is stored (sfc32_prev, the exact inverse).  Seven SFC32 steps per
                // sample become at most two.
                (void)sfc32_next(pixel_seed); (void)sfc32_next(pixel_seed); (void)sfc32_next(pixel_seed);
            } else {
                put_entry(0, pos, normal, primary, mk(1.0f, 1.0f, 1.0f), h0.idx, 0u, 0u);
                n_entries = 1;
            }
            // children waiting for this lane: origin, direction, throughput, seed, step index (RayState, Trace.hs:45); scratch
            // memory.  (Before the start record existed, every sample of a glass pixel pushed a child and the first entry
            // lived in LDS: 23 GB -> 1.4 GB of scratch writes per launch.  With the primary split cached, pushes are rare,
            // and an LDS entry beside the start record would cost a wave of occupancy: 10.2 ms instead of 9.1.)
            // The lane's waiting children, a stack.  Its first kTreeFastLevels entries are 64-byte records in a global-memory block
            // laid out [tile][level][lane] -- a push is four 16-byte stores to ONE line, a pop four loads -- and only deeper
            // entries live in scratch memory, where a push is fourteen lane-strided dwords, each a partial line: with the
            // whole stack in scratch the kernel wrote 4 GB per 1080p / 64-spp launch (70 times the planes).
            uint32_t stack_w[kTreeStackDepth - kTreeFastLevels][14];
            float4 *const fast = a.tree_stack + ((size_t)wg * kTreeFastLevels * kRenderBlock + threadIdx.x) * 4;   // (never NULL: the call fails without the block)
            int sp = 0, entry_i = 0;
            int s = 0, idx = h0.idx;
            unsigned int steps = 0, deepest = 0;                 // deepest: traceSteps of the sample's longest lineage
            V3 d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f);
            Sfc32 seed = pixel_seed;
            bool pending = false, has_ray = false;
            auto begin_sample = [&]() {                           // what every sample of this pixel has already behind it
                if (prefix) { acc = acc + emit0; live += 2u; }
                entry_i = 0; deepest = prefix ? 2u : 1u;
            };
            // the next hit this sample starts from -- or, when it has none left, the end of the sample and the next one
            auto next_start = [&]() {
                for (;;) {
                    if (entry_i < n_entries) {
                        const uint32_t *q = rec + (size_t)entry_i * kEntry * kRenderBlock;
                        pos = mk(u2f(q[0]), u2f(q[kRenderBlock]), u2f(q[2 * kRenderBlock]));
                        d = mk(u2f(q[3 * kRenderBlock]), u2f(q[4 * kRenderBlock]), u2f(q[5 * kRenderBlock]));
                        throughput = mk(u2f(q[6 * kRenderBlock]), u2f(q[7 * kRenderBlock]), u2f(q[8 * kRenderBlock]));
                        idx = (int)(q[9 * kRenderBlock] & 0xffffu);
                        const uint32_t meta = q[9 * kRenderBlock] >> 16;
                        ++entry_i;
                        steps = meta & 0xffu;
                        normal = PTMI_TREE_NORMAL_LDS ? mk(u2f(q[10 * kRenderBlock]), u2f(q[11 * kRenderBlock]), u2f(q[12 * kRenderBlock])) : normal_at(S, ns, idx, pos);
                        seed = pixel_seed;                        // (with a prefix: the lead seed -- the draws the ray's ancestors made)
                        if (meta >> 8) (void)sfc32_next(seed);    // the refraction child: one more
                        pending = true; has_ray = false;
                        return;
                    }
                    (void)random_float(pixel_seed);               // updateSeed: the sample's tree is done
                    ++s; longest = deepest > longest ? deepest : longest;
                    if (s >= n_spp) { pending = false; has_ray = false; return; }
                    begin_sample();
                }
            };
            auto lineage_ended = [&]() {
                if (sp > 0) {                                     // the most recent waiting child
                    --sp;
                    uint32_t e[14];
                    if (sp < kTreeFastLevels) {
                        const float4 *r = fast + (size_t)sp * kRenderBlock * 4;
                        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                        e[0] = f2u(r0.x); e[1] = f2u(r0.y); e[2] = f2u(r0.z); e[3] = f2u(r0.w); e[4] = f2u(r1.x); e[5] = f2u(r1.y); e[6] = f2u(r1.z);
                        e[7] = f2u(r1.w); e[8] = f2u(r2.x); e[9] = f2u(r2.y); e[10] = f2u(r2.z); e[11] = f2u(r2.w); e[12] = f2u(r3.x); e[13] = f2u(r3.y);
                    } else {
                        const int q0 = sp - kTreeFastLevels;
                        for (int q = 0; q < 14; ++q) e[q] = stack_w[q0 < kTreeStackDepth - kTreeFastLevels ? q0 : 0][q];
                    }
                    pos = mk(u2f(e[0]), u2f(e[1]), u2f(e[2]));
                    d = mk(u2f(e[3]), u2f(e[4]), u2f(e[5]));
                    throughput = mk(u2f(e[6]), u2f(e[7]), u2f(e[8]));
                    seed.a = e[9]; seed.b = e[10]; seed.c = e[11]; seed.counter = e[12];
                    steps = e[13];
                    pending = false; has_ray = true;
                } else {
                    next_start();
                }
            };
            if (n_spp > 0) { begin_sample(); next_start(); }
            diag::TreeProbe probe;                                // (diagnostic builds: ptmi_diag.h)
            bool ended = false;                                   // the lane's lineage is over: its next piece of work is fetched at the top of the trip
            while (pending || has_ray || ended) {
                ++trips;
                probe.dead(pending && !has_ray && near_zero(throughput));
                // shade round.  A ray whose throughput is already near zero dies at this hit (numNewRays): the hit adds its
                // emittance and nothing else of it survives, so such lanes skip the expensive half and go on with their most
                // recent waiting child, their sample's next start hit or the pixel's next sample -- in the latter cases they
                // take part in this round's full shade.  A lineage that ended in the previous trip (miss, cap) fetches its next
                // piece of work here too: lineage_ended is a large block -- next start hit, its normal, its seed -- and is expanded
                // once.
                if (pending && !has_ray && near_zero(throughput)) {
                    const float4 ma = M[2 * idx];
                    acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);   // computeResult (Trace.hs:318-323)
                    ++steps;
                    pending = false; ended = true;
                }
                if (ended) { lineage_ended(); ended = false; }     // the one expansion of that block (it is large)
                // (The start hit that the block above may just have loaded can itself belong to a dead ray -- a reflection of
                // weight ~0: it must not be shaded; it waits for the next trip's dead-ray block.  A test inside next_start
                // instead cost 12 %.)
                if (pending && !has_ray && !near_zero(throughput)) {   // alive
                    probe.shade();
                    // GLASS is an ARM of the one shade (as in the stream form's split kernel, ptmi_stream_split.hip): genVec's three draws come
                    // before the match for every material, ia = dir . n and the mirror direction are Glossy's too, and the reflection child is
                    // what the common tail makes of `next = reflection` and the factor R; only Schlick's R, the refraction direction and the
                    // SECOND child -- which waits on the lane's stack -- are a divergent block.  Operation for operation the oracle's
                    // glass_children (glass_refraction_child, ptmi_device.h), so the planes stay bit-identical.
                    const float4 ma = M[2 * idx], mb = M[2 * idx + 1];
                    const bool capped = steps + 1u >= step_cap;
                    const bool glass = f2u(mb.x) == 2u, matte = f2u(mb.x) == 0u;
                    const V3 color = mk(ma.x, ma.y, ma.z);
                    V3 rv;
                    rv.x = gen_component(seed); rv.y = gen_component(seed); rv.z = gen_component(seed);
                    const float ia = dot(d, normal);
                    const V3 reflection = d - scale_l(2.0f * ia, normal);
                    ++steps;
                    float glass_R = 0.0f;
                    if (glass) {                                  // the refraction child (extension; spec = the oracle's glass_children)
                        V3 ko, kd, kt; Sfc32 ks;
                        glass_R = glass_refraction_child(color, glass_constants_of<LDS_SCENE>(mb), pos, normal, d, ia, reflection, throughput, seed, ko, kd, kt, ks);
                        // while the cached reflection's subtree is walked, the cached refraction "waits": one slot less
                        if (capped) {
                        } else if (sp < kTreeStackDepth - ((prefix && entry_i == 1 && first_is_reflection) ? 1 : 0)) {
                            if (sp < kTreeFastLevels) {
                                float4 *r = fast + (size_t)sp * kRenderBlock * 4;
                                r[0] = float4{ko.x, ko.y, ko.z, kd.x};
                                r[1] = float4{kd.y, kd.z, kt.x, kt.y};
                                r[2] = float4{kt.z, u2f(ks.a), u2f(ks.b), u2f(ks.c)};
                                r[3] = float4{u2f(ks.counter), u2f(steps), 0.0f, 0.0f};
                            } else {
                                const uint32_t e[14] = {f2u(ko.x), f2u(ko.y), f2u(ko.z), f2u(kd.x), f2u(kd.y), f2u(kd.z),
                                                        f2u(kt.x), f2u(kt.y), f2u(kt.z), ks.a, ks.b, ks.c, ks.counter, steps};
                                const int q0 = sp - kTreeFastLevels;
                                for (int q = 0; q < 14; ++q) stack_w[q0 < kTreeStackDepth - kTreeFastLevels ? q0 : 0][q] = e[q];
                            }
                            ++sp;
                        } else {
                            ++dropped;
                        }
                    }
                    // Matte: rotate (anglesToQuaternion $ pi *^ rv) iNormal | Glossy: rotate (anglesToQuaternion $ (1 - p) *^ rv) reflection
                    const V3 axis = matte ? normal : reflection;
                    const float hk = matte ? 0.5f * kPi : mb.w;
                    const V3 rotated = rotate(quaternion_from_half_angles(hk * rv.x, hk * rv.y, hk * rv.z), axis);
                    const float nd = dot(rotated, axis);
                    const float brdf = matte ? mb.z * nd : __builtin_fmaxf(0.0f, nd);
                    constexpr float next_ray_prob = 1.0f / (kPi * 2.0f);
                    const float factor = glass ? glass_R : brdf * next_ray_prob;
                    const V3 next = mk(glass ? reflection.x : rotated.x, glass ? reflection.y : rotated.y, glass ? reflection.z : rotated.z);
                    // results: colour += emittance * throughput for EVERY hit (computeResult, Trace.hs:318-323); then the new ray
                    acc = acc + (scale_r(color, ma.w) * throughput);
                    pos = pos + scale_r(next, kEpsilon);
                    d = next;
                    throughput = throughput * scale_r(color, factor);
                    live += glass ? 2u : 1u;
                    pending = false;
                    if (capped) { cut += glass ? 2u : 1u; ended = true; }
                    else has_ray = true;
                }
                if (has_ray) {
                    probe.trace();
                    deepest = steps + 1u > deepest ? steps + 1u : deepest;
                    const HitSel h = check_hit<LDS_SCENE && kStagedWalk>(S, ns, np, pos, d);
                    has_ray = false;
                    if (h.just) {
                        hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                        idx = h.idx;
                        pending = true;
                    } else {
                        ended = true;
                    }
                }
            }
            probe.flush_lane(a.work_counter, trips);
            if (prefix) { sfc32_prev(pixel_seed); sfc32_prev(pixel_seed); sfc32_prev(pixel_seed); }     // the lead seed back to the pixel's
        }
        diag::TreeProbe().cost_map(acc.x, trips);
        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = pixel_seed.a; a.planes.sb[pixel] = pixel_seed.b;
        a.planes.sc[pixel] = pixel_seed.c; a.planes.sctr[pixel] = pixel_seed.counter;
    }
    leave_sample_chunk<TILE_W>(a, wg, chunk);
    diag::TreeProbe().flush_wave(a.work_counter, trips);
    if (TILE_W > 0) record_cost(a, quad, trips);
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
    if (a.stream_iterations) {
        for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
        if ((threadIdx.x & 63) == 0 && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
    if (__any((cut | dropped) != 0u)) {                               // rare
        const unsigned long long n_cut = wave_sum(cut), n_dropped = wave_sum(dropped);
        if ((threadIdx.x & 63) == 0) {
            if (n_cut) atomicAdd(a.stream_counters + kScTruncated, n_cut);
            if (n_dropped) atomicAdd(a.stream_counters + kScDropped, n_dropped);
        }
    }
}

}  // namespace

// workgroups (per copy of the grid) of the tree walk = records' worth of RenderArgs.tree_stack: x kTreeFastLevels x 64 lanes x 64 B
unsigned int tree_workgroups(int width, int rows_local)
{
    if (tiles_pay_dims(width, rows_local)) return quad_positions(width, rows_local) * 4u;
    return (unsigned int)(((long long)width * rows_local + kRenderBlock - 1) / kRenderBlock);
}

hipError_t launch_render_streams_tree(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    hipError_t e = hipMemsetAsync(a.stream_iterations, 0, (size_t)kStatShards * 2 * kStatStride * sizeof(unsigned int), stream);   // every shard: the figure is per launch
    if (e != hipSuccess) return e;
    const bool scalar_scene = variant == 5 || variant == 17 || lds > kMaxSceneLds;
    const bool tiles = variant == 4 || variant == 5 ? false : tiles_pay(a);
    if (tiles) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t ce = choose_sample_chunks(b, per_copy, PTMI_TREE_WAVES, stream, 10)) return ce;
        const dim3 tgrid(per_copy * (unsigned int)b.spp_chunks);
        if (scalar_scene) return launch(render_streams_tree_kernel<false, 8>, tgrid, block, 0, stream, b);
        else              return launch(render_streams_tree_kernel<true, 8>, tgrid, block, lds, stream, b);
    } else {
        if (scalar_scene) return launch(render_streams_tree_kernel<false>, grid, block, 0, stream, a);
        else              return launch(render_streams_tree_kernel<true>, grid, block, lds, stream, a);
    }
}

}  // namespace ptmi
