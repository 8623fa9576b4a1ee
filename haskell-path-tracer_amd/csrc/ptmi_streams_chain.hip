// ptmi_streams_chain.hip -- render Streams (src/Scene/Trace.hs:141-191, 272-331), one chain per pixel: the default of Streams for
// every scene whose rays never split, and the per-pixel TAIL of the stream form (ptmi_stream_pixels.hip).
#include "ptmi_device.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// render Streams (Trace.hs:141-191, 272-331).  The reference keeps one ray per pixel in a stream
// that `expand` compacts after every step (numNewRays is 0 or 1, Trace.hs:329-331) and scatters the
// colours back with `permute (+)`; because a pixel never owns more than one ray, the stream is the
// per-pixel chain below and the compaction becomes "a lane whose ray died starts its pixel's next
// sample" -- the wave stays dense without moving ray state through memory.  What differs from
// Inline, and is reproduced literally:
//   * every hit adds emittance * throughput straight into the accumulator, also in the step where the
//     throughput is already near zero (computeResult runs for every intersection, Trace.hs:290-293);
//   * the ray dies when nearZero throughput || miss (Trace.hs:329-331); there is NO bounce limit -- a
//     non-empty stream is never stopped by the iteration count (Trace.hs:166-170).  a.stream_step_cap only
//     guarantees that the kernel terminates (rays it cuts are counted, stream_counters[kScTruncated]);
//   * which seed the pixel carries out of `combine` (Trace.hs:179-184) is Accelerate-backend behaviour
//     (assumption A5).  Default: the pixel keeps its OLD seed while the sample runs;
//     a.seed_from_result: the seed of the ray that made the sample's LAST hit replaces it (kept in the lane's LDS
//     column, not in registers).  Either way updateSeed then advances the pixel's seed by one
//     draw (Trace.hs:151, :190-191).
// ---------------------------------------------------------------------------------------

#ifndef PTMI_STREAMS_WAVES
#define PTMI_STREAMS_WAVES 7     // 72 VGPRs (one pair spilled around the loop, not in it): C2 4.27 -> 4.17 ms
#endif
template <bool LDS_SCENE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, PTMI_STREAMS_WAVES) render_streams_kernel(const RenderArgs a)
{
    __shared__ float pixel_const[11][kRenderBlock];         // per-lane restart record (rows 0..6) and the last hit's seed (7..10)
    extern __shared__ float4 lds_scene[];
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    // As the tail of the stream form the grid covers every dispatch position, and its workgroups start where the stream form's part
    // ends (a device word): those that would pass the last position have nothing to do.
    if (TILE_W > 0 && a.first_position && blockIdx.x + 4u * *a.first_position >= gridDim.x) return;
    if (LDS_SCENE) {
        const int total = a.scene.total_f4();
        for (int i = threadIdx.x; i < total; i += kRenderBlock) lds_scene[i] = a.scene.packed[i];
        __syncthreads();
    }
    const float4 *S = LDS_SCENE ? lds_scene : a.scene.packed;
    const float4 *M = S + a.scene.geom_f4();

    unsigned int wg; int chunk, n_spp_chunk;
    enter_sample_chunk<TILE_W>(a, wg, chunk, n_spp_chunk);            // sample chunks, as in render_inline_kernel
    long long pixel;
    unsigned int quad, trips = 0;
    const bool valid = lane_pixel<TILE_W>(a, pixel, quad, wg);
    unsigned int live = 0, longest = 0, cut = 0;
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
            float *mine = &pixel_const[0][threadIdx.x];
            auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
            auto get = [&](int k) { return mine[k * kRenderBlock]; };
            V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
            hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
            const int idx0 = h0.idx;
            {   // what every sample of this pixel starts from: the primary hit and the axis / half-angle scale of its bounce
                const float4 mb0 = M[2 * idx0 + 1];
                V3 axis0; float hk0;
                bounce_axis(mb0, normal, primary, axis0, hk0);
                put(0, pos.x); put(1, pos.y); put(2, pos.z);
                put(3, axis0.x); put(4, axis0.y); put(5, axis0.z); put(6, hk0);
            }
            // PTMI_SEED_FROM_RESULT (combine new old): every hit leaves the seed its ray carried in rows 7..10 of the lane's
            // LDS column (four ds_writes per hit instead of four more registers); the sample's last one survives.
            auto note_hit_seed = [&](const Sfc32 &sd) {
                if (a.seed_from_result) { put(7, u2f(sd.a)); put(8, u2f(sd.b)); put(9, u2f(sd.c)); put(10, u2f(sd.counter)); }
            };
            int s = -1, idx = idx0;                               // s: the sample being rendered (the first pass through the block below makes it 0)
            unsigned int steps = 0;
            V3 d = primary;
            V3 throughput = mk(1.0f, 1.0f, 1.0f);
            Sfc32 seed = pixel_seed;
            bool pending = false, has_ray = false, over = n_spp > 0;
            // Loop shape [finish dead rays][next sample][shade][trace]: a lane comes round with a hit to shade (`pending`) or with
            // its sample over (`over`: the trace missed, or the cap cut the child).  ONE block per trip ends the samples that
            // are over and starts the pixel's next one from the cached primary hit, so that those lanes take part in this
            // trip's full shade.
            while (pending || over) {
                ++trips;
                float4 mb = M[2 * idx + 1];
                V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
                if (pending) {
                    // A ray whose throughput is already near zero dies at this hit (numNewRays, Trace.hs:329-331): the hit still
                    // adds its emittance (computeResult runs for every intersection) and nothing else of it survives -- no
                    // child, and the ray's seed is discarded -- so such lanes skip the three sin/cos pairs and the rotation.
                    if (near_zero(throughput)) {
                        const float4 ma = M[2 * idx];
                        acc = acc + (scale_r(mk(ma.x, ma.y, ma.z), ma.w) * throughput);
                        note_hit_seed(seed);
                        ++steps;
                        pending = false; over = true;
                    } else {
                        bounce_axis(mb, normal, d, axis, hk);
                    }
                }
                if (over) {
                    if (s >= 0) {                                  // a sample has been rendered
                        if (a.seed_from_result && steps > 0u) {    // combine new old: the seed the sample's last hit carried
                            pixel_seed.a = f2u(get(7)); pixel_seed.b = f2u(get(8)); pixel_seed.c = f2u(get(9)); pixel_seed.counter = f2u(get(10));
                        }
                        (void)random_float(pixel_seed);            // updateSeed
                        longest = steps > longest ? steps : longest;
                    }
                    seed = pixel_seed;
                    ++s; steps = 0;
                    throughput = mk(1.0f, 1.0f, 1.0f);
                    pos = mk(get(0), get(1), get(2)); idx = idx0;
                    mb = M[2 * idx0 + 1];
                    axis = mk(get(3), get(4), get(5)); hk = get(6);
                    over = false; pending = s < n_spp;
                }
                if (pending) {                                     // alive (a fresh sample starts with throughput 1)
                    const bool capped = steps + 1u >= step_cap;
                    note_hit_seed(seed);
                    // results: colour += emittance * throughput for EVERY hit; then the new ray (shade, with the axis in hand)
                    V3 next; float brdf;
                    next_about_axis(mb, axis, hk, seed, next, brdf);
                    apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, acc);
                    ++steps; ++live;                               // the child exists even if the cap then cuts it
                    pending = false;
                    if (capped) { ++cut; over = true; }
                    else has_ray = true;
                }
                if (has_ray) {
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
        }
        a.planes.r[pixel] = acc.x; a.planes.g[pixel] = acc.y; a.planes.b[pixel] = acc.z;
        a.planes.sa[pixel] = pixel_seed.a; a.planes.sb[pixel] = pixel_seed.b;
        a.planes.sc[pixel] = pixel_seed.c; a.planes.sctr[pixel] = pixel_seed.counter;
    }
    leave_sample_chunk<TILE_W>(a, wg, chunk);
    if (TILE_W > 0) {
        if (a.first_position) {                               // the stream form's unit: the hits the tile's pixels shaded (record_item_cost)
            const unsigned long long hits = wave_sum(live);
            if (a.quad_cost && (threadIdx.x & 63) == 0) atomicAdd(a.quad_cost + quad, (unsigned int)hits);
        } else {
            record_cost(a, quad, trips);
        }
    }
    if (a.live_counter) {
        const unsigned long long total = wave_sum(live);
        if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.live_counter + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride, total);
    }
    if (a.stream_iterations) {
        for (int off = 32; off > 0; off >>= 1) { const unsigned int other = __shfl_xor(longest, off, 64); longest = other > longest ? other : longest; }
        if ((threadIdx.x & 63) == 0 && longest) atomicMax(a.stream_iterations + (size_t)(blockIdx.x & (kStatShards - 1)) * (2 * kStatStride), longest);
    }
    if (__any(cut != 0u)) {                                           // rare: only when the safety cap bites
        const unsigned long long total = wave_sum(cut);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.stream_counters + kScTruncated, total);
    }
}

}  // namespace

hipError_t launch_render_streams(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    hipError_t e = hipMemsetAsync(a.stream_iterations, 0, (size_t)kStatShards * 2 * kStatStride * sizeof(unsigned int), stream);   // every shard: the figure is per launch
    if (e != hipSuccess) return e;
    const bool scalar_scene = variant == 5 || variant == 6 || variant == 17 || lds > kMaxSceneLds;
    const bool tiles = variant == 4 || variant == 5 ? false : tiles_pay(a);      // 4 / 5 keep the row mapping (ablation)
    if (tiles) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t ce = choose_sample_chunks(b, per_copy, PTMI_STREAMS_WAVES, stream)) return ce;
        const dim3 tgrid(per_copy * (unsigned int)b.spp_chunks);
        if (scalar_scene) return launch(render_streams_kernel<false, 8>, tgrid, block, 0, stream, b);
        else              return launch(render_streams_kernel<true, 8>, tgrid, block, lds, stream, b);
    } else {
        if (scalar_scene) return launch(render_streams_kernel<false>, grid, block, 0, stream, a);
        else              return launch(render_streams_kernel<true>, grid, block, lds, stream, a);
    }
}

// The per-pixel chain kernel as the TAIL of a stream-form launch: a grid over every dispatch position whose workgroups start at
// *first_position (RenderArgs.first_position); no sample chunks; stream_iterations is the stream form's to clear.
hipError_t launch_render_streams_tail(const RenderArgs &a, const unsigned int *first_position, hipStream_t stream)
{
    if (!tiles_pay(a) || !first_position) return hipSuccess;
    RenderArgs b = a;
    b.spp_chunks = 1; b.first_position = first_position;
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const dim3 grid(tile_grid(a, 8)), block(kRenderBlock);
    if (lds > kMaxSceneLds) return launch(render_streams_kernel<false, 8>, grid, block, 0, stream, b);
    else                    return launch(render_streams_kernel<true, 8>, grid, block, lds, stream, b);
}

}  // namespace ptmi
