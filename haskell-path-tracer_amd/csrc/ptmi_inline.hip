// ptmi_inline.hip -- render Inline (src/Scene/Trace.hs:193-200 + 344-383) for gfx950: the default kernel at every size.
//
//   * one lane per pixel; the seven state planes are read once and written once per launch, coalesced (x fastest), whatever the
//     sample count;
//   * the sample loop AND the bounce loop live in the kernel.  A lane whose path ends starts its pixel's next sample at once
//     ("regeneration"), so the 64 lanes of a wave stay busy although paths end after different numbers of bounces -- the
//     per-pixel order of RNG draws and of floating-point additions is exactly that of n_spp successive `render` calls;
//   * the primitive list is staged into LDS once per workgroup and read as wave-wide broadcasts (every lane walks the same
//     primitive at the same time);
//   * a shade whose outcome the next prepareRay is certain to freeze only adds its emittance and draws (surely_frozen_after);
//   * each LARGE block -- "start the pixel's next sample" -- is expanded once per trip: the sites that end a sample only set a
//     per-lane flag, and one block at the top of the next trip acts on it (three inlined copies cost 3 %);
//   * no MFMA: the work is scalar-per-lane f32/f64 VALU with divergent control flow.
// The ablation loops of DESIGN.md 5.2 (round 1's loop, regenerate-only, lock step, pooled shade round, persistent hand-out) are in
// ptmi_inline_ablations.hip, which only builds with -DPTMI_ABLATIONS.
// This unit is compiled a second time with -DPTMI_CONTRACTED_BUILD -ffp-contract=fast -Dptmi=ptmi_contracted (a * b + c fused: a
// labelled measurement mode, see the end of the file).
#include "ptmi_device.h"

namespace ptmi {

namespace {

// ---------------------------------------------------------------------------------------
// render Inline.  LDS_SCENE: primitives staged in LDS (default) or read straight from global memory through scalar loads
// (scenes over 3 KB).  TILE_W: the wave's pixels are an 8 x 8 tile (8) or 64 consecutive pixels of a row (0).
// The primary hit is evaluated once per pixel; loop [finish frozen shades + restart][shade][trace].
// ---------------------------------------------------------------------------------------
#ifndef PTMI_INLINE_WAVES
#define PTMI_INLINE_WAVES 7      // 72 VGPRs (three registers spilled around the loop, not in it) and a 10-word LDS column: C2 3.10 -> 3.04 ms
#endif
template <bool LDS_SCENE, int TILE_W = 0>
__global__ void __launch_bounds__(kRenderBlock, PTMI_INLINE_WAVES) render_inline_kernel(const RenderArgs a)
{
    __shared__ float pixel_const[10][kRenderBlock];         // per-lane restart record (see below)
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
        } else {
            // primaryRays has no sub-pixel jitter (Trace.hs:244-262): every sample of a pixel shoots the same primary ray, so
            // its checkHit + hit are evaluated ONCE per pixel and every sample starts from that record.
            // Loop shape: [finish frozen shades][restart][shade][trace].  A lane comes round with a hit to shade (`pending`) or
            // with its sample over (`over`: the trace missed, or the last shade left a throughput that the next prepareRay
            // freezes).  The shades whose outcome is CERTAIN to be frozen (the iteration limit, or surely_frozen_after) are
            // finished first -- emittance + three draws, no sin/cos, no rotation -- and those lanes are `over` too; then ONE
            // block restarts every `over` lane on its pixel's next sample, from the cached primary hit, with the rotation axis
            // and half-angle scale that every first shade of the pixel uses; then one full shade and one trace for all.  A
            // sample whose path ends by a certain freeze -- 64 % of them on C2 -- costs k-1 full shades and k-1 traces.
            const HitSel h0 = check_hit<LDS_SCENE>(S, ns, np, origin, primary);
            if (!h0.just) {
                if (n_spp > 0) acc = mk(0.0f, 0.0f, 0.0f) + acc;     // every sample: result 0, seed untouched
            } else {
                // What a sample restarts from lives in a lane-private LDS column (10 words), not in VGPRs: the position of the
                // primary hit, the axis and half-angle scale of its bounce, and the pixel's accumulator (touched once per sample).
                float *mine = &pixel_const[0][threadIdx.x];
                auto put = [&](int k, float v) { mine[k * kRenderBlock] = v; };
                auto get = [&](int k) { return mine[k * kRenderBlock]; };
                V3 pos, normal;                                       // pos: the hit to shade, then the next ray's origin
                hit_record(S, ns, h0.idx, origin, primary, h0.t, pos, normal);
                const int idx0 = h0.idx;
                {
                    const float4 mb0 = M[2 * idx0 + 1];
                    V3 axis0; float hk0;
                    bounce_axis(mb0, normal, primary, axis0, hk0);
                    put(0, pos.x); put(1, pos.y); put(2, pos.z);
                    put(3, axis0.x); put(4, axis0.y); put(5, axis0.z); put(6, hk0);
                    put(7, acc.x); put(8, acc.y); put(9, acc.z);
                }
                int s = -1, it = 0, idx = idx0;                       // s: the sample being rendered (the first restart makes it 0)
                V3 d = primary;
                V3 throughput = mk(1.0f, 1.0f, 1.0f), result = mk(0.0f, 0.0f, 0.0f);
                bool pending = false, has_ray = false, over = n_spp > 0;
                diag::PhaseProbe phase;                               // (diagnostic builds: ptmi_diag.h)
                while (pending || over) {
                    ++trips;
                    phase.trip(); phase.round_a(pending || over);
                    float4 mb = M[2 * idx + 1];
                    V3 axis = mk(0.0f, 0.0f, 0.0f); float hk = 0.0f;
                    phase.check(pending);
                    if (pending) {
                        bounce_axis(mb, normal, d, axis, hk);
                        const float4 ma = M[2 * idx];
                        if (it + 1 >= limit || surely_frozen_after(ma, mb, axis, throughput)) {
                            finish_frozen(ma, throughput, result, seed);
                            ++live;
                            phase.frozen();
                            pending = false; over = true;
                        }
                    }
                    phase.restart(over);
                    if (over) {                                        // next sample of this pixel
                        // \(new, seed') (old, _) -> (new + old, seed') -- once a sample has been rendered (the first time round
                        // the lane only starts sample 0)
                        if (s >= 0) { put(7, result.x + get(7)); put(8, result.y + get(8)); put(9, result.z + get(9)); }
                        ++s; it = 0;
                        throughput = mk(1.0f, 1.0f, 1.0f); result = mk(0.0f, 0.0f, 0.0f);
                        pos = mk(get(0), get(1), get(2)); idx = idx0;
                        mb = M[2 * idx0 + 1];
                        axis = mk(get(3), get(4), get(5)); hk = get(6);
                        over = false; pending = s < n_spp;
                    }
                    phase.shade(pending);
                    if (pending) {
                        V3 next; float brdf;
                        next_about_axis(mb, axis, hk, seed, next, brdf);
                        apply_bounce(M, idx, pos, next, brdf, pos, d, throughput, result);
                        ++it; ++live;
                        pending = false;
                        // the next prepareRay would freeze the path (Trace.hs:364-365)
                        if (it >= limit || near_zero(throughput)) over = true;
                        else has_ray = true;
                    }
                    phase.end_a(); phase.round_c(has_ray);
                    if (has_ray) {
                        const HitSel h = check_hit<LDS_SCENE>(S, ns, np, pos, d, diag::sphere_counters(a.work_counter));
                        has_ray = false;
                        if (h.just) {
                            hit_record(S, ns, h.idx, pos, d, h.t, pos, normal);
                            idx = h.idx;
                            pending = true;
                        } else {
                            over = true;
                        }
                    }
                    phase.end_c();
                }
                acc = mk(get(7), get(8), get(9));
                phase.flush(a.work_counter);
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

}  // namespace

// Which render Inline kernel a variant is (ptmi_set_variant):
//   0 auto | 4 cached, a wave = 64 consecutive pixels of a row (LDS scene) | 5 the same with the scene through scalar loads
//   13 = 4 with 8x8 pixel tiles per wave | 17 = 5 with 8x8 tiles -- these are what auto chooses from.
// Only in builds with -DPTMI_ABLATIONS (ptmi_set_variant refuses them otherwise):
//   1 / 6 persistent hand-out (LDS / scalar-load scene) | 2 lock step | 3 regenerate | 7 / 8 capped occupancy
//   10-12 pooled second shade round | 14-16 other tile shapes | 18 round 1's loop
hipError_t launch_render_inline(const RenderArgs &a, int variant, hipStream_t stream)
{
    const long long n_local = (long long)a.rows_local * a.width;
    if (n_local <= 0) return hipSuccess;
    const dim3 grid(blocks_for(n_local, kRenderBlock)), block(kRenderBlock);
    const size_t lds = (size_t)a.scene.total_f4() * sizeof(float4);
    const bool big_scene = lds > kMaxSceneLds;               // every route reads such a scene through scalar loads, not LDS
    const bool degenerate = a.bounce_limit <= 0 || a.n_spp <= 0;   // the cached kernel handles both (iterate 0; no sample at all)
    if (variant == 0 || degenerate) {
        // static mapping wins at every size measured; a scene so big that staging it per wave would cost more occupancy than scalar loads cost speed is
        // read through scalar loads; 8x8 tiles once the image is big enough for whole tiles to dominate
        const bool tiles = tiles_pay(a);
        variant = !big_scene ? (tiles ? 13 : 4) : (tiles ? 17 : 5);
    }
    if (big_scene) {                                         // the LDS forms would not fit or would cap occupancy
        if (variant == 1) variant = 6;
        else if (variant == 4 || variant == 7 || variant == 8) variant = 5;
        else if (variant >= 13 && variant <= 16) variant = 17;
    }
    if (variant == 17 || variant == 13) {
        RenderArgs b = a;
        const unsigned int per_copy = tile_grid(a, 8);
        if (hipError_t e = choose_sample_chunks(b, per_copy, PTMI_INLINE_WAVES, stream)) return e;
        const dim3 cgrid(per_copy * (unsigned int)b.spp_chunks);
        return variant == 17 ? launch(render_inline_kernel<false, 8>, cgrid, block, 0, stream, b)
                             : launch(render_inline_kernel<true, 8>, cgrid, block, lds, stream, b);
    }
    if (variant == 5) return launch(render_inline_kernel<false>, grid, block, 0, stream, a);
    if (variant == 4) return launch(render_inline_kernel<true>, grid, block, lds, stream, a);
#if defined(PTMI_ABLATIONS) && !defined(PTMI_CONTRACTED_BUILD)
    if (variant >= 14 && variant <= 16) {                     // other pixel tiles per wave: 16x4 / 4x16 / 32x2 (8x8 is handled above)
        const int tw = variant == 14 ? 16 : variant == 15 ? 4 : 32;
        const dim3 tgrid(tile_grid(a, tw));
        if (tw == 16) return launch(render_inline_kernel<true, 16>, tgrid, block, lds, stream, a);
        if (tw == 4)  return launch(render_inline_kernel<true, 4>, tgrid, block, lds, stream, a);
        return launch(render_inline_kernel<true, 32>, tgrid, block, lds, stream, a);
    }
    if (variant == 7 || variant == 8) {                       // capped occupancy through dynamic LDS: 4 / 3 waves per SIMD
        return launch(render_inline_kernel<true>, grid, block, (variant == 7 ? 33 : 41) * 1024, stream, a);
    }
    return launch_render_inline_ablation(a, variant, big_scene, stream);      // ptmi_inline_ablations.hip
#else
    return hipErrorInvalidValue;                             // ptmi_set_variant admits only what the build holds
#endif
}

#ifndef PTMI_CONTRACTED_BUILD
bool variant_available(int variant)
{
    if (variant == 0 || variant == 4 || variant == 5 || variant == 9 || variant == 13 || variant == 17) return true;
#ifdef PTMI_ABLATIONS
    return variant >= 0 && variant <= 18;
#else
    return false;
#endif
}
#endif

}  // namespace ptmi

#ifdef PTMI_CONTRACTED_BUILD
// THE CONTRACTED-ARITHMETIC OBJECT.  This unit is compiled a second time with -ffp-contract=fast and -Dptmi=ptmi_contracted
// (every name above then lives in namespace ptmi_contracted): the same kernel with a * b + c contracted into fused
// multiply-adds wherever the source writes it -- dot products, cross products, the rotation, the quaternion.  It is NOT the
// reference's arithmetic as this repository reads it (every operation rounded on its own); it exists to
// MEASURE how much of the kernel's time that reading costs (PTMI_OPT_ARITHMETIC, never the default, never the headline).  One C
// entry, because the two objects' RenderArgs are distinct types of identical layout.
extern "C" int ptmi_contracted_launch_inline(const void *args, int variant, void *stream)
{
    return (int)ptmi::launch_render_inline(*static_cast<const ptmi::RenderArgs *>(args), variant, static_cast<hipStream_t>(stream));
}
#endif
