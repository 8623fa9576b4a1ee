// ptmi_stream_primary.hip -- the stream form's small kernels: the start-hit list (streams_primary_kernel), updateSeed for the
// pixels without start hits, and the seed snapshots of the split kernel's passes.
#include "ptmi_stream_form.h"

namespace ptmi {

namespace {

template <bool TILES>
__global__ void __launch_bounds__(256) streams_primary_kernel(const RenderArgs a, const HitList out, unsigned int *counters)
{
    const int ns = a.scene.n_spheres, np = a.scene.n_planes;
    const float4 *S = a.scene.packed;                        // one evaluation per pixel and call: scalar loads will do
    const float4 *M = S + a.scene.geom_f4();
    const int lane = threadIdx.x & 63;
    unsigned int quad, region; long long pixel;
    const bool valid = primary_pixel<TILES>(a, quad, region, pixel);
    int n_rec = 0;                                            // records this pixel contributes: 0, 1 or 2
    V3 pos[2], nor[2], dir[2], thr[2];
    int prim[2] = {0, 0};
    uint32_t meta[2] = {0u, 0u};
    bool split = false;
    if (valid) {
        const int local_row = (int)(pixel / a.width);
        const int col = (int)(pixel - (long long)local_row * a.width);
        const V3 primary = primary_direction(a.cam, col, global_row(local_row, a.stripe_rows, a.n_parts, a.part));
        const HitSel h = check_hit(S, ns, np, a.cam.pos, primary);
        if (h.just) {
            V3 p0, n0;
            hit_record(S, ns, h.idx, a.cam.pos, primary, h.t, p0, n0);
            const float4 ma = M[2 * h.idx], mb = M[2 * h.idx + 1];
            const V3 emit = scale_r(mk(ma.x, ma.y, ma.z), ma.w) * mk(1.0f, 1.0f, 1.0f);
            split = out.region_slots > 64u && f2u(mb.x) == 2u && a.stream_step_cap >= 3 && emit.x == 0.0f && emit.y == 0.0f && emit.z == 0.0f;
            if (split) {
                V3 ko[2], kd[2], kt[2]; Sfc32 ks[2]; Sfc32 dummy; dummy.a = dummy.b = dummy.c = dummy.counter = 0;
                glass_children(mk(ma.x, ma.y, ma.z), glass_constants(mb.y), p0, n0, primary, mk(1.0f, 1.0f, 1.0f), dummy, ko, kd, kt, ks);
                for (int k = 0; k < 2; ++k) {
                    const V3 ro = k == 0 ? ko[0] : ko[1], rd = k == 0 ? kd[0] : kd[1], rt = k == 0 ? kt[0] : kt[1];
                    const HitSel hc = check_hit(S, ns, np, ro, rd);
                    if (hc.just) {
                        V3 hp, hn;
                        hit_record(S, ns, hc.idx, ro, rd, hc.t, hp, hn);
                        const int e = n_rec++;
                        if (e == 0) { pos[0] = hp; nor[0] = hn; dir[0] = rd; thr[0] = rt; prim[0] = hc.idx; meta[0] = 1u | ((3u + (unsigned int)k) << 8); }
                        else        { pos[1] = hp; nor[1] = hn; dir[1] = rd; thr[1] = rt; prim[1] = hc.idx; meta[1] = 1u | ((3u + (unsigned int)k) << 8); }
                    }
                }
            } else if (out.region_slots == 64u) {
                // streams_pixels_kernel's record: what every first shade of the pixel uses -- the axis and the half-angle scale
                // of its bounce (the same operations on the same inputs, once per pixel) -- in the normal's and direction's place
                V3 axis0; float hk0;
                bounce_axis(mb, n0, primary, axis0, hk0);
                pos[0] = p0; nor[0] = axis0; dir[0] = mk(hk0, 0.0f, 0.0f); thr[0] = mk(1.0f, 1.0f, 1.0f); prim[0] = h.idx; meta[0] = 0u;
                n_rec = 1;
            } else {
                pos[0] = p0; nor[0] = n0; dir[0] = primary; thr[0] = mk(1.0f, 1.0f, 1.0f); prim[0] = h.idx; meta[0] = 0u;
                n_rec = 1;
            }
        }
    }
    const unsigned long long m0 = __ballot(n_rec > 0), m1 = __ballot(n_rec > 1), ms = __ballot(split);
    const unsigned long long missed = __ballot(valid && n_rec == 0);      // updateSeed is all a sample does to these pixels
    const unsigned int c0 = (unsigned int)__builtin_popcountll(m0), c1 = (unsigned int)__builtin_popcountll(m1);
    if (lane == 0) {
        out.counts[region] = c0 + c1;
        out.missed[region] = missed;
        if (ms) {
            atomicAdd(counters + kLvSplitPixels * kCounterStride, (unsigned int)__builtin_popcountll(ms));
            if (2u > __hip_atomic_load(counters + kLvDeepest * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(counters + kLvDeepest * kCounterStride, 2u);       // the children's traceStep
        }
    }
    const unsigned int slot0 = region * out.region_slots;
    for (int e = 0; e < 2; ++e) {
        if (n_rec > e) {
            const unsigned int i = slot0 + (e == 0 ? rank_in(m0) : c0 + rank_in(m1));
            const V3 p = e == 0 ? pos[0] : pos[1], n = e == 0 ? nor[0] : nor[1], dd = e == 0 ? dir[0] : dir[1], tt = e == 0 ? thr[0] : thr[1];
            float4 *r = out.record(i);
            r[0] = float4{p.x, p.y, p.z, n.x};
            r[1] = float4{n.y, n.z, dd.x, dd.y};
            r[2] = float4{dd.z, tt.x, tt.y, tt.z};
            const uint32_t mt = e == 0 ? meta[0] : meta[1], draws = mt >> 8;                  // 0, or 3 / 4 for the children of a glass primary hit
            r[3] = float4{u2f((uint32_t)(e == 0 ? prim[0] : prim[1])), u2f((uint32_t)pixel), u2f(mt), u2f(quad)};
            out.slot_key[i] = (uint32_t)pixel | ((draws ? draws - 2u : 0u) << 30);            // (the stream form holds < 2^30 pixels)
        }
    }
}

// updateSeed (Trace.hs:190-191) for the pixels WITHOUT start hits (their primary ray misses, or both children of their glass
// primary hit do): `draws` draws each.  The pixels with start hits are streams_pixels_kernel's.  Same pixel mapping as the
// primary kernel, whose missed[] masks (one per region) say which lanes have work.
template <bool TILES>
__global__ void __launch_bounds__(256) streams_advance_missed_kernel(const RenderArgs a, const HitList hits, int draws, const unsigned int *tail_start)
{
    if (tail_start && blockIdx.x >= *tail_start) return;      // (the per-pixel tail renders those positions whole)
    unsigned int quad, region; long long pixel;
    const bool valid = primary_pixel<TILES>(a, quad, region, pixel);
    const unsigned long long missed = hits.missed[region];
    if (!valid || !((missed >> (threadIdx.x & 63)) & 1ull)) return;
    Sfc32 sd; sd.a = a.planes.sa[pixel]; sd.b = a.planes.sb[pixel]; sd.c = a.planes.sc[pixel]; sd.counter = a.planes.sctr[pixel];
    for (int j = 0; j < draws; ++j) (void)random_float(sd);
    a.planes.sa[pixel] = sd.a; a.planes.sb[pixel] = sd.b; a.planes.sc[pixel] = sd.c; a.planes.sctr[pixel] = sd.counter;
}

// The seed every ITEM of the split kernel starts from, per record SLOT of the start-hit list and per pass: the pixel's seed after
// pass_first[pass] draws (what that many updateSeeds leave), advanced by the 3 or 4 raw draws the ancestors of a cached glass child made.
// Indexed by slot, not by pixel: a refilling lane knows its slot before it has seen its record, so the record and the snapshot are two
// INDEPENDENT loads (indexed by the pixel the second waited for the first: two memory latencies in a row, in three trips of four), lanes that
// take consecutive records read consecutive snapshots, and the advance is not repeated at every refill.  Reads the pixels' seeds BEFORE
// streams_advance_seeds_kernel moves them.
__global__ void __launch_bounds__(kBlock) streams_slot_seeds_kernel(Planes p, HitList hits, uint4 *snapshots, unsigned int n_slots, int passes, const int *pass_first)
{
    const unsigned int slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n_slots) return;
    const unsigned int region = slot / hits.region_slots;
    if (slot - region * hits.region_slots >= hits.counts[region]) return;
    // (the slot's key word, not its record: rec[13] and rec[14] cost the record's 64-byte line -- 190 MB per 1080p call of the glass scene)
    const uint32_t key = hits.slot_key[slot], pixel = key & 0x3fffffffu, code = key >> 30, draws = code ? code + 2u : 0u;
    Sfc32 s; s.a = p.sa[pixel]; s.b = p.sb[pixel]; s.c = p.sc[pixel]; s.counter = p.sctr[pixel];
    int done = 0;
    for (int k = 0; k < passes; ++k) {
        for (const int upto = pass_first[k]; done < upto; ++done) (void)random_float(s);
        Sfc32 t = s;
        for (uint32_t q = 0; q < draws; ++q) (void)sfc32_next(t);
        snapshots[(size_t)k * n_slots + slot] = uint4{t.a, t.b, t.c, t.counter};
    }
}

// updateSeed (Trace.hs:190-191) for every sample of the call: `draws` draws per held pixel
__global__ void __launch_bounds__(kBlock) streams_advance_seeds_kernel(Planes p, long long n, int draws)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Sfc32 s; s.a = p.sa[i]; s.b = p.sb[i]; s.c = p.sc[i]; s.counter = p.sctr[i];
    for (int k = 0; k < draws; ++k) (void)random_float(s);
    p.sa[i] = s.a; p.sb[i] = s.b; p.sc[i] = s.c; p.sctr[i] = s.counter;
}

}  // namespace

// regions of the start-hit list: one per 8x8 tile (padded as the tiled render kernels pad their grids), or one per 64
// consecutive pixels for images too small for tiles, a multiple of four either way (a workgroup of the primary kernel makes four)
unsigned int streams_regions(int width, int rows_local)
{
    if (tiles_pay_dims(width, rows_local)) return quad_positions(width, rows_local) * 4u;
    const unsigned long long n = (unsigned long long)width * (unsigned long long)rows_local;
    return (unsigned int)((((n + 63ull) / 64ull) + 3ull) & ~3ull);
}

hipError_t launch_streams_primary(const RenderArgs &a, HitList hits, unsigned int *counters, hipStream_t stream)
{
    if (hits.n_regions == 0) return hipSuccess;
    const dim3 g(hits.n_regions / 4u), b(256);
    if (tiles_pay(a)) return launch(streams_primary_kernel<true>, g, b, 0, stream, a, hits, counters);
    else              return launch(streams_primary_kernel<false>, g, b, 0, stream, a, hits, counters);
}

hipError_t launch_streams_advance_missed(const RenderArgs &a, HitList hits, int draws, const unsigned int *tail_start, hipStream_t stream)
{
    if (hits.n_regions == 0 || draws <= 0) return hipSuccess;
    const dim3 g(hits.n_regions / 4u), b(256);
    if (tiles_pay(a)) return launch(streams_advance_missed_kernel<true>, g, b, 0, stream, a, hits, draws, tail_start);
    else              return launch(streams_advance_missed_kernel<false>, g, b, 0, stream, a, hits, draws, tail_start);
}

hipError_t launch_streams_seeds(Planes p, HitList hits, uint4 *snapshots, long long n, int passes, const int *pass_first, int draws, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const unsigned int n_slots = hits.n_regions * hits.region_slots;
    if (n_slots)
        if (hipError_t e = launch(streams_slot_seeds_kernel, dim3(blocks_for(n_slots)), dim3(kBlock), 0, stream, p, hits, snapshots, n_slots, passes, pass_first)) return e;
    return launch(streams_advance_seeds_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, p, n, draws);      // (after it, in stream order)
}

}  // namespace ptmi
