// ptmi_kernels.h -- launch interface between the C ABI (ptmi_api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptmi_core.h"

namespace ptmi {

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
    unsigned long long *live_counter;     // device counter, += live bounces
    unsigned int *work_counter;           // device counter for dynamic pixel hand-out (variants)
    unsigned int *stream_iterations;      // Streams: steps taken by the last sample (max over waves)
    // Cost-ordered dispatch of the tiled kernels (see lane_pixel in ptmi_kernels.hip).  A "quad" is a run of four
    // x-adjacent 8x8 tiles.  quad_order: the quad each dispatch position works on (NULL = image order).
    // quad_cost: where each wave adds the loop trips it paid (NULL = do not record).
    const unsigned int *quad_order;
    unsigned int *quad_cost;
};

// Ray stream of the wavefront Streams path: struct-of-arrays, `capacity` rays (type RayState, Trace.hs:46)
struct RayQueue {
    float *f[9];            // origin xyz, direction xyz, throughput xyz
    uint32_t *pixel;        // local pixel index
    uint32_t *seed[4];      // SFC32 a, b, c, counter
    uint32_t *depth;        // traceSteps already taken by the ray's ancestors (= the awhile iteration it belongs to)
    unsigned int capacity;  // total; split into kStreamShards equal regions, each with its own length counter
};
constexpr int kRayQueueWords = 15;
constexpr int kStreamShards = 8;            // one append counter per shard: a single counter word serves ~90 requests/us
constexpr int kCounterStride = 32;          // the shard counters sit 128 B apart (one per cache line)
constexpr int kStreamStepCap = 64;          // traceSteps per ray lineage; the reference has no bound (Trace.hs:166-170)
// device counters of one step launch, each kCounterStride words apart:
//   [0, 8) next-stream length per shard | 8 dropped children | [9, 17) rays continued or emitted per shard | [17, 25) deepest step + 1
constexpr int kStreamCounters = 3 * kStreamShards + 1;
constexpr int kCtrDropped = kStreamShards, kCtrLive = kStreamShards + 1, kCtrDeepest = 2 * kStreamShards + 1;
struct StreamLayout { unsigned int prefix[kStreamShards + 1]; };   // ray i of the input lives in shard k: prefix[k] <= i < prefix[k+1]

hipError_t launch_streams_init(const RenderArgs &a, RayQueue q, int batch, hipStream_t stream);
hipError_t launch_streams_step(const RenderArgs &a, RayQueue in, StreamLayout layout, RayQueue out,
                               unsigned int *counters, hipStream_t stream);
hipError_t launch_streams_update_seed(Planes p, long long n, int draws, hipStream_t stream);

hipError_t launch_render_inline(const RenderArgs &a, int variant, hipStream_t stream);
hipError_t launch_render_streams(const RenderArgs &a, int variant, hipStream_t stream);
unsigned int quad_positions(int width, int rows_local);                    // entries of quad_order / quad_cost (0 = tiles not used)
hipError_t launch_quad_order(const unsigned int *cost, unsigned int *order, unsigned int *cls, unsigned int n, hipStream_t stream);
bool uses_quad_order(const RenderArgs &a, int algorithm_inline, int variant);
hipError_t launch_seed(Planes p, int width, int rows_local, int stripe_rows, int n_parts, int part,
                       uint64_t seed0, bool clear_color, hipStream_t stream);
hipError_t launch_create_with(Planes p, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2,
                              int64_t n, hipStream_t stream);
hipError_t launch_eval_sphere(const float *spheres10, const float *rays, int n,
                              int32_t *is_just, float *t, float *normalp, hipStream_t stream);
hipError_t launch_eval_plane(const float *planes12, const float *rays, int n,
                             int32_t *is_just, float *t, float *normalp, hipStream_t stream);
hipError_t launch_present(Planes p, long long n, int iterations, float *rgb, uint32_t *rgba, hipStream_t stream);
hipError_t launch_eval_sincos(const float *x, int n, float *s, float *c, hipStream_t stream);

}  // namespace ptmi
