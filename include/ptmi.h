/*
 * ptmi.h -- C ABI of libptmi, the MI355X-native replacement for the one compute
 * call of robbert-vdh/haskell-path-tracer:
 *
 *     dewit = runN (render config) screenPixels            (app/Main.hs:190)
 *     \c (it, acc) -> (it + 1, dewit (scalar c) acc)        (app/Main.hs:191)
 *
 * plus the two array programs that create / replace the RNG planes
 * (`run <$> initialOutput` app/Main.hs:155,306 ; `run <$> reseed ...` :231).
 *
 * The reference has no FFI for this path (the boundary is a Haskell closure,
 * `type CompiledFunction`, app/Main.hs:83-84); these entry points are what a
 * `foreign import ccall` module replacing line 190 binds (INTEGRATION.md shows it).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (PTMI_OK) or a negative
 *     PTMI_E* code; ptmi_last_error(ctx) holds the message.  Nothing throws or aborts.
 *     (The reference surfaces failures as Haskell exceptions out of runN; the Haskell
 *     wrapper turns a non-zero code into throwIO.)
 *   - A RenderResult = Matrix (Color, SFC32) (src/Scene/Objects.hs:36) is SEVEN planes,
 *     row-major [rows][width], x fastest (src/Util.hs:213-214): r, g, b (binary32) and
 *     the SFC32 state a, b, c, counter (uint32) -- Accelerate's struct-of-arrays layout
 *     as seen through A.toVectors (app/Main.hs:350).
 *   - The colour planes hold a SUM over samples; averaging is the presenter's job
 *     (app/assets/fs.glsl:12).
 *   - A context may be entered from any OS thread (app/Main.hs:178-180 forks two bound
 *     threads); calls on one context are serialised by an internal mutex.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_VERSION 600   /* 0.6.0: the chained closure (ptmi_render1_chained, ptmi_chain_*: device residency behind compileFor's pure type), PTMI_ESTALE,
                            * options 14 and 15, PTMI_FORM_PIXEL; ordered passes hand off with release / acquire per region; ptmi_build_id names the
                            * compiled code, not the source text.  0.5.0: ptmi_build_id; ptmi_debug_counters writes 64 words again (as in 0.3) and ptmi_debug_counters_n takes a
                            * capacity; option 13 and ptmi_stream_tickets; the stream form's overflow streams grow instead of dropping children;
                            * no option is read from the environment any more.  (0.4.0: options 8-12, ptmi_stream_schedule.) */

/* ---- error codes ------------------------------------------------------------ */
enum {
    PTMI_OK        =  0,
    PTMI_EINVAL    = -1,   /* bad argument (null pointer, empty scene, size <= 0 ...)        */
    PTMI_ENODEVICE = -2,   /* no HIP device / device index out of range                       */
    PTMI_EHIP      = -3,   /* a HIP runtime call failed; message in ptmi_last_error          */
    PTMI_ENOMEM    = -4,   /* device or host allocation failed                                */
    PTMI_ESTATE    = -5,   /* call order violated (render before set_scene / resize ...)     */
    PTMI_ELIMIT    = -6,   /* scene larger than PTMI_MAX_PRIMITIVES; an image of more than 2^40 pixels (or beyond the stream form's limits) */
    PTMI_ESTALE    = -7    /* a token names no state this context holds (released, consumed, or another context's) */
};

/* data Algorithm = Streams | Inline                       (src/Scene/Trace.hs:68) */
enum { PTMI_STREAMS = 0, PTMI_INLINE = 1 };

/* data Brdf = Matte Float | Glossy Float                 (src/Scene/Objects.hs:77-87)
 * PTMI_GLASS is a build-defined EXTENSION (parameter = index of refraction): a hit spawns a reflection and a
 * refraction ray.  The reference only announces such materials (src/Scene/Trace.hs:109-118, :306-307, :327-328),
 * so GLASS has no reference semantics; it is accepted by PTMI_STREAMS only (Inline cannot split rays,
 * src/Scene/Trace.hs:62-67).  On one GPU a scene with GLASS is rendered by the per-pixel TREE WALK (deterministic, bit-exact
 * against the oracle: one adder per colour word); one PART of a partitioned image at >= 256 samples per call by the stream ("wavefront")
 * form -- BASELINE configs[4]'s path: faster where the job is bounded by its slowest part, colours through float atomics in no defined
 * order.  PTMI_OPT_STREAMS_FORM chooses explicitly.  See DESIGN.md 5.5. */
enum { PTMI_MATTE = 0, PTMI_GLOSSY = 1, PTMI_GLASS = 2 };

#define PTMI_MAX_PRIMITIVES 1024   /* spheres + planes staged in LDS per workgroup */

/* ---- scene types: field order = src/Scene/Objects.hs ------------------------ */
typedef struct ptmi_sphere {       /* data Sphere   Objects.hs:126-131, Material :90-100 */
    float   position[3];
    float   radius;
    float   color[3];
    float   illuminance;
    int32_t brdf_tag;              /* PTMI_MATTE | PTMI_GLOSSY (| PTMI_GLASS, extension) */
    float   brdf_param;
} ptmi_sphere;                     /* 10 words */

typedef struct ptmi_plane {        /* data Plane    Objects.hs:103-108 */
    float   position[3];
    float   direction[3];          /* the normal; used as stored (Intersection.hs:64) */
    float   color[3];
    float   illuminance;
    int32_t brdf_tag;
    float   brdf_param;
} ptmi_plane;                      /* 12 words */

typedef struct ptmi_camera {       /* data Camera   Objects.hs:67-74 */
    float   position[3];
    float   rotation[3];           /* (roll, pitch, yaw) Euler angles */
    int64_t fov;                   /* horizontal field of view, degrees; Haskell Int */
} ptmi_camera;

typedef struct ptmi_stats {
    uint64_t live_bounces;         /* Inline: iterations that took computeRay (Trace.hs:374); Streams: child rays emitted -- since the last reset */
    uint64_t nominal_bounces;      /* pixels x samples x bounce_limit since the last reset                  */
    uint64_t samples;              /* pixels x samples                                                       */
    float    last_render_ms;       /* device time of the last ptmi_render launch(es); 0 unless timing is on */
    uint32_t stream_iterations;    /* Streams: the longest chain of traceSteps any ray lineage took (stream form: of the last sample) */
    uint64_t stream_rays_dropped;  /* Streams with ray splitting: children that found no room -- tree walk: a lane's stack of 16 waiting children full; stream form: the overflow streams full even at 64 rays per pixel (below that the call is redone with longer streams) */
    uint64_t stream_rays_truncated;/* Streams: rays still alive when PTMI_OPT_STREAM_STEP_CAP cut their lineage (the reference has no cap) */
    uint64_t stream_rays_spilled;  /* Streams, stream form: children that found their wave's ring in LDS full and travelled through HBM (nothing is lost) */
    uint64_t stream_rays_overflowed; /* ... of which those that found the wave's spill queue full too and were traced by a later launch (an overflow level) */
} ptmi_stats;

typedef struct ptmi_ctx ptmi_ctx;

/* ---- lifetime ---------------------------------------------------------------- */
int         ptmi_version(void);
/* Which code this binary holds: 16 hex digits of a sha256 over the allocated sections of the objects it was linked from -- host code and
 * the gfx950 code objects (haskell-path-tracer_amd/_build.py: code_id) -- followed by "+<flags>" for a non-default build (ablations,
 * diagnostic builds).  A comment or documentation edit leaves it unchanged; an instruction, a constant or a kernel's name changes it.
 * Every measurement names it (bench.py: binary_build_id; profiles/): a number belongs to the binary that carries the id, and the
 * Python binding refuses a library that does not hold the code the sources beside it compile to.  Never NULL; static storage. */
const char *ptmi_build_id(void);
const char *ptmi_strerror(int code);
/* Create a context on HIP device `device` (>= 0).  Fails with PTMI_ENODEVICE when the
 * machine has no usable GPU: there is NO CPU fallback in this library. */
int         ptmi_create(ptmi_ctx **out, int device);
void        ptmi_destroy(ptmi_ctx *ctx);
/* The message of the calling THREAD's last failed call on `ctx` (ctx NULL: of its last failed ptmi_create); if this thread has not failed on
 * `ctx`, the context's latest message whoever saw it.  The text stays valid until the same thread's next failing call or next ptmi_last_error:
 * threads that share a context never read each other's strings.  A HIP error reported through a return code is taken out of the runtime's
 * own sticky slot (hipGetLastError) -- the caller's next launch check does not find it again -- and an error another library left there
 * is neither mistaken for this library's nor cleared by it.
 * After a failure the context stays usable and consistent: a failed ptmi_set_scene keeps the scene it had; a failed ptmi_resize leaves the
 * context UNSIZED (calls that need planes answer PTMI_ESTATE until a ptmi_resize succeeds; ptmi_group_resize likewise for the group); a
 * render call that fails with PTMI_EHIP / PTMI_ENOMEM half-way leaves the CONTENT of the planes unspecified (re-initialise them), never a
 * dangling pointer.  tests/test_host_sanitized.py walks every failure point under AddressSanitizer. */
const char *ptmi_last_error(const ptmi_ctx *ctx);

/* ---- configuration ----------------------------------------------------------- */
/* mainScene (src/Scene/World.hs:15-77) as run-time data.  Order is kept: checkHit folds
 * over spheres then planes (src/Util.hs:156-158) and ties keep the earlier primitive. */
int ptmi_set_scene(ptmi_ctx *ctx, const ptmi_sphere *spheres, int n_spheres,
                   const ptmi_plane *planes, int n_planes);

/* screenWidth / screenHeight (src/Util.hs:186-188) as run-time values.  Allocates the
 * seven device planes the context owns (for the rows of its partition, see below) and
 * zero-fills them. */
int ptmi_resize(ptmi_ctx *ctx, int width, int height);

/* Row-stripe partition for multi-GPU runs: the image is cut into stripes of `stripe_rows`
 * rows dealt round-robin to `n_parts` contexts; this context is number `part`.  The context
 * then holds ptmi_local_rows() rows, stored contiguously in ascending global order.
 * Must be called before ptmi_resize.  Default: one part (the whole image). */
int ptmi_set_partition(ptmi_ctx *ctx, int stripe_rows, int n_parts, int part);
int ptmi_local_rows(const ptmi_ctx *ctx);                    /* rows held, or a negative code */
int ptmi_global_row(const ptmi_ctx *ctx, int local_row);     /* image row of a held row       */

/* Use caller-owned DEVICE planes (e.g. torch tensors) of ptmi_local_rows() x width elements
 * instead of the context's own; pass all NULL to return to the owned planes. */
int ptmi_bind_planes(ptmi_ctx *ctx, float *r, float *g, float *b,
                     uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr);
/* The device addresses of the seven planes the context currently renders into (its own or the bound ones), for
 * interop: a collective, a graphics-API import, another kernel.  Any pointer argument may be NULL. */
int ptmi_get_planes(ptmi_ctx *ctx, float **r, float **g, float **b,
                    uint32_t **sa, uint32_t **sb, uint32_t **sc, uint32_t **sctr);
/* Launch on this HIP stream (a hipStream_t cast to void*; NULL = the context's own). */
int ptmi_set_stream(ptmi_ctx *ctx, void *hip_stream);
/* Record HIP events around every ptmi_render on the launch stream (ptmi_stats.last_render_ms). */
int ptmi_set_timing(ptmi_ctx *ctx, int enabled);
/* Kernel variant selection for measurements: 0 = default.  See DESIGN.md "kernel variants". */
int ptmi_set_variant(ptmi_ctx *ctx, int variant);

/* Options.  1-5 and 8-13 concern `render Streams` (src/Scene/Trace.hs:141-191) and never change `render Inline`; 6 is a
 * scheduling knob of the per-pixel kernels that changes no result; 7 is a labelled measurement mode of `render Inline`. */
enum {
    /* Which seed a pixel carries out of `combine` (Trace.hs:179-184): the combination function keeps the seed of its
     * FIRST argument, and which of (new value, accumulator element) `permute` hands it first is backend behaviour
     * (assumption A5).  In the reference a pixel receives at most ONE result per traceStep
     * (numNewRays is 0 or 1, Trace.hs:329-331), so either reading is deterministic there.
     *   PTMI_SEED_FROM_RESULT       `f new old`, the order Accelerate's interpreter and LLVM code generators apply the
     *                               function in: every hit replaces the pixel's seed by the seed its ray carried into the
     *                               hit (computeResult, Trace.hs:317-321); updateSeed advances the survivor.  Refused for
     *                               scenes with GLASS: several results per pixel and step race for the seed there.
     *   PTMI_SEED_KEEP_ACCUMULATOR  `f old new`: the pixel keeps its seed through the sample; updateSeed advances it one draw.
     *   PTMI_SEED_AUTO (default)    FROM_RESULT -- the likelier reading -- for scenes in which no primitive splits rays (every
     *                               scene the reference can express); KEEP_ACCUMULATOR for scenes with the build-defined
     *                               GLASS, where the reference defines nothing. */
    PTMI_OPT_STREAMS_SEED_RULE = 1,
    /* Safety cap on the traceSteps of one ray lineage (children inherit their ancestors' count).  The reference has NO
     * bound (notFinished never stops a non-empty stream, Trace.hs:166-170); the default, 65536, only guarantees
     * termination, for both forms.  Rays it cuts are counted in ptmi_stats.stream_rays_truncated. */
    PTMI_OPT_STREAM_STEP_CAP = 2,
    /* Stream form only: how many rays per pixel the overflow streams hold to begin with (default 4).  (Children wait in their wave's ring in
     * LDS, then in its spill queue; the overflow streams are the last resort and practically unused: ptmi_stats.stream_rays_overflowed.)
     * The reference's vectors grow as needed (expand, Trace.hs:284-293), and since 0.5 so do these: a call that WOULD drop children puts the
     * colour planes back (they are copied aside before the launch), doubles the streams and runs again, up to 64 rays per pixel or what the
     * device's memory allows; only beyond that are children dropped and counted in ptmi_stats.stream_rays_dropped.  ptmi_get_option returns
     * the capacity in force (what the context has grown to); setting the option starts over from the value given. */
    PTMI_OPT_STREAM_CAPACITY = 3,
    /* PTMI_FORM_PIXEL: the per-pixel kernels (one lane walks its pixel's rays; with GLASS: its ray trees) -- deterministic, bit-exact against the
     * oracle also with GLASS (a colour word has one adder).
     * PTMI_FORM_STREAM: the stream ("wavefront") form -- the start hits of the pixels as a compacted list, persistent waves whose
     * lanes take items from it by ballot + prefix, refraction children compacted into a ring per wave; ONE launch per call.  With GLASS a
     * pixel's contributions are added through float atomics in no defined order (as Accelerate's permute): the same additions in another
     * association -- colours agree with the per-pixel kernels to ~1e-4 relative (1.4e-4 seen at 256 spp), the RNG planes exactly.
     * PTMI_FORM_AUTO (default): PIXEL -- except for a scene with GLASS on a PARTITIONED context (ptmi_set_partition with more than one part:
     * one rank of a multi-GPU job) at 256 samples per call or more, where it is STREAM: there the job is as fast as its slowest part, and the
     * stream form is 5 % faster on that part and evens the parts out (BASELINE configs[4]: 29.1 against 30.7 ms on the bounding part, imbalance
     * 1.02 against 1.03-1.14; at 1080p / 64 spp on one GPU it is 1 % slower, so nothing changes there).  A caller who wants the tree walk's
     * bit-exactness on such a part sets PTMI_FORM_PIXEL. */
    PTMI_OPT_STREAMS_FORM = 4,
    /* Stream form only: how many samples of a pixel make one item.  0 (default) = automatic: without GLASS a pixel's whole sample
     * chain stays in one lane, in order (bit-identical to the per-pixel kernels; few long items are cut into ORDERED passes, still
     * bit-identical); with GLASS, where the order of a pixel's additions is undefined anyway, items of a few dozen samples run in
     * whatever lanes take them.  k > 0: items of k samples -- under PTMI_SEED_FROM_RESULT (a pixel's samples are one serial chain)
     * as ordered passes, bit-identical; under PTMI_SEED_KEEP_ACCUMULATOR as unordered items: colours then agree to rounding only
     * (Accelerate's `permute` does not define the order either); the RNG planes stay exact.  A launch holds at most 64 passes: where
     * n_spp / k exceeds that, the samples are split evenly into 64 (ordered passes) or k is raised to n_spp / 64 (unordered items); with
     * PTMI_OPT_STREAM_GRADED (the default) k is the LARGEST item of the unordered form -- the last items of a launch are shorter. */
    PTMI_OPT_STREAM_BATCH = 5,
    /* The per-pixel kernels of both algorithms, a scheduling knob that changes no result: a launch of few pixels and
     * many samples (one part of a multi-GPU image) is cut into this many chained copies of the tile grid, each rendering
     * a slice of the samples, so that the end of the launch does not run on a partly empty chip.
     * 0 (default) = automatic, 1 = off, k = k copies. */
    PTMI_OPT_SPP_CHUNKS = 6,
    /* A MEASUREMENT mode of `render Inline`, never the default: PTMI_ARITH_CONTRACTED runs the same kernel compiled with
     * a * b + c contracted into fused multiply-adds (dot and cross products, the rotation, the quaternion) -- what a compiler
     * with -ffp-contract=fast, or Accelerate's fast-math LLVM backends, may do to the reference's source.  Its planes are NOT
     * the oracle's (the RNG planes still are: integer arithmetic); DESIGN.md reports how far they are and what the literal
     * reading -- every operation rounded on its own, PTMI_ARITH_EXACT -- costs.  Variants and Streams ignore it. */
    PTMI_OPT_ARITHMETIC = 7,
    /* 8-13: scheduling knobs of the stream form of Streams.  None changes a ray, a seed or (without GLASS) a bit of the planes.
     *
     * PTMI_OPT_STREAM_TAIL: scenes without GLASS, one-pass launches: the cheapest quads of the dispatch order -- the cheapest classes
     * that together hold at most this many THOUSANDTHS of the recorded cost -- are rendered by the per-pixel chain kernel on a
     * low-priority stream beside the persistent launch, whose waves' slots it fills as they end.  -1 (default) = automatic (150, or 400
     * where a lane sees fewer than four pixels), 0 = no tail. */
    PTMI_OPT_STREAM_TAIL = 8,
    /* PTMI_OPT_ORDERED_PASSES: scenes without GLASS: a pixel's samples cut into this many ORDERED passes inside the one launch; the pixel's
     * seven words are handed from the lane that rendered pass p to whichever lane -- of any wave, on any XCD -- takes pass p + 1
     * (PTMI_OPT_PASS_HANDOFF says how).  0 (default) = automatic: passes only for parts of an image at >= 256 spp, where a lane sees fewer
     * than three pixels and they are worth 5-8 % -- and only with the FENCED hand-off; 1 = OFF: one pass per launch, no hand-off between
     * waves at all; k in [2, 64] = k passes.  1 also overrides PTMI_OPT_STREAM_BATCH for scenes without GLASS.  (Until 0.4 the environment
     * variables PTMI_ORDERED_PASSES / PTMI_STREAM_TAIL overrode these two options at creation; nothing is read from the environment any more.) */
    PTMI_OPT_ORDERED_PASSES = 9,
    /* PTMI_OPT_GLASS_BATCH: scenes with GLASS: a GLASS hit waits in its lane until this many lanes of its wave hold one (or the wave has
     * nothing else to shade or trace), so that the refraction block runs for that many lanes at a time.  0 (default) = automatic,
     * 1 = off, k in [2, 64]. */
    PTMI_OPT_GLASS_BATCH = 10,
    /* PTMI_OPT_STREAM_GRADED: scenes with GLASS (or unordered items): 1 (default) = a pixel's samples are cut into GRADED passes -- long
     * items first, short ones (4 samples where the long ones hold 16 or more) last (ptmi_stream_schedule) -- so that the launch does not end with long items in few lanes;
     * 0 = uniform passes (round 3). */
    PTMI_OPT_STREAM_GRADED = 11,
    /* PTMI_OPT_SNAPSHOT_BUDGET_MB: the stream form's split kernel starts every pass of every start hit from a 16-byte seed snapshot
     * (passes x record slots x 16 bytes of device memory: 66 MB per pass at 1080p).  This caps that block: where the schedule's passes
     * would need more, its LAST passes are merged (longer items at the end of the launch; no seed and no ray changes), and a call whose
     * single pass still does not fit fails with PTMI_ELIMIT.  0 (default) = an eighth of the device's memory; otherwise megabytes
     * in [1, 2^20]. */
    PTMI_OPT_SNAPSHOT_BUDGET_MB = 12,
    /* PTMI_OPT_STREAM_PASS_GROUPS: in which order the split kernel hands out its items.  Pass by pass -- every region of the start-hit list in pass
     * 0, then every region in pass 1 ... -- a region's 64-byte records and the colour lines of its pixels come from HBM once per pass.  In GROUPS of
     * consecutive passes, each group region by region (region r in every pass of the group, then region r + 1), the items of a start hit that belong
     * to one group are taken within microseconds of each other from one ticket queue, by waves behind one L2, and only the first of them reads HBM.
     * Changes no ray and no seed, only which lane renders which item when.  0 (default) = automatic: a group is a run of passes of EQUAL size (16,
     * 16, 16 | 8 | 4, 4 at 1080p / 64 spp) -- the grading of the passes, which keeps the end of the launch short, is then untouched; 1 = every pass on
     * its own (round 4); k in [2, 64] = the LAST k passes as one group; 100 + g (g in [2, 64]) = groups of g passes all the way.
     * ptmi_stream_tickets is the order as a pure function. */
    PTMI_OPT_STREAM_PASS_GROUPS = 13,
    /* PTMI_OPT_CHAIN_SLOTS: how many states of the chained closure (ptmi_render1_chained below) may stay on the DEVICE at a time; the oldest one
     * beyond that moves to host memory the library owns (nothing is lost; ptmi_chain_fetch serves it from there).  0 (default) = automatic: what
     * a sixteenth of the device's memory holds, at least 3, at most 64; otherwise k in [2, 4096]. */
    PTMI_OPT_CHAIN_SLOTS = 14,
    /* PTMI_OPT_PASS_HANDOFF: how ordered passes (PTMI_OPT_ORDERED_PASSES) hand a pixel's words from wave to wave.
     * PTMI_HANDOFF_FENCED (default): what the HSA memory model promises -- the storing wave's vmcnt(0), an agent-scope RELEASE, the region's
     * counter; the taking wave's poll, an agent-scope ACQUIRE, its loads -- paid once per (region, pass), not per item: all items of a region's
     * pass run in the one wave that drew its ticket, and the lane that ends the last of them releases for all.
     * PTMI_HANDOFF_FENCE_FREE: rounds 3-5's form -- write-through (sc1) stores, a counter that moves after the storing wave's vmcnt(0), sc1
     * loads after the poll, NO fence: the "valid form" of MI355X_MICROARCH.md, MEASURED valid on gfx950 (5 billion hand-offs compared bit for
     * bit, profiles/r05_soak_ordered_passes.json) -- not a promise of the memory model.  Never chosen automatically: with this value
     * PTMI_OPT_ORDERED_PASSES = 0 means one pass. */
    PTMI_OPT_PASS_HANDOFF = 15
};
enum { PTMI_HANDOFF_FENCED = 0, PTMI_HANDOFF_FENCE_FREE = 1 };
enum { PTMI_ARITH_EXACT = 0, PTMI_ARITH_CONTRACTED = 1 };
enum { PTMI_SEED_KEEP_ACCUMULATOR = 0, PTMI_SEED_FROM_RESULT = 1, PTMI_SEED_AUTO = 2 };
enum { PTMI_FORM_AUTO = 0, PTMI_FORM_STREAM = 1, PTMI_FORM_PIXEL = 2 };
int ptmi_set_option(ptmi_ctx *ctx, int option, int64_t value);
/* The passes the stream form's split kernel (GLASS, or PTMI_OPT_STREAM_BATCH under PTMI_SEED_KEEP_ACCUMULATOR) cuts n_spp samples into
 * for n_pixels held pixels on `lanes` persistent lanes (64 x 4 x 6 x compute units): pass p renders samples [first[p], first[p + 1]).
 * batch = PTMI_OPT_STREAM_BATCH, graded = PTMI_OPT_STREAM_GRADED.  Pure host arithmetic (no device needed).  Returns the number of
 * passes (first[] receives passes + 1 entries), PTMI_ELIMIT if `capacity` entries do not hold them (65 always do). */
int ptmi_stream_schedule(int n_spp, uint64_t n_pixels, uint64_t lanes, int batch, int graded, int32_t *first, int capacity);
/* The schedule of the cost-ordered dispatch, host arithmetic: given the launches made so far with one (camera, scene,
 * shape, limit, algorithm), does the next launch rebuild the order from the recorded costs (before launch 1, 2, 4, 8, ...; never once
 * the recording limit -- 2^20 launches, 2^11 for the stream form -- is reached) and does it record its costs (only if the launch after it
 * rebuilds: launch 0, 1, 3, 7, 15, ...)?  Returns the state after it. */
int ptmi_order_schedule(int launches, int stream_form, int *rebuild, int *record);
/* The order in which ONE of the eight ticket queues of the split kernel hands out its passes x queue_regions items under
 * PTMI_OPT_STREAM_PASS_GROUPS = option, for the schedule first[0 .. passes] of ptmi_stream_schedule: ticket j is (pass_out[j], region_out[j]),
 * region = the region's index within the queue.  Every (pass, region) pair appears exactly once.  Pure host arithmetic.  Returns the number of
 * tickets, PTMI_ELIMIT if `capacity` entries do not hold them. */
int ptmi_stream_tickets(int option, const int32_t *first, int passes, int queue_regions, int32_t *pass_out, int32_t *region_out, int capacity);
int ptmi_get_option(ptmi_ctx *ctx, int option, int64_t *value);

/* ---- state: initialOutput / genSeeds / reseed -------------------------------- */
/* initialOutput (src/Util.hs:204-205): colour := 0, RNG := genSeeds.  genSeeds draws three
 * words per pixel from OS entropy (src/Util.hs:122-127); here they come from a counter-based
 * hash of (seed0, global pixel index) so that "the same RNG seed" is meaningful, then go
 * through createWith (3-word sfc32 seeding).  Runs on the device. */
int ptmi_init_output(ptmi_ctx *ctx, uint64_t seed0);
/* reseed (src/Util.hs:134-135): keep colour, replace every RNG state. */
int ptmi_reseed(ptmi_ctx *ctx, uint64_t seed0);
/* createWith . use (src/Util.hs:125): three host word planes -> RNG states (colour untouched). */
int ptmi_create_with(ptmi_ctx *ctx, const uint32_t *w0, const uint32_t *w1, const uint32_t *w2);
/* Host <-> device copies of the held rows (synchronous). NULL plane pointers are skipped. */
int ptmi_upload_state(ptmi_ctx *ctx, const float *r, const float *g, const float *b,
                      const uint32_t *sa, const uint32_t *sb, const uint32_t *sc, const uint32_t *sctr);
int ptmi_download_state(ptmi_ctx *ctx, float *r, float *g, float *b,
                        uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr);
int ptmi_download_color(ptmi_ctx *ctx, float *r, float *g, float *b);   /* what graphicsLoop reads, Main.hs:350 */

/* ---- the hot path ------------------------------------------------------------- */
/* Resident form: equivalent to n_spp successive applications of
 *     render algorithm screenPixels camera            (src/Scene/Trace.hs:135-200)
 * to the planes held by the context.  `bounce_limit` is the `15` of Trace.hs:200 /
 * maxIterations (:80-81) made a parameter; PTMI_STREAMS IGNORES it, as the reference's Streams does (a non-empty
 * stream is never stopped by the iteration count, Trace.hs:166-170) -- ptmi_stats.nominal_bounces is therefore 0 for
 * Streams launches and live_bounces counts the child rays they emitted.  Asynchronous on the launch stream. */
int ptmi_render(ptmi_ctx *ctx, const ptmi_camera *camera, int algorithm,
                int bounce_limit, int n_spp);
int ptmi_synchronize(ptmi_ctx *ctx);
/* 1 if ptmi_render with `algorithm` would wait for device work before it returns under the context's present scene and options
 * (the stream form of Streams with GLASS, or with PTMI_OPT_STREAM_BATCH, reads its overflow counters back), 0 if it only enqueues. */
int ptmi_render_blocks(ptmi_ctx *ctx, int algorithm);

/* Compatibility form = exactly one call of the closure built by compileFor: host planes in,
 * host planes out, one sample.  screen_x / screen_y are the two Int planes of the
 * Matrix (V2 Int) argument (src/Util.hs:209-210) or NULL for the implicit x = column,
 * y = row.  Pointers are borrowed for the duration of the call; in and out may alias.
 * Ignores any partition: the whole width x height image is rendered. */
int ptmi_render1(ptmi_ctx *ctx, const ptmi_camera *camera, int algorithm, int bounce_limit,
                 int width, int height,
                 const int64_t *screen_x, const int64_t *screen_y,
                 const float *r_in, const float *g_in, const float *b_in,
                 const uint32_t *sa_in, const uint32_t *sb_in, const uint32_t *sc_in, const uint32_t *sctr_in,
                 float *r_out, float *g_out, float *b_out,
                 uint32_t *sa_out, uint32_t *sb_out, uint32_t *sc_out, uint32_t *sctr_out);

/* ---- the closure, chained: device residency behind compileFor's pure type ------------------------------------------------
 * `dewit = runN (render config) screenPixels` (app/Main.hs:190) is a pure function of arrays, and on the reference's own GPU backend
 * those arrays LIVE ON THE DEVICE: runN leaves its result there and copies it to the host lazily, when somebody reads it
 * (graphicsLoop, app/Main.hs:350), and an argument that is a previous result is found on the device again and not uploaded.
 * ptmi_render1 above moves all seven planes both ways on every call (1.1 ms per sample at 800x600 around a 0.03-ms kernel);
 * these entry points give the closure runN's behaviour.  A STATE is one RenderResult (seven planes, width x height) held by the
 * context under a TOKEN (never 0, never reused, meaningless to other contexts).  To the caller a state is an immutable value:
 *   - ptmi_render1_chained(token_in) renders ONE sample from the state token_in names into a NEW state and returns its token; the
 *     input state stands as it was, for ptmi_chain_fetch or for another call (rendering twice from one token gives the same planes
 *     twice -- the closure is a function).  Nothing crosses PCIe; the call only enqueues.
 *   - if token_in names no state the context holds (0; a released token; a RenderResult that came from elsewhere), the seven host
 *     planes r_in .. sctr_in are uploaded instead -- ptmi_render1's copy path; if they are NULL too: PTMI_ESTALE.
 *   - planes_out: each may be NULL; the ones given are downloaded at once (the call then waits for the render).  A lazy consumer
 *     passes none and calls ptmi_chain_fetch when somebody reads a plane: colour for graphicsLoop, all seven on demand.
 *   - a state is held until ptmi_chain_release(token) (a Haskell finalizer, a C++ destructor), or until a call CONSUMES it
 *     (PTMI_CHAIN_CONSUME: the caller gives up token_in with the call; the sample is then rendered in place, without the
 *     device-to-device copy that otherwise keeps the input intact -- the cost of a resident ptmi_render; should the call then fail, the
 *     input is gone with it).  At most
 *     PTMI_OPT_CHAIN_SLOTS states stay on the device; older ones move to host memory the library owns and are served from there.
 * Any thread; calls are serialised by the context's mutex like all others.  The resident planes of ptmi_resize / ptmi_render and
 * their partition are a separate matter and untouched. */
enum { PTMI_CHAIN_CONSUME = 1 };
int ptmi_render1_chained(ptmi_ctx *ctx, const ptmi_camera *camera, int algorithm, int bounce_limit, int width, int height,
                         uint64_t token_in, int flags,
                         const float *r_in, const float *g_in, const float *b_in,
                         const uint32_t *sa_in, const uint32_t *sb_in, const uint32_t *sc_in, const uint32_t *sctr_in,
                         uint64_t *token_out,
                         float *r_out, float *g_out, float *b_out,
                         uint32_t *sa_out, uint32_t *sb_out, uint32_t *sc_out, uint32_t *sctr_out);
/* run <$> initialOutput (src/Util.hs:204-205; app/Main.hs:155, :306) as a state: colour 0, RNG from seed0 (ptmi_init_output's seeding). */
int ptmi_chain_init_output(ptmi_ctx *ctx, int width, int height, uint64_t seed0, uint64_t *token_out);
/* run <$> reseed acc (src/Util.hs:134-135; app/Main.hs:231): a NEW state with token_in's colour and fresh RNG states from seed0.  If
 * token_in is not held the three colour planes r_in, g_in, b_in (host) are uploaded instead.  flags: PTMI_CHAIN_CONSUME as above. */
int ptmi_chain_reseed(ptmi_ctx *ctx, uint64_t seed0, int width, int height, uint64_t token_in, int flags,
                      const float *r_in, const float *g_in, const float *b_in, uint64_t *token_out);
/* The planes of a held state into host memory; NULL pointers are skipped.  Waits for the renders that produce the state. */
int ptmi_chain_fetch(ptmi_ctx *ctx, uint64_t token, float *r, float *g, float *b,
                     uint32_t *sa, uint32_t *sb, uint32_t *sc, uint32_t *sctr);
/* The caller will not name this token again.  Releasing a token that is not held (already released or consumed) is PTMI_OK. */
int ptmi_chain_release(ptmi_ctx *ctx, uint64_t token);
typedef struct ptmi_chain_stats {
    uint32_t states_on_device, states_on_host;    /* held now */
    uint32_t device_slots;                         /* PTMI_OPT_CHAIN_SLOTS in force */
    uint32_t width, height;                        /* of the newest state (0 when none is held) */
    uint64_t renders_chained;                      /* ptmi_render1_chained / ptmi_chain_reseed calls that found their input on the device ... */
    uint64_t renders_in_place;                     /* ... of which PTMI_CHAIN_CONSUME spared the copy                                         */
    uint64_t renders_uploaded;                     /* calls that took the copy path (host planes in)                                           */
    uint64_t evictions;                            /* states moved to host memory to make room                                                 */
    uint64_t fetches;                              /* ptmi_chain_fetch calls and planes_out downloads                                          */
} ptmi_chain_stats;
int ptmi_chain_info(ptmi_ctx *ctx, ptmi_chain_stats *out);

/* ---- present (what graphicsLoop + fs.glsl do with the result) ------------------- */
/* graphicsLoop interleaves the three colour planes (`V.zipWith3 V3 r g b`, app/Main.hs:351), uploads them
 * as an RGB32F texture (:383-393) and the fragment shader shows `texture.rgb / u_iterations`
 * (app/assets/fs.glsl:12) into an 8-bit framebuffer.  This does the same on the device for the held rows:
 *   rgb32f_out : [rows][W][3] float, colour / float(iterations)            (may be NULL)
 *   rgba8_out  : [rows][W][4] bytes, round(clamp(colour / iterations, 0, 1) * 255), alpha 255   (may be NULL)
 * The text overlay of the iteration counter (fs.glsl:15) is UI and not reproduced. */
int ptmi_present(ptmi_ctx *ctx, int iterations, float *rgb32f_out, uint8_t *rgba8_out);

/* The three colour planes of the held rows, copied device-to-device into ONE contiguous buffer [3][rows][W] on the
 * context's device, ordered behind the renders already issued; the copy runs on `hip_stream` (a hipStream_t cast to
 * void*; NULL = the context's launch stream), so a consumer (a collective, a display interop) can take it from there
 * while the next render runs.  Asynchronous. */
int ptmi_snapshot_color(ptmi_ctx *ctx, float *dst_device, void *hip_stream);

int ptmi_get_stats(ptmi_ctx *ctx, ptmi_stats *out);   /* synchronises the launch stream */
int ptmi_reset_stats(ptmi_ctx *ctx);
/* Diagnostics: the raw device counters (hand-out counter in [0]; the diagnostic builds of the kernels --
 * -DPTMI_PHASE_STATS and the others listed in csrc/ptmi_diag.h -- add their statistics, see tools/phase_stats.py).
 * ptmi_debug_counters writes the first 64 words (its contract since 0.3; 0.4 wrote 256 into the same argument and overran a
 * caller built against 0.3); ptmi_debug_counters_n writes min(capacity, 256) words and returns how many, or a negative code.
 * A caller built against 0.4 that passes a 256-word buffer to ptmi_debug_counters gets PTMI_OK and words 64 to 255 LEFT AS THEY WERE:
 * it must move to ptmi_debug_counters_n to see the diagnostic builds' statistics.  Both synchronise the launch stream. */
int ptmi_debug_counters(ptmi_ctx *ctx, uint32_t out[64]);
int ptmi_debug_counters_n(ptmi_ctx *ctx, uint32_t *out, int capacity);

/* ---- groups: the GPUs of one node behind ONE host process ------------------------ */
/* The reference's host is a single process holding one compiled function (app/Main.hs:188-191); a group lets that
 * process use several GPUs: the image is cut into stripes of `stripe_rows` rows (0 = 8) dealt round-robin to the
 * devices (ptmi_set_partition), every member renders its rows without any exchange (seeds come from the global pixel
 * index: the stitched planes equal the single-device image bit for bit), and the one exchange is the read-out of the
 * colour planes -- to host planes for graphicsLoop (app/Main.hs:346-351), or to a root device over RCCL / xGMI.
 * (SURVEY.md 8b proposed ptmi_create(out, device_first, n_devices); a group of contexts keeps ptmi_create unchanged.) */
typedef struct ptmi_group ptmi_group;
int         ptmi_group_create(ptmi_group **out, const int *devices, int n_devices, int stripe_rows);
void        ptmi_group_destroy(ptmi_group *group);
int         ptmi_group_size(const ptmi_group *group);
ptmi_ctx   *ptmi_group_member(ptmi_group *group, int i);        /* member i's context (options, statistics, planes) */
const char *ptmi_group_last_error(const ptmi_group *group);      /* per thread, as ptmi_last_error */
int ptmi_group_set_scene(ptmi_group *group, const ptmi_sphere *spheres, int n_spheres, const ptmi_plane *planes, int n_planes);
int ptmi_group_resize(ptmi_group *group, int width, int height);
int ptmi_group_init_output(ptmi_group *group, uint64_t seed0);
int ptmi_group_reseed(ptmi_group *group, uint64_t seed0);
/* ptmi_set_option / ptmi_set_variant on every member (ptmi_group_member gives access to a single one). */
int ptmi_group_set_option(ptmi_group *group, int option, int64_t value);
int ptmi_group_set_variant(ptmi_group *group, int variant);
/* ptmi_render on every member; asynchronous: all devices are busy before the call returns. */
int ptmi_group_render(ptmi_group *group, const ptmi_camera *camera, int algorithm, int bounce_limit, int n_spp);
int ptmi_group_synchronize(ptmi_group *group);
/* The whole image's colour planes [height][width] into HOST planes: every member's rows come down through its own
 * copy path, all members at once, and the stripes are stitched in place.  Synchronous. */
int ptmi_group_download_color(ptmi_group *group, float *r, float *g, float *b);
/* The whole image's colour planes into DEVICE planes [height][width] on member `root`'s device: RCCL grouped
 * ncclSend / ncclRecv over xGMI (every peer has its own link to the root), then a stitch kernel.  Synchronous.  The
 * caller's current HIP device is left as it was.  (What has run: on real RCCL, one member sending to itself; with 3 and 8 members
 * sharing one device, the whole n > 1 call sequence -- every root, unequal row counts, members without rows, communicator reuse, an
 * error inside the group -- against a test-only stand-in for librccl that turns paired sends and receives into device copies
 * (tests/test_gpu_group_rccl_stub.py: the host logic).  Between PHYSICAL devices, over xGMI, it has never run: no box this build has
 * seen holds two GPUs; tests/test_group.py arms itself on the first one that does.) */
int ptmi_group_gather_color(ptmi_group *group, int root, float *r_device, float *g_device, float *b_device);
int ptmi_group_get_stats(ptmi_group *group, ptmi_stats *sum);  /* sums (maxima for the time and step fields) over the members */
/* The partition's arithmetic, usable without a device: rows part `part` holds, and the image row of one of them. */
int ptmi_partition_rows(int height, int stripe_rows, int n_parts, int part);
int ptmi_partition_global_row(int height, int stripe_rows, int n_parts, int part, int local_row);

/* ---- point queries (the reference's unit-test surface) ------------------------ */
/* Evaluates distanceTo / hit (src/Scene/Intersection.hs:16-64) on the DEVICE for n independent
 * cases -- ray i against primitive i -- the way test/Scene/Intersection/Tests.hs:122-123
 * evaluates single expressions through the backend.  rays: n x 6 floats (origin, direction).
 * Outputs (host): is_just[n], t[n], hit_normalp[n x 6] (position, normal); the last may be NULL. */
int ptmi_eval_distance_to_sphere(ptmi_ctx *ctx, const ptmi_sphere *spheres, const float *rays, int n,
                                 int32_t *is_just, float *t, float *hit_normalp);
int ptmi_eval_distance_to_plane(ptmi_ctx *ctx, const ptmi_plane *planes, const float *rays, int n,
                                int32_t *is_just, float *t, float *hit_normalp);
/* sin/cos of the device math used by anglesToQuaternion, for pinning against libm. */
int ptmi_eval_sincos(ptmi_ctx *ctx, const float *x, int n, float *sin_out, float *cos_out);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
