// ptmi_diag.h -- the diagnostic builds of the render kernels, in ONE place.  A kernel calls a probe at the sites a measurement
// needs (`phase.trip()`, `probe.shade(alive, glass)`, ...); without the probe's macro every method is empty and STATIC and the probe
// holds nothing, so the product's code is the code without them, instruction for instruction (tools/isa_diff.py compares the ISA; a
// non-static empty method -- a `this` that is never used -- was enough to move a few register assignments).  The macros, each set by the tool
// that reads the counters back through ptmi_debug_counters (RenderArgs.work_counter):
//   -DPTMI_PHASE_STATS   render_inline_kernel: lane participation and wave cycles per round       tools/phase_stats.py
//   -DPTMI_SPHERE_STATS  check_hit: sphere tests that take the square-root path                   tools/phase_stats.py
//   -DPTMI_TREE_STATS    render_streams_tree_kernel: lane participation, the wave's tail         tools/tree_stats.py
//   -DPTMI_TREE_STATS_MAP   ... the red plane becomes the per-pixel cost map                     tools/tree_cost_map.py
//   -DPTMI_TAIL_STATS    streams_pixels_kernel: when waves end, lanes with an item per trip      tools/tail_stats.py
//   -DPTMI_TAIL_PHASES      ... cycles in the refill block and in the trips that end an item
//   -DPTMI_SPLIT_STATS   streams_split_kernel: lane participation per round, wave durations      tools/split_stats.py
//   -DPTMI_SPLIT_ENDS       ... only when its waves end (two atomics per wave)                    tools/split_stats.py ends
// (-DPTMI_POOL_STATS belongs to an ablation kernel and lives beside it, ptmi_inline_ablations.hip.)
#pragma once

#include <hip/hip_runtime.h>

namespace ptmi {
namespace {
namespace diag {

#define PTMI_PROBE __device__ __forceinline__
#ifndef PTMI_SPLIT_HIST_SHIFT
#define PTMI_SPLIT_HIST_SHIFT 14       // bins of 164 us
#endif
#ifndef PTMI_SPLIT_HIST_FIRST
#define PTMI_SPLIT_HIST_FIRST 32ull    // the first bin of the 64-bin window of end times
#endif

PTMI_PROBE unsigned int lanes(bool on) { return (unsigned int)__builtin_popcountll(__ballot(on)); }
PTMI_PROBE bool first_active_lane() { return (threadIdx.x & 63) == (int)__builtin_ctzll(__ballot(1)); }
PTMI_PROBE unsigned int wave_max(unsigned int v)
{
    for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// ---- check_hit: [16] sphere tests (per wave), [17] ... that took the square-root path, [18] candidate lanes, [19] active lanes,
// [20] plane tests (per wave), [21] ... that took the division path
#ifdef PTMI_SPHERE_STATS
PTMI_PROBE unsigned int *sphere_counters(unsigned int *work_counter) { return work_counter; }
PTMI_PROBE void sphere_test(unsigned int *wc, bool cand)
{
    const unsigned long long cm = __ballot(cand), am = __ballot(1);
    if (wc && (threadIdx.x & 63) == (int)__builtin_ctzll(am)) {
        atomicAdd(wc + 16, 1u);
        if (cm) atomicAdd(wc + 17, 1u);
        atomicAdd(wc + 18, (unsigned int)__builtin_popcountll(cm));
        atomicAdd(wc + 19, (unsigned int)__builtin_popcountll(am));
    }
}
PTMI_PROBE void plane_test(unsigned int *wc, bool cand)
{
    const unsigned long long cm = __ballot(cand), am = __ballot(1);
    if (wc && (threadIdx.x & 63) == (int)__builtin_ctzll(am)) {
        atomicAdd(wc + 20, 1u);
        if (cm) atomicAdd(wc + 21, 1u);
    }
}
#else
PTMI_PROBE unsigned int *sphere_counters(unsigned int *) { return nullptr; }
PTMI_PROBE static void sphere_test(unsigned int *, bool) {}
PTMI_PROBE static void plane_test(unsigned int *, bool) {}
#endif

// ---- render Inline: [1] lane-trips, [2..4] lane participations in the shade round(s) and the trace round, [5] trips of the wave's
// slowest lane x 64 (what the wave paid for), [6] shades finished by the frozen-shade shortcut, [8..13] wave cycles per round (u64;
// the waves of a SIMD interleave, so these are shares, not costs); WAVE-level executions of the loop's blocks (how often a wave runs
// the block, whatever its lanes: what instruction counts multiply with, tools/isa_other.py): [24] trips, [25] the frozen check (any lane
// pending at the top), [26] the frozen finish (any lane), [27] the restart block (any lane over), [28] the full shade (any lane pending),
// [29] the trace (any lane with a ray)
struct PhaseProbe {
#ifdef PTMI_PHASE_STATS
    unsigned int iter = 0, a = 0, b = 0, c = 0, f = 0;
    unsigned int w_trip = 0, w_check = 0, w_finish = 0, w_restart = 0, w_shade = 0, w_trace = 0;     // wave-level (uniform across the lanes that count)
    bool froze = false;
    unsigned long long cyc_a = 0, cyc_b = 0, cyc_c = 0, t_prev = 0;
    PTMI_PROBE void trip() { ++iter; ++w_trip; froze = false; t_prev = __builtin_amdgcn_s_memtime(); }
    PTMI_PROBE void round_a(bool on) { if (on) ++a; }
    PTMI_PROBE void round_b(bool on) { if (on) ++b; }
    PTMI_PROBE void round_c(bool on) { if (on) ++c; if (__any(on)) ++w_trace; }
    PTMI_PROBE void frozen() { ++f; froze = true; }
    PTMI_PROBE void check(bool pending) { if (__any(pending)) ++w_check; }
    PTMI_PROBE void restart(bool over) { if (__any(froze)) ++w_finish; if (__any(over)) ++w_restart; }
    PTMI_PROBE void shade(bool pending) { if (__any(pending)) ++w_shade; }
    PTMI_PROBE void end_a() { const unsigned long long t = __builtin_amdgcn_s_memtime(); cyc_a += t - t_prev; t_prev = t; }
    PTMI_PROBE void end_b() { const unsigned long long t = __builtin_amdgcn_s_memtime(); cyc_b += t - t_prev; t_prev = t; }
    PTMI_PROBE void end_c() { cyc_c += __builtin_amdgcn_s_memtime() - t_prev; }
    PTMI_PROBE void flush(unsigned int *wc)
    {
        const unsigned int mx = wave_max(iter);
        atomicAdd(wc + 1, iter); atomicAdd(wc + 2, a); atomicAdd(wc + 3, b); atomicAdd(wc + 4, c); atomicAdd(wc + 6, f);
        if (first_active_lane()) {
            atomicAdd(wc + 5, mx * 64u);
            atomicAdd(reinterpret_cast<unsigned long long *>(wc + 8), cyc_a);
            atomicAdd(reinterpret_cast<unsigned long long *>(wc + 10), cyc_b);
            atomicAdd(reinterpret_cast<unsigned long long *>(wc + 12), cyc_c);
            // (a lane that entered the loop late or left it early counted fewer wave-level events than the wave ran: the wave's figure is the
            // maximum over its lanes; the first active lane adds it)
        }
        const unsigned int wt = wave_max(w_trip), wk = wave_max(w_check), wf = wave_max(w_finish), wr = wave_max(w_restart), ws = wave_max(w_shade), wc_ = wave_max(w_trace);
        if (first_active_lane()) {
            atomicAdd(wc + 24, wt); atomicAdd(wc + 25, wk); atomicAdd(wc + 26, wf); atomicAdd(wc + 27, wr); atomicAdd(wc + 28, ws); atomicAdd(wc + 29, wc_);
        }
    }
    PTMI_PROBE void flush_lanes_only(unsigned int *wc) { atomicAdd(wc + 1, iter); atomicAdd(wc + 2, a); atomicAdd(wc + 3, b); atomicAdd(wc + 4, c); }
#else
    PTMI_PROBE static void trip() {}
    PTMI_PROBE static void round_a(bool) {}
    PTMI_PROBE static void round_b(bool) {}
    PTMI_PROBE static void round_c(bool) {}
    PTMI_PROBE static void frozen() {}
    PTMI_PROBE static void check(bool) {}
    PTMI_PROBE static void restart(bool) {}
    PTMI_PROBE static void shade(bool) {}
    PTMI_PROBE static void end_a() {}
    PTMI_PROBE static void end_b() {}
    PTMI_PROBE static void end_c() {}
    PTMI_PROBE static void flush(unsigned int *) {}
    PTMI_PROBE static void flush_lanes_only(unsigned int *) {}
#endif
};

// ---- tree walk: [1] lane-trips needed, [2] dead-ray finishes, [3] shades, [4] traces (lane participations), [5] lane-trips the
// wave paid for (its longest lane x 64)
struct TreeProbe {
#ifdef PTMI_TREE_STATS
    unsigned int n_dead = 0, n_shade = 0, n_trace = 0;
    PTMI_PROBE void dead(bool on) { if (on) ++n_dead; }
    PTMI_PROBE void shade() { ++n_shade; }
    PTMI_PROBE void trace() { ++n_trace; }
    PTMI_PROBE void flush_lane(unsigned int *wc, unsigned int trips)
    {
        atomicAdd(wc + 1, trips); atomicAdd(wc + 2, n_dead); atomicAdd(wc + 3, n_shade); atomicAdd(wc + 4, n_trace);
    }
    PTMI_PROBE void flush_wave(unsigned int *wc, unsigned int trips)
    {
        const unsigned int mx = wave_max(trips);
        if ((threadIdx.x & 63) == 0) atomicAdd(wc + 5, mx * 64u);
    }
#else
    PTMI_PROBE static void dead(bool) {}
    PTMI_PROBE static void shade() {}
    PTMI_PROBE static void trace() {}
    PTMI_PROBE static void flush_lane(unsigned int *, unsigned int) {}
    PTMI_PROBE static void flush_wave(unsigned int *, unsigned int) {}
#endif
#ifdef PTMI_TREE_STATS_MAP
    PTMI_PROBE void cost_map(float &red, unsigned int trips) { red = (float)trips; }
#else
    PTMI_PROBE static void cost_map(float &, unsigned int) {}
#endif
};

// ---- streams_pixels_kernel (all u64, shader-clock ticks): [8] first start (negated), [10] last end, [12] sum of the waves' durations,
// [14] waves, [16] lanes-with-item x trips, [18] trips, [20] longest wave; then either (-DPTMI_TAIL_PHASES) [24..] cycles in the refill
// block / in the sample-end block of trips that end an item / refills / such trips / items taken, or [24, 64): waves by duration,
// bins of 2^19 cycles
struct TailProbe {
#ifdef PTMI_TAIL_STATS
    unsigned long long t_start = 0, lane_trips = 0, wave_trips = 0;
    unsigned long long ph_refill = 0, ph_over = 0, ph_refills = 0, ph_ends = 0, ph_taken = 0, ph_t = 0;
    bool ending = false;
    PTMI_PROBE void begin() { t_start = __builtin_readcyclecounter(); }
    PTMI_PROBE void refill_begin() { ph_t = __builtin_readcyclecounter(); ++ph_refills; }
    PTMI_PROBE void refill_end(unsigned int take)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ph_taken += take; ph_refill += __builtin_readcyclecounter() - ph_t;
    }
    PTMI_PROBE void trip(bool busy) { lane_trips += (unsigned long long)lanes(busy); ++wave_trips; }
    PTMI_PROBE void ending_begin(bool over, int s, const float *samples) { ending = __any(over && s + 1 >= (int)__float_as_uint(*samples)); if (ending) { ph_t = __builtin_readcyclecounter(); ++ph_ends; } }
    PTMI_PROBE void ending_end() { if (ending) ph_over += __builtin_readcyclecounter() - ph_t; }
    PTMI_PROBE void flush(unsigned int *work_counter)
    {
        if ((threadIdx.x & 63) != 0) return;
        unsigned long long *wc = reinterpret_cast<unsigned long long *>(work_counter + 8);
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicMax(wc + 0, ~t_start); atomicMax(wc + 1, t_end); atomicAdd(wc + 2, t_end - t_start); atomicAdd(wc + 3, 1ull);
        atomicAdd(wc + 4, lane_trips); atomicAdd(wc + 5, wave_trips);
        atomicMax(wc + 6, t_end - t_start);
#ifdef PTMI_TAIL_PHASES
        atomicAdd(wc + 8, ph_refill); atomicAdd(wc + 9, ph_over); atomicAdd(wc + 10, ph_refills); atomicAdd(wc + 11, ph_ends); atomicAdd(wc + 12, ph_taken);
#else
        const unsigned long long bin = (t_end - t_start) >> 19;
        atomicAdd(work_counter + 24 + (bin < 39ull ? (unsigned int)bin : 39u), 1u);
#endif
    }
#else
    PTMI_PROBE static void begin() {}
    PTMI_PROBE static void refill_begin() {}
    PTMI_PROBE static void refill_end(unsigned int) {}
    PTMI_PROBE static void trip(bool) {}
    PTMI_PROBE static void ending_begin(bool, int, const float *) {}   // (arguments are values the kernel already holds: a load or an `a && b` in an argument is code the optimiser must first remove, and that moved the product's instructions)
    PTMI_PROBE static void ending_end() {}
    PTMI_PROBE static void flush(unsigned int *) {}
#endif
};

// -DPTMI_TRAFFIC_SKIP=<bits>: MEASUREMENT builds of the split kernel that leave one source of its HBM traffic out, so that the bytes of
// a launch can be attributed term by term (tools/traffic_terms.py; their planes are wrong on purpose): 1 = the item-end colour atomics,
// 2 = the colour atomics of rays taken from the ring, 4 = the cost record of an item (quad_cost atomics).  0 in every other build: the
// conditions fold away and the kernel is the kernel without them, instruction for instruction.
#ifndef PTMI_TRAFFIC_SKIP
#define PTMI_TRAFFIC_SKIP 0
#endif
constexpr unsigned int kTrafficSkip = PTMI_TRAFFIC_SKIP;

// ---- streams_split_kernel, summed over the waves: [1] trips, [2] dead hits finished, [3] lanes free for a next ray, [4] ... that took
// one from the ring, [5] ... that started a sample of their item, [6] ... whose item ended, [7] refill blocks run, [8] hits shaded,
// [9] ... of which GLASS, [10] rays traced, [11] lanes holding an item, [12] sum and [14] maximum of the waves' durations (u64),
// [16] waves, [17] GLASS hits parked for a later trip, [18] trips that ran the glass block, [19] lanes in it,
// [20] children taken from the XCD's shared queue, [21] ... given to it, [22] the first wave's start (u64, negated); per XCD x: [24+x] waves, [32+x] sum of durations >> 12,
// [40+x] longest >> 12, [48+x] trips
struct SplitProbe {
#if defined(PTMI_SPLIT_ENDS) && !defined(PTMI_SPLIT_STATS)
    // -DPTMI_SPLIT_ENDS: ONLY when the waves end -- two atomics per wave, so that the measurement does not make the tail it looks for (the
    // full statistics' flush is 6 144 waves x 45 atomics on a dozen lines: milliseconds of atomics that the still-running waves' ticket
    // atomics queue behind; round 3 read that as "waves end between 85 and 100 % of the launch").  [22] first start (u64, negated),
    // [96, 160): waves by the time they end, bins of 2^PTMI_SPLIT_HIST_SHIFT ticks of 10 ns, a window of 64 bins from PTMI_SPLIT_HIST_FIRST on.
    unsigned int *counters = nullptr;
    PTMI_PROBE void begin(unsigned int *wc)
    {
        counters = wc;
        if ((threadIdx.x & 63) == 0 && wc) atomicMax(reinterpret_cast<unsigned long long *>(wc + 22), ~__builtin_amdgcn_s_memrealtime());
    }
    PTMI_PROBE static void trip(bool, bool) {}
    PTMI_PROBE static void refill() {}
    PTMI_PROBE static void next_ray(unsigned long long, unsigned int, bool, bool) {}
    PTMI_PROBE static void shade(bool, bool) {}
    PTMI_PROBE static void parked(unsigned int) {}
    PTMI_PROBE static void glass_block(unsigned int) {}
    PTMI_PROBE static void stolen(unsigned int) {}
    PTMI_PROBE static void shared(unsigned int) {}
    PTMI_PROBE static void trace(bool) {}
    PTMI_PROBE static void stamp(int) {}
    PTMI_PROBE static void tickets(bool, bool, int, unsigned int, unsigned int) {}
    PTMI_PROBE void flush(unsigned int *wc, unsigned int)
    {
        if ((threadIdx.x & 63) != 0 || !wc) return;
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const unsigned long long t0 = ~__hip_atomic_load(reinterpret_cast<unsigned long long *>(wc + 22), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long be = (t_end - t0) >> PTMI_SPLIT_HIST_SHIFT;
        atomicAdd(wc + 96 + (be < PTMI_SPLIT_HIST_FIRST ? 0u : (be < PTMI_SPLIT_HIST_FIRST + 63ull ? (unsigned int)(be - PTMI_SPLIT_HIST_FIRST) : 63u)), 1u);
    }
#elif defined(PTMI_SPLIT_STATS)
    unsigned int n_trips = 0, n_dead = 0, n_free = 0, n_ring = 0, n_start = 0, n_end = 0, n_refill = 0, n_shade = 0, n_glass = 0, n_trace = 0, n_busy = 0;
    unsigned int n_parked = 0, n_glass_trips = 0, n_glass_lanes = 0, n_stolen = 0, n_shared = 0;
    unsigned long long t_start = 0, t_start_real = 0;
    PTMI_PROBE void begin(unsigned int *wc)
    {
        t_start = __builtin_readcyclecounter();
        // (the shader-clock counter is not one clock across the chip: launch-wide times come from the 100-MHz s_memrealtime)
        t_start_real = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && wc) atomicMax(reinterpret_cast<unsigned long long *>(wc + 22), ~t_start_real);     // (the first start, negated)
    }
    PTMI_PROBE void trip(bool dead, bool busy) { ++n_trips; n_dead += lanes(dead); n_busy += lanes(busy); }
    PTMI_PROBE void refill() { ++n_refill; }
    PTMI_PROBE void next_ray(unsigned long long free_m, unsigned int ring_n, bool starts, bool ends)
    {
        const unsigned int n = (unsigned int)__builtin_popcountll(free_m);
        n_free += n; n_ring += ring_n < n ? ring_n : n; n_start += lanes(starts); n_end += lanes(ends);
    }
    PTMI_PROBE void shade(bool alive, bool glass) { n_shade += lanes(alive); n_glass += lanes(glass); }
    PTMI_PROBE void parked(unsigned int n) { n_parked += n; }
    PTMI_PROBE void glass_block(unsigned int n) { ++n_glass_trips; n_glass_lanes += n; }
    PTMI_PROBE void stolen(unsigned int n) { n_stolen += n; }
    PTMI_PROBE void shared(unsigned int n) { n_shared += n; }
    PTMI_PROBE void trace(bool has_ray) { n_trace += lanes(has_ray); }
    // [224, 240): wave cycles (shader clock, u64) from one stamp to the next, by the block that ends at the stamp: 0 dead hits, 1 refill, 2 next
    // ray / sample start / item end, 3 shade (first piece), 4 GLASS block + expand, 5 shade (rest), 6 trace, 7 loop control.  The waves of a SIMD
    // interleave, so these are SHARES of a wave's time, not costs: a block whose share exceeds its share of the instructions is where the wave waits.
    unsigned long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    PTMI_PROBE void stamp(int k) { const unsigned long long t = __builtin_readcyclecounter(); if (t_last) cyc[k] += t - t_last; t_last = t; }
    // what the wave still holds when it finds the last queue exhausted: [56] lanes with an item, [57] their samples left, [58] spill records,
    // [59] ring records, [60] the trips it makes after that (sum over the waves), [61] the most of any wave
    unsigned int at_end_busy = 0, at_end_samples = 0, at_end_spill = 0, at_end_ring = 0, trips_at_end = 0;
    bool ended = false;
    PTMI_PROBE void tickets(bool left, bool busy, int samples_left, unsigned int spill_n, unsigned int ring_n)
    {
        if (left || ended) return;
        ended = true; trips_at_end = n_trips;
        at_end_busy = lanes(busy);
        unsigned int s = busy && samples_left > 0 ? (unsigned int)samples_left : 0u;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        at_end_samples = s; at_end_spill = spill_n; at_end_ring = ring_n;
    }
    PTMI_PROBE void flush(unsigned int *wc, unsigned int xcd)
    {
        if ((threadIdx.x & 63) != 0 || !wc) return;
        const unsigned long long t_end_real = __builtin_amdgcn_s_memrealtime();    // BEFORE the atomics below: 6 144 waves x 40 atomics on the same few lines take milliseconds
        const unsigned long long dur = __builtin_readcyclecounter() - t_start;
        atomicAdd(wc + 56, at_end_busy); atomicAdd(wc + 57, at_end_samples); atomicAdd(wc + 58, at_end_spill); atomicAdd(wc + 59, at_end_ring);
        atomicAdd(wc + 60, n_trips - trips_at_end); atomicMax(wc + 61, n_trips - trips_at_end);
        for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(wc + 224 + 2 * k), cyc[k]);
        atomicAdd(wc + 1, n_trips); atomicAdd(wc + 2, n_dead); atomicAdd(wc + 3, n_free); atomicAdd(wc + 4, n_ring);
        atomicAdd(wc + 5, n_start); atomicAdd(wc + 6, n_end); atomicAdd(wc + 7, n_refill); atomicAdd(wc + 8, n_shade);
        atomicAdd(wc + 9, n_glass); atomicAdd(wc + 10, n_trace); atomicAdd(wc + 11, n_busy);
        atomicAdd(reinterpret_cast<unsigned long long *>(wc + 12), dur);
        atomicMax(reinterpret_cast<unsigned long long *>(wc + 14), dur);
        atomicAdd(wc + 16, 1u);
        atomicAdd(wc + 17, n_parked); atomicAdd(wc + 18, n_glass_trips); atomicAdd(wc + 19, n_glass_lanes);
        atomicAdd(wc + 20, n_stolen); atomicAdd(wc + 21, n_shared);
        atomicAdd(wc + 24 + xcd, 1u); atomicAdd(wc + 32 + xcd, (unsigned int)(dur >> 12)); atomicMax(wc + 40 + xcd, (unsigned int)(dur >> 12));
        atomicAdd(wc + 48 + xcd, n_trips);
        // [64, 96): the waves by the time they START, [96, 160): by the time they END (from bin 32 on), [160, 224): the trips of the waves that
        // ended in the bin (did the late waves do more, or did they run slower?) -- times from the first wave's start, bins of 2^PTMI_SPLIT_HIST_SHIFT ticks of 10 ns
        const unsigned long long t0 = ~__hip_atomic_load(reinterpret_cast<unsigned long long *>(wc + 22), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long bs = (t_start_real - t0) >> PTMI_SPLIT_HIST_SHIFT, be = (t_end_real - t0) >> PTMI_SPLIT_HIST_SHIFT;
        const unsigned int bin = be < PTMI_SPLIT_HIST_FIRST ? 0u : (be < PTMI_SPLIT_HIST_FIRST + 63ull ? (unsigned int)(be - PTMI_SPLIT_HIST_FIRST) : 63u);
        atomicAdd(wc + 64 + (bs < 31ull ? (unsigned int)bs : 31u), 1u);
        atomicAdd(wc + 96 + bin, 1u); atomicAdd(wc + 160 + bin, n_trips);
    }
#else
    PTMI_PROBE static void begin(unsigned int *) {}
    PTMI_PROBE static void trip(bool, bool) {}
    PTMI_PROBE static void refill() {}
    PTMI_PROBE static void next_ray(unsigned long long, unsigned int, bool, bool) {}
    PTMI_PROBE static void shade(bool, bool) {}
    PTMI_PROBE static void parked(unsigned int) {}
    PTMI_PROBE static void glass_block(unsigned int) {}
    PTMI_PROBE static void stolen(unsigned int) {}
    PTMI_PROBE static void shared(unsigned int) {}
    PTMI_PROBE static void trace(bool) {}
    PTMI_PROBE static void stamp(int) {}
    PTMI_PROBE static void tickets(bool, bool, int, unsigned int, unsigned int) {}
    PTMI_PROBE static void flush(unsigned int *, unsigned int) {}
#endif
};

}  // namespace diag
}  // namespace
}  // namespace ptmi
