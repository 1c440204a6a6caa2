// Pedestal removal, second part: the recurrence over the runs in reference row order
// (spectra.c:66-78 taken run by run) -- run_links (which earlier runs hold a run's end slots, and
// what they added there), run_solve (the recurrence as a triangular system: a few sweeps inside one
// launch) and run_chain (the serial form, for the levels the sweeps leave; run by the chunk that
// leaves run_solve last).  Included by pedestal.h after pedestal_runs.h.
#pragma once

namespace lbl {

// ---------------------------------------------------------------------------------------
// The recurrence in the pedestals alone, solved by relaxation (round 4).
//
// The value accumulated on a slot c before run r is
//     sum_{q<r, c in W_q} G_q[c]  -  sum_{q<r, c in W_q} P_q
// (G: profile sums of run q on its slots, W_q its window, P_q the sum of its pedestals), so
//     P_r = min( Ks_r - sum_{q<r} P_q [fs_r in W_q] ,  Ke_r - sum_{q<r} P_q [ls_r in W_q] )
// with fs_r / ls_r the run's end slots and Ks / Ke the profile sums on them (earlier runs' and
// its own).  Which earlier runs hold a slot, and the G sums, do not depend on the pedestals:
// run_links_kernel sums them in parallel, one wavefront per run.  What is left is a triangular
// system in the P's -- every P_r is a fixed function of earlier ones -- and a triangular system
// has exactly one solution, which plain iteration P <- F(P) reaches from any start after as many
// sweeps as its longest chain of dependences that MATTER: a run whose last slot is the smaller end
// takes its pedestal from there, and a last slot is fresh -- only the run's own window, and the
// neighbour's where pressure shifts make two windows alternate, has added to it -- so such a run
// does not look at history at all.  On every line table tried (uniform, banded, sparse:
// profiles/r04_pedestal_branches.txt) that is all but a handful of runs, and chains are 2-4 runs
// long.  run_solve_kernel: one wavefront per 64 consecutive runs (lane = run), the chunk solved
// exactly by forward substitution, earlier chunks' values taken from their previous sweep.  The
// second and later sweeps report whether anything changed, and a sweep that changed nothing
// has verified a fixed point, i.e. the solution a serial evaluation of the same formula gives, bit
// for bit and independent of how it was reached.  Levels that have not settled after the last
// sweep get the serial chain.  ~25 us for the 400 k-line benchmark table, where the serial
// forms take a wavefront 0.2-0.75 ms: one wavefront issues an instruction every fourth cycle at
// best, and 5 300 dependent steps of ~100 instructions are 2 M cycles however they are arranged.
//
// WHICH earlier runs: a window is fixed by its bin b (RunMeta::bin) -- slots max(b - 2 cut_off - 1,
// 0) to min(b, slot of the last grid point) -- so a slot c is held by the bins c ... c + 2 cut_off
// + 1 and by no other.  In a table in ascending order whose pressure shifts move a line by less
// than a wavenumber every run before r has a bin <= b_r + 1 (checked: prefix maxima of the bins;
// the serial chain takes a level where it fails), so the prefix maxima rise with the run index and
// the runs that can hold c begin where the prefix maximum reaches c: found by a search, then one
// stretch of runs up to r, each tested against the slot.  However many runs that is -- 60 for the
// benchmark's tables, 1 800 where a 4 M-line table has 800 lines to the wavenumber and dozens of
// them alternate between two windows at every integer -- nothing is out of sight.  (The first form
// of this kept bit masks over the previous 256 runs and gave up beyond.)  What still costs
// sweeps is a chain that matters ACROSS chunks: one sweep per boundary it crosses.
// ---------------------------------------------------------------------------------------
// Per level (kChainState ints, reset by run_find_kernel): [0] 1 while the relaxation applies
// (cleared by run_links_kernel where rows are too far out of order, by the first sweep where one
// window's runs are spread over more than kMaxStretch runs: its total would be one lane's walk of
// thousands, and by a chunk whose wait for an earlier chunk ran out), [k] something changed in
// sweep k (k = 1 .. sweeps-1), [8] chunks of the level that have left run_solve_kernel.
constexpr int kMaxRelaxLaunches = 7;
constexpr int kMaxStretch = 1024;
constexpr int kMaxHistory = 16384;  // earlier runs a run may have to look at (a 4 M-line table: 1 800)
constexpr int kStateFinished = 8;

// A sweep that changed nothing has verified the values it was handed.
__device__ __forceinline__ bool chain_verified_before(const int * state, int sweep)
{
    for (int k = 1; k < sweep; ++k)
    {
        if (state[k] == 0) return true;
    }
    return false;
}

__device__ __forceinline__ bool chain_settled(const int * state, int sweeps)
{
    return state[0] != 0 && chain_verified_before(state, sweeps);
}

// First index in [lo, hi) whose value is >= x in a non-decreasing array, found by a whole
// wavefront: 64 probes per step.
__device__ inline int wave_lower_bound(const int * __restrict__ values, int lo, int hi, int x)
{
    const int lane = threadIdx.x & 63;
    while (hi - lo > 64)
    {
        const int stride = (hi - lo + 63) >> 6;
        const int at = lo + lane*stride;
        const bool below = at < hi && values[at] < x;
        const int count = __builtin_popcountll(__ballot(below));
        if (count == 0) return lo;
        const int base = lo + (count - 1)*stride;
        hi = min(base + stride, hi);
        lo = base + 1;
    }
    const int at = lo + lane;
    const bool below = at < hi && values[at] < x;
    return lo + __builtin_popcountll(__ballot(below));
}

// One wavefront per run: the stretch of earlier runs that can hold its first slot (from where
// the prefix maximum of the bins reaches that slot), what they added on its end slots, which runs
// of its own chunk of 64 hold them, and whether it is the first run of its bin.
__global__ __launch_bounds__(64) void run_links_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int slot_stride, int n_bins,
                                                       const RunMeta * __restrict__ runs,
                                                       const double * __restrict__ slot_sums,
                                                       const int * __restrict__ prefix_bin,
                                                       long long prefix_stride,
                                                       int * __restrict__ bin_end,
                                                       int * __restrict__ bin_first,
                                                       int * __restrict__ progress, int max_chunks,
                                                       RunLink * __restrict__ links,
                                                       int2 * __restrict__ run_slots,
                                                       int * __restrict__ run_bin,
                                                       int * __restrict__ state)
{
    const int level = blockIdx.y;
    const int lane = threadIdx.x;
    const int count = run_count[level];
    const RunMeta * meta = runs + (long long)level*max_runs;
    const double * sums = slot_sums + (long long)level*max_runs*slot_stride;
    const int * prefix = prefix_bin + (long long)level*prefix_stride;
    for (int r = blockIdx.x; r < count; r += gridDim.x)
    {
        const RunMeta m = meta[r];
        const bool bin_ok = m.bin >= 0 && m.bin < n_bins;
        // (the slots' bins: a run holds slot c exactly if its bin lies in c ... c + 2 cut_off + 1,
        // so the earliest holder of the first slot is the first run whose prefix maximum is >= it)
        const int begin = wave_lower_bound(prefix, 0, r, m.first_slot);
        // (rows in no order at all: every row its own run and every stretch the whole table --
        // such a level is the serial chain's, and nobody walks its stretches)
        const bool too_long = r - begin > kMaxHistory;
        const bool given_up = __builtin_amdgcn_readfirstlane(state[level*kChainState]) == 0;
        double gs = 0., ge = 0.;
        bool seen_before = false;
        for (int q0 = (too_long || given_up) ? r : begin; q0 < r; q0 += 64)
        {
            const int q = q0 + lane;
            if (q < r)
            {
                const RunMeta e = meta[q];
                if (e.first_slot <= m.first_slot && m.first_slot <= e.last_slot)
                {
                    gs += sums[(long long)q*slot_stride + (m.first_slot - e.first_slot)];
                }
                if (e.first_slot <= m.last_slot && m.last_slot <= e.last_slot)
                {
                    ge += sums[(long long)q*slot_stride + (m.last_slot - e.first_slot)];
                }
                seen_before = seen_before || e.bin == m.bin;
            }
        }
        for (int offset = 32; offset > 0; offset >>= 1)
        {
            gs += __shfl_xor(gs, offset, 64);
            ge += __shfl_xor(ge, offset, 64);
        }
        // The runs of the same chunk of 64: bit j is run r-1-j.
        const int q = r - 1 - lane;
        bool holds_s = false, holds_e = false;
        if (q >= (r & ~63))
        {
            const RunMeta e = meta[q];
            holds_s = e.first_slot <= m.first_slot && m.first_slot <= e.last_slot;
            holds_e = e.first_slot <= m.last_slot && m.last_slot <= e.last_slot;
        }
        const unsigned long long in_s = __ballot(holds_s), in_e = __ballot(holds_e);
        const bool first_of_bin = __ballot(seen_before) == 0ull;
        if (lane == 0)
        {
            // Rows too far out of order for the stretch to be what it is taken for.
            const bool displaced = !bin_ok || (r > 0 && prefix[r - 1] > m.bin + 1);
            if (displaced || too_long) atomicAnd(&state[level*kChainState], 0);
            if (bin_ok)
            {
                atomicMax(&bin_end[(long long)level*n_bins + m.bin], r + 1);
                atomicMin(&bin_first[(long long)level*n_bins + m.bin], r);
            }
            // (the sweeps this chunk of 64 runs has completed: run_solve_kernel)
            if ((r & 63) == 0) progress[(long long)level*max_chunks + (r >> 6)] = 0;
            RunLink link;
            link.ks = gs + m.vs;
            link.ke = ge + m.ve;
            link.in_s = in_s;
            link.in_e = in_e;
            link.bin = m.bin;
            link.n_slots = m.n_slots;
            link.begin = begin;
            link.first_of_bin = first_of_bin ? 1 : 0;
            links[(long long)level*max_runs + r] = link;
            run_slots[(long long)level*max_runs + r] = make_int2(m.first_slot, m.last_slot);
            run_bin[(long long)level*max_runs + r] = m.bin;
        }
    }
}

// min(k_s, k_e), taken like the serial chain takes it (run_chain_kernel).
__device__ __forceinline__ double run_pedestal(double k_s, double k_e, int n_slots)
{
    return (n_slots == 1 || !(k_s - k_e > 0.)) ? k_s : k_e;
}

constexpr int kChainChunk = 16;     // runs whose slot sums are staged in LDS at a time
// Small-LDS form: the slots of the spectrum live in HBM, kChainRing consecutive ones of them in LDS
// (the register window moves inside that ring at the price of an LDS round trip; only when it
// leaves the ring -- every ~140 bins of a sorted table -- does the chain wait for HBM).  Staging
// and ring together ask for 15.5 KB: less than one accumulate workgroup holds, so the kernel -- which
// usually only reads the flags and leaves -- finds a place on a busy chip at once.
constexpr int kChainRing = 256;
constexpr int kChainRingBack = 64;  // slots kept behind the window that re-bases the ring

// One wavefront per level: the serial recurrence over runs.  Slots (the accumulated
// spectrum on integer wavenumbers) and the per-window pedestal totals live in LDS; the
// inputs of the next kChainChunk runs are staged cooperatively so that no global-memory
// latency sits on the serial chain.
template <bool USE_LDS, bool WINDOW>
__device__ __forceinline__ void run_chain(int level, const int * __restrict__ run_count,
                                          int max_runs, int slot_stride, GridSpec g, int n_cells,
                                          int n_bins, const RunMeta * __restrict__ runs,
                                          const double * __restrict__ slot_sums,
                                          double * __restrict__ global_slots,
                                          double * __restrict__ bin_sum, double * lds)
{
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    // LDS carve: [2 x staged slot sums][slots][bin sums]; without LDS room the last two are in HBM.
    double * staged = lds;
    double * a = USE_LDS ? lds + 2*kChainChunk*slot_stride
                         : global_slots + (long long)level*(n_cells + 1);
    double * bins = USE_LDS ? a + (n_cells + 1) : bin_sum + (long long)level*n_bins;
    for (int s = lane; s <= n_cells; s += 64) a[s] = 0.;
    for (int s = lane; s < n_bins; s += 64) bins[s] = 0.;
    __syncthreads();
    const int count = run_count[level];
    const RunMeta * meta = runs + (long long)level*max_runs;
    const double * sums = slot_sums + (long long)level*max_runs*slot_stride;
    // Inputs of chunk c+1 are fetched into registers while chunk c is being chained, and
    // parked in the other half of the LDS staging area afterwards: no global-memory latency
    // on the serial path.  (kStageLoads*64 doubles cover a chunk for cut_off <= 30.)
    constexpr int kStageLoads = kChainChunk;
    const bool prefetch = kChainChunk*slot_stride <= kStageLoads*64;
    double ahead[kStageLoads];
    RunMeta mine_next;
    auto fetch = [&](int base) {
        const int chunk = min(kChainChunk, count - base);
        mine_next = meta[base + min(lane, max(chunk - 1, 0))];
#pragma unroll
        for (int u = 0; u < kStageLoads; ++u)
        {
            const int i = u*64 + lane;
            ahead[u] = i < chunk*slot_stride ? sums[(long long)base*slot_stride + i] : 0.;
        }
    };
    auto park = [&](double * where, int chunk) {
#pragma unroll
        for (int u = 0; u < kStageLoads; ++u)
        {
            const int i = u*64 + lane;
            if (i < chunk*slot_stride) where[i] = ahead[u];
        }
    };
    double * stage_a = lds;
    double * stage_b = lds + kChainChunk*slot_stride;
    if (prefetch && count > 0)
    {
        fetch(0);
        park(stage_a, min(kChainChunk, count));
    }
    int zone = 0;           // WINDOW: first slot held in registers
    double window = 0.;     // WINDOW: slot zone + lane (all slots start at zero)
    // !USE_LDS && WINDOW: slots [ring_base, ring_base + kChainRing) are current in `ring`, the
    // others in HBM.
    double * ring = lds + 2*kChainChunk*slot_stride;
    int ring_base = 0;
    if (!USE_LDS && WINDOW)
    {
        for (int s = lane; s < kChainRing; s += 64) ring[s] = 0.;
        __syncthreads();
    }
    for (int base = 0; base < count; base += kChainChunk)
    {
        const int chunk = min(kChainChunk, count - base);
        RunMeta mine;
        if (prefetch)
        {
            mine = mine_next;
            staged = ((base/kChainChunk) & 1) ? stage_b : stage_a;
            if (base + kChainChunk < count) fetch(base + kChainChunk);
        }
        else
        {
            mine = meta[base + min(lane, chunk - 1)];
            for (int i = lane; i < chunk*slot_stride; i += 64)
            {
                staged[i] = sums[(long long)base*slot_stride + i];
            }
        }
        __syncthreads();
        for (int r = 0; r < chunk; ++r)
        {
            // Run r's scalars are wave-uniform: v_readlane, no LDS traffic.
            const int n_slots = __builtin_amdgcn_readlane(mine.n_slots, r);
            const int bin = __builtin_amdgcn_readlane(mine.bin, r);
            const int first_slot = __builtin_amdgcn_readlane(mine.first_slot, r);
            const int last_slot = __builtin_amdgcn_readlane(mine.last_slot, r);
            const double vs = read_lane(mine.vs, r);
            const double ve = read_lane(mine.ve, r);
            const bool bin_ok = bin >= 0 && bin < n_bins;
            if (WINDOW)
            {
                // Register window: lane l holds slot zone + l.  A window is a contiguous
                // range of at most 64 slots (the last grid point counts as slot n_cells), so
                // consecutive windows almost always fit the zone already loaded and the step
                // touches no memory on its dependent path: the end values come out of the
                // registers with v_readlane, the update is one masked vector add.
                if (first_slot < zone || last_slot > zone + 63)
                {
                    if (USE_LDS)
                    {
                        if (zone + lane <= n_cells) a[zone + lane] = window;
                        __builtin_amdgcn_wave_barrier();
                        zone = first_slot;
                        window = zone + lane <= n_cells ? a[zone + lane] : 0.;
                    }
                    else
                    {
                        ring[(zone + lane) & (kChainRing - 1)] = window;
                        __builtin_amdgcn_wave_barrier();
                        if (first_slot < ring_base || first_slot + 63 >= ring_base + kChainRing)
                        {
                            // The ring goes back to HBM and is filled again around the new window.
#pragma unroll
                            for (int k = 0; k < kChainRing/64; ++k)
                            {
                                const int slot = ring_base + k*64 + lane;
                                if (slot <= n_cells) a[slot] = ring[slot & (kChainRing - 1)];
                            }
                            __syncthreads();
                            ring_base = max(first_slot - kChainRingBack, 0);
#pragma unroll
                            for (int k = 0; k < kChainRing/64; ++k)
                            {
                                const int slot = ring_base + k*64 + lane;
                                ring[slot & (kChainRing - 1)] = slot <= n_cells ? a[slot] : 0.;
                            }
                            __syncthreads();
                        }
                        zone = first_slot;
                        window = ring[(zone + lane) & (kChainRing - 1)];
                    }
                }
                const int f = first_slot - zone, e = last_slot - zone;
                const bool interior = lane > f && lane < e;
                const double add = interior ? staged[r*slot_stride + (lane - f)] : 0.;
                const double a_s = read_lane(window, f);
                const double a_e = read_lane(window, e);
                // The run leaves (k_s, k_e) - min(k_s, k_e) on its end slots and has
                // subtracted min(k_s, k_e) in all (see the header): taken from the smaller side
                // directly, so that a line peak on the other end slot costs no accuracy.
                const double k_s = a_s + vs, k_e = a_e + ve;
                const double delta_n = k_s - k_e;
                const double s_new = delta_n > 0. ? delta_n : 0.;
                const double e_new = delta_n < 0. ? -delta_n : 0.;
                const double pedestal = (n_slots == 1 || !(delta_n > 0.)) ? k_s : k_e;
                double value = window + (add - pedestal);
                if (!interior) value = window;
                if (lane == e) value = e_new;
                if (lane == f) value = n_slots == 1 ? 0. : s_new;
                window = value;
                if (lane == 0 && bin_ok) atomicAdd(&bins[bin], pedestal);
                continue;
            }
            // Everything the step reads is requested up front (one LDS round trip); the
            // interior slots do not depend on the end slots.
            const bool interior = lane > 0 && lane < n_slots - 1 && lane < 64;
            const double a_s = a[first_slot];
            const double a_e = a[last_slot];
            const double bin_old = bins[bin_ok ? bin : 0];
            const double mid = interior ? a[first_slot + lane] : 0.;
            const double add = interior ? staged[r*slot_stride + lane] : 0.;
            const double k_s = a_s + vs, k_e = a_e + ve;
            const double delta_n = k_s - k_e;
            const double s_new = delta_n > 0. ? delta_n : 0.;
            const double e_new = delta_n < 0. ? -delta_n : 0.;
            const double pedestal = (n_slots == 1 || !(delta_n > 0.)) ? k_s : k_e;
            // One wavefront owns this memory: its LDS accesses execute in program order, so
            // only the compiler has to be kept from reordering them.
            __builtin_amdgcn_wave_barrier();
            if (n_slots <= 64)
            {
                double value = mid + (add - pedestal);
                if (lane == 0) value = s_new;
                if (lane == n_slots - 1) value = e_new;
                if (n_slots == 1) value = 0.;
                const int slot = (lane == n_slots - 1) ? last_slot : first_slot + lane;
                if (lane < n_slots) a[slot] = value;
            }
            else
            {
                // Windows wider than 64 slots (cut_off > 30).
                for (int q = lane; q < n_slots; q += 64)
                {
                    const int slot = (q == n_slots - 1) ? last_slot : first_slot + q;
                    double value;
                    if (q == 0) value = s_new;
                    else if (q == n_slots - 1) value = e_new;
                    else value = a[slot] + (staged[r*slot_stride + q] - pedestal);
                    a[slot] = value;
                }
            }
            if (lane == 0 && bin_ok) bins[bin] = bin_old + pedestal;
            if (USE_LDS)
            {
                __builtin_amdgcn_wave_barrier();
            }
            else
            {
                __syncthreads();    // HBM fallback: wait for the stores before the next reads
            }
        }
        if (prefetch && base + kChainChunk < count)
        {
            park(((base/kChainChunk) & 1) ? stage_a : stage_b,
                 min(kChainChunk, count - base - kChainChunk));
        }
        __syncthreads();
    }
    if (WINDOW && USE_LDS && zone + lane <= n_cells) a[zone + lane] = window;
    __syncthreads();
    if (USE_LDS)
    {
        for (int s = lane; s < n_bins; s += 64) bin_sum[(long long)level*n_bins + s] = bins[s];
    }
}

// The serial chain by itself, one wavefront per level (engine option scan_chain = 0).
template <bool USE_LDS, bool WINDOW>
__global__ __launch_bounds__(64) void run_chain_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int slot_stride,
                                                       GridSpec g, int n_cells, int n_bins,
                                                       const RunMeta * __restrict__ runs,
                                                       const double * __restrict__ slot_sums,
                                                       double * __restrict__ global_slots,
                                                       double * __restrict__ bin_sum)
{
    extern __shared__ double chain_lds[];
    run_chain<USE_LDS, WINDOW>(blockIdx.x, run_count, max_runs, slot_stride, g, n_cells, n_bins,
                               runs, slot_sums, global_slots, bin_sum, chain_lds);
}

// ---------------------------------------------------------------------------------------------
// The relaxation in ONE launch (rounds 4-5: one launch per sweep, three to five of them, and the
// serial chain as a launch of its own behind them).
//
// One wavefront per 64 consecutive runs (lane = run), alive for all `sweeps` sweeps.  Inside the
// chunk the system is solved exactly, run by run (forward substitution: run t's value is broadcast,
// the later lanes whose masks name it add it to their sums -- ~20 instructions a step); what earlier
// chunks hold comes from their PREVIOUS sweep (sweep 0: zero): the stretch of runs from the first
// that can hold a slot of this chunk up to the chunk, staged through LDS, oldest first, every lane
// testing the run against its own two end slots.  So sweep k is exact for every chain of
// dependences that crosses at most k chunk boundaries, whatever its length inside a chunk (the runs
// of the 2 cut_off + 2 windows clipped at either end of the grid form such chains: each holds the
// end slot of all the others).  Sweeps >= 1 report a change; a sweep that changed nothing anywhere
// has verified a fixed point, i.e. the solution a serial evaluation of the same formula gives, bit
// for bit and independent of how it was reached.
//
// Between sweeps a chunk waits for exactly what it reads: progress[c'] >= k for the chunks c' of
// its stretch -- all EARLIER chunks, which the dispatcher started before it, so the wait ends.  It is
// bounded all the same (kSolveSpinLimit polls): a chunk whose wait runs out clears the level's
// flag [0], everybody stops waiting, and the serial chain takes the level.  Every sweep writes a
// buffer of its own (an earlier chunk may be a sweep ahead of a later one that still reads).
// After its last sweep the LAST run of every bin sums the bin's pedestals (its runs in row order,
// the earlier ones waited for like the stretch): bins without a run keep run_find_kernel's zero.
// The chunk that leaves last (per level: an atomic count) looks at the flags and, where the
// relaxation does not apply or has not settled, runs the serial chain itself.
// ---------------------------------------------------------------------------------------------
constexpr int kHistoryTile = 256;   // earlier runs staged in LDS at a time
constexpr int kSolveSpinLimit = 1 << 21;

__device__ __forceinline__ int load_agent(const int * p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A pedestal total another chunk (on any XCD, each with an L2 of its own) has written / will read
// during this launch: device-scope accesses, coherent by themselves.  The order against the count
// that announces them is kept by a workgroup-scope fence (the stores have completed before the
// count is stored), not by an agent-scope one -- that would write the XCD's whole L2 back, every
// sweep of every chunk, under the accumulate grid that is filling it (profiles/r06_ab_combine_in_kernel.txt
// is what such fences cost).
__device__ __forceinline__ double load_shared(const double * p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_shared(double * p, double value)
{
    __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits until the chunks [first, last) of the level have completed `needed` sweeps.  false: given up
// (this wait ran out, or somebody else's did: flags[0] == 0).
__device__ __forceinline__ bool wait_for_chunks(const int * progress, int first, int last,
                                                int needed, int * flags, int lane)
{
    for (int c0 = first; c0 < last; c0 += 64)
    {
        const int c = c0 + lane;
        int polls = 0;
        while (true)
        {
            const int done = c < last ? load_agent(&progress[c]) : needed;
            if (__ballot(done < needed) == 0ull) break;
            if (++polls > kSolveSpinLimit || load_agent(&flags[0]) == 0)
            {
                if (lane == 0) atomicAnd(&flags[0], 0);
                return false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    // (The values those chunks wrote before they raised their count are read by loads that go to
    // the device's coherence point themselves -- load_shared below --, like the counts: no cache
    // has to be written back or invalidated for them.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

template <bool WINDOW>
__global__ __launch_bounds__(64) void run_solve_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int max_chunks, int n_bins,
                                                       int sweeps,
                                                       const RunLink * __restrict__ links,
                                                       const int2 * __restrict__ run_slots,
                                                       const int * __restrict__ run_bin,
                                                       const int * __restrict__ bin_end,
                                                       const int * __restrict__ bin_first,
                                                       double * __restrict__ pedestals,
                                                       long long sweep_stride,
                                                       int * __restrict__ progress,
                                                       int * __restrict__ state,
                                                       double * __restrict__ bin_sum,
                                                       // the serial chain's own arguments
                                                       int slot_stride, GridSpec g, int n_cells,
                                                       const RunMeta * __restrict__ runs,
                                                       const double * __restrict__ slot_sums,
                                                       double * __restrict__ global_slots)
{
    extern __shared__ double chain_lds[];
    __shared__ double history_p[kHistoryTile];
    __shared__ int2 history_slots[kHistoryTile];
    __shared__ double own_p[64];
    const int level = blockIdx.y;
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    const int count = run_count[level];
    const int base = chunk*64;
    int * flags = state + level*kChainState;
    int * done = progress + (long long)level*max_chunks;
    const int chunks = (count + 63) >> 6;       // chunks of this level that hold a run
    if (base < count && load_agent(&flags[0]) != 0)
    {
        const int2 * slots_of = run_slots + (long long)level*max_runs;
        const int r = base + lane;
        const bool valid = r < count;
        RunLink mine;
        mine.ks = mine.ke = 0.;
        mine.in_s = mine.in_e = 0ull;
        mine.bin = -1; mine.n_slots = 0; mine.begin = base; mine.first_of_bin = 0;
        int2 ends = make_int2(-1, -1);
        if (valid)
        {
            mine = links[(long long)level*max_runs + r];
            ends = slots_of[r];
        }
        // The earliest run any lane of the chunk looks back to.
        int oldest = min(mine.begin, base);
        for (int offset = 32; offset > 0; offset >>= 1)
        {
            oldest = min(oldest, __shfl_xor(oldest, offset, 64));
        }
        const int last = min(64, count - base);
        bool waiting = true;
        double p = 0., given = 0.;
        for (int sweep = 0; sweep < sweeps && waiting; ++sweep)
        {
            const double * from = pedestals + (sweep > 0 ? sweep - 1 : 0)*sweep_stride +
                                  (long long)level*max_runs;
            double * to = pedestals + sweep*sweep_stride + (long long)level*max_runs;
            double before_s = 0., before_e = 0.;
            given = p;          // this lane's value of the sweep before
            if (sweep > 0 && base > 0)
            {
                waiting = wait_for_chunks(done, oldest >> 6, chunk, sweep, flags, lane);
                if (!waiting) break;
                for (int tile = oldest; tile < base; tile += kHistoryTile)
                {
                    const int length = min(kHistoryTile, base - tile);
                    __builtin_amdgcn_wave_barrier();        // the previous tile has been read
                    for (int t = lane; t < length; t += 64)
                    {
                        history_p[t] = load_shared(&from[tile + t]);
                        history_slots[t] = slots_of[tile + t];
                    }
                    __builtin_amdgcn_wave_barrier();        // one wavefront: LDS keeps program order
                    for (int t = 0; t < length; ++t)
                    {
                        const double value = history_p[t];
                        const int2 window = history_slots[t];
                        if (window.x <= ends.x && ends.x <= window.y) before_s += value;
                        if (window.x <= ends.y && ends.y <= window.y) before_e += value;
                    }
                }
            }
            // The chunk itself: run t of the chunk is bit lane-1-t of the later lanes' masks.
            p = 0.;
            for (int t = 0; t < last; ++t)
            {
                const double candidate = run_pedestal(mine.ks - before_s, mine.ke - before_e,
                                                      mine.n_slots);
                const double settled = read_lane(candidate, t);
                if (lane == t) p = candidate;
                const int j = lane - 1 - t;
                if (j >= 0)
                {
                    if ((mine.in_s >> j) & 1ull) before_s += settled;
                    if ((mine.in_e >> j) & 1ull) before_e += settled;
                }
            }
            if (valid) store_shared(&to[r], p);
            if (sweep == 0 && valid && mine.first_of_bin && mine.bin >= 0 && mine.bin < n_bins &&
                bin_end[(long long)level*n_bins + mine.bin] - r > kMaxStretch)
            {
                atomicAnd(&flags[0], 0);
            }
            if (sweep > 0)
            {
                const bool moved = valid && __double_as_longlong(p) != __double_as_longlong(given);
                if (__ballot(moved) != 0ull && lane == 0) atomicOr(&flags[sweep], 1);
            }
            // this sweep's values (and flags) before the count that says so
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&done[chunk], sweep + 1, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        }
        if (waiting && sweeps > 0)
        {
            // The bins whose last run lies in this chunk: their totals of the last sweep's values,
            // runs in row order from the bin's first (same order of additions as rounds 4-5).
            const double * final_p = pedestals + (sweeps - 1)*sweep_stride + (long long)level*max_runs;
            const int * bins = run_bin + (long long)level*max_runs;
            own_p[lane] = p;
            const bool closes = valid && mine.bin >= 0 && mine.bin < n_bins &&
                                bin_end[(long long)level*n_bins + mine.bin] == r + 1;
            const int first = closes ? bin_first[(long long)level*n_bins + mine.bin] : r;
            int earliest = min(first, base);
            for (int offset = 32; offset > 0; offset >>= 1)
            {
                earliest = min(earliest, __shfl_xor(earliest, offset, 64));
            }
            __builtin_amdgcn_wave_barrier();
            if (load_agent(&flags[0]) != 0 &&
                (earliest >= base || wait_for_chunks(done, earliest >> 6, chunk, sweeps, flags, lane)))
            {
                if (closes)
                {
                    double total = 0.;
                    for (int q = first; q <= r; ++q)
                    {
                        if (bins[q] == mine.bin) total += q >= base ? own_p[q - base] : load_shared(&final_p[q]);
                    }
                    store_shared(&bin_sum[(long long)level*n_bins + mine.bin], total);
                }
            }
        }
    }
    // Whoever leaves last looks at the flags (every chunk's are in by then: device-scope atomics,
    // like the bin totals above, completed before the chunk counts itself out) and, where the sweeps
    // did not apply or have not settled, runs the serial chain for the level.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int left = 0;
    if (lane == 0) left = atomicAdd(&flags[kStateFinished], 1);
    left = __builtin_amdgcn_readfirstlane(left);
    if (left != (int)gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bool settled = load_agent(&flags[0]) != 0;
    if (settled)
    {
        settled = false;
        for (int k = 1; k < sweeps; ++k)
        {
            if (load_agent(&flags[k]) == 0) settled = true;
        }
    }
    if (settled || count == 0) return;
    // (bin totals some chunks may have written from unsettled values: the chain starts from zero)
    for (int s = lane; s < n_bins; s += 64) bin_sum[(long long)level*n_bins + s] = 0.;
    __syncthreads();
    run_chain<false, WINDOW>(level, run_count, max_runs, slot_stride, g, n_cells, n_bins, runs,
                             slot_sums, global_slots, bin_sum, chain_lds);
}

}  // namespace lbl
