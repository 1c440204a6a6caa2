// Pedestal removal, first part: the runs of rows with one window (run_find: one single-pass scan
// in reference row order, with the running maximum of the runs' bins) and their profile sums on the
// window's slots (run_sums: one wavefront per run).  The factorisation is stated in pedestal.h,
// which includes this file after the workspace types; spectra.c:66-78 is the reference.
#pragma once

namespace lbl {

// A lane's value, the same in every lane (the lane index is wave-uniform): v_readlane.
__device__ __forceinline__ double read_lane(double value, int lane)
{
    const long long bits = __double_as_longlong(value);
    const int lo = __builtin_amdgcn_readlane((int)bits, lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// A row opens a run when its window is not empty and differs from the previous row's
// (an empty window in between also ends a run) -- and at every kRunCut-th row of the table: a
// run is one wavefront's work in run_sums_kernel, row after row, and the recurrence holds for any
// grouping of same-window rows, so thousands of lines in one window (a 4 M-line table has 8 000 to
// the wavenumber at a band centre: 2.5 ms for that one wavefront) become several runs side by side.
// `centre` receives the row's shifted line centre (its window's bin follows from it).
__device__ __forceinline__ int opens_run(const LineWing * __restrict__ wing,
                                         const int * __restrict__ sorted_of_row,
                                         long long r, long long n_lines, double & centre)
{
    centre = 0.;
    if (r >= n_lines) return 0;
    const LineWing w = wing[sorted_of_row[r]];
    centre = w.centre;
    if (w.first > w.last) return 0;
    if (r % kRunCut == 0) return 1;
    const LineWing p = wing[sorted_of_row[r - 1]];
    return (p.first == w.first && p.last == w.last) ? 0 : 1;
}

// The run-finding kernel uses 256-thread workgroups with a handful of registers so that it can be
// placed beside a resident accumulate grid (which leaves ~56 VGPRs and two wave slots per SIMD
// free); 1024-thread workgroups had to wait for it to drain.
constexpr int kScanThreads = 256;

// Inclusive scan over the workgroup of (sum of `value`, maximum of `top`), both at once.
__device__ __forceinline__ void block_scan_sum_max(int & value, int & top, int (*wave_part)[2],
                                                   int & block_sum, int & block_top)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int offset = 1; offset < 64; offset <<= 1)
    {
        const int up_value = __shfl_up(value, offset, 64);
        const int up_top = __shfl_up(top, offset, 64);
        if (lane >= offset)
        {
            value += up_value;
            top = max(top, up_top);
        }
    }
    if (lane == 63)
    {
        wave_part[wave][0] = value;
        wave_part[wave][1] = top;
    }
    __syncthreads();
    int before = 0, before_top = -1;
    block_sum = 0;
    block_top = -1;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i)
    {
        if (i < wave)
        {
            before += wave_part[i][0];
            before_top = max(before_top, wave_part[i][1]);
        }
        block_sum += wave_part[i][0];
        block_top = max(block_top, wave_part[i][1]);
    }
    __syncthreads();
    value += before;
    top = max(top, before_top);
}

// ---------------------------------------------------------------------------------------------
// The runs of every level in ONE launch (rounds 1-5: count / offset / compact, three launches, and a
// fourth for the prefix maxima of the runs' bins): a single-pass scan over the blocks of
// kScanThreads rows with decoupled look-back.  Every block publishes a 64-bit descriptor
//     status (2 bits: 0 nothing yet, 1 the block's own aggregate, 2 inclusive prefix)
//     | runs opened (31 bits) | largest bin + 1 (31 bits)
// first with its own aggregate, then -- once its first wavefront has walked back over the
// descriptors of the blocks before it, 64 at a time, adding aggregates until it meets an inclusive
// prefix -- with its inclusive prefix.  A block only ever waits for blocks with a smaller index in
// the same level, which the dispatcher started before it, so the wait ends; it is bounded all the
// same (kScanSpinLimit polls, ~1 s): a block that gives up publishes a poisoned count, which every
// later block inherits, and the host reads a run count it refuses (pedestal_finish).  Two
// descriptor arrays take turns: a launch works in one and clears the other for the next launch on
// this workspace (launches on one stream do not overlap).
// The kernel also resets what the later kernels of the pass fill: the level's chain state, and
// bin_end / bin_first / bin_sum (bins without a run keep zero).
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kScanAggregate = 1ull, kScanInclusive = 2ull;
constexpr int kScanPoison = 0x40000000;
constexpr int kScanSpinLimit = 1 << 22;

__device__ __forceinline__ unsigned long long scan_word(unsigned long long status, int count,
                                                        int top)
{
    return (status << 62) | ((unsigned long long)(unsigned)min(count, kScanPoison) << 31) |
           (unsigned long long)(unsigned)(top + 1);
}

__device__ __forceinline__ int scan_count(unsigned long long word)
{
    return (int)((word >> 31) & 0x7fffffffull);
}

__device__ __forceinline__ int scan_top(unsigned long long word)
{
    return (int)(word & 0x7fffffffull) - 1;
}

__device__ __forceinline__ unsigned long long load_agent(const unsigned long long * p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_agent(unsigned long long * p, unsigned long long value)
{
    __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kChainState = 16;     // ints of chain state per level, see pedestal_chain.h

__global__ __launch_bounds__(kScanThreads) void run_find_kernel(
    const LineWing * __restrict__ wing, const int * __restrict__ sorted_of_row, long long n_lines,
    int n_blocks, int bin_origin, unsigned long long * __restrict__ scan_now,
    unsigned long long * __restrict__ scan_next, long long scan_capacity,
    int * __restrict__ run_start, int * __restrict__ prefix_bin, int * __restrict__ run_count,
    int n_bins, int * __restrict__ bin_end, int * __restrict__ bin_first,
    double * __restrict__ bin_sum, int * __restrict__ state, int start_state)
{
    __shared__ int wave_part[kScanThreads/64][2];
    __shared__ int before[2];
    const int level = blockIdx.y, block = blockIdx.x;
    // Housekeeping for the rest of the pass and for the next launch.
    {
        const long long threads = (long long)gridDim.x*gridDim.y*kScanThreads;
        const long long me = ((long long)level*n_blocks + block)*kScanThreads + threadIdx.x;
        for (long long i = me; i < scan_capacity; i += threads) scan_next[i] = 0ull;
        for (int i = block*kScanThreads + threadIdx.x; i < n_bins; i += n_blocks*kScanThreads)
        {
            bin_end[(long long)level*n_bins + i] = 0;
            bin_first[(long long)level*n_bins + i] = 0x7fffffff;
            bin_sum[(long long)level*n_bins + i] = 0.;
        }
        if (block == 0 && threadIdx.x < kChainState)
        {
            state[level*kChainState + threadIdx.x] = threadIdx.x == 0 ? start_state : 0;
        }
    }
    const long long r = (long long)block*kScanThreads + threadIdx.x;
    double centre;
    const int flag = opens_run(wing + (long long)level*n_lines, sorted_of_row, r, n_lines, centre);
    // (the run's bin as run_sums_kernel forms it, RunMeta::bin, kept inside what a descriptor holds)
    const int bin = flag ? min(max((int)floor(centre) - bin_origin, -1), kScanPoison) : -1;
    int inclusive = flag, top = bin, total, block_top;
    block_scan_sum_max(inclusive, top, wave_part, total, block_top);
    if (threadIdx.x < 64)
    {
        const int lane = threadIdx.x;
        unsigned long long * descriptors = scan_now + (long long)level*n_blocks;
        int count_before = 0, top_before = -1;
        if (block > 0)
        {
            if (lane == 0) store_agent(&descriptors[block], scan_word(kScanAggregate, total, block_top));
            bool gave_up = false;
            for (int base = block - 1; base >= 0; base -= 64)
            {
                const int at = base - lane;
                unsigned long long word = 0ull;
                int polls = 0;
                while (true)
                {
                    word = at >= 0 ? load_agent(&descriptors[at]) : (kScanAggregate << 62) | 0ull;
                    if (__ballot((word >> 62) == 0ull) == 0ull) break;
                    if (++polls > kScanSpinLimit)
                    {
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (gave_up) break;
                const unsigned long long closed = __ballot(at >= 0 && (word >> 62) == kScanInclusive);
                const int stop = closed != 0ull ? __builtin_ctzll(closed) : 63;
                const bool take = at >= 0 && lane <= stop;
                int count = take ? scan_count(word) : 0;
                int best = take ? scan_top(word) : -1;
                for (int offset = 32; offset > 0; offset >>= 1)
                {
                    count = min(count + __shfl_xor(count, offset, 64), kScanPoison);
                    best = max(best, __shfl_xor(best, offset, 64));
                }
                count_before = min(count_before + count, kScanPoison);
                top_before = max(top_before, best);
                if (closed != 0ull) break;
            }
            if (gave_up) count_before = kScanPoison;
        }
        if (lane == 0)
        {
            store_agent(&descriptors[block],
                        scan_word(kScanInclusive, count_before + total, max(top_before, block_top)));
            before[0] = count_before;
            before[1] = top_before;
        }
    }
    __syncthreads();
    const long long at = (long long)before[0] + inclusive - 1;
    if (flag && at < n_lines)
    {
        run_start[(long long)level*n_lines + at] = (int)r;
        prefix_bin[(long long)level*n_lines + at] = max(before[1], top);
    }
    if (block == n_blocks - 1 && threadIdx.x == 0)
    {
        run_count[level] = min(before[0] + total, kScanPoison);
    }
}

__device__ __forceinline__ int slot_point(int slot, int n_cells, int n_per_v, int n)
{
    return slot < n_cells ? slot*n_per_v : n - 1;
}

// One wavefront per run (grid-stride over runs): evaluates every row of the run on the
// run's slots (lane = slot) with the same profile code the accumulate kernel uses.
// The rows of a run are taken 64 at a time: lane i fetches row i's records (index, LineWing,
// LineCore: three dependent loads, side by side for 64 rows) into LDS, then every lane walks the
// staged rows in the reference's row order.  (Walking the rows straight from HBM put those three
// round trips on every row: 80-130 us for the benchmark's tables, most of it waiting.)
// A slot inside a row's core range takes the region chain (wells_profile: a few hundred
// instructions against a dozen for the far wing).  Slots are whole wavenumbers, so that is one
// lane in every third row or so -- and the wavefront paid the chain for each such row.  Those
// (row, lane) pairs are set aside and evaluated together, one pair per lane, once per batch of
// rows; every lane then adds its own in row order.
struct StagedRow
{
    double centre, g2, bl;
    double repwid, y, amp;
    int first, last, core_first, core_last;
};

constexpr int kCorePairs = 256;     // (row, lane) pairs set aside before they are evaluated

__global__ __launch_bounds__(64) void run_sums_kernel(const LineWing * __restrict__ wing,
                                                      const LineCore * __restrict__ core,
                                                      const int * __restrict__ sorted_of_row,
                                                      long long n_lines, GridSpec g, int n_cells,
                                                      const int * __restrict__ run_start,
                                                      const int * __restrict__ run_count,
                                                      int max_runs, int slot_stride,
                                                      RunMeta * __restrict__ runs,
                                                      double * __restrict__ slot_sums)
{
    __shared__ StagedRow staged[64];
    __shared__ unsigned short pair_of[kCorePairs];      // row << 6 | lane
    __shared__ double pair_value[kCorePairs];
    const int level = blockIdx.y;
    const int lane = threadIdx.x;
    const int count = run_count[level];
    const LineWing * w = wing + (long long)level*n_lines;
    const LineCore * c = core + (long long)level*n_lines;
    const int * starts = run_start + (long long)level*n_lines;
    for (int run = blockIdx.x; run < count; run += gridDim.x)
    {
        const int row_begin = starts[run];
        const int row_end = run + 1 < count ? starts[run + 1] : (int)n_lines;
        const LineWing head = w[sorted_of_row[row_begin]];
        const int first_slot = head.first/g.n_per_v;
        const int last_int = head.last/g.n_per_v;
        const bool extra = (last_int*g.n_per_v != head.last);
        const int n_slots = last_int - first_slot + 1 + (extra ? 1 : 0);
        double * sums = slot_sums + ((long long)level*max_runs + run)*slot_stride;
        // Slots of this lane in the passes q0 = 0, 64, ... (one pass unless cut_off > 30).
        const int passes = (n_slots + 63)/64;
        double vs = 0., ve = 0.;
        for (int pass = 0; pass < passes; ++pass)
        {
            const int q0 = pass*64;
            auto point_of = [&](int in_pass) {
                const int q = q0 + in_pass;
                const int slot = (extra && q == n_slots - 1) ? n_cells : first_slot + q;
                return slot_point(q < n_slots ? slot : first_slot, n_cells, g.n_per_v, g.n);
            };
            auto wavenumber_of = [&](int point) {
                const double step = (double)point*g.dv;        // absorption.c:39
                return (double)g.v0 + step;
            };
            const bool active = q0 + lane < n_slots;
            const int point = point_of(lane);
            const double v = wavenumber_of(point);
            double total = 0.;
            bool open = true;
            for (int base = row_begin; base < row_end && open; base += 64)
            {
                const int rows = min(64, row_end - base);
                __builtin_amdgcn_wave_barrier();    // the previous batch has been read
                // Lane r keeps row r of the batch in registers as well: the row loop below takes
                // what it needs of a row from there by v_readlane (wave-uniform, into scalar
                // registers) and waits for no LDS round trip; the LDS copy serves settle_pairs,
                // where every lane wants a different row.
                StagedRow row;
                row.centre = 0.; row.g2 = 1.; row.bl = 0.; row.repwid = 1.; row.y = 0.; row.amp = 0.;
                row.first = -1; row.last = -2; row.core_first = 0; row.core_last = -1;
                if (lane < rows)
                {
                    const int j = sorted_of_row[base + lane];
                    const LineWing l = w[j];
                    const LineCore k = c[j];
                    row.centre = l.centre; row.g2 = l.g2; row.bl = l.bl;
                    row.repwid = k.repwid; row.y = k.y; row.amp = k.amp;
                    row.first = l.first; row.last = l.last;
                    row.core_first = k.core_first; row.core_last = k.core_last;
                    staged[lane] = row;
                }
                __builtin_amdgcn_wave_barrier();    // one wavefront: LDS keeps program order
                int n_pairs = 0;
                // The pairs set aside so far: the region chain for 64 of them at a time, then
                // every lane adds its own (in the order they were set aside: row order).
                auto settle_pairs = [&]() {
                    for (int e0 = 0; e0 < n_pairs; e0 += 64)
                    {
                        if (e0 + lane < n_pairs)
                        {
                            const int pair = pair_of[e0 + lane];
                            const StagedRow l = staged[pair >> 6];
                            const double d = wavenumber_of(point_of(pair & 63)) - l.centre;
                            pair_value[e0 + lane] = l.amp*wells_profile(d*l.repwid, l.y);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int e = 0; e < n_pairs; ++e)
                    {
                        if ((pair_of[e] & 63) == lane) total += pair_value[e];
                    }
                    __builtin_amdgcn_wave_barrier();
                    n_pairs = 0;
                };
                for (int r = 0; r < rows; ++r)
                {
                    if (__builtin_amdgcn_readlane(row.first, r) != head.first ||
                        __builtin_amdgcn_readlane(row.last, r) != head.last)
                    {
                        open = false;
                        break;      // an empty or different window ends the run
                    }
                    const int core_first = __builtin_amdgcn_readlane(row.core_first, r);
                    const int core_last = __builtin_amdgcn_readlane(row.core_last, r);
                    const bool in_core = active && point >= core_first && point <= core_last;
                    const double d = v - read_lane(row.centre, r);
                    const double far_wing = read_lane(row.bl, r)*
                                            rcp_newton(__builtin_fma(d, d, read_lane(row.g2, r)));
                    total += in_core ? 0. : far_wing;
                    const unsigned long long cores = __ballot(in_core);
                    if (cores != 0ull)
                    {
                        if (in_core)
                        {
                            const int at = n_pairs + __builtin_popcountll(cores & ((1ull << lane) - 1ull));
                            pair_of[at] = (unsigned short)(r << 6 | lane);
                        }
                        n_pairs += __builtin_popcountll(cores);
                        if (n_pairs > kCorePairs - 64)
                        {
                            __builtin_amdgcn_wave_barrier();
                            settle_pairs();
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (n_pairs > 0) settle_pairs();
            }
            if (active) sums[q0 + lane] = total;
            // The run's own values on its end slots are those lanes' totals.
            if (q0 == 0) vs = __shfl(total, 0, 64);
            if (q0 + 64 >= n_slots) ve = __shfl(total, n_slots - 1 - q0, 64);
        }
        if (lane == 0)
        {
            RunMeta meta;
            meta.row_begin = row_begin;
            meta.first = head.first;
            meta.last = head.last;
            meta.n_slots = n_slots;
            meta.vs = vs;
            meta.ve = ve;
            meta.bin = (int)floor(head.centre) - (g.v0 - g.cut_off - 1);
            meta.first_slot = first_slot;
            meta.last_slot = extra ? n_cells : last_int;
            meta.pad = 0;
            runs[(long long)level*max_runs + run] = meta;
        }
    }
}

}  // namespace lbl
