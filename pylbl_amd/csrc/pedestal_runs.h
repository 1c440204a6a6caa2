// Pedestal removal, first part: the runs of rows with one window (run_count / run_offset /
// run_compact: a three-pass scan in reference row order) and their profile sums on the window's
// slots (run_sums: one wavefront per run).  The factorisation is stated in pedestal.h, which
// includes this file after the workspace types; spectra.c:66-78 is the reference.
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
__device__ __forceinline__ int opens_run(const LineWing * __restrict__ wing,
                                         const int * __restrict__ sorted_of_row,
                                         long long r, long long n_lines)
{
    if (r >= n_lines) return 0;
    const LineWing w = wing[sorted_of_row[r]];
    if (w.first > w.last) return 0;
    if (r % kRunCut == 0) return 1;
    const LineWing p = wing[sorted_of_row[r - 1]];
    return (p.first == w.first && p.last == w.last) ? 0 : 1;
}

__device__ __forceinline__ int block_inclusive_scan(int value, int * wave_total, int & block_total)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int scan = value;
    for (int offset = 1; offset < 64; offset <<= 1)
    {
        const int up = __shfl_up(scan, offset, 64);
        if (lane >= offset) scan += up;
    }
    if (lane == 63) wave_total[wave] = scan;
    __syncthreads();
    int before = 0;
    block_total = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i)
    {
        if (i < wave) before += wave_total[i];
        block_total += wave_total[i];
    }
    __syncthreads();
    return before + scan;
}

// The three run-finding kernels use 256-thread workgroups with a handful of registers so that
// they can be placed beside a resident accumulate grid (which leaves ~56 VGPRs and two wave
// slots per SIMD free); 1024-thread workgroups had to wait for it to drain.
constexpr int kScanThreads = 256;

// Pass 1: runs opened inside every block of kScanThreads rows.
__global__ __launch_bounds__(kScanThreads) void run_count_kernel(const LineWing * __restrict__ wing,
                                                         const int * __restrict__ sorted_of_row,
                                                         long long n_lines, int n_blocks,
                                                         int * __restrict__ block_count)
{
    __shared__ int wave_total[16];
    const int level = blockIdx.y;
    const long long r = (long long)blockIdx.x*kScanThreads + threadIdx.x;
    const int flag = opens_run(wing + (long long)level*n_lines, sorted_of_row, r, n_lines);
    int total;
    block_inclusive_scan(flag, wave_total, total);
    if (threadIdx.x == 0) block_count[(long long)level*n_blocks + blockIdx.x] = total;
}

// Pass 2 (one block per level): exclusive scan of the block counts, in place.
__global__ __launch_bounds__(kScanThreads) void run_offset_kernel(int n_blocks, int * __restrict__ block_count,
                                                          int * __restrict__ run_count)
{
    __shared__ int wave_total[16];
    __shared__ int carry;
    int * counts = block_count + (long long)blockIdx.x*n_blocks;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += kScanThreads)
    {
        const int i = base + threadIdx.x;
        const int value = i < n_blocks ? counts[i] : 0;
        int total;
        const int inclusive = block_inclusive_scan(value, wave_total, total);
        const int before = carry;
        if (i < n_blocks) counts[i] = before + inclusive - value;
        __syncthreads();
        if (threadIdx.x == 0) carry = before + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) run_count[blockIdx.x] = carry;
}

// Pass 3: rows that open a run, compacted in row order.
__global__ __launch_bounds__(kScanThreads) void run_compact_kernel(const LineWing * __restrict__ wing,
                                                           const int * __restrict__ sorted_of_row,
                                                           long long n_lines, int n_blocks,
                                                           const int * __restrict__ block_offset,
                                                           int * __restrict__ run_start)
{
    __shared__ int wave_total[16];
    const int level = blockIdx.y;
    const long long r = (long long)blockIdx.x*kScanThreads + threadIdx.x;
    const int flag = opens_run(wing + (long long)level*n_lines, sorted_of_row, r, n_lines);
    int total;
    const int inclusive = block_inclusive_scan(flag, wave_total, total);
    if (flag)
    {
        const int at = block_offset[(long long)level*n_blocks + blockIdx.x] + inclusive - 1;
        run_start[(long long)level*n_lines + at] = (int)r;
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
