// Pedestal removal ("K3").
//
// Reference statement: pyLBL/c_lib/spectra.c:66-78.  After line j has been added to k[],
// the reference subtracts  p_j = min(k[first_j], k[last_j])  from its whole window -- taken
// from the ACCUMULATED spectrum, so p_j depends on every earlier row.  A literal
// implementation is a serial sweep of the full-resolution window per line.
//
// Factorisation used here (exact in real arithmetic):
//  * Windows begin and end on integer wavenumbers (or on the last grid point n-1), so the
//    recurrence only ever looks at k on those "slots": v0+c for c = 0..cells-1, plus n-1.
//  * Rows that follow each other with the SAME window form a run.  Inside a run only the
//    two end slots matter, and after every subtraction one of them is exactly zero, so the
//    state collapses to their difference:  delta <- delta + (V_i(first) - V_i(last)).
//    Hence for a run with sums VS = sum V_i(first), VE = sum V_i(last), entered with
//    end-slot values (a_s, a_e):  k_s = a_s + VS,  k_e = a_e + VE,
//        a_s' = max(k_s - k_e, 0),   a_e' = max(k_e - k_s, 0),
//        sum of the run's pedestals  P = k_s - a_s' = k_e - a_e' = min(k_s, k_e),
//    taken from the smaller side (the identity that subtracts nothing), so that a line
//    peak sitting on the other end slot does not cost the small side its accuracy;
//    every interior slot c receives  sum_i V_i(c) - P.
//  * The spectrum is then  sum_j V_j(x) - sum_j p_j [x in window_j]; the second sum is
//    piecewise constant between integer wavenumbers, so the accumulate kernel subtracts one
//    table value per grid point in its epilogue.
//
// Kernels: run_count / run_offset / run_compact (find runs in reference row order: a
// three-pass parallel scan), run_sums (one wavefront per run: profile values on the run's
// slots, in parallel over runs), run_prefix / run_links (the stretch of earlier runs that can
// hold a run's end slots, and what they added there), run_relax (the recurrence in the runs'
// pedestal totals as a triangular system: a few launches, each exact inside chunks of 64 runs), run_chain (one
// wavefront per level: the serial form, for the levels the relaxation leaves),
// pedestal_tables (per 1 cm-1 cell: total pedestal covering its interior / integer point).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "line_prep.h"
#include "tile_schedule.h"
#include "wave_ops.h"
#include "voigt_profile.h"

namespace lbl {

struct RunMeta
{
    int row_begin;      // first row (reference order) of the run
    int first, last;    // the run's window (grid indices, inclusive)
    int n_slots;        // integer slots first/npv .. last/npv (+1 if last is not an integer point)
    double vs;          // sum over the run of V_i(first)
    double ve;          // sum over the run of V_i(last)
    int bin;            // floor(centre) - (v0 - cut_off - 1): identifies the unclipped window
    int first_slot;     // first/n_per_v
    int last_slot;      // slot of `last`: last/n_per_v, or n_cells when last is the point n-1
    int pad;
};

struct alignas(16) RunLink
{
    double ks;          // GS + VS: first-slot candidate before subtracting earlier pedestals
    double ke;          // GE + VE: last-slot candidate
    unsigned long long in_s;    // bit j: run r-1-j, of the same chunk of 64 runs, holds this run's first slot
    unsigned long long in_e;    // ... this run's last slot
    int bin;            // the run's window (RunMeta::bin)
    int n_slots;
    int begin;          // the first earlier run that can hold this run's first slot
    int first_of_bin;   // 1: no earlier run has this bin
};
static_assert(sizeof(RunLink) == 48, "RunLink");

template <typename T>
struct RawBuffer
{
    T * data = nullptr;
    size_t capacity = 0;
    void reserve(size_t count)
    {
        if (count <= capacity) return;
        if (data != nullptr) (void)hipFree(data);
        data = nullptr;
        capacity = 0;
        if (hipMalloc(reinterpret_cast<void **>(&data), count*sizeof(T)) != hipSuccess)
        {
            throw std::runtime_error("hipMalloc failed in the pedestal workspace.");
        }
        capacity = count;
    }
    ~RawBuffer() { if (data != nullptr) (void)hipFree(data); }
    RawBuffer() = default;
    RawBuffer(const RawBuffer &) = delete;
    RawBuffer & operator=(const RawBuffer &) = delete;
};

struct PedestalWorkspace
{
    RawBuffer<int> block_count;     // [levels][blocks of kScanThreads rows]
    RawBuffer<int> run_start;       // [levels][n_lines]
    RawBuffer<int> run_count;       // [levels]
    RawBuffer<RunMeta> runs;        // [levels][max_runs]
    RawBuffer<double> slot_sums;    // [levels][max_runs][slot_stride]
    RawBuffer<RunLink> links;       // [levels][max_runs]
    RawBuffer<double> pedestals[2]; // [levels][max_runs]: the relaxation's two sets of values
    RawBuffer<int2> run_slots;      // [levels][max_runs]: the runs' end slots, packed for the relaxation
    RawBuffer<int> run_bin;         // [levels][max_runs]
    RawBuffer<int> bin_end;         // [levels][bins]: 1 + the last run of every bin
    RawBuffer<int> prefix_bin;      // [levels][max_runs]: running maximum of the runs' bins
    RawBuffer<int> state;           // [levels][kChainState], see run_prefix_kernel
    RawBuffer<double> slots;        // [levels][cells+1]       (only when LDS is too small)
    RawBuffer<double> bin_sum;      // [levels][cells+2*cut+3]
    RawBuffer<double> cell_sum;     // [levels][cells]
    RawBuffer<double> point_sum;    // [levels][cells]
    std::vector<int> host_counts;
};

constexpr int kRunCut = 1024;     // see opens_run

inline long long pedestal_bytes_per_level(long long n_lines, int n_cells, int cut_off)
{
    const long long stride = 2*cut_off + 3;
    const long long runs = std::min<long long>(n_lines, 4ll*(n_cells + 2*cut_off + 2) + n_lines/kRunCut);
    return n_lines*4 + runs*((long long)(sizeof(RunMeta) + sizeof(RunLink)) + stride*8 + 40) +
           4ll*(n_cells + stride)*8;
}

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
// long.  run_relax_kernel: one wavefront per 64 consecutive runs (lane = run), the chunk solved
// exactly by forward substitution, earlier chunks' values taken from the previous launch.  The
// second and later launches report whether anything changed, and a launch that changed nothing
// has verified a fixed point, i.e. the solution a serial evaluation of the same formula gives, bit
// for bit and independent of how it was reached.  Levels that have not settled after the last
// launch keep run_chain_kernel.  ~25 us for the 400 k-line benchmark table, where the serial
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
// launches is a chain that matters ACROSS chunks: one launch per boundary it crosses.
// ---------------------------------------------------------------------------------------
// Per level: [0] 1 while the relaxation applies (cleared by run_links_kernel where rows are too far
// out of order, by the first relaxation launch where one window's runs are spread over more than
// kMaxStretch runs: its total would be one lane's walk of thousands), [k] something
// changed in relaxation launch k (k = 1 .. launches-1), [7] the number of launches queued.
constexpr int kChainState = 8;
constexpr int kMaxRelaxLaunches = 7;
constexpr int kMaxStretch = 1024;
constexpr int kMaxHistory = 16384;  // earlier runs a run may have to look at (a 4 M-line table: 1 800)

// A launch that changed nothing has verified the values it was handed.
__device__ __forceinline__ bool chain_verified_before(const int * state, int launch)
{
    for (int k = 1; k < launch; ++k)
    {
        if (state[k] == 0) return true;
    }
    return false;
}

__device__ __forceinline__ bool chain_settled(const int * state)
{
    return state[0] != 0 && chain_verified_before(state, state[kChainState - 1]);
}

// prefix_bin[r] = max over q <= r of the runs' bins (one workgroup per level: every thread takes
// a contiguous share of the runs, the shares' maxima are scanned, the shares written back).
// Also resets the level's chain state and clears what the relaxation fills: bin_end (1 + the
// last run of every bin, run_links_kernel) and bin_sum (bins without a run keep zero).
__global__ __launch_bounds__(kScanThreads) void run_prefix_kernel(const int * __restrict__ run_count,
                                                          int max_runs, int n_bins,
                                                          const RunMeta * __restrict__ runs,
                                                          int * __restrict__ prefix_bin,
                                                          int * __restrict__ bin_end,
                                                          double * __restrict__ bin_sum,
                                                          int * __restrict__ state, int start_state,
                                                          int launches)
{
    __shared__ int wave_max[kScanThreads/64];
    const int level = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int count = run_count[level];
    const RunMeta * meta = runs + (long long)level*max_runs;
    int * out = prefix_bin + (long long)level*max_runs;
    if (start_state != 0)
    {
        for (int b = threadIdx.x; b < n_bins; b += kScanThreads)
        {
            bin_end[(long long)level*n_bins + b] = 0;
            bin_sum[(long long)level*n_bins + b] = 0.;
        }
    }
    if (threadIdx.x < kChainState)
    {
        state[level*kChainState + threadIdx.x] = threadIdx.x == 0 ? start_state
                                                 : threadIdx.x == kChainState - 1 ? launches : 0;
    }
    const int share = (count + kScanThreads - 1)/kScanThreads;
    const int begin = min(threadIdx.x*share, count), end = min(begin + share, count);
    int mine = -1;
    for (int r = begin; r < end; ++r) mine = max(mine, meta[r].bin);
    int scan = mine;
    for (int offset = 1; offset < 64; offset <<= 1)
    {
        const int up = __shfl_up(scan, offset, 64);
        if (lane >= offset) scan = max(scan, up);
    }
    if (lane == 63) wave_max[wave] = scan;
    __syncthreads();
    int before = -1;
    for (int i = 0; i < wave; ++i) before = max(before, wave_max[i]);
    const int up = __shfl_up(scan, 1, 64);
    int running = max(before, lane > 0 ? up : -1);      // everything before this thread's share
    for (int r = begin; r < end; ++r)
    {
        running = max(running, meta[r].bin);
        out[r] = running;
    }
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
                                                       int * __restrict__ bin_end,
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
    const int * prefix = prefix_bin + (long long)level*max_runs;
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
            if (bin_ok) atomicMax(&bin_end[(long long)level*n_bins + m.bin], r + 1);
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

constexpr int kHistoryTile = 256;   // earlier runs staged in LDS at a time

// One wavefront per 64 consecutive runs (lane = run).  Inside the chunk the system is solved
// exactly, run by run (forward substitution: run t's value is broadcast, the later lanes whose
// masks name it add it to their sums -- ~20 instructions a step); what earlier chunks hold comes
// from the previous launch (launch 0: zero): the stretch of runs from the first that can hold a
// slot of this chunk up to the chunk, staged through LDS, oldest first, every lane testing the run
// against its own two end slots.  So launch k is exact for every chain of dependences that
// crosses at most k chunk boundaries, whatever its length inside a chunk (the runs of the
// 2 cut_off + 2 windows clipped at either end of the grid form such chains: each holds the end
// slot of all the others).  Launches >= 1 report a change and sum the bins' totals from the
// values they were handed -- final if the launch changes nothing anywhere.
__global__ __launch_bounds__(64) void run_relax_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int n_bins, int launch,
                                                       const RunLink * __restrict__ links,
                                                       const int2 * __restrict__ run_slots,
                                                       const int * __restrict__ run_bin,
                                                       const int * __restrict__ bin_end,
                                                       const double * __restrict__ p_in,
                                                       double * __restrict__ p_out,
                                                       int * __restrict__ state,
                                                       double * __restrict__ bin_sum)
{
    __shared__ double history_p[kHistoryTile];
    __shared__ int2 history_slots[kHistoryTile];
    const int level = blockIdx.y;
    const int lane = threadIdx.x;
    const int count = run_count[level];
    const int base = blockIdx.x*64;
    int * flags = state + level*kChainState;
    if (base >= count || flags[0] == 0) return;
    if (chain_verified_before(flags, launch)) return;       // an earlier launch changed nothing
    const double * from = p_in + (long long)level*max_runs;
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
    double given = 0., before_s = 0., before_e = 0.;
    if (launch > 0 && valid) given = from[r];
    if (launch > 0 && base > 0)
    {
        // The earliest run any lane of the chunk looks back to.
        int oldest = min(mine.begin, base);
        for (int offset = 32; offset > 0; offset >>= 1)
        {
            oldest = min(oldest, __shfl_xor(oldest, offset, 64));
        }
        for (int tile = oldest; tile < base; tile += kHistoryTile)
        {
            const int length = min(kHistoryTile, base - tile);
            __builtin_amdgcn_wave_barrier();        // the previous tile has been read
            for (int t = lane; t < length; t += 64)
            {
                history_p[t] = from[tile + t];
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
    double p = 0.;
    const int last = min(64, count - base);
    for (int t = 0; t < last; ++t)
    {
        const double candidate = run_pedestal(mine.ks - before_s, mine.ke - before_e, mine.n_slots);
        const double settled = read_lane(candidate, t);
        if (lane == t) p = candidate;
        const int j = lane - 1 - t;
        if (j >= 0)
        {
            if ((mine.in_s >> j) & 1ull) before_s += settled;
            if ((mine.in_e >> j) & 1ull) before_e += settled;
        }
    }
    if (valid) p_out[(long long)level*max_runs + r] = p;
    if (launch == 0 && valid && mine.first_of_bin && mine.bin >= 0 && mine.bin < n_bins &&
        bin_end[(long long)level*n_bins + mine.bin] - r > kMaxStretch)
    {
        atomicAnd(&flags[0], 0);
    }
    if (launch > 0)
    {
        const bool moved = valid && __double_as_longlong(p) != __double_as_longlong(given);
        if (__ballot(moved) != 0ull && lane == 0)
        {
            atomicOr(&flags[launch], 1);        // (run_prefix_kernel cleared the flags)
        }
        if (valid && mine.first_of_bin && mine.bin >= 0 && mine.bin < n_bins)
        {
            // The bin's total of the values handed in: its runs in row order, up to its last.
            const int * bins = run_bin + (long long)level*max_runs;
            const int end = bin_end[(long long)level*n_bins + mine.bin];
            double total = 0.;
            for (int q = r; q < end; ++q)
            {
                if (bins[q] == mine.bin) total += from[q];
            }
            bin_sum[(long long)level*n_bins + mine.bin] = total;
        }
    }
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
__global__ __launch_bounds__(64) void run_chain_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int slot_stride,
                                                       GridSpec g, int n_cells, int n_bins,
                                                       const RunMeta * __restrict__ runs,
                                                       const double * __restrict__ slot_sums,
                                                       const int * __restrict__ state,
                                                       double * __restrict__ global_slots,
                                                       double * __restrict__ bin_sum)
{
    extern __shared__ double lds[];
    const int level = blockIdx.x;
    const int lane = threadIdx.x;
    if (state != nullptr && chain_settled(state + level*kChainState)) return;   // relaxation did it
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

// One thread per 1 cm-1 cell: the interior points of cell c lie in the windows of bins
// b = c+v0-cut .. c+v0+cut, its integer point also in bin c+v0-cut-1 (the window that closes
// there).  Sums of non-negative totals in a fixed order: reproducible, and exactly zero where
// no line reaches.
__global__ __launch_bounds__(256) void pedestal_tables_kernel(GridSpec g, int n_cells, int n_bins,
                                                              const double * __restrict__ bin_sum,
                                                              double * __restrict__ cell_sum,
                                                              double * __restrict__ point_sum)
{
    const int level = blockIdx.y;
    const int cell = blockIdx.x*blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const double * bins = bin_sum + (long long)level*n_bins;
    double interior = 0.;
    for (int k = 1; k <= 2*g.cut_off + 1; ++k)
    {
        interior += bins[cell + k];
    }
    cell_sum[(long long)level*n_cells + cell] = interior;
    point_sum[(long long)level*n_cells + cell] = interior + bins[cell];
}

// k = (sums - pedestal total of the windows holding the point) [* number density] [+ k].
// Windows start and end on integer wavenumbers, so the total is constant inside a 1 cm-1
// cell and has one extra bin of lines on the integer point that closes a window.
__global__ __launch_bounds__(256) void pedestal_apply_kernel(const double * __restrict__ sums,
                                                             long long sums_stride,
                                                             double * __restrict__ out,
                                                             long long out_stride,
                                                             const double * __restrict__ cell_sum,
                                                             const double * __restrict__ point_sum,
                                                             const LevelScalars * __restrict__ levels,
                                                             int first, int end, int n_per_v,
                                                             int n_cells, int scale_density,
                                                             int accumulate)
{
    // Points [first, end): the whole grid, or the columns of one piece of a streamed call.
    const int level = blockIdx.y;
    const int i = first + blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= end) return;
    const int cell = i/n_per_v;
    const bool on_integer = (cell*n_per_v == i);
    const double * table = on_integer ? point_sum : cell_sum;
    double value = sums[(long long)level*sums_stride + i] - table[(long long)level*n_cells + cell];
    if (scale_density) value *= levels[level].density;
    double * k = out + (long long)level*out_stride;
    if (accumulate) value += k[i];
    k[i] = value;
}

// The pedestal pre-pass for `count` levels whose LineWing/LineCore arrays are already in
// HBM, in two halves (pedestal_find_runs, pedestal_finish) on the same stream.
inline void pedestal_check(hipError_t status, const char * what)
{
    if (status != hipSuccess)
    {
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(status));
    }
}

// First half: finds the runs (three short scan kernels) and starts the copy of the run
// counts to the host.  The engine orders them before the accumulate launch of the same call
// (a resident accumulate grid of another call does not hold them up, see kScanThreads).
inline void pedestal_find_runs(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                               const LineWing * wing, int count)
{
    auto check = pedestal_check;
    const long long n_lines = t.n_lines;
    const int n_blocks = (int)((n_lines + kScanThreads - 1)/kScanThreads);
    ws.block_count.reserve((size_t)count*n_blocks);
    ws.run_start.reserve((size_t)(count*n_lines));
    ws.run_count.reserve((size_t)count);
    hipLaunchKernelGGL(run_count_kernel, dim3(n_blocks, count), dim3(kScanThreads), 0, stream, wing,
                       t.sorted_of_row, n_lines, n_blocks, ws.block_count.data);
    hipLaunchKernelGGL(run_offset_kernel, dim3(count), dim3(kScanThreads), 0, stream, n_blocks,
                       ws.block_count.data, ws.run_count.data);
    hipLaunchKernelGGL(run_compact_kernel, dim3(n_blocks, count), dim3(kScanThreads), 0, stream, wing,
                       t.sorted_of_row, n_lines, n_blocks, ws.block_count.data,
                       ws.run_start.data);
    check(hipGetLastError(), "run scan kernels");
    ws.host_counts.resize((size_t)count);
    check(hipMemcpyAsync(ws.host_counts.data(), ws.run_count.data, count*sizeof(int),
                         hipMemcpyDeviceToHost, stream), "run count copy");
}

// Second half: waits for the run counts, then sums, links, chain and tables on `stream`;
// leaves cell_sum / point_sum for pedestal_apply_kernel.  parallel_chain: the relaxation
// (run_relax_kernel) with the serial chain behind it for the levels it leaves; else the serial
// chain alone.
inline void pedestal_finish(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                            const LineWing * wing, const LineCore * core, const GridSpec & g,
                            int count, int n_cells, bool parallel_chain = true, int relax_launches = 0)
{
    auto check = pedestal_check;
    const long long n_lines = t.n_lines;
    const int slot_stride = 2*g.cut_off + 3;
    const int n_bins = n_cells + 2*g.cut_off + 3;
    ws.bin_sum.reserve((size_t)count*n_bins);
    ws.bin_end.reserve((size_t)count*n_bins);
    ws.cell_sum.reserve((size_t)count*n_cells);
    ws.point_sum.reserve((size_t)count*n_cells);
    ws.state.reserve((size_t)count*kChainState);
    // (Sizing the pass by a host-side bound on the runs instead -- no wait here -- was built twice,
    // rounds 3 and 4: the user-facing call gains 1 % at most, calls queued in numbers lose the
    // pacing this wait gives them: profiles/r03_ab_prepass.txt, r04_ab_total_order.txt.  Reading the
    // counts back into page-locked instead of pageable memory, so that the copy in
    // pedestal_find_runs does not hold the host either: 0 ... -4 % on the same legs,
    // profiles/r04_ab_pinned_counts.txt.)
    check(hipStreamSynchronize(stream), "run count sync");
    int max_runs = 1;
    for (int c : ws.host_counts) max_runs = std::max(max_runs, c);
    // How many relaxation launches to queue (0: by the table): three settle every table whose runs
    // are about as many as its windows; where dozens of lines alternate between two windows at
    // every integer wavenumber (a 4 M-line table: three runs per window) chains cross more chunk
    // boundaries and five are needed.  Every launch is ~6 us of the host's time, used or not.
    if (relax_launches <= 0) relax_launches = max_runs > 2*n_bins ? 5 : 3;
    relax_launches = std::min(std::max(relax_launches, 2), kMaxRelaxLaunches);
    ws.runs.reserve((size_t)count*max_runs);
    ws.slot_sums.reserve((size_t)count*max_runs*slot_stride);
    hipLaunchKernelGGL(run_sums_kernel, dim3(std::min(max_runs, 65535), count), dim3(64), 0,
                       stream, wing, core, t.sorted_of_row, n_lines, g, n_cells,
                       ws.run_start.data, ws.run_count.data, max_runs, slot_stride,
                       ws.runs.data, ws.slot_sums.data);
    check(hipGetLastError(), "run_sums_kernel");
    ws.prefix_bin.reserve((size_t)count*max_runs);
    hipLaunchKernelGGL(run_prefix_kernel, dim3(count), dim3(kScanThreads), 0, stream,
                       ws.run_count.data, max_runs, n_bins, ws.runs.data, ws.prefix_bin.data,
                       ws.bin_end.data, ws.bin_sum.data, ws.state.data, parallel_chain ? 1 : 0,
                       relax_launches);
    if (parallel_chain)
    {
        ws.links.reserve((size_t)count*max_runs);
        ws.run_slots.reserve((size_t)count*max_runs);
        ws.run_bin.reserve((size_t)count*max_runs);
        ws.pedestals[0].reserve((size_t)count*max_runs);
        ws.pedestals[1].reserve((size_t)count*max_runs);
        hipLaunchKernelGGL(run_links_kernel, dim3(std::min(max_runs, 65535), count), dim3(64), 0,
                           stream, ws.run_count.data, max_runs, slot_stride, n_bins, ws.runs.data,
                           ws.slot_sums.data, ws.prefix_bin.data, ws.bin_end.data, ws.links.data,
                           ws.run_slots.data, ws.run_bin.data, ws.state.data);
        const dim3 chunks((max_runs + 63)/64, count);
        for (int launch = 0; launch < relax_launches; ++launch)
        {
            hipLaunchKernelGGL(run_relax_kernel, chunks, dim3(64), 0, stream, ws.run_count.data,
                               max_runs, n_bins, launch, ws.links.data, ws.run_slots.data,
                               ws.run_bin.data, ws.bin_end.data,
                               ws.pedestals[(launch + 1) & 1].data, ws.pedestals[launch & 1].data,
                               ws.state.data, ws.bin_sum.data);
        }
        check(hipGetLastError(), "run_relax_kernel");
    }
    // The serial chain takes the levels the relaxation left (rows far out of order; or not settled
    // after the last launch: chains of dependences across many chunks of runs).  Behind the
    // relaxation it is launched in its small-LDS form (slots of the spectrum in HBM, the active
    // ones in registers): it usually only looks at the flags and returns, and must not queue for
    // most of a CU's LDS to do that (the accumulate workgroups beside it hold 12-27 KB each).
    const size_t staged_bytes = (size_t)2*kChainChunk*slot_stride*sizeof(double);
    const size_t ring_bytes = (size_t)kChainRing*sizeof(double);
    const size_t lds_bytes = staged_bytes + (size_t)(n_cells + 1 + n_bins)*sizeof(double);
    if (lds_bytes <= 160*1024 - 512 && !parallel_chain)
    {
        if (lds_bytes > 64*1024)
        {
            check(hipFuncSetAttribute(reinterpret_cast<const void *>(run_chain_kernel<true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes), "LDS opt-in");
            check(hipFuncSetAttribute(reinterpret_cast<const void *>(run_chain_kernel<true, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes), "LDS opt-in");
        }
        // Windows of at most 64 slots (cut_off <= 30) keep the active slots in registers.
        auto chain = slot_stride <= 64 ? run_chain_kernel<true, true> : run_chain_kernel<true, false>;
        hipLaunchKernelGGL(chain, dim3(count), dim3(64), lds_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, n_bins,
                           ws.runs.data, ws.slot_sums.data, ws.state.data, (double *)nullptr,
                           ws.bin_sum.data);
    }
    else
    {
        ws.slots.reserve((size_t)count*(n_cells + 1));
        auto chain = slot_stride <= 64 ? run_chain_kernel<false, true> : run_chain_kernel<false, false>;
        hipLaunchKernelGGL(chain, dim3(count), dim3(64), staged_bytes + ring_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, n_bins,
                           ws.runs.data, ws.slot_sums.data, ws.state.data, ws.slots.data,
                           ws.bin_sum.data);
    }
    check(hipGetLastError(), "run_chain_kernel");
    hipLaunchKernelGGL(pedestal_tables_kernel, dim3((n_cells + 255)/256, count), dim3(256), 0,
                       stream, g, n_cells, n_bins, ws.bin_sum.data, ws.cell_sum.data,
                       ws.point_sum.data);
    check(hipGetLastError(), "pedestal_tables_kernel");
}

}  // namespace lbl
