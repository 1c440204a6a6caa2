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
// slots, in parallel over runs), run_chain (one wavefront per level: the only serial part,
// ~20 flops + one slot-vector update per run, everything it touches staged in LDS),
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
    unsigned long long mask_s;  // bit j: run r-1-j holds this run's first slot in its window
    unsigned long long mask_e;  // bit j: run r-1-j holds this run's last slot
    int bin, pad0;
    long long pad1;
};

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
    RawBuffer<int> prefix_last;     // [levels][max_runs]
    RawBuffer<int> regular;         // [levels] 1: monotone windows, fast chain
    RawBuffer<double> slots;        // [levels][cells+1]       (only when LDS is too small)
    RawBuffer<double> bin_sum;      // [levels][cells+2*cut+3]
    RawBuffer<double> cell_sum;     // [levels][cells]
    RawBuffer<double> point_sum;    // [levels][cells]
    std::vector<int> host_counts;
};

inline long long pedestal_bytes_per_level(long long n_lines, int n_cells, int cut_off)
{
    const long long stride = 2*cut_off + 3;
    const long long runs = std::min<long long>(n_lines, 4ll*(n_cells + 2*cut_off + 2));
    return n_lines*4 + runs*((long long)sizeof(RunMeta) + stride*8) + 4ll*(n_cells + stride)*8;
}

// A row opens a run when its window is not empty and differs from the previous row's
// (an empty window in between also ends a run).
__device__ __forceinline__ int opens_run(const LineWing * __restrict__ wing,
                                         const int * __restrict__ sorted_of_row,
                                         long long r, long long n_lines)
{
    if (r >= n_lines) return 0;
    const LineWing w = wing[sorted_of_row[r]];
    if (w.first > w.last) return 0;
    if (r == 0) return 1;
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
        double vs = 0., ve = 0.;
        for (int q0 = 0; q0 < n_slots; q0 += 64)
        {
            const int q = q0 + lane;
            const bool active = q < n_slots;
            const int slot = (extra && q == n_slots - 1) ? n_cells : first_slot + q;
            const int point = slot_point(active ? slot : first_slot, n_cells, g.n_per_v, g.n);
            const double step = (double)point*g.dv;        // absorption.c:39
            const double v = (double)g.v0 + step;
            const bool holds_first = (q0 == 0);
            const bool holds_last = (q0 + 64 >= n_slots);
            double total = 0.;
            for (int r = row_begin; r < row_end; ++r)
            {
                const int j = sorted_of_row[r];
                const LineWing l = w[j];
                if (l.first != head.first || l.last != head.last)
                {
                    break;      // an empty or different window ends the run
                }
                const LineCore k = c[j];
                const double d = v - l.centre;
                double value;
                if (point < k.core_first || point > k.core_last)
                {
                    value = l.bl*rcp_newton(__builtin_fma(d, d, l.g2));
                }
                else
                {
                    value = k.amp*wells_profile(d*k.repwid, k.y);
                }
                total += value;
                if (holds_first)
                {
                    const double at_first = __shfl(value, 0, 64);
                    vs += at_first;
                }
                if (holds_last)
                {
                    ve += __shfl(value, n_slots - 1 - q0, 64);
                }
            }
            if (active) sums[q] = total;
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
// Fast chain: the recurrence in the pedestals alone.
//
// The value accumulated on a slot c before run r is
//     sum_{q<r, c in W_q} G_q[c]  -  sum_{q<r, c in W_q} P_q
// (G: profile sums of run q on its slots, W_q its window, P_q the sum of its pedestals), so
//     P_r = min( GS_r + VS_r - sum_{q<r} P_q [fs_r in W_q] ,  GE_r + VE_r - sum_{q<r} P_q [ls_r in W_q] )
// with fs_r / ls_r the run's end slots and VS / VE its own end values.  Which earlier runs
// hold a slot, and the G sums, do not depend on the pedestals: run_links_kernel finds them in
// parallel (bit masks over the previous 64 runs).  What is left is a recurrence in the P's.
// Take a block of consecutive runs in which every earlier in-block run holds the later runs'
// first slots, and holds either all or none of their last slots.  With L_r the in-block
// prefix sum of P,
//     L_r = min(Ks_r, L_{r-1} + Ke_r)     (no in-block run holds the last slot), or
//     L_r = min(Ks_r, Ke_r)               (all of them do),
// Ks/Ke being the candidates minus the pedestals of covering runs BEFORE the block: a first-
// order recurrence in the (min,+) semiring, i.e. one wavefront scan per block of up to 32
// runs instead of one serial step per run.  Blocks end where the pattern breaks (a window
// that steps backwards: a few dozen places in a 5 000-window spectrum).  Levels where some
// slot is shared by runs more than 64 apart (rows far out of order) keep run_chain_kernel.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double read_lane(double value, int lane)
{
    const long long bits = __double_as_longlong(value);
    const int lo = __builtin_amdgcn_readlane((int)bits, lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int kLinkReach = 64;      // history visible to a run: the previous 64 runs

// prefix_last[r] = max over q <= r of the runs' last slots (one workgroup per level).
__global__ __launch_bounds__(kScanThreads) void run_prefix_kernel(const int * __restrict__ run_count,
                                                          int max_runs,
                                                          const RunMeta * __restrict__ runs,
                                                          int * __restrict__ prefix_last)
{
    __shared__ int wave_max[16];
    __shared__ int carry;
    const int level = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int count = run_count[level];
    const RunMeta * meta = runs + (long long)level*max_runs;
    int * out = prefix_last + (long long)level*max_runs;
    if (threadIdx.x == 0) carry = -1;
    __syncthreads();
    for (int base = 0; base < count; base += kScanThreads)
    {
        const int r = base + threadIdx.x;
        int value = r < count ? meta[r].last_slot : -1;
        for (int offset = 1; offset < 64; offset <<= 1)
        {
            const int up = __shfl_up(value, offset, 64);
            if (lane >= offset) value = max(value, up);
        }
        if (lane == 63) wave_max[wave] = value;
        __syncthreads();
        int before = carry;
        for (int i = 0; i < wave; ++i) before = max(before, wave_max[i]);
        value = max(value, before);
        if (r < count) out[r] = value;
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) carry = value;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void run_links_kernel(const int * __restrict__ run_count,
                                                        int max_runs, int slot_stride,
                                                        const RunMeta * __restrict__ runs,
                                                        const double * __restrict__ slot_sums,
                                                        const int * __restrict__ prefix_last,
                                                        RunLink * __restrict__ links,
                                                        int * __restrict__ regular)
{
    const int level = blockIdx.y;
    const int r = blockIdx.x*blockDim.x + threadIdx.x;
    const int count = run_count[level];
    if (r >= count) return;
    const RunMeta * meta = runs + (long long)level*max_runs;
    const double * sums = slot_sums + (long long)level*max_runs*slot_stride;
    const RunMeta m = meta[r];
    // Nothing older than the visible history may hold one of this run's end slots.
    if (r > kLinkReach &&
        prefix_last[(long long)level*max_runs + r - kLinkReach - 1] >= m.first_slot)
    {
        atomicAnd(&regular[level], 0);
    }
    double gs = 0., ge = 0.;
    unsigned long long mask_s = 0, mask_e = 0;
    for (int j = 0; j < kLinkReach && r - 1 - j >= 0; ++j)
    {
        const int q = r - 1 - j;
        const RunMeta e = meta[q];
        if (e.first_slot <= m.first_slot && m.first_slot <= e.last_slot)
        {
            mask_s |= 1ull << j;
            gs += sums[(long long)q*slot_stride + (m.first_slot - e.first_slot)];
        }
        if (e.first_slot <= m.last_slot && m.last_slot <= e.last_slot)
        {
            mask_e |= 1ull << j;
            ge += sums[(long long)q*slot_stride + (m.last_slot - e.first_slot)];
        }
    }
    RunLink link;
    link.ks = gs + m.vs;
    link.ke = ge + m.ve;
    link.mask_s = mask_s;
    link.mask_e = mask_e;
    link.bin = m.bin;
    link.pad0 = 0;
    link.pad1 = 0;
    links[(long long)level*max_runs + r] = link;
}

__device__ __forceinline__ double shuffle_up(double value, int offset)
{
    return __shfl_up(value, offset, 64);
}

// Inclusive scan over the 64 lanes of the composition of L -> min(L + a, c): the element (a, c)
// of a lane becomes (sum of the a's up to it, the smallest c_i + (a's after i)).
__device__ __forceinline__ void wave_min_plus_scan(double & a, double & c)
{
    const double inf = __builtin_inf();
#define LBL_MIN_PLUS_STEP(CONTROL, ROWS)                                   \
    {                                                                       \
        const double a_left = dpp_from<CONTROL, ROWS>(0., a);               \
        const double c_left = dpp_from<CONTROL, ROWS>(inf, c);              \
        c = fmin(c_left + a, c);                                            \
        a = a_left + a;                                                     \
    }
    LBL_MIN_PLUS_STEP(kRowShr1, 0xf)
    LBL_MIN_PLUS_STEP(kRowShr2, 0xf)
    LBL_MIN_PLUS_STEP(kRowShr4, 0xf)
    LBL_MIN_PLUS_STEP(kRowShr8, 0xf)
    LBL_MIN_PLUS_STEP(kRowBcast15, 0xa)
    LBL_MIN_PLUS_STEP(kRowBcast31, 0xc)
#undef LBL_MIN_PLUS_STEP
}

// Runs per (min,+) scan.  A block ends where a run's first slot is no longer held by every run
// before it in the block (after 2*cut_off + 1 = 51 runs of one-run-per-cell tables) or where its
// last slot is held by some but not all of them -- which is what lines within a pressure shift of
// an integer wavenumber do: they alternate between two windows, and each alternation ends a block.
// The benchmark tables (shifts up to 0.01 cm-1 at 1 atm) take ~1 700 steps for their 4 999 cells, so
// the wider scan buys little there (0.52 -> 0.48 ms); tables without shifts take a tenth of that.
constexpr int kScanBlock = 64;
// (The chain kernels are single workgroups that start beside a resident accumulate grid, whose
// workgroups hold 12-27 KB of LDS each, six or seven to a CU: what a chain kernel asks for must
// fit in what they leave, ~50 KB, or it waits for a CU to drain -- 1.4 ms in a kernel trace when
// the fallback below asked for 108 KB only to find nothing to do.)
constexpr int kLinkChunk = 128;     // run links staged in LDS at a time

// One wavefront per level; only acts on levels run_links_kernel left flagged regular.
__global__ __launch_bounds__(64) void run_chain_scan_kernel(const int * __restrict__ run_count,
                                                            int max_runs, int n_bins,
                                                            const RunLink * __restrict__ links,
                                                            const int * __restrict__ regular,
                                                            double * __restrict__ bin_sum)
{
    extern __shared__ double lds[];
    const int level = blockIdx.x;
    const int lane = threadIdx.x;
    if (!regular[level]) return;
    // The chain is the critical path of the pedestal pass and shares its SIMD with
    // accumulate wavefronts: let the arbiter prefer it.
    __builtin_amdgcn_s_setprio(3);
    double * bins = lds;                                    // [n_bins]
    double * history = bins + n_bins;                       // pedestals of the last 128 runs
    RunLink * staged = reinterpret_cast<RunLink *>(history + 128);   // [kLinkChunk]
    for (int s = lane; s < n_bins; s += 64) bins[s] = 0.;
    for (int s = lane; s < 128; s += 64) history[s] = 0.;
    const int count = run_count[level];
    const RunLink * link = links + (long long)level*max_runs;
    const double inf = __builtin_inf();
    int staged_from = 0, staged_to = 0;
    __syncthreads();
    int b = 0;
    while (b < count)
    {
        if (b + kScanBlock > staged_to && staged_to < count)
        {
            // Stage the next chunk of links (coalesced) so that the serial part reads LDS.
            staged_from = b;
            staged_to = min(count, b + kLinkChunk);
            for (int i = lane; i < staged_to - staged_from; i += 64)
            {
                staged[i] = link[staged_from + i];
            }
            __builtin_amdgcn_wave_barrier();   // one wavefront: LDS keeps its order
        }
        const int r = b + lane;
        const bool candidate = r < staged_to;
        RunLink mine;
        mine.ks = mine.ke = 0.;
        mine.mask_s = mine.mask_e = 0;
        mine.bin = -1;
        if (candidate) mine = staged[r - staged_from];
        // In-block part of the masks: bit j < lane is run r-1-j >= b.
        const unsigned long long in_block = lane == 0 ? 0ull : ((1ull << lane) - 1ull);
        const bool first_held = (mine.mask_s & in_block) == in_block;
        const bool last_free = (mine.mask_e & in_block) == 0ull;
        const bool last_held = (mine.mask_e & in_block) == in_block;
        const bool fits = candidate && first_held && (last_free || last_held);
        const unsigned long long fit_mask = __ballot(fits);
        const int size = __builtin_ctzll(~fit_mask);        // leading run of ones (>= 1)
        const bool active = lane < size;

        // Pedestals of covering runs before the block.  History entry t is run b-1-t, which
        // is bit t + lane of this lane's masks; lane t keeps entry t in a register.
        const unsigned long long old_s = active ? mine.mask_s >> lane : 0ull;
        const unsigned long long old_e = active ? mine.mask_e >> lane : 0ull;
        const double entry = b - 1 - lane >= 0 ? history[(b - 1 - lane) & 127] : 0.;
        double prev_s = 0., prev_e = 0.;
        // Usual case: the covering runs are the most recent J ones (masks 0..01..1), so the
        // sums are prefix sums of the history.
        const bool contiguous = ((old_s & (old_s + 1ull)) | (old_e & (old_e + 1ull))) == 0ull;
        if (__ballot(!contiguous) == 0ull)
        {
            const double prefix = wave_prefix_sum(entry);
            const int count_s = __builtin_popcountll(old_s), count_e = __builtin_popcountll(old_e);
            const double at_s = __shfl(prefix, (count_s - 1) & 63, 64);
            const double at_e = __shfl(prefix, (count_e - 1) & 63, 64);
            prev_s = count_s ? at_s : 0.;
            prev_e = count_e ? at_e : 0.;
        }
        else
        {
            // Arbitrary masks: walk the history up to the farthest entry any lane needs.
            const unsigned long long both = old_s | old_e;
            int reach = both ? 64 - __builtin_clzll(both) : 0;
            for (int offset = 32; offset > 0; offset >>= 1)
            {
                reach = max(reach, __shfl_xor(reach, offset, 64));
            }
            for (int t = 0; t < reach; ++t)
            {
                const double p = read_lane(entry, t);
                if ((old_s >> t) & 1ull) prev_s += p;
                if ((old_e >> t) & 1ull) prev_e += p;
            }
        }

        // Element of the (min,+) scan: L_r = min(L_{r-1} + a, c).
        double a = 0., c = inf;      // identity (right of the block)
        if (active)
        {
            const double k_s = mine.ks - prev_s;
            const double k_e = mine.ke - prev_e;
            if (last_held && lane > 0)
            {
                a = inf;
                c = fmin(k_s, k_e);
            }
            else
            {
                a = k_e;
                c = k_s;
            }
        }
        const double a_own = a, c_own = c;
        static_assert(kScanBlock == 64, "wave_min_plus_scan covers the wavefront");
        wave_min_plus_scan(a, c);
        const double total = fmin(a, c);                    // L_r with L_{b-1} = 0
        const double before = dpp_from<kWaveShr1, 0xf>(0., total);     // lane 0 keeps 0
        // P_r = L_r - L_{r-1}; on the branch L_r = L_{r-1} + a it is a itself, exactly.
        const double pedestal = (before + a_own <= c_own) ? a_own : c_own - before;
        if (active)
        {
            history[r & 127] = pedestal;
            if (mine.bin >= 0 && mine.bin < n_bins) atomicAdd(&bins[mine.bin], pedestal);
        }
        __builtin_amdgcn_wave_barrier();
        b += size;
    }
    __syncthreads();
    for (int s = lane; s < n_bins; s += 64) bin_sum[(long long)level*n_bins + s] = bins[s];
}

constexpr int kChainChunk = 32;     // runs whose slot sums are staged in LDS at a time

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
                                                       const int * __restrict__ regular,
                                                       double * __restrict__ global_slots,
                                                       double * __restrict__ bin_sum)
{
    extern __shared__ double lds[];
    const int level = blockIdx.x;
    const int lane = threadIdx.x;
    if (regular != nullptr && regular[level]) return;    // run_chain_scan_kernel took it
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
                    if (zone + lane <= n_cells) a[zone + lane] = window;
                    if (USE_LDS) __builtin_amdgcn_wave_barrier(); else __syncthreads();
                    zone = first_slot;
                    window = zone + lane <= n_cells ? a[zone + lane] : 0.;
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
    if (WINDOW && zone + lane <= n_cells) a[zone + lane] = window;
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
__global__ void fill_int_kernel(int * data, int n, int value)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) data[i] = value;
}

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
// leaves cell_sum / point_sum for pedestal_apply_kernel.
inline void pedestal_finish(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                            const LineWing * wing, const LineCore * core, const GridSpec & g,
                            int count, int n_cells, bool scan_chain = true)
{
    auto check = pedestal_check;
    const long long n_lines = t.n_lines;
    const int slot_stride = 2*g.cut_off + 3;
    const int n_bins = n_cells + 2*g.cut_off + 3;
    ws.bin_sum.reserve((size_t)count*n_bins);
    ws.cell_sum.reserve((size_t)count*n_cells);
    ws.point_sum.reserve((size_t)count*n_cells);
    check(hipStreamSynchronize(stream), "run count sync");
    int max_runs = 1;
    for (int c : ws.host_counts) max_runs = std::max(max_runs, c);
    ws.runs.reserve((size_t)count*max_runs);
    ws.slot_sums.reserve((size_t)count*max_runs*slot_stride);
    hipLaunchKernelGGL(run_sums_kernel, dim3(std::min(max_runs, 65535), count), dim3(64), 0,
                       stream, wing, core, t.sorted_of_row, n_lines, g, n_cells,
                       ws.run_start.data, ws.run_count.data, max_runs, slot_stride,
                       ws.runs.data, ws.slot_sums.data);
    check(hipGetLastError(), "run_sums_kernel");
    // Fast chain where the windows are monotone (flag per level), serial chain otherwise.
    ws.links.reserve((size_t)count*max_runs);
    ws.regular.reserve((size_t)count);
    const bool try_scan = scan_chain && n_bins*sizeof(double) + 128*sizeof(double) +
                          kLinkChunk*sizeof(RunLink) <= 150*1024;
    check(hipMemsetAsync(ws.regular.data, 0, count*sizeof(int), stream), "regular flags");
    if (try_scan)
    {
        hipLaunchKernelGGL(fill_int_kernel, dim3((count + 255)/256), dim3(256), 0, stream,
                           ws.regular.data, count, 1);
        ws.prefix_last.reserve((size_t)count*max_runs);
        hipLaunchKernelGGL(run_prefix_kernel, dim3(count), dim3(kScanThreads), 0, stream,
                           ws.run_count.data, max_runs, ws.runs.data, ws.prefix_last.data);
        hipLaunchKernelGGL(run_links_kernel, dim3((max_runs + 255)/256, count), dim3(256), 0,
                           stream, ws.run_count.data, max_runs, slot_stride, ws.runs.data,
                           ws.slot_sums.data, ws.prefix_last.data, ws.links.data,
                           ws.regular.data);
        const size_t scan_lds = (size_t)(n_bins + 128)*sizeof(double) + kLinkChunk*sizeof(RunLink);
        if (scan_lds > 64*1024)
        {
            check(hipFuncSetAttribute(reinterpret_cast<const void *>(run_chain_scan_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scan_lds), "LDS opt-in");
        }
        hipLaunchKernelGGL(run_chain_scan_kernel, dim3(count), dim3(64), scan_lds, stream,
                           ws.run_count.data, max_runs, n_bins, ws.links.data, ws.regular.data,
                           ws.bin_sum.data);
        check(hipGetLastError(), "run_chain_scan_kernel");
    }
    // The serial chain takes the levels the scan left (flag `regular` cleared; with an ascending
    // table: none, or the few whose windows are crowded with more runs than the scan's masks
    // reach).  Behind a scan it is launched in its small-LDS form (slots of the spectrum in HBM,
    // the active ones in registers): it usually only looks at the flags and returns, and must
    // not queue for most of a CU's LDS to do that.
    const size_t staged_bytes = (size_t)2*kChainChunk*slot_stride*sizeof(double);
    const size_t lds_bytes = staged_bytes + (size_t)(n_cells + 1 + n_bins)*sizeof(double);
    if (lds_bytes <= 160*1024 - 512 && !try_scan)
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
                           ws.runs.data, ws.slot_sums.data, ws.regular.data, (double *)nullptr,
                           ws.bin_sum.data);
    }
    else
    {
        ws.slots.reserve((size_t)count*(n_cells + 1));
        auto chain = slot_stride <= 64 ? run_chain_kernel<false, true> : run_chain_kernel<false, false>;
        hipLaunchKernelGGL(chain, dim3(count), dim3(64), staged_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, n_bins,
                           ws.runs.data, ws.slot_sums.data, ws.regular.data, ws.slots.data,
                           ws.bin_sum.data);
    }
    check(hipGetLastError(), "run_chain_kernel");
    hipLaunchKernelGGL(pedestal_tables_kernel, dim3((n_cells + 255)/256, count), dim3(256), 0,
                       stream, g, n_cells, n_bins, ws.bin_sum.data, ws.cell_sum.data,
                       ws.point_sum.data);
    check(hipGetLastError(), "pedestal_tables_kernel");
}

}  // namespace lbl
