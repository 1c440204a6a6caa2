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
// Kernels (five launches, round 6; eleven to thirteen before): run_find (the runs in reference row
// order, the prefix maxima of their bins: one single-pass scan), run_sums (one wavefront per run:
// profile values on the run's slots, in parallel over runs), run_links (the stretch of earlier runs
// that can hold a run's end slots, and what they added there), run_solve (the recurrence in the
// runs' pedestal totals as a triangular system: a few sweeps inside one launch, each exact inside
// chunks of 64 runs; the bins' totals; and, for the levels the sweeps leave, the serial form, one
// wavefront per level), pedestal_apply (per point: the total pedestal of the windows that hold it,
// summed from the bins' totals per 1 cm-1 cell, subtracted).
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
    RawBuffer<unsigned long long> scan[2];  // [levels][blocks of kScanThreads rows]: run_find_kernel's descriptors, used in turn
    int scan_turn = 0;
    RawBuffer<int> run_start;       // [levels][n_lines]
    RawBuffer<int> prefix_bin;      // [levels][n_lines]: running maximum of the runs' bins
    RawBuffer<int> run_count;       // [levels]
    RawBuffer<RunMeta> runs;        // [levels][max_runs]
    RawBuffer<double> slot_sums;    // [levels][max_runs][slot_stride]
    RawBuffer<RunLink> links;       // [levels][max_runs]
    RawBuffer<double> pedestals;    // [sweeps][levels][max_runs]: every sweep's values
    RawBuffer<int> progress;        // [levels][chunks of 64 runs]: sweeps completed
    RawBuffer<int2> run_slots;      // [levels][max_runs]: the runs' end slots, packed for the relaxation
    RawBuffer<int> run_bin;         // [levels][max_runs]
    RawBuffer<int> bin_end;         // [levels][bins]: 1 + the last run of every bin
    RawBuffer<int> bin_first;       // [levels][bins]: the first run of every bin
    RawBuffer<int> state;           // [levels][kChainState], see pedestal_chain.h
    RawBuffer<double> slots;        // [levels][cells+1]: the serial chain's slots of the spectrum
    RawBuffer<double> bin_sum;      // [levels][cells+2*cut+3]: total pedestal of every window
    std::vector<int> host_counts;
};

constexpr int kRunCut = 1024;     // see opens_run
constexpr int kPedestalSweepBuffers = 7;    // kMaxRelaxLaunches (pedestal_chain.h)

inline long long pedestal_bytes_per_level(long long n_lines, int n_cells, int cut_off)
{
    const long long stride = 2*cut_off + 3;
    const long long runs = std::min<long long>(n_lines, 4ll*(n_cells + 2*cut_off + 2) + n_lines/kRunCut);
    return n_lines*8 + runs*((long long)(sizeof(RunMeta) + sizeof(RunLink)) + stride*8 + 16 +
                             8ll*kPedestalSweepBuffers) + 4ll*(n_cells + stride)*8;
}

}  // namespace lbl

#include "pedestal_runs.h"
#include "pedestal_chain.h"

namespace lbl {

// k = (sums - pedestal total of the windows holding the point) [* number density] [+ k].
// Windows start and end on integer wavenumbers, so the total is constant inside a 1 cm-1 cell and
// has one extra bin of lines on the integer point that closes a window: the interior points of cell
// c lie in the windows of bins c+1 .. c+2 cut_off+1 (bin b: the window of the lines with
// floor(centre) = b + v0 - cut_off - 1), its integer point also in bin c (the window that closes
// there).  Every workgroup forms the two totals of the (at most 256) cells its 256 points lie in
// from the bins' totals, in LDS -- sums of non-negative totals in a fixed order: reproducible, and
// exactly zero where no line reaches.  (Rounds 1-5 kept the two tables in HBM, written by a
// launch of their own.)  Four points per thread: the kernel moves 16-24 bytes per point and
// nothing else, and one point per thread left it at a third of what HBM gives.
constexpr int kApplyPoints = 4;     // grid points per thread of pedestal_apply_kernel

__global__ __launch_bounds__(256) void pedestal_apply_kernel(const double * __restrict__ sums,
                                                             long long sums_stride,
                                                             double * __restrict__ out,
                                                             long long out_stride,
                                                             const double * __restrict__ bin_sum,
                                                             const LevelScalars * __restrict__ levels,
                                                             int first, int end, int n_per_v,
                                                             int n_bins, int cut_off, int max_span,
                                                             int scale_density, int accumulate)
{
    // Points [first, end): the whole grid, or the columns of one piece of a streamed call; a
    // workgroup takes 256 x kApplyPoints consecutive ones (thread t: t, t + 256, ...).
    extern __shared__ double apply_lds[];
    constexpr int kBlock = 256*kApplyPoints;
    const int level = blockIdx.y;
    const int i_lo = first + blockIdx.x*kBlock;
    if (i_lo >= end) return;
    const int i_hi = min(i_lo + kBlock - 1, end - 1);
    const int cell_lo = i_lo/n_per_v;
    const int span = i_hi/n_per_v - cell_lo + 1;            // <= max_span cells
    const int wanted = span + 2*cut_off + 1;                // bins cell_lo .. cell_hi + 2 cut_off + 1
    double * bins = apply_lds;
    double * cell_sum = apply_lds + max_span + 2*cut_off + 2;
    double * point_sum = cell_sum + max_span;
    const double * source = bin_sum + (long long)level*n_bins + cell_lo;
    for (int t = threadIdx.x; t < wanted; t += 256) bins[t] = source[t];
    __syncthreads();
    for (int c = threadIdx.x; c < span; c += 256)
    {
        double interior = 0.;
        for (int k = 1; k <= 2*cut_off + 1; ++k)
        {
            interior += bins[c + k];
        }
        cell_sum[c] = interior;
        point_sum[c] = interior + bins[c];
    }
    __syncthreads();
    const double density = scale_density ? levels[level].density : 1.;
    const double * from = sums + (long long)level*sums_stride;
    double * k = out + (long long)level*out_stride;
    double value[kApplyPoints], before[kApplyPoints];
#pragma unroll
    for (int j = 0; j < kApplyPoints; ++j)
    {
        const int i = i_lo + (int)threadIdx.x + 256*j;
        value[j] = i < end ? from[i] : 0.;
        before[j] = (accumulate && i < end) ? k[i] : 0.;
    }
#pragma unroll
    for (int j = 0; j < kApplyPoints; ++j)
    {
        const int i = i_lo + (int)threadIdx.x + 256*j;
        if (i >= end) continue;
        const int cell = i/n_per_v;
        const bool on_integer = (cell*n_per_v == i);
        const double pedestal = on_integer ? point_sum[cell - cell_lo] : cell_sum[cell - cell_lo];
        double result = value[j] - pedestal;
        if (scale_density) result *= density;
        if (accumulate) result += before[j];
        k[i] = result;
    }
}

// Cells a workgroup's points can lie in, and the dynamic LDS the kernel asks for.
inline int pedestal_apply_span(int n_per_v)
{
    return 256*kApplyPoints/std::max(n_per_v, 1) + 2;
}

inline size_t pedestal_apply_lds_bytes(int cut_off, int n_per_v)
{
    return (size_t)(3*pedestal_apply_span(n_per_v) + 2*cut_off + 2)*sizeof(double);
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

// First half: finds the runs (one launch) and starts the copy of the run counts to the host.  The
// engine orders it before the accumulate launch of the same call (a resident accumulate grid of
// another call does not hold it up, see kScanThreads).
inline void pedestal_find_runs(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                               const LineWing * wing, const GridSpec & g, int count, int n_cells,
                               bool parallel_chain)
{
    auto check = pedestal_check;
    const long long n_lines = t.n_lines;
    const int n_blocks = (int)((n_lines + kScanThreads - 1)/kScanThreads);
    const int n_bins = n_cells + 2*g.cut_off + 3;
    const size_t descriptors = (size_t)count*n_blocks;
    if (descriptors > ws.scan[0].capacity)
    {
        // Fresh descriptor arrays start out clear; from then on every launch clears the array the
        // next one uses.
        for (auto & array : ws.scan)
        {
            array.reserve(descriptors);
            check(hipMemsetAsync(array.data, 0, array.capacity*sizeof(unsigned long long), stream),
                  "descriptor clear");
        }
    }
    ws.run_start.reserve((size_t)(count*n_lines));
    ws.prefix_bin.reserve((size_t)(count*n_lines));
    ws.run_count.reserve((size_t)count);
    ws.bin_sum.reserve((size_t)count*n_bins);
    ws.bin_end.reserve((size_t)count*n_bins);
    ws.bin_first.reserve((size_t)count*n_bins);
    ws.state.reserve((size_t)count*kChainState);
    unsigned long long * now = ws.scan[ws.scan_turn].data;
    unsigned long long * next = ws.scan[ws.scan_turn ^ 1].data;
    ws.scan_turn ^= 1;
    hipLaunchKernelGGL(run_find_kernel, dim3(n_blocks, count), dim3(kScanThreads), 0, stream, wing,
                       t.sorted_of_row, n_lines, n_blocks, g.v0 - g.cut_off - 1, now, next,
                       (long long)ws.scan[0].capacity, ws.run_start.data, ws.prefix_bin.data,
                       ws.run_count.data, n_bins, ws.bin_end.data, ws.bin_first.data,
                       ws.bin_sum.data, ws.state.data, parallel_chain ? 1 : 0);
    check(hipGetLastError(), "run_find_kernel");
    ws.host_counts.resize((size_t)count);
    check(hipMemcpyAsync(ws.host_counts.data(), ws.run_count.data, count*sizeof(int),
                         hipMemcpyDeviceToHost, stream), "run count copy");
}

// Second half: waits for the run counts, then sums, links and the solve on `stream`; leaves bin_sum
// for pedestal_apply_kernel.  parallel_chain: the relaxation (run_solve_kernel) with the serial
// chain inside it for the levels it leaves; else the serial chain alone.
inline void pedestal_finish(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                            const LineWing * wing, const LineCore * core, const GridSpec & g,
                            int count, int n_cells, bool parallel_chain = true, int relax_launches = 0)
{
    auto check = pedestal_check;
    const long long n_lines = t.n_lines;
    const int slot_stride = 2*g.cut_off + 3;
    const int n_bins = n_cells + 2*g.cut_off + 3;
    // (Sizing the pass by a host-side bound on the runs instead -- no wait here -- was built twice,
    // rounds 3 and 4: the user-facing call gains 1 % at most, calls queued in numbers lose the
    // pacing this wait gives them: profiles/r03_ab_prepass.txt, r04_ab_total_order.txt.  Reading the
    // counts back into page-locked instead of pageable memory, so that the copy in
    // pedestal_find_runs does not hold the host either: 0 ... -4 % on the same legs,
    // profiles/r04_ab_pinned_counts.txt.)
    check(hipStreamSynchronize(stream), "run count sync");
    int max_runs = 1;
    for (int c : ws.host_counts)
    {
        if (c < 0 || c > n_lines)
        {
            throw std::runtime_error("the scan for the pedestal's runs did not complete.");
        }
        max_runs = std::max(max_runs, c);
    }
    // How many sweeps of the relaxation (0: by the table): three settle every table whose runs are
    // about as many as its windows; where dozens of lines alternate between two windows at every
    // integer wavenumber (a 4 M-line table: three runs per window) chains cross more chunk
    // boundaries and five are needed.
    if (relax_launches <= 0) relax_launches = max_runs > 2*n_bins ? 5 : 3;
    relax_launches = std::min(std::max(relax_launches, 2), kMaxRelaxLaunches);
    ws.runs.reserve((size_t)count*max_runs);
    ws.slot_sums.reserve((size_t)count*max_runs*slot_stride);
    hipLaunchKernelGGL(run_sums_kernel, dim3(std::min(max_runs, 65535), count), dim3(64), 0,
                       stream, wing, core, t.sorted_of_row, n_lines, g, n_cells,
                       ws.run_start.data, ws.run_count.data, max_runs, slot_stride,
                       ws.runs.data, ws.slot_sums.data);
    check(hipGetLastError(), "run_sums_kernel");
    // The serial chain in its small-LDS form (slots of the spectrum in HBM, the active ones in
    // registers and a ring in LDS): 15.5 KB, less than one accumulate workgroup holds.
    const size_t staged_bytes = (size_t)2*kChainChunk*slot_stride*sizeof(double);
    const size_t ring_bytes = (size_t)kChainRing*sizeof(double);
    if (parallel_chain)
    {
        const int max_chunks = (max_runs + 63)/64;
        ws.links.reserve((size_t)count*max_runs);
        ws.run_slots.reserve((size_t)count*max_runs);
        ws.run_bin.reserve((size_t)count*max_runs);
        ws.pedestals.reserve((size_t)relax_launches*count*max_runs);
        ws.progress.reserve((size_t)count*max_chunks);
        ws.slots.reserve((size_t)count*(n_cells + 1));
        hipLaunchKernelGGL(run_links_kernel, dim3(std::min(max_runs, 65535), count), dim3(64), 0,
                           stream, ws.run_count.data, max_runs, slot_stride, n_bins, ws.runs.data,
                           ws.slot_sums.data, ws.prefix_bin.data, n_lines, ws.bin_end.data,
                           ws.bin_first.data, ws.progress.data, max_chunks, ws.links.data,
                           ws.run_slots.data, ws.run_bin.data, ws.state.data);
        check(hipGetLastError(), "run_links_kernel");
        // Windows of at most 64 slots (cut_off <= 30) keep the active slots in registers.
        auto solve = slot_stride <= 64 ? run_solve_kernel<true> : run_solve_kernel<false>;
        hipLaunchKernelGGL(solve, dim3(max_chunks, count), dim3(64), staged_bytes + ring_bytes,
                           stream, ws.run_count.data, max_runs, max_chunks, n_bins, relax_launches,
                           ws.links.data, ws.run_slots.data, ws.run_bin.data, ws.bin_end.data,
                           ws.bin_first.data, ws.pedestals.data, (long long)count*max_runs,
                           ws.progress.data, ws.state.data, ws.bin_sum.data, slot_stride, g,
                           n_cells, ws.runs.data, ws.slot_sums.data, ws.slots.data);
        check(hipGetLastError(), "run_solve_kernel");
        return;
    }
    // The serial chain alone (engine option scan_chain = 0): slots and bin totals in LDS where they
    // fit, else in HBM.
    const size_t lds_bytes = staged_bytes + (size_t)(n_cells + 1 + n_bins)*sizeof(double);
    if (lds_bytes <= 160*1024 - 512)
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
        auto chain = slot_stride <= 64 ? run_chain_kernel<true, true> : run_chain_kernel<true, false>;
        hipLaunchKernelGGL(chain, dim3(count), dim3(64), lds_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, n_bins,
                           ws.runs.data, ws.slot_sums.data, (double *)nullptr, ws.bin_sum.data);
    }
    else
    {
        ws.slots.reserve((size_t)count*(n_cells + 1));
        auto chain = slot_stride <= 64 ? run_chain_kernel<false, true> : run_chain_kernel<false, false>;
        hipLaunchKernelGGL(chain, dim3(count), dim3(64), staged_bytes + ring_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, n_bins,
                           ws.runs.data, ws.slot_sums.data, ws.slots.data, ws.bin_sum.data);
    }
    check(hipGetLastError(), "run_chain_kernel");
}

}  // namespace lbl
