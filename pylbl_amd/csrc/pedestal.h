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

}  // namespace lbl

#include "pedestal_runs.h"
#include "pedestal_chain.h"

namespace lbl {

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
