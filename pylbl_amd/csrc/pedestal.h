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
//    Hence for a run with sums VS = sum V_i(first), D = sum (V_i(first) - V_i(last)),
//    entered with end-slot values (a_s, a_e):
//        delta_n = (a_s - a_e) + D,   a_s' = max(delta_n, 0),   a_e' = max(-delta_n, 0),
//        sum of the run's pedestals  P = a_s + VS - a_s',
//    and every interior slot c receives  sum_i V_i(c) - P.
//  * The spectrum is then  sum_j V_j(x) - sum_j p_j [x in window_j]; the second sum is
//    piecewise constant between integer wavenumbers, so the accumulate kernel subtracts one
//    table value per grid point in its epilogue.
//
// Kernels: run_scan (find runs in reference row order), run_sums (one wavefront per run:
// profile values on the run's slots, in parallel over runs), run_chain (one wavefront per
// level: the only serial part, ~20 flops + one slot-vector update per run, slots in LDS),
// pedestal_tables (per 1 cm-1 cell: total pedestal covering its interior / its integer point).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "line_prep.h"
#include "tile_schedule.h"
#include "voigt_profile.h"

namespace lbl {

struct RunMeta
{
    int row_begin;      // first row (reference order) of the run
    int first, last;    // the run's window (grid indices, inclusive)
    int n_slots;        // integer slots first/npv .. last/npv (+1 if last is not an integer point)
    double vs;          // sum over the run of V_i(first)
    double d;           // sum over the run of V_i(first) - V_i(last)
    double pedestal;    // filled by run_chain: sum of the run's pedestals
    double pad;
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
    RawBuffer<int> run_start;       // [levels][n_lines]
    RawBuffer<int> run_count;       // [levels]
    RawBuffer<RunMeta> runs;        // [levels][max_runs]
    RawBuffer<double> slot_sums;    // [levels][max_runs][slot_stride]
    RawBuffer<double> slots;        // [levels][cells+1]  (only when LDS is too small)
    RawBuffer<double> cell_sum;     // [levels][cells]
    RawBuffer<double> point_sum;    // [levels][cells]
    std::vector<int> host_counts;
};

inline long long pedestal_bytes_per_level(long long n_lines, int n_cells, int cut_off)
{
    const long long stride = 2*cut_off + 3;
    const long long runs = std::min<long long>(n_lines, 4ll*(n_cells + 2*cut_off + 2));
    return n_lines*4 + runs*((long long)sizeof(RunMeta) + stride*8) + 3ll*(n_cells + 1)*8;
}

__device__ __forceinline__ bool window_of_row(const LineWing * __restrict__ wing,
                                              const int * __restrict__ sorted_of_row,
                                              long long r, int & first, int & last)
{
    const LineWing w = wing[sorted_of_row[r]];
    first = w.first;
    last = w.last;
    return w.first <= w.last;
}

// One 1024-thread block per level: marks rows that open a run and compacts their indices.
__global__ __launch_bounds__(1024) void run_scan_kernel(const LineWing * __restrict__ wing,
                                                        const int * __restrict__ sorted_of_row,
                                                        long long n_lines,
                                                        int * __restrict__ run_start,
                                                        int * __restrict__ run_count)
{
    __shared__ int wave_total[16];
    __shared__ int carry;
    const int level = blockIdx.x;
    const LineWing * w = wing + (long long)level*n_lines;
    int * out = run_start + (long long)level*n_lines;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (long long base = 0; base < n_lines; base += 1024)
    {
        const long long r = base + threadIdx.x;
        int flag = 0;
        if (r < n_lines)
        {
            int f, l;
            if (window_of_row(w, sorted_of_row, r, f, l))
            {
                flag = 1;
                if (r > 0)
                {
                    int pf, pl;
                    if (window_of_row(w, sorted_of_row, r - 1, pf, pl) && pf == f && pl == l)
                    {
                        flag = 0;
                    }
                }
            }
        }
        // Inclusive scan of the flags inside the wavefront, then across the 16 wavefronts.
        int scan = flag;
        for (int offset = 1; offset < 64; offset <<= 1)
        {
            const int up = __shfl_up(scan, offset, 64);
            if (lane >= offset) scan += up;
        }
        if (lane == 63) wave_total[wave] = scan;
        __syncthreads();
        int before = carry;
        for (int i = 0; i < wave; ++i) before += wave_total[i];
        if (flag) out[before + scan - 1] = (int)r;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + scan;
        __syncthreads();
    }
    if (threadIdx.x == 0) run_count[level] = carry;
}

__device__ __forceinline__ int slot_point(int slot, int n_cells, int n_per_v, int n)
{
    return slot < n_cells ? slot*n_per_v : n - 1;
}

// One wavefront per run (grid-stride over runs): evaluates every row of the run on the
// run's slots with the same profile code the accumulate kernel uses.
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
        double vs = 0., dd = 0.;
        for (int q0 = 0; q0 < n_slots; q0 += 64)
        {
            const int q = q0 + lane;
            const bool active = q < n_slots;
            const int slot = (extra && q == n_slots - 1) ? n_cells : first_slot + q;
            const int point = slot_point(active ? slot : first_slot, n_cells, g.n_per_v, g.n);
            const double step = (double)point*g.dv;        // absorption.c:39
            const double v = (double)g.v0 + step;
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
                if (q0 == 0)
                {
                    const double at_first = __shfl(value, 0, 64);
                    const int last_lane = (n_slots - 1) & 63;
                    // V_i(last) lives in the final 64-slot group; fetched there.
                    if (n_slots <= 64)
                    {
                        const double at_last = __shfl(value, last_lane, 64);
                        vs += at_first;
                        dd += at_first - at_last;
                    }
                    else
                    {
                        vs += at_first;
                        dd += at_first;
                    }
                }
                else if (q0 + 64 >= n_slots)
                {
                    const double at_last = __shfl(value, (n_slots - 1 - q0), 64);
                    dd -= at_last;
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
            meta.d = dd;
            meta.pedestal = 0.;
            meta.pad = 0.;
            runs[(long long)level*max_runs + run] = meta;
        }
    }
}

// One wavefront per level: the serial recurrence over runs.  `a` holds the accumulated
// spectrum on the slots (LDS when it fits, HBM otherwise).
template <bool USE_LDS>
__global__ __launch_bounds__(64) void run_chain_kernel(const int * __restrict__ run_count,
                                                       int max_runs, int slot_stride,
                                                       GridSpec g, int n_cells,
                                                       RunMeta * __restrict__ runs,
                                                       const double * __restrict__ slot_sums,
                                                       double * __restrict__ global_slots)
{
    extern __shared__ double lds_slots[];
    const int level = blockIdx.x;
    const int lane = threadIdx.x;
    double * a = USE_LDS ? lds_slots : global_slots + (long long)level*(n_cells + 1);
    for (int s = lane; s <= n_cells; s += 64) a[s] = 0.;
    __syncthreads();
    const int count = run_count[level];
    RunMeta * meta = runs + (long long)level*max_runs;
    const double * sums = slot_sums + (long long)level*max_runs*slot_stride;
    for (int run = 0; run < count; ++run)
    {
        const RunMeta m = meta[run];
        const int first_slot = m.first/g.n_per_v;
        const int last_int = m.last/g.n_per_v;
        const bool extra = (last_int*g.n_per_v != m.last);
        const int last_slot = extra ? n_cells : last_int;
        const double a_s = a[first_slot];
        const double a_e = a[last_slot];
        const double delta_n = (a_s - a_e) + m.d;
        const double s_new = delta_n > 0. ? delta_n : 0.;
        const double e_new = delta_n < 0. ? -delta_n : 0.;
        const double pedestal = (a_s + m.vs) - s_new;
        __syncthreads();   // single wavefront: orders the LDS reads above before the writes
        for (int q = lane; q < m.n_slots; q += 64)
        {
            const int slot = (extra && q == m.n_slots - 1) ? n_cells : first_slot + q;
            double value;
            if (q == 0) value = s_new;
            else if (q == m.n_slots - 1) value = e_new;
            else value = a[slot] + (sums[(long long)run*slot_stride + q] - pedestal);
            if (m.n_slots == 1) value = 0.;
            a[slot] = value;
        }
        __syncthreads();
        if (lane == 0) meta[run].pedestal = pedestal;
    }
}

// One thread per 1 cm-1 cell: total pedestal of the runs whose window holds the cell's
// interior points / its integer point.  Runs are read in order (wave-uniform loads), so the
// sums are reproducible.
__global__ __launch_bounds__(256) void pedestal_tables_kernel(const int * __restrict__ run_count,
                                                              int max_runs, GridSpec g,
                                                              int n_cells,
                                                              const RunMeta * __restrict__ runs,
                                                              double * __restrict__ cell_sum,
                                                              double * __restrict__ point_sum)
{
    const int level = blockIdx.y;
    const int cell = blockIdx.x*blockDim.x + threadIdx.x;
    const int count = run_count[level];
    const RunMeta * meta = runs + (long long)level*max_runs;
    const long long lo = (long long)cell*g.n_per_v;           // the integer point
    const long long in_lo = lo + 1;                            // interior of the cell
    const long long in_hi = lo + g.n_per_v - 1;
    double interior = 0., point = 0.;
    for (int run = 0; run < count; ++run)
    {
        const int first = meta[run].first, last = meta[run].last;
        const double p = meta[run].pedestal;
        if (first <= lo && lo <= last) point += p;
        if (first <= in_lo && in_hi <= last) interior += p;
    }
    if (cell < n_cells)
    {
        cell_sum[(long long)level*n_cells + cell] = interior;
        point_sum[(long long)level*n_cells + cell] = point;
    }
}

// Runs the whole pedestal pre-pass for `count` levels whose LineWing/LineCore arrays are
// already in HBM; leaves cell_sum / point_sum for the accumulate kernel's epilogue.
inline void pedestal_pass(PedestalWorkspace & ws, hipStream_t stream, const LineTableView & t,
                          const LineWing * wing, const LineCore * core, const GridSpec & g,
                          int count, int n_cells)
{
    auto check = [](hipError_t status, const char * what) {
        if (status != hipSuccess)
        {
            throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(status));
        }
    };
    const long long n_lines = t.n_lines;
    const int slot_stride = 2*g.cut_off + 3;
    ws.run_start.reserve((size_t)(count*n_lines));
    ws.run_count.reserve((size_t)count);
    ws.cell_sum.reserve((size_t)count*n_cells);
    ws.point_sum.reserve((size_t)count*n_cells);
    hipLaunchKernelGGL(run_scan_kernel, dim3(count), dim3(1024), 0, stream, wing,
                       t.sorted_of_row, n_lines, ws.run_start.data, ws.run_count.data);
    check(hipGetLastError(), "run_scan_kernel");
    ws.host_counts.resize((size_t)count);
    check(hipMemcpyAsync(ws.host_counts.data(), ws.run_count.data, count*sizeof(int),
                         hipMemcpyDeviceToHost, stream), "run count copy");
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
    const size_t lds_bytes = (size_t)(n_cells + 1)*sizeof(double);
    if (lds_bytes <= 160*1024 - 256)
    {
        if (lds_bytes > 64*1024)
        {
            check(hipFuncSetAttribute(reinterpret_cast<const void *>(run_chain_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes), "LDS opt-in");
        }
        hipLaunchKernelGGL(run_chain_kernel<true>, dim3(count), dim3(64), lds_bytes, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, ws.runs.data,
                           ws.slot_sums.data, (double *)nullptr);
    }
    else
    {
        ws.slots.reserve((size_t)count*(n_cells + 1));
        hipLaunchKernelGGL(run_chain_kernel<false>, dim3(count), dim3(64), 0, stream,
                           ws.run_count.data, max_runs, slot_stride, g, n_cells, ws.runs.data,
                           ws.slot_sums.data, ws.slots.data);
    }
    check(hipGetLastError(), "run_chain_kernel");
    hipLaunchKernelGGL(pedestal_tables_kernel, dim3((n_cells + 255)/256, count), dim3(256), 0,
                       stream, ws.run_count.data, max_runs, g, n_cells, ws.runs.data,
                       ws.cell_sum.data, ws.point_sum.data);
    check(hipGetLastError(), "pedestal_tables_kernel");
}

}  // namespace lbl
