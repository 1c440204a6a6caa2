// Launch-side kernels around the accumulate kernel: per-line scalars for a batch of
// levels ("K1", device form of line_prep.h) and the per-tile cut points into the
// wavenumber-sorted line table.
#pragma once

#include <hip/hip_runtime.h>

#include "accumulate.h"
#include "line_prep.h"

namespace lbl {

// Line table of one molecule, sorted by wavenumber (stable), resident in HBM.
struct LineTableView
{
    const double * nu;
    const double * sw;
    const double * gamma_air;
    const double * gamma_self;
    const double * n_air;
    const double * elower;
    const double * delta_air;
    const int * iso_slot;       // local_iso_id - 1 with 0 -> 9 (spectral_database.c:173-178)
    const int * row;            // position in the reference's row order
    const int * sorted_of_row;  // inverse of row: sorted position of reference row r
    long long n_lines;
    // Where the sorted table reaches every 1/cell_scale cm-1: cell_first[i] = the first line with
    // nu >= cell_base + i/cell_scale (i < cell_entries; made once per molecule, lbl_molecule_load).
    // A search for a wavenumber starts in the cell that holds it -- a dozen lines -- instead of the
    // whole table: 4-5 dependent loads instead of 19 for 400 k lines (round 6; the searches are
    // the serial part of the prologue kernel).  nullptr: search the whole table.
    const int * cell_first;
    double cell_base, cell_scale;
    int cell_entries;
};

// The index range [lo, hi] inside which the first line with nu >= x (or > x) lies.
__device__ __forceinline__ void search_bracket(const LineTableView & t, double x, int & lo, int & hi)
{
    lo = 0;
    hi = (int)t.n_lines;
    if (t.cell_first == nullptr || !(x == x)) return;
    // (one cell either side of the one x falls in: the cell edges and this product are rounded)
    const double at = floor((x - t.cell_base)*t.cell_scale);
    const int last = t.cell_entries - 1;
    if (at >= 1.) lo = t.cell_first[min((int)fmin(at, (double)last) - 1, last)];
    if (at + 2. <= (double)last) hi = t.cell_first[max((int)fmax(at, -2.) + 2, 0)];
}

struct RangeRule
{
    int policy;                 // LBL_RANGE_REFERENCE / LBL_RANGE_SKIP
    int row_limit;              // rows [0,row_limit) are reached before the break (absorption.c:80-83)
    double nu_min, nu_max;      // v0-(cut_off+1), vn+cut_off+1
};

__host__ __device__ inline bool line_accepted(const RangeRule & r, double nu, int row)
{
    if (r.policy == 0)
    {
        return row < r.row_limit;
    }
    return !(nu > r.nu_max || nu < r.nu_min);
}

__device__ inline int first_not_below(const double * __restrict__ nu, int lo, int hi, double x)
{
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (nu[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ inline int first_above(const double * __restrict__ nu, int lo, int hi, double x)
{
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (nu[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Eight threads per (tile, level), one binary search each (the searches are chains of
// dependent loads: side by side they take the time of one).  A line's window is fixed by
// b = floor(nu + p*delta): [b-cut, b+cut+1] cm-1, both ends included (spectra.c:48-62).  With
// |p*delta| <= shift_max the tests on b become tests on nu, which is the sort key:
//   may overlap the tile    b in [ceil(i0/npv)+v0-cut-1, floor(i1/npv)+v0+cut]
//   covers the whole tile   b in [ceil(i1/npv)+v0-cut-1, floor(i0/npv)+v0+cut]
//   tile may touch |x|<xlim0    |nu - tile| <= core_reach*nu (+ shift)
__device__ __forceinline__ void schedule_tile(const LineTableView & table,
                                              const LevelScalars & lv, const GridSpec & g,
                                              const Tiling & tiling, const int farfield,
                                              int tile, int which, bool active,
                                              TileSchedule * __restrict__ out)
{
    long long i0 = 0, i1 = 0;
    if (active) tile_bounds(tiling, tile, g.n_per_v, g.n, i0, i1);
    const double smax = lv.shift_max;
    const long long npv = g.n_per_v;
    const double any_lo = (double)((i0 + npv - 1)/npv + g.v0 - g.cut_off - 1);
    const double any_hi = (double)(i1/npv + g.v0 + g.cut_off);
    const double full_lo = (double)((i1 + npv - 1)/npv + g.v0 - g.cut_off - 1);
    const double full_hi = (double)(i0/npv + g.v0 + g.cut_off);
    const double v_lo = (double)g.v0 + (double)i0*g.dv;
    const double v_hi = (double)g.v0 + (double)i1*g.dv;
    const double kk = lv.core_reach;
    const bool bounded = kk < 0.5;
    const double core = bounded ? kk*(v_hi + smax)/(1. - kk)*(1. + 1.e-9) + smax + 1.e-9 : 0.;
    const double half = 0.5*(v_hi - v_lo);
    const double u0 = tile_centre(g.v0, g.dv, i0, i1);
    const double radius = fmax(kFarRatio*half, core + half)*(1. + 1.e-9) + 1.e-6;
    // which: 0 lo, 1 a1, 2 c1, 3 c2, 4 a2, 5 hi, 6 f1, 7 f2 (the order of TileSchedule).
    double key;
    bool above = false;         // first index with nu > key (else: nu >= key)
    switch (which)
    {
    case 0: key = any_lo - smax; break;
    case 1: key = full_lo + smax; break;
    case 2: key = v_lo - core; break;
    case 3: key = v_hi + core; above = true; break;
    case 4: key = full_hi + 1. - smax; break;
    case 5: key = any_hi + 1. + smax; break;
    case 6: key = u0 - radius - smax; above = true; break;
    default: key = u0 + radius + smax; break;
    }
    const int n_lines = (int)table.n_lines;
    int found = 0;
    if (active)
    {
        int lo, hi;
        search_bracket(table, key, lo, hi);
        found = above ? first_above(table.nu, lo, hi, key) : first_not_below(table.nu, lo, hi, key);
    }
    const int lane = threadIdx.x & 63;
    const int leader = lane & ~7;
    int value[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
    {
        value[k] = __shfl(found, leader + k, 64);
    }
    if (which != 0 || !active)
    {
        return;
    }
    TileSchedule s;
    s.lo = value[0]; s.a1 = value[1]; s.c1 = value[2]; s.c2 = value[3];
    s.a2 = value[4]; s.hi = value[5];
    if (!bounded)
    {
        s.c1 = 0;
        s.c2 = n_lines;
    }
    if (s.hi < s.lo) s.hi = s.lo;
    s.a1 = min(max(s.a1, s.lo), s.hi);
    s.a2 = min(max(s.a2, s.a1), s.hi);
    s.c1 = min(max(s.c1, s.a1), s.a2);
    s.c2 = min(max(s.c2, s.c1), s.a2);
    // Far-field split (farfield.h): lines at least kFarRatio half-widths from the tile
    // centre, and beyond every possible core, go to the series; without it the ranges
    // [a1,f1) and [f2,a2) are empty.
    s.f1 = s.a1;
    s.f2 = s.a2;
    if (farfield && bounded)
    {
        s.f1 = min(max(value[6], s.a1), s.c1);
        s.f2 = min(max(value[7], s.c2), s.a2);
    }
    *out = s;
}

__global__ __launch_bounds__(256) void schedule_kernel(const LineTableView table,
                                                       const LevelScalars * __restrict__ levels,
                                                       const GridSpec g, const Tiling tiling,
                                                       const int farfield,
                                                       TileSchedule * __restrict__ schedule)
{
    const int thread = blockIdx.x*blockDim.x + threadIdx.x;
    const int tile = thread >> 3;
    const int level = blockIdx.y;
    const bool active = tile < tiling.n_tiles;
    schedule_tile(table, levels[level], g, tiling, farfield, tile, thread & 7, active,
                  schedule + (long long)level*tiling.n_tiles + (active ? tile : 0));
}

// prepare_kernel and schedule_kernel in one launch (they are independent: the schedule only
// reads the sorted wavenumbers and the level scalars): blocks [0, prepare_blocks) prepare
// lines, the rest schedule tiles.  For a few levels the level scalars travel as kernel
// arguments (no host-to-device copy in front of the launch) and block 0 of each level stores
// them where the later kernels read them.
constexpr int kInlineLevels = 4;
struct InlineLevels
{
    LevelScalars level[kInlineLevels];
};

__device__ __forceinline__ void prepare_block(const LineTableView & t, const LevelScalars & lv,
                                              const GridSpec & g, const RangeRule & rule,
                                              int block, int level,
                                              LineWing * __restrict__ wing,
                                              LineCore * __restrict__ core,
                                              double * __restrict__ derived,
                                              unsigned long long * __restrict__ evals)
{
    const long long j = (long long)block*blockDim.x + threadIdx.x;
    unsigned long long count = 0;
    if (j < t.n_lines)
    {
        LineWing w;
        LineCore c;
        const double nu = t.nu[j];
        const int slot = t.iso_slot[j];     // -1: no mass / partition function (never accepted)
        const bool ok = slot >= 0 && line_accepted(rule, nu, t.row[j]);
        double * d = derived != nullptr ? derived + ((long long)level*t.n_lines + j)*8 : nullptr;
        const int status = prepare_line(lv, g, nu, t.sw[j], t.gamma_air[j],
                                        t.gamma_self[j], t.n_air[j], t.elower[j],
                                        t.delta_air[j], max(slot, 0), ok, w, c, d);
        wing[(long long)level*t.n_lines + j] = w;
        core[(long long)level*t.n_lines + j] = c;
        if (status == 1 && w.last >= w.first)
        {
            count = (unsigned long long)(w.last - w.first + 1);
        }
    }
    if (evals != nullptr)
    {
        // Closed-form count of the reference's inner-loop iterations (spectra.c:48-62).
        for (int offset = 32; offset > 0; offset >>= 1)
        {
            count += __shfl_down(count, offset, 64);
        }
        if ((threadIdx.x & 63) == 0 && count != 0)
        {
            atomicAdd(evals, count);
        }
    }
}

__global__ __launch_bounds__(256) void prologue_kernel(const LineTableView t,
                                                       LevelScalars * __restrict__ levels,
                                                       const InlineLevels inline_levels,
                                                       const int use_inline,
                                                       const GridSpec g, const RangeRule rule,
                                                       const Tiling tiling, const int farfield,
                                                       const int prepare_blocks,
                                                       LineWing * __restrict__ wing,
                                                       LineCore * __restrict__ core,
                                                       TileSchedule * __restrict__ schedule,
                                                       double * __restrict__ derived,
                                                       unsigned long long * __restrict__ evals)
{
    const int level = blockIdx.y;
    const LevelScalars & lv = use_inline ? inline_levels.level[level] : levels[level];
    if (use_inline && blockIdx.x == 0)
    {
        // sizeof(LevelScalars) is a multiple of 8: copy as doubles, one per thread.
        const double * from = reinterpret_cast<const double *>(&inline_levels.level[level]);
        double * to = reinterpret_cast<double *>(levels + level);
        for (int i = threadIdx.x; i < (int)(sizeof(LevelScalars)/8); i += blockDim.x) to[i] = from[i];
    }
    if ((int)blockIdx.x < prepare_blocks)
    {
        prepare_block(t, lv, g, rule, blockIdx.x, level, wing, core, derived, evals);
        return;
    }
    const int thread = (blockIdx.x - prepare_blocks)*blockDim.x + threadIdx.x;
    const int tile = thread >> 3;
    const bool active = tile < tiling.n_tiles;
    schedule_tile(t, lv, g, tiling, farfield, tile, thread & 7, active,
                  schedule + (long long)level*tiling.n_tiles + (active ? tile : 0));
}

}  // namespace lbl
