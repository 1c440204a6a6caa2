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
};

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

// One thread per (line, level).
__global__ __launch_bounds__(256) void prepare_kernel(const LineTableView t,
                                                      const LevelScalars * __restrict__ levels,
                                                      const GridSpec g, const RangeRule rule,
                                                      LineWing * __restrict__ wing,
                                                      LineCore * __restrict__ core,
                                                      double * __restrict__ derived,
                                                      unsigned long long * __restrict__ evals)
{
    const long long j = (long long)blockIdx.x*blockDim.x + threadIdx.x;
    const int level = blockIdx.y;
    unsigned long long count = 0;
    if (j < t.n_lines)
    {
        LineWing w;
        LineCore c;
        const double nu = t.nu[j];
        const int slot = t.iso_slot[j];     // -1: no mass / partition function (never accepted)
        const bool ok = slot >= 0 && line_accepted(rule, nu, t.row[j]);
        double * d = derived != nullptr ? derived + ((long long)level*t.n_lines + j)*8 : nullptr;
        const int status = prepare_line(levels[level], g, nu, t.sw[j], t.gamma_air[j],
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

__device__ inline int first_not_below(const double * __restrict__ nu, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (nu[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ inline int first_above(const double * __restrict__ nu, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (nu[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per (tile, level).  A line's window is fixed by b = floor(nu + p*delta):
// [b-cut, b+cut+1] cm-1, both ends included (spectra.c:48-62).  With |p*delta| <= shift_max
// the tests on b become tests on nu, which is the sort key:
//   may overlap the tile    b in [ceil(i0/npv)+v0-cut-1, floor(i1/npv)+v0+cut]
//   covers the whole tile   b in [ceil(i1/npv)+v0-cut-1, floor(i0/npv)+v0+cut]
//   tile may touch |x|<xlim0    |nu - tile| <= core_reach*nu (+ shift)
__global__ __launch_bounds__(256) void schedule_kernel(const double * __restrict__ nu, int n_lines,
                                                       const LevelScalars * __restrict__ levels,
                                                       const GridSpec g, const Tiling tiling,
                                                       const int farfield,
                                                       TileSchedule * __restrict__ schedule)
{
    const int tile = blockIdx.x*blockDim.x + threadIdx.x;
    const int level = blockIdx.y;
    const int n_tiles = tiling.n_tiles;
    if (tile >= n_tiles)
    {
        return;
    }
    long long i0, i1;
    tile_bounds(tiling, tile, g.n_per_v, g.n, i0, i1);
    const double smax = levels[level].shift_max;
    const long long npv = g.n_per_v;
    const double any_lo = (double)((i0 + npv - 1)/npv + g.v0 - g.cut_off - 1);
    const double any_hi = (double)(i1/npv + g.v0 + g.cut_off);
    const double full_lo = (double)((i1 + npv - 1)/npv + g.v0 - g.cut_off - 1);
    const double full_hi = (double)(i0/npv + g.v0 + g.cut_off);
    TileSchedule s;
    s.lo = first_not_below(nu, n_lines, any_lo - smax);
    s.hi = first_not_below(nu, n_lines, any_hi + 1. + smax);
    s.a1 = first_not_below(nu, n_lines, full_lo + smax);
    s.a2 = first_not_below(nu, n_lines, full_hi + 1. - smax);
    const double v_lo = (double)g.v0 + (double)i0*g.dv;
    const double v_hi = (double)g.v0 + (double)i1*g.dv;
    const double kk = levels[level].core_reach;
    if (kk < 0.5)
    {
        const double reach = kk*(v_hi + smax)/(1. - kk)*(1. + 1.e-9) + smax + 1.e-9;
        s.c1 = first_not_below(nu, n_lines, v_lo - reach);
        s.c2 = first_above(nu, n_lines, v_hi + reach);
    }
    else
    {
        s.c1 = 0;
        s.c2 = n_lines;
    }
    if (s.hi < s.lo) s.hi = s.lo;
    s.a1 = min(max(s.a1, s.lo), s.hi);
    s.a2 = min(max(s.a2, s.a1), s.hi);
    // Normally every line that may have its core in the tile also covers it completely
    // (cores reach ~1 cm-1, windows 25); with a tiny cut-off the clipping ranges can hold such
    // lines too, and the accumulate kernel then takes them through its core path.
    s.clip_core = (s.c1 < s.a1 && s.a1 > s.lo) || (s.c2 > s.a2 && s.a2 < s.hi) ? 1 : 0;
    s.pad[0] = s.pad[1] = s.pad[2] = 0;
    s.c1 = min(max(s.c1, s.a1), s.a2);
    s.c2 = min(max(s.c2, s.c1), s.a2);
    // Far-field split (farfield.h): lines at least kFarRatio half-widths from the tile
    // centre, and beyond every possible core, go to the series; without it the ranges
    // [a1,f1) and [f2,a2) are empty.
    s.f1 = s.a1;
    s.f2 = s.a2;
    if (farfield && kk < 0.5)
    {
        const double half = 0.5*(v_hi - v_lo);
        const double u0 = tile_centre(g.v0, g.dv, i0, i1);
        const double core = kk*(v_hi + smax)/(1. - kk)*(1. + 1.e-9) + smax + 1.e-9;
        const double radius = fmax(kFarRatio*half, core + half)*(1. + 1.e-9) + 1.e-6;
        s.f1 = first_above(nu, n_lines, u0 - radius - smax);
        s.f2 = first_not_below(nu, n_lines, u0 + radius + smax);
        s.f1 = min(max(s.f1, s.a1), s.c1);
        s.f2 = min(max(s.f2, s.c2), s.a2);
    }
    schedule[(long long)level*n_tiles + tile] = s;
}

}  // namespace lbl
