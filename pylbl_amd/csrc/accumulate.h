// The hot kernel ("K2"): sums every line's profile onto the wavenumber grid.
//
// Reference statement: the row loop of pyLBL/c_lib/absorption.c:76-86 calling
// spectra() -> voigt() (pyLBL/c_lib/voigt.c:21-25, :74-189), i.e. for every line a
// read-modify-write sweep over its +-cut_off window of k[].  Here the loop nest is turned
// inside out (a gather): a workgroup owns a tile of 64*P consecutive grid points, every lane
// keeps P partial sums in registers, and the workgroup's four wavefronts walk the lines
// whose windows overlap the tile (a quarter of them each).  k[] is written exactly once; the
// grid itself is never read (v[i] = v0 + i*dv is formed in registers the way
// absorption.c:36-40 forms it).
//
// Line scalars are wave-uniform, so they travel through the scalar unit (s_load into
// SGPRs) rather than through vector registers or LDS: every VALU instruction of the inner
// loop takes its line operand straight from an SGPR pair.
//
// Lines are kept sorted by wavenumber.  For a tile the schedule kernel (tile_schedule.h)
// gives cut points lo <= a1 <= f1 <= c1 <= c2 <= f2 <= a2 <= hi into that order:
//   [f1,c1) and [c2,f2): windows certainly cover the whole tile and the tile is certainly
//                        in the Lorentz far wing of the line  -> branch-free fast loop,
//                        eight lines per reciprocal;
//   [a1,f1) and [f2,a2): the same, and far enough away for the optional power series of
//                        farfield.h (empty, f1 = a1 and f2 = a2, when that is off);
//   [c1,c2):             the tile may hold points of the line's core -> core_lines(): per line
//                        a bit mask of the rows that lie wholly in w4 region 1 (one rational
//                        function, no selection), of the rows that may hold core points (the
//                        reference's region chain lane by lane) and of the rest (far wing); the
//                        inner points (|x| < xlim1) of all of them in a pass of their own,
//                        inner_ranges();
//   [lo,a1) and [a2,hi): the window ends inside the tile -> clipped_ranges(): eight lines with
//                        the same window as one far-wing group with the row masked; what does not
//                        group, and the 0-3 left-over lines of the fast ranges, line by line
//                        (general_line).
// Work is handed out as WorkItems (a tile, or a share of a heavy tile's lines), heaviest first.
#pragma once

#include <hip/hip_runtime.h>

#include "line_prep.h"
#include "voigt_profile.h"

namespace lbl {

// How the grid is cut into tiles of at most 64*P points.  When the points per wavenumber
// allow it (little padding), tiles are aligned to the 1 cm-1 cells: line windows begin and
// end on integer wavenumbers (spectra.c:48-62), so an aligned tile is never cut by a window
// edge and every overlapping line covers it completely (bar the single closing point).
struct Tiling
{
    int aligned;        // 1: `per_cell` tiles of `length` points inside every 1 cm-1 cell
    int per_cell;
    int length;         // points per tile (<= 64*P)
    int n_tiles;
};

__host__ __device__ inline void tile_bounds(const Tiling & t, int tile, int n_per_v, int n,
                                            long long & i0, long long & i1)
{
    if (t.aligned)
    {
        const int cell = tile/t.per_cell;
        const int sub = tile - cell*t.per_cell;
        i0 = (long long)cell*n_per_v + (long long)sub*t.length;
        i1 = i0 + t.length - 1;
        const long long cell_end = (long long)(cell + 1)*n_per_v - 1;
        if (i1 > cell_end) i1 = cell_end;
    }
    else
    {
        i0 = (long long)tile*t.length;
        i1 = i0 + t.length - 1;
    }
    if (i1 > n - 1) i1 = n - 1;
}

struct alignas(32) TileSchedule
{
    int lo, a1, c1, c2, a2, hi;
    int f1, f2;         // [a1,f1) and [f2,a2): far enough for the series of farfield.h
};

// Centre of a tile in wavenumber: the expansion point of the far-field series.
__host__ __device__ inline double tile_centre(int v0, double dv, long long i0, long long i1)
{
    const double v_lo = (double)v0 + (double)i0*dv;
    const double v_hi = (double)v0 + (double)i1*dv;
    return 0.5*(v_lo + v_hi);
}

#ifndef LBL_WING_GROUP
#define LBL_WING_GROUP 8        // far-wing lines per reciprocal: 4 or 8
#endif
#ifndef LBL_FAR_TERMS
#define LBL_FAR_TERMS 21
#define LBL_FAR_RATIO 4.
#endif
constexpr int kFarTerms = LBL_FAR_TERMS;     // series order 20: truncation <= ~1.5e-11 relative at ratio 1/4
constexpr double kFarRatio = LBL_FAR_RATIO;  // far lines are at least 4 tile half-widths from the centre

// One unit of work for a workgroup: part `part` of `parts` of tile `tile`'s lines.  Dense
// spectral bands give some tiles many times the average number of lines; the host splits
// those into several items (and orders items heaviest first) so that no single workgroup
// holds up the kernel.  Items of a split tile store plain partial sums into
// partial[slot + part], which combine_kernel adds up in a fixed order.
struct alignas(16) WorkItem
{
    int tile;
    int part, parts;
    int slot;           // first partial slot of the tile (parts > 1), else -1
};

struct AccumulateArgs
{
    const LineWing * wing;          // [levels][n_lines]
    const LineCore * core;          // [levels][n_lines]
    const TileSchedule * schedule;  // [levels][n_tiles]
    const LevelScalars * levels;    // [levels]
    const WorkItem * items;         // [n_items]
    const double * far_series;      // [levels][n_tiles][kFarTerms] or nullptr (farfield.h)
    double * partial;               // [levels][partial_slots][64*P] sums of split tiles
    long long partial_slots;
    double * k;                     // [levels][level_stride]
    long long level_stride;
    long long n_lines;
    Tiling tiling;
    int n_tiles;
    int n;                          // grid points
    int v0, n_per_v;
    double dv;
    double v0_real;                 // (double)v0: a kernel argument lives in scalar registers, the
                                    // conversion would hold two vector registers all kernel long
    int scale_density;
    int accumulate;
    int inner_everywhere;           // 0: only the core-range lines can have inner points in a tile
#ifdef LBL_ABLATE
    int ablate;                     // diagnostics build only (-DLBL_ABLATE, scripts/ablate_*.sh): 1 skips
                                    // the general ranges, 2 the fast ranges, ..., 64 sends the
                                    // clipped windows line by line (accumulate_tile)
#endif
};

// Parts of the kernel switched off for timing diagnostics (results are then wrong): only in a
// library built with -DLBL_ABLATE; the shipped one has no such switch anywhere.
#ifdef LBL_ABLATE
#define LBL_ABLATED(a, bits) (((a).ablate & (bits)) != 0)
#else
#define LBL_ABLATED(a, bits) false
#endif

}  // namespace lbl

#include "accumulate_lines.h"

namespace lbl {

// Contiguous share `part` of `parts` of the index range [j0, j1), in units of 1 << unit_shift lines
// (the last share also takes the remainder).  The cut after share x is f(x) = trunc(units * (x *
// (1/parts))) in single precision: any non-decreasing f does -- every wavefront forms its two cuts
// from the same expression, so the shares tile the range whatever the rounding -- and this one
// costs a handful of instructions where the exact 64-bit quotient cost the scalar unit some 150,
// ten times per wavefront (a third of all scalar instructions of the far-field kernel).
// `inverse` = 1.f/parts.
__device__ __forceinline__ void share_of(int j0, int j1, int part, int parts, float inverse,
                                         int unit_shift, int & begin, int & end)
{
    const int units = (j1 - j0) >> unit_shift;      // units of 1, 4 or 8 lines
    const float scale = (float)units;
    const int cut0 = min((int)(scale*((float)part*inverse)), units);
    const int cut1 = min((int)(scale*((float)(part + 1)*inverse)), units);
    begin = j0 + (cut0 << unit_shift);
    end = (part + 1 == parts) ? j1 : j0 + (cut1 << unit_shift);
}

// One 256-thread workgroup per work item.  Its four wavefronts own the SAME 64*P grid points
// and split the item's lines four ways (every cut-point range is first cut into the item's
// share, then quartered), so an item is four independent wavefronts for the dispatcher; the
// four partial sums meet in LDS and each wavefront finishes P/4 of the rows (scaling, the
// one store of k -- or of the item's partial sums when the tile was split).
template <int P>
__device__ __forceinline__ void accumulate_tile(const AccumulateArgs & a)
{
    __shared__ double partial[4][P][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int level = blockIdx.y;
    const WorkItem item = a.items[blockIdx.x];
    const int tile = item.tile;
    long long first_point, last_point;
    tile_bounds(a.tiling, tile, a.n_per_v, a.n, first_point, last_point);
    const int i0 = (int)first_point, i1 = (int)last_point;
    const TileSchedule sc = a.schedule[(long long)level*a.n_tiles + tile];
    const LineWing * __restrict__ wing = a.wing + (long long)level*a.n_lines;
    const LineCore * __restrict__ core = a.core + (long long)level*a.n_lines;

    double v[P], acc[P];
    // This wavefront's block of the LDS sums: what the inner pass adds to, and where the
    // register sums join it at the end.
    double * slab = &partial[wave][0][0];
    __shared__ InnerStage inner_stage[4];
    // (The point indices as doubles: the first one converted, the others 64, 128, ... more --
    // exact, and no integer per row stays behind in a vector register for the rest of the kernel.)
    const double first_index = (double)(i0 + lane);
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        // absorption.c:39: v[i] = v0 + i*dv (product rounded, then the sum).
        const double step = (first_index + (double)(p*64))*a.dv;
        v[p] = a.v0_real + step;
        acc[p] = 0.;
    }

    // This wavefront's share of each of the five cut-point ranges: the item's part of the
    // range, quartered.
    const int piece = item.part*4 + wave, pieces = item.parts*4;
    GeneralList g;
    int fa0, fa1, fb0, fb1, e;
    // (The clipped windows are handed out eight lines at a time where clipped_ranges() sums them
    // in groups of eight.)
    const bool clipped_in_groups = !a.inner_everywhere && !LBL_ABLATED(a, 64);
    const float inverse = 1.f/(float)pieces;
    share_of(sc.lo, sc.a1, piece, pieces, inverse, clipped_in_groups ? 3 : 0, g.begin[0], e);
    g.count[0] = e - g.begin[0];
    share_of(sc.c1, sc.c2, piece, pieces, inverse, 0, g.begin[1], e);
    g.count[1] = e - g.begin[1];
    share_of(sc.a2, sc.hi, piece, pieces, inverse, clipped_in_groups ? 3 : 0, g.begin[2], e);
    g.count[2] = e - g.begin[2];
    // [a1,f1) and [f2,a2) are summed by the far-field series (empty when that is off).
    share_of(sc.f1, sc.c1, piece, pieces, inverse, 2, fa0, fa1);
    share_of(sc.c2, sc.f2, piece, pieces, inverse, 2, fb0, fb1);
    // Left-over lines of the far-wing ranges (fewer than four each) take the general path.
    g.begin[3] = fa0 + ((fa1 - fa0) & ~3);
    g.count[3] = (fa1 - fa0) & 3;
    g.begin[4] = fb0 + ((fb1 - fb0) & ~3);
    g.count[4] = (fb1 - fb0) & 3;
    if (LBL_ABLATED(a, 4)) { g.count[0] = 0; g.count[2] = 0; }    // diagnostics: no clipping lines
    if (LBL_ABLATED(a, 8)) { g.count[3] = 0; g.count[4] = 0; }    // ... no left-overs of the fast ranges
    if (LBL_ABLATED(a, 16)) { g.count[1] = 0; }                   // ... no core lines
    if (!LBL_ABLATED(a, 2))
    {
        fast_ranges<P>(wing, fa0, fa1, fb0, fb1, v, acc);
    }
    bool slab_in_use = false;
    if (!LBL_ABLATED(a, 1))
    {
        GeneralList walk = g;
        if (clipped_in_groups)
        {
            clipped_ranges<P>(wing, core, g.begin[0], g.count[0], g.begin[2], g.count[2], i0, i1,
                              lane, v, acc);
            walk.count[0] = walk.count[2] = 0;
        }
        general_ranges<P>(wing, core, walk, i0, i1, lane, v, acc);
        // (32: diagnostics, leaves the inner points out.)  Levels at which no line of the call
        // can have an inner point -- the host's bound on y, engine.hip -- skip the look.
        if (!LBL_ABLATED(a, 32) && a.levels[level].inner_possible != 0.)
        {
            // With the reference's cut-off of 25 cm-1 a line whose window clips the tile, or
            // whose wing covers it, is far from having its core there: only the core-range
            // lines need the second look.  (Cut-offs of a wavenumber or two: everything.)
            GeneralList inner = g;
            if (!a.inner_everywhere)
            {
                inner.count[0] = inner.count[2] = inner.count[3] = inner.count[4] = 0;
            }
            slab_in_use = inner_ranges(wing, core, inner, inner_stage[wave], i0, i1, a.v0,
                                       a.n_per_v, a.dv, lane, slab, P);
        }
    }

#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        partial[wave][p][lane] = slab_in_use ? partial[wave][p][lane] + acc[p] : acc[p];
    }
    __syncthreads();

    // Far-field series of this tile (added once per tile, by part 0): sum_k s_k u^k with
    // u = v - tile centre.
    const bool add_series = a.far_series != nullptr && item.part == 0;
    const double * __restrict__ series = a.far_series +
        ((long long)level*a.n_tiles + tile)*kFarTerms;
    const double u0 = tile_centre(a.v0, a.dv, first_point, last_point);
    auto series_at = [&](double vv) {
        const double u = vv - u0;
        double value = series[kFarTerms - 1];
#pragma unroll
        for (int k = kFarTerms - 2; k >= 0; --k)
        {
            value = __builtin_fma(value, u, series[k]);
        }
        return value;
    };

    if (item.parts > 1)
    {
        // A split tile: plain sums to this item's slot; combine_kernel finishes.
        double * __restrict__ slot = a.partial +
            ((long long)level*a.partial_slots + item.slot + item.part)*(64*P);
        for (int p = wave; p < P; p += 4)
        {
            // Fixed order of the four partial sums: results do not depend on scheduling.
            double value = (partial[0][p][lane] + partial[1][p][lane]) +
                           (partial[2][p][lane] + partial[3][p][lane]);
            if (add_series)
            {
                const int i = i0 + p*64 + lane;
                value += series_at(a.v0_real + (double)i*a.dv);
            }
            slot[p*64 + lane] = value;
        }
        return;
    }

    double scale = 1.;
    if (a.scale_density)
    {
        scale = a.levels[level].density;
    }
    double * __restrict__ out = a.k + (long long)level*a.level_stride;
    for (int p = wave; p < P; p += 4)
    {
        const int i = i0 + p*64 + lane;
        if (i <= i1)
        {
            double value = (partial[0][p][lane] + partial[1][p][lane]) +
                           (partial[2][p][lane] + partial[3][p][lane]);
            if (add_series)
            {
                value += series_at(a.v0_real + (double)i*a.dv);
            }
            value *= scale;
            if (a.accumulate)
            {
                value += out[i];
            }
            out[i] = value;
        }
    }
}

template <int P>
__global__ __launch_bounds__(256) void accumulate_kernel(const AccumulateArgs a)
{
    accumulate_tile<P>(a);
}

// Occupancy hints on the two forms that carry the benchmarks.  The eight-points-per-lane form
// (far-field series on: few lines per tile are evaluated point by point, so the tile is made
// wide): left alone the scheduler spends 108 VGPRs on it (4 wavefronts per SIMD); asked for 4 per
// SIMD or more it made do with 78 and the far-field step gained 10-11 %
// (profiles/r03_ab_occupancy.txt); with clipped_ranges() in the kernel that hint gives 82 (5
// wavefronts), asked for 6 it gives 80 and another 4 % (profiles/r03_ab_clipped.txt).  The
// four-points-per-lane form took the same hint in round 3 (81 VGPRs without it, one wavefront per
// SIMD fewer) -- at the price of five registers spilled around the far-wing loop: stored and
// re-read once per wavefront, 96 MB of scratch writes per launch on the 5 M-point workload, 70 %
// of the kernel's HBM writes.  Round 4 removed what was spilled instead: the point indices of the
// rows (now formed as doubles from the first one) and (double)v0 (now a kernel argument, i.e. in
// scalar registers).  <4> compiles to 71 VGPRs without a spill (seven wavefronts per SIMD), <8> to
// 80 with two instead of four spilled; same speed within +-1 % (profiles/r04_ab_spill.txt), where
// the hint (6, 6) -- also free of spills, 77 VGPRs -- lost 2 %.
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6)))
void accumulate_kernel<8>(const AccumulateArgs a)
{
    accumulate_tile<8>(a);
}

template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6)))
void accumulate_kernel<4>(const AccumulateArgs a)
{
    accumulate_tile<4>(a);
}

// Adds up the partial sums of the split tiles, part 0 first (fixed order), and writes k.
// One 64-thread group per (64-point row of a split tile, level).
struct SplitTile
{
    int tile, parts, slot, pad;
};

__global__ __launch_bounds__(256) void combine_kernel(const AccumulateArgs a,
                                                      const SplitTile * __restrict__ tiles,
                                                      int n_split, int points)
{
    const int level = blockIdx.y;
    const int rows = points >> 6;
    const int unit = blockIdx.x*4 + (threadIdx.x >> 6);     // (split tile, row)
    const int lane = threadIdx.x & 63;
    if (unit >= n_split*rows) return;
    const SplitTile t = tiles[unit/rows];
    const int p = unit - (unit/rows)*rows;
    long long first_point, last_point;
    tile_bounds(a.tiling, t.tile, a.n_per_v, a.n, first_point, last_point);
    const long long i = first_point + p*64 + lane;
    if (i > last_point) return;
    const double * slot = a.partial + ((long long)level*a.partial_slots + t.slot)*points +
                          p*64 + lane;
    // Same order of additions as a plain loop over the parts; the loads of eight parts are issued
    // together (on small grids every one of them is a miss).
    double value = 0.;
    int part = 0;
    for (; part + 8 <= t.parts; part += 8)
    {
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = slot[(long long)(part + j)*points];
#pragma unroll
        for (int j = 0; j < 8; ++j) value += x[j];
    }
    for (; part < t.parts; ++part)
    {
        value += slot[(long long)part*points];
    }
    if (a.scale_density) value *= a.levels[level].density;
    double * out = a.k + (long long)level*a.level_stride;
    if (a.accumulate) value += out[i];
    out[i] = value;
}

}  // namespace lbl
