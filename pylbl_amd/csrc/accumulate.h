// The hot kernel ("K2"): sums every line's profile onto the wavenumber grid.
//
// Reference statement: the row loop of pyLBL/c_lib/absorption.c:76-86 calling
// spectra() -> voigt() (pyLBL/c_lib/voigt.c:21-25, :74-189), i.e. for every line a
// read-modify-write sweep over its +-cut_off window of k[].  Here the loop nest is turned
// inside out (a gather): one wavefront owns a tile of 64*P consecutive grid points, keeps
// the P partial sums of each lane in registers, and walks the lines whose windows overlap
// the tile.  k[] is written exactly once; the grid itself is never read (v[i] = v0 + i*dv
// is formed in registers the way absorption.c:36-40 forms it).
//
// Line scalars are wave-uniform, so they travel through the scalar unit (s_load into
// SGPRs) rather than through vector registers or LDS: every VALU instruction of the inner
// loop takes its line operand straight from an SGPR pair.
//
// Lines are kept sorted by wavenumber.  For a tile the schedule kernel (tile_schedule.h)
// gives five cut points lo <= a1 <= c1 <= c2 <= a2 <= hi into that order:
//   [a1,c1) and [c2,a2): windows certainly cover the whole tile and the tile is certainly
//                        in the Lorentz far wing of the line  -> branch-free fast loop,
//                        four lines per reciprocal;
//   [lo,a1), [c1,c2), [a2,hi): anything else -> per-line, per-64-point-row decisions
//                        (window clipping, exact reference region chain near the core).
#pragma once

#include <hip/hip_runtime.h>

#include "line_prep.h"
#include "voigt_profile.h"

namespace lbl {

struct alignas(32) TileSchedule
{
    int lo, a1, c1, c2, a2, hi;
    int pad0, pad1;
};

struct AccumulateArgs
{
    const LineWing * wing;          // [levels][n_lines]
    const LineCore * core;          // [levels][n_lines]
    const TileSchedule * schedule;  // [levels][n_tiles]
    const LevelScalars * levels;    // [levels]
    const double * pedestal_cell;   // [levels][cells] or nullptr
    const double * pedestal_point;  // [levels][cells] or nullptr
    int n_cells;                    // vn - v0
    double * k;                     // [levels][level_stride]
    long long level_stride;
    long long n_lines;
    int n_tiles;
    int n;                          // grid points
    int v0, n_per_v;
    double dv;
    int scale_density;
    int accumulate;
};

template <int P>
__device__ __forceinline__ void fast_range(const LineWing * __restrict__ wing, int j0, int j1,
                                           const double (&v)[P], double (&acc)[P])
{
    int j = j0;
    for (; j + 4 <= j1; j += 4)
    {
        const LineWing l1 = wing[j], l2 = wing[j + 1], l3 = wing[j + 2], l4 = wing[j + 3];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            acc[p] += lorentz_four(v[p], l1.centre, l1.g2, l1.bl, l2.centre, l2.g2, l2.bl,
                                   l3.centre, l3.g2, l3.bl, l4.centre, l4.g2, l4.bl);
        }
    }
    for (; j < j1; ++j)
    {
        const LineWing l = wing[j];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            acc[p] += lorentz_one(v[p], l.centre, l.g2, l.bl);
        }
    }
}

template <int P>
__device__ __forceinline__ void general_range(const LineWing * __restrict__ wing,
                                              const LineCore * __restrict__ core,
                                              int j0, int j1, int i0, int i1, int lane,
                                              const double (&v)[P], double (&acc)[P])
{
    for (int j = j0; j < j1; ++j)
    {
        const LineWing l = wing[j];
        if (l.last < i0 || l.first > i1)
        {
            continue;   // also skips empty windows (first > last)
        }
        const LineCore c = core[j];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            const int r0 = i0 + p*64;
            const int r1 = r0 + 63;
            if (l.last < r0 || l.first > r1)
            {
                continue;
            }
            const int i = r0 + lane;
            const bool inside = (i >= l.first) && (i <= l.last);
            const double d = v[p] - l.centre;
            double value;
            if (c.core_last < r0 || c.core_first > r1)
            {
                // Whole row beyond xlim0 (or a y >= 70.55 line): voigt.c:82 / :24.
                value = l.bl*rcp_newton(__builtin_fma(d, d, l.g2));
            }
            else
            {
                // voigt.c:76,188 with the reference's region chain.
                const double xi = d*c.repwid;
                value = c.amp*wells_profile(xi, c.y);
            }
            acc[p] += inside ? value : 0.;
        }
    }
}

template <int P>
__global__ __launch_bounds__(256) void accumulate_kernel(const AccumulateArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int level = blockIdx.y;
    const int tile = blockIdx.x*4 + wave;
    if (tile >= a.n_tiles)
    {
        return;
    }
    const int i0 = tile*(64*P);
    int i1 = i0 + 64*P - 1;
    if (i1 > a.n - 1) i1 = a.n - 1;
    const TileSchedule sc = a.schedule[(long long)level*a.n_tiles + tile];
    const LineWing * __restrict__ wing = a.wing + (long long)level*a.n_lines;
    const LineCore * __restrict__ core = a.core + (long long)level*a.n_lines;

    double v[P], acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        // absorption.c:39: v[i] = v0 + i*dv (product rounded, then the sum).
        const int i = i0 + p*64 + lane;
        const double step = (double)i*a.dv;
        v[p] = (double)a.v0 + step;
        acc[p] = 0.;
    }

    general_range<P>(wing, core, sc.lo, sc.a1, i0, i1, lane, v, acc);
    fast_range<P>(wing, sc.a1, sc.c1, v, acc);
    general_range<P>(wing, core, sc.c1, sc.c2, i0, i1, lane, v, acc);
    fast_range<P>(wing, sc.c2, sc.a2, v, acc);
    general_range<P>(wing, core, sc.a2, sc.hi, i0, i1, lane, v, acc);

    double scale = 1.;
    if (a.scale_density)
    {
        scale = a.levels[level].density;
    }
    double * __restrict__ out = a.k + (long long)level*a.level_stride;
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        const int i = i0 + p*64 + lane;
        if (i < a.n)
        {
            double value = acc[p];
            if (a.pedestal_cell != nullptr)
            {
                // Sum of the pedestals of every line whose window holds point i
                // (spectra.c:66-78 factorised; see pedestal.h).  Windows start and end on
                // integer wavenumbers, so the sum is constant inside a 1 cm-1 cell and has
                // one extra bin of lines on the integer point that closes a window.
                const int cell = i/a.n_per_v;
                const bool on_integer = (cell*a.n_per_v == i);
                const double * table = on_integer ? a.pedestal_point : a.pedestal_cell;
                value -= table[(long long)level*a.n_cells + cell];
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

}  // namespace lbl
