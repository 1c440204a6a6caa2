// Far-field series ("K2f", optional: engine option farfield=1).
//
// Reference statement: the same sums as accumulate.h -- voigt.c:82 / :24 for every line
// whose window covers a tile and whose distance from it puts the whole tile in the line's
// Lorentz wing.  For such a line, with u = v - u0 (u0 the tile centre), a = nu' - u0 and
// D = a^2 + gamma^2,
//
//     S gamma/pi / ((v - nu')^2 + gamma^2) = b / (D - 2 a u + u^2) = sum_k q_k u^k,
//     q_0 = b/D,  q_1 = (2a/D) q_0,  q_k = (2a/D) q_{k-1} - (1/D) q_{k-2},
//
// a power series with convergence radius sqrt(D) >= |a|.  The schedule kernel only hands
// over lines with |a| >= kFarRatio x the tile's half width, so |u|/|a| <= 1/4 and kFarTerms
// = 21 terms leave a relative truncation below ~1.5e-11 (measured on the 5 M-point benchmark:
// max 2.8e-12 against the direct kernel) -- five orders inside the 1e-6 parity bar.  The series
// of all far lines of a tile are ADDED coefficient by coefficient (3 flops per line and term
// instead of ~5 flops per line and grid point), and the accumulate kernel evaluates the summed
// polynomial once per point.
//
// This is an algorithmic shortcut, not the reference's evaluation order: it is off by
// default and `bench.py` reports it separately.
#pragma once

#include <hip/hip_runtime.h>

#include "accumulate.h"
#include "line_prep.h"
#include "wave_ops.h"

namespace lbl {

// Two levels (round 3).  Every tile used to walk the ~50 cm-1 worth of lines whose windows cover
// it (~3800 of a 400 k-line table: 63 fp64 operations each, and 133 KB of records through L2),
// yet most of those lines are far enough away to be expanded about the centre of a GROUP of kFarGroup adjacent tiles just as well: a line at least
// kFarRatio group half-widths from the group's centre (and beyond every core) converges at the
// same rate over the whole group as a tile's lines do over the tile.  So
//   per group (farfield_series_kernel, first half): the very far lines, read once for kFarGroup
//                           tiles, and the cut points that say which lines those are;
//   per tile (second half): what is left -- the lines between the tile's own limit
//                           and the group's, and the few whose windows cover this tile but not
//                           the whole group -- plus the group's polynomial re-centred on the
//                           tile (a Taylor shift by the distance d of the two centres,
//                           s'_j = sum_{k>=j} C(k,j) s_k d^(k-j): an identity of polynomials,
//                           so nothing is truncated twice);
// and the accumulate kernel evaluates one polynomial per point as before.  Reads and arithmetic
// drop ~2.9x for kFarGroup = 4.
constexpr int kFarGroup = 4;       // = wavefronts per workgroup of farfield_series_kernel
struct GroupCuts
{
    int a1, g1, g2, a2;     // the group's very far lines: [a1, g1) below it, [g2, a2) above it
};

__device__ __forceinline__ void group_bounds(const Tiling & tiling, int group, int n_per_v, int n,
                                             long long & i0, long long & i1)
{
    const int t0 = group*kFarGroup;
    const int t1 = min(t0 + kFarGroup, tiling.n_tiles) - 1;
    long long unused;
    tile_bounds(tiling, t0, n_per_v, n, i0, unused);
    tile_bounds(tiling, t1, n_per_v, n, unused, i1);
}

// Series about u0 of every THREADS-th line of a list (index ranges laid end to end by `line_at`),
// from `first` up to `end`, added term by term into c.
template <int THREADS, typename LineAt>
__device__ __forceinline__ void series_terms(const LineWing * __restrict__ w, int first, int end,
                                             LineAt line_at, double u0, double (&c)[kFarTerms])
{
    auto add_line = [&](const LineWing & l) {
        const double a = l.centre - u0;
        const double r = rcp_newton(__builtin_fma(a, a, l.g2));
        const double s = (a + a)*r;
        double q0 = l.bl*r;
        double q1 = s*q0;
        c[0] += q0;
        c[1] += q1;
#pragma unroll
        for (int k = 2; k < kFarTerms; ++k)
        {
            const double q2 = __builtin_fma(s, q1, -(r*q0));
            c[k] += q2;
            q0 = q1;
            q1 = q2;
        }
    };
    int at = first;
    for (; at + 3*THREADS < end; at += 4*THREADS)   // four loads in flight
    {
        const LineWing l0 = w[line_at(at)];
        const LineWing l1 = w[line_at(at + THREADS)];
        const LineWing l2 = w[line_at(at + 2*THREADS)];
        const LineWing l3 = w[line_at(at + 3*THREADS)];
        add_line(l0);
        add_line(l1);
        add_line(l2);
        add_line(l3);
    }
    for (; at + THREADS < end; at += 2*THREADS)
    {
        const LineWing l0 = w[line_at(at)];
        const LineWing l1 = w[line_at(at + THREADS)];
        add_line(l0);
        add_line(l1);
    }
    if (at < end)
    {
        add_line(w[line_at(at)]);
    }
}

// One 256-thread workgroup per (group of kFarGroup = 4 tiles, level): first the series of the
// group's very far lines about the group's centre (all four wavefronts, into LDS), then every
// wavefront takes one tile of the group -- the lines between the tile's own limit and the group's,
// and the few whose windows cover this tile but not the whole group -- and adds the group's
// polynomial re-centred on its tile.  (Rounds 3-5 ran the two steps as two launches,
// farfield_group_kernel and farfield_kernel, with the group's cuts and series going through HBM in
// between, and a whole workgroup per tile.)
__global__ __launch_bounds__(256) void farfield_series_kernel(
    const LineWing * __restrict__ wing, const TileSchedule * __restrict__ schedule,
    const double * __restrict__ nu, const LevelScalars * __restrict__ levels, long long n_lines,
    Tiling tiling, int n_groups, int v0, int n_per_v, int n, double dv,
    double * __restrict__ far_series)
{
    __shared__ double wave_sum[4][kFarTerms];
    __shared__ GroupCuts shared_cuts;
    __shared__ double group_term[kFarTerms];
    const int level = blockIdx.y;
    // Neighbouring groups read almost the same lines: one contiguous eighth of the spectrum per
    // XCD (workgroup b runs on XCD b mod 8) keeps them in that XCD's L2 (speed only).
    const int per_xcd = (n_groups + 7) >> 3;
    const int group = (blockIdx.x & 7)*per_xcd + (blockIdx.x >> 3);
    if (group >= n_groups)
    {
        return;
    }
    long long i0, i1;
    group_bounds(tiling, group, n_per_v, n, i0, i1);
    const double group_centre = tile_centre(v0, dv, i0, i1);
    const int t0 = group*kFarGroup, t1 = min(t0 + kFarGroup, tiling.n_tiles);
    if (threadIdx.x < 128)
    {
        // The schedule's far limit (schedule_tile: cases 6 and 7) with the group's half width:
        // wavefront 0 finds the cut below the group, wavefront 1 the one above it.
        const bool below = threadIdx.x < 64;
        const LevelScalars & lv = levels[level];
        const double v_lo = (double)v0 + (double)i0*dv, v_hi = (double)v0 + (double)i1*dv;
        const double half = 0.5*(v_hi - v_lo);
        const double kk = lv.core_reach;
        const bool bounded = kk < 0.5;
        const double core = bounded ? kk*(v_hi + lv.shift_max)/(1. - kk)*(1. + 1.e-9) +
                                      lv.shift_max + 1.e-9 : 0.;
        const double radius = fmax(kFarRatio*half, core + half)*(1. + 1.e-9) + 1.e-6;
        // Lines that cover every tile of the group: the tightest of the tiles' ranges.
        int a1 = 0, a2 = (int)n_lines, f1 = (int)n_lines, f2 = 0;
        for (int t = t0; t < t1; ++t)
        {
            const TileSchedule sc = schedule[(long long)level*tiling.n_tiles + t];
            a1 = max(a1, sc.a1);
            a2 = min(a2, sc.a2);
            f1 = min(f1, sc.f1);
            f2 = max(f2, sc.f2);
        }
        // Never beyond what every tile hands to its own series (f1, f2), never outside [a1, a2):
        // the search is confined to that range.
        if (below)
        {
            const int g1 = bounded && f1 > a1
                ? wave_search<true>(nu, a1, f1, group_centre - radius - lv.shift_max) : a1;
            if (threadIdx.x == 0)
            {
                shared_cuts.a1 = a1;
                shared_cuts.g1 = g1;
            }
        }
        else
        {
            const int g2 = bounded && a2 > f2
                ? wave_search<false>(nu, f2, a2, group_centre + radius + lv.shift_max) : a2;
            if (threadIdx.x == 64)
            {
                shared_cuts.a2 = a2;
                shared_cuts.g2 = g2;
            }
        }
    }
    __syncthreads();
    // (wave-uniform: into scalar registers)
    GroupCuts gc;
    gc.a1 = __builtin_amdgcn_readfirstlane(shared_cuts.a1);
    gc.g1 = __builtin_amdgcn_readfirstlane(shared_cuts.g1);
    gc.g2 = __builtin_amdgcn_readfirstlane(shared_cuts.g2);
    gc.a2 = __builtin_amdgcn_readfirstlane(shared_cuts.a2);
    const LineWing * __restrict__ w = wing + (long long)level*n_lines;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The group's own series about the group's centre: all 256 threads, summed over the 64 lanes
    // (wave_ops.h: 21 butterflies through the LDS crossbar were most of this kernel's time), then
    // over the 4 wavefronts, in a fixed order.
    {
        const int left = max(gc.g1 - gc.a1, 0);
        const int total = left + max(gc.a2 - gc.g2, 0);
        double c[kFarTerms];
#pragma unroll
        for (int k = 0; k < kFarTerms; ++k) c[k] = 0.;
        series_terms<256>(w, (int)threadIdx.x, total,
                          [&](int at) { return at < left ? gc.a1 + at : gc.g2 + (at - left); },
                          group_centre, c);
        int index;
        bool valid;
        wave_sums(c, index, valid);
        if (valid) wave_sum[wave][index] = c[0];
        __syncthreads();
        if (threadIdx.x < kFarTerms)
        {
            const int k = threadIdx.x;
            group_term[k] = (wave_sum[0][k] + wave_sum[1][k]) + (wave_sum[2][k] + wave_sum[3][k]);
        }
        __syncthreads();
    }
    // The tiles of the group, one per wavefront, side by side: each about its own centre, what only
    // the tile can take, plus the group's polynomial re-centred on it.
    static_assert(kFarGroup == 4, "one tile of the group per wavefront");
    const int tile = t0 + wave;
    if (tile >= t1) return;
    const TileSchedule sc = schedule[(long long)level*tiling.n_tiles + tile];
    long long q0, q1;
    tile_bounds(tiling, tile, n_per_v, n, q0, q1);
    const double u0 = tile_centre(v0, dv, q0, q1);
    // This tile's far lines [a1, f1) and [f2, a2) without the group's [gc.a1, gc.g1),
    // [gc.g2, gc.a2): four pieces laid end to end.
    const int l1 = min(max(gc.a1, sc.a1), sc.f1);       // [a1, l1): cover this tile, not the group
    const int l2 = min(max(gc.g1, l1), sc.f1);          // [l2, f1): nearer than the group's limit
    const int r2 = max(min(gc.a2, sc.a2), sc.f2);       // [r2, a2)
    const int r1 = max(min(gc.g2, r2), sc.f2);          // [f2, r1)
    const int n0 = l1 - sc.a1, n1 = sc.f1 - l2, n2 = r1 - sc.f2, n3 = sc.a2 - r2;
    double c[kFarTerms];
#pragma unroll
    for (int k = 0; k < kFarTerms; ++k) c[k] = 0.;
    series_terms<64>(w, lane, n0 + n1 + n2 + n3,
                     [&](int at) {
                         if (at < n0) return sc.a1 + at;
                         at -= n0;
                         if (at < n1) return l2 + at;
                         at -= n1;
                         if (at < n2) return sc.f2 + at;
                         return r2 + (at - n2);
                     }, u0, c);
    int j;
    bool valid;
    wave_sums(c, j, valid);
    if (valid)
    {
        // Lane `lane` holds term j of the tile's own series; the group's polynomial in
        // w = u + d about this tile's centre (a Taylor shift) is added to it.
        const double d = u0 - group_centre;
        double shifted = 0., weight = 1.;       // weight = C(j+m, j) d^m
        for (int m = 0; j + m < kFarTerms; ++m)
        {
            shifted = __builtin_fma(group_term[j + m], weight, shifted);
            weight *= d*((double)(j + m + 1)*rcp_newton((double)(m + 1)));
        }
        far_series[((long long)level*tiling.n_tiles + tile)*kFarTerms + j] = c[0] + shifted;
    }
}

}  // namespace lbl
