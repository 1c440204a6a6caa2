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
// max 2.8e-12 against the direct kernel) -- five orders inside the 1e-6 parity bar.  The series of all far lines of a tile are ADDED coefficient by coefficient
// (3 flops per line and term instead of ~5 flops per line and grid point), and the
// accumulate kernel evaluates the summed polynomial once per point.
//
// This is an algorithmic shortcut, not the reference's evaluation order: it is off by
// default and `bench.py` reports it separately.
#pragma once

#include <hip/hip_runtime.h>

#include "accumulate.h"

namespace lbl {

// One 256-thread workgroup per (tile, level); thread = line (strided), then a block sum.
__global__ __launch_bounds__(256) void farfield_kernel(const LineWing * __restrict__ wing,
                                                       const TileSchedule * __restrict__ schedule,
                                                       long long n_lines, Tiling tiling,
                                                       int v0, int n_per_v, int n, double dv,
                                                       double * __restrict__ far_series)
{
    __shared__ double wave_sum[4][kFarTerms];
    const int level = blockIdx.y;
    // Neighbouring tiles read almost the same lines: one contiguous eighth of the spectrum
    // per XCD keeps them in that XCD's L2 (speed only).
    const int per_xcd = (tiling.n_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7)*per_xcd + (blockIdx.x >> 3);
    if (tile >= tiling.n_tiles)
    {
        return;
    }
    const TileSchedule sc = schedule[(long long)level*tiling.n_tiles + tile];
    const LineWing * __restrict__ w = wing + (long long)level*n_lines;
    long long i0, i1;
    tile_bounds(tiling, tile, n_per_v, n, i0, i1);
    const double u0 = tile_centre(v0, dv, i0, i1);

    double c[kFarTerms];
#pragma unroll
    for (int k = 0; k < kFarTerms; ++k) c[k] = 0.;
    const int left = sc.f1 - sc.a1;
    const int total = left + (sc.a2 - sc.f2);
    for (int at = threadIdx.x; at < total; at += 256)
    {
        const int j = at < left ? sc.a1 + at : sc.f2 + (at - left);
        const LineWing l = w[j];
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
    }
    // Sum over the 64 lanes (butterfly), then over the 4 wavefronts: fixed order.
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kFarTerms; ++k)
    {
        double value = c[k];
        for (int offset = 32; offset > 0; offset >>= 1)
        {
            value += __shfl_xor(value, offset, 64);
        }
        if (lane == 0) wave_sum[wave][k] = value;
    }
    __syncthreads();
    if (threadIdx.x < kFarTerms)
    {
        const int k = threadIdx.x;
        far_series[((long long)level*tiling.n_tiles + tile)*kFarTerms + k] =
            (wave_sum[0][k] + wave_sum[1][k]) + (wave_sum[2][k] + wave_sum[3][k]);
    }
}

}  // namespace lbl
