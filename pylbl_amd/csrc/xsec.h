// ARTS-crossfit absorption cross-sections on the device: the counterpart of the reference's
// mechanism slot 2 (pyLBL/spectroscopy.py:199-203 -> CrossSection.absorption_coefficient,
// pyLBL/arts_crossfit/cross_section.py:19-48).
//
// A molecule has a few bands; a band has a frequency grid [Hz] and four fit coefficients per
// frequency.  Per level the reference evaluates p00 + p10 T + p01 P + p20 T^2 on the band's
// grid, removes negative values without changing the band integral
// (xsec_aux_functions.py:80-121), interpolates linearly to the user's grid converted to Hz
// (scipy interp1d, zero outside the band) and adds the bands up.
//
// Two kernels:
//   xsec_model_kernel    workgroup = one (band, level): fit, the two sums the clipping rule
//                        needs (workgroup reduction), rescale, slope of every interval;
//   xsec_interp_kernel   thread = PT points of the user's grid x LV levels: the band grids are
//                        not uniform, so the interval comes from a binary search -- narrowed
//                        per workgroup to the few intervals its (ascending) points span.
//                        HBM-bound like continuum_interp_kernel: 16 B per point and level.
#pragma once

#include <hip/hip_runtime.h>

#include "continuum.h"      // load_points / store_points
#include "wave_ops.h"       // wave_search

namespace lbl {

constexpr int kMaxXsecBands = 16;
constexpr int kXsecStage = 1024;         // frequencies of a band staged in LDS per workgroup and band
constexpr double kSpeedOfLight = 299792458.0;       // cross_section.py:31
constexpr double kBoltzmann = 1.38064852e-23;       // spectroscopy.py:15

struct XsecBand
{
    int size;               // frequencies
    long long offset;       // of this band in the concatenated frequency axis
};

struct XsecSet
{
    int n_bands;
    int total;              // sum of sizes = workspace doubles per level
    XsecBand band[kMaxXsecBands];
};

struct XsecLevel
{
    double t, p;            // [K], [Pa]
    double density;         // P x /(kb T) [m-3] (spectroscopy.py:18-29), 1 if not scaled
};

constexpr int kModelThreads = 1024;

__device__ __forceinline__ double block_sum(double value, double * scratch)
{
    // 16 wavefronts; fixed order, so the result does not depend on timing.
    for (int offset = 32; offset > 0; offset >>= 1) value += __shfl_down(value, offset, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = value;
    __syncthreads();
    double total = 0.;
    for (int w = 0; w < kModelThreads/64; ++w) total += scratch[w];
    return total;
}

// grid = (bands, levels), kModelThreads threads.  coeffs: per band [4][size] at 4*offset.
__global__ __launch_bounds__(kModelThreads) void xsec_model_kernel(XsecSet set,
                                                         const double * __restrict__ fgrid,
                                                         const double * __restrict__ coeffs,
                                                         const XsecLevel * __restrict__ levels,
                                                         double * __restrict__ values,
                                                         double * __restrict__ slopes)
{
    __shared__ double scratch[kModelThreads/64];
    const XsecBand b = set.band[blockIdx.x];
    const XsecLevel s = levels[blockIdx.y];
    const double * c = coeffs + 4*b.offset;
    double * out = values + (long long)blockIdx.y*set.total + b.offset;
    double * slope = slopes + (long long)blockIdx.y*set.total + b.offset;
    const double * f = fgrid + b.offset;
    const double tt = s.t*s.t;
    double raw_sum = 0., kept_sum = 0., negatives = 0.;
    for (int j = threadIdx.x; j < b.size; j += kModelThreads)
    {
        // Rows added in the reference's order (xsec_aux_functions.py:49-75).
        const double value = ((c[j]*1. + c[b.size + j]*s.t) + c[2*b.size + j]*s.p) +
                             c[3*b.size + j]*tt;
        out[j] = value;
        raw_sum += value;
        if (value < 0.) negatives += 1.; else kept_sum += value;
    }
    raw_sum = block_sum(raw_sum, scratch);
    kept_sum = block_sum(kept_sum, scratch);
    negatives = block_sum(negatives, scratch);
    if (negatives > 0.)
    {
        // xsec_aux_functions.py:104-119: clip; rescale only when the integral was not negative.
        const double weight = raw_sum >= 0. ? raw_sum/kept_sum : 1.;
        const bool rescale = raw_sum >= 0.;
        for (int j = threadIdx.x; j < b.size; j += kModelThreads)
        {
            double value = out[j];
            value = value < 0. ? 0. : value;
            out[j] = rescale ? value*weight : value;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < b.size; j += kModelThreads)
    {
        slope[j] = j + 1 < b.size ? (out[j + 1] - out[j])/(f[j + 1] - f[j]) : 0.;
    }
}

// First index with f[index] >= x within [lo, hi] (numpy.searchsorted, side="left").
__device__ __forceinline__ int lower_bound(const double * __restrict__ f, int lo, int hi, double x)
{
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (f[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// grid = (points / (256 PT), levels / LV).  out[level][i] (+)= density * sum over bands.
template <int PT, int LV>
__global__ __launch_bounds__(256) void xsec_interp_kernel(XsecSet set,
                                                          const double * __restrict__ fgrid,
                                                          const double * __restrict__ values,
                                                          const double * __restrict__ slopes,
                                                          const XsecLevel * __restrict__ levels,
                                                          GridForm form,
                                                          long long n, int n_levels, int ascending,
                                                          double * __restrict__ out,
                                                          long long level_stride, int accumulate)
{
    __shared__ int window[kMaxXsecBands][2];
    // The frequencies of a band that this workgroup's points can fall between, staged once per
    // band (round 6): the binary search then walks LDS, where rounds 1-5 walked HBM -- six dependent
    // round trips per point, 77 % of the kernel's wave-cycles spent waiting
    // (profiles/r05b_valu_counters.json).
    __shared__ double staged[kXsecStage];
    // Points of a thread as in continuum_interp_kernel: neighbouring pairs, pairs 512 apart.
    static_assert(PT == 1 || PT % 2 == 0, "points come in pairs");
    const long long block_first = (long long)blockIdx.x*(256*PT);
    auto point_index = [&](int p) -> long long {
        return PT == 1 ? block_first + threadIdx.x
                       : block_first + (p >> 1)*512 + 2*threadIdx.x + (p & 1);
    };
    const int level0 = blockIdx.y*LV;
    const int count = min(LV, n_levels - level0);
    // Search window of every band for this workgroup's points: exact when the grid ascends.  Two
    // searches per band over its whole frequency axis -- the serial head of every workgroup: one
    // wavefront each, 64 probes a step (wave_ops.h: two dependent loads for 1 300 frequencies where
    // a thread's binary search took eleven), two bands at a time.
    if (ascending)
    {
        const int wave = threadIdx.x >> 6;
        const long long block_last = min(block_first + 256*PT, n) - 1;
        const double x_lo = wavenumber_at(form, block_first)*kSpeedOfLight*100;
        const double x_hi = wavenumber_at(form, block_last)*kSpeedOfLight*100;
        for (int k0 = 0; k0 < set.n_bands; k0 += 2)
        {
            const int k = k0 + (wave >> 1);
            if (k < set.n_bands)
            {
                const XsecBand b = set.band[k];
                const double * f = fgrid + b.offset;
                // (a band none of this workgroup's points can lie in -- most bands for most
                // workgroups -- gets the empty window and is skipped below without a search)
                const bool touches = x_hi >= f[0] && x_lo <= f[b.size - 1];
                const int found = touches ? wave_search<false>(f, 0, b.size, (wave & 1) ? x_hi : x_lo)
                                          : -1;
                if ((threadIdx.x & 63) == 0) window[k][wave & 1] = found;
            }
        }
    }
    else if ((int)threadIdx.x < set.n_bands)
    {
        window[threadIdx.x][0] = 0;
        window[threadIdx.x][1] = set.band[threadIdx.x].size;
    }
    double x[PT], total[PT][LV], before[PT][LV];
    load_wavenumbers<PT>(form, n, point_index, x);
#pragma unroll
    for (int p = 0; p < PT; ++p)
    {
        x[p] = x[p]*kSpeedOfLight*100;      // Hz, as the reference forms it (cross_section.py:32)
    }
#pragma unroll
    for (int l = 0; l < LV; ++l)
    {
        double row[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) row[p] = 0.;
        if (accumulate && l < count)
        {
            load_points<PT>(out + (long long)(level0 + l)*level_stride, n, point_index, row, 0.);
        }
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
            total[p][l] = 0.;
            before[p][l] = row[p];
        }
    }
    for (int k = 0; k < set.n_bands; ++k)
    {
        const XsecBand b = set.band[k];
        const double * f = fgrid + b.offset;
        __syncthreads();        // window[] is set; the previous band's frequencies have been read
        // The search reads f[w0 .. w1-1], the interval's lower knot may be f[w0-1].
        const int w0 = window[k][0], w1 = window[k][1];
        if (w0 < 0) continue;   // (the same for every thread of the workgroup)
        const int base = max(w0 - 1, 0);
        const int length = w1 - base;
        const bool in_lds = length <= kXsecStage;
        if (in_lds)
        {
            for (int t = threadIdx.x; t < length; t += blockDim.x) staged[t] = f[base + t];
        }
        __syncthreads();
        const double f_first = f[0], f_last = f[b.size - 1];
        int j[PT];
        double dx[PT];
        bool inside[PT];
        bool any = false;
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
            inside[p] = (x[p] >= f_first) && (x[p] <= f_last);     // zero outside (fill_value)
            int at = 1;
            if (inside[p])
            {
                // scipy interp1d (linear): searchsorted, clipped to [1, size-1]; the interval
                // is (at-1, at).
                at = in_lds ? base + lower_bound(staged, w0 - base, w1 - base, x[p])
                            : lower_bound(f, w0, w1, x[p]);
                at = at < 1 ? 1 : (at > b.size - 1 ? b.size - 1 : at);
            }
            j[p] = at - 1;
            const bool knot_staged = in_lds && at - 1 >= base && at - 1 < w1;
            const double knot = !inside[p] ? 0. : (knot_staged ? staged[at - 1 - base] : f[at - 1]);
            dx[p] = inside[p] ? x[p] - knot : 0.;
            any = any || inside[p];
        }
        if (__ballot(any) == 0ull) continue;
        const long long table = (long long)level0*set.total + b.offset;
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
#pragma unroll
            for (int l = 0; l < LV; ++l)
            {
                const long long at = table + (long long)(l < count ? l : 0)*set.total + j[p];
                const double value = slopes[at]*dx[p] + values[at];
                if (inside[p]) total[p][l] += value;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LV; ++l)
    {
        if (l >= count) continue;
        const double density = levels[level0 + l].density;
        double row[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
            const double value = density*total[p][l];
            row[p] = accumulate ? value + before[p][l] : value;
        }
        store_points<PT>(out + (long long)(level0 + l)*level_stride, n, point_index, row);
    }
}

}  // namespace lbl
