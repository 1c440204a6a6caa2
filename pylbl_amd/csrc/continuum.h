// MT-CKD continua on the device: the counterpart of the reference's mechanism slot 1
// (pyLBL/spectroscopy.py:193-197 -> BandedContinuum.spectra, pyLBL/mt_ckd/utils.py:157-174).
//
// A continuum is a short list of bands.  Each band has coefficient columns on its own
// uniform coarse wavenumber grid and one formula (Continuum.spectra of the 16 classes in
// water_vapor.py, carbon_dioxide.py, nitrogen.py, oxygen.py, ozone.py); the reference
// evaluates the formula on the coarse grid, interpolates linearly to the user's grid with
// numpy.interp (zero outside the band), multiplies by 100 (cm-1 -> m-1) and adds the bands up.
//
// Two kernels:
//   band_spectra_kernel      thread = coarse point of one (level, band): a few hundred to a
//                            few thousand points per band, negligible time;
//   continuum_interp_kernel  thread = point of the user's grid for one level: reads the
//                            wavenumber (8 B), looks the enclosing coarse interval of every
//                            band up by arithmetic (the coarse grids are uniform), writes the
//                            extinction (8 B).  HBM-bound: 16 B per point and level.
#pragma once

#include <hip/hip_runtime.h>

namespace lbl {

// Band formulas.  The numbering is part of the C ABI (include/lbl_amd.h, LBL_BAND_*).
enum BandKind : int
{
    kBandH2OSelf = 0,       // water_vapor.py:23-32     columns: bs296, bs260
    kBandH2OForeign = 1,    // water_vapor.py:71-78     columns: bfh2o, scale
    kBandCO2 = 2,           // carbon_dioxide.py:33-39  columns: bfco2, chi factor, T exponent
    kBandN2Rotation = 3,    // nitrogen.py:19-30        columns: ct_296, ct_220, sf_296, sf_220
    kBandN2Fundamental = 4, // nitrogen.py:40-54        columns: xn2_272, xn2_228, a_h2o
    kBandN2Overtone = 5,    // nitrogen.py:63-68        columns: xn2
    kBandO2Fundamental = 6, // oxygen.py:22-30          columns: o2_f, o2_t
    kBandO2NIR = 7,         // oxygen.py:39-47          columns: o2_inf1
    kBandO2NIR2 = 8,        // oxygen.py:69-74          columns: analytic shape / wavenumber
    kBandO2NIR3 = 9,        // oxygen.py:90-94          columns: o2_inf3
    kBandO2Visible = 10,    // oxygen.py:105-111        columns: o2_invis
    kBandO2Herzberg = 11,   // oxygen.py:126-130        columns: analytic shape
    kBandO2UV = 12,         // oxygen.py:138-141        columns: o2_infuv
    kBandO3Chappuis = 13,   // ozone.py:23-28           columns: x_o3, y_o3, z_o3
    kBandO3Hartley = 14,    // ozone.py:46-52           columns: o3_hh0, o3_hh1, o3_hh2
    kBandO3UV = 15,         // ozone.py:68-71           columns: o3_huv
    kBandKinds = 16
};

constexpr int kMaxBands = 8;

struct Band
{
    int kind;
    int size;               // coarse points
    double lower;           // coarse wavenumber j = lower + j*resolution (utils.py:142-143)
    double resolution;
    double per_step;        // 1/resolution, formed once on the host (correctly rounded there as
                            // here: the same bits as the division every thread used to do)
    long long column[4];    // offsets of the coefficient columns in the table, -1 = unused
    long long spectrum;     // offset of this band's coarse spectrum in a level's workspace
};

struct BandSet
{
    int n_bands;
    int coarse_points;      // sum of sizes = workspace doubles per level
    Band band[kMaxBands];
};

// What the formulas need from a level (utils.py:16-42 evaluated on the host).
struct ContinuumLevel
{
    double t;               // [K]
    double p;               // [mb]
    double dry;             // dry-air number density [cm-3]
    double air;             // air number density [cm-3]
    double self;            // mole fraction of the gas the continuum belongs to
    double h2o, o2, n2;     // mole fractions the formulas refer to by name
};

namespace mtckd {
constexpr double kLoschmidt = 2.6867775e19;     // utils.py:7
constexpr double kP0 = 1013.25;                 // utils.py:8
constexpr double kC2 = 1.4387752;               // utils.py:9
constexpr double kT0 = 296.;                    // utils.py:10
constexpr double kT273 = 273.15;                // utils.py:11
}

// utils.py:45-59.  Its x <= 0.01 branch is always overwritten by the x <= 10 one.
__device__ __forceinline__ double radiation_term(double w, double t)
{
    const double x = w/(t/mtckd::kC2);
    if (x <= 10.)
    {
        const double e = exp(-x);
        return w*(1. - e)/(1. + e);
    }
    return w;
}

// Continuum.spectra of one band at coarse point j [cm-1].
__device__ inline double band_value(const Band & b, const double * __restrict__ table,
                                    const ContinuumLevel & s, int j)
{
    using namespace mtckd;
    const double w = b.lower + (double)j*b.resolution;
    const double rad = radiation_term(w, s.t);
    const double * c0 = table + b.column[0];
    const double * c1 = table + b.column[1];
    const double * c2 = table + b.column[2];
    const double * c3 = table + b.column[3];
    switch (b.kind)
    {
    case kBandH2OSelf:
    {
        const double n = s.dry*s.h2o;
        return n*(n/s.air)*(s.p/kP0)*(kT0/s.t)*1.e-20*rad*
               c0[j]*pow(c1[j]/c0[j], (s.t - kT0)/(260. - kT0));
    }
    case kBandH2OForeign:
    {
        const double n = s.dry*s.h2o;
        return (1. - (n/s.air))*(s.p/kP0)*(kT0/s.t)*1.e-20*n*rad*c1[j]*c0[j];
    }
    case kBandCO2:
    {
        const double n = s.dry*s.self;
        return n*1.e-20*(s.p/kP0)*(kT0/s.t)*rad*c1[j]*pow(s.t/246., c2[j])*c0[j];
    }
    case kBandN2Rotation:
    {
        const double tau = (s.dry*s.n2/kLoschmidt)*(s.p/kP0)*(kT273/s.t);
        const double f = (s.t - kT0)/(220. - kT0);
        const double c = c0[j]*pow(c1[j]/c0[j], f);
        const double sf = c2[j]*pow(c3[j]/c2[j], f);
        const double fo2 = (sf - 1.)*s.n2/s.o2;
        return tau*rad*c*(s.n2 + fo2*s.o2 + s.h2o);
    }
    case kBandN2Fundamental:
    {
        const double tau = (s.dry*s.n2/kLoschmidt)*(s.p/kP0)*(kT273/s.t);
        const double xt = (1./s.t - 1./272.)/(1./228. - 1./272.);
        double c = 0.;
        if (j > 0 && j < b.size - 1) c = c0[j]*pow(c1[j]/c0[j], xt);
        c = c/w;
        const double with_o2 = (1.294 - 0.4545*s.t/kT0)*c;
        const double with_h2o = (9./7.)*c2[j]*c;
        return tau*rad*(c*s.n2 + s.o2*with_o2 + s.h2o*with_h2o);
    }
    case kBandN2Overtone:
    {
        const double tau = (s.dry*s.n2/kLoschmidt)*(s.p/kP0)*(kT273/s.t)*(s.n2 + s.o2 + s.h2o);
        return tau*rad*c0[j]/w;
    }
    case kBandO2Fundamental:
    {
        const double tau = s.dry*s.o2*1.e-20*(s.p/kP0)*(kT273/s.t);
        return tau*rad*(1.e20/kLoschmidt)*c0[j]*exp(c1[j]*((1./kT0) - (1./s.t)))/w;
    }
    case kBandO2NIR:
    {
        const double tau = (s.dry*s.o2/kLoschmidt)*(s.p/kP0)*(kT273/s.t)*
                           ((1./0.446)*s.o2 + (0.3/0.446)*s.n2 + s.h2o);
        return tau*rad*c0[j]/w;
    }
    case kBandO2NIR2:
    {
        const double n = s.dry*s.o2;
        const double adj = (n/s.air)*(1./s.o2)*n*1.e-20*(s.p/kP0)*(kT0/s.t);
        return adj*rad*c0[j];
    }
    case kBandO2NIR3:
        return (s.dry*s.o2/kLoschmidt)*(s.p/kP0)*(kT273/s.t)*rad*c0[j]/w;
    case kBandO2Visible:
    {
        const double n = s.dry*s.o2;
        const double adj = (n/s.air)*n*1.e-20*(s.p/kP0)*(kT273/s.t);
        const double factor = 1./(kLoschmidt*1.e-20*(55.*kT273/kT0)*(55.*kT273/kT0)*89.5);
        return adj*rad*factor*c0[j]/w;
    }
    case kBandO2Herzberg:
        return 1.e-20*(s.dry*s.o2)*rad*(1. + 0.83*(s.p/kP0)*(kT273/s.t))*c0[j]/w;
    case kBandO2UV:
        return 1.e-20*(s.dry*s.o2)*rad*c0[j]/w;
    case kBandO3Chappuis:
    {
        const double dt = s.t - kT273;
        return 1.e-20*(s.dry*s.self)*rad*(c0[j] + c1[j]*dt + c2[j]*dt*dt)/w;
    }
    case kBandO3Hartley:
    {
        const double dt = s.t - kT273;
        return 1.e-20*(s.dry*s.self)*rad*(c0[j]/w)*(1. + c1[j]*dt + c2[j]*dt*dt);
    }
    case kBandO3UV:
        return (s.dry*s.self)*rad*c0[j]/w;
    default:
        return 0.;
    }
}

// grid = (coarse points of the widest band / 256, bands, levels).  Besides the spectrum the
// kernel stores the slope of every coarse interval, (fp[j+1]-fp[j])/(xp[j+1]-xp[j]): what
// numpy.interp precomputes when the target grid is longer than the table.
__global__ __launch_bounds__(256) void band_spectra_kernel(BandSet set,
                                                           const double * __restrict__ table,
                                                           const ContinuumLevel * __restrict__ levels,
                                                           double * __restrict__ coarse,
                                                           double * __restrict__ slopes)
{
    const Band & b = set.band[blockIdx.y];
    const int j = blockIdx.x*256 + threadIdx.x;
    if (j >= b.size) return;
    const ContinuumLevel s = levels[blockIdx.z];
    const long long at = (long long)blockIdx.z*set.coarse_points + b.spectrum + j;
    const double here = band_value(b, table, s, j);
    coarse[at] = here;
    double slope = 0.;
    if (j + 1 < b.size)
    {
        const double xj = b.lower + (double)j*b.resolution;
        const double xn = b.lower + (double)(j + 1)*b.resolution;
        slope = (band_value(b, table, s, j + 1) - here)/(xn - xj);
    }
    slopes[at] = slope;
}

// continuum_interp_kernel<PT, LV>: PT points per thread (pairs of neighbours, 16 B accesses)
// for LV levels.

// PT values of one thread from / to a row of n doubles: neighbouring pairs move as one
// 16-byte access when the row is 16-byte aligned (the pair index is even by construction).
template <int PT, typename Index>
__device__ __forceinline__ void load_points(const double * __restrict__ row, long long n,
                                            const Index & index, double (&value)[PT], double fill)
{
    const bool aligned = ((unsigned long long)row & 15ull) == 0ull;
#pragma unroll
    for (int p = 0; p < PT; p += 2)
    {
        const long long i = index(p);
        if (PT > 1 && aligned && i + 1 < n)
        {
            const double2 pair = *reinterpret_cast<const double2 *>(row + i);
            value[p] = pair.x;
            if (p + 1 < PT) value[p + 1] = pair.y;
        }
        else
        {
            value[p] = i < n ? row[i] : fill;
            if (p + 1 < PT) value[p + 1] = index(p + 1) < n ? row[index(p + 1)] : fill;
        }
    }
}

template <int PT, typename Index>
__device__ __forceinline__ void store_points(double * __restrict__ row, long long n,
                                             const Index & index, const double (&value)[PT])
{
    const bool aligned = ((unsigned long long)row & 15ull) == 0ull;
#pragma unroll
    for (int p = 0; p < PT; p += 2)
    {
        const long long i = index(p);
        if (PT > 1 && aligned && i + 1 < n)
        {
            *reinterpret_cast<double2 *>(row + i) = double2{value[p], p + 1 < PT ? value[p + 1] : 0.};
        }
        else
        {
            if (i < n) row[i] = value[p];
            if (p + 1 < PT && index(p + 1) < n) row[index(p + 1)] = value[p + 1];
        }
    }
}

// The caller's spectral grid as the interpolation kernels see it.  The reference hands its
// continua and cross-sections whatever array the user built (spectroscopy.py:195,203), almost
// always numpy.arange(lo, hi, step), whose elements are first + i*(second - first) in double
// precision.  lbl_grid_load checks exactly that, element by element; where it holds the kernels
// form the wavenumber in registers -- the same bits -- instead of reading 8 bytes per point and
// waiting for them before the table look-ups can start.
struct GridForm
{
    const double * wavenumber;
    int arithmetic;             // 1: wavenumber[i] == start + (double)i*step for every i (verified)
    double start, step;
};

// (double)i for a grid index: grids have fewer than 2^31 points (lbl_grid_load), and a 32-bit
// conversion is one instruction where the 64-bit one is a dozen.
__device__ __forceinline__ double index_as_double(long long i)
{
    return (double)(int)i;
}

__device__ __forceinline__ double wavenumber_at(const GridForm & form, long long i)
{
    if (form.arithmetic)
    {
        const double offset = index_as_double(i)*form.step;     // (product rounded, then the sum)
        return form.start + offset;
    }
    return form.wavenumber[i];
}

template <int PT, typename Index>
__device__ __forceinline__ void load_wavenumbers(const GridForm & form, long long n,
                                                 const Index & index, double (&x)[PT])
{
    if (form.arithmetic)
    {
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
            const long long i = index(p);
            const double offset = index_as_double(i)*form.step;
            x[p] = i < n ? form.start + offset : __builtin_nan("");
        }
        return;
    }
    load_points<PT>(form.wavenumber, n, index, x, __builtin_nan(""));
}

// numpy.interp(x, xp, fp, left=0, right=0) for every band, xp[j] = lower + j*resolution
// (numpy/core/src/multiarray/compiled_base.c, arr_interp: interval by search, then
// slope*(x - xp[j]) + fp[j], fp[j] itself on a knot, the same fallbacks for a NaN result).
// The coarse grids are uniform, so the interval comes from one multiplication and at most a
// step of correction; it does not depend on the level, so a thread keeps its points for
// kInterpLevels levels: the wavenumber is read once, the interval found once per band.
// The kernel is bound by the latency of two dependent round trips (wavenumber from HBM, then
// the coarse spectrum from L2), so every thread carries kInterpPoints independent points.
//
// grid = (points / (256 PT), levels / LV).  extinction[level][i] (+)= 100 * sum over bands.
// numpy.interp's interval for x in a band with knots xp[j] = lower + j*resolution: the largest j
// with xp[j] <= x (0 outside the band), and xp[j] itself.  One multiplication lands on it or next
// to it; the knots beside the guess decide (formed exactly as xp is, so the answer is numpy's
// search result whatever the product's rounding did).  Every knot is formed once: the two loops
// used to re-form theirs at every test, a quarter of the kernel's instructions.
__device__ __forceinline__ int interval_of(const Band & b, int last, double x, bool inside,
                                           double & xj)
{
    int at = inside ? (int)((x - b.lower)*b.per_step) : 0;
    at = at < 0 ? 0 : (at > last ? last : at);
    xj = b.lower + (double)at*b.resolution;
    if (inside)
    {
        while (at > 0 && xj > x)
        {
            --at;
            xj = b.lower + (double)at*b.resolution;
        }
        double xn = b.lower + (double)(at + 1)*b.resolution;
        while (at < last && xn <= x)
        {
            ++at;
            xj = xn;
            xn = b.lower + (double)(at + 1)*b.resolution;
        }
    }
    return at;
}

// The run of consecutive grid points a wavefront holds (G points per thread: 64 G points, lane l
// holding points 2l, 2l + 1 of every 128 of them): the wavenumbers of its first and last point.  `usable`: the grid is arithmetic with a positive step (x rises with the
// index) and the whole run lies inside the grid.
struct WaveRun
{
    double lo, hi;
    bool usable;
};

// Grid index of point p of this thread (see WaveRun).
template <int G>
__device__ __forceinline__ long long wave_point(long long block_first, int p)
{
    if (G == 1) return block_first + threadIdx.x;
    return block_first + (long long)G*(threadIdx.x & ~63) + (p >> 1)*128 + 2*(threadIdx.x & 63) +
           (p & 1);
}

template <int G>
__device__ __forceinline__ WaveRun wave_run(const GridForm & form, long long n, long long block_first)
{
    const long long first = block_first + (long long)G*(threadIdx.x & ~63);
    const long long last = first + 64*G - 1;
    WaveRun run;
    run.usable = form.arithmetic != 0 && form.step > 0. && last < n;
    run.lo = run.usable ? wavenumber_at(form, first) : 0.;
    run.hi = run.usable ? wavenumber_at(form, last) : 0.;
    return run;
}

// One band's contribution for G points (a pair of neighbours, or one point) x LV levels of a
// thread: total[p][l] += 100 * interp.  `row` = offset of the band's coarse spectrum of the first of
// the thread's levels; a level's spectra are `level_points` doubles apart.  Returns false when the
// whole wavefront lies outside the band (nothing was read).
//
// Round 5: on a fine grid a wavefront's run of 64 G points is far narrower than a coarse interval
// (0.13-0.26 cm-1 against 2-10 cm-1), so the interval is looked for at the run's two ENDS -- the search
// result is the largest j with xp[j] <= x, non-decreasing in x, and x rises with the index on an
// arithmetic grid -- and where both ends give the same j every point of the run has it: what is left
// per point is x - xp[j], the knot test and slope*dx + f with f and slope arriving by scalar loads
// (8 vector instructions per point and band where the search per point took ~45).  Runs that hold
// a knot or a band edge take the search per point as before; the values are the same expressions on
// the same operands either way.
template <int G, int LV>
__device__ __forceinline__ bool add_band(const Band & b, const double * __restrict__ coarse,
                                         const double * __restrict__ slopes, long long row,
                                         long long level_points, int count, const double (&x)[G],
                                         const WaveRun & run, double (&total)[G][LV])
{
    const int last = b.size - 1;
    const double x_last = b.lower + (double)last*b.resolution;
    if (run.usable)
    {
        if (run.hi < b.lower || run.lo > x_last) return false;      // the whole run lies outside
        if (run.lo >= b.lower && run.hi <= x_last)
        {
            // (the run's last point has the interval of its first iff it lies below the next knot)
            double xj;
            const int j_lo = interval_of(b, last, run.lo, true, xj);
            const double next_knot = b.lower + (double)(j_lo + 1)*b.resolution;
            if (j_lo == last || run.hi < next_knot)
            {
                const int j = __builtin_amdgcn_readfirstlane(j_lo);
                const bool last_knot = j == last;
#pragma unroll
                for (int l = 0; l < LV; ++l)
                {
                    const long long at = row + (long long)(l < count ? l : 0)*level_points + j;
                    const double f = coarse[at];
                    const double slope = slopes[at];
#pragma unroll
                    for (int p = 0; p < G; ++p)
                    {
                        const bool on_knot = last_knot || xj == x[p];
                        double value = on_knot ? f : slope*(x[p] - xj) + f;
                        if (value != value && !on_knot && l < count)
                        {
                            const double xn = b.lower + (double)(j + 1)*b.resolution;
                            const double fn = coarse[at + 1];
                            value = slope*(x[p] - xn) + fn;
                            if (value != value && f == fn) value = f;
                        }
                        total[p][l] += value*100.;                  // utils.py:171-173
                    }
                }
                return true;
            }
        }
    }
    int j[G];
    double dx[G];
    bool inside[G], on_knot[G];
    bool any = false;
#pragma unroll
    for (int p = 0; p < G; ++p)
    {
        inside[p] = (x[p] >= b.lower) && (x[p] <= x_last);   // zero outside; false for NaN
        double xj;
        j[p] = interval_of(b, last, x[p], inside[p], xj);
        dx[p] = x[p] - xj;
        on_knot[p] = (j[p] == last || xj == x[p]);
        any = any || inside[p];
    }
    if (__ballot(any) == 0ull) return false;            // the whole wavefront lies outside
    double f[G][LV], slope[G][LV];
#pragma unroll
    for (int p = 0; p < G; ++p)
    {
#pragma unroll
        for (int l = 0; l < LV; ++l)
        {
            const long long at = row + (long long)(l < count ? l : 0)*level_points + j[p];
            f[p][l] = coarse[at];
            slope[p][l] = slopes[at];
        }
    }
#pragma unroll
    for (int p = 0; p < G; ++p)
    {
#pragma unroll
        for (int l = 0; l < LV; ++l)
        {
            double value = on_knot[p] ? f[p][l] : slope[p][l]*dx[p] + f[p][l];
            if (value != value && inside[p] && !on_knot[p] && l < count)
            {
                const long long at = row + (long long)l*level_points + j[p];
                const double xn = b.lower + (double)(j[p] + 1)*b.resolution;
                const double fn = coarse[at + 1];
                value = slope[p][l]*(x[p] - xn) + fn;
                if (value != value && f[p][l] == fn) value = f[p][l];
            }
            if (inside[p]) total[p][l] += value*100.;         // utils.py:171-173
        }
    }
    return true;
}

template <int kInterpPoints, int kInterpLevels>
__global__ __launch_bounds__(256) void continuum_interp_kernel(BandSet set,
                                                               const double * __restrict__ coarse,
                                                               const double * __restrict__ slopes,
                                                               GridForm form,
                                                               long long n, int n_levels,
                                                               double * __restrict__ out,
                                                               long long level_stride, int accumulate)
{
    // Points of a thread: pairs of neighbours (one 16-byte access where alignment allows), the
    // lanes' pairs side by side (a store instruction of the wavefront writes 1 KB of consecutive
    // points), a thread's pairs 128 points apart: a wavefront holds ONE run of 64 PT consecutive
    // points.  point_index(p) for p = 0..PT-1.
    static_assert(kInterpPoints == 1 || kInterpPoints % 2 == 0, "points come in pairs");
    const long long block_first = (long long)blockIdx.x*(256*kInterpPoints);
    auto point_index = [&](int p) -> long long {
        return wave_point<kInterpPoints>(block_first, p);
    };
    const int level0 = blockIdx.y*kInterpLevels;
    const int count = min(kInterpLevels, n_levels - level0);
    double x[kInterpPoints];
    load_wavenumbers<kInterpPoints>(form, n, point_index, x);
    const WaveRun run = wave_run<kInterpPoints>(form, n, block_first);
    double total[kInterpPoints][kInterpLevels], before[kInterpPoints][kInterpLevels];
#pragma unroll
    for (int l = 0; l < kInterpLevels; ++l)
    {
        double row[kInterpPoints];
#pragma unroll
        for (int p = 0; p < kInterpPoints; ++p) row[p] = 0.;
        // What is already there, fetched beside the wavenumbers rather than at the end.
        if (accumulate && l < count)
        {
            load_points<kInterpPoints>(out + (long long)(level0 + l)*level_stride, n,
                                       point_index, row, 0.);
        }
#pragma unroll
        for (int p = 0; p < kInterpPoints; ++p)
        {
            total[p][l] = 0.;
            before[p][l] = row[p];
        }
    }
    for (int k = 0; k < set.n_bands; ++k)
    {
        const Band & b = set.band[k];
        add_band<kInterpPoints, kInterpLevels>(
            b, coarse, slopes, (long long)level0*set.coarse_points + b.spectrum,
            set.coarse_points, count, x, run, total);
    }
#pragma unroll
    for (int l = 0; l < kInterpLevels; ++l)
    {
        if (l >= count) continue;
        double row[kInterpPoints];
#pragma unroll
        for (int p = 0; p < kInterpPoints; ++p)
        {
            row[p] = accumulate ? total[p][l] + before[p][l] : total[p][l];
        }
        store_points<kInterpPoints>(out + (long long)(level0 + l)*level_stride, n, point_index,
                                    row);
    }
}

// ---------------------------------------------------------------------------------------------
// Several continua in one pass (round 5).  Spectroscopy.compute_absorption adds, per gas, every
// continuum of the gas into mechanism slot 1 (spectroscopy.py:193-197; H2O has two, :58-61), and
// the "gas" / "total" formats then add slots and gases (:225-234): as separate launches every
// continuum after the first is a read-modify-write pass over the block (16 + 24 (owners - 1) bytes
// per point and level), each behind its own level copy and band kernel.  A group evaluates all of
// its continua per grid point and writes once: 8 bytes per point and level on an arithmetic grid
// (+ 8 when it adds into the block).  The sum is formed in the order the separate launches form
// it -- every continuum's bands from zero, then continuum after continuum onto what was there --
// so the values are the same bits.
struct GroupBand
{
    Band band;              // band.spectrum: offset inside the GROUP's per-level row of coarse spectra
    const double * table;   // the owning continuum's coefficient table
    int owner;              // index of the continuum in the group: its level scalars
    int first_of_owner;     // 1: the first band of its continuum
};

constexpr int kMaxGroupBands = 64;

// grid = (coarse points of the widest band / 256, bands of the group, levels);
// levels[owner*n_levels + level].
__global__ __launch_bounds__(256) void group_band_spectra_kernel(const GroupBand * __restrict__ bands,
                                                                 int level_points,
                                                                 const ContinuumLevel * __restrict__ levels,
                                                                 int n_levels,
                                                                 double * __restrict__ coarse,
                                                                 double * __restrict__ slopes)
{
    const GroupBand gb = bands[blockIdx.y];
    const Band & b = gb.band;
    const int j = blockIdx.x*256 + threadIdx.x;
    if (j >= b.size) return;
    const ContinuumLevel s = levels[(long long)gb.owner*n_levels + blockIdx.z];
    const long long at = (long long)blockIdx.z*level_points + b.spectrum + j;
    const double here = band_value(b, gb.table, s, j);
    coarse[at] = here;
    double slope = 0.;
    if (j + 1 < b.size)
    {
        const double xj = b.lower + (double)j*b.resolution;
        const double xn = b.lower + (double)(j + 1)*b.resolution;
        slope = (band_value(b, gb.table, s, j + 1) - here)/(xn - xj);
    }
    slopes[at] = slope;
}

// grid = (points / (256 PT), levels / LV).  out[level][i] (+)= sum over the group's continua.
// (Taking the bands two to four at a time -- their table values requested together -- was
// measured and only cost registers: profiles/r05_perf_continuum_group.txt.)
template <int PT, int LV>
__global__ __launch_bounds__(256) void group_interp_kernel(const GroupBand * __restrict__ bands,
                                                           int n_bands, int level_points,
                                                           const double * __restrict__ coarse,
                                                           const double * __restrict__ slopes,
                                                           GridForm form, long long n, int n_levels,
                                                           double * __restrict__ out,
                                                           long long level_stride, int accumulate)
{
    static_assert(PT == 1 || PT % 2 == 0, "points come in pairs");
    const long long block_first = (long long)blockIdx.x*(256*PT);
    auto point_index = [&](int p) -> long long {
        return wave_point<PT>(block_first, p);
    };
    const int level0 = blockIdx.y*LV;
    const int count = min(LV, n_levels - level0);
    double x[PT];
    load_wavenumbers<PT>(form, n, point_index, x);
    const WaveRun run = wave_run<PT>(form, n, block_first);
    double sum[PT][LV], total[PT][LV];
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
            sum[p][l] = row[p];
        }
    }
    // (`started`: something has been added to -- or was there before -- the running sum: the
    // first continuum of a group that writes REPLACES, like the first of the separate launches.)
    bool started = accumulate != 0;
    auto fold = [&]() {
#pragma unroll
        for (int p = 0; p < PT; ++p)
        {
#pragma unroll
            for (int l = 0; l < LV; ++l)
            {
                // (the order of the separate launches: what this continuum sums to, onto what
                // was there)
                sum[p][l] = started ? total[p][l] + sum[p][l] : total[p][l];
                total[p][l] = 0.;
            }
        }
        started = true;
    };
    for (int k = 0; k < n_bands; ++k)
    {
        const GroupBand * gb = bands + k;           // wave-uniform: scalar loads
        if (k > 0 && gb->first_of_owner) fold();
        add_band<PT, LV>(gb->band, coarse, slopes,
                         (long long)level0*level_points + gb->band.spectrum, level_points, count, x,
                         run, total);
    }
    fold();
#pragma unroll
    for (int l = 0; l < LV; ++l)
    {
        if (l >= count) continue;
        double row[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) row[p] = sum[p][l];
        store_points<PT>(out + (long long)(level0 + l)*level_stride, n, point_index, row);
    }
}

}  // namespace lbl
