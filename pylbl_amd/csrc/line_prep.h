// Per-line physics ("K1"): turns one HITRAN transition and one atmospheric level into
// the handful of scalars the accumulate kernel needs.
//
// Reference statement: pyLBL/c_lib/spectra.c:12-62 (pressure shift, Lorentz and Doppler
// widths, line-strength temperature scaling, window indices) and the first lines of
// pyLBL/c_lib/voigt.c:7-15,33-34 (repwid, y, far-wing limit).  Operation order of every
// expression follows the reference; the translation unit is compiled with
// -ffp-contract=off so no multiply-add is fused here.
//
// What is hoisted out of the per-line work (it depends on the level and isotopologue
// only) and done once on the host: p, p*x, 296/T, sqrt(2 ln2 R T / M_iso) and the TIPS
// ratio Q_iso(296)/Q_iso(T) (spectral_database.c:97-104) -- see LevelScalars.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace lbl {

constexpr int kMassSlots = 32;          // absorption.c:62 (double mass[32])
constexpr double kPi = 3.14159265358979323846;

// Scalars of one atmospheric level, prepared on the host.
struct LevelScalars
{
    double temperature;     // T [K]
    double p_atm;           // P*9.86923e-6                      spectra.c:17
    double p_partial;       // p*x                               spectra.c:18
    double tfact;           // 296/T                             spectra.c:19
    double t_minus_ref;     // T - 296                           spectra.c:33
    double t_times_ref;     // T*296                             spectra.c:33
    double shift_max;       // bound on |p*delta_air| over the table (+ margin)
    double core_reach;      // bound on (far-wing limit / repwid) / nu over the table
    double density;         // P x /(kb T)                       spectroscopy.py:18-29
    double inner_possible;  // 0: y >= 8.425 for every line of this call at this level, i.e. no line
                            // has points in the inner regions (voigt.c:35-43); else 1
    double doppler[kMassSlots];   // sqrt(2 ln2 * 8314.472 * T / mass[iso])   spectra.c:29
    double q_ratio[kMassSlots];   // Q(296)/Q(T) per isotopologue             spectra.c:41-42
};

struct GridSpec
{
    int v0, vn, n_per_v, cut_off;
    int n;                  // (vn - v0)*n_per_v                 absorption.c:34
    double dv;              // 1./n_per_v                        absorption.c:33
};

// What the accumulate kernel reads per line and level.  Two 32-byte halves so that the
// far-wing loop (the >99 % case) touches only the first.
struct alignas(32) LineWing
{
    double centre;          // nu + p*delta_air                  spectra.c:22
    double g2;              // gamma^2
    double bl;              // S*gamma/pi: Lorentz amplitude (voigt.c:24 / :82 rewritten in cm-1)
    int first, last;        // window [first, last] inclusive    spectra.c:48-62; empty if first > last
};

struct alignas(16) LineCore
{
    double repwid;          // sqrt(ln2)/alpha                   voigt.c:13
    double y;               // repwid*gamma                      voigt.c:14
    double amp;             // S/sqrt(pi)*repwid                 voigt.c:188
    double xlim0;           // far-wing limit                    voigt.c:34
    double xlim1;           // w4 region-1 limit                 voigt.c:35-43, :48-53
    int core_first, core_last;  // grid indices that may fall inside |x| < xlim0 (conservative)
    // Grid indices that CERTAINLY lie inside the window and inside |x| < xlim0 (one index and
    // 1e-9 of the reach inside the limit), and the indices that may lie inside |x| < xlim1 (the
    // range of inner_index_range(); empty when the line has no inner regions).  A 64-point row
    // inside the first range and clear of the second is w4 region 1 on every lane
    // (voigt.c:95-96): accumulate.h evaluates it without the per-lane region chain.
    int mid_first, mid_last;
    int hole_first, hole_last;
};
static_assert(sizeof(LineCore) == 64, "LineCore is read by scalar loads: 64 bytes");

// An empty index range that intersects no row, tile or grid (including index 0).
constexpr int kEmptyFirst = 0x3fffffff;
constexpr int kEmptyLast = -0x3fffffff;

__host__ __device__ inline void mark_empty(LineWing & w, LineCore & c)
{
    w.centre = 0.; w.g2 = 1.; w.bl = 0.; w.first = kEmptyFirst; w.last = kEmptyLast;
    c.repwid = 1.; c.y = 100.; c.amp = 0.; c.xlim0 = 0.; c.xlim1 = 0.;
    c.core_first = kEmptyFirst; c.core_last = kEmptyLast;
    c.mid_first = kEmptyFirst; c.mid_last = kEmptyLast;
    c.hole_first = kEmptyFirst; c.hole_last = kEmptyLast;
}

// Grid indices that may fall inside |x| < xlim1, the inner regions of a line (w4 regions 2-3,
// CPF12; voigt.c:98-186): conservative, with the margins of core_first / core_last above.
__host__ __device__ inline void inner_index_range(double centre, double repwid, double xlim1,
                                                  int v0, int n_per_v, int & first, int & last)
{
    const double near = (xlim1/repwid)*(1. + 1.e-9);
    double lo = floor((centre - near - (double)v0)*n_per_v) - 1.;
    double hi = floor((centre + near - (double)v0)*n_per_v) + 2.;
    if (lo < -1.e9) lo = -1.e9;
    if (hi > 1.e9) hi = 1.e9;
    if (hi < -1.e9) hi = -1.e9;
    if (lo > 1.e9) lo = 1.e9;
    first = (int)lo;
    last = (int)hi;
}

// status: 1 evaluated, 0 window right of the grid / empty, -1 not accepted by the range rule.
// derived (optional, 8 doubles): centre, alpha, gamma, strength, first, last, status, 0.
__host__ __device__ inline int prepare_line(const LevelScalars & lv, const GridSpec & g,
                                            double nu, double sw, double gamma_air,
                                            double gamma_self, double n_air, double elower,
                                            double delta_air, int iso_slot, bool accepted,
                                            LineWing & w, LineCore & c, double * derived)
{
    const double vlight = 2.99792458e8;
    const double c2 = 1.4387752;
    const double rsqrpi = 1./sqrt(kPi);
    const double sqrln2 = sqrt(log(2.));
    mark_empty(w, c);
    if (derived != nullptr)
    {
        for (int i = 0; i < 8; ++i) derived[i] = 0.;
        derived[6] = -1.;
    }
    if (!accepted)
    {
        return -1;
    }
    // spectra.c:22-45
    const double centre = nu + lv.p_atm*delta_air;
    const double gamma = (gamma_air*(lv.p_atm - lv.p_partial) + gamma_self*lv.p_partial)*
                         pow(lv.tfact, n_air);
    const double alpha = (nu/vlight)*lv.doppler[iso_slot];
    const double sb = exp(elower*c2*lv.t_minus_ref/lv.t_times_ref);
    const double gg = exp((-c2*nu)/lv.temperature);
    const double gref = exp((-c2*nu)/296.);
    const double se = (1. - gg)/(1. - gref);
    const double strength = sw*sb*se*lv.q_ratio[iso_slot]*0.01*0.01;

    // spectra.c:48-62 (v[0] == v0 exactly: absorption.c:39 with i = 0)
    const double fl = floor(centre);
    int first = (int)((fl - g.cut_off - (double)g.v0)*g.n_per_v);
    int last = 0;
    int status = 1;
    if (first >= g.n)
    {
        status = 0;
    }
    else
    {
        if (first < 0) first = 0;
        last = (int)((fl + g.cut_off + 1 - (double)g.v0)*g.n_per_v);
        if (last >= g.n) last = g.n - 1;
    }
    if (derived != nullptr)
    {
        derived[0] = centre; derived[1] = alpha; derived[2] = gamma; derived[3] = strength;
        derived[4] = first; derived[5] = last; derived[6] = status;
    }
    if (status == 0 || last < first)
    {
        // Nothing to add (a window wholly left of the grid gives last < 0).
        return status;
    }
    // voigt.c:13-15
    const double repwid = sqrln2/alpha;
    const double y = repwid*gamma;
    w.centre = centre;
    w.g2 = gamma*gamma;
    w.bl = strength*gamma/kPi;
    w.first = first;
    w.last = last;
    c.repwid = repwid;
    c.y = y;
    c.amp = strength*rsqrpi*repwid;
    if (y < 70.55)
    {
        // voigt.c:34: beyond xlim0 Doppler half-widths the profile is the Lorentz wing.
        const double xlim0 = sqrt(15100. + y*(40. - y*3.6));
        double xlim1 = (y >= 8.425) ? 0. : sqrt(164. - y*(4.3 + y*1.8));
        if (y <= 0.000001) xlim1 = xlim0;               // voigt.c:48-53
        c.xlim0 = xlim0;
        c.xlim1 = xlim1;
        const double reach = (xlim0/repwid)*(1. + 1.e-9);
        double lo = floor((centre - reach - (double)g.v0)*g.n_per_v) - 1.;
        double hi = floor((centre + reach - (double)g.v0)*g.n_per_v) + 2.;
        if (lo < -1.e9) lo = -1.e9;
        if (hi > 1.e9) hi = 1.e9;
        if (hi < -1.e9) hi = -1.e9;
        if (lo > 1.e9) lo = 1.e9;
        c.core_first = (int)lo;
        c.core_last = (int)hi;
        const double sure = (xlim0/repwid)*(1. - 1.e-9);
        double mid_lo = ceil((centre - sure - (double)g.v0)*g.n_per_v) + 1.;
        double mid_hi = floor((centre + sure - (double)g.v0)*g.n_per_v) - 1.;
        if (mid_lo < (double)first) mid_lo = (double)first;
        if (mid_hi > (double)last) mid_hi = (double)last;
        if (mid_hi >= mid_lo)
        {
            c.mid_first = (int)mid_lo;
            c.mid_last = (int)mid_hi;
        }
        if (xlim1 > 0.)
        {
            inner_index_range(centre, repwid, xlim1, g.v0, g.n_per_v, c.hole_first, c.hole_last);
        }
    }
    return status;
}

}  // namespace lbl
