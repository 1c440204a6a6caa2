// Device-side profile evaluation.
//
// Reference statement: pyLBL/c_lib/voigt.c:4-191 -- Wells' HUMLIK form of the Voigt
// function K(x,y) = Re w(x+iy): Lorentz far wing (region 0), Humlicek w4 rational
// regions 1-3, CPF12 regions I/II, chosen per point by an if / else-if chain on |x|.
//
// Entry points:
//   lorentz_*()      the far wing.  voigt.c:82 gives  K = (y/sqrt(pi))/(x^2+y^2); with the
//                    amplitude of voigt.c:188 and x = (v-nu')*repwid, y = repwid*gamma this
//                    is S*gamma/pi / ((v-nu')^2 + gamma^2): no repwid, one reciprocal.
//                    The same expression is the y >= 70.55 branch (voigt.c:17-27).
//   inner_pass()     the points nearer to the centre than xlim1 (w4 regions 2-3, CPF12) of one
//                    line, in two stages of this file's own (see below),
//   wells_profile()  the whole chain for one point.  Region SELECTION is the reference's,
//                    comparison for comparison (this file is compiled with
//                    -ffp-contract=off); values use reciprocal + Newton instead of division.
#pragma once

#include <hip/hip_runtime.h>

namespace lbl {

// 1/t for t well inside the normal range: v_rcp_f64 (about 25 good bits on gfx950) plus
// one Newton step; measured max relative error 1.4e-15 (scripts/ubench/ubench_fp64.hip).
__device__ __forceinline__ double rcp_newton(double t)
{
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}

// One far-wing line at one point, added to `sum`.
__device__ __forceinline__ double lorentz_one(double v, double centre, double g2, double bl,
                                              double sum)
{
    const double d = v - centre;
    return __builtin_fma(bl, rcp_newton(__builtin_fma(d, d, g2)), sum);
}

// Four far-wing lines at one point with a single reciprocal, added to `sum`:
//   sum b_i/t_i = N/(t1 t2 t3 t4).  Products stay far from over/underflow because
//   t = d^2 + gamma^2 lies in [~1e-12, ~1e3] for every window the reference allows.
__device__ __forceinline__ double lorentz_four(double v,
                                               double c1, double g1, double b1,
                                               double c2, double g2, double b2,
                                               double c3, double g3, double b3,
                                               double c4, double g4, double b4, double sum)
{
    const double d1 = v - c1, d2 = v - c2, d3 = v - c3, d4 = v - c4;
    const double t1 = __builtin_fma(d1, d1, g1);
    const double t2 = __builtin_fma(d2, d2, g2);
    const double t3 = __builtin_fma(d3, d3, g3);
    const double t4 = __builtin_fma(d4, d4, g4);
    const double n12 = __builtin_fma(b1, t2, b2*t1);
    const double n34 = __builtin_fma(b3, t4, b4*t3);
    const double t12 = t1*t2, t34 = t3*t4;
    const double num = __builtin_fma(n12, t34, n34*t12);
    return __builtin_fma(num, rcp_newton(t12*t34), sum);
}

// Eight far-wing lines at one point with a single reciprocal (two levels more of the same
// pairing): 40 FMA-class operations + 1 v_rcp_f64 per 8 evaluations.
struct WingTerm
{
    double centre, g2, bl;
};

__device__ __forceinline__ void wing_pair(double v, const WingTerm & p, const WingTerm & q,
                                          double & num, double & den)
{
    const double d1 = v - p.centre, d2 = v - q.centre;
    const double t1 = __builtin_fma(d1, d1, p.g2);
    const double t2 = __builtin_fma(d2, d2, q.g2);
    num = __builtin_fma(p.bl, t2, q.bl*t1);
    den = t1*t2;
}

__device__ __forceinline__ double lorentz_eight(double v, const WingTerm (&l)[8], double sum)
{
    double n12, t12, n34, t34, n56, t56, n78, t78;
    wing_pair(v, l[0], l[1], n12, t12);
    wing_pair(v, l[2], l[3], n34, t34);
    wing_pair(v, l[4], l[5], n56, t56);
    wing_pair(v, l[6], l[7], n78, t78);
    const double na = __builtin_fma(n12, t34, n34*t12), ta = t12*t34;
    const double nb = __builtin_fma(n56, t78, n78*t56), tb = t56*t78;
    const double num = __builtin_fma(na, tb, nb*ta);
    return __builtin_fma(num, rcp_newton(ta*tb), sum);
}

// ---------------------------------------------------------------------------------------------
// The inner regions: points nearer to the line centre than xlim1 (voigt.c:98-186).
//
// The reference evaluates them point by point inside its per-line loop and caches what depends
// on y alone behind three flags.  Here the work is cut in two stages of its own:
//
//   stage 1, per line   InnerLimits (where each region begins) and, where a pass over the
//                       line's inner points meets a rational region, that region's coefficient
//                       set -- polynomials in y whose coefficients are Wells' published
//                       constants, kept as tables and evaluated by one Horner routine;
//   stage 2, per point  region_two_value / region_three_value (one shared reciprocal each) and
//                       the 12-node CPF sum, rearranged so that a node pair needs ONE reciprocal
//                       in either CPF region (the reference divides two resp. four times per
//                       node; here 1/(a-)(a+) resp. 1/(a- b- a+ b+) is formed once and multiplied
//                       back).
//
// Region SELECTION is the reference's comparison chain on the same |x| and the same limits:
//   |x| >= xlim2 -> w4 region 2;  |x| < xlim3 -> w4 region 3;  else CPF12, region I when
//   |x| <= xlim4, region II beyond (voigt.c:98, :116, :148, :168).
// ---------------------------------------------------------------------------------------------

template <int N>
__device__ __forceinline__ double horner(const double (&c)[N], double t)
{
    double value = c[N - 1];
#pragma unroll
    for (int k = N - 2; k >= 0; --k)
    {
        value = c[k] + t*value;
    }
    return value;
}

struct InnerLimits
{
    double two, three, four;    // xlim2, xlim3, xlim4 (voigt.c:44-53)
};

__device__ __forceinline__ InnerLimits inner_limits(double y)
{
    InnerLimits lim;
    // For y <= 1e-6 the reference lifts xlim2 to xlim0 (> any inner |x|): region 2 never applies.
    lim.two = (y <= 0.000001) ? 1.e300 : 6.8 - y;
    lim.three = 2.4*y;
    lim.four = 18.1*y + 1.65;
    return lim;
}

// w4 region 2:  K = y/sqrt(pi) * E(x^2)/H(x^2),  H = x^8 + h6 x^6 + h4 x^4 + h2 x^2 + h0,
// E = x^6 + e4 x^4 + e2 x^2 + e0; every coefficient a polynomial in y^2 (voigt.c:101-114).
struct RegionTwo
{
    double h0, h2, h4, h6, e0, e2, e4;
};

__device__ __forceinline__ RegionTwo region_two_model(double y)
{
    const double kTwoH0[5] = {0.5625, 4.5, 10.5, 6.0, 1.0};
    const double kTwoH2[4] = {-4.5, 9.0, 6.0, 4.0};
    const double kTwoH4[3] = {10.5, -6.0, 6.0};
    const double kTwoH6[2] = {-6.0, 4.0};
    const double kTwoE0[4] = {1.875, 8.25, 5.5, 1.0};
    const double kTwoE2[3] = {5.25, 1.0, 3.0};
    const double yq = y*y;
    RegionTwo m;
    m.h0 = horner(kTwoH0, yq);
    m.h2 = horner(kTwoH2, yq);
    m.h4 = horner(kTwoH4, yq);
    m.h6 = horner(kTwoH6, yq);
    m.e0 = horner(kTwoE0, yq);
    m.e2 = horner(kTwoE2, yq);
    m.e4 = 0.75*m.h6;
    return m;
}

__device__ __forceinline__ double region_two_value(const RegionTwo & m, double y, double xq)
{
    const double rsqrpi = 0.56418958354775628695;   // 1/sqrt(pi)
    const double den = m.h0 + xq*(m.h2 + xq*(m.h4 + xq*(m.h6 + xq)));
    const double num = m.e0 + xq*(m.e2 + xq*(m.e4 + xq));
    return (rsqrpi*rcp_newton(den))*y*num;
}

// w4 region 3:  K = 1.7724538 * P(x^2)/Z(x^2) with Z of degree 5 (monic) and P of degree 4 in
// x^2, coefficients polynomials in y (voigt.c:119-146; the 8-digit sqrt(pi) is the reference's).
struct RegionThree
{
    double z0, z2, z4, z6, z8, p0, p2, p4, p6, p8;
};

__device__ __forceinline__ RegionThree region_three_model(double y)
{
    const double kThreeZ0[11] = {272.1014, 1280.829, 2802.870, 3764.966, 3447.629, 2256.981,
                                 1074.409, 369.1989, 88.26741, 13.39880, 1.0};
    const double kThreeZ2[9] = {211.678, 902.3066, 1758.336, 2037.310, 1549.675, 793.4273,
                                266.2987, 53.59518, 5.0};
    const double kThreeZ4[7] = {78.86585, 308.1852, 497.3014, 479.2576, 269.2916, 80.39278,
                                10.0};
    const double kThreeZ6[5] = {22.03523, 55.02933, 92.75679, 53.59518, 10.0};
    const double kThreeZ8[3] = {1.496460, 13.39880, 5.0};
    const double kThreeP0[10] = {153.5168, 549.3954, 919.4955, 946.8970, 662.8097, 328.2151,
                                 115.3772, 27.93941, 4.264678, 0.3183291};
    const double kThreeP2[8] = {-34.16955, -1.322256, 124.5975, 189.7730, 139.4665, 56.81652,
                                12.79458, 1.2733163};
    const double kThreeP4[6] = {2.584042, 10.46332, 24.01655, 29.81482, 12.79568, 1.9099744};
    const double kThreeP6[4] = {-0.07272979, 0.9377051, 4.266322, 1.273316};
    const double kThreeP8[2] = {0.0005480304, 0.3183291};
    RegionThree m;
    m.z0 = horner(kThreeZ0, y);
    m.z2 = horner(kThreeZ2, y);
    m.z4 = horner(kThreeZ4, y);
    m.z6 = horner(kThreeZ6, y);
    m.z8 = horner(kThreeZ8, y);
    m.p0 = horner(kThreeP0, y);
    m.p2 = horner(kThreeP2, y);
    m.p4 = horner(kThreeP4, y);
    m.p6 = horner(kThreeP6, y);
    m.p8 = horner(kThreeP8, y);
    return m;
}

__device__ __forceinline__ double region_three_value(const RegionThree & m, double xq)
{
    const double den = m.z0 + xq*(m.z2 + xq*(m.z4 + xq*(m.z6 + xq*(m.z8 + xq))));
    const double num = m.p0 + xq*(m.p2 + xq*(m.p4 + xq*(m.p6 + xq*m.p8)));
    return (1.7724538*rcp_newton(den))*num;
}

// Humlicek's 12-node complex probability function, nodes +-t_j with weights (c_j, s_j)
// (voigt.c:55-60).  With Y = y + 1.5, a = (x -+ t)^2 + Y^2 and b = (x -+ t)^2 + 1.5^2:
//   region I   sum_j  c_j Y (1/a- + 1/a+)  -  s_j ((x-t)/a- - (x+t)/a+)          (voigt.c:170-175)
//   region II  y sum_j [c_j((x-t)^2 - 1.5 Y) + s_j(y+3)(x-t)]/(a- b-)
//                    + [c_j((x+t)^2 - 1.5 Y) - s_j(y+3)(x+t)]/(a+ b+)  + exp(-x^2) (voigt.c:178-186)
// (the reference's mq*mf - y0*ym is ((x-t)^2 - y0 Y)/a-, and so on).  `outer` selects region II
// per lane; a pass in which no lane is in region II skips its arithmetic and the exponential.
__device__ __forceinline__ double cpf12_value(double xi, double y, bool outer)
{
    const double kCpfWeightC[6] = {1.0117281, -0.75197147, 0.012557727,
                                   0.010022008, -0.00024206814, 0.00000050084806};
    const double kCpfWeightS[6] = {1.393237, 0.23115241, -0.15535147,
                                   0.0062183662, 0.000091908299, -0.00000062752596};
    const double kCpfNode[6] = {0.31424038, 0.94778839, 1.5976826,
                                2.2795071, 3.0206370, 3.8897249};
    const double y0 = 1.5;
    const double big_y = y + y0;
    const double big_y2 = big_y*big_y;
    double sum = 0.;
    if (!outer)
    {
#pragma unroll
        for (int j = 0; j < 6; ++j)
        {
            const double t = kCpfNode[j];
            const double dm = xi - t, dp = xi + t;
            const double am = __builtin_fma(dm, dm, big_y2);
            const double ap = __builtin_fma(dp, dp, big_y2);
            const double r = rcp_newton(am*ap);
            const double rm = r*ap, rp = r*am;          // 1/a-, 1/a+
            sum += kCpfWeightC[j]*(big_y*(rm + rp)) - kCpfWeightS[j]*(dm*rm - dp*rp);
        }
        return sum;
    }
    const double y0q = y0*y0;
    const double shift = y0*big_y;
    const double yf = y + (y0 + y0);
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
        const double t = kCpfNode[j];
        const double dm = xi - t, dp = xi + t;
        const double qm = dm*dm, qp = dp*dp;
        const double pm = (qm + big_y2)*(qm + y0q);     // a- b-
        const double pp = (qp + big_y2)*(qp + y0q);     // a+ b+
        const double r = rcp_newton(pm*pp);
        const double syf = kCpfWeightS[j]*yf;
        const double nm = __builtin_fma(kCpfWeightC[j], qm - shift, syf*dm);
        const double np = __builtin_fma(kCpfWeightC[j], qp - shift, -(syf*dp));
        sum += nm*(r*pp) + np*(r*pm);
    }
    return y*sum + exp(-(xi*xi));
}

// The inner points are evaluated one CLASS at a time (accumulate.h packs the points of many
// lines by class before it calls these): a wavefront that runs the reference's if-chain on mixed
// points executes every branch for every point, and the branches differ tenfold in cost.
//   class A  |x| >= xlim2               w4 region 2
//   class B  |x| <= xlim4 (and < xlim2)   CPF12 region I, or w4 region 3 where |x| < xlim3
//   class C  the rest                    CPF12 region II
// Each lane is one (line, point) pair, so stage 1 (what depends on y) runs per lane here; it is
// the smaller part of every class.  Real functions on purpose, each small enough for the
// registers a callee may clobber (v0-v39): inlined, their temporaries would be the caller's,
// whose far-wing loop lives or dies by its occupancy.
constexpr int kInnerClasses = 3;

__device__ __forceinline__ int inner_class(double abx, const InnerLimits & lim)
{
    return abx >= lim.two ? 0 : (abx <= lim.four ? 1 : 2);
}

__device__ __noinline__ double inner_class_a(double xi, double y)
{
    const double abx = fabs(xi);
    return region_two_value(region_two_model(y), y, abx*abx);
}

__device__ __noinline__ double inner_class_b(double xi, double y)
{
    const double abx = fabs(xi);
    const bool in_three = abx < 2.4*y;              // xlim3, tested after xlim2 (voigt.c:116)
    double value = 0.;
    if (__any(in_three))
    {
        const double r3 = region_three_value(region_three_model(y), abx*abx);
        value = in_three ? r3 : value;
    }
    if (!in_three)
    {
        value = cpf12_value(xi, y, false);
    }
    return value;
}

__device__ __noinline__ double inner_class_c(double xi, double y)
{
    return cpf12_value(xi, y, true);
}

// One inner point of one line, for callers whose lanes hold different lines (the pedestal's
// window-edge slots): every lane runs its own stage 1.
__device__ __noinline__ double inner_point(double xi, double y)
{
    const InnerLimits lim = inner_limits(y);
    const double abx = fabs(xi);
    const double xq = abx*abx;
    if (abx >= lim.two)
    {
        return region_two_value(region_two_model(y), y, xq);
    }
    if (abx < lim.three)
    {
        return region_three_value(region_three_model(y), xq);
    }
    return cpf12_value(xi, y, abx > lim.four);
}

// K(x,y) for y < 70.55 with the reference's full region chain (voigt.c:74-187): used where
// a single point is evaluated at a time (the pedestal slots).
__device__ __forceinline__ double wells_profile(double xi, double y)
{
    const double rsqrpi = 0.56418958354775628695;
    const double yq = y*y;
    const double xlim0 = sqrt(15100. + y*(40. - y*3.6));
    double xlim1 = (y >= 8.425) ? 0. : sqrt(164. - y*(4.3 + y*1.8));
    if (y <= 0.000001) xlim1 = xlim0;
    const double abx = fabs(xi);
    const double xq = abx*abx;
    if (abx >= xlim0)
    {
        return y*rsqrpi*rcp_newton(xq + yq);
    }
    if (abx >= xlim1)
    {
        const double a0 = yq + 0.5;
        const double d0 = a0*a0;
        const double d2 = yq + yq - 1.;
        return rsqrpi*rcp_newton(d0 + xq*(d2 + xq))*y*(a0 + xq);
    }
    return inner_point(xi, y);
}

}  // namespace lbl
