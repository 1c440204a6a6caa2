// Device-side profile evaluation.
//
// Reference statement: pyLBL/c_lib/voigt.c:4-191 -- Wells' HUMLIK form of the Voigt
// function K(x,y) = Re w(x+iy): Lorentz far wing (region 0), Humlicek w4 rational
// regions 1-3, CPF12 regions I/II, chosen per point by an if / else-if chain on |x|.
//
// Two entry points:
//   lorentz_*()      the far wing.  voigt.c:82 gives  K = (y/sqrt(pi))/(x^2+y^2); with the
//                    amplitude of voigt.c:188 and x = (v-nu')*repwid, y = repwid*gamma this
//                    is S*gamma/pi / ((v-nu')^2 + gamma^2): no repwid, one reciprocal.
//                    The same expression is the y >= 70.55 branch (voigt.c:17-27).
//   wells_inner()    the points nearer to the centre than xlim1 (w4 regions 2-3, CPF12),
//   wells_profile()  the whole chain for one point.  Region SELECTION is the reference's,
//                    comparison for comparison (this file is compiled with
//                    -ffp-contract=off); values use reciprocal + Newton instead of division.
#pragma once

#include <hip/hip_runtime.h>

namespace lbl {

// 1/t for t well inside the normal range: v_rcp_f64 (about 25 good bits on gfx950) plus
// one Newton step; measured max relative error 1.4e-15 (scripts/ubench_fp64.hip).
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

// K(x,y) for the points nearer to the line centre than xlim1 (w4 regions 2 and 3, CPF12
// regions I and II), selected exactly as voigt.c:98-186 selects them (same comparisons on
// the same abx and limits).  Values: the reference's expressions with its divisions replaced
// by reciprocal + Newton step (1.4e-15), far inside the 1e-6 parity bar.
// Not inlined: reached by a few rows per line only, and large.
__device__ __noinline__ double wells_inner(double xi, double y)
{
    const double rsqrpi = 0.56418958354775628695;   // 1/sqrt(pi)
    const double y0 = 1.5;
    const double y0py0 = y0 + y0;
    const double y0q = y0*y0;
    const double yq = y*y;
    // voigt.c:44-53: for y <= 1e-6 the w4 regions 1 and 2 are switched off.
    const double xlim2 = (y <= 0.000001) ? 1.e300 : 6.8 - y;
    const double xlim3 = 2.4*y;
    const double xlim4 = 18.1*y + 1.65;
    const double abx = fabs(xi);
    const double xq = abx*abx;
    double buf;
    if (abx >= xlim2)
    {
        const double h0 = 0.5625 + yq*(4.5 + yq*(10.5 + yq*(6.0 + yq)));
        const double h2 = -4.5 + yq*(9.0 + yq*(6.0 + yq*4.0));
        const double h4 = 10.5 - yq*(6.0 - yq*6.0);
        const double h6 = -6.0 + yq*4.0;
        const double e0 = 1.875 + yq*(8.25 + yq*(5.5 + yq));
        const double e2 = 5.25 + yq*(1.0 + yq*3.0);
        const double e4 = 0.75*h6;
        const double d = rsqrpi*rcp_newton(h0 + xq*(h2 + xq*(h4 + xq*(h6 + xq))));
        buf = d*y*(e0 + xq*(e2 + xq*(e4 + xq)));
    }
    else if (abx < xlim3)
    {
        const double z0 = 272.1014 + y*(1280.829 + y*(2802.870 + y*(3764.966
                          + y*(3447.629 + y*(2256.981 + y*(1074.409 + y*(369.1989
                          + y*(88.26741 + y*(13.39880 + y)))))))));
        const double z2 = 211.678 + y*(902.3066 + y*(1758.336 + y*(2037.310
                          + y*(1549.675 + y*(793.4273 + y*(266.2987
                          + y*(53.59518 + y*5.0)))))));
        const double z4 = 78.86585 + y*(308.1852 + y*(497.3014 + y*(479.2576
                          + y*(269.2916 + y*(80.39278 + y*10.0)))));
        const double z6 = 22.03523 + y*(55.02933 + y*(92.75679 + y*(53.59518
                          + y*10.0)));
        const double z8 = 1.496460 + y*(13.39880 + y*5.0);
        const double p0 = 153.5168 + y*(549.3954 + y*(919.4955 + y*(946.8970
                          + y*(662.8097 + y*(328.2151 + y*(115.3772 + y*(27.93941
                          + y*(4.264678 + y*0.3183291))))))));
        const double p2 = -34.16955 + y*(-1.322256 + y*(124.5975 + y*(189.7730
                          + y*(139.4665 + y*(56.81652 + y*(12.79458
                          + y*1.2733163))))));
        const double p4 = 2.584042 + y*(10.46332 + y*(24.01655 + y*(29.81482
                          + y*(12.79568 + y*1.9099744))));
        const double p6 = -0.07272979 + y*(0.9377051 + y*(4.266322 + y*1.273316));
        const double p8 = 0.0005480304 + y*0.3183291;
        const double d = 1.7724538*rcp_newton(z0 + xq*(z2 + xq*(z4 + xq*(z6 + xq*(z8 + xq)))));
        buf = d*(p0 + xq*(p2 + xq*(p4 + xq*(p6 + xq*p8))));
    }
    else
    {
        const double cc[6] = {1.0117281, -0.75197147, 0.012557727,
                              0.010022008, -0.00024206814, 0.00000050084806};
        const double ss[6] = {1.393237, 0.23115241, -0.15535147,
                              0.0062183662, 0.000091908299, -0.00000062752596};
        const double tt[6] = {0.31424038, 0.94778839, 1.5976826,
                              2.2795071, 3.0206370, 3.8897249};
        const double ypy0 = y + y0;
        const double ypy0q = ypy0*ypy0;
        const bool inner = abx <= xlim4;
        const double yf = y + y0py0;
        buf = 0.;
#pragma unroll
        for (int j = 0; j < 6; ++j)
        {
            double d = xi - tt[j];
            const double mq = d*d;
            const double mf = rcp_newton(mq + ypy0q);
            const double xm = mf*d;
            const double ym = mf*ypy0;
            d = xi + tt[j];
            const double pq = d*d;
            const double pf = rcp_newton(pq + ypy0q);
            const double xp = pf*d;
            const double yp = pf*ypy0;
            if (inner)
            {
                buf += cc[j]*(ym + yp) - ss[j]*(xm - xp);
            }
            else
            {
                buf += (cc[j]*(mq*mf - y0*ym) + ss[j]*yf*xm)*rcp_newton(mq + y0q)
                       + (cc[j]*(pq*pf - y0*yp) - ss[j]*yf*xp)*rcp_newton(pq + y0q);
            }
        }
        if (!inner)
        {
            buf = y*buf + exp(-xq);
        }
    }
    return buf;
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
    return wells_inner(xi, y);
}

}  // namespace lbl
