// Micro-benchmark for a candidate far-wing loop body: instead of eight lines' Lorentzians merged
// pairwise at every point (lorentz_eight: 40 FMA-class + 1 v_rcp_f64 per 8 lines and point), the
// four lines of a "quad" are pre-multiplied into one rational function N(u)/D(u) in a local
// coordinate u (deg 6 / monic deg 8, 15 coefficients per quad), two quads share a reciprocal:
// 33 FMA-class + 1 v_rcp_f64 per 8 lines and point.  Both loops read wave-uniform records
// through the scalar cache like the product kernel.  Not part of the product.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/ubench/ubench_poly.hip -o /tmp/ubench_poly
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct alignas(32) Line { double centre, g2, bl; int first, last; };
struct alignas(128) Quad { double c[8]; double n[7]; double pad; };

__device__ __forceinline__ double rcp_newton(double t)
{
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}

template <int P>
__global__ __launch_bounds__(256) void direct_loop(const Line * __restrict__ lines, int n_lines,
                                                   double v0, double dv, double * __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long base = (long)blockIdx.x*64*P;
    double v[P], acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { v[p] = v0 + (double)(base + p*64 + lane)*dv; acc[p] = 0.; }
    const int share = n_lines/4;
    const Line * __restrict__ l = lines + wave*share;
    for (int j = 0; j + 8 <= share; j += 8)
    {
        Line r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = l[j + i];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            double n[4], t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                const double d1 = v[p] - r[2*i].centre, d2 = v[p] - r[2*i + 1].centre;
                const double t1 = __builtin_fma(d1, d1, r[2*i].g2);
                const double t2 = __builtin_fma(d2, d2, r[2*i + 1].g2);
                n[i] = __builtin_fma(r[2*i].bl, t2, r[2*i + 1].bl*t1);
                t[i] = t1*t2;
            }
            const double na = __builtin_fma(n[0], t[1], n[1]*t[0]), ta = t[0]*t[1];
            const double nb = __builtin_fma(n[2], t[3], n[3]*t[2]), tb = t[2]*t[3];
            const double num = __builtin_fma(na, tb, nb*ta);
            acc[p] = __builtin_fma(num, rcp_newton(ta*tb), acc[p]);
        }
    }
    __shared__ double partial[4][P][64];
#pragma unroll
    for (int p = 0; p < P; ++p) partial[wave][p][lane] = acc[p];
    __syncthreads();
    for (int p = wave; p < P; p += 4)
        out[base + p*64 + lane] = (partial[0][p][lane] + partial[1][p][lane]) +
                                  (partial[2][p][lane] + partial[3][p][lane]);
}

template <int P, int KEEP_U>
__global__ __launch_bounds__(256) void poly_loop(const Quad * __restrict__ quads, int n_quads,
                                                 double v0, double dv, double u0,
                                                 double * __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long base = (long)blockIdx.x*64*P;
    double v[P], acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        v[p] = v0 + (double)(base + p*64 + lane)*dv;
        if (KEEP_U) v[p] -= u0 + (double)base*dv;
        acc[p] = 0.;
    }
    const double centre = u0 + (double)base*dv;
    const int share = n_quads/4;
    const Quad * __restrict__ q = quads + wave*share;
    for (int j = 0; j + 2 <= share; j += 2)
    {
        const Quad a = q[j], b = q[j + 1];
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            const double u = KEEP_U ? v[p] : v[p] - centre;
            double da = u + a.c[7], db = u + b.c[7];
#pragma unroll
            for (int k = 6; k >= 0; --k) { da = __builtin_fma(da, u, a.c[k]); db = __builtin_fma(db, u, b.c[k]); }
            double na = __builtin_fma(a.n[6], u, a.n[5]), nb = __builtin_fma(b.n[6], u, b.n[5]);
#pragma unroll
            for (int k = 4; k >= 0; --k) { na = __builtin_fma(na, u, a.n[k]); nb = __builtin_fma(nb, u, b.n[k]); }
            const double num = __builtin_fma(na, db, nb*da);
            acc[p] = __builtin_fma(num, rcp_newton(da*db), acc[p]);
        }
    }
    __shared__ double partial[4][P][64];
#pragma unroll
    for (int p = 0; p < P; ++p) partial[wave][p][lane] = acc[p];
    __syncthreads();
    for (int p = wave; p < P; p += 4)
        out[base + p*64 + lane] = (partial[0][p][lane] + partial[1][p][lane]) +
                                  (partial[2][p][lane] + partial[3][p][lane]);
}

// Host: coefficients of one quad around u0.
static void quad_of(const Line * l, double u0, Quad & q)
{
    double m[4], c[4], b[4];
    for (int i = 0; i < 4; ++i) { const double a = l[i].centre - u0; m[i] = -2.*a; c[i] = a*a + l[i].g2; b[i] = l[i].bl; }
    auto pair = [&](int i, int j, double * D, double * N) {
        D[4] = 1.; D[3] = m[i] + m[j]; D[2] = c[i] + c[j] + m[i]*m[j]; D[1] = m[i]*c[j] + m[j]*c[i]; D[0] = c[i]*c[j];
        N[2] = b[i] + b[j]; N[1] = b[i]*m[j] + b[j]*m[i]; N[0] = b[i]*c[j] + b[j]*c[i];
    };
    double D1[5], N1[3], D2[5], N2[3];
    pair(0, 1, D1, N1); pair(2, 3, D2, N2);
    double D[9] = {0}, N[7] = {0};
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) D[i + j] += D1[i]*D2[j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 5; ++j) N[i + j] += N1[i]*D2[j] + N2[i]*D1[j];
    for (int k = 0; k < 8; ++k) q.c[k] = D[k];
    for (int k = 0; k < 7; ++k) q.n[k] = N[k];
    q.pad = 0.;
}

int main()
{
    const int n_lines = 5120, P = 4;
    const int blocks = 19531;           // tiles of the 5 M-point benchmark
    const double dv = 0.001, v0 = 2000.;
    std::vector<Line> lines(n_lines);
    srand(7);
    for (int i = 0; i < n_lines; ++i)
    {
        // Lines 1..26 cm-1 away on either side of the first tile (far wing everywhere).
        const double off = 1. + 25.*(rand()/(double)RAND_MAX);
        lines[i].centre = v0 + ((i & 1) ? off : -off);
        lines[i].g2 = 0.07*0.07*(0.5 + rand()/(double)RAND_MAX);
        lines[i].bl = 1e-25*(0.5 + rand()/(double)RAND_MAX);
        lines[i].first = 0; lines[i].last = 1 << 30;
    }
    const double u0 = v0 + 0.128;
    std::vector<Quad> quads(n_lines/4);
    for (int q = 0; q < n_lines/4; ++q) quad_of(&lines[4*q], u0, quads[q]);
    Line * d_lines; Quad * d_quads; double * d_out;
    CHECK(hipMalloc(&d_lines, lines.size()*sizeof(Line)));
    CHECK(hipMalloc(&d_quads, quads.size()*sizeof(Quad)));
    CHECK(hipMalloc(&d_out, (size_t)blocks*64*P*8));
    CHECK(hipMemcpy(d_lines, lines.data(), lines.size()*sizeof(Line), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_quads, quads.data(), quads.size()*sizeof(Quad), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> ref(256), got(256);
    auto run = [&](const char * name, auto launch, std::vector<double> & first) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(first.data(), d_out, 256*8, hipMemcpyDeviceToHost));
        const double evals = (double)blocks*64*P*n_lines;
        printf("%-44s %8.3f ms  %.3e evals/s\n", name, ms/reps, evals/(ms/reps*1e-3));
    };
    // Every block uses the same records (L2-resident): the loop bodies are what is compared.  The
    // first tile is the one the quads were expanded around; only its values are compared.
    run("direct: 8 lines per reciprocal", [&] { hipLaunchKernelGGL((direct_loop<P>), dim3(blocks), dim3(256), 0, 0, d_lines, n_lines, v0, dv, d_out); }, ref);
    run("quads, u recomputed per point", [&] { hipLaunchKernelGGL((poly_loop<P, 0>), dim3(blocks), dim3(256), 0, 0, d_quads, n_lines/4, v0, dv, u0, d_out); }, got);
    double worst = 0.;
    for (int i = 0; i < 256; ++i) worst = fmax(worst, fabs(got[i] - ref[i])/ref[i]);
    printf("   first tile: max relative difference to the direct loop %.3e\n", worst);
    run("quads, u kept in registers", [&] { hipLaunchKernelGGL((poly_loop<P, 1>), dim3(blocks), dim3(256), 0, 0, d_quads, n_lines/4, v0, dv, u0, d_out); }, got);
    return 0;
}
