// Micro-benchmarks that size the far-wing (Lorentz) inner loop of the Voigt
// accumulate kernel on gfx950: fp64 instruction issue rates and candidate loop
// bodies (one reciprocal per evaluation, shared reciprocal for 2 / 4 lines,
// fp32-seeded reciprocal, IEEE division).  Not part of the product.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/ubench/ubench_fp64.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct LineRec { double nu, g2, b, pad; };

__device__ __forceinline__ double rcp_nr1(double t)
{
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}

__device__ __forceinline__ double rcp_nr2(double t)
{
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}

__device__ __forceinline__ double rcp_f32seed(double t)
{
    float tf = (float)t;
    double r = (double)__builtin_amdgcn_rcpf(tf);
    double e = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}

// VARIANT: 0 = IEEE divide, 1 = rcp + 1 NR, 2 = rcp + 2 NR, 3 = f32 seed + 2 NR,
//          4 = pair (one rcp for two lines), 5 = quad (one rcp for four lines),
//          6 = raw rcp only (accuracy probe)
template <int VARIANT, int P>
__global__ __launch_bounds__(256) void lorentz_loop(const LineRec * __restrict__ lines, int n_lines,
                                                    double v0, double dv, double * __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x*blockDim.x + threadIdx.x) >> 6;
    const long base = (long)wave*64*P;
    double v[P], acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        v[p] = v0 + (double)(base + p*64 + lane)*dv;
        acc[p] = 0.;
    }
    if (VARIANT <= 3 || VARIANT == 6)
    {
        for (int l = 0; l < n_lines; ++l)
        {
            const double nu = lines[l].nu, g2 = lines[l].g2, b = lines[l].b;
#pragma unroll
            for (int p = 0; p < P; ++p)
            {
                double d = v[p] - nu;
                double t = __builtin_fma(d, d, g2);
                double r;
                if (VARIANT == 0) r = 1.0/t;
                else if (VARIANT == 1) r = rcp_nr1(t);
                else if (VARIANT == 2) r = rcp_nr2(t);
                else if (VARIANT == 3) r = rcp_f32seed(t);
                else r = __builtin_amdgcn_rcp(t);
                acc[p] = __builtin_fma(b, r, acc[p]);
            }
        }
    }
    else if (VARIANT == 4)
    {
        for (int l = 0; l + 1 < n_lines; l += 2)
        {
            const double nu1 = lines[l].nu, g21 = lines[l].g2, b1 = lines[l].b;
            const double nu2 = lines[l+1].nu, g22 = lines[l+1].g2, b2 = lines[l+1].b;
#pragma unroll
            for (int p = 0; p < P; ++p)
            {
                double d1 = v[p] - nu1;
                double d2 = v[p] - nu2;
                double t1 = __builtin_fma(d1, d1, g21);
                double t2 = __builtin_fma(d2, d2, g22);
                double num = __builtin_fma(b1, t2, b2*t1);
                double r = rcp_nr1(t1*t2);
                acc[p] = __builtin_fma(num, r, acc[p]);
            }
        }
    }
    else if (VARIANT == 5)
    {
        for (int l = 0; l + 3 < n_lines; l += 4)
        {
            const double nu1 = lines[l].nu, g21 = lines[l].g2, b1 = lines[l].b;
            const double nu2 = lines[l+1].nu, g22 = lines[l+1].g2, b2 = lines[l+1].b;
            const double nu3 = lines[l+2].nu, g23 = lines[l+2].g2, b3 = lines[l+2].b;
            const double nu4 = lines[l+3].nu, g24 = lines[l+3].g2, b4 = lines[l+3].b;
#pragma unroll
            for (int p = 0; p < P; ++p)
            {
                double d1 = v[p] - nu1, d2 = v[p] - nu2, d3 = v[p] - nu3, d4 = v[p] - nu4;
                double t1 = __builtin_fma(d1, d1, g21);
                double t2 = __builtin_fma(d2, d2, g22);
                double t3 = __builtin_fma(d3, d3, g23);
                double t4 = __builtin_fma(d4, d4, g24);
                double n12 = __builtin_fma(b1, t2, b2*t1);
                double n34 = __builtin_fma(b3, t4, b4*t3);
                double t12 = t1*t2, t34 = t3*t4;
                double num = __builtin_fma(n12, t34, n34*t12);
                double r = rcp_nr1(t12*t34);
                acc[p] = __builtin_fma(num, r, acc[p]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
    {
        out[base + p*64 + lane] = acc[p];
    }
}

// Raw issue-rate probes: CHAINS independent dependency chains per lane.
template <int OP>
__global__ __launch_bounds__(256) void op_rate(double * out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i*0.001 + threadIdx.x*1e-6;
    const double c1 = 1.0000001, c2 = 1e-9;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            if (OP == 0) a[i] = __builtin_fma(a[i], c1, c2);
            else if (OP == 1) a[i] = a[i] + c2;
            else if (OP == 2) a[i] = a[i]*c1;
            else if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
            else if (OP == 4) a[i] = (double)(float)a[i] + c2;              // cvt both ways + add
            else if (OP == 5) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]);
            else if (OP == 6) a[i] = __builtin_amdgcn_rsq(a[i]);
            else if (OP == 7) a[i] = __builtin_fmax(a[i], c1);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int VARIANT, int P>
double run_loop(const LineRec * d_lines, int n_lines, double * d_out, int waves, std::vector<double> * host_out,
                const char * name)
{
    int blocks = waves/4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    lorentz_loop<VARIANT, P><<<blocks, 256>>>(d_lines, n_lines, 1000., 0.001, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r)
        lorentz_loop<VARIANT, P><<<blocks, 256>>>(d_lines, n_lines, 1000., 0.001, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double evals = (double)waves*64*P*(double)n_lines;
    printf("%-28s P=%2d waves=%6d lines=%5d  %8.3f ms  %10.4g evals/s\n", name, P, waves, n_lines, ms,
           evals/(ms*1e-3));
    if (host_out)
    {
        host_out->resize((size_t)waves*64*P);
        CHECK(hipMemcpy(host_out->data(), d_out, host_out->size()*sizeof(double), hipMemcpyDeviceToHost));
    }
    return ms;
}

template <int OP>
void run_op(double * d_out, const char * name, int n_ops_per_iter_elem)
{
    const int blocks = 256*8, iters = 20000;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    op_rate<OP><<<blocks, 256>>>(d_out, 10, 1.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    op_rate<OP><<<blocks, 256>>>(d_out, iters, 1.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_insts = (double)blocks*4*iters*8*n_ops_per_iter_elem;
    double per_simd = wave_insts/1024.;   // 1024 SIMDs
    double ns_per = ms*1e6/per_simd;
    printf("%-34s %8.3f ms  %7.3f ns per wave-instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", name, ms,
           ns_per, ns_per*2.4);
}

double max_rel(const std::vector<double> & a, const std::vector<double> & b)
{
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i)
    {
        double r = fabs(a[i] - b[i])/fabs(b[i]);
        if (r > m) m = r;
    }
    return m;
}

int main()
{
    const int n_lines = 2048;
    std::vector<LineRec> lines(n_lines);
    srand(7);
    for (int i = 0; i < n_lines; ++i)
    {
        double u = rand()/(double)RAND_MAX;
        lines[i].nu = 990. + 60.*u;   // mostly outside the tile range 1000..1000+tile
        if (lines[i].nu > 999.5 && lines[i].nu < 1020.) lines[i].nu += 25.;
        double g = 0.01 + 0.1*(rand()/(double)RAND_MAX);
        lines[i].g2 = g*g;
        lines[i].b = 1e-22*(0.1 + rand()/(double)RAND_MAX);
        lines[i].pad = 0;
    }
    LineRec * d_lines;
    double * d_out;
    CHECK(hipMalloc(&d_lines, n_lines*sizeof(LineRec)));
    CHECK(hipMemcpy(d_lines, lines.data(), n_lines*sizeof(LineRec), hipMemcpyHostToDevice));
    const int max_waves = 1024*16;
    CHECK(hipMalloc(&d_out, (size_t)max_waves*64*16*sizeof(double)));

    printf("== issue rates (8 waves/SIMD, 8 independent chains per lane) ==\n");
    run_op<0>(d_out, "v_fma_f64", 1);
    run_op<1>(d_out, "v_add_f64", 1);
    run_op<2>(d_out, "v_mul_f64", 1);
    run_op<3>(d_out, "v_rcp_f64", 1);
    run_op<4>(d_out, "cvt_f32_f64+cvt_f64_f32+add (3 ops)", 3);
    run_op<5>(d_out, "cvt+rcp_f32+cvt (3 ops)", 3);
    run_op<6>(d_out, "v_rsq_f64", 1);
    run_op<7>(d_out, "v_max_f64", 1);

    printf("== Lorentz loop bodies ==\n");
    std::vector<double> ref, got;
    for (int waves : {1024, 2048, 4096, 8192})
    {
        run_loop<1, 8>(d_lines, n_lines, d_out, waves, nullptr, "rcp+1NR");
    }
    const int waves = 4096;
    run_loop<0, 8>(d_lines, n_lines, d_out, waves, &ref, "IEEE divide");
    run_loop<1, 8>(d_lines, n_lines, d_out, waves, &got, "rcp+1NR");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    run_loop<2, 8>(d_lines, n_lines, d_out, waves, &got, "rcp+2NR");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    run_loop<3, 8>(d_lines, n_lines, d_out, waves, &got, "f32 seed+2NR");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    run_loop<4, 8>(d_lines, n_lines, d_out, waves, &got, "pair (1 rcp / 2 lines)");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    run_loop<5, 8>(d_lines, n_lines, d_out, waves, &got, "quad (1 rcp / 4 lines)");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    run_loop<6, 8>(d_lines, n_lines, d_out, waves, &got, "raw rcp (no NR)");
    printf("   max rel err vs divide: %.3e\n", max_rel(got, ref));
    for (int w : {1024, 2048, 4096, 8192})
    {
        run_loop<4, 4>(d_lines, n_lines, d_out, w, nullptr, "pair P=4");
        run_loop<4, 8>(d_lines, n_lines, d_out, w, nullptr, "pair P=8");
        run_loop<4, 16>(d_lines, n_lines, d_out, w, nullptr, "pair P=16");
        run_loop<5, 4>(d_lines, n_lines, d_out, w, nullptr, "quad P=4");
        run_loop<5, 8>(d_lines, n_lines, d_out, w, nullptr, "quad P=8");
        run_loop<1, 4>(d_lines, n_lines, d_out, w, nullptr, "rcp+1NR P=4");
        run_loop<1, 16>(d_lines, n_lines, d_out, w, nullptr, "rcp+1NR P=16");
    }
    return 0;
}
