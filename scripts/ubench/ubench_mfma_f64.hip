// Does v_mfma_f64_16x16x4_f64 run beside fp64 VALU work on gfx950?  One wave per SIMD (and
// four), a loop of one MFMA plus K independent v_fma_f64 fillers; cycles per iteration from
// s_memtime.  Also prints the C/D register layout of the instruction.  Not part of the product.
//
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/ubench_mfma_f64.hip -o /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int FILL, int MFMAS>
__global__ __launch_bounds__(256) void loop_kernel(int iterations, double seed, double * out,
                                                   long long * cycles)
{
    double4_t acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = double4_t{0., 0., 0., 0.};
    double a = seed + threadIdx.x, b = seed*0.5 + threadIdx.x;
    double f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = seed + k + threadIdx.x;
    const long long start = __builtin_readcyclecounter();
    for (int i = 0; i < iterations; ++i)
    {
#pragma unroll
        for (int m = 0; m < MFMAS; ++m)
        {
            acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < FILL; ++k) f[k] = __builtin_fma(f[k], 1.0000001, 0.5);
    }
    const long long stop = __builtin_readcyclecounter();
    double total = 0.;
    for (int m = 0; m < 4; ++m) total += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) total += f[k];
    out[blockIdx.x*blockDim.x + threadIdx.x] = total;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = stop - start;
}

__global__ void layout_kernel(double * out)
{
    // A[i][k] = 100 i + k at lane (k*16 + i)?  B[k][j] = 1 if k == 0 -> D[i][j] = A[i][0].
    const int lane = threadIdx.x;
    const double a = (double)(lane % 16) + 1000.*(lane/16);      // row = lane%16, k = lane/16
    const double b = (lane/16 == 0) ? 1. : 0.;                   // picks k = 0
    double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, double4_t{0., 0., 0., 0.}, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane*4 + r] = d[r];
    // second probe: B[k][j] = j for k == 0, A = 1 for k == 0 -> D[i][j] = j
    const double a2 = (lane/16 == 0) ? 1. : 0.;
    const double b2 = (lane/16 == 0) ? (double)(lane % 16) : 0.;
    double4_t e = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, double4_t{0., 0., 0., 0.}, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + lane*4 + r] = e[r];
}

template <int FILL, int MFMAS>
void run(int blocks, int threads, double * d_out, long long * d_cycles)
{
    const int iterations = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop_kernel<FILL, MFMAS>), dim3(blocks), dim3(threads), 0, 0, 100, 1.0, d_out, d_cycles);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop_kernel<FILL, MFMAS>), dim3(blocks), dim3(threads), 0, 0, iterations, 1.0, d_out, d_cycles);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cycles; CHECK(hipMemcpy(&cycles, d_cycles, 8, hipMemcpyDeviceToHost));
    printf("mfma/iter %d fillers %2d waves/SIMD %d: %7.1f clock ticks/iter (s_memtime), %7.2f ns/iter\n",
           MFMAS, FILL, threads/256 > 0 ? threads/256 : 1, (double)cycles/iterations, ms*1e6/iterations);
}

int main()
{
    double * d_out; long long * d_cycles;
    CHECK(hipMalloc(&d_out, 1 << 24)); CHECK(hipMalloc(&d_cycles, 8));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<double> h(512);
    CHECK(hipMemcpy(h.data(), d_out, 512*8, hipMemcpyDeviceToHost));
    printf("D layout: lane -> (row i = value of probe 1 [A row index], col j = probe 2)\n");
    for (int lane = 0; lane < 64; lane += 5)
    {
        printf("  lane %2d:", lane);
        for (int r = 0; r < 4; ++r) printf(" r%d=(i=%g,j=%g)", r, h[lane*4 + r], h[256 + lane*4 + r]);
        printf("\n");
    }
    const int cus = 256;
    // one wave per SIMD: 256-thread blocks, one block per CU
    printf("--- one wave per SIMD ---\n");
    run<0, 1>(cus, 256, d_out, d_cycles);
    run<4, 1>(cus, 256, d_out, d_cycles);
    run<8, 1>(cus, 256, d_out, d_cycles);
    run<12, 1>(cus, 256, d_out, d_cycles);
    run<14, 1>(cus, 256, d_out, d_cycles);
    run<16, 1>(cus, 256, d_out, d_cycles);
    run<0, 4>(cus, 256, d_out, d_cycles);
    run<16, 4>(cus, 256, d_out, d_cycles);
    run<16, 0>(cus, 256, d_out, d_cycles);
    run<8, 0>(cus, 256, d_out, d_cycles);
    printf("--- four waves per SIMD (4 blocks per CU) ---\n");
    run<0, 1>(cus*4, 256, d_out, d_cycles);
    run<8, 1>(cus*4, 256, d_out, d_cycles);
    run<12, 1>(cus*4, 256, d_out, d_cycles);
    run<14, 1>(cus*4, 256, d_out, d_cycles);
    run<16, 1>(cus*4, 256, d_out, d_cycles);
    run<16, 0>(cus*4, 256, d_out, d_cycles);
    return 0;
}
