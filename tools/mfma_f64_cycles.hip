// cycles per v_mfma_f64_16x16x4_f64 (s_memtime) for different accumulator counts / waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int threads, int grid, const char* tag)
{
    double* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, sizeof(double) * 1 << 22); (void)hipMalloc(&cyc, 8);
    int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * NACC;
    printf("%-28s NACC=%2d thr=%4d grid=%5d: %.1f memtime-ticks/MFMA/wave, wall %.3f ms -> %.1f ns/MFMA/wave\n", tag, NACC, threads, grid,
           h / n, ms, ms * 1e6 / n);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<1>(64, 1, "1 wave, dependent chain");
    run<2>(64, 1, "1 wave");
    run<4>(64, 1, "1 wave");
    run<8>(64, 1, "1 wave");
    run<4>(256, 1, "4 waves (1/SIMD), 1 CU");
    run<4>(512, 1, "8 waves (2/SIMD), 1 CU");
    run<4>(256, 256, "1/SIMD, all CUs");
    run<4>(256, 1024, "4 blocks/CU, all CUs");
    run<8>(256, 256, "1/SIMD, all CUs");
    run<16>(256, 256, "1/SIMD, all CUs");
    return 0;
}
