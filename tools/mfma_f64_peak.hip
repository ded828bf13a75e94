// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on the whole chip.
// ("AMD public spec 78.6 TFLOP/s f64 matrix = vector -- microbenchmark the ceiling first".)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* out;
    hipMalloc(&out, 4096 * 256 * sizeof(double));
    const int iters = 4000;
    for (int blocks_per_cu : { 1, 2, 4 }) {
        int grid = 256 * blocks_per_cu;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters); });
        double fl = (double)grid * 4 * iters * 4 * 2048.0;
        printf("mfma_f64 NACC=4  %d blocks/CU: %.2f TFLOP/s\n", blocks_per_cu, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(grid), dim3(256), 0, 0, out, iters / 4); });
        fl = (double)grid * 4 * (iters / 4) * 16 * 2048.0;
        printf("mfma_f64 NACC=16 %d blocks/CU: %.2f TFLOP/s\n", blocks_per_cu, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters); });
        fl = (double)grid * 256 * iters * 16 * 2.0;
        printf("v_fma_f64        %d blocks/CU: %.2f TFLOP/s\n", blocks_per_cu, fl / ms / 1e9);
    }
    return 0;
}
