// Round trip v_mfma_f64_16x16x4_f64 -> VALU -> MFMA operand on gfx950: how long a chain "MFMA result feeds one VALU
// instruction that feeds the next MFMA's A operand" takes per link, against the MFMA -> MFMA accumulator chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NVALU>
__global__ void k(double* out, long long* cyc, int iters)
{
    double4_t acc = { 1, 1, 1, 1 };
    double a = 1e-3 * threadIdx.x, b = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        double x = acc[it & 3];
#pragma unroll
        for (int v = 0; v < NVALU; ++v) x = __builtin_fma(x, 0.999, 1e-9);
        if (NVALU > 0) a = x;
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NVALU>
void run()
{
    double* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8);
    const int iters = 4000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<NVALU>, dim3(1), dim3(64), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("MFMA + %d dependent VALU fma per link: %.1f cycles / link\n", NVALU, (double)h / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() { run<0>(); run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
