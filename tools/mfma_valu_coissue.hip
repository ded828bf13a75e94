// Do f64 matrix operations and f64 vector FMAs of two wavefronts on ONE SIMD overlap on gfx950?  (development probe)
// Eight wavefronts per workgroup, one workgroup per compute unit: wavefronts 0-3 issue independent v_mfma_f64_16x16x4, wavefronts 4-7
// (the same four SIMDs) independent v_fma_f64; each role alone, then both together; shader clocks per role.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, long long* out, double* sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = wave < 4;
    double4_t c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = lane + i;
    __syncthreads();
    const long long t0 = clock64();
    if (mf && (mode & 1)) {
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    if (!mf && (mode & 2)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], a, b);
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], a, b);
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], a, b);
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], a, b);
        }
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[mode * 8 + wave] = t1 - t0;
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += f[i];
    if (s == 12345.678) sink[threadIdx.x] = s;
}
int main()
{
    long long* d; double* sink; hipMalloc(&d, 8 * 4 * sizeof(long long)); hipMalloc(&sink, 4096);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 1; mode <= 3; ++mode) { hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d, sink); hipDeviceSynchronize(); }
    long long h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = { "", "matrix only", "vector only", "both" };
    for (int mode = 1; mode <= 3; ++mode) {
        printf("%-12s", names[mode]);
        if (mode & 1) printf("  matrix wavefront: %6.1f cycles per v_mfma_f64_16x16x4 (1024 FMA)", (double)h[mode * 8 + 0] / (4.0 * iters));
        if (mode & 2) printf("  vector wavefront: %5.2f cycles per v_fma_f64 (64 FMA)", (double)h[mode * 8 + 4] / (64.0 * iters));
        printf("\n");
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
