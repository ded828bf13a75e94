// Issue-rate probe (one wavefront): cycles per instruction for the f64 operations the diagonal factorisation uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void k(long long* out, double* sink)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0000001, c = 0.5;
    long long t[8];
    t[0] = clock64();
    REP64(asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    t[1] = clock64();
    REP64(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    t[2] = clock64();
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0" : "+v"(a0) : "v"(c));)
    t[3] = clock64();
    REP64(asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %4, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %4, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a1), "+v"(b), "+v"(a2), "+v"(a3), "+v"(a4));)
    t[4] = clock64();
    REP64(asm volatile("v_rsq_f64 %0, %1\n v_rsq_f64 %2, %1\n v_rsq_f64 %3, %1\n v_rsq_f64 %4, %1\n v_rsq_f64 %0, %1\n v_rsq_f64 %2, %1\n v_rsq_f64 %3, %1\n v_rsq_f64 %4, %1" : "+v"(a1), "+v"(b), "+v"(a2), "+v"(a3), "+v"(a4));)
    t[5] = clock64();
    // dependent chain through rsq
    REP64(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0" : "+v"(a5));)
    t[6] = clock64();
    // dependent chain through dpp fmac (dest as accumulator)
    REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a6) : "v"(b), "v"(c));)
    t[7] = clock64();
    if (threadIdx.x == 0) for (int i = 0; i < 7; ++i) out[i] = t[i + 1] - t[i];
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    long long* d; double* s; hipMalloc(&d, 64); hipMalloc(&s, 8 * 64);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s); hipDeviceSynchronize(); }
    long long o[7]; hipMemcpy(o, d, 56, hipMemcpyDeviceToHost);
    const char* names[7] = { "v_fma_f64 independent", "v_fmac_f64_dpp independent", "v_fma_f64 dependent", "v_mov_b64_dpp independent", "v_rsq_f64 independent", "v_rsq_f64 dependent", "v_fmac_f64_dpp dependent" };
    for (int i = 0; i < 7; ++i) printf("%-30s %.2f cycles / instruction\n", names[i], o[i] / 512.0);
    return 0;
}
