// Latency probe: cycles per DEPENDENT f64 operation on a lone wavefront (what the 16 x 16 pivot block's recurrence is made of), and how
// many independent operations fit between two dependent ones for free.
//   hipcc --offload-arch=gfx950 -O2 tools/f64_latency.hip -o tools/_bin/f64_latency && tools/_bin/f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void k(long long* out, double* sink, double a, double b)
{
    double x = threadIdx.x * 1e-3 + 1.0, y0 = 1.0, y1 = 2.0, y2 = 3.0, y3 = 4.0, y4 = 5.0, y5 = 6.0;
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));) }
    if (MODE == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(y0) : "v"(a), "v"(b));) }
    if (MODE == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(a), "v"(b));) }
    if (MODE == 3) { REP64(asm volatile("v_fma_f64 %0, %0, %7, %8\n v_fma_f64 %1, %1, %7, %8\n v_fma_f64 %2, %2, %7, %8\n v_fma_f64 %3, %3, %7, %8\n v_fma_f64 %4, %4, %7, %8\n v_fma_f64 %5, %5, %7, %8\n v_fma_f64 %6, %6, %7, %8" : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5) : "v"(a), "v"(b));) }
    if (MODE == 4) { REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(x));) }
    if (MODE == 5) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));) }
    if (MODE == 6) { REP64(asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a));) }
    if (MODE == 7) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_nop 1\n v_fmac_f64_dpp %1, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y0) : "v"(a), "v"(b));) }
    if (MODE == 8) { REP64(asm volatile("v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %7, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %7, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %7, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5) : "v"(a), "v"(b));) }
    if (MODE == 9) { REP64(asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %0, %2, %3" : "+v"(x), "+v"(y0) : "v"(a), "v"(b));) }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = x + y0 + y1 + y2 + y3 + y4 + y5;
}

template <int MODE> void run(const char* what, int per_rep)
{
    long long* d; double* s; hipMalloc(&d, 8); hipMalloc(&s, 64 * 8);
    long long best = 1ll << 60;
    for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, s, 0.999, 1e-3); long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); if (h < best) best = h; }
    printf("%-70s %7.2f cycles per repetition (%d instruction(s)), %6.2f per instruction\n", what, best / 64.0, per_rep, best / 64.0 / per_rep);
    hipFree(d); hipFree(s);
}
int main()
{
    run<0>("v_fma_f64, each depending on the one before", 1);
    run<1>("1 dependent + 1 independent v_fma_f64", 2);
    run<2>("1 dependent + 3 independent", 4);
    run<3>("1 dependent + 6 independent", 7);
    run<4>("v_rsq_f64, dependent", 1);
    run<5>("v_mul_f64, dependent", 1);
    run<6>("v_fmac_f64_dpp (row_newbcast) of its own result, behind s_nop 1", 1);
    run<7>("v_fma_f64 -> s_nop 1 -> v_fmac_f64_dpp reading it", 2);
    run<8>("7 independent v_fmac_f64_dpp (row_newbcast)", 7);
    run<9>("v_mov_b64_dpp of a value -> v_fma_f64 on it (dependent pair, no s_nop)", 2);
    return 0;
}
