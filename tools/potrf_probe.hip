// Cycle probe for the diagonal-tile factorisation pieces (development tool).
#include "../mageslam_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <vector>
using namespace mage;
namespace mage { namespace {
#ifndef PROBE_NW
#define PROBE_NW 4
#endif
__global__ __launch_bounds__(64 * PROBE_NW) void k_probe(const double* __restrict__ S, int ld, double* __restrict__ Linv, long long* __restrict__ out)
{
    extern __shared__ double sm[];
    double* A = sm; double* Li = sm + PACKED_TILE_DOUBLES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) load_tile_packed(A, S, ld, tid);              // the library's layout of a diagonal tile (LayPacked)
    __syncthreads();
    long long t0 = clock64();
    bool f = false;
    if (wave == 0) {                                             // block (0, 0) of the tile with four payload rows: the identity and three strips
        if (lane < 64) { for (int q = 0; q < 4; ++q) { const int e = lane * 4 + q; Li[e] = (e >> 4) == (e & 15) ? 1.0 : 0.0; } }
        const int g = lane >> 4, l = lane & 15;
        const int poff = g == 0 ? (int)(Li - A) + l : LayPacked::blk(g, 0) + l;
        double a_ss[NB];
        f = factor_block16_rows<LayPacked::PITCH>(A, LayPacked::blk(0, 0), poff, NB, lane, true, a_ss);
    }
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    if (tid < 256) load_tile_packed(A, S, ld, tid);
    __syncthreads();
    long long t3 = clock64();
#ifdef PROBE_PUBLISH
    f |= potrf_tile_rows<false, LayPacked, 4, PROBE_NW>(A, Li, Linv, tid, NBLK, TilePublish{ Linv + 8 * 256, reinterpret_cast<int*>(out + 8), 0 });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    f |= potrf_tile_rows<false, LayPacked, 0, PROBE_NW>(A, Li, Linv, tid);
#endif
    long long t4 = clock64();
    __syncthreads();
    if (tid < 256) store_tile_packed(const_cast<double*>(S) + (size_t)ld * ld, A, ld, tid);      // second ld x ld matrix of the buffer receives L
    if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t4 - t3; out[3] = f; }
    if (tid == 64) { out[4] = t1 - t0; }
}
} }
int main()
{
    const int n = 128;
    std::vector<double> A((size_t)n * n, 0.0);
    srand(7);
    for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) { double v = (double)rand() / RAND_MAX - 0.5; A[(size_t)j * n + i] = v; A[(size_t)i * n + j] = v; }
    for (int i = 0; i < n; ++i) A[(size_t)i * n + i] = 70.0;
    double *dS, *dL; long long* dout;
    hipMalloc(&dS, sizeof(double) * n * n * 2); hipMalloc(&dL, sizeof(double) * (8 * 256 + LPUB_TILE_DOUBLES)); hipMalloc(&dout, 128);
    hipMemcpy(dS, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    const size_t lds = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64 * PROBE_NW), lds, 0, dS, n, dL, dout);
        hipDeviceSynchronize();
        long long o[5]; hipMemcpy(o, dout, 40, hipMemcpyDeviceToHost);
        printf("pivot block + three strips + inverse %lld cycles (another wave %lld), barrier %lld, potrf_tile_rows %lld cycles, failed %lld\n", o[0], o[4], o[1], o[2], o[3]);
    }
    {   // residuals: || A - L L^T || / || A || over the lower triangle, and || Linv_b L_bb - I || for the eight diagonal blocks
        std::vector<double> L((size_t)n * n), Li(8 * 256);
        hipMemcpy(L.data(), dS + (size_t)n * n, sizeof(double) * n * n, hipMemcpyDeviceToHost);
        hipMemcpy(Li.data(), dL, sizeof(double) * 8 * 256, hipMemcpyDeviceToHost);
        double num = 0, den = 0;
        for (int j = 0; j < n; ++j) for (int i = j; i < n; ++i) {
            double acc = 0;
            for (int k = 0; k <= j; ++k) acc += L[(size_t)k * n + i] * L[(size_t)k * n + j];
            const double d = acc - A[(size_t)j * n + i];
            num += d * d; den += A[(size_t)j * n + i] * A[(size_t)j * n + i];
        }
        double worst = 0;
        for (int b = 0; b < 8; ++b) for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            double acc = 0;       // (Linv L)[r][c], Linv stored as Li[b][row * 16 + col]
            for (int k = 0; k < 16; ++k) {
                const double lkc = (k >= c) ? L[(size_t)(b * 16 + c) * n + b * 16 + k] : 0.0;
                acc += Li[b * 256 + r * 16 + k] * lkc;
            }
            worst = std::max(worst, std::abs(acc - (r == c ? 1.0 : 0.0)));
        }
        printf("residual ||A - L L^T|| / ||A|| = %.3e   max |Linv L - I| = %.3e\n", std::sqrt(num / den), worst);
    }
#ifdef CHOL_TILE_STAMPS
    {   // per in-tile iteration: top -> strips done -> past the middle barrier -> pivot block / trailing update done -> past the end barrier
        long long st[4][NBLK][5];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tile_stamps), sizeof(st));
        for (int s = 0; s + 1 < NBLK; ++s) {
            printf("iteration %d (from wavefront 0's top stamp):", s);
            for (int w = 0; w < 4; ++w) { printf("  w%d:", w); for (int p = 0; p < 5; ++p) printf(" %5lld", st[w][s][p] - st[0][s][0]); }
            printf("\n");
        }
    }
#endif
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
