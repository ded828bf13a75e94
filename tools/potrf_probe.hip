// Cycle probe for the diagonal-tile factorisation pieces (development tool).
#include "../mageslam_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
using namespace mage;
namespace mage { namespace {
__global__ __launch_bounds__(256) void k_probe(const double* __restrict__ S, int ld, double* __restrict__ Linv, long long* __restrict__ out)
{
    extern __shared__ double sm[];
    double* A = sm; double* Li = sm + TILE * LDC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    load_tile<LDC>(A, S, ld, tid);
    __syncthreads();
    long long t0 = clock64();
    bool f = false;
    if (wave == 0) f = factor_block16(A, 0, lane, Li, Linv);
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    load_tile<LDC>(A, S, ld, tid);
    __syncthreads();
    long long t3 = clock64();
    f |= potrf_tile_lds(A, Li, Linv, tid);
    long long t4 = clock64();
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
    hipMalloc(&dS, sizeof(double) * n * n); hipMalloc(&dL, sizeof(double) * 8 * 256); hipMalloc(&dout, 64);
    hipMemcpy(dS, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    const size_t lds = ((size_t)TILE * LDC + 2 * NB * NB) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), lds, 0, dS, n, dL, dout);
        hipDeviceSynchronize();
        long long o[5]; hipMemcpy(o, dout, 40, hipMemcpyDeviceToHost);
        printf("factor chol %lld cycles (inverse wave %lld), barrier %lld, potrf_tile_lds %lld cycles, failed %lld\n", o[0], o[4], o[1], o[2], o[3]);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
