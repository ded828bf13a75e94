// Is v_mfma_f64_16x16x4's result the chain fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c))))?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
// operand layout (chol_kernels.hip): A operand of lane l = A[row l & 15][k = l >> 4], B operand = B[k = l >> 4][col l & 15],
// accumulator register r of lane l = D[row 4 r + (l >> 4)][col l & 15]
__global__ void k(const double* A, const double* B, const double* C, double* D)
{
    const int l = threadIdx.x;
    double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * r + (l >> 4)) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * r + (l >> 4)) * 16 + (l & 15)] = c[r];
}
int main()
{
    std::vector<double> A(64), B(64), C(256), D(256);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    long asc = 0, desc = 0, pair = 0, total = 0;
    srand(3);
    for (int trial = 0; trial < 200; ++trial) {
        auto rnd = [&] { return ((double)rand() / RAND_MAX - 0.5) * std::pow(2.0, rand() % 20 - 10); };
        for (auto& v : A) v = rnd(); for (auto& v : B) v = rnd(); for (auto& v : C) v = rnd();
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            // the kernel above computes D[m][n] with A[m][k], B[k][n] in the layout of its comment: m = row of the accumulator
            double up = C[i * 16 + j], down = C[i * 16 + j];
            for (int kk = 0; kk < 4; ++kk) up = std::fma(A[i * 4 + kk], B[kk * 16 + j], up);
            for (int kk = 3; kk >= 0; --kk) down = std::fma(A[i * 4 + kk], B[kk * 16 + j], down);
            const double pr = std::fma(A[i * 4 + 1], B[16 + j], std::fma(A[i * 4 + 0], B[j], 0.0)) + std::fma(A[i * 4 + 3], B[48 + j], std::fma(A[i * 4 + 2], B[32 + j], 0.0)) + C[i * 16 + j];
            const double d = D[i * 16 + j];
            asc += std::memcmp(&d, &up, 8) == 0; desc += std::memcmp(&d, &down, 8) == 0; pair += std::memcmp(&d, &pr, 8) == 0; ++total;
        }
    }
    printf("%ld results: equal to the ascending FMA chain %ld, to the descending chain %ld, to pairwise sums %ld\n", total, asc, desc, pair);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
