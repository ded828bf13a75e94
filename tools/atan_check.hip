#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cfloat>
#include <cstdlib>
__host__ __device__ inline float fa(float y, float x, int mode)
{
#pragma clang fp contract(off)
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float)2.2204460492503131e-16;
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + eps); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + eps); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (mode == 1) return c;
    if (mode == 2) return c2;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
__global__ void k(const float* y, const float* x, float* o, int n, int mode) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = fa(y[i], x[i], mode); }
int main()
{
    const int n = 100000;
    float *hy = new float[n], *hx = new float[n], *ho = new float[n];
    srand(1);
    for (int i = 0; i < n; ++i) { hy[i] = (float)(rand() % 200001 - 100000); hx[i] = (float)(rand() % 200001 - 100000); }
    float *dy, *dx, *dout; hipMalloc(&dy, 4 * n); hipMalloc(&dx, 4 * n); hipMalloc(&dout, 4 * n);
    hipMemcpy(dy, hy, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dx, hx, 4 * n, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dy, dx, dout, n, mode);
        hipMemcpy(ho, dout, 4 * n, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) { float r = fa(hy[i], hx[i], mode); if (r != ho[i]) { if (bad < 3) printf("mode %d: y %g x %g host %.9g dev %.9g\n", mode, hy[i], hx[i], r, ho[i]); ++bad; } }
        printf("mode %d mismatches %d / %d\n", mode, bad, n);
    }
    return 0;
}
