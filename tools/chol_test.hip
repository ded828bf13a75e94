// Standalone check + timing of chol_factor_solve (no Python): SPD matrix, residual, per-run time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../mageslam_amd/csrc/chol_kernels.h"
#include "../mageslam_amd/csrc/chol_dag.h"
using namespace mage;
extern "C" int mage_debug_chol_schedule(int nt, int n_cu, int fuse_max, unsigned long long* out, int cap, int* quarter_from, int* group_len);
extern "C" int mage_debug_chol_schedule_env(int nt, int n_cu, int fuse_max, const int* env, unsigned long long* out, int cap, int* quarter_from, int* group_len);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    int reps = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<int> sizes = { 128, 256, 640, 6016 };
    if (argc > 1 && atoi(argv[1]) > 0) sizes = { atoi(argv[1]) };
    chol_init_device();
    for (int n : sizes) {
        // banded-ish SPD: A = B B^T + n I with random B entries in a band, dense storage
        std::vector<double> A((size_t)n * n, 0.0), b(n), x(n);
        srand(1234 + n);
        auto rnd = [] { return (double)rand() / RAND_MAX - 0.5; };
        int bw = n < 512 ? n : 200;
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n && i < j + bw; ++i) { double v = rnd(); A[(size_t)j * n + i] = v; A[(size_t)i * n + j] = v; }
        for (int i = 0; i < n; ++i) { A[(size_t)i * n + i] = bw * 0.5 + 1.0 + rnd(); b[i] = rnd(); }
        double *dS, *dS0, *dy, *dy0, *dx, *dok, *dws;
        CK(hipMalloc(&dS, sizeof(double) * n * n)); CK(hipMalloc(&dS0, sizeof(double) * n * n));
        CK(hipMalloc(&dy, sizeof(double) * n)); CK(hipMalloc(&dy0, sizeof(double) * n)); CK(hipMalloc(&dx, sizeof(double) * n));
        CK(hipMalloc(&dok, 16)); CK(hipMalloc(&dws, sizeof(double) * chol_workspace_doubles(n)));
        CK(hipMemcpy(dS0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(dy0, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        int* dq; CK(hipMalloc(&dq, sizeof(int) * chol_sync_ints(n)));
        // CHOL_ENV=1: the skyline of the test matrix (band 200) by tile rows: the task-graph schedule skips the tiles left of it
        std::vector<int> envh;
        if (getenv("CHOL_ENV") && n >= 1024) {
            envh.assign(n / 128, 0);
            for (int R = 0; R < n / 128; ++R) { const int first_col = std::max(0, R * 128 - (bw - 1)); envh[R] = first_col / 128; }
        }
#ifdef DAG_TRACE
        // trace build: per task four stamps + two per tile column, written by the task-graph launch (dumped below)
        std::vector<unsigned long long> tasks(600000);
        int qf = 0;
        const int fuse = getenv("MAGE_CHOL_DAG_FUSE") ? atoi(getenv("MAGE_CHOL_DAG_FUSE")) : 8;
        const int n_tasks = envh.empty() ? mage_debug_chol_schedule(n / 128, 256, fuse, tasks.data(), (int)tasks.size(), &qf, nullptr)
                                         : mage_debug_chol_schedule_env(n / 128, 256, fuse, envh.data(), tasks.data(), (int)tasks.size(), &qf, nullptr);
        const size_t n_stamps = 8 * (size_t)(n_tasks > 0 ? n_tasks : 0) + 2 * (n / 128) + 32;
        long long* ddbg; CK(hipMalloc(&ddbg, sizeof(long long) * n_stamps));
        CholWorkspace ws{ dws, dq, ddbg };
#else
        long long* ddbg; CK(hipMalloc(&ddbg, sizeof(long long) * 4 * (n / 128 + 1)));
        CholWorkspace ws{ dws, dq, getenv("CHOL_DBG") ? ddbg : nullptr };
#endif
        if (!envh.empty()) ws.env_host = envh.data();
        { double bms = 0; const bool dag = chol_dag_wait_schedule(n, &bms, ws.env_host); if (dag) printf("  task lists of %d tile columns built in %.1f ms (worker thread)\n", n / 128, bms); }
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(dS, dS0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(dy, dy0, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
            CK(hipEventRecord(e0, st));
            chol_factor_solve(dS, dy, dx, n, ws, dok, st);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
#ifdef DAG_TRACE
        if (n_tasks > 0) {
            CK(hipMemset(ddbg, 0, sizeof(long long) * n_stamps));
            CK(hipMemcpyAsync(dS, dS0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(dy, dy0, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
            chol_factor_solve(dS, dy, dx, n, ws, dok, st);
            CK(hipStreamSynchronize(st));
            std::vector<long long> stamps(n_stamps);
            CK(hipMemcpy(stamps.data(), ddbg, sizeof(long long) * n_stamps, hipMemcpyDeviceToHost));
            char name[256];
            snprintf(name, sizeof(name), "gpurun_out/dag_trace_%d.bin", n);
            if (FILE* f = fopen(name, "wb")) {
                long long hdr[4] = { n_tasks, n / 128, qf, (long long)n_stamps };
                fwrite(hdr, sizeof(long long), 4, f);
                fwrite(tasks.data(), sizeof(unsigned long long), n_tasks, f);
                fwrite(stamps.data(), sizeof(long long), n_stamps, f);
                fclose(f);
                printf("  trace -> %s (%d tasks)\n", name, n_tasks);
            }
        }
#endif
        double ok, okst[2];
        CK(hipMemcpy(x.data(), dx, sizeof(double) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(okst, dok, 16, hipMemcpyDeviceToHost));
        ok = okst[0];
        if (okst[1] != 0.0) printf("  STALL code %g (a bounded wait ran out: the schedule under test did not finish)\n", okst[1]);
        double rn = 0, bn = 0;
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int j = 0; j < n; ++j) s += A[(size_t)j * n + i] * x[j];
            rn += (s - b[i]) * (s - b[i]); bn += b[i] * b[i];
        }
        if (getenv("CHOL_DBG")) {
            int nt = n / 128;
            std::vector<long long> d(4 * nt);
            CK(hipMemcpy(d.data(), ddbg, sizeof(long long) * 4 * nt, hipMemcpyDeviceToHost));
            long long t0 = d[(nt - 1) * 4];
            for (int j = nt - 1; j >= 0; --j)
                printf("  col %2d: start %8.2f us  precompute done %8.2f  last wait begins %8.2f  published %8.2f\n", j, (d[j * 4] - t0) * 0.01, (d[j * 4 + 1] - t0) * 0.01,
                       (d[j * 4 + 2] - t0) * 0.01, (d[j * 4 + 3] - t0) * 0.01);
        }
        unsigned long long hx = 1469598103934665603ull;      // FNV-1a over the bits of x: two schedules that claim the same numbers can be compared
        for (int i = 0; i < n; ++i) { unsigned long long u; memcpy(&u, &x[i], 8); hx = (hx ^ u) * 1099511628211ull; }
        printf("n=%5d ok=%g  |Ax-b|/|b| = %.3e   best %.3f ms  -> %.2f TFLOP/s (n^3/3)  x-hash %016llx\n", n, ok, sqrt(rn / bn), best,
               (double)n * n * n / 3.0 / (best * 1e-3) / 1e12, hx);
        // indefinite matrix must be flagged
        if (n <= 640) {
            A[(size_t)(n / 2) * n + n / 2] = -5.0;
            CK(hipMemcpy(dS, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(dy, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            chol_factor_solve(dS, dy, dx, n, ws, dok, st);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&ok, dok, 8, hipMemcpyDeviceToHost));
            printf("         indefinite input -> ok=%g (expect 0)\n", ok);
        }
        CK(hipGetLastError());
        CK(hipFree(dS)); CK(hipFree(dS0)); CK(hipFree(dy)); CK(hipFree(dy0)); CK(hipFree(dx)); CK(hipFree(dok)); CK(hipFree(dws));
    }
    return 0;
}
