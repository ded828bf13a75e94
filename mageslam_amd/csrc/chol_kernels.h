// chol_kernels.h -- dense f64 Cholesky factor + solve on gfx950 (see chol_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mage {

constexpr int CHOL_TILE = 128;

struct CholWorkspace {
    double* Linv;      // chol_workspace_doubles(n_pad): inverses of the 16x16 diagonal blocks of L, per tile
    int* sync;         // chol_sync_ints(n_pad) ints: [0] arrival counter of the split diagonal tile, [1] last tile column factored, [2] parts of first-column tiles written, [4] phased strips: block columns of the tile being factored that are published (8 tile + column); from [8] on: the task-graph launch's state (chol_dag.h)
    long long* dbg = nullptr;   // development only: 4 x nt time stamps of the backward solve (tools/chol_test.hip, CHOL_DBG=1); a -DDAG_TRACE build hands it to the task-graph launch instead (tools/dag_trace.py)
    double* stall = nullptr;    // device double: set to 1.0 when a bounded cross-workgroup wait timed out (a device fault, not a property of S)
    const int* env_host = nullptr;   // optional, HOST array of n_pad / CHOL_TILE ints: env[i] = first tile column of tile row i that can hold a non-zero (env[i] <= i).
                                     // The task-graph schedule then skips every tile left of it (they are zero and stay zero in the factor: the same numbers, less work);
                                     // the column-by-column schedules ignore it
};
size_t chol_workspace_doubles(int n_pad);
size_t chol_sync_ints(int n_pad);      // hand-off counters of the launches + the state words of the task-graph launch
constexpr int CHOL_MAX_ORDER = 256 * CHOL_TILE;   // the persistent backward solve needs one resident workgroup per tile column
bool chol_merge_fallback_active();   // true once chol_report_stall(2) has switched this process to separate panel-solve launches
void chol_report_stall(int code);   // the host saw *stall = code (1 split diagonal tile, 2 merged panel solve, 3 backward solve, 4 task-graph launch, 9 a launch was refused): adapts the schedule
void chol_forget_stream(hipStream_t st);   // before hipStreamDestroy of a stream that chol_factor_solve has been given (chol_dag.hip: the launches of a device take turns by events)
void chol_init_device();   // once per device (after hipSetDevice): opt the LDS-heavy kernels in

// Factor S = L L^T in place (lower triangle, column-major, n_pad multiple of CHOL_TILE) and solve
// S x = y.  y is overwritten by the forward-substituted rhs, x receives the solution.  *ok (device
// double) is set to 1.0 first and to 0.0 when a pivot is not positive; *ws.stall (ok[1] when ws.stall is null) is set to 0.0
// first and to 1.0 when a cross-workgroup hand-off timed out.
void chol_factor_solve(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, hipStream_t st);

// Order n <= CHOL_TILE: the same solve as ONE launch of one workgroup (S with ld >= 16 ceil(n / 16), identity beyond n; Linv_ws >=
// chol_workspace_doubles(CHOL_TILE) doubles).  *ok as above, *stall = 0.
void chol_small_solve(const double* S, const double* y, double* x, int n, int ld, double* Linv_ws, double* ok, double* stall, hipStream_t st);

}  // namespace mage
