// chol_dag.h -- the dense Cholesky factorisation + forward substitution as ONE persistent launch over a static task list
// (chol_dag.hip); called by chol_factor_solve (chol_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "chol_kernels.h"

namespace mage {

// ints of CholWorkspace::sync (behind the first 8, which the column-by-column launches use) that the task-graph launch needs for nt tile columns
size_t chol_dag_sync_ints(int nt);
void chol_dag_init_device(int n_cu);       // once per device, from chol_init_device
void chol_dag_prefetch(int n_pad, const int* env_host = nullptr);         // a system of this order is coming: start building its task lists (a worker thread; returns at once)
bool chol_dag_wait_schedule(int n_pad, double* build_ms, const int* env_host = nullptr);      // blocks until they are on the device; false when this order is not the task graph's
// Queues the state reset + the launch on `st` and returns true; false when this size is served by the column-by-column launches
// (fewer than CHOL_DAG_MIN_TILES tile columns, more than 255, no schedule could be built -- or it is still being built: see chol_dag_prefetch).  The caller queues the backward solve.
// ws.env_host (optional): the skyline of S by tile rows -- tiles left of it are never touched; *kmax_dev then receives the device array the
// backward solve needs (per tile column the last tile row inside the skyline), nullptr for a dense system.
bool chol_dag_factor(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, double* stall, hipStream_t st, const int** kmax_dev = nullptr);

}  // namespace mage
