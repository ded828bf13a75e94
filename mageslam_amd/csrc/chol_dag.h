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
// Queues the state reset + the launch on `st` and returns true; false when this size is served by the column-by-column launches
// (fewer than CHOL_DAG_MIN_TILES tile columns, more than 255, or no schedule could be built).  The caller queues the backward solve.
bool chol_dag_factor(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, double* stall, hipStream_t st);

}  // namespace mage
