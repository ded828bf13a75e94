// chol_kernels.h -- dense f64 Cholesky factor + solve on gfx950 (see chol_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mage {

constexpr int CHOL_TILE = 128;

struct CholWorkspace {
    double* Ld;        // CHOL_TILE x CHOL_TILE  read-only copy of the current diagonal factor
    double* inv_diag;  // CHOL_TILE
};

// Factor S = L L^T in place (lower triangle, column-major, n_pad multiple of CHOL_TILE) and solve
// S x = y.  y is overwritten by the forward-substituted rhs, x receives the solution.  *ok (device
// double) must be 1.0 on entry and is set to 0.0 when a pivot is not positive.
void chol_factor_solve(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, hipStream_t st);

}  // namespace mage
