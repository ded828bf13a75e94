// chol_kernels.hip -- dense float64 Cholesky factorisation + solve of the reduced camera system on gfx950.
//
// Replaces g2o::LinearSolverDense::solve (Eigen::LDLT<MatrixXd>), which
// BundlerLib selects at Dependencies/BundlerLib/Source/BundlerLib.cpp:188-190.  The reduced camera
// matrix S is symmetric positive definite whenever the reference's LDLT reports isPositive(), so an
// unpivoted Cholesky S = L L^T gives the same solution up to rounding; a non-positive pivot is
// reported through *ok = 0 (the reference's "solve failed" branch).
//
// Layout: S is n_pad x n_pad, column-major, lower triangle referenced, n_pad a multiple of TILE.
// Right-looking tile algorithm, one panel of TILE columns per step k:
//     k_potrf_diag   : L_kk = chol(S_kk)                          (one workgroup, LDS resident)
//     k_trsm_panel   : L_ik = S_ik L_kk^-T for i > k, and the rhs row  y_k = L_kk^-1 y_k
//     k_syrk_update  : S_ij -= L_ik L_jk^T  (i >= j > k)  on v_mfma_f64_16x16x4_f64, and  y_i -= L_ik y_k
// so the forward substitution rides along with the factorisation (the rhs is treated as one more row
// of the matrix).  The backward substitution L^T x = y is k_bsolve_step, one launch per tile column.
//
// This is the only MFMA-bound stage of the path (n^3/3 = 72 GFLOP at 1000 poses): f64 MFMA peak on
// MI355X is 78.6 TFLOP/s (= the f64 vector peak; one 16x16x4 instruction = 2048 flop per 64 cycles/SIMD).
#include <hip/hip_runtime.h>
#include "chol_kernels.h"

namespace mage {
namespace {

constexpr int TILE = CHOL_TILE;      // 128
constexpr int LDT = TILE + 1;        // padded LDS leading dimension for row/column walks
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// diagonal tile: unblocked right-looking Cholesky in LDS.  Also exports a read-only copy of L_kk
// (row-major, dense TILE x TILE, zeros above the diagonal) and 1/diag for the panel solve.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ S, int ld, int k, double* __restrict__ Ld,
                                                    double* __restrict__ inv_diag, double* __restrict__ ok)
{
    extern __shared__ double A[];   // TILE x LDT, A[r * LDT + c]
    __shared__ int fail;
    const int tid = threadIdx.x;
    double* T = S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
    if (tid == 0) fail = 0;
    // coalesced load: column c of the tile is contiguous in rows
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int c = e / TILE, r = e % TILE;
        A[r * LDT + c] = (r >= c) ? T[(size_t)c * ld + r] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < TILE; ++j) {
        if (tid == 0) {
            const double d = A[j * LDT + j];
            if (!(d > 0.0)) fail = 1;
            A[j * LDT + j] = sqrt(d);
        }
        __syncthreads();
        const double inv = 1.0 / A[j * LDT + j];
        for (int r = j + 1 + tid; r < TILE; r += 256) A[r * LDT + j] *= inv;
        __syncthreads();
        // trailing rank-1 update of the lower triangle: (r, c), j < c <= r
        const int m = TILE - 1 - j;               // trailing order
        for (int e = tid; e < m * m; e += 256) {
            const int rr = e / m, cc = e % m;
            if (cc <= rr) {
                const int r = j + 1 + rr, c = j + 1 + cc;
                A[r * LDT + c] -= A[r * LDT + j] * A[c * LDT + j];
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int c = e / TILE, r = e % TILE;
        if (r >= c) T[(size_t)c * ld + r] = A[r * LDT + c];
    }
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int r = e / TILE, c = e % TILE;
        Ld[e] = A[r * LDT + c];
    }
    if (tid < TILE) inv_diag[tid] = 1.0 / A[tid * LDT + tid];
    if (tid == 0 && fail) *ok = 0.0;
}

// ---------------------------------------------------------------------------------------------
// panel solve X L_kk^T = A, one matrix row per lane (64 rows per workgroup); the row's solution is
// kept in LDS as xs[j][lane] (conflict-free), L_kk comes through the scalar cache from the
// read-only copy.  The last workgroup solves the rhs row (y_k) the same way.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_trsm_panel(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                   const double* __restrict__ Ld, const double* __restrict__ inv_diag)
{
    __shared__ double xs[TILE * 64];
    const int lane = threadIdx.x;
    const int n_row_blocks = (nt - k - 1) * (TILE / 64);
    const bool is_rhs = (int)blockIdx.x == n_row_blocks;
    double* rowbase;       // element j of this lane's row lives at rowbase[j * stride]
    size_t stride;
    bool active = true;
    if (!is_rhs) {
        const int row = (k + 1) * TILE + blockIdx.x * 64 + lane;
        rowbase = S + (size_t)(k * TILE) * ld + row;
        stride = (size_t)ld;
    } else {
        rowbase = y + (size_t)k * TILE;
        stride = 1;
        active = lane == 0;
    }
    for (int j = 0; j < TILE; ++j) {
        double a = active ? rowbase[(size_t)j * stride] : 0.0;
        const double* Lj = Ld + (size_t)j * TILE;
        double acc0 = 0, acc1 = 0;
        int c = 0;
        for (; c + 1 < j; c += 2) {
            acc0 += xs[c * 64 + lane] * Lj[c];
            acc1 += xs[(c + 1) * 64 + lane] * Lj[c + 1];
        }
        if (c < j) acc0 += xs[c * 64 + lane] * Lj[c];
        const double x = (a - (acc0 + acc1)) * inv_diag[j];
        xs[j * 64 + lane] = x;
        if (active) rowbase[(size_t)j * stride] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// trailing update on the f64 matrix cores.  One workgroup = one 128x128 tile of the lower triangle,
// 4 wavefronts as 2x2, each owning 64x64 = 4x4 MFMA tiles (128 accumulator VGPRs).  The MFMA "M"
// index runs over tile COLUMNS and "N" over tile ROWS so that the accumulator's lane&15 direction is
// the memory-contiguous one and the read-modify-write of S is done in 128-byte row segments.
// ---------------------------------------------------------------------------------------------
constexpr int KC = 32;               // panel columns staged per LDS round
constexpr int LDP = TILE + 16;       // LDS row pitch (doubles): +32 banks between consecutive k rows

__global__ __launch_bounds__(256) void k_syrk_update(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt)
{
    __shared__ double Ps[2][KC * LDP];   // [0]: panel rows of tile i (N side), [1]: panel rows of tile j (M side)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = nt - k - 1;
    const int n_tiles = m * (m + 1) / 2;
    if ((int)blockIdx.x >= n_tiles) {
        // rhs row: y_i -= L_ik y_k
        const int i = k + 1 + (blockIdx.x - n_tiles);
        double* red = &Ps[0][0];
        const int r = tid & 127, half = tid >> 7;
        const double* Lik = S + (size_t)(k * TILE) * ld + (size_t)i * TILE;
        const double* yk = y + (size_t)k * TILE;
        double acc = 0;
        for (int c = half * 64; c < half * 64 + 64; ++c) acc += Lik[(size_t)c * ld + r] * yk[c];
        red[tid] = acc;
        __syncthreads();
        if (tid < 128) y[(size_t)i * TILE + tid] -= red[tid] + red[tid + 128];
        return;
    }
    // linear tile index -> (row tile, col tile) of the trailing lower triangle
    int t = blockIdx.x;
    int rt = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((rt + 1) * (rt + 2) / 2 <= t) ++rt;
    while (rt * (rt + 1) / 2 > t) --rt;
    const int ct = t - rt * (rt + 1) / 2;
    const int ti = k + 1 + rt, tj = k + 1 + ct;
    const double* Pi = S + (size_t)(k * TILE) * ld + (size_t)ti * TILE;   // L_ik : rows of tile i, panel columns
    const double* Pj = S + (size_t)(k * TILE) * ld + (size_t)tj * TILE;

    const int wn = wave & 1, wm = wave >> 1;     // wave's 64-row (N) / 64-col (M) half
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){ 0, 0, 0, 0 };

    for (int kc = 0; kc < TILE; kc += KC) {
        // stage KC panel columns of both tiles: each column is 128 contiguous rows (1 KiB)
#pragma unroll
        for (int it = 0; it < (KC * TILE) / (256 * 2); ++it) {
            const int e = (it * 256 + tid) * 2;
            const int kk = e / TILE, r = e % TILE;
            const double2 vi = *reinterpret_cast<const double2*>(Pi + (size_t)(kc + kk) * ld + r);
            const double2 vj = *reinterpret_cast<const double2*>(Pj + (size_t)(kc + kk) * ld + r);
            *reinterpret_cast<double2*>(&Ps[0][kk * LDP + r]) = vi;
            *reinterpret_cast<double2*>(&Ps[1][kk * LDP + r]) = vj;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < KC; k4 += 4) {
            const int kk = k4 + (lane >> 4);
            double bn[4], am[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bn[q] = Ps[0][kk * LDP + wn * 64 + q * 16 + (lane & 15)];   // B[k][n] : row n of tile i
                am[q] = Ps[1][kk * LDP + wm * 64 + q * 16 + (lane & 15)];   // A[m][k] : row m of tile j
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[a], bn[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[m][n]: n = lane & 15 (matrix row, contiguous), m = (lane >> 4) + 4 * reg (matrix column)
    double* C = S + (size_t)(tj * TILE) * ld + (size_t)ti * TILE;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = wm * 64 + a * 16 + (lane >> 4) + 4 * r;
                const int row = wn * 64 + b * 16 + (lane & 15);
                C[(size_t)col * ld + row] -= acc[a][b][r];
            }
}

// ---------------------------------------------------------------------------------------------
// backward substitution, tile column k: every workgroup solves x_k = L_kk^-T y_k with one wavefront
// (two unknowns per lane, broadcasts through readlane, no barriers), then workgroup j < k applies
// y_j -= L_kj^T x_k; workgroup k stores x_k.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bsolve_step(const double* __restrict__ S, double* __restrict__ y, double* __restrict__ x,
                                                     int ld, int k)
{
    extern __shared__ double T[];     // TILE x LDT
    __shared__ double xk[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Lkk = S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int c = e / TILE, r = e % TILE;
        T[r * LDT + c] = (r >= c) ? Lkk[(size_t)c * ld + r] : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        double y0 = y[(size_t)k * TILE + lane], y1 = y[(size_t)k * TILE + 64 + lane];
        for (int c = TILE - 1; c >= 0; --c) {
            // x_c = y_c / L[c][c] ; y_j -= L[c][j] x_c for j < c
            const double yc = __shfl(c < 64 ? y0 : y1, c & 63, 64);
            const double xc = yc / T[c * LDT + c];
            if (lane == (c & 63)) { if (c < 64) y0 = xc; else y1 = xc; }
            if (lane < c) y0 -= T[c * LDT + lane] * xc;
            if (lane + 64 < c) y1 -= T[c * LDT + 64 + lane] * xc;
        }
        xk[lane] = y0; xk[64 + lane] = y1;
    }
    __syncthreads();
    const int j = blockIdx.x;
    if (j == k) {
        if (tid < TILE) x[(size_t)k * TILE + tid] = xk[tid];
        return;
    }
    // y_j[c] -= sum_r L(k-block row r, j-block col c) * x_k[r]
    const double* Lkj = S + (size_t)(j * TILE) * ld + (size_t)k * TILE;
    __syncthreads();
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int c = e / TILE, r = e % TILE;
        T[r * LDT + c] = Lkj[(size_t)c * ld + r];
    }
    __syncthreads();
    {
        const int c = tid & 127, half = tid >> 7;
        double acc = 0;
        for (int r = half * 64; r < half * 64 + 64; ++r) acc += T[r * LDT + c] * xk[r];
        __syncthreads();
        T[tid] = acc;
        __syncthreads();
        if (tid < 128) y[(size_t)j * TILE + tid] -= T[tid] + T[tid + 128];
    }
}

__global__ void k_set_scalar(double* p, double v) { *p = v; }

}  // namespace

void chol_factor_solve(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, hipStream_t st)
{
    const int nt = n_pad / TILE;
    const size_t lds_tile = (size_t)TILE * LDT * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bsolve_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, st, ok, 1.0);
    for (int k = 0; k < nt; ++k) {
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), lds_tile, st, S, n_pad, k, ws.Ld, ws.inv_diag, ok);
        const int n_row_blocks = (nt - k - 1) * (TILE / 64);
        hipLaunchKernelGGL(k_trsm_panel, dim3(n_row_blocks + 1), dim3(64), 0, st, S, y, n_pad, k, nt, ws.Ld, ws.inv_diag);
        const int m = nt - k - 1;
        if (m > 0) hipLaunchKernelGGL(k_syrk_update, dim3(m * (m + 1) / 2 + m), dim3(256), 0, st, S, y, n_pad, k, nt);
    }
    for (int k = nt - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_bsolve_step, dim3(k + 1), dim3(256), lds_tile, st, S, y, x, n_pad, k);
}

}  // namespace mage
