// ba_kernels.hip -- HIP kernels (gfx950) for the bundle-adjustment hot path.
//
// What each kernel computes, and the reference code it replaces (g2o is not vendored in the
// reference tree; semantics per SURVEY.md appendix A, call sites in
// Dependencies/BundlerLib/Source/BundlerLib.cpp):
//
//   k_error            EdgeProjectXYZ2UV::computeError + RobustKernelHuber (A.2, A.3)  -> chi2
//   k_linearize_lm     linearizeOplus + constructQuadraticForm, landmark side: V, b_p, W (A.2, A.3)
//   k_linearize_cam    same, camera side: U, b_c, one wavefront per camera, shuffle reduction
//   k_lm_invert        BlockSolver::solve, (V + lambda I)^-1 and D^-1 b_p (A.5)
//   k_schur_block      S_ij = [i==j](U_i + lambda I) - sum W_i D^-1 W_j^T, one wavefront per 6x6 block
//   k_schur_rhs        b_s = b_c - sum W D^-1 b_p, one wavefront per camera
//   k_backsub          x_l = D^-1 (b_p - sum W^T x_c), point update, scale term (A.4, A.5)
//   k_pose_update      VertexSE3Expmap::oplusImpl: pose <- exp(x_c) * pose (A.1)
//   k_classify         StepBundleAdjustment post-pass, BundlerLib.cpp:384-427
//
// All accumulations use a fixed order (per-lane strided partial sums, xor-butterfly across the
// 64 lanes, ordered combination of wave partials), so results are bit-reproducible run to run --
// the property the reference checks with mira::determinator (BundleAdjust.cpp:43-44,250,320,389).
// HBM-bound integer/f64 streaming work: no MFMA here (the dense factorisation is in chol_kernels.hip).
#include <algorithm>
#include <cstdlib>

#include "ba_kernels.h"
#include "chol_kernels.h"

namespace mage {
namespace {

constexpr int WAVE = 64;
constexpr int SPLIT_BLOCKS_BELOW = SCHUR_SPLIT_BLOCKS_BELOW;       // Schur blocks: at most this many -> four wavefronts per block
constexpr int SPLIT_CAMERAS_BELOW = 256;      // per-camera kernels: at most this many free cameras -> a workgroup per camera

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}
// The butterfly sums of N <= 32 values AT ONCE (round 4).  wave_sum spends six shuffles per value and leaves the total in every lane;
// here a stage's shuffle carries TWO values -- the lanes whose bit m is clear send their y and keep x, the others send x and keep y, so
// either half forms "own + received" of the value it keeps, the very additions of the plain butterfly -- and the register count
// halves from stage to stage: 29 shuffles instead of 162 for the 27 sums of a camera block (21 of U, 6 of b), the same bits.
// Returns the total of value number wave_sum_slot(lane) (meaningless where that number is >= N).
__device__ __forceinline__ int wave_sum_slot(int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}
template <int N, int M>
__device__ __forceinline__ double wave_sum_packed_stage(const double (&v)[N], int lane)
{
    if constexpr (M == 1) {
        static_assert(N == 1, "five halvings take at most 32 values to one");
        return v[0] + __shfl_xor(v[0], 1, WAVE);
    } else {
        constexpr int H = (N + 1) / 2;
        double h[H];
        const bool up = (lane & M) != 0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const double x = v[2 * j], y = (2 * j + 1 < N) ? v[2 * j + 1] : 0.0;
            const double r = __shfl_xor(up ? x : y, M, WAVE);
            h[j] = (up ? y : x) + r;
        }
        return wave_sum_packed_stage<H, M / 2>(h, lane);
    }
}
template <int N>
__device__ __forceinline__ double wave_sum_packed(const double (&v)[N], int lane)
{
    static_assert(N >= 1 && N <= 32, "at most 32 values");
    return wave_sum_packed_stage<N, 32>(v, lane);
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, WAVE));
    return v;
}

// Deterministic block sum: butterfly inside each wave, then wave partials added in wave order.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* sm /* NW doubles */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += sm[i];
    return r;
}

// Camera handled by wavefront `wave` of workgroup `bid` when workgroups of 4 wavefronts cover n cameras: workgroups go to the
// eight XCDs round-robin, so XCD x = bid % 8 gets the contiguous run of cameras [x * run, (x + 1) * run) -- neighbouring
// cameras observe the same landmarks and then meet in one L2.  Returns n (= nothing to do) past the end of a run.
__device__ __forceinline__ int xcd_camera(int bid, int wave, int n)
{
    const int run = ((n + 7) / 8 + 3) / 4 * 4;                 // cameras per XCD, a multiple of the 4 per workgroup
    const int within = (bid / 8) * 4 + wave;
    const int hc = (bid % 8) * run + within;
    return within < run && hc < n ? hc : n;
}
__host__ inline int xcd_camera_grid(int n) { return (((n + 7) / 8 + 3) / 4) * 8; }

struct PoseD { double qx, qy, qz, qw, tx, ty, tz; };

// a value that is the same in every lane, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ PoseD load_pose(const double* __restrict__ p, int cam)
{
    const double2* q = reinterpret_cast<const double2*>(p + (size_t)cam * 8);
    double2 a = q[0], b = q[1], c = q[2], d = q[3];
    PoseD P = { a.x, a.y, b.x, b.y, c.x, c.y, d.x };
    return P;
}
__device__ __forceinline__ void store_pose(double* __restrict__ p, int cam, const PoseD& P)
{
    double2* q = reinterpret_cast<double2*>(p + (size_t)cam * 8);
    q[0] = make_double2(P.qx, P.qy); q[1] = make_double2(P.qz, P.qw);
    q[2] = make_double2(P.tx, P.ty); q[3] = make_double2(P.tz, 0.0);
}

// v' = q v q^-1  (Eigen QuaternionBase::_transformVector form)
__device__ __forceinline__ void q_rot(double qx, double qy, double qz, double qw, double vx, double vy, double vz,
                                      double& ox, double& oy, double& oz)
{
    double ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
    ux += ux; uy += uy; uz += uz;
    ox = vx + qw * ux + (qy * uz - qz * uy);
    oy = vy + qw * uy + (qz * ux - qx * uz);
    oz = vz + qw * uz + (qx * uy - qy * ux);
}

__device__ __forceinline__ void q_to_R(double qx, double qy, double qz, double qw, double R[9])
{
    double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__device__ __forceinline__ void huber(double chi2, double delta, double& rho0, double& rho1)
{
    double d2 = delta * delta;
    if (chi2 <= d2) { rho0 = chi2; rho1 = 1.0; }
    else { double s = sqrt(chi2); rho0 = 2 * s * delta - d2; rho1 = delta / s; }
}

struct EdgeGeom { double x, y, z, e0, e1; };

__device__ __forceinline__ EdgeGeom edge_geom(const PoseD& P, const double* __restrict__ camK, int cam,
                                               double X, double Y, double Z, float2 uv)
{
    EdgeGeom g;
    q_rot(P.qx, P.qy, P.qz, P.qw, X, Y, Z, g.x, g.y, g.z);
    g.x += P.tx; g.y += P.ty; g.z += P.tz;
    const double f = camK[cam * 4 + 0], cx = camK[cam * 4 + 1], cy = camK[cam * 4 + 2];
    g.e0 = (double)uv.x - (g.x / g.z * f + cx);
    g.e1 = (double)uv.y - (g.y / g.z * f + cy);
    return g;
}

// 2x6 pose Jacobian (columns omega then upsilon), appendix A.2
__device__ __forceinline__ void jac_pose(const EdgeGeom& g, double f, double Jc[12])
{
    const double x = g.x, y = g.y, z = g.z, z2 = z * z;
    Jc[0] = x * y / z2 * f;        Jc[1] = -(1 + (x * x / z2)) * f; Jc[2] = y / z * f;
    Jc[3] = -1.0 / z * f;          Jc[4] = 0;                       Jc[5] = x / z2 * f;
    Jc[6] = (1 + y * y / z2) * f;  Jc[7] = -x * y / z2 * f;         Jc[8] = -x / z * f;
    Jc[9] = 0;                     Jc[10] = -1.0 / z * f;           Jc[11] = y / z2 * f;
}
// 2x3 point Jacobian: -1/z * [[f,0,-f x/z],[0,f,-f y/z]] * R
__device__ __forceinline__ void jac_point(const EdgeGeom& g, double f, const double R[9], double Jp[6])
{
    const double t02 = -g.x / g.z * f, t12 = -g.y / g.z * f, iz = -1.0 / g.z;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Jp[c] = iz * (f * R[c] + t02 * R[6 + c]);
        Jp[3 + c] = iz * (f * R[3 + c] + t12 * R[6 + c]);
    }
}

// ---------------------------------------------------------------------------------------------
// final stage of the two-level reductions: one block, fixed strided order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_sum(const double* __restrict__ in, int n, int stride_out,
                                                    double* __restrict__ out, int n_out)
{
    // in is laid out [n_out][n] ; out[o * stride_out].  One block per output when the grid has several (the sums are independent; the same
    // additions in the same order either way), all outputs in turn when it is one block.
    __shared__ double sm[4];
    for (int o = (int)blockIdx.x; o < n_out; o += (int)gridDim.x) {
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += 256) acc += in[(size_t)o * n + i];
        double r = block_sum<4>(acc, sm);
        if (threadIdx.x == 0) out[o * stride_out] = r;
    }
}

// k_reduce_sum's body for one block: in is [n_out][n], out[o * stride_out]
__device__ __forceinline__ void fold_partials_strided(const double* __restrict__ in, int n, int stride_out, double* __restrict__ out, int n_out, double* sm4)
{
    for (int o = 0; o < n_out; ++o) {
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += 256) acc += in[(size_t)o * n + i];
        const double r = block_sum<4>(acc, sm4);
        if (threadIdx.x == 0) out[o * stride_out] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// residual + robustified chi2  (one thread per observation, landmark order => coalesced records)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_error(BaDeviceView v, int trial, double delta, int n_part)
{
    __shared__ double sm[4];
    const double* pose = trial ? v.pose_trial : v.pose_cur;
    const double* pts = trial ? v.pt_trial : v.pt_cur;
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_L; i += gridDim.x * 256) {
        if (!v.L_active[i]) continue;                      // removed observation (outlier): not part of the graph any more
        const int cam = v.L_cam[i], pt = v.L_pt[i];
        PoseD P = load_pose(pose, cam);
        const double2 xy = *reinterpret_cast<const double2*>(pts + (size_t)pt * 4);
        const double Z = pts[(size_t)pt * 4 + 2];
        EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
        *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
        double rho0, rho1;
        huber((double)v.L_info[i] * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
        acc += rho0;
    }
    double r = block_sum<4>(acc, sm);
    if (threadIdx.x == 0) v.partial[blockIdx.x] = r;
    (void)n_part;
}

// ---------------------------------------------------------------------------------------------
// landmark side of the linearisation: one thread per landmark, observations in order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linearize_lm(BaDeviceView v, double delta)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= v.n_lm) return;
    const int pt = v.lm_pt[l];
    const double X = v.pt_cur[(size_t)pt * 4], Y = v.pt_cur[(size_t)pt * 4 + 1], Z = v.pt_cur[(size_t)pt * 4 + 2];
    double V[6] = { 0, 0, 0, 0, 0, 0 }, bp[3] = { 0, 0, 0 };
    double Wacc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) Wacc[k] = 0;
    const int beg = v.lm_ptr[l], end = v.lm_ptr[l + 1];
    int cur_slot = -1;
    for (int i = beg; i < end; ++i) {
        const int slot = v.L_slot[i];
        if (slot != cur_slot) {
            if (cur_slot >= 0) {
                double2* Wd = reinterpret_cast<double2*>(v.W + (size_t)cur_slot * 18);        // 144-byte block, 16-byte aligned: nine 128-bit stores
#pragma unroll
                for (int k = 0; k < 9; ++k) { Wd[k] = make_double2(Wacc[2 * k], Wacc[2 * k + 1]); Wacc[2 * k] = 0; Wacc[2 * k + 1] = 0; }
            }
            cur_slot = slot;
        }
        if (!v.L_active[i]) continue;                      // removed observation: contributes nothing (its slot may end up all-zero)
        const int cam = v.L_cam[i];
        PoseD P = load_pose(v.pose_cur, cam);
        EdgeGeom g = edge_geom(P, v.camK, cam, X, Y, Z, v.L_uv[i]);
        const double f = v.camK[cam * 4];
        const double info = (double)v.L_info[i];
        double rho0, rho1;
        huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
        const double w = info * rho1;
        const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
        double R[9], Jp[6];
        q_to_R(P.qx, P.qy, P.qz, P.qw, R);
        jac_point(g, f, R, Jp);
#pragma unroll
        for (int a = 0; a < 3; ++a) bp[a] += Jp[a] * r0 + Jp[3 + a] * r1;
        V[0] += Jp[0] * w * Jp[0] + Jp[3] * w * Jp[3];
        V[1] += Jp[0] * w * Jp[1] + Jp[3] * w * Jp[4];
        V[2] += Jp[0] * w * Jp[2] + Jp[3] * w * Jp[5];
        V[3] += Jp[1] * w * Jp[1] + Jp[4] * w * Jp[4];
        V[4] += Jp[1] * w * Jp[2] + Jp[4] * w * Jp[5];
        V[5] += Jp[2] * w * Jp[2] + Jp[5] * w * Jp[5];
        if (slot >= 0) {
            double Jc[12];
            jac_pose(g, f, Jc);
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) Wacc[a * 3 + b] += Jc[a] * w * Jp[b] + Jc[6 + a] * w * Jp[3 + b];
        }
    }
    if (cur_slot >= 0) {
        double2* Wd = reinterpret_cast<double2*>(v.W + (size_t)cur_slot * 18);
#pragma unroll
        for (int k = 0; k < 9; ++k) Wd[k] = make_double2(Wacc[2 * k], Wacc[2 * k + 1]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) v.V[(size_t)l * 6 + k] = V[k];
    v.bp[(size_t)l * 4 + 0] = bp[0]; v.bp[(size_t)l * 4 + 1] = bp[1]; v.bp[(size_t)l * 4 + 2] = bp[2]; v.bp[(size_t)l * 4 + 3] = 0;
}

// ---------------------------------------------------------------------------------------------
// camera side: one wavefront per free camera, lanes stride over its observations
// ---------------------------------------------------------------------------------------------
// SPLIT = false: one wavefront per camera (four cameras per workgroup, XCD runs).  SPLIT = true: the four wavefronts of a
// workgroup share ONE camera (observations dealt out 64 at a time), partial sums combined in wavefront order through LDS --
// for problems with few cameras (local BA: ~15 free keyframes x 2 500 observations) where one wavefront per camera leaves
// the GPU empty and walks 40 dependent gathers.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_linearize_cam(BaDeviceView v, double delta)
{
    const int wave = threadIdx.x >> 6;
    const int hc = SPLIT ? (int)blockIdx.x : xcd_camera((int)blockIdx.x, wave, v.n_fc);
    const int lane = threadIdx.x & 63;
    if (hc >= v.n_fc) return;
    const int first = SPLIT ? wave * WAVE + lane : lane, stride = SPLIT ? 4 * WAVE : WAVE;
    const int cam = v.hc2cam[hc];
    PoseD P = load_pose(v.pose_cur, cam);
    const double f = v.camK[cam * 4];
    double A[21], b[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) A[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = 0;
    for (int idx = v.camE_ptr[hc] + first; idx < v.camE_ptr[hc + 1]; idx += stride) {
        const int i = v.camE[idx];
        if (!v.L_active[i]) continue;
        const int pt = v.L_pt[i];
        const double2 xy = *reinterpret_cast<const double2*>(v.pt_cur + (size_t)pt * 4);
        const double Z = v.pt_cur[(size_t)pt * 4 + 2];
        EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
        const double info = (double)v.L_info[i];
        double rho0, rho1;
        huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
        const double w = info * rho1;
        const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
        double Jc[12];
        jac_pose(g, f, Jc);
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            b[a] += Jc[a] * r0 + Jc[6 + a] * r1;
#pragma unroll
            for (int c = 0; c <= a; ++c) A[k++] += Jc[a] * w * Jc[c] + Jc[6 + a] * w * Jc[6 + c];
        }
    }
    // the 27 sums of the block in one packed butterfly (wave_sum_packed: 29 shuffles instead of 162, the same bits); value k's total
    // sits in the even lanes whose wave_sum_slot is k
    double s27[27];
#pragma unroll
    for (int k = 0; k < 21; ++k) s27[k] = A[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) s27[21 + k] = b[k];
    double tot = wave_sum_packed<27>(s27, lane);
    const int slot = wave_sum_slot(lane);
    if (SPLIT) {
        __shared__ double part[4][28];
        if ((lane & 1) == 0 && slot < 27) part[wave][slot] = tot;
        __syncthreads();
        if (wave != 0) return;
        if (slot < 27) tot = ((part[0][slot] + part[1][slot]) + part[2][slot]) + part[3][slot];
    }
    if ((lane & 1) == 0 && slot < 27) {
        if (slot < 21) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= slot) ++a;              // slot = a (a + 1) / 2 + c
            const int c = slot - a * (a + 1) / 2;
            v.U[(size_t)hc * 36 + a * 6 + c] = tot; v.U[(size_t)hc * 36 + c * 6 + a] = tot;
        } else v.bc[(size_t)hc * 6 + (slot - 21)] = tot;
    }
}

// max |diagonal| over pose and landmark blocks (computeLambdaInit, A.4): grid-stride partial maxima, then one block folds them
// (a single block over 100 k landmarks took 345 us -- and lambda is re-seeded after every outlier removal)
// udiag: the diagonal of U as 6 doubles per free camera when it was summed over the ranks of a landmark-sharded map (null: read U)
__global__ __launch_bounds__(256) void k_maxdiag(BaDeviceView v, const double* __restrict__ udiag)
{
    __shared__ double sm[4];
    double m = 0;
    const int nU = v.n_fc * 6, nV = v.points_free ? v.n_lm * 3 : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nU; i += gridDim.x * 256) m = fmax(m, fabs(udiag ? udiag[i] : v.U[(size_t)(i / 6) * 36 + (i % 6) * 7]));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nV; i += gridDim.x * 256) {
        const int l = i / 3, d = i % 3;
        m = fmax(m, fabs(v.V[(size_t)l * 6 + (d == 0 ? 0 : d == 1 ? 3 : 5)]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) v.partial[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void k_reduce_max(const double* __restrict__ in, int n, double* __restrict__ out)
{
    __shared__ double sm[4];
    double m = 0;
    for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, in[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// ---------------------------------------------------------------------------------------------
// Schur complement
// ---------------------------------------------------------------------------------------------
// (V + lambda I)^-1 in the cofactor form Eigen uses for fixed 3x3 (the inverse of a symmetric matrix
// computed that way is bitwise symmetric, so 6 values are stored), and D^-1 b_p.
// Blocks [0, nb_lm): (V_l + lambda I)^-1 and D^-1 b_p per landmark.  The blocks behind them prepare the reduced system's small
// arrays in the same launch (they were a memset, a copy and a kernel of their own, ~4 us each on the stream): y = b_c (the diagonal
// blocks of k_schur_block subtract the landmark part; y_from_bc = 0: plain zero) with a zero tail, and the identity on the padded
// tail of S's diagonal so that the padded system stays SPD.
// pad_diag: 1 (0 on the ranks of a landmark-sharded map that do not add the camera damping: the shards' matrices are summed).
__global__ __launch_bounds__(256) void k_lm_invert(BaDeviceView v, double lambda, int nb_lm, int y_from_bc, double pad_diag)
{
    if ((int)blockIdx.x >= nb_lm) {
        const int n = v.n_fc * 6;
        for (int i = ((int)blockIdx.x - nb_lm) * 256 + threadIdx.x; i < v.n_pad; i += ((int)gridDim.x - nb_lm) * 256) {
            v.y[i] = (i < n && y_from_bc) ? v.bc[i] : 0.0;
            if (i >= n) v.S[(size_t)i * v.n_pad + i] = pad_diag;
        }
        return;
    }
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= v.n_lm) return;
    const double* Vl = v.V + (size_t)l * 6;
    const double m00 = Vl[0] + lambda, m01 = Vl[1], m02 = Vl[2], m11 = Vl[3] + lambda, m12 = Vl[4], m22 = Vl[5] + lambda;
    const double c00 = m11 * m22 - m12 * m12;
    const double c10 = m12 * m02 - m22 * m01;     // cofactor(1,0)
    const double c20 = m01 * m12 - m02 * m11;     // cofactor(2,0)
    const double det = c00 * m00 + c10 * m01 + c20 * m02;
    const double id = 1.0 / det;
    const double i00 = c00 * id, i01 = c10 * id, i02 = c20 * id;
    const double i11 = (m22 * m00 - m02 * m02) * id;
    const double i12 = (m02 * m01 - m00 * m12) * id;
    const double i22 = (m00 * m11 - m01 * m01) * id;
    double* D = v.Dinv + (size_t)l * 6;
    D[0] = i00; D[1] = i01; D[2] = i02; D[3] = i11; D[4] = i12; D[5] = i22;
    const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
    double* db = v.db + (size_t)l * 4;
    db[0] = i00 * b0 + i01 * b1 + i02 * b2;
    db[1] = i01 * b0 + i11 * b1 + i12 * b2;
    db[2] = i02 * b0 + i12 * b1 + i22 * b2;
    db[3] = 0;
}

// Zero-fill of the part of S the factorisation reads: per column, from the first row of its 128-row tile downwards (the strict
// upper tiles are never referenced).  Half the bytes of a memset of the whole matrix (145 MB instead of 289 MB at 1k poses).
__global__ __launch_bounds__(256) void k_zero_lower(double* __restrict__ S, int n_pad, int tile)
{
    const int c = blockIdx.y;
    const int r0 = (c / tile) * tile;
    double2* col = reinterpret_cast<double2*>(S + (size_t)c * n_pad + r0);
    const int n2 = (n_pad - r0) / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) col[i] = make_double2(0.0, 0.0);
}

// The same zero-fill restricted to the SKYLINE of S (round 4).  Row i of a Cholesky factor is zero left of the first non-zero of row
// i of the matrix, and the dense factorisation reproduces those zeros exactly (0 - 0 x = 0, 0 x = 0 while no NaN is about), so a tile
// left of its tile row's envelope that held zeros before a factorisation holds zeros after it: only the tiles inside the envelope --
// the Schur blocks, the tether pairs and the fill-in between them -- carry the last factor and need clearing.  The host hands
// tile_env over only while that invariant is known to hold (first trial after a build, a failed or stalled factorisation: full clear).
// On the 1k-pose rail 94 of 1 128 tiles: 19 -> 2 us per trial.  One workgroup per lower tile (R, t): the zero role of k_schur_prepare.
__device__ __forceinline__ void lm_dinv_from(const double* Vl, double lambda, double D[6]);
// k_lm_invert + the skyline zero-fill + the linearisation's chi2 fold in ONE launch (round 4, late: three launches of ~6 us each became one):
//   blocks [0, nb_lm)                      (V_l + lambda I)^-1 and D^-1 b_p per landmark            (k_lm_invert)
//   blocks [nb_lm, nb_lm + nb_y)           y = b_c with a zero tail (+ the padded diagonal when no zero role runs here)
//   block   nb_lm + nb_y, if fold_n > 0    scal[SC_CHI] = sum of the fold_n partials the linearisation left (k_reduce_sum's order)
//   the n_zero blocks behind               one lower tile of S each: cleared inside the skyline (k_zero_skyline); a diagonal tile then also
//                                          gets the identity on its padded rows -- by the workgroup that cleared it, behind a barrier
__global__ __launch_bounds__(256) void k_schur_prepare(BaDeviceView v, double lambda, int nb_lm, int nb_y, int y_from_bc, double pad_diag, int fold_n, int n_zero, int tile)
{
    __shared__ double sm4[4];
    const int bid = blockIdx.x;
    if (bid < nb_lm) {
        const int l = bid * 256 + threadIdx.x;
        if (l >= v.n_lm) return;
        double D[6];
        lm_dinv_from(v.V + (size_t)l * 6, lambda, D);
        double* Do = v.Dinv + (size_t)l * 6;
        Do[0] = D[0]; Do[1] = D[1]; Do[2] = D[2]; Do[3] = D[3]; Do[4] = D[4]; Do[5] = D[5];
        const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
        double* db = v.db + (size_t)l * 4;
        db[0] = D[0] * b0 + D[1] * b1 + D[2] * b2;
        db[1] = D[1] * b0 + D[3] * b1 + D[4] * b2;
        db[2] = D[2] * b0 + D[4] * b1 + D[5] * b2;
        db[3] = 0;
        return;
    }
    const int n = v.n_fc * 6;
    if (bid < nb_lm + nb_y) {
        for (int i = (bid - nb_lm) * 256 + threadIdx.x; i < v.n_pad; i += nb_y * 256) {
            v.y[i] = (i < n && y_from_bc) ? v.bc[i] : 0.0;
            if (i >= n && n_zero == 0) v.S[(size_t)i * v.n_pad + i] = pad_diag;
        }
        return;
    }
    int z = bid - nb_lm - nb_y;
    if (fold_n > 0) {
        if (z == 0) { fold_partials_strided(v.partial, fold_n, 1, v.scal + SC_CHI, 1, sm4); return; }
        --z;
    }
    if (z >= n_zero) return;
    int R = (int)((sqrt(8.0 * (double)z + 1.0) - 1.0) * 0.5);
    while ((R + 1) * (R + 2) / 2 <= z) ++R;
    while (R * (R + 1) / 2 > z) --R;
    const int t = z - R * (R + 1) / 2;
    if (t < v.tile_env[R]) return;
    double2* base = reinterpret_cast<double2*>(v.S + (size_t)(t * tile) * v.n_pad + (size_t)R * tile);
    for (int e = threadIdx.x; e < tile * tile / 2; e += 256) {
        const int c = e / (tile / 2), r2 = e % (tile / 2);
        base[(size_t)c * (v.n_pad / 2) + r2] = make_double2(0.0, 0.0);
    }
    if (t == R && (R + 1) * tile > n) {            // the diagonal tile that holds padded rows
        __syncthreads();
        for (int i = R * tile + threadIdx.x; i < (R + 1) * tile; i += 256) if (i >= n) v.S[(size_t)i * v.n_pad + i] = pad_diag;
    }
}

// env[R] = min over the blocks (i, j), i <= j, whose rows 6 j .. fall into tile row R, of the tile column of 6 i
__global__ __launch_bounds__(256) void k_tile_envelope(BaDeviceView v, int* __restrict__ env, int n_tiles, int tile)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < n_tiles) atomicMin(env + g, g);                                  // (env arrives filled with INT_MAX: never right of the diagonal)
    if (g < v.n_blk) {
        const int2 ij = v.blk_ij[g];
        for (int r = ij.y * 6; r < ij.y * 6 + 6; r += 5) atomicMin(env + r / tile, (ij.x * 6) / tile);      // first and last row of the block
    }
    if (g < v.n_tp) {
        const int2 ij = v.tp_ij[g];
        for (int r = ij.y * 6; r < ij.y * 6 + 6; r += 5) atomicMin(env + r / tile, (ij.x * 6) / tile);
    }
}

// Landmark-sharded maps: what the ranks add up is the part of S the factorisation reads (same trapezoid as k_zero_lower), packed
// column after column, with the right-hand side behind it.  Column c of tile column t starts at 128 (t n_pad - 128 t (t - 1) / 2)
// + (c % 128)(n_pad - 128 t).  TO_PACKED = false copies back.
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void k_pack_lower(double* __restrict__ S, double* __restrict__ y, double* __restrict__ packed, int n_pad, int tile)
{
    const int c = blockIdx.y;
    if (c == n_pad) {                                       // the extra "column": y
        double* py = packed + (size_t)n_pad * (n_pad + tile) / 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n_pad; i += gridDim.x * 256) { if (TO_PACKED) py[i] = y[i]; else y[i] = py[i]; }
        return;
    }
    const int t = c / tile, r0 = t * tile;
    const size_t off = (size_t)tile * ((size_t)t * n_pad - (size_t)tile * t * (t - 1) / 2) + (size_t)(c - r0) * (n_pad - r0);
    double2* col = reinterpret_cast<double2*>(S + (size_t)c * n_pad + r0);
    double2* pk = reinterpret_cast<double2*>(packed + off);
    const int n2 = (n_pad - r0) / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) { if (TO_PACKED) pk[i] = col[i]; else col[i] = pk[i]; }
}

__global__ __launch_bounds__(256) void k_gather_udiag(BaDeviceView v, double* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < v.n_fc * 6) out[i] = v.U[(size_t)(i / 6) * 36 + (i % 6) * 7];
}

// mage_device_allreduce_local: every buffer becomes the sum in rank order (or the maximum) of all of them
constexpr int LOCAL_REDUCE_MAX = 16;
struct LocalReduceArgs { double* buf[LOCAL_REDUCE_MAX]; int n; };
__global__ __launch_bounds__(256) void k_allreduce_local(LocalReduceArgs a, size_t count, int op)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        double r = a.buf[0][i];
        for (int k = 1; k < a.n; ++k) { const double x = a.buf[k][i]; r = op == 0 ? r + x : fmax(r, x); }
        for (int k = 0; k < a.n; ++k) a.buf[k][i] = r;
    }
}

// One wavefront per non-empty upper block (i <= j).  Workgroups go to the eight XCDs round-robin and every XCD has its own
// L2, so the slot -> block table (blk_order, ba_host.hip) hands the workgroups of XCD x a CONTIGUOUS run of block rows (runs cut
// at equal shares of the contributions): camera i's W blocks are then fetched into ONE L2 instead of eight, and consecutive rows
// of an XCD share most of their column cameras.  Placement only: any table is correct.  Each lane owns a strided subset of the block's
// landmark contributions, forms (W_a D^-1) W_b^T in registers, then the 36 partial sums are combined
// with a butterfly.  The block is written to the lower triangle of S (column-major), i.e. as the
// transposed (j, i) block, plus the full diagonal block.
// SPLIT = false: one wavefront (= one workgroup) per block.  SPLIT = true (few blocks, each with thousands of contributions:
// local BA): four wavefronts share a block, contributions dealt out 64 at a time, the four partial blocks added in wavefront order.
// STAGED: the W blocks of a wavefront's 64 contributions are fetched COOPERATIVELY -- nine consecutive lanes read the nine 16-byte
// pieces of one 144-byte block, so a wave-wide load touches ~14 cache lines instead of 64 -- into the wavefront's LDS area in
// piece-major order, and every lane then picks up its own two blocks from LDS without bank conflicts.  The texture addresser
// handles about one cache line per cycle: lane-per-contribution gathers are 18 loads x 64 lines per 64 contributions, the
// cooperative form 18 x ~14.  Measured at the 1k-pose map: 258 -> 246 us (the kernel is bound by the latency of its misses at
// 8 wavefronts per compute unit, not by the addresser).  Same arithmetic in the same order: the results are bit-identical
// (tools/ab_schur.py; MAGE_BA_SCHUR_GATHER=1 selects the old loads).
template <bool SPLIT, bool STAGED>
__global__ __launch_bounds__(SPLIT ? 256 : 64 * SCHUR_WAVES) void k_schur_block(BaDeviceView v, double lambda)
{
    constexpr int NW = SPLIT ? 4 : SCHUR_WAVES;
    __shared__ __attribute__((aligned(16))) double red[NW][64 * 37];      // 18944 bytes per wavefront = 128 blocks x 144 + 128 slot indices x 4
    __shared__ double part[SPLIT ? 4 : 1][36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b;
    if (SPLIT) {
        b = (int)blockIdx.x;
        if (b >= v.n_blk) return;
    } else {
        const int slot = blockIdx.x * SCHUR_WAVES + wave;
        if (slot >= v.n_blk_slots) return;
        b = v.blk_order[slot];
        if (b < 0) return;
    }
    const int stride = SPLIT ? 4 * WAVE : WAVE;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0;
    // A diagonal block's contributions are the camera's own slots (a, a): the same pass over W also gives the camera's part of the
    // reduced right-hand side, b_s = b_c - sum W D^-1 b_p (k_schur_rhs read every W block a second time for it)
    const int2 ij = v.blk_ij[b];
    const bool diag = ij.x == ij.y;
    double yv[6] = { 0, 0, 0, 0, 0, 0 };
    const int c_end = v.blk_ptr[b + 1];
    double2* stg = reinterpret_cast<double2*>(red[wave]);                 // [piece 0..8][block 0..127], then 128 slot indices
    int* sl = reinterpret_cast<int*>(red[wave]) + 128 * 36;
    for (int c0 = v.blk_ptr[b] + (SPLIT ? wave * WAVE : 0); c0 < c_end; c0 += stride) {      // uniform per wavefront
        const int c = c0 + lane;
        const bool live = c < c_end;
        const int2 sab = live ? v.con[c] : make_int2(0, 0);               // idle lanes read slot 0 (valid memory) and add nothing
        // 144-byte W blocks and 48-byte D^-1 records are 16-byte aligned: 128-bit loads (21 per contribution instead of 42)
        const int lm = v.w_lm[sab.x];
        const double2* D2 = reinterpret_cast<const double2*>(v.Dinv + (size_t)lm * 6);
        const double2 da = D2[0], db = D2[1], dc = D2[2];
        const double d00 = da.x, d01 = da.y, d02 = db.x, d11 = db.y, d12 = dc.x, d22 = dc.y;
        double wa[18], wb[18];
        if (STAGED) {
            sl[lane] = sab.x; sl[64 + lane] = sab.y;
            double2 piece[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int q = i * 64 + lane, blk = (q * 7282) >> 16, part = q - 9 * blk;          // q / 9 for q < 1152
                piece[i] = reinterpret_cast<const double2*>(v.W + (size_t)sl[blk] * 18)[part];
            }
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int q = i * 64 + lane, blk = (q * 7282) >> 16, part = q - 9 * blk;
                stg[part * 128 + blk] = piece[i];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = stg[k * 128 + lane]; wa[2 * k] = t.x; wa[2 * k + 1] = t.y; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = stg[k * 128 + 64 + lane]; wb[2 * k] = t.x; wb[2 * k + 1] = t.y; }
        } else {
            const double2* Wa2 = reinterpret_cast<const double2*>(v.W + (size_t)sab.x * 18);
            const double2* Wb2 = reinterpret_cast<const double2*>(v.W + (size_t)sab.y * 18);
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = Wa2[k]; wa[2 * k] = t.x; wa[2 * k + 1] = t.y; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = Wb2[k]; wb[2 * k] = t.x; wb[2 * k + 1] = t.y; }
        }
        if (!live) continue;
        if (diag) {
            const double2* db2 = reinterpret_cast<const double2*>(v.db + (size_t)lm * 4);
            const double2 dba = db2[0];
            const double e0 = dba.x, e1 = dba.y, e2 = db2[1].x;
#pragma unroll
            for (int r = 0; r < 6; ++r) yv[r] += wa[r * 3] * e0 + wa[r * 3 + 1] * e1 + wa[r * 3 + 2] * e2;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double a0 = wa[r * 3], a1 = wa[r * 3 + 1], a2 = wa[r * 3 + 2];
            const double t0 = a0 * d00 + a1 * d01 + a2 * d02;
            const double t1 = a0 * d01 + a1 * d11 + a2 * d12;
            const double t2 = a0 * d02 + a1 * d12 + a2 * d22;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) acc[r * 6 + cc] += t0 * wb[cc * 3] + t1 * wb[cc * 3 + 1] + t2 * wb[cc * 3 + 2];
        }
    }
    // 36 sums over 64 lanes: a butterfly costs 36 x 6 cross-lane exchanges; instead every lane parks its 36 partials in LDS
    // (lane-major, odd pitch: conflict-free both ways) and lane k < 36 adds the 64 partials of entry k in lane order.
    double* R = red[wave];
#pragma unroll
    for (int k = 0; k < 36; ++k) R[lane * 37 + k] = acc[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane k < 36 writes entry (r, c) of the block
    double val = 0;
    if (lane < 36) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int j = 0; j < 64; j += 4) {
            s0 += R[(j + 0) * 37 + lane]; s1 += R[(j + 1) * 37 + lane]; s2 += R[(j + 2) * 37 + lane]; s3 += R[(j + 3) * 37 + lane];
        }
        val = (s0 + s1) + (s2 + s3);
    }
    double yval = 0;
    if (diag) {                                           // uniform per workgroup
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 6; ++k) R[lane * 7 + k] = yv[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 6) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) { s0 += R[(j + 0) * 7 + lane]; s1 += R[(j + 1) * 7 + lane]; s2 += R[(j + 2) * 7 + lane]; s3 += R[(j + 3) * 7 + lane]; }
            yval = (s0 + s1) + (s2 + s3);
        }
    }
    if (SPLIT) {
        __shared__ double ypart[4][6];
        if (lane < 36) part[wave][lane] = val;
        if (diag && lane < 6) ypart[wave][lane] = yval;
        __syncthreads();
        if (wave != 0) return;
        if (lane < 36) val = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        if (diag && lane < 6) yval = ((ypart[0][lane] + ypart[1][lane]) + ypart[2][lane]) + ypart[3][lane];
    }
    if (diag && lane < 6) v.y[ij.x * 6 + lane] = v.bc[(size_t)ij.x * 6 + lane] - yval;
    if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        if (ij.x == ij.y) {
            double u = v.U[(size_t)ij.x * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
            v.S[(size_t)(ij.x * 6 + c) * v.n_pad + (ij.x * 6 + r)] = u - val;
        } else {
            // S(row = 6j + c, col = 6i + r) = block(i,j)(r,c)
            v.S[(size_t)(ij.x * 6 + r) * v.n_pad + (ij.y * 6 + c)] = -val;
        }
    }
}

// COMPACT form of the W blocks (BaDeviceView::compact): a slot holds (a, b, iz, w) = x/z, y/z, 1/z of the point in the slot's camera
// and the observation's robust weight.  With the camera's focal length f and rotation R (camR, uniform per block):
//     Jc  = f [ a b, -(1 + a^2), b, -iz, 0, a iz ;  1 + b^2, -a b, -a, 0, -iz, b iz ]         (2 x 6, jac_pose)
//     Q   = w Jp = -w iz f [ R_0 - a R_2 ; R_1 - b R_2 ]                                        (2 x 3, jac_point; R_k = row k of R)
//     W   = Jc^T Q                                                                              (6 x 3, rank 2)
// so a contribution W_a D W_b^T = Jc_a^T (Q_a D Q_b^T) Jc_b goes through a 2 x 2 middle: the same ~220 operations as the product of
// materialised blocks, on 112 gathered bytes instead of 336 (9 loads instead of 23), and the linearisation writes 32 bytes per
// observation instead of 144.  Same block -> workgroup placement, same reduction over the lanes, same outputs as k_schur_block.
struct CamConst { double f, R[9]; };
__device__ __forceinline__ CamConst load_cam_const(const double* __restrict__ camR, int hc)
{
    const double2* p = reinterpret_cast<const double2*>(camR + (size_t)hc * 12);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    CamConst k;
    k.f = a.x; k.R[0] = a.y; k.R[1] = b.x; k.R[2] = b.y; k.R[3] = c.x; k.R[4] = c.y; k.R[5] = d.x; k.R[6] = d.y; k.R[7] = e.x; k.R[8] = e.y;
    return k;
}
__device__ __forceinline__ double uniform_d(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ CamConst uniform_cam_const(const CamConst& k)
{
    CamConst u;
    u.f = uniform_d(k.f);
#pragma unroll
    for (int i = 0; i < 9; ++i) u.R[i] = uniform_d(k.R[i]);
    return u;
}
// Jc rows (J0, J1) and Q rows (Q0, Q1) of a slot
__device__ __forceinline__ void slot_factors(const CamConst& k, double a, double b, double iz, double w, double J0[6], double J1[6], double Q0[3], double Q1[3])
{
    const double f = k.f, af = a * f, bf = b * f, izf = iz * f;
    J0[0] = a * bf;        J0[1] = -(f + a * af); J0[2] = bf;  J0[3] = -izf; J0[4] = 0;    J0[5] = a * izf;
    J1[0] = f + b * bf;    J1[1] = -(a * bf);     J1[2] = -af; J1[3] = 0;    J1[4] = -izf; J1[5] = b * izf;
    const double s = -(w * izf);
#pragma unroll
    for (int c = 0; c < 3; ++c) { Q0[c] = s * (k.R[c] - a * k.R[6 + c]); Q1[c] = s * (k.R[3 + c] - b * k.R[6 + c]); }
}

template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 256 : 64 * SCHUR_WAVES) void k_schur_block_compact(BaDeviceView v, double lambda)
{
    constexpr int NW = SPLIT ? 4 : SCHUR_WAVES;
    __shared__ double red[NW][32 * 37];      // the two halves of a wavefront are added in registers first: 9.5 KB per wavefront
    __shared__ double part[SPLIT ? 4 : 1][36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b;
    if (SPLIT) {
        b = (int)blockIdx.x;
        if (b >= v.n_blk) return;
    } else {
        const int slot = blockIdx.x * SCHUR_WAVES + wave;
        if (slot >= v.n_blk_slots) return;
        b = (v.slot_order ? v.slot_order : v.blk_order)[slot];
        if (b < 0) return;
    }
    b = __builtin_amdgcn_readfirstlane(b);
    const int stride = SPLIT ? 4 * WAVE : WAVE;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0;
    // The block's header is two chains of dependent loads -- (i, j) -> the cameras' constants, and the list bounds -> entries -> landmarks
    // -> records.  A block is ~4 trips of the loop below, so its start-up latency counts as much as its arithmetic (two wavefronts per
    // SIMD): both chains are ISSUED before the first wait -- the constants are moved to scalar registers (a wait) only after the first
    // records have been requested.
    const int2 ij = v.blk_ij[b];
    const int c_begin = v.blk_ptr[b], c_end = v.blk_ptr[b + 1];
    const bool diag = ij.x == ij.y;
    const CamConst ci_v = load_cam_const(v.camR, ij.x), cj_v = load_cam_const(v.camR, ij.y);
    double yv[6] = { 0, 0, 0, 0, 0, 0 };
    const double2* G2 = reinterpret_cast<const double2*>(v.W);
    // Software pipeline (round 4): a wavefront keeps 36 sums + the factors of one contribution in ~200 registers, so only two
    // wavefronts share a SIMD and nothing hides the chain  list entry -> records  (the landmark rides in the entry).  Before the ~260
    // f64 operations of contribution c are issued, the records of c + stride and the list entry of c + 2 stride are requested --
    // unconditionally (a load under a branch is waited for at the join): lanes past the end of the block re-read its last entry and
    // add nothing.
    struct Rec { double2 da, db, dc, ga0, ga1, gb0, gb1, e01, e2x; };
    auto fetch = [&](const ConPos e, Rec& r) {
        const double2* D2 = reinterpret_cast<const double2*>(v.Dinv + (size_t)e.lm * 6);
        r.da = D2[0]; r.db = D2[1]; r.dc = D2[2];
        r.ga0 = G2[(size_t)e.a * 2]; r.ga1 = G2[(size_t)e.a * 2 + 1];
        r.gb0 = G2[(size_t)e.b * 2]; r.gb1 = G2[(size_t)e.b * 2 + 1];
        if (diag) {
            const double2* db2 = reinterpret_cast<const double2*>(v.db + (size_t)e.lm * 4);
            r.e01 = db2[0]; r.e2x = db2[1];
        }
    };
    const int c_first = c_begin + (SPLIT ? wave * WAVE : 0);                             // uniform
    const int trips = c_first < c_end ? (c_end - c_first + stride - 1) / stride : 0;     // uniform
    const int c_last = c_end - 1;
    int c = c_first + lane;
    Rec cur, nxt;
    ConPos e1 = { 0, 0, 0 };
    if (trips > 0) {
        const ConPos e0 = v.con_pos[min(c, c_last)];
        e1 = v.con_pos[min(c + stride, c_last)];
        fetch(e0, cur);
    }
    const CamConst ci = uniform_cam_const(ci_v), cj = uniform_cam_const(cj_v);          // in SGPRs: 40 VGPRs less
    for (int t = 0; t < trips; ++t, c += stride) {
        fetch(e1, nxt);
        const ConPos e2 = v.con_pos[min(c + 2 * stride, c_last)];
        if (c < c_end) {
            const double2 da = cur.da, db = cur.db, dc = cur.dc, ga0 = cur.ga0, ga1 = cur.ga1, gb0 = cur.gb0, gb1 = cur.gb1;
            const double d00 = da.x, d01 = da.y, d02 = db.x, d11 = db.y, d12 = dc.x, d22 = dc.y;
            double Ja0[6], Ja1[6], Qa0[3], Qa1[3], Jb0[6], Jb1[6], Qb0[3], Qb1[3];
            slot_factors(ci, ga0.x, ga0.y, ga1.x, ga1.y, Ja0, Ja1, Qa0, Qa1);
            slot_factors(cj, gb0.x, gb0.y, gb1.x, gb1.y, Jb0, Jb1, Qb0, Qb1);
            if (diag) {
                const double e0 = cur.e01.x, e1 = cur.e01.y, e2 = cur.e2x.x;
                const double s0 = Qa0[0] * e0 + Qa0[1] * e1 + Qa0[2] * e2, s1 = Qa1[0] * e0 + Qa1[1] * e1 + Qa1[2] * e2;
#pragma unroll
                for (int r = 0; r < 6; ++r) yv[r] += r == 4 ? Ja1[r] * s1 : r == 3 ? Ja0[r] * s0 : Ja0[r] * s0 + Ja1[r] * s1;
            }
            // M = Qa D Qb^T (2 x 2)
            const double t00 = Qa0[0] * d00 + Qa0[1] * d01 + Qa0[2] * d02, t01 = Qa0[0] * d01 + Qa0[1] * d11 + Qa0[2] * d12, t02 = Qa0[0] * d02 + Qa0[1] * d12 + Qa0[2] * d22;
            const double t10 = Qa1[0] * d00 + Qa1[1] * d01 + Qa1[2] * d02, t11 = Qa1[0] * d01 + Qa1[1] * d11 + Qa1[2] * d12, t12 = Qa1[0] * d02 + Qa1[1] * d12 + Qa1[2] * d22;
            const double m00 = t00 * Qb0[0] + t01 * Qb0[1] + t02 * Qb0[2], m01 = t00 * Qb1[0] + t01 * Qb1[1] + t02 * Qb1[2];
            const double m10 = t10 * Qb0[0] + t11 * Qb0[1] + t12 * Qb0[2], m11 = t10 * Qb1[0] + t11 * Qb1[1] + t12 * Qb1[2];
            // (Jc has two structural zeros, J0[4] and J1[3]: the products with them are left out by hand -- the compiler may not drop 0 * x)
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double n0 = r == 4 ? Ja1[r] * m10 : r == 3 ? Ja0[r] * m00 : Ja0[r] * m00 + Ja1[r] * m10;
                const double n1 = r == 4 ? Ja1[r] * m11 : r == 3 ? Ja0[r] * m01 : Ja0[r] * m01 + Ja1[r] * m11;
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) acc[r * 6 + cc] += cc == 4 ? n1 * Jb1[cc] : cc == 3 ? n0 * Jb0[cc] : n0 * Jb0[cc] + n1 * Jb1[cc];
            }
        }
        cur = nxt; e1 = e2;
    }
    double* R = red[wave];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
    if (lane < 32) {
#pragma unroll
        for (int k = 0; k < 36; ++k) R[lane * 37 + k] = acc[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double val = 0;
    if (lane < 36) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            s0 += R[(j + 0) * 37 + lane]; s1 += R[(j + 1) * 37 + lane]; s2 += R[(j + 2) * 37 + lane]; s3 += R[(j + 3) * 37 + lane];
        }
        val = (s0 + s1) + (s2 + s3);
    }
    double yval = 0;
    if (diag) {                                           // uniform per workgroup
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 6; ++k) yv[k] += __shfl_xor(yv[k], 32, 64);
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 6; ++k) R[lane * 7 + k] = yv[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 6) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int j = 0; j < 32; j += 4) { s0 += R[(j + 0) * 7 + lane]; s1 += R[(j + 1) * 7 + lane]; s2 += R[(j + 2) * 7 + lane]; s3 += R[(j + 3) * 7 + lane]; }
            yval = (s0 + s1) + (s2 + s3);
        }
    }
    if (SPLIT) {
        __shared__ double ypart[4][6];
        if (lane < 36) part[wave][lane] = val;
        if (diag && lane < 6) ypart[wave][lane] = yval;
        __syncthreads();
        if (wave != 0) return;
        if (lane < 36) val = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        if (diag && lane < 6) yval = ((ypart[0][lane] + ypart[1][lane]) + ypart[2][lane]) + ypart[3][lane];
    }
    if (diag && lane < 6) v.y[ij.x * 6 + lane] = v.bc[(size_t)ij.x * 6 + lane] - yval;
    if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        if (ij.x == ij.y) {
            double u = v.U[(size_t)ij.x * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
            v.S[(size_t)(ij.x * 6 + c) * v.n_pad + (ij.x * 6 + r)] = u - val;
        } else {
            v.S[(size_t)(ij.x * 6 + r) * v.n_pad + (ij.y * 6 + c)] = -val;
        }
    }
}

// ---- the same blocks taken as STREAMS OF TRIPS by resident wavefronts (round 6).  k_schur_block_compact<false> gives every 6 x 6 block a
// wavefront of its own: ~16 k wavefronts of ~6 trips each (a third of them ONE trip), every one starting with three dependent trips to
// memory (block header -> list entries -> records) while its SIMD's only other wavefront is, often enough, in the same state.  Here one
// workgroup of eight wavefronts per compute unit (two per SIMD, all resident from the first cycle) works through the unit's LIST OF BLOCKS
// (k_build_stream_lists: the XCD's run of blocks, longest first, dealt out to the XCD's compute units); a wavefront CLAIMS its next block
// from the list with an LDS counter one block ahead of need, so the arithmetic of trip t is issued with the records of trip t + 1, the
// list entries of trip t + 2 and the descriptor of trip t + 3 in flight ACROSS block boundaries -- a block's header costs nothing -- and
// the two wavefronts of a SIMD, of which the hardware favours the older one, end together all the same (per-wavefront lists, however
// evenly cut, left the favoured one done 14 us before the other: profiles/HISTORY.md).  The lane <-> contribution mapping, each lane's
// order of additions and the reduction are those of k_schur_block_compact: the same bits whichever wavefront takes a block
// (tests/test_ba_gpu.py; MAGE_BA_SCHUR_BLOCKS=1 keeps the per-block launch for A/B).
#define MAGE_CONSTANT __attribute__((address_space(4)))
#ifndef SCHUR_ABL
#define SCHUR_ABL 0
#endif
typedef int trip4 __attribute__((ext_vector_type(4)));
struct StreamBlk { int c_begin, c_end, i, j; };
constexpr int STREAM_WAVES = 8;
__global__ __launch_bounds__(64 * STREAM_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_schur_stream(BaDeviceView v, double lambda)
{
    __shared__ double red[STREAM_WAVES][32 * 37];
    __shared__ int head;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* R = red[wave];
    const MAGE_CONSTANT int* gp = (const MAGE_CONSTANT int*)v.stream_ptr;
    const int l_begin = gp[blockIdx.x], l_n = gp[blockIdx.x + 1] - l_begin;
    const MAGE_CONSTANT trip4* LIST = (const MAGE_CONSTANT trip4*)v.stream_blks + l_begin;
    const MAGE_CONSTANT double* CR = (const MAGE_CONSTANT double*)v.camR;
    long long t_start = 0;
    if (v.stream_stamps) t_start = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) head = 0;
    __syncthreads();
    auto claim = [&]() -> int {
        int k = 0;
        if (lane == 0) k = atomicAdd(&head, 1);
        return __builtin_amdgcn_readfirstlane(k);
    };
    const double2* G2 = reinterpret_cast<const double2*>(v.W);
    struct Rec { double2 da, db, dc, ga0, ga1, gb0, gb1, e01, e2x; };
    auto fetch = [&](const ConPos e, const bool dg, Rec& r) {
#if SCHUR_ABL == 1
        const double q = (double)(e.a + lane);
        r.da = r.db = r.dc = r.ga0 = r.ga1 = r.gb0 = r.gb1 = r.e01 = r.e2x = double2{ q, q + 1.0 };
#elif SCHUR_ABL == 3
        // the same bytes requested COOPERATIVELY: three lanes per 48-byte landmark record, two per 32-byte slot record (e.a carries the
        // trip's first contribution, e.b its last valid one) -- a third of the distinct lines per load instruction
        const int* CAx = v.con_soa; const int* CBx = CAx + v.con_soa_pitch; const int* CLx = CBx + v.con_soa_pitch;
        const int c0 = e.a, cl = e.b;
        const char* DI = reinterpret_cast<const char*>(v.Dinv); const char* WW = reinterpret_cast<const char*>(v.W);
        const int l3 = lane == 63 ? 21 : lane / 3, p3 = lane == 63 ? 0 : lane % 3;
        const unsigned lm0 = CLx[min(c0 + l3, cl)], lm1 = CLx[min(c0 + l3 + 21, cl)], lm2 = CLx[min(c0 + l3 + 42, cl)];
        const unsigned a0 = CAx[min(c0 + lane / 2, cl)], a1 = CAx[min(c0 + lane / 2 + 32, cl)];
        const unsigned b0 = CBx[min(c0 + lane / 2, cl)], b1 = CBx[min(c0 + lane / 2 + 32, cl)];
        r.da = *reinterpret_cast<const double2*>(DI + lm0 * 48u + p3 * 16u);
        r.db = *reinterpret_cast<const double2*>(DI + lm1 * 48u + p3 * 16u);
        r.dc = *reinterpret_cast<const double2*>(DI + lm2 * 48u + p3 * 16u);
        r.ga0 = *reinterpret_cast<const double2*>(WW + a0 * 32u + (lane & 1) * 16u);
        r.ga1 = *reinterpret_cast<const double2*>(WW + a1 * 32u + (lane & 1) * 16u);
        r.gb0 = *reinterpret_cast<const double2*>(WW + b0 * 32u + (lane & 1) * 16u);
        r.gb1 = *reinterpret_cast<const double2*>(WW + b1 * 32u + (lane & 1) * 16u);
        r.e01 = r.e2x = double2{ 0, 0 };
        (void)G2;
#else
        const double2* D2 = reinterpret_cast<const double2*>(v.Dinv + (size_t)e.lm * 6);
        r.da = D2[0]; r.db = D2[1]; r.dc = D2[2];
        r.ga0 = G2[(size_t)e.a * 2]; r.ga1 = G2[(size_t)e.a * 2 + 1];
        r.gb0 = G2[(size_t)e.b * 2]; r.gb1 = G2[(size_t)e.b * 2 + 1];
        if (dg) {
            const double2* db2 = reinterpret_cast<const double2*>(v.db + (size_t)e.lm * 4);
            r.e01 = db2[0]; r.e2x = db2[1];
        }
#endif
    };
    // a trip: (first contribution, the block's end, i, j | last << 31); none left: end = -1
#if SCHUR_ABL == 1
    auto entry = [&](const trip4 D) { return ConPos{ D.x + lane, D.y, D.z }; };
#elif SCHUR_ABL == 3
    auto entry = [&](const trip4 D) { return ConPos{ D.x, max(D.y - 1, 0), D.z }; };
#else
    auto entry = [&](const trip4 D) { return v.con_pos[max(min(D.x + lane, D.y - 1), 0)]; };
#endif
    auto is_diag = [](const trip4 D) { return D.z == (D.w & 0xffffff); };
    auto cam = [&](int hc) {
        CamConst k;
        const MAGE_CONSTANT double* p = CR + (size_t)hc * 12;
        k.f = p[0];
#pragma unroll
        for (int q = 0; q < 9; ++q) k.R[q] = p[1 + q];
        return k;
    };
    // the cursor that turns the claimed blocks into trips: `cb` is being cut into trips, `nb` was claimed behind it (its descriptor has been
    // on its way since the claim) and takes over when cb runs out -- at which moment the block after it is claimed
    trip4 cb = { 0, -1, 0, 0 }, nb = { 0, -1, 0, 0 };
    {
        const int k0 = claim();
        if (k0 < l_n) cb = LIST[k0];
        const int k1 = k0 < l_n ? claim() : l_n;
        if (k1 < l_n) nb = LIST[k1];
    }
    int n_trips = 0, n_blocks = 0, n_diag = 0;
    auto next_trip = [&]() -> trip4 {
        if (cb.y < 0) return cb;
        trip4 D = cb;
        cb.x += 64;
        const bool last = cb.x >= cb.y;
        D.w |= last ? (int)0x80000000u : 0;
        if (last) {
            cb = nb;
            nb.y = -1;
            if (cb.y >= 0) { const int k = claim(); if (k < l_n) nb = LIST[k]; }
        }
        return D;
    };
    trip4 D0 = next_trip(), D1 = next_trip(), D2 = next_trip();
    CamConst ci = cam(D0.z), cj = cam(D0.w & 0xffffff);
    Rec cur, nxt;
    ConPos e1;
    {
        const ConPos e0 = entry(D0);
        e1 = entry(D1);
        fetch(e0, is_diag(D0), cur);
    }
    double acc[36], yv[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) yv[k] = 0;
    while (D0.y >= 0) {
        fetch(e1, is_diag(D1), nxt);
        const ConPos e2 = entry(D2);
        const trip4 D3 = next_trip();
        const int bi = D0.z, bj = D0.w & 0xffffff;
        const bool diag = bi == bj;
        ++n_trips; n_diag += diag ? 1 : 0;
        if (D0.x + lane < D0.y) {
#if SCHUR_ABL == 2 || SCHUR_ABL == 3
            acc[0] += (cur.da.x + cur.db.x + cur.dc.x) + (cur.ga0.x + cur.ga1.x) + (cur.gb0.x + cur.gb1.x) + (diag ? cur.e01.x + cur.e2x.x : 0.0) + ci.f + cj.f;
#else
            const double2 da = cur.da, db = cur.db, dc = cur.dc, ga0 = cur.ga0, ga1 = cur.ga1, gb0 = cur.gb0, gb1 = cur.gb1;
            const double d00 = da.x, d01 = da.y, d02 = db.x, d11 = db.y, d12 = dc.x, d22 = dc.y;
            double Ja0[6], Ja1[6], Qa0[3], Qa1[3], Jb0[6], Jb1[6], Qb0[3], Qb1[3];
            slot_factors(ci, ga0.x, ga0.y, ga1.x, ga1.y, Ja0, Ja1, Qa0, Qa1);
            slot_factors(cj, gb0.x, gb0.y, gb1.x, gb1.y, Jb0, Jb1, Qb0, Qb1);
            if (diag) {
                const double e0 = cur.e01.x, e1 = cur.e01.y, e2 = cur.e2x.x;
                const double s0 = Qa0[0] * e0 + Qa0[1] * e1 + Qa0[2] * e2, s1 = Qa1[0] * e0 + Qa1[1] * e1 + Qa1[2] * e2;
#pragma unroll
                for (int r = 0; r < 6; ++r) yv[r] += r == 4 ? Ja1[r] * s1 : r == 3 ? Ja0[r] * s0 : Ja0[r] * s0 + Ja1[r] * s1;
            }
            const double t00 = Qa0[0] * d00 + Qa0[1] * d01 + Qa0[2] * d02, t01 = Qa0[0] * d01 + Qa0[1] * d11 + Qa0[2] * d12, t02 = Qa0[0] * d02 + Qa0[1] * d12 + Qa0[2] * d22;
            const double t10 = Qa1[0] * d00 + Qa1[1] * d01 + Qa1[2] * d02, t11 = Qa1[0] * d01 + Qa1[1] * d11 + Qa1[2] * d12, t12 = Qa1[0] * d02 + Qa1[1] * d12 + Qa1[2] * d22;
            const double m00 = t00 * Qb0[0] + t01 * Qb0[1] + t02 * Qb0[2], m01 = t00 * Qb1[0] + t01 * Qb1[1] + t02 * Qb1[2];
            const double m10 = t10 * Qb0[0] + t11 * Qb0[1] + t12 * Qb0[2], m11 = t10 * Qb1[0] + t11 * Qb1[1] + t12 * Qb1[2];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double n0 = r == 4 ? Ja1[r] * m10 : r == 3 ? Ja0[r] * m00 : Ja0[r] * m00 + Ja1[r] * m10;
                const double n1 = r == 4 ? Ja1[r] * m11 : r == 3 ? Ja0[r] * m01 : Ja0[r] * m01 + Ja1[r] * m11;
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) acc[r * 6 + cc] += cc == 4 ? n1 * Jb1[cc] : cc == 3 ? n0 * Jb0[cc] : n0 * Jb0[cc] + n1 * Jb1[cc];
            }
#endif
        }
        if (D0.w < 0) {                                      // the block's last trip (uniform): reduce, write, clear
            ++n_blocks;
            ci = cam(D1.z); cj = cam(D1.w & 0xffffff);       // the next block's cameras (scalar loads): requested here, back by the end of the reduction
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 36; ++k) R[lane * 37 + k] = acc[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double val = 0;
            if (lane < 36) {
                double rr[32];                               // all 32 reads in flight before the first addition (same order of additions)
#pragma unroll
                for (int j = 0; j < 32; ++j) rr[j] = R[j * 37 + lane];
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                for (int j = 0; j < 32; j += 4) { s0 += rr[j + 0]; s1 += rr[j + 1]; s2 += rr[j + 2]; s3 += rr[j + 3]; }
                val = (s0 + s1) + (s2 + s3);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (diag) {
                double yval = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) yv[k] += __shfl_xor(yv[k], 32, 64);
                if (lane < 32) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) R[lane * 7 + k] = yv[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 6) {
                    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                    for (int j = 0; j < 32; j += 4) { s0 += R[(j + 0) * 7 + lane]; s1 += R[(j + 1) * 7 + lane]; s2 += R[(j + 2) * 7 + lane]; s3 += R[(j + 3) * 7 + lane]; }
                    yval = (s0 + s1) + (s2 + s3);
                    v.y[bi * 6 + lane] = v.bc[(size_t)bi * 6 + lane] - yval;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < 6; ++k) yv[k] = 0;
            }
            if (lane < 36) {
                const int r = lane / 6, c = lane % 6;
                if (diag) {
                    double u = v.U[(size_t)bi * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
                    v.S[(size_t)(bi * 6 + c) * v.n_pad + (bi * 6 + r)] = u - val;
                } else {
                    v.S[(size_t)(bi * 6 + r) * v.n_pad + (bj * 6 + c)] = -val;
                }
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k] = 0;
        }
        cur = nxt; e1 = e2; D0 = D1; D1 = D2; D2 = D3;
    }
    if (v.stream_stamps && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* o = v.stream_stamps + ((size_t)blockIdx.x * STREAM_WAVES + wave) * 4;
        o[0] = t_start; o[1] = (long long)__builtin_amdgcn_s_memrealtime(); o[2] = (long long)hw | ((long long)xcc << 32); o[3] = (long long)n_trips | ((long long)n_blocks << 32) | ((long long)n_diag << 48);
    }
}
// The block lists of k_schur_stream.  Workgroup g runs on XCD g % 8 and is that XCD's compute unit q = g / 8 of Q.  The XCD's run of
// blocks, longest first (slot_order), is dealt out in ROUNDS of Q: in every round the unit with the least work so far takes the round's
// longest block, and so on.  A block costs its trips + the reduction and write-out (an eighth of a trip) + a quarter per trip for a diagonal
// block's right-hand side (fitted to the units' end times, MAGE_BA_SCHUR_TRACE).  A unit's list is in round order: longest first.
// k_assign_blocks: one workgroup per XCD, thread = unit; k_build_stream_lists: one workgroup: count, scan, fill.
constexpr int STREAM_Q_MAX = 1024;
__global__ __launch_bounds__(STREAM_Q_MAX) void k_assign_blocks(BaDeviceView v, int Q, int rounds, int* __restrict__ group_blocks)
{
    __shared__ int load[STREAM_Q_MAX];
    const int x = blockIdx.x, q = threadIdx.x;
    const int* order = v.slot_order ? v.slot_order : v.blk_order;
    const int n = v.n_blk_slots > x ? (v.n_blk_slots - x + 7) / 8 : 0;
    int mine = 0;
    if (q < Q) load[q] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        int rank = 0;
        if (q < Q)
            for (int o = 0; o < Q; ++o) { const int lo = load[o]; rank += (lo < mine || (lo == mine && o < q)) ? 1 : 0; }
        __syncthreads();
        if (q < Q) {
            const int e = r * Q + rank;
            const int b = e < n ? order[x + 8 * e] : -1;
            group_blocks[(size_t)(q * 8 + x) * rounds + r] = b;
            if (b >= 0) {
                const int trips = max(1, (v.blk_ptr[b + 1] - v.blk_ptr[b] + 63) >> 6);
                const int2 ij = v.blk_ij[b];
                mine += 8 * trips + 1 + (ij.x == ij.y ? 2 * trips : 0);
            }
            load[q] = mine;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_build_stream_lists(BaDeviceView v, int n_groups, int rounds, const int* __restrict__ group_blocks, int* __restrict__ group_ptr, StreamBlk* __restrict__ blks)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n_groups + 1023) / 1024;                    // consecutive groups per thread
    int mine = 0;
    for (int k = 0; k < per; ++k) {
        const int g = tid * per + k;
        if (g < n_groups)
            for (int r = 0; r < rounds; ++r) mine += group_blocks[(size_t)g * rounds + r] >= 0 ? 1 : 0;
    }
    part[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int at = part[tid] - mine;
    for (int k = 0; k < per; ++k) {
        const int g = tid * per + k;
        if (g >= n_groups) break;
        group_ptr[g] = at;
        for (int r = 0; r < rounds; ++r) {
            const int b = group_blocks[(size_t)g * rounds + r];
            if (b < 0) continue;
            const int2 ij = v.blk_ij[b];
            blks[at++] = StreamBlk{ v.blk_ptr[b], v.blk_ptr[b + 1], ij.x, ij.y };
        }
    }
    if (tid == 1023) group_ptr[n_groups] = part[1023];
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_schur_rhs(BaDeviceView v)
{
    const int wave = threadIdx.x >> 6;
    const int hc = SPLIT ? (int)blockIdx.x : xcd_camera((int)blockIdx.x, wave, v.n_fc);
    const int lane = threadIdx.x & 63;
    if (hc >= v.n_fc) return;
    const int first = SPLIT ? wave * WAVE + lane : lane, stride = SPLIT ? 4 * WAVE : WAVE;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int idx = v.camS_ptr[hc] + first; idx < v.camS_ptr[hc + 1]; idx += stride) {
        const int s = v.camS[idx];
        const double2* W2 = reinterpret_cast<const double2*>(v.W + (size_t)s * 18);
        const double2* db2 = reinterpret_cast<const double2*>(v.db + (size_t)v.w_lm[s] * 4);
        double W[18];
#pragma unroll
        for (int k = 0; k < 9; ++k) { const double2 t = W2[k]; W[2 * k] = t.x; W[2 * k + 1] = t.y; }
        const double2 da = db2[0];
        const double d0 = da.x, d1 = da.y, d2 = db2[1].x;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[r] += W[r * 3] * d0 + W[r * 3 + 1] * d1 + W[r * 3 + 2] * d2;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[r] = wave_sum(acc[r]);
    if (SPLIT) {
        __shared__ double part[4][6];
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r) part[wave][r] = acc[r];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[r] = ((part[0][r] + part[1][r]) + part[2][r]) + part[3][r];
    }
    if (lane < 6) {
        double a = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) if (lane == r) a = acc[r];
        v.y[hc * 6 + lane] = v.bc[(size_t)hc * 6 + lane] - a;
    }
}

// ---------------------------------------------------------------------------------------------
// back substitution, state update, scale = sum x (lambda x + b)
// ---------------------------------------------------------------------------------------------
// BACKSUB_LPL lanes per landmark: lane `sub` takes the landmark's W blocks s0 + sub, s0 + sub + 8, ... (a landmark has ~10 of them and a
// thread that walks them alone waits out ten dependent 144-byte reads with 1.5 workgroups per compute unit to hide them), the three
// sums are added over the lanes in a fixed tree, lane 0 of the group finishes the landmark.
constexpr int BACKSUB_LPL = 8;
// WITH_ERROR: every lane of the landmark's group ends with the trial point and evaluates its share of the landmark's observations
// against the trial poses (k_pose_update has run): the residuals and chi2 partials of k_error(trial) without reading the trial state
// back, in partial[chi_off + block].
// (7 wavefronts per SIMD: at 74 registers six fitted and the 3 125 blocks of the 1k-pose map were 2.03 rounds of 1 536; at 72 seven fit --
//  36.0 -> 34.4 us; eight cost a spill and 35.7 us)
template <bool WITH_ERROR>
__global__ __launch_bounds__(256, 7) void k_backsub(BaDeviceView v, double lambda, double delta, int chi_off)
{
    __shared__ double sm[4];
    const int gl = blockIdx.x * 256 + threadIdx.x, l = gl / BACKSUB_LPL, sub = gl % BACKSUB_LPL;
    double sc = 0;
    double c0 = 0, c1 = 0, c2 = 0;
    if (l < v.n_lm) {
        const int s1 = v.lm_wptr[l + 1];
        for (int s = v.lm_wptr[l] + sub; s < s1; s += BACKSUB_LPL) {
            if (v.compact) {          // W^T x = Q^T (Jc x), uniform branch
                const int hc = v.w_hc[s];
                const double2* G2 = reinterpret_cast<const double2*>(v.W + (size_t)v.w_pos[s] * 4);
                const double2 g0 = G2[0], g1 = G2[1];
                const double2* x2 = reinterpret_cast<const double2*>(v.xc + (size_t)hc * 6);
                const double2 xa = x2[0], xb = x2[1], xd = x2[2];
                const CamConst k = load_cam_const(v.camR, hc);
                double J0[6], J1[6], Q0[3], Q1[3];
                slot_factors(k, g0.x, g0.y, g1.x, g1.y, J0, J1, Q0, Q1);
                const double u0 = -(J0[0] * xa.x + J0[1] * xa.y + J0[2] * xb.x + J0[3] * xb.y + J0[4] * xd.x + J0[5] * xd.y);
                const double u1 = -(J1[0] * xa.x + J1[1] * xa.y + J1[2] * xb.x + J1[3] * xb.y + J1[4] * xd.x + J1[5] * xd.y);
                c0 += Q0[0] * u0 + Q1[0] * u1; c1 += Q0[1] * u0 + Q1[1] * u1; c2 += Q0[2] * u0 + Q1[2] * u1;
                continue;
            }
            const double2* W2 = reinterpret_cast<const double2*>(v.W + (size_t)s * 18);
            const double2* x2 = reinterpret_cast<const double2*>(v.xc + (size_t)v.w_hc[s] * 6);
            double W[18], x[6];
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = W2[k]; W[2 * k] = t.x; W[2 * k + 1] = t.y; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double2 t = x2[k]; x[2 * k] = t.x; x[2 * k + 1] = t.y; }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double mx = -x[r];
                c0 += W[r * 3] * mx; c1 += W[r * 3 + 1] * mx; c2 += W[r * 3 + 2] * mx;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < BACKSUB_LPL; m <<= 1) { c0 += __shfl_xor(c0, m, 64); c1 += __shfl_xor(c1, m, 64); c2 += __shfl_xor(c2, m, 64); }
    double chi = 0;
    if (l < v.n_lm && (WITH_ERROR || sub == 0)) {
        const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
        c0 += b0; c1 += b1; c2 += b2;
        const double* D = v.Dinv + (size_t)l * 6;
        const double x0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
        const double x1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
        const double x2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
        const int pt = v.lm_pt[l];
        const double* pc = v.pt_cur + (size_t)pt * 4;
        const double X = pc[0] + x0, Y = pc[1] + x1, Z = pc[2] + x2;
        if (sub == 0) {
            double* xl = v.xl + (size_t)l * 4;
            xl[0] = x0; xl[1] = x1; xl[2] = x2; xl[3] = 0;
            double* pt_t = v.pt_trial + (size_t)pt * 4;
            pt_t[0] = X; pt_t[1] = Y; pt_t[2] = Z;
            sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
        if (WITH_ERROR) {
            const int end = v.lm_ptr[l + 1];
            for (int i = v.lm_ptr[l] + sub; i < end; i += BACKSUB_LPL) {
                if (!v.L_active[i]) continue;
                const int cam = v.L_cam[i];
                const PoseD P = load_pose(v.pose_trial, cam);
                EdgeGeom g = edge_geom(P, v.camK, cam, X, Y, Z, v.L_uv[i]);
                *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
                double rho0, rho1;
                huber((double)v.L_info[i] * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
                chi += rho0;
            }
        }
    }
    const double r = block_sum<4>(sc, sm);
    if (threadIdx.x == 0) v.partial[blockIdx.x] = r;
    if (WITH_ERROR) {
        const double r1 = block_sum<4>(chi, sm);
        if (threadIdx.x == 0) v.partial[chi_off + blockIdx.x] = r1;
    }
}

__device__ __forceinline__ void m3mul(const double A[9], const double B[9], double C[9])
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

// Eigen matrix -> quaternion.  (The branch for a non-positive trace picks the largest diagonal element at run time; it is written
// out once per choice with constant indices -- indexed by a run-time i the matrix went to scratch memory in every caller.)
template <int I>
__device__ __forceinline__ void R_to_q_branch(const double m[9], double c[4])
{
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    double t = sqrt(m[I * 4] - m[J * 4] - m[K * 4] + 1.0);
    const double ci = 0.5 * t;
    t = 0.5 / t;
    c[3] = (m[K * 3 + J] - m[J * 3 + K]) * t;
    c[I] = ci;
    c[J] = (m[J * 3 + I] + m[I * 3 + J]) * t;
    c[K] = (m[K * 3 + I] + m[I * 3 + K]) * t;
}
__device__ __forceinline__ void R_to_q(const double m[9], double c[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        c[3] = 0.5 * t;
        t = 0.5 / t;
        c[0] = (m[7] - m[5]) * t; c[1] = (m[2] - m[6]) * t; c[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > (i == 1 ? m[4] : m[0])) i = 2;
        if (i == 0) R_to_q_branch<0>(m, c);
        else if (i == 1) R_to_q_branch<1>(m, c);
        else R_to_q_branch<2>(m, c);
    }
}

// pose <- exp(x) * pose for one free camera (VertexSE3Expmap::oplusImpl, appendix A.1); returns its share of the scale term
__device__ __forceinline__ double pose_update_compute(const BaDeviceView& v, double lambda, int hc, PoseD& O)
{
    double sc = 0;
    const int cam = v.hc2cam[hc];
    const double* u = v.xc + (size_t)hc * 6;
    const double* b = v.bc + (size_t)hc * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) sc += u[k] * (lambda * u[k] + b[k]);
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double Om[9] = { 0, -w2, w1, w2, 0, -w0, -w1, w0, 0 };
    double Om2[9], R[9], Vm[9];
    m3mul(Om, Om, Om2);
    double a, bb, d;
    if (theta < 0.00001) { a = 1.0; bb = 0.5; d = 1.0 / 6.0; }
    else {
        double s, c;
        sincos(theta, &s, &c);                 // one argument reduction for both (the two calls were ~1 us of dependent f64 operations on the pose-only kernel's lone thread)
        a = s / theta; bb = (1 - c) / (theta * theta); d = (theta - s) / (theta * theta * theta);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double I = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
        R[k] = I + a * Om[k] + bb * Om2[k];
        Vm[k] = I + bb * Om[k] + d * Om2[k];
    }
    double q[4];
    R_to_q(R, q);
    double ex = Vm[0] * u[3] + Vm[1] * u[4] + Vm[2] * u[5];
    double ey = Vm[3] * u[3] + Vm[4] * u[4] + Vm[5] * u[5];
    double ez = Vm[6] * u[3] + Vm[7] * u[4] + Vm[8] * u[5];
    // SE3Quat(q, t) ctor normalises
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
    PoseD P = load_pose(v.pose_cur, cam);
    // result = E * P : t = E.t + E.r * P.t ; r = E.r * P.r ; normalise
    double rx, ry, rz;
    q_rot(q[0], q[1], q[2], q[3], P.tx, P.ty, P.tz, rx, ry, rz);
    O.tx = ex + rx; O.ty = ey + ry; O.tz = ez + rz;
    O.qw = q[3] * P.qw - q[0] * P.qx - q[1] * P.qy - q[2] * P.qz;
    O.qx = q[3] * P.qx + q[0] * P.qw + q[1] * P.qz - q[2] * P.qy;
    O.qy = q[3] * P.qy + q[1] * P.qw + q[2] * P.qx - q[0] * P.qz;
    O.qz = q[3] * P.qz + q[2] * P.qw + q[0] * P.qy - q[1] * P.qx;
    if (O.qw < 0) { O.qx = -O.qx; O.qy = -O.qy; O.qz = -O.qz; O.qw = -O.qw; }
    n = sqrt(O.qx * O.qx + O.qy * O.qy + O.qz * O.qz + O.qw * O.qw);
    O.qx /= n; O.qy /= n; O.qz /= n; O.qw /= n;
    return sc;
}
__device__ __forceinline__ double pose_update_one(const BaDeviceView& v, double lambda, int hc)
{
    PoseD O;
    const double sc = pose_update_compute(v, lambda, hc, O);
    store_pose(v.pose_trial, v.hc2cam[hc], O);
    return sc;
}

// every free camera; one thread per camera; scale partials
__global__ __launch_bounds__(256) void k_pose_update(BaDeviceView v, double lambda, int part_off, int zero_off)
{
    __shared__ double sm[4];
    const int hc = blockIdx.x * 256 + threadIdx.x;
    double sc = 0;
    if (hc < v.n_fc) sc = pose_update_one(v, lambda, hc);
    double r = block_sum<4>(sc, sm);
    if (threadIdx.x == 0) {
        v.partial[part_off + blockIdx.x] = r;
        if (zero_off >= 0) v.partial[zero_off + blockIdx.x] = 0.0;       // this row of the second sum of a fused reduction
    }
}

// ---------------------------------------------------------------------------------------------
// outlier classification (BundlerLib.cpp:384-427): uses the residuals of the LAST error evaluation
// (errL) and the KEPT estimates for the in-front-of-camera test, exactly as the reference does.
// ---------------------------------------------------------------------------------------------
// Outliers are appended (in no particular order; the host sorts them) to out_ids as ORIGINAL observation indices: what crosses
// PCIe is the list, not a flag per observation.  The cursor *out_count only ever grows between structure builds; out_base is its
// value before this launch (the host has read every earlier count), so nothing has to be cleared.
// The host's decision on an LM trial, repeated on the device from the same scalars in the same arithmetic (lm_solve in ba_host.hip):
// is this StepBundleAdjustment call over with the trial just evaluated, and which estimate is kept?
__device__ __forceinline__ bool call_is_over(const BaDeviceView& v, const ClassifyAfterTrial& spec, bool& accept)
{
    const bool stalled = v.scal[SC_CHOL_STALL] != 0.0;
    const bool ok2 = v.scal[SC_CHOL_OK] != 0.0 && !stalled;
    const double temp = v.scal[SC_CHI_TRIAL];
    const double cur = spec.chi_on_device ? v.scal[SC_CHI] : spec.chi_ref;
    const double rho = ok2 ? (cur - temp) / (v.scal[SC_SCALE] + 1e-3) : -1.0;
    accept = ok2 && rho > 0 && isfinite(temp);
    const bool loop_over = !(rho < 0 && spec.trials_done < 10);
    const bool terminate = spec.trials_done == 10 || rho == 0;
    return !stalled && loop_over && (spec.last_iteration || terminate);
}

template <bool AFTER_TRIAL>
__global__ __launch_bounds__(256) void k_classify(BaDeviceView v, double max_err_sq, uint32_t* __restrict__ out_ids, int* __restrict__ out_count, int out_base, int nb,
                                                  ClassifyAfterTrial spec)
{
    __shared__ double sm[4];
    const double* pose_kept = v.pose_cur;
    const double* pt_kept = v.pt_cur;
    if (AFTER_TRIAL) {
        bool accept;
        const bool fin = call_is_over(v, spec, accept);
        if (blockIdx.x == 0 && threadIdx.x == 0) v.scal[SC_SPEC_DONE] = fin ? 1.0 : 0.0;
        if (!fin) return;
        if (accept) { pose_kept = v.pose_trial; pt_kept = v.pt_trial; }
    }
    double es = 0, ec = 0, no = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_L; i += gridDim.x * 256) {
        if (!v.L_active[i]) continue;
        const double2 e = *reinterpret_cast<const double2*>(v.errL + (size_t)i * 2);
        const double ss = e.x * e.x + e.y * e.y;
        const int cam = v.L_cam[i], pt = v.L_pt[i];
        PoseD P = load_pose(pose_kept, cam);
        double wx, wy, wz, fx, fy, fz;
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, -P.tx, -P.ty, -P.tz, wx, wy, wz);   // camera centre
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, 0.0, 0.0, 1.0, fx, fy, fz);        // forward axis in world
        const double* X = pt_kept + (size_t)pt * 4;
        const double dot = (X[0] - wx) * fx + (X[1] - wy) * fy + (X[2] - wz) * fz;
        const bool out = (dot <= 0) || (ss > max_err_sq);
        if (out) { no += 1.0; v.L_active[i] = 0; out_ids[atomicAdd(out_count, 1) - out_base] = v.L_edge[i]; }   // removeEdge: the observation leaves the graph on the device right here
        else { es += ss; ec += 1.0; }
    }
    double r0 = block_sum<4>(es, sm);
    double r1 = block_sum<4>(ec, sm);
    double r2 = block_sum<4>(no, sm);
    if (threadIdx.x == 0) { v.partial[blockIdx.x] = r0; v.partial[nb + blockIdx.x] = r1; v.partial[2 * nb + blockIdx.x] = r2; }
}

// ---------------------------------------------------------------------------------------------
// pose exchange of a window-sharded map: 8 doubles per pose, one thread per double
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_poses(const double* __restrict__ pose, const uint32_t* __restrict__ cam,
                                                      const uint32_t* __restrict__ row, size_t n, double* __restrict__ block)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 8) return;
    const size_t k = i >> 3, a = i & 7;
    block[(size_t)row[k] * 8 + a] = pose[(size_t)cam[k] * 8 + a] + 0.0;      // -0.0 -> +0.0
}
__global__ __launch_bounds__(256) void k_import_poses(double* __restrict__ pose0, double* __restrict__ pose1, const uint32_t* __restrict__ cam,
                                                      const uint32_t* __restrict__ row, size_t n, const double* __restrict__ block)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 8) return;
    const size_t k = i >> 3, a = i & 7;
    const double val = block[(row ? (size_t)row[k] : k) * 8 + a];
    pose0[(size_t)cam[k] * 8 + a] = val;
    pose1[(size_t)cam[k] * 8 + a] = val;
}

// =================================================================================================
// SMALL PROBLEMS: the reduced camera system fits one 128x128 tile (local bundle adjustment: ~15 free keyframes; map
// initialisation; the tracker's pose-only refinement).  There the step is launch-latency, not bandwidth: the large-problem
// path spends ~22 launches and 4-5 us each on a 78x78 system.  Here one LM trial is FIVE launches,
//     k_small_linearize   landmark side + camera side + zero-fill of S, y (roles by block range); the LAST block to finish
//                         adds the chi2 partials in block order and (first iteration) takes max |diag|
//     k_small_schur       all blocks of S + the reduced rhs ((V + lambda I)^-1 formed where it is used, no D^-1 array)
//     k_small_solve       (chol_kernels.hip) Cholesky of the leading ceil(n/16) blocks + both substitutions in LDS, one workgroup
//     k_small_update      back-substitution + pose update; last block adds the scale partials
//     k_small_error       residuals of the trial state; last block adds the chi2 partials
// plus one scalar read-back.  Same arithmetic per element as the large-problem kernels (the device functions are shared);
// sums keep a fixed order (partials in block order), so results stay bit-reproducible run to run.
// Callers: Tasks/MappingWorker.cpp:330-371 (local BA), Tracking/TrackLocalMap.cpp:421-501, Tracking/PoseEstimator.cpp:168-207.
// =================================================================================================
// Every block calls this once, after writing its partial(s): true in the block that arrives last (it then sees all partials).
__device__ __forceinline__ bool last_block_arrives(int* __restrict__ counter, int n_blocks)
{
    __shared__ int is_last;
    // Every wavefront waits until its own stores have reached L2; then ONE release fence in thread 0 writes this XCD's L2 back
    // (the blocks of a launch sit on different XCDs) before the count.  __threadfence() in every thread costs a write-back per
    // wavefront and, at a few dozen blocks per launch, more than the kernels' arithmetic.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = prev == n_blocks - 1;
        if (is_last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    __syncthreads();
    if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return is_last != 0;
}
// sum of partial[0..n) in a fixed order by one block; result to *out (all 256 threads call)
__device__ __forceinline__ void fold_partials(const double* __restrict__ partial, int n, double* __restrict__ out, double* sm4)
{
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double r = block_sum<4>(acc, sm4);
    if (threadIdx.x == 0) *out = r;
}

// (V_l + lambda I)^-1 in the cofactor form of k_lm_invert (same operations, same order)
__device__ __forceinline__ void lm_dinv_from(const double* Vl, double lambda, double D[6])
{
    const double m00 = Vl[0] + lambda, m01 = Vl[1], m02 = Vl[2], m11 = Vl[3] + lambda, m12 = Vl[4], m22 = Vl[5] + lambda;
    const double c00 = m11 * m22 - m12 * m12;
    const double c10 = m12 * m02 - m22 * m01;
    const double c20 = m01 * m12 - m02 * m11;
    const double det = c00 * m00 + c10 * m01 + c20 * m02;
    const double id = 1.0 / det;
    D[0] = c00 * id; D[1] = c10 * id; D[2] = c20 * id;
    D[3] = (m22 * m00 - m02 * m02) * id;
    D[4] = (m02 * m01 - m00 * m12) * id;
    D[5] = (m00 * m11 - m01 * m01) * id;
}
__device__ __forceinline__ void lm_dinv(const BaDeviceView& v, int l, double lambda, double D[6]) { lm_dinv_from(v.V + (size_t)l * 6, lambda, D); }

// Lanes per landmark and workgroups per camera of the linearisation below: a landmark has ~10 observations and a camera of a local
// window ~2500, and a thread that walks them one after another waits out three dependent loads per observation with nothing else
// on its compute unit to hide them (a local BA is ~200 wavefronts on 256 compute units).
constexpr int SMALL_LPL = 8, SMALL_CPC = 4;

// cpc = workgroups per camera: SMALL_CPC for the few cameras of a small problem (their quarters are added by the last block), 1 for
// the large-problem use of the same kernel (ba_fused_linearize: hundreds of cameras, the workgroup writes U and b_c itself);
// zero_role = 0 drops the S / y zero-fill block (large systems clear S with k_zero_lower).
// DUP = some observations share a W slot (the landmark role then walks a landmark's observations on one thread and sums the blocks of
// a slot in order): a launch is entirely one kind or the other, and compiled together the rare kind's 27 accumulators set the spills of both.
template <bool DUP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DUP ? 3 : 4, 4))) void k_small_linearize(BaDeviceView v, double delta, int nbL, int want_maxdiag, int* __restrict__ counter, int cpc,
                                                         int zero_role)
{
    __shared__ double sm[4];
    __shared__ double part[4][28];
    __shared__ double udiag[128];
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_blocks = gridDim.x;
    const int nbC = cpc == 1 ? ((v.n_fc + 7) / 8) * 8 + 8 : v.n_fc * cpc;       // cpc == 1: runs of cameras per XCD (camera role below)
    double* cam_part = v.partial + n_blocks;           // n_fc x cpc x 28 partial (U, b_c) sums, behind the chi2 partials
    double chi = 0;                                    // this thread's share of the robust chi2 (one role per problem kind owns it)
    if (bid < nbL && !DUP) {
        // ---- landmark role, SMALL_LPL lanes per landmark: lane `sub` takes observations beg + sub, beg + sub + 8, ...; every
        // observation owns its W block (no two share a slot), V_l and b_l are summed over the lanes in a fixed tree
        const int gl = bid * 256 + tid, l = gl / SMALL_LPL, sub = gl % SMALL_LPL;
        double V[6] = { 0, 0, 0, 0, 0, 0 }, bp[3] = { 0, 0, 0 };
        if (l < v.n_lm) {
            const int pt = v.lm_pt[l];
            const double X = v.pt_cur[(size_t)pt * 4], Y = v.pt_cur[(size_t)pt * 4 + 1], Z = v.pt_cur[(size_t)pt * 4 + 2];
            const int end = v.lm_ptr[l + 1];
            for (int i = v.lm_ptr[l] + sub; i < end; i += SMALL_LPL) {
                const int slot = v.L_slot[i];
                double Wacc[18];
#pragma unroll
                for (int k = 0; k < 18; ++k) Wacc[k] = 0;
                double cg0 = 0, cg1 = 0, cg2 = 0, cg3 = 0;          // COMPACT form of the slot (an inactive observation weighs nothing)
                if (v.L_active[i]) {
                    const int cam = v.L_cam[i];
                    PoseD P = load_pose(v.pose_cur, cam);
                    EdgeGeom g = edge_geom(P, v.camK, cam, X, Y, Z, v.L_uv[i]);
                    *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
                    const double f = v.camK[cam * 4];
                    const double info = (double)v.L_info[i];
                    double rho0, rho1;
                    huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
                    chi += rho0;
                    const double w = info * rho1;
                    cg0 = g.x / g.z; cg1 = g.y / g.z; cg2 = 1.0 / g.z; cg3 = w;
                    const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
                    double R[9], Jp[6];
                    q_to_R(P.qx, P.qy, P.qz, P.qw, R);
                    jac_point(g, f, R, Jp);
#pragma unroll
                    for (int a = 0; a < 3; ++a) bp[a] += Jp[a] * r0 + Jp[3 + a] * r1;
                    V[0] += Jp[0] * w * Jp[0] + Jp[3] * w * Jp[3];
                    V[1] += Jp[0] * w * Jp[1] + Jp[3] * w * Jp[4];
                    V[2] += Jp[0] * w * Jp[2] + Jp[3] * w * Jp[5];
                    V[3] += Jp[1] * w * Jp[1] + Jp[4] * w * Jp[4];
                    V[4] += Jp[1] * w * Jp[2] + Jp[4] * w * Jp[5];
                    V[5] += Jp[2] * w * Jp[2] + Jp[5] * w * Jp[5];
                    if (slot >= 0 && !v.compact) {
                        double Jc[12];
                        jac_pose(g, f, Jc);
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int c = 0; c < 3; ++c) Wacc[a * 3 + c] += Jc[a] * w * Jp[c] + Jc[6 + a] * w * Jp[3 + c];
                    }
                }
                if (slot >= 0) {
                    if (v.compact) {
                        double2* Gd = reinterpret_cast<double2*>(v.W + (size_t)v.w_pos[slot] * 4);
                        Gd[0] = make_double2(cg0, cg1); Gd[1] = make_double2(cg2, cg3);
                    } else {
                        double2* Wd = reinterpret_cast<double2*>(v.W + (size_t)slot * 18);            // 144-byte block, 16-byte aligned: nine 128-bit stores
#pragma unroll
                        for (int k = 0; k < 9; ++k) Wd[k] = make_double2(Wacc[2 * k], Wacc[2 * k + 1]);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 1; m < SMALL_LPL; m <<= 1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) V[k] += __shfl_xor(V[k], m, 64);
#pragma unroll
            for (int k = 0; k < 3; ++k) bp[k] += __shfl_xor(bp[k], m, 64);
        }
        if (l < v.n_lm && sub == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) v.V[(size_t)l * 6 + k] = V[k];
            v.bp[(size_t)l * 4 + 0] = bp[0]; v.bp[(size_t)l * 4 + 1] = bp[1]; v.bp[(size_t)l * 4 + 2] = bp[2]; v.bp[(size_t)l * 4 + 3] = 0;
        }
    } else if (bid < nbL) {
        // ---- landmark role, one thread per landmark (some observations share a W slot: their blocks are summed in order):
        // k_linearize_lm, plus the residuals / chi2 k_error would have produced
        const int l = bid * 256 + tid;
        if (l < v.n_lm) {
            const int pt = v.lm_pt[l];
            const double X = v.pt_cur[(size_t)pt * 4], Y = v.pt_cur[(size_t)pt * 4 + 1], Z = v.pt_cur[(size_t)pt * 4 + 2];
            double V[6] = { 0, 0, 0, 0, 0, 0 }, bp[3] = { 0, 0, 0 };
            double Wacc[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) Wacc[k] = 0;
            const int beg = v.lm_ptr[l], end = v.lm_ptr[l + 1];
            int cur_slot = -1;
            for (int i = beg; i < end; ++i) {
                const int slot = v.L_slot[i];
                if (slot != cur_slot) {
                    if (cur_slot >= 0) {
                        double2* Wd = reinterpret_cast<double2*>(v.W + (size_t)cur_slot * 18);        // 144-byte block, 16-byte aligned: nine 128-bit stores
#pragma unroll
                        for (int k = 0; k < 9; ++k) { Wd[k] = make_double2(Wacc[2 * k], Wacc[2 * k + 1]); Wacc[2 * k] = 0; Wacc[2 * k + 1] = 0; }
                    }
                    cur_slot = slot;
                }
                if (!v.L_active[i]) continue;
                const int cam = v.L_cam[i];
                PoseD P = load_pose(v.pose_cur, cam);
                EdgeGeom g = edge_geom(P, v.camK, cam, X, Y, Z, v.L_uv[i]);
                *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
                const double f = v.camK[cam * 4];
                const double info = (double)v.L_info[i];
                double rho0, rho1;
                huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
                chi += rho0;
                const double w = info * rho1;
                const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
                double R[9], Jp[6];
                q_to_R(P.qx, P.qy, P.qz, P.qw, R);
                jac_point(g, f, R, Jp);
#pragma unroll
                for (int a = 0; a < 3; ++a) bp[a] += Jp[a] * r0 + Jp[3 + a] * r1;
                V[0] += Jp[0] * w * Jp[0] + Jp[3] * w * Jp[3];
                V[1] += Jp[0] * w * Jp[1] + Jp[3] * w * Jp[4];
                V[2] += Jp[0] * w * Jp[2] + Jp[3] * w * Jp[5];
                V[3] += Jp[1] * w * Jp[1] + Jp[4] * w * Jp[4];
                V[4] += Jp[1] * w * Jp[2] + Jp[4] * w * Jp[5];
                V[5] += Jp[2] * w * Jp[2] + Jp[5] * w * Jp[5];
                if (slot >= 0) {
                    double Jc[12];
                    jac_pose(g, f, Jc);
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) Wacc[a * 3 + b] += Jc[a] * w * Jp[b] + Jc[6 + a] * w * Jp[3 + b];
                }
            }
            if (cur_slot >= 0) {
                double2* Wd = reinterpret_cast<double2*>(v.W + (size_t)cur_slot * 18);
#pragma unroll
                for (int k = 0; k < 9; ++k) Wd[k] = make_double2(Wacc[2 * k], Wacc[2 * k + 1]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) v.V[(size_t)l * 6 + k] = V[k];
            v.bp[(size_t)l * 4 + 0] = bp[0]; v.bp[(size_t)l * 4 + 1] = bp[1]; v.bp[(size_t)l * 4 + 2] = bp[2]; v.bp[(size_t)l * 4 + 3] = 0;
        }
    } else if (bid < nbL + nbC) {
        // ---- camera role: k_linearize_cam<true>, SMALL_CPC workgroups per camera, each a contiguous quarter of the camera's
        // observations; the last block adds the quarters in order.  Owns the chi2 when the points are fixed
        int hc = (bid - nbL) / cpc;
        const int quarter = (bid - nbL) % cpc;
        if (cpc == 1) {
            // large problems: workgroups go to the eight XCDs round-robin by their index in the GRID, and a camera's observation
            // records are gathered from lines it shares with the cameras next to it -- so XCD x takes the contiguous run of cameras
            // [x run, (x + 1) run) and those lines are fetched into one L2 instead of eight (263 -> ~150 MB of L2 misses per launch)
            const int x = bid & 7, first = nbL + ((x - (nbL & 7) + 8) & 7), run = (v.n_fc + 7) >> 3, within = (bid - first) >> 3;
            hc = (bid >= first && within < run) ? x * run + within : v.n_fc;
            if (hc >= v.n_fc) {
                if (tid == 0) v.partial[bid] = 0.0;
                return;
            }
        }
        // the workgroup's camera is the same in every lane: its pose and focal length live in scalar registers (14 + 2 of them instead
        // of 16 vector registers per lane -- what the 27 accumulators of the loop below were being spilled for)
        const int cam = __builtin_amdgcn_readfirstlane(v.hc2cam[hc]);
        PoseD P = load_pose(v.pose_cur, cam);
        P.qx = uniform_f64(P.qx); P.qy = uniform_f64(P.qy); P.qz = uniform_f64(P.qz); P.qw = uniform_f64(P.qw);
        P.tx = uniform_f64(P.tx); P.ty = uniform_f64(P.ty); P.tz = uniform_f64(P.tz);
        const double f = uniform_f64(v.camK[cam * 4]);
        if (v.compact && quarter == 0 && tid == 0) {
            double R[9];
            q_to_R(P.qx, P.qy, P.qz, P.qw, R);
            double* cr = v.camR + (size_t)hc * 12;
            cr[0] = f;
#pragma unroll
            for (int k = 0; k < 9; ++k) cr[1 + k] = R[k];
        }
        double A[21], b[6];
#pragma unroll
        for (int k = 0; k < 21; ++k) A[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = 0;
        const int e0 = v.camE_ptr[hc], ne = v.camE_ptr[hc + 1] - e0, per = (ne + cpc - 1) / cpc;
        const int q0 = e0 + min(quarter * per, ne), q1 = e0 + min((quarter + 1) * per, ne);
        for (int idx = q0 + wave * WAVE + lane; idx < q1; idx += 4 * WAVE) {
            const int i = v.camE[idx];
            if (!v.L_active[i]) continue;
            const int pt = v.L_pt[i];
            const double2 xy = *reinterpret_cast<const double2*>(v.pt_cur + (size_t)pt * 4);
            const double Z = v.pt_cur[(size_t)pt * 4 + 2];
            EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
            const double info = (double)v.L_info[i];
            double rho0, rho1;
            huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
            if (!v.points_free) { chi += rho0; *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1); }
            const double w = info * rho1;
            const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
            double Jc[12];
            jac_pose(g, f, Jc);
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                b[a] += Jc[a] * r0 + Jc[6 + a] * r1;
#pragma unroll
                for (int c = 0; c <= a; ++c) A[k++] += Jc[a] * w * Jc[c] + Jc[6 + a] * w * Jc[6 + c];
            }
        }
        {   // the 27 sums of the block in one packed butterfly per wavefront (wave_sum_packed: 29 shuffles instead of 162, the same bits)
            double s27[27];
#pragma unroll
            for (int k = 0; k < 21; ++k) s27[k] = A[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) s27[21 + k] = b[k];
            const double tot = wave_sum_packed<27>(s27, lane);
            const int slot = wave_sum_slot(lane);
            if ((lane & 1) == 0 && slot < 27) part[wave][slot] = tot;
        }
        __syncthreads();
        if (cpc > 1) {
            if (tid < 27) cam_part[(size_t)(bid - nbL) * 28 + tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        } else if (tid == 0) {
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c <= a; ++c) {
                    const double val = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
                    v.U[(size_t)hc * 36 + a * 6 + c] = val; v.U[(size_t)hc * 36 + c * 6 + a] = val; ++k;
                }
#pragma unroll
            for (int a = 0; a < 6; ++a) v.bc[(size_t)hc * 6 + a] = ((part[0][21 + a] + part[1][21 + a]) + part[2][21 + a]) + part[3][21 + a];
        }
    } else if (zero_role) {
        // ---- zero role: S (n_pad x n_pad), y, identity on the padded tail of the diagonal
        const int n = v.n_fc * 6, np = v.n_pad;
        for (int i = tid; i < np * np; i += 256) v.S[i] = 0.0;
        for (int i = tid; i < np; i += 256) v.y[i] = 0.0;
        __syncthreads();
        for (int i = n + tid; i < np; i += 256) v.S[(size_t)i * np + i] = 1.0;
    }
    const double r = block_sum<4>(chi, sm);
    // Large problems (cpc == 1): the "last block" pattern needs a release fence per block -- a write-back of the XCD's L2 -- and with
    // thousands of blocks streaming W through those L2s that costs more than the whole kernel; a write-through form (round 4) cost what
    // the k_reduce_sum launch it saved costs.  Their partials are added by that launch (or by k_schur_prepare, which follows).
    if (tid == 0) v.partial[bid] = r;
    if (cpc == 1) return;
    if (!last_block_arrives(counter, n_blocks)) return;
    fold_partials(v.partial, n_blocks, v.scal + SC_CHI, sm);
    // U and b_c of every camera from its quarters, in order
    for (int e = tid; e < v.n_fc * 27; e += 256) {
        const int hc = e / 27, k = e % 27;
        const double* q = cam_part + (size_t)hc * SMALL_CPC * 28 + k;
        const double val = ((q[0] + q[28]) + q[56]) + q[84];
        if (k < 21) {
            int a = 0, rem = k;                          // k = a (a + 1) / 2 + c, c <= a
            while (rem > a) { rem -= a + 1; ++a; }
            const int c = rem;
            v.U[(size_t)hc * 36 + a * 6 + c] = val; v.U[(size_t)hc * 36 + c * 6 + a] = val;
            if (a == c) udiag[hc * 6 + a] = val;
        } else v.bc[(size_t)hc * 6 + (k - 21)] = val;
    }
    __syncthreads();
    if (want_maxdiag) {
        double m = 0;
        const int nU = v.n_fc * 6, nV = v.points_free ? v.n_lm * 3 : 0;
        for (int i = tid; i < nU; i += 256) m = fmax(m, fabs(udiag[i]));
        for (int i = tid; i < nV; i += 256) {
            const int l = i / 3, d = i % 3;
            m = fmax(m, fabs(v.V[(size_t)l * 6 + (d == 0 ? 0 : d == 1 ? 3 : 5)]));
        }
        m = wave_max(m);
        __syncthreads();
        if (lane == 0) sm[wave] = m;
        __syncthreads();
        if (tid == 0) v.scal[SC_MAXDIAG] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    }
}

// blocks [0, n_blk): one workgroup per 6x6 block of S (k_schur_block<true> with D^-1 formed in place);
// blocks [n_blk, n_blk + n_fc): the reduced rhs of one camera (k_schur_rhs<true>, D^-1 b_p formed in place)
__global__ __launch_bounds__(256) void k_small_schur(BaDeviceView v, double lambda)
{
    if (lambda < 0) lambda = 1e-5 * v.scal[SC_MAXDIAG];      // first trial after a (re-)initialisation: g2o's tau * max |diag| (A.4), seeded without a host round trip

    __shared__ double red[4][64 * 37];
    __shared__ double part[4][36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = wave * WAVE + lane, stride = 4 * WAVE;
    if ((int)blockIdx.x < v.n_blk) {
        const int b = blockIdx.x;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0;
        // the index chain (contribution -> slots -> landmark) of the NEXT contribution is fetched while this one is computed: a
        // thread has ~5 contributions and nothing else on its compute unit hides three dependent loads each.  (Issuing the loads
        // of all of a thread's contributions in bulk, 8-way unrolled, was slower: 44 us against 34 us.)
        const int c_end = v.blk_ptr[b + 1];
        int c = v.blk_ptr[b] + first;
        int2 sab_next = make_int2(0, 0);
        int l_next = 0;
        if (c < c_end) { sab_next = v.con[c]; l_next = v.w_lm[sab_next.x]; }
        for (; c < c_end; c += stride) {
            const int2 sab = sab_next;
            const int l_cur = l_next;
            if (c + stride < c_end) { sab_next = v.con[c + stride]; l_next = v.w_lm[sab_next.x]; }
            const double2* Wa2 = reinterpret_cast<const double2*>(v.W + (size_t)sab.x * 18);
            const double2* Wb2 = reinterpret_cast<const double2*>(v.W + (size_t)sab.y * 18);
            double D[6];
            lm_dinv(v, l_cur, lambda, D);
            const double d00 = D[0], d01 = D[1], d02 = D[2], d11 = D[3], d12 = D[4], d22 = D[5];
            double wa[18], wb[18];
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = Wa2[k]; wa[2 * k] = t.x; wa[2 * k + 1] = t.y; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = Wb2[k]; wb[2 * k] = t.x; wb[2 * k + 1] = t.y; }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double a0 = wa[r * 3], a1 = wa[r * 3 + 1], a2 = wa[r * 3 + 2];
                const double t0 = a0 * d00 + a1 * d01 + a2 * d02;
                const double t1 = a0 * d01 + a1 * d11 + a2 * d12;
                const double t2 = a0 * d02 + a1 * d12 + a2 * d22;
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) acc[r * 6 + cc] += t0 * wb[cc * 3] + t1 * wb[cc * 3 + 1] + t2 * wb[cc * 3 + 2];
            }
        }
        double* R = red[wave];
#pragma unroll
        for (int k = 0; k < 36; ++k) R[lane * 37 + k] = acc[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int2 ij = v.blk_ij[b];
        double val = 0;
        if (lane < 36) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                s0 += R[(j + 0) * 37 + lane]; s1 += R[(j + 1) * 37 + lane]; s2 += R[(j + 2) * 37 + lane]; s3 += R[(j + 3) * 37 + lane];
            }
            val = (s0 + s1) + (s2 + s3);
            part[wave][lane] = val;
        }
        __syncthreads();
        if (wave == 0 && lane < 36) {
            val = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
            const int r = lane / 6, c = lane % 6;
            if (ij.x == ij.y) {
                const double u = v.U[(size_t)ij.x * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
                v.S[(size_t)(ij.x * 6 + c) * v.n_pad + (ij.x * 6 + r)] = u - val;
            } else {
                v.S[(size_t)(ij.x * 6 + r) * v.n_pad + (ij.y * 6 + c)] = -val;
            }
        }
        return;
    }
    const int hc = (int)blockIdx.x - v.n_blk;
    if (hc >= v.n_fc) return;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int idx = v.camS_ptr[hc] + first; idx < v.camS_ptr[hc + 1]; idx += stride) {
        const int s = v.camS[idx];
        const double* W = v.W + (size_t)s * 18;
        const int l = v.w_lm[s];
        double D[6];
        lm_dinv(v, l, lambda, D);
        const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
        const double d0 = D[0] * b0 + D[1] * b1 + D[2] * b2;
        const double d1 = D[1] * b0 + D[3] * b1 + D[4] * b2;
        const double d2 = D[2] * b0 + D[4] * b1 + D[5] * b2;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[r] += W[r * 3] * d0 + W[r * 3 + 1] * d1 + W[r * 3 + 2] * d2;
    }
    {   // six sums, one packed butterfly (9 shuffles instead of 36, the same bits)
        const double tot = wave_sum_packed<6>(acc, lane);
        const int slot = wave_sum_slot(lane);
        if ((lane & 1) == 0 && slot < 6) part[wave][slot] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int r = threadIdx.x;
        const double a = ((part[0][r] + part[1][r]) + part[2][r]) + part[3][r];
        v.y[hc * 6 + r] = v.bc[(size_t)hc * 6 + r] - a;
    }
}

// blocks [0, nbL): k_backsub with D^-1 formed in place; blocks [nbL, nbL + nbC): k_pose_update; last block adds the scale partials
__global__ __launch_bounds__(256) void k_small_update(BaDeviceView v, double lambda, int nbL, int* __restrict__ counter)
{
    if (lambda < 0) lambda = 1e-5 * v.scal[SC_MAXDIAG];      // first trial after a (re-)initialisation: g2o's tau * max |diag| (A.4), seeded without a host round trip

    __shared__ double sm[4];
    const int bid = blockIdx.x;
    double sc = 0;
    if (bid < nbL) {
        const int l = bid * 256 + threadIdx.x;
        if (l < v.n_lm) {
            const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
            double c0 = b0, c1 = b1, c2 = b2;
            for (int s = v.lm_wptr[l]; s < v.lm_wptr[l + 1]; ++s) {
                const double2* W2 = reinterpret_cast<const double2*>(v.W + (size_t)s * 18);
                const double2* x2 = reinterpret_cast<const double2*>(v.xc + (size_t)v.w_hc[s] * 6);
                double W[18], xx[6];
#pragma unroll
                for (int k = 0; k < 9; ++k) { const double2 t = W2[k]; W[2 * k] = t.x; W[2 * k + 1] = t.y; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double2 t = x2[k]; xx[2 * k] = t.x; xx[2 * k + 1] = t.y; }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const double mx = -xx[r];
                    c0 += W[r * 3] * mx; c1 += W[r * 3 + 1] * mx; c2 += W[r * 3 + 2] * mx;
                }
            }
            double D[6];
            lm_dinv(v, l, lambda, D);
            const double x0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
            const double x1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
            const double x2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
            const int pt = v.lm_pt[l];
            const double* pc = v.pt_cur + (size_t)pt * 4;
            double* pt_t = v.pt_trial + (size_t)pt * 4;
            pt_t[0] = pc[0] + x0; pt_t[1] = pc[1] + x1; pt_t[2] = pc[2] + x2;
            sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
    } else {
        const int hc = (bid - nbL) * 256 + threadIdx.x;
        if (hc < v.n_fc) sc = pose_update_one(v, lambda, hc);
    }
    const double r = block_sum<4>(sc, sm);
    if (threadIdx.x == 0) v.partial[bid] = r;
    if (!last_block_arrives(counter, gridDim.x)) return;
    fold_partials(v.partial, gridDim.x, v.scal + SC_SCALE, sm);
}

// k_error with the reduction folded in: last block adds the partials
__global__ __launch_bounds__(256) void k_small_error(BaDeviceView v, int trial, double delta, int* __restrict__ counter)
{
    __shared__ double sm[4];
    const double* pose = trial ? v.pose_trial : v.pose_cur;
    const double* pts = trial ? v.pt_trial : v.pt_cur;
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_L; i += gridDim.x * 256) {
        if (!v.L_active[i]) continue;
        const int cam = v.L_cam[i], pt = v.L_pt[i];
        PoseD P = load_pose(pose, cam);
        const double2 xy = *reinterpret_cast<const double2*>(pts + (size_t)pt * 4);
        const double Z = pts[(size_t)pt * 4 + 2];
        EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
        *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
        double rho0, rho1;
        huber((double)v.L_info[i] * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
        acc += rho0;
    }
    const double r = block_sum<4>(acc, sm);
    if (threadIdx.x == 0) v.partial[blockIdx.x] = r;
    if (!last_block_arrives(counter, gridDim.x)) return;
    fold_partials(v.partial, gridDim.x, v.scal + (trial ? SC_CHI_TRIAL : SC_CHI), sm);
}

// k_small_update + k_small_error in one launch (points free).  Every workgroup recomputes the trial poses of the <= 21 free cameras
// into LDS (workgroup 0 also stores them); SMALL_LPL lanes per landmark share its W slots for the back-substitution (fixed shuffle
// tree, as k_backsub), every lane then holds the trial point and evaluates its share of the landmark's observations against the
// trial poses -- the residuals k_small_error would have read the trial state back for.  The last workgroup folds both sums.
__global__ __launch_bounds__(256) void k_small_update_error(BaDeviceView v, double lambda, double delta, int* __restrict__ counter)
{
    if (lambda < 0) lambda = 1e-5 * v.scal[SC_MAXDIAG];      // first trial after a (re-)initialisation: g2o's tau * max |diag| (A.4), seeded without a host round trip

    __shared__ double sm[4];
    __shared__ PoseD lp[24];
    const int bid = blockIdx.x, tid = threadIdx.x, n_blocks = gridDim.x;
    double sc = 0, chi = 0;
    if (tid < v.n_fc) {
        PoseD O;
        const double s = pose_update_compute(v, lambda, tid, O);
        lp[tid] = O;
        if (bid == 0) { store_pose(v.pose_trial, v.hc2cam[tid], O); sc = s; }
    }
    __syncthreads();
    const int gl = bid * 256 + tid, l = gl / SMALL_LPL, sub = gl % SMALL_LPL;
    const bool live = l < v.n_lm;
    double c0 = 0, c1 = 0, c2 = 0;
    if (live) {
        const int s1 = v.lm_wptr[l + 1];
        for (int s = v.lm_wptr[l] + sub; s < s1; s += SMALL_LPL) {
            const double2* W2 = reinterpret_cast<const double2*>(v.W + (size_t)s * 18);
            const double2* x2 = reinterpret_cast<const double2*>(v.xc + (size_t)v.w_hc[s] * 6);
            double W[18], xx[6];
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double2 t = W2[k]; W[2 * k] = t.x; W[2 * k + 1] = t.y; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double2 t = x2[k]; xx[2 * k] = t.x; xx[2 * k + 1] = t.y; }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double mx = -xx[r];
                c0 += W[r * 3] * mx; c1 += W[r * 3 + 1] * mx; c2 += W[r * 3 + 2] * mx;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < SMALL_LPL; m <<= 1) { c0 += __shfl_xor(c0, m, 64); c1 += __shfl_xor(c1, m, 64); c2 += __shfl_xor(c2, m, 64); }
    if (live) {
        const double b0 = v.bp[(size_t)l * 4], b1 = v.bp[(size_t)l * 4 + 1], b2 = v.bp[(size_t)l * 4 + 2];
        c0 += b0; c1 += b1; c2 += b2;
        double D[6];
        lm_dinv(v, l, lambda, D);
        const double x0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
        const double x1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
        const double x2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
        const int pt = v.lm_pt[l];
        const double* pc = v.pt_cur + (size_t)pt * 4;
        const double X = pc[0] + x0, Y = pc[1] + x1, Z = pc[2] + x2;
        if (sub == 0) {
            double* pt_t = v.pt_trial + (size_t)pt * 4;
            pt_t[0] = X; pt_t[1] = Y; pt_t[2] = Z;
            sc += x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
        const int end = v.lm_ptr[l + 1];
        for (int i = v.lm_ptr[l] + sub; i < end; i += SMALL_LPL) {
            if (!v.L_active[i]) continue;
            const int cam = v.L_cam[i], hc = v.cam2hc[cam];
            const PoseD P = hc >= 0 ? lp[hc] : load_pose(v.pose_cur, cam);       // a fixed camera's trial pose is its pose
            EdgeGeom g = edge_geom(P, v.camK, cam, X, Y, Z, v.L_uv[i]);
            *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
            double rho0, rho1;
            huber((double)v.L_info[i] * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
            chi += rho0;
        }
    }
    const double r0 = block_sum<4>(sc, sm);
    const double r1 = block_sum<4>(chi, sm);
    if (tid == 0) { v.partial[bid] = r0; v.partial[n_blocks + bid] = r1; }
    if (!last_block_arrives(counter, n_blocks)) return;
    fold_partials(v.partial, n_blocks, v.scal + SC_SCALE, sm);
    __syncthreads();
    fold_partials(v.partial + n_blocks, n_blocks, v.scal + SC_CHI_TRIAL, sm);
}

// k_classify with the three reductions folded in
// result != nullptr: the KEPT estimate (n_cams x 8 pose doubles, then n_pts x 4) is also written there -- right behind the scalars, so
// that it rides their read-back and the caller's GetPose / GetPoint loop needs no trip to the device of its own (round 4: the
// reference reads every pose and point back after every bundler, BundleAdjust.cpp:318-347; that trip was 20 of a 2 000-observation
// bundler's 167 us)
// PUBLISH (round 4, the queued post-pass): what the host reads after the call -- the scalars, the kept estimate behind them, the first
// outlier ids behind that: one contiguous run starting at v.scal -- is written into the pinned mirror by the launch itself, so no copy
// command (a blit launch and a dependency, ~5 us of a 150 us local-BA call) is queued behind it.  Loads that skip this unit's cache: the
// scalars were read when the launch began and rewritten since, result and ids come from every workgroup.
__device__ __forceinline__ void publish_run(const double* __restrict__ src, double* __restrict__ mirror, int n_doubles)
{
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(mirror);
    for (int i = threadIdx.x; i < n_doubles; i += 256) d[i] = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool AFTER_TRIAL>
__global__ __launch_bounds__(256) void k_small_classify(BaDeviceView v, double max_err_sq, uint32_t* __restrict__ out_ids, int* __restrict__ out_count,
                                                        int out_base, int* __restrict__ counter, ClassifyAfterTrial spec, double* __restrict__ result,
                                                        double* __restrict__ mirror, int mirror_scalars, int mirror_doubles, int ids_prefix)
{
    __shared__ double sm[4];
    const double* pose_kept = v.pose_cur;
    const double* pt_kept = v.pt_cur;
    if (AFTER_TRIAL) {
        bool accept;
        const bool fin = call_is_over(v, spec, accept);
        if (blockIdx.x == 0 && threadIdx.x == 0) v.scal[SC_SPEC_DONE] = fin ? 1.0 : 0.0;
        if (!fin) {                                         // every workgroup alike: the arrival counter stays untouched
            if (mirror && blockIdx.x == 0) {               // the host still reads the trial's scalars
                __threadfence();
                __syncthreads();
                publish_run(v.scal, mirror, mirror_scalars);
            }
            return;
        }
        if (accept) { pose_kept = v.pose_trial; pt_kept = v.pt_trial; }
    }
    if (result) {
        const int np8 = v.n_cams * 8, nt4 = v.n_pts * 4;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < np8 + nt4; i += gridDim.x * 256) result[i] = i < np8 ? pose_kept[i] : pt_kept[i - np8];
    }
    double es = 0, ec = 0, no = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_L; i += gridDim.x * 256) {
        if (!v.L_active[i]) continue;
        const double2 e = *reinterpret_cast<const double2*>(v.errL + (size_t)i * 2);
        const double ss = e.x * e.x + e.y * e.y;
        const int cam = v.L_cam[i], pt = v.L_pt[i];
        PoseD P = load_pose(pose_kept, cam);
        double wx, wy, wz, fx, fy, fz;
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, -P.tx, -P.ty, -P.tz, wx, wy, wz);
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, 0.0, 0.0, 1.0, fx, fy, fz);
        const double* X = pt_kept + (size_t)pt * 4;
        const double dot = (X[0] - wx) * fx + (X[1] - wy) * fy + (X[2] - wz) * fz;
        const bool out = (dot <= 0) || (ss > max_err_sq);
        if (out) { no += 1.0; v.L_active[i] = 0; out_ids[atomicAdd(out_count, 1) - out_base] = v.L_edge[i]; }
        else { es += ss; ec += 1.0; }
    }
    const int nb = gridDim.x;
    const double r0 = block_sum<4>(es, sm);
    const double r1 = block_sum<4>(ec, sm);
    const double r2 = block_sum<4>(no, sm);
    if (threadIdx.x == 0) { v.partial[blockIdx.x] = r0; v.partial[nb + blockIdx.x] = r1; v.partial[2 * nb + blockIdx.x] = r2; }
    if (!last_block_arrives(counter, nb)) return;
    fold_partials(v.partial, nb, v.scal + SC_ERRSUM, sm);
    fold_partials(v.partial + nb, nb, v.scal + SC_ERRCNT, sm);
    fold_partials(v.partial + 2 * nb, nb, v.scal + SC_NOUT, sm);
    if (mirror) {      // scalars + kept estimate, and as many of the ids behind them as there are (up to the prefix the host reads)
        __threadfence();
        __syncthreads();
        const int n_ids = min(max(__hip_atomic_load(out_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - out_base, 0), ids_prefix);
        publish_run(v.scal, mirror, mirror_doubles + (n_ids + 1) / 2);
    }
}

// =================================================================================================
// POSE-ONLY problems (BundlerParameters::ArePointsFixed, a handful of free cameras, a few thousand observations): the tracker's
// per-frame refinement (Tracking/TrackLocalMap.cpp:421-501 OptimizeCameraPose, Tracking/PoseEstimator.cpp:168-207).  With the
// points fixed the system is one 6x6 block per camera, so the WHOLE StepBundleAdjustment call -- every LM iteration with its
// damped trials (OptimizationAlgorithmLevenberg::solve, appendix A.4), and the outlier post-pass (BundlerLib.cpp:384-427) -- runs
// in ONE launch of one workgroup; the host reads one record back.  Control flow and arithmetic per element are those of
// lm_solve / the kernels above; sums keep a fixed order.
// =================================================================================================
// 1 / sqrt(d) to full double precision for d > 0 (v_rsq_f64 + two Goldschmidt steps), and sqrt(d) = d * that: a pivot of the 6 x 6
// factorisation below costs ~12 dependent operations instead of the ~40 of an IEEE sqrt followed by a division -- on the pose-only
// kernel's lone thread a dependent f64 operation is ~25 cycles, and this solve runs once per LM trial.
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double& s, double& rs)
{
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    s = g; rs = h + h;
}
__device__ __forceinline__ bool solve6_spd(const double* __restrict__ U36, double lambda, const double* __restrict__ b, double* __restrict__ x)
{
    double L[6][6], inv[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = U36[j * 6 + j] + lambda;
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) ok = false;
        double sq, rs;
        sqrt_and_rsqrt(d, sq, rs);
        L[j][j] = sq; inv[j] = rs;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double a = U36[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k];
            L[i][j] = a * rs;
        }
    }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double a = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) a -= L[i][k] * z[k];
        z[i] = a * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double a = z[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) a -= L[k][i] * x[k];
        x[i] = a * inv[i];
    }
    return ok;
}

// STAGED (round 4, the frame path of ba_host.hip): everything the solve touches is copied into LDS first and only the two pose
// buffers go back at the end.  With the arrays in HBM every LM trial is six to eight DEPENDENT memory round trips of one workgroup
// (residuals, linearisation, the 6x6 system written and read back, the trial's residuals): 47-63 us for the tracker's 3-4 iterations
// on 300 observations, where one CPU core needs 33-47 us for the whole call.  Same code, same order of every sum: bit-identical.
size_t pose_lm_staged_bytes(int nc, int np, int nL, int nfc)
{
    return ((size_t)nc * 20 + (size_t)np * 4 + (size_t)nL * 2 + (size_t)nfc * 48) * 8 + (size_t)nL * (8 + 4 * 4) + (size_t)(2 * nfc + 1) * 4 + (size_t)nL + 64;
}

#ifdef POSE_LM_CLOCKS      // development build only (tools/pose_lm_clocks.sh): thread 0 prints the shader clock at the phases of the solve
#define PLC(tag) do { if (tid == 0 && plc_n < 48) { plc_t[plc_n] = clock64(); plc_tag[plc_n++] = tag; } } while (0)
#else
#define PLC(tag) do { } while (0)
#endif
// NW wavefronts: the tracker's 300 observations are one pass of 512 threads where 256 took two (the second 17 % full); two wavefronts per
// SIMD interleave their dependent f64 chains almost for free.
constexpr int POSE_LM_NW = 8;
template <bool STAGED, int NW>
__global__ __launch_bounds__(64 * NW) void k_pose_lm(BaDeviceView vin, PoseLmArgs a, PoseLmResult* __restrict__ out, uint8_t* __restrict__ flagL, double* __restrict__ out_pose)
{
    extern __shared__ __align__(16) double dyn[];
    constexpr int NT = 64 * NW;
    __shared__ double sm[NW];
    __shared__ double part[NW][28];
    __shared__ double s_lambda, s_ni, s_rho, s_cur_chi;
    __shared__ int s_ok, s_accept;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    BaDeviceView v = vin;
#ifdef POSE_LM_CLOCKS
    long long plc_t[48]; int plc_tag[48]; int plc_n = 0;
#endif
    PLC(0);
    if (STAGED) {
        const int nc = vin.n_cams, np = vin.n_pts, nL = vin.n_L, nfc = vin.n_fc;
        // LDS: the caller's image first, byte for byte -- [pose0 | pose1 | camK | pt | L_uv | L_info | L_cam | L_pt | camE | camE_ptr |
        // hc2cam | L_active] (ba_host.hip keeps it in exactly this order, packed) -- then the solve's own arrays.  ONE flat copy in 16-byte
        // pieces, eight loads of a thread in flight before the first store: the image costs one trip to memory -- or across PCIe, when
        // the pointer is the pinned image itself -- where array-by-array loops cost one trip each (4.9 us from HBM, 14 us from the
        // host for the tracker's 300 observations).
        double* d = dyn;
        double* pose0 = d; d += (size_t)nc * 8;
        double* pose1 = d; d += (size_t)nc * 8;
        double* camK = d; d += (size_t)nc * 4;
        double* pt = d; d += (size_t)np * 4;
        float2* L_uv = reinterpret_cast<float2*>(d);
        float* L_info = reinterpret_cast<float*>(L_uv + nL);
        uint32_t* L_cam = reinterpret_cast<uint32_t*>(L_info + nL);
        uint32_t* L_pt = L_cam + nL;
        int* camE = reinterpret_cast<int*>(L_pt + nL);
        int* camE_ptr = camE + nL;
        int* hc2cam = camE_ptr + nfc + 1;
        uint8_t* L_active = reinterpret_cast<uint8_t*>(hc2cam + nfc);
        const int image_bytes = (nc * 20 + np * 4) * 8 + nL * 25 + (2 * nfc + 1) * 4;
        const int n16 = (image_bytes + 15) / 16;
        d = dyn + (size_t)n16 * 2;
        double* errL = d; d += (size_t)nL * 2;
        double* U = d; d += (size_t)nfc * 36;
        double* bc = d; d += (size_t)nfc * 6;
        double* xc = d; d += (size_t)nfc * 6;
        {
            const uint4* src = reinterpret_cast<const uint4*>(vin.pose_cur);
            uint4* dst = reinterpret_cast<uint4*>(dyn);
            for (int base = 0; base < n16; base += NT * 8) {
                uint4 r[8];          // (every load is issued -- past the end the last piece again -- so that r stays in registers: predicated loads sent it to scratch)
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = base + u * NT + tid; r[u] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = base + u * NT + tid; if (i < n16) dst[i] = r[u]; }
            }
        }
        v.pose_cur = pose0; v.pose_trial = pose1; v.camK = camK; v.pt_cur = v.pt_trial = pt; v.errL = errL; v.U = U; v.bc = bc; v.xc = xc;
        v.L_uv = L_uv; v.L_info = L_info; v.L_cam = L_cam; v.L_pt = L_pt; v.camE = camE; v.camE_ptr = camE_ptr; v.hc2cam = hc2cam; v.L_active = L_active;
        __syncthreads();
    }
    PLC(1);
    double lambda = a.lambda, ni = a.ni;
    int iteration = a.iteration, n_stats = 0, flips = 0, cont = 1;
    BaDeviceView w = v;                                 // w.pose_cur / w.pose_trial swap on every accepted trial

    // robust chi2 of state `pose` over all active observations (residuals to errL): thread-strided, fixed order
    auto chi2_of = [&](const double* pose, double delta) -> double {
        double acc = 0;
        for (int i = tid; i < v.n_L; i += NT) {
            if (!v.L_active[i]) continue;
            const int cam = v.L_cam[i], pt = v.L_pt[i];
            PoseD P = load_pose(pose, cam);
            const double2 xy = *reinterpret_cast<const double2*>(v.pt_cur + (size_t)pt * 4);
            const double Z = v.pt_cur[(size_t)pt * 4 + 2];
            EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
            *reinterpret_cast<double2*>(v.errL + (size_t)i * 2) = make_double2(g.e0, g.e1);
            double rho0, rho1;
            huber((double)v.L_info[i] * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
            acc += rho0;
        }
        return block_sum<NW>(acc, sm);
    };

    double carried_chi = 0;
    for (int it = 0; it < a.n_huber && cont; ++it) {
        const double delta = (double)a.huber[it];
        // ---- linearise at the current estimate: chi2, U, b per free camera
        // (with an unchanged Huber width the robust chi2 of the current estimate is the value the last iteration ended with -- the same
        // function on the same pose gives the same bits, accepted trial or not -- so one of the three passes over the observations per
        // iteration is saved; the residuals the post-pass reads are those of the LAST error evaluation either way, BundlerLib.cpp:386-425)
        const double chi_cur0 = (it > 0 && a.huber[it] == a.huber[it - 1]) ? carried_chi : chi2_of(w.pose_cur, delta);
        PLC(2);
        for (int hc = 0; hc < v.n_fc; ++hc) {
            const int cam = v.hc2cam[hc];
            PoseD P = load_pose(w.pose_cur, cam);
            const double f = v.camK[cam * 4];
            double A[21], b[6];
#pragma unroll
            for (int k = 0; k < 21; ++k) A[k] = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] = 0;
            for (int idx = v.camE_ptr[hc] + tid; idx < v.camE_ptr[hc + 1]; idx += NT) {
                const int i = v.camE[idx];
                if (!v.L_active[i]) continue;
                const int pt = v.L_pt[i];
                const double2 xy = *reinterpret_cast<const double2*>(v.pt_cur + (size_t)pt * 4);
                const double Z = v.pt_cur[(size_t)pt * 4 + 2];
                EdgeGeom g = edge_geom(P, v.camK, cam, xy.x, xy.y, Z, v.L_uv[i]);
                const double info = (double)v.L_info[i];
                double rho0, rho1;
                huber(info * (g.e0 * g.e0 + g.e1 * g.e1), delta, rho0, rho1);
                const double wgt = info * rho1;
                const double r0 = -info * g.e0 * rho1, r1 = -info * g.e1 * rho1;
                double Jc[12];
                jac_pose(g, f, Jc);
                int k = 0;
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    b[p] += Jc[p] * r0 + Jc[6 + p] * r1;
#pragma unroll
                    for (int c = 0; c <= p; ++c) A[k++] += Jc[p] * wgt * Jc[c] + Jc[6 + p] * wgt * Jc[6 + c];
                }
            }
            PLC(3);
            {   // the 27 sums of the block: one packed butterfly per wavefront, the wavefronts' partials added in wavefront order by 27 threads
                double s27[27];
#pragma unroll
                for (int k = 0; k < 21; ++k) s27[k] = A[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) s27[21 + k] = b[k];
                const double tot = wave_sum_packed<27>(s27, lane);
                const int slot = wave_sum_slot(lane);
                __syncthreads();
                if ((lane & 1) == 0 && slot < 27) part[wave][slot] = tot;
                __syncthreads();
                if (tid < 27) {
                    double val = part[0][tid];
#pragma unroll
                    for (int w2 = 1; w2 < NW; ++w2) val += part[w2][tid];          // wavefront order
                    if (tid < 21) {
                        int p = 0;
                        while ((p + 1) * (p + 2) / 2 <= tid) ++p;          // tid = p (p + 1) / 2 + c
                        const int c = tid - p * (p + 1) / 2;
                        v.U[(size_t)hc * 36 + p * 6 + c] = val; v.U[(size_t)hc * 36 + c * 6 + p] = val;
                    } else v.bc[(size_t)hc * 6 + (tid - 21)] = val;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        PLC(4);
        if (tid == 0) {
            if (iteration == 0) {
                double m = 0;
                for (int i = 0; i < v.n_fc * 6; ++i) m = fmax(m, fabs(v.U[(size_t)(i / 6) * 36 + (i % 6) * 7]));
                lambda = a.user_lambda > 0 ? a.user_lambda : 1e-5 * m;
                ni = 2;
            }
            s_lambda = lambda; s_ni = ni; s_cur_chi = chi_cur0;
        }
        __syncthreads();
        lambda = s_lambda; ni = s_ni;
        double cur_chi = s_cur_chi, rho = 0;
        int qmax = 0;
        do {
            // ---- one damped trial: x = (U + lambda I)^-1 b per camera, pose_trial = exp(x) pose_cur, scale, chi2 of the trial
            if (tid == 0) s_ok = 1;
            __syncthreads();
            double sc = 0;
            if (tid < v.n_fc) {
                double x[6];
                const bool ok = solve6_spd(v.U + (size_t)tid * 36, lambda, v.bc + (size_t)tid * 6, x);
#pragma unroll
                for (int k = 0; k < 6; ++k) v.xc[(size_t)tid * 6 + k] = x[k];
                if (!ok) s_ok = 0;
                sc = pose_update_one(w, lambda, tid);
            }
            const double scale = block_sum<NW>(sc, sm);
            __threadfence_block();
            __syncthreads();
            PLC(5);
            const double chi_trial = chi2_of(w.pose_trial, delta);
            PLC(6);
            if (tid == 0) {
                const bool ok2 = s_ok != 0;
                double temp = chi_trial, r;
                if (!ok2) { temp = 1.7976931348623157e308; r = -1.0; }
                else r = (cur_chi - temp) / (scale + 1e-3);
                int acc = 0;
                if (ok2 && r > 0 && isfinite(temp)) {
                    const double t3 = 2 * r - 1;
                    double alpha = 1. - t3 * t3 * t3;          // (2 rho - 1)^3 spelled out: pow() is a hundred dependent f64 operations on this lone thread
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    cur_chi = temp;
                    acc = 1;
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                s_rho = r; s_accept = acc; s_lambda = lambda; s_ni = ni; s_cur_chi = cur_chi;
            }
            __syncthreads();
            rho = s_rho; lambda = s_lambda; ni = s_ni; cur_chi = s_cur_chi;
            if (s_accept) { double* t = w.pose_cur; w.pose_cur = w.pose_trial; w.pose_trial = t; ++flips; }
            ++qmax;
            __syncthreads();
            PLC(7);
        } while (rho < 0 && qmax < 10);
        const int code = (qmax == 10 || rho == 0) ? 1 : 0;
        if (tid == 0 && n_stats < POSE_LM_MAX_ITERS) {
            out->stats[n_stats].code = code; out->stats[n_stats].trials = qmax; out->stats[n_stats].chi2_before = chi_cur0;
            out->stats[n_stats].chi2_after = cur_chi; out->stats[n_stats].lambda = lambda;
        }
        ++n_stats;
        ++iteration;
        cont = code == 0;
        carried_chi = cur_chi;
    }
    // ---- post-pass: classification with the residuals of the LAST error evaluation and the kept estimate (k_classify)
    PLC(8);
    double es = 0, ec = 0, no = 0;
    for (int i = tid; i < v.n_L; i += NT) {
        if (!v.L_active[i]) { flagL[i] = 0; continue; }
        const double2 e = *reinterpret_cast<const double2*>(v.errL + (size_t)i * 2);
        const double ss = e.x * e.x + e.y * e.y;
        const int cam = v.L_cam[i], pt = v.L_pt[i];
        PoseD P = load_pose(w.pose_cur, cam);
        double wx, wy, wz, fx, fy, fz;
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, -P.tx, -P.ty, -P.tz, wx, wy, wz);
        q_rot(-P.qx, -P.qy, -P.qz, P.qw, 0.0, 0.0, 1.0, fx, fy, fz);
        const double* X = v.pt_cur + (size_t)pt * 4;
        const double dot = (X[0] - wx) * fx + (X[1] - wy) * fy + (X[2] - wz) * fz;
        const bool o = (dot <= 0) || (ss > a.max_err_sq);
        flagL[i] = o ? 1 : 0;
        if (o) { no += 1.0; v.L_active[i] = 0; }
        else { es += ss; ec += 1.0; }
    }
    const double r0 = block_sum<NW>(es, sm);
    const double r1 = block_sum<NW>(ec, sm);
    const double r2 = block_sum<NW>(no, sm);
    if (tid == 0) {
        out->lambda = lambda; out->ni = ni; out->iteration = iteration; out->n_stats = n_stats; out->flips = flips;
        out->err_sum = r0; out->err_cnt = r1; out->n_out = r2;
    }
    // the two pose buffers leave for out_pose (the host picks the current one by `flips`); without one they are the caller's own arrays,
    // already in place (the staged form always has one: its inputs are read-only)
    if (out_pose) {
        __syncthreads();
        for (int i = tid; i < vin.n_cams * 8; i += NT) { out_pose[i] = v.pose_cur[i]; out_pose[vin.n_cams * 8 + i] = v.pose_trial[i]; }
    }
    PLC(9);
#ifdef POSE_LM_CLOCKS
    if (tid == 0) for (int i = 1; i < plc_n; ++i) printf("PLC %d %lld\n", plc_tag[i], plc_t[i] - plc_t[i - 1]);
#endif
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int RED_BLOCKS = 1024;   // grid-stride blocks for the streaming reductions (<= partial capacity / 3)

}  // namespace

void ba_launch_error(const BaDeviceView& v, bool trial, double delta, hipStream_t st)
{
    int nb = v.n_L > 0 ? (cdiv(v.n_L, 256) < RED_BLOCKS ? cdiv(v.n_L, 256) : RED_BLOCKS) : 1;
    hipLaunchKernelGGL(k_error, dim3(nb), dim3(256), 0, st, v, trial ? 1 : 0, delta, nb);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, st, v.partial, nb, 1, v.scal + (trial ? SC_CHI_TRIAL : SC_CHI), 1);
    tether_launch_error(v, trial, st);
}

void ba_launch_linearize(const BaDeviceView& v, double delta, hipStream_t st)
{
    if (v.points_free && v.n_lm > 0) hipLaunchKernelGGL(k_linearize_lm, dim3(cdiv(v.n_lm, 256)), dim3(256), 0, st, v, delta);
    if (v.n_fc > 0) {
        if (v.n_fc <= SPLIT_CAMERAS_BELOW) hipLaunchKernelGGL(k_linearize_cam<true>, dim3(v.n_fc), dim3(256), 0, st, v, delta);
        else hipLaunchKernelGGL(k_linearize_cam<false>, dim3(xcd_camera_grid(v.n_fc)), dim3(256), 0, st, v, delta);
    }
    tether_launch_linearize(v, st);
}

void ba_launch_maxdiag(const BaDeviceView& v, hipStream_t st, const double* udiag_sum)
{
    const int work = v.n_fc * 6 + (v.points_free ? v.n_lm * 3 : 0);
    const int nb = std::max(1, std::min(256, (work + 1023) / 1024));
    hipLaunchKernelGGL(k_maxdiag, dim3(nb), dim3(256), 0, st, v, udiag_sum);
    hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(256), 0, st, v.partial, nb, v.scal + SC_MAXDIAG);
}

void ba_launch_tile_envelope(const BaDeviceView& v, int* tile_env, hipStream_t st)
{
    const int nt = v.n_pad / 128;
    (void)hipMemsetAsync(tile_env, 0x7f, (size_t)nt * sizeof(int), st);
    const int n = std::max(nt, std::max(v.n_blk, v.n_tp));
    hipLaunchKernelGGL(k_tile_envelope, dim3(cdiv(std::max(n, 1), 256)), dim3(256), 0, st, v, tile_env, nt, 128);
}

size_t ba_packed_doubles(int n_pad) { return (size_t)n_pad * (n_pad + 128) / 2 + (size_t)n_pad; }
void ba_launch_pack_lower(const BaDeviceView& v, double* packed, bool to_packed, hipStream_t st)
{
    const dim3 grid(std::max(1, v.n_pad / 2048), v.n_pad + 1);
    if (to_packed) hipLaunchKernelGGL(k_pack_lower<true>, grid, dim3(256), 0, st, v.S, v.y, packed, v.n_pad, 128);
    else hipLaunchKernelGGL(k_pack_lower<false>, grid, dim3(256), 0, st, v.S, v.y, packed, v.n_pad, 128);
}
void ba_launch_gather_udiag(const BaDeviceView& v, double* out, hipStream_t st)
{
    if (v.n_fc > 0) hipLaunchKernelGGL(k_gather_udiag, dim3(cdiv(v.n_fc * 6, 256)), dim3(256), 0, st, v, out);
}
bool ba_launch_allreduce_local(double* const* bufs, int n, size_t count, int op, hipStream_t st)
{
    if (n < 1 || n > LOCAL_REDUCE_MAX) return false;
    if (count == 0) return true;
    LocalReduceArgs a{};
    for (int k = 0; k < n; ++k) a.buf[k] = bufs[k];
    a.n = n;
    const int nb = (int)std::min<size_t>(2048, (count + 255) / 256);
    hipLaunchKernelGGL(k_allreduce_local, dim3(nb), dim3(256), 0, st, a, count, op);
    return true;
}

void ba_launch_schur(const BaDeviceView& v, double lambda, hipStream_t st) { ba_launch_schur(v, lambda, lambda, 1.0, st); }
// lambda_cam: the damping of the camera blocks (lambda; 0 on the ranks of a landmark-sharded map that leave it to rank 0)
void ba_launch_schur(const BaDeviceView& v, double lambda, double lambda_cam, double pad_diag, hipStream_t st) { ba_launch_schur(v, lambda, lambda_cam, pad_diag, 0, st); }
// fold_n > 0: scal[SC_CHI] = the sum of the fold_n partials the linearisation left in v.partial (ba_fused_linearize with its fold deferred)
void ba_launch_schur(const BaDeviceView& v, double lambda, double lambda_cam, double pad_diag, int fold_n, hipStream_t st)
{
    const bool skyline = v.n_pad >= 1024 && v.tile_env;
    const bool merged = v.n_pad >= 1024;          // zero-fill of the skyline, landmark inverses and the chi2 fold in ONE launch (k_schur_prepare)
    if (v.n_pad >= 1024 && !skyline) hipLaunchKernelGGL(k_zero_lower, dim3(std::max(1, v.n_pad / 2048), v.n_pad), dim3(256), 0, st, v.S, v.n_pad, 128);
    else if (v.n_pad < 1024) (void)hipMemsetAsync(v.S, 0, (size_t)v.n_pad * v.n_pad * sizeof(double), st);
    // the diagonal blocks of k_schur_block also write their camera's reduced rhs; a camera without a block (no free landmark) keeps b_c
    const bool rhs_in_blocks = v.points_free && v.n_blk > 0;
    {
        const int nb_lm = (v.points_free && v.n_lm > 0) ? cdiv(v.n_lm, 256) : 0;
        const int nb_y = std::max(1, std::min(8, cdiv(v.n_pad, 256)));
        if (merged) {
            const int nt = v.n_pad / 128, n_zero = skyline ? nt * (nt + 1) / 2 : 0;
            hipLaunchKernelGGL(k_schur_prepare, dim3(nb_lm + nb_y + (fold_n > 0 ? 1 : 0) + n_zero), dim3(256), 0, st, v, lambda, nb_lm, nb_y, rhs_in_blocks ? 1 : 0, pad_diag, fold_n, n_zero, 128);
        } else {
            if (fold_n > 0) hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, st, v.partial, fold_n, 1, v.scal + SC_CHI, 1);
            hipLaunchKernelGGL(k_lm_invert, dim3(nb_lm + nb_y), dim3(256), 0, st, v, lambda, nb_lm, rhs_in_blocks ? 1 : 0, pad_diag);
        }
    }
    if (v.n_blk > 0 && v.compact) {
        if (v.n_blk <= SPLIT_BLOCKS_BELOW) hipLaunchKernelGGL(k_schur_block_compact<true>, dim3(v.n_blk), dim3(256), 0, st, v, lambda_cam);
        else if (v.stream_blks) hipLaunchKernelGGL(k_schur_stream, dim3(v.n_stream_groups), dim3(64 * STREAM_WAVES), 0, st, v, lambda_cam);
        else hipLaunchKernelGGL(k_schur_block_compact<false>, dim3(cdiv(v.n_blk_slots, SCHUR_WAVES)), dim3(64 * SCHUR_WAVES), 0, st, v, lambda_cam);
    } else if (v.n_blk > 0) {
        if (v.n_blk <= SPLIT_BLOCKS_BELOW) hipLaunchKernelGGL((k_schur_block<true, true>), dim3(v.n_blk), dim3(256), 0, st, v, lambda_cam);
        else hipLaunchKernelGGL((k_schur_block<false, true>), dim3(cdiv(v.n_blk_slots, SCHUR_WAVES)), dim3(64 * SCHUR_WAVES), 0, st, v, lambda_cam);
    }
    tether_launch_schur(v, st);
    if (v.n_fc > 0 && !rhs_in_blocks) {
        if (v.n_fc <= SPLIT_CAMERAS_BELOW) hipLaunchKernelGGL(k_schur_rhs<true>, dim3(v.n_fc), dim3(256), 0, st, v);
        else hipLaunchKernelGGL(k_schur_rhs<false>, dim3(xcd_camera_grid(v.n_fc)), dim3(256), 0, st, v);
    }
}

// Back-substitution + state update + the trial's chi2 for problems with free points: k_pose_update first (the trial poses), then
// k_backsub<true> (trial points, scale partials, the residuals of every observation against the trial state), one k_reduce_sum
// for both scalars, the tether edges' chi2 on top.  = ba_launch_update + ba_launch_error(trial) in three launches instead of five.
bool ba_update_and_trial_error_fuses(const BaDeviceView& v)
{
    return v.points_free && v.n_lm > 0 && v.n_fc > 0;
}
void ba_launch_update_and_trial_error(const BaDeviceView& v, double lambda, double lambda_cam, double delta, hipStream_t st)
{
    const int nb_l = cdiv(v.n_lm * BACKSUB_LPL, 256), nb_c = cdiv(v.n_fc, 256), n = nb_l + nb_c;
    hipLaunchKernelGGL(k_pose_update, dim3(nb_c), dim3(256), 0, st, v, lambda_cam, nb_l, n + nb_l);       // scale partials nb_l .. n, chi2 partials of those rows: zero
    hipLaunchKernelGGL(k_backsub<true>, dim3(nb_l), dim3(256), 0, st, v, lambda, delta, n);
    // partial = [scale: n][chi2: n] -> scal[SC_SCALE], scal[SC_SCALE + 6] = scal[SC_CHI_TRIAL] (folding the sums into the last block of their
    // producer was measured no faster than this launch: profiles/HISTORY.md)
    static_assert(SC_CHI_TRIAL - SC_SCALE == 6, "the two outputs of the fused reduction");
    hipLaunchKernelGGL(k_reduce_sum, dim3(2), dim3(256), 0, st, v.partial, n, SC_CHI_TRIAL - SC_SCALE, v.scal + SC_SCALE, 2);
    tether_launch_error(v, true, st);
}

void ba_launch_update(const BaDeviceView& v, double lambda, hipStream_t st) { ba_launch_update(v, lambda, lambda, st); }
void ba_launch_update(const BaDeviceView& v, double lambda, double lambda_cam, hipStream_t st)
{
    int nb_l = 0;
    if (v.points_free && v.n_lm > 0) {
        nb_l = cdiv(v.n_lm * BACKSUB_LPL, 256);
        hipLaunchKernelGGL(k_backsub<false>, dim3(nb_l), dim3(256), 0, st, v, lambda, 0.0, 0);
    }
    int nb_c = 0;
    if (v.n_fc > 0) {
        nb_c = cdiv(v.n_fc, 256);
        hipLaunchKernelGGL(k_pose_update, dim3(nb_c), dim3(256), 0, st, v, lambda_cam, nb_l, -1);
    }
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, st, v.partial, nb_l + nb_c, 1, v.scal + SC_SCALE, 1);
}

// ---- small-problem path (ba_kernels.h: ba_small_*)
// ONE predicate for the host's structure build (which skips the Schur block order for small problems) and for lm_solve (which then
// takes the small path): the two must never disagree, so both call this.
bool ba_small_shape_applies(int n_fc, int n_tethers, long long n_L)
{
    static const bool off = std::getenv("MAGE_BA_NO_SMALL_PATH") != nullptr;
    return !off && n_fc > 0 && n_fc * 6 <= 128 && n_tethers == 0 && n_L <= (1 << 20);
}
bool ba_small_applies(const BaDeviceView& v) { return ba_small_shape_applies(v.n_fc, v.n_T, v.n_L); }
static int small_error_blocks(const BaDeviceView& v) { return v.n_L > 0 ? std::min(cdiv(v.n_L, 256), RED_BLOCKS) : 1; }
void ba_small_linearize(const BaDeviceView& v, double delta, bool want_maxdiag, int* counter, hipStream_t st)
{
    const int nbL = (v.points_free && v.n_lm > 0) ? cdiv(v.n_lm * (v.dup_slots ? 1 : SMALL_LPL), 256) : 0;
    if (v.dup_slots) hipLaunchKernelGGL(k_small_linearize<true>, dim3(nbL + v.n_fc * SMALL_CPC + 1), dim3(256), 0, st, v, delta, nbL, want_maxdiag ? 1 : 0, counter, SMALL_CPC, 1);
    else hipLaunchKernelGGL(k_small_linearize<false>, dim3(nbL + v.n_fc * SMALL_CPC + 1), dim3(256), 0, st, v, delta, nbL, want_maxdiag ? 1 : 0, counter, SMALL_CPC, 1);
}
// Positions of the compact records (BaDeviceView::w_pos): camera-major = the inverse of camS, or the identity.
__global__ __launch_bounds__(256) void k_build_positions(BaDeviceView v, int* __restrict__ w_pos, int* __restrict__ pos_lm, int camera_major)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= v.n_w) return;
    const int s = camera_major ? v.camS[p] : p;
    w_pos[s] = p;
    pos_lm[p] = v.w_lm[s];
}
__global__ __launch_bounds__(256) void k_build_con_pos(BaDeviceView v, const int* __restrict__ w_pos, ConPos* __restrict__ con_pos, int* __restrict__ soa, int pitch)
{
    const int end = v.blk_ptr[v.n_blk];
    for (int c = blockIdx.x * 256 + threadIdx.x; c < end; c += gridDim.x * 256) {
        const int2 ab = v.con[c];
        const ConPos e = { w_pos[ab.x], w_pos[ab.y], v.w_lm[ab.x] };
        con_pos[c] = e;
        if (soa) { soa[c] = e.a; soa[pitch + c] = e.b; soa[2 * (size_t)pitch + c] = e.lm; }
    }
}
// slot_order: workgroup w of k_schur_block_compact lands on XCD w % 8 and takes entry w / 8 of that XCD's run of blocks (blk_order).  The
// run is in row order, so a long block (a diagonal one: every landmark of the camera) starts every ~20 entries, the last of them when
// the grid is almost drained.  Here every XCD's run is reordered longest first (stable counting sort by trips of 64 contributions):
// the same blocks on the same XCD, each block's sum untouched.  One workgroup per XCD.
__global__ __launch_bounds__(256) void k_order_slots_longest_first(BaDeviceView v, int* __restrict__ out)
{
    constexpr int NBIN = 64;
    __shared__ int cnt[NBIN * 256];        // [bin][thread]
    __shared__ int binbase[NBIN];
    const int x = blockIdx.x, tid = threadIdx.x;
    const int n = v.n_blk_slots > x ? (v.n_blk_slots - x + 7) / 8 : 0;
    const int per = (n + 255) / 256, e0 = min(tid * per, n), e1 = min(e0 + per, n);
    for (int k = 0; k < NBIN; ++k) cnt[k * 256 + tid] = 0;
    auto bin_of = [&](int b) -> int {
        if (b < 0) return NBIN - 1;                                       // empty slots last
        const int trips = (v.blk_ptr[b + 1] - v.blk_ptr[b] + 63) >> 6;
        return NBIN - 2 - min(trips, NBIN - 2);
    };
    for (int e = e0; e < e1; ++e) cnt[bin_of(v.blk_order[x + 8 * e]) * 256 + tid] += 1;
    __syncthreads();
    if (tid < NBIN) {                                                     // exclusive prefix inside the bin, in thread (= entry) order
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int c = cnt[tid * 256 + t]; cnt[tid * 256 + t] = run; run += c; }
        binbase[tid] = run;
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int k = 0; k < NBIN; ++k) { const int c = binbase[k]; binbase[k] = run; run += c; } }
    __syncthreads();
    for (int e = e0; e < e1; ++e) {
        const int b = v.blk_order[x + 8 * e], k = bin_of(b);
        const int rank = binbase[k] + cnt[k * 256 + tid];
        cnt[k * 256 + tid] += 1;
        out[x + 8 * rank] = b;
    }
}
int ba_schur_stream_groups(int n_cu) { return std::min(n_cu, STREAM_Q_MAX * 8) / 8 * 8; }
int ba_schur_stream_rounds(int n_blk_slots, int n_groups) { const int Q = n_groups / 8, n = (n_blk_slots + 7) / 8; return std::max(1, (n + Q - 1) / Q); }
void ba_launch_build_stream_lists(const BaDeviceView& v, int n_groups, int* group_blocks, int* group_ptr, void* blks, hipStream_t st)
{
    const int rounds = ba_schur_stream_rounds(v.n_blk_slots, n_groups);
    hipLaunchKernelGGL(k_assign_blocks, dim3(8), dim3(STREAM_Q_MAX), 0, st, v, n_groups / 8, rounds, group_blocks);
    hipLaunchKernelGGL(k_build_stream_lists, dim3(1), dim3(1024), 0, st, v, n_groups, rounds, group_blocks, group_ptr, static_cast<StreamBlk*>(blks));
}
void ba_launch_build_positions(const BaDeviceView& v, int* w_pos, int* pos_lm, ConPos* con_pos, int* slot_order, int* con_soa, int con_soa_pitch, hipStream_t st)
{
    if (v.n_w <= 0) return;
    if (v.n_blk_slots > 0 && slot_order) hipLaunchKernelGGL(k_order_slots_longest_first, dim3(8), dim3(256), 0, st, v, slot_order);
    hipLaunchKernelGGL(k_build_positions, dim3(cdiv(v.n_w, 256)), dim3(256), 0, st, v, w_pos, pos_lm, 1);
    if (v.n_blk > 0) hipLaunchKernelGGL(k_build_con_pos, dim3(2048), dim3(256), 0, st, v, w_pos, con_pos, con_soa, con_soa_pitch);
}

bool ba_compact_w_enabled()
{
    static const bool off = std::getenv("MAGE_BA_MATERIAL_W") != nullptr;
    return !off;
}
// The same kernel for LARGE problems in which every observation owns its W block: k_error + k_linearize_lm +
// k_linearize_cam in one launch -- the residuals are computed once, a landmark's observations by eight lanes, the chi2 folded by the
// last block.  S is cleared by ba_launch_schur, max |diag| comes from ba_launch_maxdiag.
bool ba_fused_linearize_applies(const BaDeviceView& v)
{
    return !v.dup_slots && v.n_fc > 0 && v.n_L > 0;
}
// defer_chi_fold: the chi2 partials stay in v.partial and their count is returned -- the caller hands it to the trial's FIRST
// ba_launch_schur, whose launch adds them (nothing reads scal[SC_CHI] before); only without tethers (their chi2 is added on top below)
int ba_fused_linearize(const BaDeviceView& v, double delta, int* counter, hipStream_t st, bool defer_chi_fold)
{
    const int nbL = (v.points_free && v.n_lm > 0) ? cdiv(v.n_lm * SMALL_LPL, 256) : 0;
    const int nbC = ((v.n_fc + 7) / 8) * 8 + 8;          // every XCD gets ceil(n_fc / 8) camera workgroups wherever its first one falls
    const bool defer = defer_chi_fold && v.n_T == 0 && v.n_pad >= 1024;
    if (v.dup_slots) hipLaunchKernelGGL(k_small_linearize<true>, dim3(nbL + nbC), dim3(256), 0, st, v, delta, nbL, 0, counter, 1, 0);
    else hipLaunchKernelGGL(k_small_linearize<false>, dim3(nbL + nbC), dim3(256), 0, st, v, delta, nbL, 0, counter, 1, 0);
    if (!defer) hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, st, v.partial, nbL + nbC, 1, v.scal + SC_CHI, 1);
    tether_launch_error(v, false, st);          // the pose-pose edges add their chi2, U and b_c on top (nothing is launched without them)
    tether_launch_linearize(v, st);
    return defer ? nbL + nbC : 0;
}
void ba_small_solve_trial(const BaDeviceView& v, double lambda, double delta, double* linv_ws, int* counter, hipStream_t st)
{
    const int n = v.n_fc * 6;
    hipLaunchKernelGGL(k_small_schur, dim3(v.n_blk + v.n_fc), dim3(256), 0, st, v, lambda);
    chol_small_solve(v.S, v.y, v.xc, n, v.n_pad, linv_ws, v.scal + SC_CHOL_OK, v.scal + SC_CHOL_STALL, st);
    if (v.points_free && v.n_lm > 0) {        // back-substitution, state update and the trial's residuals in one launch
        hipLaunchKernelGGL(k_small_update_error, dim3(cdiv(v.n_lm * SMALL_LPL, 256)), dim3(256), 0, st, v, lambda, delta, counter);
        return;
    }
    const int nbL = (v.points_free && v.n_lm > 0) ? cdiv(v.n_lm, 256) : 0;
    hipLaunchKernelGGL(k_small_update, dim3(nbL + cdiv(v.n_fc, 256)), dim3(256), 0, st, v, lambda, nbL, counter);
    hipLaunchKernelGGL(k_small_error, dim3(small_error_blocks(v)), dim3(256), 0, st, v, 1, delta, counter);
}
void ba_small_classify(const BaDeviceView& v, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, int* counter, double* result, hipStream_t st)
{
    hipLaunchKernelGGL(k_small_classify<false>, dim3(small_error_blocks(v)), dim3(256), 0, st, v, max_err_sq, out_ids, out_count, out_base, counter, ClassifyAfterTrial{}, result,
                       (double*)nullptr, 0, 0, 0);
}
void ba_small_classify_after_trial(const BaDeviceView& v, const ClassifyAfterTrial& c, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, int* counter, double* result,
                                   double* mirror, int mirror_scalars, int mirror_doubles, int ids_prefix, hipStream_t st)
{
    hipLaunchKernelGGL(k_small_classify<true>, dim3(small_error_blocks(v)), dim3(256), 0, st, v, max_err_sq, out_ids, out_count, out_base, counter, c, result,
                       mirror, mirror_scalars, mirror_doubles, ids_prefix);
}
static bool g_pose_lm_staged_ok = true;      // false when this device refused the staged kernel's LDS opt-in: the arrays then stay in HBM
void ba_small_init_device()
{
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_lm<true, POSE_LM_NW>), hipFuncAttributeMaxDynamicSharedMemorySize, POSE_LM_STAGED_MAX_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        g_pose_lm_staged_ok = false;
    }
}

bool ba_pose_lm_applies(const BaDeviceView& v, size_t n_huber)
{
    static const bool off = std::getenv("MAGE_BA_NO_SMALL_PATH") != nullptr;
    return !off && !v.points_free && v.n_T == 0 && v.n_fc >= 1 && v.n_fc <= 64 && v.n_L <= 16384 && n_huber >= 1 && n_huber <= (size_t)POSE_LM_MAX_ITERS;
}
void ba_launch_pose_lm(const BaDeviceView& v, const PoseLmArgs& a, PoseLmResult* out, uint8_t* flagL, double* out_pose, hipStream_t st)
{
    hipLaunchKernelGGL((k_pose_lm<false, POSE_LM_NW>), dim3(1), dim3(64 * POSE_LM_NW), 0, st, v, a, out, flagL, out_pose);
}
bool ba_pose_lm_staged_fits(const BaDeviceView& v)
{
    return g_pose_lm_staged_ok && pose_lm_staged_bytes(v.n_cams, v.n_pts, v.n_L, v.n_fc) <= (size_t)POSE_LM_STAGED_MAX_BYTES;
}
bool ba_launch_pose_lm_staged(const BaDeviceView& v, const PoseLmArgs& a, PoseLmResult* out, uint8_t* flagL, double* out_pose, hipStream_t st)
{
    const size_t lds = pose_lm_staged_bytes(v.n_cams, v.n_pts, v.n_L, v.n_fc);
    if (!g_pose_lm_staged_ok || lds > (size_t)POSE_LM_STAGED_MAX_BYTES || !out_pose) return false;
    hipLaunchKernelGGL((k_pose_lm<true, POSE_LM_NW>), dim3(1), dim3(64 * POSE_LM_NW), lds, st, v, a, out, flagL, out_pose);
    return true;
}

void ba_launch_export_poses(const double* pose, const uint32_t* cam, const uint32_t* row, size_t n, double* block, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_export_poses, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, st, pose, cam, row, n, block);
}
void ba_launch_import_poses(double* pose0, double* pose1, const uint32_t* cam, const uint32_t* row, size_t n, const double* block, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_import_poses, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, st, pose0, pose1, cam, row, n, block);
}

void ba_launch_classify(const BaDeviceView& v, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, hipStream_t st)
{
    int nb = v.n_L > 0 ? (cdiv(v.n_L, 256) < RED_BLOCKS ? cdiv(v.n_L, 256) : RED_BLOCKS) : 1;
    hipLaunchKernelGGL(k_classify<false>, dim3(nb), dim3(256), 0, st, v, max_err_sq, out_ids, out_count, out_base, nb, ClassifyAfterTrial{});
    hipLaunchKernelGGL(k_reduce_sum, dim3(3), dim3(256), 0, st, v.partial, nb, 1, v.scal + SC_ERRSUM, 3);
}
void ba_launch_classify_after_trial(const BaDeviceView& v, const ClassifyAfterTrial& c, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, hipStream_t st)
{
    int nb = v.n_L > 0 ? (cdiv(v.n_L, 256) < RED_BLOCKS ? cdiv(v.n_L, 256) : RED_BLOCKS) : 1;
    hipLaunchKernelGGL(k_classify<true>, dim3(nb), dim3(256), 0, st, v, max_err_sq, out_ids, out_count, out_base, nb, c);
    // (when the kernel found the call unfinished the three sums are stale: nobody reads them then)
    hipLaunchKernelGGL(k_reduce_sum, dim3(3), dim3(256), 0, st, v.partial, nb, 1, v.scal + SC_ERRSUM, 3);
}

}  // namespace mage
