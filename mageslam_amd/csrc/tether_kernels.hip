// tether_kernels.hip -- pose-pose constraint edges of the bundle-adjustment path (gfx950).
//
// Replaces (Dependencies/BundlerLib/Source/BundlerLib.cpp):
//   EdgeScaleConstraint     :24-54    e = (d - |t_b - t_a|) * w                      information 1
//   EdgeRotationConstraint  :56-90    e = angularDistance((T_a^-1 T_b).rot, q) * w   information 1
//   g2o EdgeSE3Expmap       :338-350  e = log(T_b^-1 * C * T_a)                      information w * I6
// The first two have no analytic Jacobian in the reference: g2o's BaseMultiEdge::linearizeOplus differentiates
// numerically, central differences of step 1e-9 through the manifold update exp(u) * T.  That amplifies every
// rounding difference of the error function by ~1e7, so this file is compiled with floating-point contraction OFF
// and spells its arithmetic in the same order as the CPU restatement (oracle/ba_oracle.c): with identical inputs the
// distance Jacobian is then bit-identical, and the rotation Jacobian differs only where atan2 differs by an ulp.
// The transform edge uses g2o's closed-form Jacobians Adj(T_b^-1 C), -Adj(T_a^-1 C^-1).
//
// There are few tethers per problem (one or two per keyframe of a stereo rig): one thread per tether, results
// parked per tether in HBM and then gathered per camera / per camera pair in list order, so the sums are deterministic.
#include "ba_kernels.h"

#pragma clang fp contract(off)

namespace mage {
namespace {

struct TP { double qx, qy, qz, qw, tx, ty, tz; };     // g2o::SE3Quat

__device__ __forceinline__ TP t_load(const double* __restrict__ p, int cam)
{
    const double* q = p + (size_t)cam * 8;
    TP T = { q[0], q[1], q[2], q[3], q[4], q[5], q[6] };
    return T;
}

// Eigen QuaternionBase::_transformVector
__device__ __forceinline__ void t_qrot(double qx, double qy, double qz, double qw, double vx, double vy, double vz,
                                       double& ox, double& oy, double& oz)
{
    double ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
    ux += ux; uy += uy; uz += uz;
    ox = vx + qw * ux + (qy * uz - qz * uy);
    oy = vy + qw * uy + (qz * ux - qx * uz);
    oz = vz + qw * uz + (qx * uy - qy * ux);
}

__device__ __forceinline__ void t_normalize(TP& T)   // SE3Quat::normalizeRotation
{
    if (T.qw < 0) { T.qx = -T.qx; T.qy = -T.qy; T.qz = -T.qz; T.qw = -T.qw; }
    const double n = sqrt(T.qx * T.qx + T.qy * T.qy + T.qz * T.qz + T.qw * T.qw);
    T.qx /= n; T.qy /= n; T.qz /= n; T.qw /= n;
}

__device__ __forceinline__ TP t_mul(const TP& a, const TP& b)   // SE3Quat::operator*
{
    TP r;
    double rx, ry, rz;
    t_qrot(a.qx, a.qy, a.qz, a.qw, b.tx, b.ty, b.tz, rx, ry, rz);
    r.tx = a.tx + rx; r.ty = a.ty + ry; r.tz = a.tz + rz;
    r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    t_normalize(r);
    return r;
}

__device__ __forceinline__ TP t_inverse(const TP& T)  // r' = conj(r), t' = r' * (t * -1)
{
    TP r;
    r.qx = -T.qx; r.qy = -T.qy; r.qz = -T.qz; r.qw = T.qw;
    t_qrot(r.qx, r.qy, r.qz, r.qw, T.tx * -1., T.ty * -1., T.tz * -1., r.tx, r.ty, r.tz);
    return r;
}

__device__ __forceinline__ void t_m3mul(const double A[9], const double B[9], double C[9])
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}

__device__ __forceinline__ void t_q_to_R(const TP& q, double R[9])
{
    double tx = 2 * q.qx, ty = 2 * q.qy, tz = 2 * q.qz;
    double twx = tx * q.qw, twy = ty * q.qw, twz = tz * q.qw;
    double txx = tx * q.qx, txy = ty * q.qx, txz = tz * q.qx;
    double tyy = ty * q.qy, tyz = tz * q.qy, tzz = tz * q.qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// SE3Quat::exp restricted to what the numeric differentiation needs: u = +-delta * e_d, |omega| < 1e-5 always,
// i.e. the small-angle branch R = I + Om + Om^2/2, V = I + Om/2 + Om^2/6.
__device__ __forceinline__ TP t_exp_small(const double u[6])
{
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double Om[9] = { 0, -w2, w1, w2, 0, -w0, -w1, w0, 0 };
    double Om2[9], R[9], V[9];
    t_m3mul(Om, Om, Om2);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double I = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
        R[k] = I + Om[k] + 0.5 * Om2[k];
        V[k] = I + 0.5 * Om[k] + (1.0 / 6.0) * Om2[k];
    }
    // Eigen matrix -> quaternion; the trace of a near-identity rotation is positive
    TP T;
    double t = R[0] + R[4] + R[8];
    t = sqrt(t + 1.0);
    T.qw = 0.5 * t;
    t = 0.5 / t;
    T.qx = (R[7] - R[5]) * t;
    T.qy = (R[2] - R[6]) * t;
    T.qz = (R[3] - R[1]) * t;
    T.tx = V[0] * u[3] + V[1] * u[4] + V[2] * u[5];
    T.ty = V[3] * u[3] + V[4] * u[4] + V[5] * u[5];
    T.tz = V[6] * u[3] + V[7] * u[4] + V[8] * u[5];
    t_normalize(T);
    return T;
}

__device__ __forceinline__ void t_log(const TP& T, double out[6])   // SE3Quat::log
{
    double R[9];
    t_q_to_R(T, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
    double om[3], c;
    if (fabs(d) > 0.99999) {
#pragma unroll
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        c = 1. / 12.;
    } else {
        const double theta = acos(d);
        const double k = theta / (2 * sqrt(1 - d * d));
#pragma unroll
        for (int i = 0; i < 3; ++i) om[i] = k * dR[i];
        c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    }
    const double Om[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double Om2[9];
    t_m3mul(Om, Om, Om2);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double vi[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) vi[j] = ((i == j) ? 1.0 : 0.0) - 0.5 * Om[i * 3 + j] + c * Om2[i * 3 + j];
        out[i] = om[i];
        out[3 + i] = vi[0] * T.tx + vi[1] * T.ty + vi[2] * T.tz;
    }
}

__device__ __forceinline__ void t_adj(const TP& T, double A[36], double sign)   // SE3Quat::adj: [R 0; skew(t) R  R]
{
    double R[9], tR[9];
    const double tx[9] = { 0, -T.tz, T.ty, T.tz, 0, -T.tx, -T.ty, T.tx, 0 };
    t_q_to_R(T, R);
    t_m3mul(tx, R, tR);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[r * 6 + c] = sign * R[r * 3 + c]; A[r * 6 + 3 + c] = 0.0;
            A[(3 + r) * 6 + c] = sign * tR[r * 3 + c]; A[(3 + r) * 6 + 3 + c] = sign * R[r * 3 + c];
        }
}

struct TMeas { int kind; double q[4]; double t[3]; double dist; double w; };

__device__ __forceinline__ TMeas t_meas(const BaDeviceView& v, int t)
{
    TMeas m;
    m.kind = v.T_kind[t];
    const double* p = v.T_meas + (size_t)t * 8;
    m.q[0] = p[0]; m.q[1] = p[1]; m.q[2] = p[2]; m.q[3] = p[3];
    m.t[0] = p[4]; m.t[1] = p[5]; m.t[2] = p[6];
    m.dist = p[7];
    m.w = v.T_w[t];
    return m;
}

// computeError of the three kinds; err[1..5] untouched for the scalar kinds
__device__ __forceinline__ void t_error(const TMeas& m, const TP& T0, const TP& T1, double err[6])
{
    if (m.kind == TETHER_DISTANCE) {
        const double dx = T1.tx - T0.tx, dy = T1.ty - T0.ty, dz = T1.tz - T0.tz;
        err[0] = (m.dist - sqrt(dx * dx + (dy * dy + dz * dz))) * m.w;
    } else if (m.kind == TETHER_ROTATION) {
        const TP inv0 = t_inverse(T0);
        const TP rel = t_mul(inv0, T1);
        // d = rel.r * conj(meas)   (Eigen 3.3 angularDistance)
        const double bx = -m.q[0], by = -m.q[1], bz = -m.q[2], bw = m.q[3];
        const double dw = rel.qw * bw - rel.qx * bx - rel.qy * by - rel.qz * bz;
        const double dx = rel.qw * bx + rel.qx * bw + rel.qy * bz - rel.qz * by;
        const double dy = rel.qw * by + rel.qy * bw + rel.qz * bx - rel.qx * bz;
        const double dz = rel.qw * bz + rel.qz * bw + rel.qx * by - rel.qy * bx;
        const double vn = sqrt(dx * dx + (dy * dy + dz * dz));
        err[0] = 2.0 * atan2(vn, fabs(dw)) * m.w;
    } else {
        TP C = { m.q[0], m.q[1], m.q[2], m.q[3], m.t[0], m.t[1], m.t[2] };
        const TP inv1 = t_inverse(T1);
        const TP a = t_mul(inv1, C);
        const TP e = t_mul(a, T0);
        t_log(e, err);
    }
}

__device__ __forceinline__ double t_chi2(const TMeas& m, const double err[6])
{
    if (m.kind != TETHER_TRANSFORM) return err[0] * err[0];
    double s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += err[i] * (m.w * err[i]);
    return s;
}

template <int NW>
__device__ __forceinline__ double t_block_sum(double v, double* sm)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += sm[i];
    return r;
}

// chi2 of the active tethers, added to the chi2 slot k_reduce_sum has just written on the same stream
__global__ __launch_bounds__(256) void k_tether_error(BaDeviceView v, int trial)
{
    __shared__ double sm[4];
    const double* pose = trial ? v.pose_trial : v.pose_cur;
    double acc = 0;
    for (int t = threadIdx.x; t < v.n_T; t += 256) {
        const TMeas m = t_meas(v, t);
        const int2 cc = v.T_cam[t];
        const TP T0 = t_load(pose, cc.x), T1 = t_load(pose, cc.y);
        double err[6] = { 0, 0, 0, 0, 0, 0 };
        t_error(m, T0, T1, err);
        acc += t_chi2(m, err);
    }
    const double r = t_block_sum<4>(acc, sm);
    if (threadIdx.x == 0) v.scal[trial ? SC_CHI_TRIAL : SC_CHI] += r;
}

// linearizeOplus + constructQuadraticForm of one tether per thread -> T_out[t] = H00 | H11 | H01 | b0 | b1
__global__ __launch_bounds__(64) void k_tether_linearize(BaDeviceView v)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= v.n_T) return;
    const TMeas m = t_meas(v, t);
    const int2 cc = v.T_cam[t];
    const int2 fx = v.T_fixed[t];
    const TP T0 = t_load(v.pose_cur, cc.x), T1 = t_load(v.pose_cur, cc.y);
    const int dim = (m.kind == TETHER_TRANSFORM) ? 6 : 1;
    const double om = (m.kind == TETHER_TRANSFORM) ? m.w : 1.0;
    double err[6] = { 0, 0, 0, 0, 0, 0 };
    t_error(m, T0, T1, err);
    double J0[36], J1[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) { J0[k] = 0; J1[k] = 0; }
    if (m.kind == TETHER_TRANSFORM) {
        TP Tij = { m.q[0], m.q[1], m.q[2], m.q[3], m.t[0], m.t[1], m.t[2] };
        const TP invTij = t_inverse(Tij);
        const TP a = t_mul(t_inverse(T1), Tij), b = t_mul(t_inverse(T0), invTij);
        t_adj(a, J0, 1.0);
        t_adj(b, J1, -1.0);
    } else {
        const double delta = 1e-9, scalar = 1 / (2 * delta);
        for (int side = 0; side < 2; ++side) {
            if (side ? fx.y : fx.x) continue;
            const TP base = side ? T1 : T0;
            for (int d = 0; d < 6; ++d) {
                double u[6] = { 0, 0, 0, 0, 0, 0 }, ep[6], em[6];
                u[d] = delta;
                const TP Tp = t_mul(t_exp_small(u), base);
                if (side) t_error(m, T0, Tp, ep); else t_error(m, Tp, T1, ep);
                u[d] = -delta;
                const TP Tm = t_mul(t_exp_small(u), base);
                if (side) t_error(m, T0, Tm, em); else t_error(m, Tm, T1, em);
                const double jv = scalar * (ep[0] - em[0]);
                if (side) J1[d] = jv; else J0[d] = jv;
            }
        }
    }
    double* out = v.T_out + (size_t)t * TETHER_OUT_STRIDE;
    for (int r = 0; r < 6; ++r) {
        double a0 = 0, a1 = 0;
        for (int d = 0; d < dim; ++d) { a0 += J0[d * 6 + r] * (-(om * err[d])); a1 += J1[d * 6 + r] * (-(om * err[d])); }
        out[108 + r] = a0; out[114 + r] = a1;
        for (int c = 0; c < 6; ++c) {
            double h00 = 0, h11 = 0, h01 = 0;
            for (int d = 0; d < dim; ++d) {
                h00 += J0[d * 6 + r] * om * J0[d * 6 + c];
                h11 += J1[d * 6 + r] * om * J1[d * 6 + c];
                h01 += J0[d * 6 + r] * om * J1[d * 6 + c];
            }
            out[r * 6 + c] = h00; out[36 + r * 6 + c] = h11; out[72 + r * 6 + c] = h01;
        }
    }
}

// One wavefront per camera that carries tethers: U_c += sum H_ii, b_c += sum b_i in list order.
__global__ __launch_bounds__(256) void k_tether_cam(BaDeviceView v)
{
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= v.n_tc || lane >= 42) return;
    const int hc = v.tc_hc[e];
    double acc = 0;
    for (int k = v.tc_ptr[e]; k < v.tc_ptr[e + 1]; ++k) {
        const int item = v.tc_item[k];
        const double* out = v.T_out + (size_t)(item >> 1) * TETHER_OUT_STRIDE;
        const int side = item & 1;
        acc += (lane < 36) ? out[side * 36 + lane] : out[108 + side * 6 + (lane - 36)];
    }
    if (lane < 36) v.U[(size_t)hc * 36 + lane] += acc;
    else v.bc[(size_t)hc * 6 + (lane - 36)] += acc;
}

// One wavefront per pair of free cameras joined by tethers: block (i, j), i < j, of Hpp added to S (lower triangle).
__global__ __launch_bounds__(256) void k_tether_pair(BaDeviceView v)
{
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= v.n_tp || lane >= 36) return;
    const int2 ij = v.tp_ij[e];
    const int r = lane / 6, c = lane % 6;
    double acc = 0;
    for (int k = v.tp_ptr[e]; k < v.tp_ptr[e + 1]; ++k) {
        const int item = v.tp_item[k];
        const double* H = v.T_out + (size_t)(item >> 1) * TETHER_OUT_STRIDE + 72;
        acc += (item & 1) ? H[c * 6 + r] : H[r * 6 + c];
    }
    // S(row = 6j + c, col = 6i + r) = block(i, j)(r, c)
    v.S[(size_t)(ij.x * 6 + r) * v.n_pad + (ij.y * 6 + c)] += acc;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

void tether_launch_error(const BaDeviceView& v, bool trial, hipStream_t st)
{
    if (v.n_T > 0) hipLaunchKernelGGL(k_tether_error, dim3(1), dim3(256), 0, st, v, trial ? 1 : 0);
}
void tether_launch_linearize(const BaDeviceView& v, hipStream_t st)
{
    if (v.n_T <= 0) return;
    hipLaunchKernelGGL(k_tether_linearize, dim3(cdiv(v.n_T, 64)), dim3(64), 0, st, v);
    if (v.n_tc > 0) hipLaunchKernelGGL(k_tether_cam, dim3(cdiv(v.n_tc, 4)), dim3(256), 0, st, v);
}
void tether_launch_schur(const BaDeviceView& v, hipStream_t st)
{
    if (v.n_tp > 0) hipLaunchKernelGGL(k_tether_pair, dim3(cdiv(v.n_tp, 4)), dim3(256), 0, st, v);
}

}  // namespace mage
