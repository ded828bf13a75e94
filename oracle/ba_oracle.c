/*
 * ba_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of the bundle-adjustment hot path of
 * microsoft/mageslam:
 *
 *   Dependencies/BundlerLib/Source/BundlerLib.cpp   (entire file; cited per function below)
 *   Dependencies/BundlerLib/Include/BundlerLib.h:20-66
 *
 * plus the semantics of the third-party code BundlerLib delegates to, which is NOT
 * vendored under /root/reference (empty submodules, no pinned SHA -- SURVEY.md section 0):
 *
 *   g2o (github.com/RainerKuemmerle/g2o, 2018-2019 master; unpinned):
 *       OptimizationAlgorithmLevenberg::solve, BlockSolver<6,3>::{buildSystem,setLambda,solve},
 *       LinearSolverDense::solve, SparseOptimizer::{initializeOptimization,push,pop,update},
 *       SE3Quat, VertexSE3Expmap, VertexSBAPointXYZ, EdgeProjectXYZ2UV, RobustKernelHuber,
 *       BaseBinaryEdge::constructQuadraticForm          
 *       tether edges: EdgeSE3Expmap::{computeError,linearizeOplus}, SE3Quat::{inverse,log,adj},
 *       BaseMultiEdge::{linearizeOplus (central differences, 1e-9), constructQuadraticForm}
 *   Eigen 3.3.x (unpinned): LDLT<MatrixXd> (pivoted, unblocked, left-looking),
 *       Quaternion<->Matrix3 conversions, Matrix3::inverse (cofactor form).
 *
 * PARITY UNPINNED: the reference tree holds no test, golden vector or known-answer value
 * for this path and the reference cannot be compiled here.
 * What pins this file instead: tests/test_oracle_ba.py checks it against an independent
 * numpy implementation (oracle/indep/ba_numpy.py: dense full-system solve, no Schur,
 * numeric Jacobians) and against committed fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BAO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* small fixed-size helpers (Eigen restatements)                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct { double x, y, z, w; } quat_t;           /* Eigen coeff order x,y,z,w */
typedef struct { quat_t r; double t[3]; } se3_t;         /* g2o::SE3Quat */

static quat_t q_mul(quat_t a, quat_t b)                  /* Eigen quat product */
{
    quat_t r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

static quat_t q_normalized(quat_t q)
{
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    quat_t r = { q.x / n, q.y / n, q.z / n, q.w / n };
    return r;
}

/* Eigen QuaternionBase::_transformVector */
static void q_rot(quat_t q, const double v[3], double out[3])
{
    double uv[3] = { q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

/* Eigen QuaternionBase::toRotationMatrix ; R row-major R[r*3+c] */
static void q_to_R(quat_t q, double R[9])
{
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen quaternionbase_assign_impl<Matrix3>: rotation matrix -> quaternion (double) */
static quat_t R_to_q(const double m[9])
{
    double c[4];
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        c[3] = 0.5 * t;
        t = 0.5 / t;
        c[0] = (m[7] - m[5]) * t;
        c[1] = (m[2] - m[6]) * t;
        c[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        c[i] = 0.5 * t;
        t = 0.5 / t;
        c[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
    quat_t q = { c[0], c[1], c[2], c[3] };
    return q;
}

/* same, single precision: BundlerLib.cpp:272 builds Eigen::Quaternionf{Matrix3f}.normalized() */
static void R_to_q_f32(const float m[9] /*row-major*/, float c[4])
{
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = sqrtf(t + 1.0f);
        c[3] = 0.5f * t;
        t = 0.5f / t;
        c[0] = (m[7] - m[5]) * t;
        c[1] = (m[2] - m[6]) * t;
        c[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0f);
        c[i] = 0.5f * t;
        t = 0.5f / t;
        c[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        c[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        c[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
    float n = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
    c[0] /= n; c[1] /= n; c[2] /= n; c[3] /= n;
}

/* g2o SE3Quat::normalizeRotation (appendix A.1) */
static void se3_normalize(se3_t* T)
{
    if (T->r.w < 0) { T->r.x = -T->r.x; T->r.y = -T->r.y; T->r.z = -T->r.z; T->r.w = -T->r.w; }
    T->r = q_normalized(T->r);
}

static void se3_map(const se3_t* T, const double X[3], double out[3])
{
    q_rot(T->r, X, out);
    out[0] += T->t[0]; out[1] += T->t[1]; out[2] += T->t[2];
}

/* g2o SE3Quat::operator* */
static se3_t se3_mul(const se3_t* a, const se3_t* b)
{
    se3_t r = *a;
    double rt[3];
    q_rot(a->r, b->t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    r.r = q_mul(a->r, b->r);
    se3_normalize(&r);
    return r;
}

static void m3_mul(const double A[9], const double B[9], double C[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}

/* g2o SE3Quat::exp (appendix A.1); update = [omega | upsilon] */
static se3_t se3_exp(const double u[6])
{
    double w0 = u[0], w1 = u[1], w2 = u[2];
    double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    double Om[9] = { 0, -w2, w1, w2, 0, -w0, -w1, w0, 0 };
    double Om2[9];
    m3_mul(Om, Om, Om2);
    double R[9], V[9];
    static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) {
            R[i] = I3[i] + Om[i] + 0.5 * Om2[i];
            V[i] = I3[i] + 0.5 * Om[i] + (1.0 / 6.0) * Om2[i];
        }
    } else {
        double s = sin(theta), c = cos(theta);
        double a = s / theta, b = (1 - c) / (theta * theta), d = (theta - s) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) {
            R[i] = I3[i] + a * Om[i] + b * Om2[i];
            V[i] = I3[i] + b * Om[i] + d * Om2[i];
        }
    }
    se3_t T;
    T.r = R_to_q(R);
    for (int r = 0; r < 3; ++r) T.t[r] = V[r * 3 + 0] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
    se3_normalize(&T);
    return T;
}

/* Eigen compute_inverse_size3 (cofactor form); row-major 3x3. */
static void m3_inverse(const double m[9], double inv[9])
{
#define M(r, c) m[(r) * 3 + (c)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
    double c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    double det = c00 * M(0, 0) + c10 * M(1, 0) + c20 * M(2, 0);
    double invdet = 1.0 / det;
    inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
    inv[3] = COF(0, 1) * invdet; inv[4] = COF(1, 1) * invdet; inv[5] = COF(2, 1) * invdet;
    inv[6] = COF(0, 2) * invdet; inv[7] = COF(1, 2) * invdet; inv[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

/* ------------------------------------------------------------------------------------------ */
/* Eigen::LDLT<MatrixXd, Lower> restated (unblocked, left-looking, diagonal pivoting) A.6     */
/* Column-major n x n, lower triangle referenced.  Returns 1 if isPositive().                 */
/* ------------------------------------------------------------------------------------------ */
enum { SIGN_ZERO = 0, SIGN_POS = 1, SIGN_NEG = 2, SIGN_INDEF = 3 };

static int ldlt_factor(double* A, int n, int* transp, double* temp)
{
    int sign = SIGN_ZERO;
    if (n == 0) return 1;
#define AT(r, c) A[(size_t)(c) * n + (r)]
    for (int k = 0; k < n; ++k) {
        /* largest |diagonal| in the (not yet updated) trailing diagonal */
        int big = k;
        double bigv = fabs(AT(k, k));
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(AT(i, i));
            if (v > bigv) { bigv = v; big = i; }
        }
        transp[k] = big;
        if (k != big) {
            int s = n - big - 1;
            for (int j = 0; j < k; ++j) { double t = AT(k, j); AT(k, j) = AT(big, j); AT(big, j) = t; }
            for (int i = 0; i < s; ++i) { double t = AT(big + 1 + i, k); AT(big + 1 + i, k) = AT(big + 1 + i, big); AT(big + 1 + i, big) = t; }
            { double t = AT(k, k); AT(k, k) = AT(big, big); AT(big, big) = t; }
            for (int i = k + 1; i < big; ++i) { double t = AT(i, k); AT(i, k) = AT(big, i); AT(big, i) = t; }
        }
        int rs = n - k - 1;
        if (k > 0) {
            double acc = 0;
            for (int j = 0; j < k; ++j) { temp[j] = AT(j, j) * AT(k, j); acc += AT(k, j) * temp[j]; }
            AT(k, k) -= acc;
            if (rs > 0) {
                double* a21 = &AT(k + 1, k);
                for (int j = 0; j < k; ++j) {
                    const double* col = &AT(k + 1, j);
                    double tj = temp[j];
                    for (int i = 0; i < rs; ++i) a21[i] -= col[i] * tj;
                }
            }
        }
        double akk = AT(k, k);
        int valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) {
            for (int j = 0; j < n; ++j) transp[j] = j;
            return 1; /* ZeroSign -> isPositive() true */
        }
        if (rs > 0 && valid) {
            double* a21 = &AT(k + 1, k);
            for (int i = 0; i < rs; ++i) a21[i] /= akk;
        }
        if (sign == SIGN_POS) { if (akk < 0) sign = SIGN_INDEF; }
        else if (sign == SIGN_NEG) { if (akk > 0) sign = SIGN_INDEF; }
        else if (sign == SIGN_ZERO) { if (akk > 0) sign = SIGN_POS; else if (akk < 0) sign = SIGN_NEG; }
    }
#undef AT
    return sign == SIGN_POS || sign == SIGN_ZERO;
}

/* Eigen LDLT::_solve_impl */
static void ldlt_solve(const double* A, int n, const int* transp, const double* b, double* x)
{
#define AT(r, c) A[(size_t)(c) * n + (r)]
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int k = 0; k < n; ++k) if (transp[k] != k) { double t = x[k]; x[k] = x[transp[k]]; x[transp[k]] = t; }
    /* L y = x (unit lower) -- column oriented */
    for (int j = 0; j < n; ++j) {
        double xj = x[j];
        if (xj != 0.0) for (int i = j + 1; i < n; ++i) x[i] -= AT(i, j) * xj;
    }
    const double tol = 1.0 / DBL_MAX;
    for (int i = 0; i < n; ++i) { double d = AT(i, i); if (fabs(d) > tol) x[i] /= d; else x[i] = 0; }
    /* L^T z = y */
    for (int j = n - 1; j >= 0; --j) {
        double acc = x[j];
        for (int i = j + 1; i < n; ++i) acc -= AT(i, j) * x[i];
        x[j] = acc;
    }
    for (int k = n - 1; k >= 0; --k) if (transp[k] != k) { double t = x[k]; x[k] = x[transp[k]]; x[transp[k]] = t; }
#undef AT
}

/* ------------------------------------------------------------------------------------------ */
/* problem state                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    se3_t est, backup;
    double f, cx, cy;          /* CameraParameters(focal=intr[2], pp=(intr[0],intr[1]))  BundlerLib.cpp:266 */
    int fixed, set;
    int hidx;                  /* hessian index or -1 */
} cam_t;

typedef struct {
    double est[3], backup[3];
    int set;
    int hidx;
} pt_t;

typedef struct {
    double uv[2], info;
    uint32_t cam, pt;
    double err[2];             /* _error of the last computeError */
    double delta;              /* Huber delta of this edge's kernel */
    int set, removed, active;
    int wblk;                  /* index of its Hpl block (both endpoints free) or -1 */
} obs_t;

/* Tether edges, BundlerLib.cpp:24-90 (EdgeScaleConstraint, EdgeRotationConstraint) and g2o EdgeSE3Expmap. */
enum { TETHER_DISTANCE = 0, TETHER_ROTATION = 1, TETHER_TRANSFORM = 2 };
typedef struct {
    uint32_t c0, c1;           /* setVertex(0, cameraIndex1), setVertex(1, cameraIndex2) */
    double dist;               /* EdgeScaleConstraint measurement */
    quat_t q;                  /* rotation measurement (rotation: as given; transform: normalised by the SE3Quat ctor) */
    double t[3];               /* transform measurement translation */
    double w;                  /* m_weight (distance, rotation) or the scalar of the information matrix w*I6 (transform) */
    int set, active;
    double err[6];             /* _error of the last computeError */
    double J0[36], J1[36];     /* _jacobianOplus[0/1], dim x 6 row-major */
} tether_t;
static const int tether_dim[3] = { 1, 1, 6 };

typedef struct { int code, trials; double chi_before, chi_after, lambda; } bao_trace_t;

typedef struct ba_oracle {
    int points_fixed;
    cam_t* cams; size_t n_cams;
    pt_t* pts; size_t n_pts;
    obs_t* obs; size_t n_obs;
    tether_t* teth[3]; size_t n_teth[3];   /* FixedDistance / RelativeRotation / RelativeTransform constraints */
    /* StepOptimizer state  BundlerLib.cpp:92-167 */
    int dirty, useless, iteration;
    /* LM state (A.4) */
    double lambda, user_lambda_init, ni;
    /* structure (rebuilt when dirty) */
    int* active; size_t n_active;          /* active edge ids, insertion order */
    int n_fc, n_fp;                        /* free cams / free points in the index map */
    int* hc2cam; int* hp2pt;
    double* U;  double* bc;                /* n_fc x 36 (row-major 6x6), n_fc x 6 */
    double* V;  double* bp;                /* n_fp x 9, n_fp x 3 */
    double* W;  int n_w;                   /* n_w x 18 : Hpl(c,l) 6x3 row-major */
    int* lm_off; int* lm_cam; int* lm_blk; /* per landmark (hessian order): sorted (hc, block) lists */
    double* S; double* bs; double* x; double* coeff; double* Dinv;
    int* transp; double* temp;
    /* trace */
    bao_trace_t trace[64]; int n_trace;
} ba_oracle;

BAO_API ba_oracle* bao_create(int points_fixed)
{
    ba_oracle* b = (ba_oracle*)calloc(1, sizeof(ba_oracle));
    b->points_fixed = points_fixed;
    b->dirty = 1;
    b->lambda = -1.0;           /* g2o: _currentLambda(-1) */
    b->user_lambda_init = 0.0;
    b->ni = 2.0;
    return b;
}

static void free_structure(ba_oracle* b)
{
    free(b->active); free(b->hc2cam); free(b->hp2pt); free(b->U); free(b->bc); free(b->V); free(b->bp);
    free(b->W); free(b->lm_off); free(b->lm_cam); free(b->lm_blk); free(b->S); free(b->bs); free(b->x);
    free(b->coeff); free(b->Dinv); free(b->transp); free(b->temp);
    b->active = NULL; b->hc2cam = b->hp2pt = NULL; b->U = b->bc = b->V = b->bp = b->W = NULL;
    b->lm_off = b->lm_cam = b->lm_blk = NULL; b->S = b->bs = b->x = b->coeff = b->Dinv = NULL;
    b->transp = NULL; b->temp = NULL;
}

BAO_API void bao_destroy(ba_oracle* b)
{
    if (!b) return;
    free_structure(b);
    free(b->cams); free(b->pts); free(b->obs);
    for (int k = 0; k < 3; ++k) free(b->teth[k]);
    free(b);
}

/* BundlerLib.cpp:198-229 */
BAO_API void bao_alloc_cameras(ba_oracle* b, size_t n) { b->cams = (cam_t*)calloc(n ? n : 1, sizeof(cam_t)); b->n_cams = n; }
BAO_API void bao_alloc_points(ba_oracle* b, size_t n) { b->pts = (pt_t*)calloc(n ? n : 1, sizeof(pt_t)); b->n_pts = n; }
BAO_API void bao_alloc_observations(ba_oracle* b, size_t n) { b->obs = (obs_t*)calloc(n ? n : 1, sizeof(obs_t)); b->n_obs = n; }

/* BundlerLib.cpp:261-276.  R is column-major 3x3 (Eigen::Map<const Matrix3f>). */
BAO_API void bao_set_camera(ba_oracle* b, size_t idx, const float t[3], const float Rcm[9], const float K[4], int fixed)
{
    cam_t* c = &b->cams[idx];
    float Rrm[9];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rrm[r * 3 + cc] = Rcm[cc * 3 + r];
    float qf[4];
    R_to_q_f32(Rrm, qf);
    c->est.r.x = qf[0]; c->est.r.y = qf[1]; c->est.r.z = qf[2]; c->est.r.w = qf[3];
    c->est.t[0] = t[0]; c->est.t[1] = t[1]; c->est.t[2] = t[2];
    se3_normalize(&c->est);                       /* SE3Quat(q,t) ctor */
    c->f = K[2]; c->cx = K[0]; c->cy = K[1];     /* fy = K[3] ignored, as the reference does */
    c->fixed = fixed; c->set = 1;
    b->dirty = 1;
}

/* Counterpart of the product's extension mage_ba_update_camera_poses (include/mage_ba.h): pose-only re-seed of cameras that are
   already in the graph; the graph is untouched, the optimiser starts over (iteration 0) as after SetCurrentLambda. */
BAO_API void bao_update_camera_poses(ba_oracle* b, size_t n, const uint32_t* idx, const float* t3, const float* R9)
{
    for (size_t k = 0; k < n; ++k) {
        cam_t* c = &b->cams[idx[k]];
        const float* Rcm = R9 + 9 * k;
        float Rrm[9], qf[4];
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rrm[r * 3 + cc] = Rcm[cc * 3 + r];
        R_to_q_f32(Rrm, qf);
        c->est.r.x = qf[0]; c->est.r.y = qf[1]; c->est.r.z = qf[2]; c->est.r.w = qf[3];
        c->est.t[0] = t3[3 * k]; c->est.t[1] = t3[3 * k + 1]; c->est.t[2] = t3[3 * k + 2];
        se3_normalize(&c->est);
    }
    b->iteration = 0;
}

/* Counterpart of the product's device-resident pose exchange (mage_ba_import_poses_device): the float64 state rows
   (qx qy qz qw tx ty tz pad) are taken as they are -- they are another solver's state, already normalised. */
BAO_API void bao_set_camera_poses_f64(ba_oracle* b, size_t n, const uint32_t* idx, const double* qt8)
{
    for (size_t k = 0; k < n; ++k) {
        cam_t* c = &b->cams[idx[k]];
        c->est.r.x = qt8[8 * k]; c->est.r.y = qt8[8 * k + 1]; c->est.r.z = qt8[8 * k + 2]; c->est.r.w = qt8[8 * k + 3];
        c->est.t[0] = qt8[8 * k + 4]; c->est.t[1] = qt8[8 * k + 5]; c->est.t[2] = qt8[8 * k + 6];
    }
    b->iteration = 0;
}

BAO_API void bao_fix_camera(ba_oracle* b, size_t idx, int fixed) { b->cams[idx].fixed = fixed; /* BundlerLib.cpp:278-281: does not dirty */ }

/* BundlerLib.cpp:283-292 */
BAO_API void bao_set_point(ba_oracle* b, size_t idx, const float p[3])
{
    pt_t* v = &b->pts[idx];
    v->est[0] = p[0]; v->est[1] = p[1]; v->est[2] = p[2];
    v->set = 1;
    b->dirty = 1;
}

/* BundlerLib.cpp:294-309 */
BAO_API void bao_set_observation(ba_oracle* b, size_t idx, const float uv[2], size_t cam, size_t pt, float info)
{
    obs_t* e = &b->obs[idx];
    e->uv[0] = uv[0]; e->uv[1] = uv[1];
    e->info = info;
    e->cam = (uint32_t)cam; e->pt = (uint32_t)pt;
    e->err[0] = e->err[1] = 0;
    e->delta = 1.0;              /* RobustKernel default delta */
    e->set = 1; e->removed = 0; e->active = 0; e->wblk = -1;
    b->dirty = 1;
}

/* BundlerLib.cpp:231-259 */
BAO_API void bao_alloc_tethers(ba_oracle* b, int kind, size_t n) { b->teth[kind] = (tether_t*)calloc(n ? n : 1, sizeof(tether_t)); b->n_teth[kind] = n; }
/* BundlerLib.cpp:311-322 */
BAO_API void bao_set_distance_tether(ba_oracle* b, size_t idx, size_t cam1, size_t cam2, float distance, float weight)
{
    tether_t* t = &b->teth[TETHER_DISTANCE][idx];
    memset(t, 0, sizeof(*t));
    t->c0 = (uint32_t)cam1; t->c1 = (uint32_t)cam2; t->dist = (double)distance; t->w = (double)weight; t->set = 1;
    b->dirty = 1;
}
/* BundlerLib.cpp:324-336; q = Eigen::Quaternionf coefficients x,y,z,w, cast to double, NOT normalised */
BAO_API void bao_set_rotation_tether(ba_oracle* b, size_t idx, size_t cam1, size_t cam2, const float q[4], float weight)
{
    tether_t* t = &b->teth[TETHER_ROTATION][idx];
    memset(t, 0, sizeof(*t));
    t->c0 = (uint32_t)cam1; t->c1 = (uint32_t)cam2; t->w = (double)weight; t->set = 1;
    t->q.x = q[0]; t->q.y = q[1]; t->q.z = q[2]; t->q.w = q[3];
    b->dirty = 1;
}
/* BundlerLib.cpp:338-350; measurement = SE3Quat(q, p) (ctor normalises), information = weight * I6 */
BAO_API void bao_set_transform_tether(ba_oracle* b, size_t idx, size_t cam1, size_t cam2, const float p[3], const float q[4], float weight)
{
    tether_t* t = &b->teth[TETHER_TRANSFORM][idx];
    memset(t, 0, sizeof(*t));
    t->c0 = (uint32_t)cam1; t->c1 = (uint32_t)cam2; t->w = (double)weight; t->set = 1;
    se3_t M;
    M.r.x = q[0]; M.r.y = q[1]; M.r.z = q[2]; M.r.w = q[3];
    M.t[0] = p[0]; M.t[1] = p[1]; M.t[2] = p[2];
    se3_normalize(&M);
    t->q = M.r; t->t[0] = M.t[0]; t->t[1] = M.t[1]; t->t[2] = M.t[2];
    b->dirty = 1;
}

/* BundlerLib.cpp:123-130, 354-362 */
BAO_API void bao_set_lambda(ba_oracle* b, float l) { b->iteration = 0; b->user_lambda_init = (double)l; }
BAO_API float bao_get_lambda(const ba_oracle* b) { return (float)b->lambda; }

/* ------------------------------------------------------------------------------------------ */
/* SparseOptimizer::initializeOptimization + BlockSolver::buildStructure  (A.5)               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int hc, edge; } hc_edge_t;
static int cmp_hc_edge(const void* a, const void* b)
{
    const hc_edge_t* x = (const hc_edge_t*)a; const hc_edge_t* y = (const hc_edge_t*)b;
    if (x->hc != y->hc) return x->hc < y->hc ? -1 : 1;
    return x->edge < y->edge ? -1 : (x->edge > y->edge);
}

static void initialize_optimization(ba_oracle* b)
{
    free_structure(b);
    size_t nc = b->n_cams, np = b->n_pts, no = b->n_obs;
    int* cam_deg = (int*)calloc(nc ? nc : 1, sizeof(int));
    int* pt_deg = (int*)calloc(np ? np : 1, sizeof(int));
    b->active = (int*)malloc((no ? no : 1) * sizeof(int));
    b->n_active = 0;
    for (size_t e = 0; e < no; ++e) {
        obs_t* o = &b->obs[e];
        o->active = 0; o->wblk = -1;
        if (!o->set || o->removed) continue;
        int all_fixed = b->cams[o->cam].fixed && b->points_fixed;
        if (all_fixed) continue;
        o->active = 1;
        b->active[b->n_active++] = (int)e;
        cam_deg[o->cam]++; pt_deg[o->pt]++;
    }
    /* tether edges are active unless both endpoints are fixed (OptimizableGraph::Edge::allVerticesFixed) */
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < b->n_teth[k]; ++i) {
            tether_t* t = &b->teth[k][i];
            t->active = 0;
            if (!t->set) continue;
            if (b->cams[t->c0].fixed && b->cams[t->c1].fixed) continue;
            t->active = 1;
            cam_deg[t->c0]++; cam_deg[t->c1]++;
        }
    /* index map: free poses ascending id, then free landmarks ascending id (= descending point idx,
       ids are INT_MAX-2-idx, BundlerLib.cpp:215) */
    b->hc2cam = (int*)malloc((nc ? nc : 1) * sizeof(int));
    b->hp2pt = (int*)malloc((np ? np : 1) * sizeof(int));
    b->n_fc = 0; b->n_fp = 0;
    for (size_t i = 0; i < nc; ++i) {
        cam_t* c = &b->cams[i];
        c->hidx = -1;
        if (cam_deg[i] > 0 && !c->fixed) { c->hidx = b->n_fc; b->hc2cam[b->n_fc++] = (int)i; }
    }
    for (size_t k = 0; k < np; ++k) {
        size_t i = np - 1 - k;
        pt_t* p = &b->pts[i];
        p->hidx = -1;
        if (pt_deg[i] > 0 && !b->points_fixed) { p->hidx = b->n_fp; b->hp2pt[b->n_fp++] = (int)i; }
    }
    free(cam_deg); free(pt_deg);
    b->useless = (b->n_fc + b->n_fp) == 0;
    int nfc = b->n_fc, nfp = b->n_fp;
    b->U = (double*)calloc((size_t)(nfc ? nfc : 1) * 36, sizeof(double));
    b->bc = (double*)calloc((size_t)(nfc ? nfc : 1) * 6, sizeof(double));
    b->V = (double*)calloc((size_t)(nfp ? nfp : 1) * 9, sizeof(double));
    b->bp = (double*)calloc((size_t)(nfp ? nfp : 1) * 3, sizeof(double));
    /* Hpl blocks: one per distinct (free cam, free point) pair; per-landmark lists sorted by hc */
    b->lm_off = (int*)calloc((size_t)nfp + 1, sizeof(int));
    size_t n_pairs = 0;
    for (size_t a = 0; a < b->n_active; ++a) {
        obs_t* o = &b->obs[b->active[a]];
        if (b->cams[o->cam].hidx >= 0 && b->pts[o->pt].hidx >= 0) { b->lm_off[b->pts[o->pt].hidx + 1]++; n_pairs++; }
    }
    for (int l = 0; l < nfp; ++l) b->lm_off[l + 1] += b->lm_off[l];
    hc_edge_t* tmp = (hc_edge_t*)malloc((n_pairs ? n_pairs : 1) * sizeof(hc_edge_t));
    int* fill = (int*)calloc((size_t)nfp + 1, sizeof(int));
    for (size_t a = 0; a < b->n_active; ++a) {
        int e = b->active[a];
        obs_t* o = &b->obs[e];
        int hc = b->cams[o->cam].hidx, hl = b->pts[o->pt].hidx;
        if (hc >= 0 && hl >= 0) { hc_edge_t he = { hc, e }; tmp[b->lm_off[hl] + fill[hl]++] = he; }
    }
    free(fill);
    b->lm_cam = (int*)malloc((n_pairs ? n_pairs : 1) * sizeof(int));
    b->lm_blk = (int*)malloc((n_pairs ? n_pairs : 1) * sizeof(int));
    int* new_off = (int*)calloc((size_t)nfp + 1, sizeof(int));
    int nblk = 0;
    for (int l = 0; l < nfp; ++l) {
        int s = b->lm_off[l], e = b->lm_off[l + 1];
        qsort(tmp + s, (size_t)(e - s), sizeof(hc_edge_t), cmp_hc_edge);
        new_off[l] = nblk;
        int prev = -1;
        for (int k = s; k < e; ++k) {
            if (tmp[k].hc != prev) { b->lm_cam[nblk] = tmp[k].hc; b->lm_blk[nblk] = nblk; nblk++; prev = tmp[k].hc; }
            b->obs[tmp[k].edge].wblk = nblk - 1;   /* duplicate (cam,pt) edges share a block */
        }
    }
    new_off[nfp] = nblk;
    free(b->lm_off); b->lm_off = new_off;
    free(tmp);
    b->n_w = nblk;
    b->W = (double*)calloc((size_t)(nblk ? nblk : 1) * 18, sizeof(double));
    size_t ns = (size_t)nfc * 6;
    b->S = (double*)malloc((ns ? ns * ns : 1) * sizeof(double));
    b->bs = (double*)calloc(ns ? ns : 1, sizeof(double));
    b->coeff = (double*)calloc(ns ? ns : 1, sizeof(double));
    b->x = (double*)calloc(ns + (size_t)nfp * 3 + 1, sizeof(double));
    b->Dinv = (double*)calloc((size_t)(nfp ? nfp : 1) * 9, sizeof(double));
    b->transp = (int*)malloc((ns ? ns : 1) * sizeof(int));
    b->temp = (double*)malloc((ns ? ns : 1) * sizeof(double));
    b->iteration = 0;
    b->dirty = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* EdgeProjectXYZ2UV (A.2) + Huber (A.3)                                                      */
/* ------------------------------------------------------------------------------------------ */
static void compute_error(ba_oracle* b, obs_t* o)
{
    const cam_t* c = &b->cams[o->cam];
    double Xc[3];
    se3_map(&c->est, b->pts[o->pt].est, Xc);
    double px = Xc[0] / Xc[2], py = Xc[1] / Xc[2];
    o->err[0] = o->uv[0] - (px * c->f + c->cx);
    o->err[1] = o->uv[1] - (py * c->f + c->cy);
}

static void huber(double e2, double delta, double rho[3])
{
    double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.0; rho[2] = 0.0; }
    else {
        double sqrte = sqrt(e2);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e2;
    }
}


/* ------------------------------------------------------------------------------------------ */
/* Tether edges: BundlerLib.cpp:24-90 + g2o EdgeSE3Expmap / SE3Quat::{inverse,log,adj} /      */
/* BaseMultiEdge::linearizeOplus (central differences, delta = 1e-9)                          */
/* ------------------------------------------------------------------------------------------ */
static se3_t se3_inverse(const se3_t* T)            /* SE3Quat::inverse: r' = conj(r), t' = r' * (t * -1) */
{
    se3_t r;
    r.r.x = -T->r.x; r.r.y = -T->r.y; r.r.z = -T->r.z; r.r.w = T->r.w;
    double nt[3] = { T->t[0] * -1., T->t[1] * -1., T->t[2] * -1. };
    q_rot(r.r, nt, r.t);
    return r;
}

static void se3_log(const se3_t* T, double out[6])  /* SE3Quat::log: [omega | upsilon] */
{
    double R[9];
    q_to_R(T->r, R);
    double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
    double om[3], Om[9], Om2[9], Vi[9];
    static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (fabs(d) > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
        memcpy(Om, O, sizeof(O));
        m3_mul(Om, Om, Om2);
        for (int i = 0; i < 9; ++i) Vi[i] = I3[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        double theta = acos(d);
        double k = theta / (2 * sqrt(1 - d * d));
        for (int i = 0; i < 3; ++i) om[i] = k * dR[i];
        double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
        memcpy(Om, O, sizeof(O));
        m3_mul(Om, Om, Om2);
        double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = I3[i] - 0.5 * Om[i] + c * Om2[i];
    }
    for (int i = 0; i < 3; ++i) {
        out[i] = om[i];
        out[3 + i] = Vi[i * 3 + 0] * T->t[0] + Vi[i * 3 + 1] * T->t[1] + Vi[i * 3 + 2] * T->t[2];
    }
}

static void se3_adj(const se3_t* T, double A[36])   /* SE3Quat::adj: [R 0; skew(t) R  R], row-major 6x6 */
{
    double R[9], tx[9] = { 0, -T->t[2], T->t[1], T->t[2], 0, -T->t[0], -T->t[1], T->t[0], 0 }, tR[9];
    q_to_R(T->r, R);
    m3_mul(tx, R, tR);
    memset(A, 0, 36 * sizeof(double));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { A[r * 6 + c] = R[r * 3 + c]; A[(3 + r) * 6 + 3 + c] = R[r * 3 + c]; A[(3 + r) * 6 + c] = tR[r * 3 + c]; }
}

/* computeError of the three tether kinds, poses passed explicitly so the numeric differentiation can perturb them */
static void tether_error(int kind, const tether_t* t, const se3_t* T0, const se3_t* T1, double err[6])
{
    if (kind == TETHER_DISTANCE) {                       /* BundlerLib.cpp:45-51 */
        double dx = T1->t[0] - T0->t[0], dy = T1->t[1] - T0->t[1], dz = T1->t[2] - T0->t[2];
        err[0] = (t->dist - sqrt(dx * dx + (dy * dy + dz * dz))) * t->w;
    } else if (kind == TETHER_ROTATION) {                /* BundlerLib.cpp:76-87: angularDistance((T0^-1 T1).rotation(), meas) * w */
        se3_t inv0 = se3_inverse(T0);
        se3_t rel = se3_mul(&inv0, T1);
        quat_t mc = { -t->q.x, -t->q.y, -t->q.z, t->q.w };
        quat_t d = q_mul(rel.r, mc);                     /* Eigen 3.3 angularDistance: d = this * other.conjugate() */
        double vn = sqrt(d.x * d.x + (d.y * d.y + d.z * d.z));
        err[0] = 2.0 * atan2(vn, fabs(d.w)) * t->w;
    } else {                                             /* g2o EdgeSE3Expmap::computeError: log(T1^-1 * C * T0) */
        se3_t C; C.r = t->q; C.t[0] = t->t[0]; C.t[1] = t->t[1]; C.t[2] = t->t[2];
        se3_t inv1 = se3_inverse(T1);
        se3_t a = se3_mul(&inv1, &C);
        se3_t e = se3_mul(&a, T0);
        se3_log(&e, err);
    }
}

static void tether_linearize(ba_oracle* b, int kind, tether_t* t)
{
    const cam_t* c0 = &b->cams[t->c0]; const cam_t* c1 = &b->cams[t->c1];
    const int dim = tether_dim[kind];
    memset(t->J0, 0, sizeof(t->J0)); memset(t->J1, 0, sizeof(t->J1));
    if (kind == TETHER_TRANSFORM) {                      /* g2o EdgeSE3Expmap::linearizeOplus (analytic) */
        se3_t Tij; Tij.r = t->q; Tij.t[0] = t->t[0]; Tij.t[1] = t->t[1]; Tij.t[2] = t->t[2];
        se3_t invTij = se3_inverse(&Tij);
        se3_t invTj = se3_inverse(&c1->est), invTi = se3_inverse(&c0->est);
        se3_t a = se3_mul(&invTj, &Tij), bq = se3_mul(&invTi, &invTij);
        se3_adj(&a, t->J0);
        se3_adj(&bq, t->J1);
        for (int i = 0; i < 36; ++i) t->J1[i] = -t->J1[i];
        return;
    }
    /* BaseMultiEdge::linearizeOplus: central differences through oplus (exp(u) * estimate), fixed vertices skipped */
    const double delta = 1e-9, scalar = 1 / (2 * delta);
    for (int side = 0; side < 2; ++side) {
        const cam_t* cv = side ? c1 : c0;
        if (cv->fixed) continue;
        double* J = side ? t->J1 : t->J0;
        for (int d = 0; d < 6; ++d) {
            double u[6] = { 0, 0, 0, 0, 0, 0 }, ep[6], em[6];
            u[d] = delta;
            se3_t E = se3_exp(u);
            se3_t Tp = se3_mul(&E, &cv->est);
            tether_error(kind, t, side ? &c0->est : &Tp, side ? &Tp : &c1->est, ep);
            u[d] = -delta;
            E = se3_exp(u);
            se3_t Tm = se3_mul(&E, &cv->est);
            tether_error(kind, t, side ? &c0->est : &Tm, side ? &Tm : &c1->est, em);
            for (int k = 0; k < dim; ++k) J[k * 6 + d] = scalar * (ep[k] - em[k]);
        }
    }
}

/* chi2 of a tether: e^T Omega e, Omega = I (distance, rotation) or w*I6 (transform); no robust kernel */
static double tether_chi2(int kind, const tether_t* t)
{
    if (kind != TETHER_TRANSFORM) return t->err[0] * t->err[0];
    double s = 0;
    for (int i = 0; i < 6; ++i) s += t->err[i] * (t->w * t->err[i]);
    return s;
}

static void compute_active_errors(ba_oracle* b)
{
    for (size_t a = 0; a < b->n_active; ++a) compute_error(b, &b->obs[b->active[a]]);
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < b->n_teth[k]; ++i) {
            tether_t* t = &b->teth[k][i];
            if (t->active) tether_error(k, t, &b->cams[t->c0].est, &b->cams[t->c1].est, t->err);
        }
}

static double active_robust_chi2(ba_oracle* b)
{
    double chi = 0;
    for (size_t a = 0; a < b->n_active; ++a) {
        obs_t* o = &b->obs[b->active[a]];
        double chi2 = o->info * (o->err[0] * o->err[0] + o->err[1] * o->err[1]);
        double rho[3];
        huber(chi2, o->delta, rho);
        chi += rho[0];
    }
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < b->n_teth[k]; ++i)
            if (b->teth[k][i].active) chi += tether_chi2(k, &b->teth[k][i]);
    return chi;
}

/* BlockSolver::buildSystem: zero, linearizeOplus + constructQuadraticForm per active edge (A.2/A.3) */
static void build_system(ba_oracle* b)
{
    memset(b->U, 0, (size_t)b->n_fc * 36 * sizeof(double));
    memset(b->bc, 0, (size_t)b->n_fc * 6 * sizeof(double));
    memset(b->V, 0, (size_t)b->n_fp * 9 * sizeof(double));
    memset(b->bp, 0, (size_t)b->n_fp * 3 * sizeof(double));
    memset(b->W, 0, (size_t)b->n_w * 18 * sizeof(double));
    for (size_t a = 0; a < b->n_active; ++a) {
        obs_t* o = &b->obs[b->active[a]];
        const cam_t* c = &b->cams[o->cam];
        const pt_t* p = &b->pts[o->pt];
        double Xc[3];
        se3_map(&c->est, p->est, Xc);
        double x = Xc[0], y = Xc[1], z = Xc[2], z2 = z * z, f = c->f;
        double R[9];
        q_to_R(c->est.r, R);
        double tmp[6] = { f, 0, -x / z * f, 0, f, -y / z * f };
        double Jp[6];   /* 2x3 */
        for (int r = 0; r < 2; ++r)
            for (int cc = 0; cc < 3; ++cc)
                Jp[r * 3 + cc] = -1.0 / z * (tmp[r * 3 + 0] * R[0 * 3 + cc] + tmp[r * 3 + 1] * R[1 * 3 + cc] + tmp[r * 3 + 2] * R[2 * 3 + cc]);
        double Jc[12];  /* 2x6 */
        Jc[0] = x * y / z2 * f;          Jc[1] = -(1 + (x * x / z2)) * f; Jc[2] = y / z * f;
        Jc[3] = -1.0 / z * f;            Jc[4] = 0;                       Jc[5] = x / z2 * f;
        Jc[6] = (1 + y * y / z2) * f;    Jc[7] = -x * y / z2 * f;         Jc[8] = -x / z * f;
        Jc[9] = 0;                       Jc[10] = -1.0 / z * f;           Jc[11] = y / z2 * f;

        int hc = c->hidx, hl = p->hidx;
        if (hc < 0 && hl < 0) continue;
        double chi2 = o->info * (o->err[0] * o->err[0] + o->err[1] * o->err[1]);
        double rho[3];
        huber(chi2, o->delta, rho);
        double w = o->info * rho[1];                                  /* weightedOmega = rho1 * Omega */
        double r0 = -o->info * o->err[0] * rho[1], r1 = -o->info * o->err[1] * rho[1]; /* omega_r */
        if (hl >= 0) {      /* "from" = vertex 0 = point */
            double* bp = &b->bp[hl * 3];
            double* V = &b->V[hl * 9];
            for (int i = 0; i < 3; ++i) {
                bp[i] += Jp[0 * 3 + i] * r0 + Jp[1 * 3 + i] * r1;
                for (int j = 0; j < 3; ++j) V[i * 3 + j] += Jp[0 * 3 + i] * w * Jp[0 * 3 + j] + Jp[1 * 3 + i] * w * Jp[1 * 3 + j];
            }
            if (hc >= 0) {
                double* W = &b->W[(size_t)o->wblk * 18];   /* Hpl(c,l) = Jc^T w Jp, 6x3 */
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 3; ++j)
                        W[i * 3 + j] += Jc[0 * 6 + i] * w * Jp[0 * 3 + j] + Jc[1 * 6 + i] * w * Jp[1 * 3 + j];
            }
        }
        if (hc >= 0) {
            double* bc = &b->bc[hc * 6];
            double* U = &b->U[hc * 36];
            for (int i = 0; i < 6; ++i) {
                bc[i] += Jc[0 * 6 + i] * r0 + Jc[1 * 6 + i] * r1;
                for (int j = 0; j < 6; ++j) U[i * 6 + j] += Jc[0 * 6 + i] * w * Jc[0 * 6 + j] + Jc[1 * 6 + i] * w * Jc[1 * 6 + j];
            }
        }
    }
    /* tether edges: BaseMultiEdge / BaseBinaryEdge::constructQuadraticForm without a robust kernel:
       H_ii += Ji^T Omega Ji, b_i += Ji^T (-Omega e), H_01 += J0^T Omega J1 (kept per tether, added to Hpp in solver_solve) */
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < b->n_teth[k]; ++i) {
            tether_t* t = &b->teth[k][i];
            if (!t->active) continue;
            tether_linearize(b, k, t);
            const int dim = tether_dim[k];
            const double om = (k == TETHER_TRANSFORM) ? t->w : 1.0;
            for (int side = 0; side < 2; ++side) {
                int hc = b->cams[side ? t->c1 : t->c0].hidx;
                if (hc < 0) continue;
                const double* J = side ? t->J1 : t->J0;
                double* bc = &b->bc[hc * 6];
                double* U = &b->U[hc * 36];
                for (int r = 0; r < 6; ++r) {
                    double acc = 0;
                    for (int d = 0; d < dim; ++d) acc += J[d * 6 + r] * (-(om * t->err[d]));
                    bc[r] += acc;
                    for (int c = 0; c < 6; ++c) {
                        double h = 0;
                        for (int d = 0; d < dim; ++d) h += J[d * 6 + r] * om * J[d * 6 + c];
                        U[r * 6 + c] += h;
                    }
                }
            }
        }
}

/* Hpp off-diagonal blocks of the tether edges, added to both triangles of the dense column-major matrix */
static void add_tether_offdiag(ba_oracle* b, int n)
{
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < b->n_teth[k]; ++i) {
            const tether_t* t = &b->teth[k][i];
            if (!t->active) continue;
            int h0 = b->cams[t->c0].hidx, h1 = b->cams[t->c1].hidx;
            if (h0 < 0 || h1 < 0) continue;
            const int dim = tether_dim[k];
            const double om = (k == TETHER_TRANSFORM) ? t->w : 1.0;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    double h = 0;
                    for (int d = 0; d < dim; ++d) h += t->J0[d * 6 + r] * om * t->J1[d * 6 + c];
                    b->S[(size_t)(h1 * 6 + c) * n + (h0 * 6 + r)] += h;
                    b->S[(size_t)(h0 * 6 + r) * n + (h1 * 6 + c)] += h;
                }
        }
}

/* BlockSolver::solve with lambda applied (setLambda + solve + restoreDiagonal folded)  A.5/A.6 */
static int solver_solve(ba_oracle* b, double lambda)
{
    int nfc = b->n_fc, nfp = b->n_fp, n = nfc * 6;
    double* x = b->x;
    if (nfp == 0) {
        /* no Schur: Hpp x = b */
        if (n == 0) return 1;
        memset(b->S, 0, (size_t)n * n * sizeof(double));
        for (int c = 0; c < nfc; ++c)
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    b->S[(size_t)(c * 6 + j) * n + (c * 6 + i)] = b->U[c * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
        add_tether_offdiag(b, n);
        if (!ldlt_factor(b->S, n, b->transp, b->temp)) return 0;
        ldlt_solve(b->S, n, b->transp, b->bc, x);
        return 1;
    }
    /* S = Hpp(+lambda) ; column-major dense, upper blocks formed then mirrored */
    if (n > 0) memset(b->S, 0, (size_t)n * n * sizeof(double));
    for (int c = 0; c < nfc; ++c)
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
                b->S[(size_t)(c * 6 + j) * n + (c * 6 + i)] = b->U[c * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
    add_tether_offdiag(b, n);           /* _Hschur starts as Hpp, tether pose-pose blocks included */
    memset(b->coeff, 0, (size_t)n * sizeof(double));
    for (int l = 0; l < nfp; ++l) {
        double D[9], db[3];
        memcpy(D, &b->V[l * 9], sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Dinv = &b->Dinv[l * 9];
        m3_inverse(D, Dinv);
        const double* bl = &b->bp[l * 3];
        for (int i = 0; i < 3; ++i) db[i] = Dinv[i * 3 + 0] * bl[0] + Dinv[i * 3 + 1] * bl[1] + Dinv[i * 3 + 2] * bl[2];
        for (int k1 = b->lm_off[l]; k1 < b->lm_off[l + 1]; ++k1) {
            int i1 = b->lm_cam[k1];
            const double* Bi = &b->W[(size_t)b->lm_blk[k1] * 18];
            double BDinv[18];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 3; ++c)
                    BDinv[r * 3 + c] = Bi[r * 3 + 0] * Dinv[0 * 3 + c] + Bi[r * 3 + 1] * Dinv[1 * 3 + c] + Bi[r * 3 + 2] * Dinv[2 * 3 + c];
            for (int r = 0; r < 6; ++r) b->coeff[i1 * 6 + r] += Bi[r * 3 + 0] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
            for (int k2 = k1; k2 < b->lm_off[l + 1]; ++k2) {
                int i2 = b->lm_cam[k2];
                const double* Bj = &b->W[(size_t)b->lm_blk[k2] * 18];
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c)
                        b->S[(size_t)(i2 * 6 + c) * n + (i1 * 6 + r)] -= BDinv[r * 3 + 0] * Bj[c * 3 + 0] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
    for (int i = 0; i < n; ++i) b->bs[i] = b->bc[i] - b->coeff[i];
    /* LinearSolverDense: mirror upper blocks into the lower triangle, LDLT */
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < c; ++r) {
            int br = r / 6, bcn = c / 6;
            if (br != bcn) b->S[(size_t)r * n + c] = b->S[(size_t)c * n + r];
        }
    if (n > 0) {
        if (!ldlt_factor(b->S, n, b->transp, b->temp)) return 0;
        ldlt_solve(b->S, n, b->transp, b->bs, x);
    }
    /* landmarks: xl = Dinv (bl - Hpl^T xp) */
    for (int l = 0; l < nfp; ++l) {
        double cl[3] = { b->bp[l * 3], b->bp[l * 3 + 1], b->bp[l * 3 + 2] };
        for (int k = b->lm_off[l]; k < b->lm_off[l + 1]; ++k) {
            const double* Bi = &b->W[(size_t)b->lm_blk[k] * 18];
            const double* xp = &x[b->lm_cam[k] * 6];
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 6; ++r) cl[c] += Bi[r * 3 + c] * (-xp[r]);
        }
        const double* Dinv = &b->Dinv[l * 9];
        for (int i = 0; i < 3; ++i) x[n + l * 3 + i] = Dinv[i * 3 + 0] * cl[0] + Dinv[i * 3 + 1] * cl[1] + Dinv[i * 3 + 2] * cl[2];
    }
    return 1;
}

static void push_state(ba_oracle* b)
{
    for (int i = 0; i < b->n_fc; ++i) { cam_t* c = &b->cams[b->hc2cam[i]]; c->backup = c->est; }
    for (int i = 0; i < b->n_fp; ++i) { pt_t* p = &b->pts[b->hp2pt[i]]; memcpy(p->backup, p->est, sizeof(p->est)); }
}
static void pop_state(ba_oracle* b)
{
    for (int i = 0; i < b->n_fc; ++i) { cam_t* c = &b->cams[b->hc2cam[i]]; c->est = c->backup; }
    for (int i = 0; i < b->n_fp; ++i) { pt_t* p = &b->pts[b->hp2pt[i]]; memcpy(p->est, p->backup, sizeof(p->est)); }
}
/* SparseOptimizer::update -> oplus  (A.1) */
static void apply_update(ba_oracle* b)
{
    int n = b->n_fc * 6;
    for (int i = 0; i < b->n_fc; ++i) {
        cam_t* c = &b->cams[b->hc2cam[i]];
        se3_t E = se3_exp(&b->x[i * 6]);
        c->est = se3_mul(&E, &c->est);
    }
    for (int i = 0; i < b->n_fp; ++i) {
        pt_t* p = &b->pts[b->hp2pt[i]];
        p->est[0] += b->x[n + i * 3]; p->est[1] += b->x[n + i * 3 + 1]; p->est[2] += b->x[n + i * 3 + 2];
    }
}

enum { LM_OK = 0, LM_TERMINATE = 1, LM_FAIL = 2 };

/* OptimizationAlgorithmLevenberg::solve  (A.4) */
static int lm_solve(ba_oracle* b, int iteration)
{
    bao_trace_t tr; memset(&tr, 0, sizeof(tr));
    compute_active_errors(b);
    double currentChi = active_robust_chi2(b);
    double tempChi = currentChi;
    tr.chi_before = currentChi;
    build_system(b);
    int n = b->n_fc * 6;
    if (iteration == 0) {
        if (b->user_lambda_init > 0) b->lambda = b->user_lambda_init;
        else {
            double maxDiag = 0;
            for (int i = 0; i < b->n_fc; ++i) for (int j = 0; j < 6; ++j) { double v = fabs(b->U[i * 36 + j * 7]); if (v > maxDiag) maxDiag = v; }
            for (int i = 0; i < b->n_fp; ++i) for (int j = 0; j < 3; ++j) { double v = fabs(b->V[i * 9 + j * 4]); if (v > maxDiag) maxDiag = v; }
            b->lambda = 1e-5 * maxDiag;
        }
        b->ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
        push_state(b);
        int ok2 = solver_solve(b, b->lambda);
        apply_update(b);
        compute_active_errors(b);
        tempChi = active_robust_chi2(b);
        if (!ok2) tempChi = DBL_MAX;
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < n; ++j) scale += b->x[j] * (b->lambda * b->x[j] + b->bc[j]);
        for (int j = 0; j < b->n_fp * 3; ++j) scale += b->x[n + j] * (b->lambda * b->x[n + j] + b->bp[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = alpha < (2. / 3.) ? alpha : (2. / 3.);
            double scaleFactor = (1. / 3.) > alpha ? (1. / 3.) : alpha;
            b->lambda *= scaleFactor;
            b->ni = 2;
            currentChi = tempChi;
        } else {
            b->lambda *= b->ni;
            b->ni *= 2;
            pop_state(b);
        }
        qmax++;
    } while (rho < 0 && qmax < 10);
    tr.chi_after = currentChi; tr.lambda = b->lambda; tr.trials = qmax;
    int code = (qmax == 10 || rho == 0) ? LM_TERMINATE : LM_OK;
    tr.code = code;
    if (b->n_trace < 64) b->trace[b->n_trace++] = tr;
    return code;
}

/* StepOptimizer::Step  BundlerLib.cpp:132-149 */
static int step_optimizer(ba_oracle* b)
{
    if (b->dirty) initialize_optimization(b);
    if (b->useless) return 0;
    int r = lm_solve(b, b->iteration);
    b->iteration++;
    return r == LM_OK;
}

/* BundlerLib::StepBundleAdjustment  BundlerLib.cpp:364-447 */
BAO_API float bao_step(ba_oracle* b, const float* huber_w, size_t n_huber, float max_err_sq,
                       unsigned* outliers, size_t cap, size_t* n_out)
{
    b->n_trace = 0;
    float prior = -1.f;
    for (size_t h = 0; h < n_huber; ++h) {
        float hw = huber_w[h];
        if (hw != prior) {
            for (size_t e = 0; e < b->n_obs; ++e) b->obs[e].delta = (double)hw;
            prior = hw;
        }
        if (!step_optimizer(b)) break;
    }
    size_t nout = 0;
    int count = 0;
    double error = 0;
    for (size_t a = 0; a < b->n_active; ++a) {
        int e = b->active[a];
        obs_t* o = &b->obs[e];
        double ss = o->err[0] * o->err[0] + o->err[1] * o->err[1];
        const cam_t* c = &b->cams[o->cam];
        /* worldPose = estimate().inverse(): r' = conj(r), t' = r' * (-t) */
        quat_t rc = { -c->est.r.x, -c->est.r.y, -c->est.r.z, c->est.r.w };
        double nt[3] = { -c->est.t[0], -c->est.t[1], -c->est.t[2] }, wt[3];
        q_rot(rc, nt, wt);
        const double* X = b->pts[o->pt].est;
        double pv[3] = { X[0] - wt[0], X[1] - wt[1], X[2] - wt[2] };
        double ez[3] = { 0, 0, 1 }, fwd[3];
        q_rot(rc, ez, fwd);
        double dot = pv[0] * fwd[0] + pv[1] * fwd[1] + pv[2] * fwd[2];
        if (dot <= 0 || ss > (double)max_err_sq) {
            o->removed = 1;             /* removeEdge */
            b->dirty = 1;
            if (outliers && nout < cap) outliers[nout] = (unsigned)e;
            nout++;
        } else {
            error += ss;
            count++;
        }
    }
    if (n_out) *n_out = nout;
    return (float)(error / count);      /* count==0 -> NaN, as the reference */
}

/* BundlerLib.cpp:457-471 */
BAO_API void bao_get_pose(const ba_oracle* b, size_t idx, float t[3], float Rcm[9])
{
    const cam_t* c = &b->cams[idx];
    double R[9];
    q_to_R(q_normalized(c->est.r), R);
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rcm[cc * 3 + r] = (float)R[r * 3 + cc];
    t[0] = (float)c->est.t[0]; t[1] = (float)c->est.t[1]; t[2] = (float)c->est.t[2];
}
BAO_API void bao_get_point(const ba_oracle* b, size_t idx, float p[3])
{
    p[0] = (float)b->pts[idx].est[0]; p[1] = (float)b->pts[idx].est[1]; p[2] = (float)b->pts[idx].est[2];
}

/* ---- test-only accessors (double precision state, LM trace) ---- */
BAO_API void bao_get_pose_f64(const ba_oracle* b, size_t idx, double qt[7])
{
    const cam_t* c = &b->cams[idx];
    qt[0] = c->est.r.x; qt[1] = c->est.r.y; qt[2] = c->est.r.z; qt[3] = c->est.r.w;
    qt[4] = c->est.t[0]; qt[5] = c->est.t[1]; qt[6] = c->est.t[2];
}
BAO_API void bao_get_point_f64(const ba_oracle* b, size_t idx, double p[3]) { memcpy(p, b->pts[idx].est, 3 * sizeof(double)); }
BAO_API int bao_trace_count(const ba_oracle* b) { return b->n_trace; }
BAO_API void bao_trace_get(const ba_oracle* b, int i, int* code, int* trials, double* chi_before, double* chi_after, double* lambda)
{
    *code = b->trace[i].code; *trials = b->trace[i].trials; *chi_before = b->trace[i].chi_before;
    *chi_after = b->trace[i].chi_after; *lambda = b->trace[i].lambda;
}
BAO_API double bao_lambda_f64(const ba_oracle* b) { return b->lambda; }
BAO_API void bao_get_errors(const ba_oracle* b, double* err2 /* n_obs x 2 */)
{
    for (size_t e = 0; e < b->n_obs; ++e) { err2[e * 2] = b->obs[e].err[0]; err2[e * 2 + 1] = b->obs[e].err[1]; }
}
/* expose pieces for unit tests */
BAO_API void bao_test_se3_exp(const double u[6], double qt[7])
{
    se3_t T = se3_exp(u);
    qt[0] = T.r.x; qt[1] = T.r.y; qt[2] = T.r.z; qt[3] = T.r.w; qt[4] = T.t[0]; qt[5] = T.t[1]; qt[6] = T.t[2];
}
BAO_API int bao_test_ldlt(double* A_colmajor, int n, const double* rhs, double* x)
{
    int* tr = (int*)malloc(sizeof(int) * (size_t)(n ? n : 1));
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(n ? n : 1));
    int ok = ldlt_factor(A_colmajor, n, tr, tmp);
    if (ok) ldlt_solve(A_colmajor, n, tr, rhs, x);
    free(tr); free(tmp);
    return ok;
}

/* bulk forms of the setters (same semantics as the scalar ones, idx = 0..n-1); test/bench convenience */
BAO_API void bao_set_cameras_bulk(ba_oracle* b, size_t n, const float* t3, const float* R9, const float* K4, const uint8_t* fixed)
{
    for (size_t i = 0; i < n; ++i) bao_set_camera(b, i, t3 + i * 3, R9 + i * 9, K4 + i * 4, fixed[i]);
}
BAO_API void bao_set_points_bulk(ba_oracle* b, size_t n, const float* p3)
{
    for (size_t i = 0; i < n; ++i) bao_set_point(b, i, p3 + i * 3);
}
BAO_API void bao_set_observations_bulk(ba_oracle* b, size_t n, const float* uv2, const uint32_t* cam, const uint32_t* pt, const float* info)
{
    for (size_t i = 0; i < n; ++i) bao_set_observation(b, i, uv2 + i * 2, cam[i], pt[i], info[i]);
}
