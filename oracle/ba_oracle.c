/*
 * oracle/ba_oracle.c -- CPU restatement of stella_vslam's local bundle adjustment (g2o backend).
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Checker for the HIP BA kernels and the timed "port" CPU
 * baseline of bench.py; never linked into or called from the product.
 *
 * Restated (reference = /root/reference/src/stella_vslam):
 *   optimize/local_bundle_adjuster_g2o.cc:149-164   solver stack: LM( BlockSolver_6_3( LinearSolverEigen ) ),
 *                                                   terminate_action gain 1e-3, force-stop flag
 *   optimize/local_bundle_adjuster_g2o.cc:195-245   per-observation edge: obs (u,v[,u_right]) f32, Omega = I*inv_sigma_sq,
 *                                                   Huber delta = sqrt(chi_sq) chosen by the keyframe's setup type
 *   optimize/local_bundle_adjuster_g2o.cc:306-348   two-stage schedule: optimize(5); chi2/depth gate -> level 1, all
 *                                                   kernels removed; optimize(10)
 *   optimize/local_bundle_adjuster_g2o.cc:352-375   final outlier list (chi2 of the last computeActiveErrors, stale
 *                                                   for level-1 edges)
 *   optimize/internal/se3/perspective_reproj_edge.h:67-120   mono: error, Jacobians, depth_is_positive, cam_project
 *   optimize/internal/se3/perspective_reproj_edge.h:173-238  stereo
 *   optimize/internal/se3/shot_vertex.h:55-58       oplus: exp(update) * estimate
 *   optimize/internal/landmark_vertex.h:50-53       oplus: additive
 *   optimize/terminate_action.cc:36-76              stop rule; writes through the optimizer's force-stop pointer
 *
 * Third-party arithmetic NOT in /root/reference: g2o, pinned `20230223_git` (Dockerfile.desktop:82).
 * Restated from its published sources: OptimizationAlgorithmLevenberg::solve / computeLambdaInit /
 * computeScale, SparseOptimizer::optimize, BlockSolver<6,3>::buildSystem / solve (Schur complement),
 * BaseBinaryEdge::constructQuadraticForm, RobustKernelHuber::robustify, SE3Quat::exp / operator* / map.
 * LinearSolverEigen's sparse Cholesky is replaced by a dense LL^T of the same reduced matrix
 * (mathematically identical solution; different rounding order).
 *
 * PARITY STATUS: the reference's OWN part of this path -- the vertex and edge classes and the wrappers that configure them
 * (optimize/internal/landmark_vertex.h, se3/shot_vertex.h, se3/*_reproj_edge.h, se3/*_pose_opt_edge.h, se3/*_wrapper.h) -- is pinned
 * bit for bit against the reference's compiled headers (oracle/ref_local -> oracle/_ref/libsvref_opt.so, tests/test_ref_local_optimize.py:
 * errors, both Jacobian blocks, depth gate, chi2, information, Huber width, levels, oplus; optimize/terminate_action.cc over a scripted
 * optimizer; optimize/pose_optimizer_g2o.cc, local_bundle_adjuster_g2o.cc and global_bundle_adjuster.cc with this file's LM behind g2o's optimize(): gather,
 * graph, schedule, gating, outlier list, write-back, return value -- tests/test_ref_local_ba.py).  g2o's side (LM schedule, block solver,
 * SE3Quat arithmetic, robust weighting) stays "parity unpinned": the reference has no test under test/stella_vslam/optimize/ and g2o
 * cannot be built here.  That part is cross-checked against scipy.optimize.least_squares and known
 * ground truth on synthetic scenes (tests/test_oracle_ba.py).  Vertex ordering in the reference is
 * unordered_map hash order, so only mathematical (<=1e-4 relative), not bitwise, parity is meaningful.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- SE3Quat (g2o/types/slam3d/se3quat.h) */
typedef struct {
    double q[4]; /* x y z w */
    double t[3];
} se3q;

static void quat_normalize(se3q* T) { /* SE3Quat::normalizeRotation */
    if (T->q[3] < 0) {
        for (int i = 0; i < 4; ++i) T->q[i] = -T->q[i];
    }
    const double n = sqrt(T->q[0] * T->q[0] + T->q[1] * T->q[1] + T->q[2] * T->q[2] + T->q[3] * T->q[3]);
    for (int i = 0; i < 4; ++i) T->q[i] /= n;
}

static void quat_from_R(const double* R /*row-major 3x3*/, double* q) { /* Eigen Quaternion(Matrix3) */
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    }
    else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

static void quat_to_R(const double* q, double* R) { /* Eigen toRotationMatrix */
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}

static void quat_rotate(const double* q, const double* v, double* out) { /* Eigen q * v */
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]);
    const double uy = 2 * (q[2] * v[0] - q[0] * v[2]);
    const double uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

static void quat_mul(const double* a, const double* b, double* o) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

static void se3_map(const se3q* T, const double* p, double* out) {
    quat_rotate(T->q, p, out);
    out[0] += T->t[0];
    out[1] += T->t[1];
    out[2] += T->t[2];
}

static void se3_exp(const double* upd /* omega(3), upsilon(3) */, se3q* out) {
    const double* w = upd;
    const double* u = upd + 3;
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double a, b, c, d;
    if (theta < 0.00001) {
        a = 1.0;
        b = 0.5;
        c = 0.5;
        d = 1.0 / 6.0;
    }
    else {
        a = sin(theta) / theta;
        b = (1 - cos(theta)) / (theta * theta);
        c = b;
        d = (theta - sin(theta)) / pow(theta, 3);
    }
    double R[9], V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * O[i] + b * O2[i];
        V[i] = I + c * O[i] + d * O2[i];
    }
    quat_from_R(R, out->q);
    for (int i = 0; i < 3; ++i) out->t[i] = V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2];
    quat_normalize(out);
}

static void se3_mul(const se3q* A, const se3q* B, se3q* out) { /* A * B */
    se3q r;
    quat_mul(A->q, B->q, r.q);
    quat_rotate(A->q, B->t, r.t);
    for (int i = 0; i < 3; ++i) r.t[i] += A->t[i];
    quat_normalize(&r);
    *out = r;
}

/* ---------------------------------------------------------------- problem state */
typedef struct {
    int P, L, E;
    const uint8_t* pose_fixed;
    const uint8_t* point_fixed; /* nullable */
    const int32_t *obs_pose, *obs_point;
    const float *obs_uvr, *obs_inv_sigma_sq, *obs_huber;
    const double* intr; /* P x 5 */
    se3q* pose;         /* current estimates */
    double* pt;
    uint8_t* level;   /* per edge: 0 active, 1 excluded */
    uint8_t* robust;  /* per edge: kernel present */
    double* err;      /* per edge cached error (3) -- as g2o caches _error */
    /* index maps over ACTIVE non-fixed vertices */
    int* pose_slot;  /* P: slot or -1 */
    int* point_slot; /* L: slot or -1 */
    int nP, nL;
} ba_t;

/* Camera model of a pose.  The flat problem carries fx fy cx cy fxb per pose; fx == fy == 0 selects the EQUIRECTANGULAR model
 * with cols = K[2], rows = K[3] (optimize/internal/se3/equirectangular_reproj_edge.h:64-134, always monocular,
 * reproj_edge_wrapper.h:143-162; depth_is_positive() is constant true for it, :247-249). */
static inline int cam_is_equirect(const double* K) { return K[0] == 0.0 && K[1] == 0.0; }
static inline void equirect_project(const double* K, const double* pc, double* u, double* v) { /* cam_project, :128-132 */
    const double theta = atan2(pc[0], pc[2]);
    const double phi = -asin(pc[1] / sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]));
    *u = K[2] * (0.5 + theta / (2 * M_PI));
    *v = K[3] * (0.5 - phi / M_PI);
}
/* linearizeOplus (:70-126): rows of d error / d [rx ry rz tx ty tz] (Bj, 2 x 6) and d error / d landmark (A, 2 x 3; Rcw = rows of R) */
static inline void equirect_jacobians(const double* K, const double* pc, const double* Rcw, double* A, double* Bj) {
    const double x = pc[0], y = pc[1], z = pc[2], L = sqrt(x * x + y * y + z * z);
    const double dX[9] = {0, z, -y, 1, 0, 0, Rcw ? Rcw[0] : 0, Rcw ? Rcw[1] : 0, Rcw ? Rcw[2] : 0};
    const double dY[9] = {-z, 0, x, 0, 1, 0, Rcw ? Rcw[3] : 0, Rcw ? Rcw[4] : 0, Rcw ? Rcw[5] : 0};
    const double dZ[9] = {y, -x, 0, 0, 0, 1, Rcw ? Rcw[6] : 0, Rcw ? Rcw[7] : 0, Rcw ? Rcw[8] : 0};
    const double c0 = -(K[2] / (2 * M_PI)) * (1.0 / (x * x + z * z));
    const double c1 = -(K[3] / M_PI) * (1.0 / (L * sqrt(x * x + z * z)));
    for (int k = 0; k < 9; ++k) {
        const double dL = (1.0 / L) * (x * dX[k] + y * dY[k] + z * dZ[k]);
        const double j0 = c0 * (z * dX[k] - x * dZ[k]);
        const double j1 = c1 * (L * dY[k] - y * dL);
        if (k < 6) {
            Bj[k] = j0;
            Bj[6 + k] = j1;
        }
        else if (A) {
            A[k - 6] = j0;
            A[3 + k - 6] = j1;
        }
    }
}
static inline int depth_ok(const double* K, const double* pc) { return cam_is_equirect(K) || 0.0 < pc[2]; }

static void edge_error(const ba_t* B, int e, double* err, double* pc_out) {
    const int p = B->obs_pose[e], l = B->obs_point[e];
    double pc[3];
    se3_map(&B->pose[p], &B->pt[3 * l], pc);
    const double* K = &B->intr[5 * p];
    double u, v;
    if (cam_is_equirect(K)) equirect_project(K, pc, &u, &v);
    else {
        u = K[0] * pc[0] / pc[2] + K[2];
        v = K[1] * pc[1] / pc[2] + K[3];
    }
    err[0] = (double)B->obs_uvr[3 * e] - u;
    err[1] = (double)B->obs_uvr[3 * e + 1] - v;
    if (B->obs_uvr[3 * e + 2] < 0 || cam_is_equirect(K)) err[2] = 0.0;
    else err[2] = (double)B->obs_uvr[3 * e + 2] - (u - K[4] / pc[2]);
    if (pc_out) memcpy(pc_out, pc, sizeof(pc));
}

static inline double edge_chi2(const ba_t* B, int e) {
    const double* r = &B->err[3 * e];
    return (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * (double)B->obs_inv_sigma_sq[e];
}

static void huber(double e, double delta, double* rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e;
        rho[1] = 1.;
    }
    else {
        const double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
    }
}

static void compute_active_errors(ba_t* B) {
    for (int e = 0; e < B->E; ++e)
        if (B->level[e] == 0) edge_error(B, e, &B->err[3 * e], NULL);
}

static double active_robust_chi2(const ba_t* B) {
    double chi = 0;
    for (int e = 0; e < B->E; ++e) {
        if (B->level[e]) continue;
        const double c = edge_chi2(B, e);
        if (B->robust[e]) {
            double rho[2];
            huber(c, (double)B->obs_huber[e], rho);
            chi += rho[0];
        }
        else chi += c;
    }
    return chi;
}

/* initializeOptimization(level 0): active vertices = endpoints of active edges. */
static void build_index(ba_t* B) {
    for (int p = 0; p < B->P; ++p) B->pose_slot[p] = -1;
    for (int l = 0; l < B->L; ++l) B->point_slot[l] = -1;
    uint8_t* pa = (uint8_t*)calloc(B->P, 1);
    uint8_t* la = (uint8_t*)calloc(B->L, 1);
    for (int e = 0; e < B->E; ++e)
        if (B->level[e] == 0) {
            pa[B->obs_pose[e]] = 1;
            la[B->obs_point[e]] = 1;
        }
    B->nP = B->nL = 0;
    for (int p = 0; p < B->P; ++p)
        if (pa[p] && !B->pose_fixed[p]) B->pose_slot[p] = B->nP++;
    for (int l = 0; l < B->L; ++l)
        if (la[l] && !(B->point_fixed && B->point_fixed[l])) B->point_slot[l] = B->nL++;
    free(pa);
    free(la);
}

/* dense LL^T, in place in the lower triangle; returns 0 on success.
 * Rows are walked inside their envelope only: first[i] = column of the first non-zero of row i of the INPUT matrix.  Entries left
 * of the envelope stay exactly zero through the factorisation, and a skipped term of the inner products is an exact 0 * x, so the
 * factor is bit-identical to the plain dense loops (checked by tests/test_oracle_ba.py) -- it only makes the 3000 x 3000 reduced
 * systems of the global-BA configuration (500 keyframes on a loop: a band plus the loop-closure corner) affordable on one core. */
static int chol_factor_env(double* A, int n, int* first) {
    for (int i = 0; i < n; ++i) {
        int f = 0;
        while (f < i && A[(size_t)i * n + f] == 0.0) ++f;
        first[i] = f;
    }
    for (int i = 0; i < n; ++i) {
        const int fi = first[i];
        for (int j = fi; j < i; ++j) {
            double s = A[(size_t)i * n + j];
            const int k0 = fi > first[j] ? fi : first[j];
            for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
        }
        double d = A[(size_t)i * n + i];
        for (int k = fi; k < i; ++k) d -= A[(size_t)i * n + k] * A[(size_t)i * n + k];
        if (!(d > 0)) return -1;
        A[(size_t)i * n + i] = sqrt(d);
    }
    return 0;
}
static int chol_factor(double* A, int n) { /* small systems (pose optimizer 6x6): no envelope bookkeeping */
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve_env(const double* Lm, int n, const int* first, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = first[i]; k < i; ++k) s -= Lm[(size_t)i * n + k] * b[k];
        b[i] = s / Lm[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k)
            if (first[k] <= i) s -= Lm[(size_t)k * n + i] * b[k];
        b[i] = s / Lm[(size_t)i * n + i];
    }
}
static void chol_solve(const double* Lm, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= Lm[(size_t)i * n + k] * b[k];
        b[i] = s / Lm[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= Lm[(size_t)k * n + i] * b[k];
        b[i] = s / Lm[(size_t)i * n + i];
    }
}

/* test hook: both factorisations on the same matrix (tests/test_oracle_ba.py checks bit equality) */
int orc_chol_envelope_check(const double* A_in, int n, const double* b_in, double* x_env, double* x_dense) {
    double* A1 = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* A2 = (double*)malloc(sizeof(double) * (size_t)n * n);
    int* first = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    memcpy(A1, A_in, sizeof(double) * (size_t)n * n);
    memcpy(A2, A_in, sizeof(double) * (size_t)n * n);
    memcpy(x_env, b_in, sizeof(double) * n);
    memcpy(x_dense, b_in, sizeof(double) * n);
    const int r1 = chol_factor_env(A1, n, first), r2 = chol_factor(A2, n);
    if (!r1) chol_solve_env(A1, n, first, x_env);
    if (!r2) chol_solve(A2, n, x_dense);
    int same = r1 == r2;
    if (!r1 && !r2)
        for (int i = 0; i < n && same; ++i)
            for (int j = 0; j <= i; ++j)
                if (A1[(size_t)i * n + j] != A2[(size_t)i * n + j]) {
                    same = 0;
                    break;
                }
    free(A1);
    free(A2);
    free(first);
    return same ? (r1 ? 1 : 0) : -1;
}

static int inv3(const double* A, double* Ai) {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0 || !isfinite(det)) return -1;
    const double id = 1.0 / det;
    Ai[0] = c00 * id;
    Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id;
    Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id;
    Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return 0;
}

/* linear system storage for one LM iteration */
typedef struct {
    double* Hpp; /* nP x 36 */
    double* bp;  /* nP x 6 */
    double* Hll; /* nL x 9 */
    double* bl;  /* nL x 3 */
    double* Hpl; /* E x 18 (6x3, row-major), valid where both endpoints free & active */
    double* xp;  /* nP x 6 */
    double* xl;  /* nL x 3 */
} lin_t;

/* linearizeOplus of the reprojection edges (perspective_reproj_edge.h:67-98 mono, :154-192 stereo rows; equirectangular_reproj_edge.h):
 * A = d e / d landmark (3 x 3, unused rows zero), Bj = d e / d [omega, upsilon] (3 x 6).  The stereo row is always filled for the
 * perspective model; callers use D rows. */
static void reproj_edge_jacobians(const double* K, const double* pc, const double* R, double* A, double* Bj) {
    const double fx = K[0], fy = K[1], fxb = K[4];
    const double x = pc[0], y = pc[1], z = pc[2], z_sq = z * z;
    for (int c = 0; A && c < 3; ++c) {
        A[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_sq;
        A[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_sq;
        A[6 + c] = A[c] - fxb * R[6 + c] / z_sq;
    }
    Bj[0] = x * y / z_sq * fx;
    Bj[1] = -(1.0 + (x * x / z_sq)) * fx;
    Bj[2] = y / z * fx;
    Bj[3] = -1.0 / z * fx;
    Bj[4] = 0.0;
    Bj[5] = x / z_sq * fx;
    Bj[6] = (1.0 + y * y / z_sq) * fy;
    Bj[7] = -x * y / z_sq * fy;
    Bj[8] = -x / z * fy;
    Bj[9] = 0.0;
    Bj[10] = -1.0 / z * fy;
    Bj[11] = y / z_sq * fy;
    Bj[12] = Bj[0] - fxb * y / z_sq;
    Bj[13] = Bj[1] + fxb * x / z_sq;
    Bj[14] = Bj[2];
    Bj[15] = Bj[3];
    Bj[16] = 0;
    Bj[17] = Bj[5] - fxb / z_sq;
    if (cam_is_equirect(K)) {
        if (A) memset(A, 0, sizeof(double) * 9);
        memset(Bj, 0, sizeof(double) * 18);
        equirect_jacobians(K, pc, A ? R : NULL, A, Bj);
    }
}

static void build_system(const ba_t* B, lin_t* S) {
    memset(S->Hpp, 0, sizeof(double) * 36 * (size_t)B->nP);
    memset(S->bp, 0, sizeof(double) * 6 * (size_t)B->nP);
    memset(S->Hll, 0, sizeof(double) * 9 * (size_t)B->nL);
    memset(S->bl, 0, sizeof(double) * 3 * (size_t)B->nL);
    for (int e = 0; e < B->E; ++e) {
        if (B->level[e]) continue;
        const int p = B->obs_pose[e], l = B->obs_point[e];
        const int ps = B->pose_slot[p], lsl = B->point_slot[l];
        if (ps < 0 && lsl < 0) continue;
        const double* K = &B->intr[5 * p];
        const int stereo = !(B->obs_uvr[3 * e + 2] < 0) && !cam_is_equirect(K);
        const int D = stereo ? 3 : 2;
        const double fx = K[0], fy = K[1], fxb = K[4];
        double pc[3], R[9];
        se3_map(&B->pose[p], &B->pt[3 * l], pc);
        quat_to_R(B->pose[p].q, R);
        const double x = pc[0], y = pc[1], z = pc[2], z_sq = z * z;
        double A[9], Bj[18]; /* A: D x 3 (d e / d landmark), Bj: D x 6 (d e / d pose) */
        (void)x, (void)y, (void)z_sq, (void)fx, (void)fy, (void)fxb;
        reproj_edge_jacobians(K, pc, R, A, Bj);
        const double* r = &B->err[3 * e];
        double w = (double)B->obs_inv_sigma_sq[e];
        double rw = w; /* weight on the residual in b: rho' * omega */
        if (B->robust[e]) {
            double rho[2];
            huber(edge_chi2(B, e), (double)B->obs_huber[e], rho);
            w *= rho[1];
            rw = w;
        }
        if (lsl >= 0) {
            double* H = &S->Hll[9 * lsl];
            double* b = &S->bl[3 * lsl];
            for (int i = 0; i < 3; ++i) {
                double s = 0;
                for (int d = 0; d < D; ++d) s += A[3 * d + i] * (-rw * r[d]);
                b[i] += s;
                for (int j = 0; j < 3; ++j) {
                    double h = 0;
                    for (int d = 0; d < D; ++d) h += A[3 * d + i] * w * A[3 * d + j];
                    H[3 * i + j] += h;
                }
            }
        }
        if (ps >= 0) {
            double* H = &S->Hpp[36 * ps];
            double* b = &S->bp[6 * ps];
            for (int i = 0; i < 6; ++i) {
                double s = 0;
                for (int d = 0; d < D; ++d) s += Bj[6 * d + i] * (-rw * r[d]);
                b[i] += s;
                for (int j = 0; j < 6; ++j) {
                    double h = 0;
                    for (int d = 0; d < D; ++d) h += Bj[6 * d + i] * w * Bj[6 * d + j];
                    H[6 * i + j] += h;
                }
            }
        }
        if (ps >= 0 && lsl >= 0) {
            double* H = &S->Hpl[18 * (size_t)e];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) {
                    double h = 0;
                    for (int d = 0; d < D; ++d) h += Bj[6 * d + i] * w * A[3 * d + j];
                    H[3 * i + j] = h;
                }
        }
    }
}

/* BlockSolver::solve with lambda on every diagonal; returns 0 ok / -1 failed */
static int solve_system(const ba_t* B, lin_t* S, double lambda, const int* lm_edge_off, const int* lm_edges) {
    const int n = 6 * B->nP;
    double* Hs = (double*)calloc((size_t)n * n + 1, sizeof(double));
    double* bs = (double*)calloc(n + 1, sizeof(double));
    double* Dinv = (double*)malloc(sizeof(double) * 9 * (size_t)(B->nL + 1));
    int fail = 0;
    for (int p = 0; p < B->nP; ++p) {
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) Hs[(size_t)(6 * p + i) * n + 6 * p + j] = S->Hpp[36 * p + 6 * i + j];
            Hs[(size_t)(6 * p + i) * n + 6 * p + i] += lambda;
            bs[6 * p + i] = S->bp[6 * p + i];
        }
    }
    for (int l = 0; l < B->L; ++l) {
        const int sl = B->point_slot[l];
        if (sl < 0) continue;
        double D[9];
        memcpy(D, &S->Hll[9 * sl], sizeof(D));
        D[0] += lambda;
        D[4] += lambda;
        D[8] += lambda;
        if (inv3(D, &Dinv[9 * sl])) {
            fail = 1;
            memset(&Dinv[9 * sl], 0, sizeof(double) * 9);
        }
        const double* Di = &Dinv[9 * sl];
        const double* bl = &S->bl[3 * sl];
        const double db[3] = {Di[0] * bl[0] + Di[1] * bl[1] + Di[2] * bl[2], Di[3] * bl[0] + Di[4] * bl[1] + Di[5] * bl[2],
                              Di[6] * bl[0] + Di[7] * bl[1] + Di[8] * bl[2]};
        for (int a = lm_edge_off[l]; a < lm_edge_off[l + 1]; ++a) {
            const int e1 = lm_edges[a];
            if (B->level[e1]) continue;
            const int p1 = B->pose_slot[B->obs_pose[e1]];
            if (p1 < 0) continue;
            const double* W1 = &S->Hpl[18 * (size_t)e1];
            double Y[18]; /* W1 * Dinv (6x3) */
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) Y[3 * i + j] = W1[3 * i] * Di[j] + W1[3 * i + 1] * Di[3 + j] + W1[3 * i + 2] * Di[6 + j];
            for (int i = 0; i < 6; ++i) bs[6 * p1 + i] -= W1[3 * i] * db[0] + W1[3 * i + 1] * db[1] + W1[3 * i + 2] * db[2];
            for (int c = lm_edge_off[l]; c < lm_edge_off[l + 1]; ++c) {
                const int e2 = lm_edges[c];
                if (B->level[e2]) continue;
                const int p2 = B->pose_slot[B->obs_pose[e2]];
                if (p2 < 0) continue;
                const double* W2 = &S->Hpl[18 * (size_t)e2];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j)
                        Hs[(size_t)(6 * p1 + i) * n + 6 * p2 + j] -= Y[3 * i] * W2[3 * j] + Y[3 * i + 1] * W2[3 * j + 1] + Y[3 * i + 2] * W2[3 * j + 2];
            }
        }
    }
    if (n > 0) {
        int* first = (int*)malloc(sizeof(int) * (size_t)(n + 1));
        if (chol_factor_env(Hs, n, first)) fail = 1;
        else chol_solve_env(Hs, n, first, bs);
        free(first);
    }
    memcpy(S->xp, bs, sizeof(double) * n);
    for (int l = 0; l < B->L; ++l) {
        const int sl = B->point_slot[l];
        if (sl < 0) continue;
        double c[3] = {S->bl[3 * sl], S->bl[3 * sl + 1], S->bl[3 * sl + 2]};
        for (int a = lm_edge_off[l]; a < lm_edge_off[l + 1]; ++a) {
            const int e1 = lm_edges[a];
            if (B->level[e1]) continue;
            const int p1 = B->pose_slot[B->obs_pose[e1]];
            if (p1 < 0) continue;
            const double* W1 = &S->Hpl[18 * (size_t)e1];
            const double* xp = &S->xp[6 * p1];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 6; ++i) c[j] -= W1[3 * i + j] * xp[i];
        }
        const double* Di = &Dinv[9 * sl];
        for (int i = 0; i < 3; ++i) S->xl[3 * sl + i] = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
    }
    free(Hs);
    free(bs);
    free(Dinv);
    return fail ? -1 : 0;
}

static void apply_update(ba_t* B, const lin_t* S) {
    for (int p = 0; p < B->P; ++p) {
        const int s = B->pose_slot[p];
        if (s < 0) continue;
        se3q ex;
        se3_exp(&S->xp[6 * s], &ex);
        se3_mul(&ex, &B->pose[p], &B->pose[p]);
    }
    for (int l = 0; l < B->L; ++l) {
        const int s = B->point_slot[l];
        if (s < 0) continue;
        for (int i = 0; i < 3; ++i) B->pt[3 * l + i] += S->xl[3 * s + i];
    }
}

typedef struct {
    double lambda, ni;
    double last_chi; /* terminate_action::_lastChi */
} lm_t;

/* terminate_action::operator() for iteration >= 0 (optimize/terminate_action.cc:52-73; _maxIterations keeps g2o's default INT_MAX):
 * iteration 0 stores the chi2, later ones stop when 0 <= (last - now) / now < threshold.  Returns 1 when the stop flag is to be raised. */
static int terminate_rule(double* last_chi, int it, double chi_now, double gain_thr) {
    if (it == 0) {
        *last_chi = chi_now;
        return 0;
    }
    const double gain = (*last_chi - chi_now) / chi_now;
    *last_chi = chi_now;
    return gain >= 0 && gain < gain_thr;
}

/* One SparseOptimizer::optimize(iterations) call incl. the terminate_action post-iteration hook.
 * stop: the optimizer's force-stop flag (never NULL here; see orc_local_ba).  Returns iterations run. */
static int optimize(ba_t* B, int iterations, double gain_thr, volatile uint8_t* stop, const int* lm_edge_off,
                    const int* lm_edges, double* trace /* per iteration: chi2 after, lambda; nullable */) {
    build_index(B);
    if (B->nP + B->nL == 0) return 0;
    lin_t S;
    S.Hpp = (double*)malloc(sizeof(double) * 36 * (size_t)(B->nP + 1));
    S.bp = (double*)malloc(sizeof(double) * 6 * (size_t)(B->nP + 1));
    S.Hll = (double*)malloc(sizeof(double) * 9 * (size_t)(B->nL + 1));
    S.bl = (double*)malloc(sizeof(double) * 3 * (size_t)(B->nL + 1));
    S.Hpl = (double*)malloc(sizeof(double) * 18 * (size_t)(B->E + 1));
    S.xp = (double*)malloc(sizeof(double) * 6 * (size_t)(B->nP + 1));
    S.xl = (double*)malloc(sizeof(double) * 3 * (size_t)(B->nL + 1));
    se3q* pose_bak = (se3q*)malloc(sizeof(se3q) * B->P);
    double* pt_bak = (double*)malloc(sizeof(double) * 3 * (size_t)B->L);
    lm_t lm = {0, 2, 0};
    int done = 0, ok = 1;
    for (int it = 0; it < iterations && !*stop && ok; ++it) {
        /* OptimizationAlgorithmLevenberg::solve(it) */
        compute_active_errors(B);
        double current_chi = active_robust_chi2(B);
        double temp_chi = current_chi;
        build_system(B, &S);
        if (it == 0) {
            double max_diag = 0;
            for (int p = 0; p < B->nP; ++p)
                for (int j = 0; j < 6; ++j) max_diag = fmax(fabs(S.Hpp[36 * p + 7 * j]), max_diag);
            for (int l = 0; l < B->nL; ++l)
                for (int j = 0; j < 3; ++j) max_diag = fmax(fabs(S.Hll[9 * l + 4 * j]), max_diag);
            lm.lambda = 1e-5 * max_diag;
            lm.ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(pose_bak, B->pose, sizeof(se3q) * B->P); /* push */
            memcpy(pt_bak, B->pt, sizeof(double) * 3 * (size_t)B->L);
            const int ok2 = solve_system(B, &S, lm.lambda, lm_edge_off, lm_edges) == 0;
            apply_update(B, &S);
            compute_active_errors(B);
            temp_chi = active_robust_chi2(B);
            if (!ok2) temp_chi = DBL_MAX;
            rho = (current_chi - temp_chi);
            double scale = 0;
            for (int j = 0; j < 6 * B->nP; ++j) scale += S.xp[j] * (lm.lambda * S.xp[j] + S.bp[j]);
            for (int j = 0; j < 3 * B->nL; ++j) scale += S.xl[j] * (lm.lambda * S.xl[j] + S.bl[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                const double scale_factor = fmax(1. / 3., alpha);
                lm.lambda *= scale_factor;
                lm.ni = 2;
                current_chi = temp_chi;
            }
            else {
                lm.lambda *= lm.ni;
                lm.ni *= 2;
                memcpy(B->pose, pose_bak, sizeof(se3q) * B->P); /* pop */
                memcpy(B->pt, pt_bak, sizeof(double) * 3 * (size_t)B->L);
                if (!isfinite(lm.lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !*stop);
        if (qmax == 10 || rho == 0 || !isfinite(lm.lambda)) ok = 0; /* Terminate */
        ++done;
        /* postIteration(it): terminate_action.cc:36-76 */
        compute_active_errors(B);
        const double chi_now = active_robust_chi2(B);
        if (trace) {
            trace[2 * it] = chi_now;
            trace[2 * it + 1] = lm.lambda;
        }
        if (terminate_rule(&lm.last_chi, it, chi_now, gain_thr)) *stop = 1;
    }
    free(S.Hpp);
    free(S.bp);
    free(S.Hll);
    free(S.bl);
    free(S.Hpl);
    free(S.xp);
    free(S.xl);
    free(pose_bak);
    free(pt_bak);
    return done;
}

/*
 * Local BA on flat arrays.
 *   pose_cw        P x 12  rows of [R|t] (3x4 row-major), world -> camera
 *   pose_fixed     P       1 = fixed keyframe
 *   points         L x 3
 *   point_fixed    L       nullable (markers kept fixed)
 *   obs_*          E       pose index, point index, (u, v, u_right<0 => mono) f32, inv_sigma_sq f32,
 *                          huber delta f32 (<= 0 => no kernel; < 0 => a marker-corner edge: also outside the gate and the outlier list)
 *   intr           P x 5   fx fy cx cy fx*baseline
 *   stop           caller's force_stop_flag, nullable.  Non-NULL: polled between iterations AND written
 *                  by the terminate rule (reference quirk) so stage 2 is skipped after an
 *                  early stage-1 stop.  NULL: g2o installs an internal flag with the same effect on stage 2's
 *                  LM loop, but the outlier gate + kernel removal still run.
 *   outlier_out    E       final outlier flags (local_bundle_adjuster_g2o.cc:352-375)
 *   stats          8       [0] chi2 before, [1] chi2 after, [2] iters stage 1, [3] iters stage 2,
 *                          [4] stage 2 entered (0/1), [5] #edges gated to level 1
 *   trace          nullable, 2*(iters1+iters2): chi2 and lambda after each LM iteration
 */
int orc_local_ba(int P, int L, int E, const double* pose_cw, const uint8_t* pose_fixed, const double* points,
                 const uint8_t* point_fixed, const int32_t* obs_pose, const int32_t* obs_point, const float* obs_uvr,
                 const float* obs_inv_sigma_sq, const float* obs_huber, const double* intr, int iters1, int iters2,
                 double gain_thr, volatile uint8_t* stop, double* pose_out, double* points_out, uint8_t* outlier_out,
                 double* stats, double* trace) {
    ba_t B;
    memset(&B, 0, sizeof(B));
    B.P = P;
    B.L = L;
    B.E = E;
    B.pose_fixed = pose_fixed;
    B.point_fixed = point_fixed;
    B.obs_pose = obs_pose;
    B.obs_point = obs_point;
    B.obs_uvr = obs_uvr;
    B.obs_inv_sigma_sq = obs_inv_sigma_sq;
    B.obs_huber = obs_huber;
    B.intr = intr;
    B.pose = (se3q*)malloc(sizeof(se3q) * (P + 1));
    B.pt = (double*)malloc(sizeof(double) * 3 * (size_t)(L + 1));
    B.level = (uint8_t*)calloc(E + 1, 1);
    B.robust = (uint8_t*)calloc(E + 1, 1);
    B.err = (double*)calloc(3 * (size_t)(E + 1), sizeof(double));
    B.pose_slot = (int*)malloc(sizeof(int) * (P + 1));
    B.point_slot = (int*)malloc(sizeof(int) * (L + 1));
    for (int p = 0; p < P; ++p) {
        const double* M = &pose_cw[12 * p];
        const double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        quat_from_R(R, B.pose[p].q);
        B.pose[p].t[0] = M[3];
        B.pose[p].t[1] = M[7];
        B.pose[p].t[2] = M[11];
        quat_normalize(&B.pose[p]);
    }
    memcpy(B.pt, points, sizeof(double) * 3 * (size_t)L);
    for (int e = 0; e < E; ++e) B.robust[e] = obs_huber && obs_huber[e] > 0;
    /* landmark -> edges CSR */
    int* off = (int*)calloc(L + 2, sizeof(int));
    int* items = (int*)malloc(sizeof(int) * (E + 1));
    for (int e = 0; e < E; ++e) off[obs_point[e] + 1]++;
    for (int l = 0; l < L; ++l) off[l + 1] += off[l];
    int* fill = (int*)calloc(L + 1, sizeof(int));
    for (int e = 0; e < E; ++e) items[off[obs_point[e]] + fill[obs_point[e]]++] = e;
    free(fill);

    uint8_t aux_flag = 0;
    volatile uint8_t* flag = stop ? stop : &aux_flag;
    double st[8] = {0};
    compute_active_errors(&B);
    st[0] = active_robust_chi2(&B);
    int rc = 0;
    if (stop && *stop) {
        rc = 1; /* local_bundle_adjuster_g2o.cc:308-310: silent early return, nothing written back */
    }
    else {
        st[2] = optimize(&B, iters1, gain_thr, flag, off, items, trace);
        int run_robust = 1;
        if (stop && *stop) run_robust = 0;
        if (run_robust) {
            st[4] = 1;
            for (int e = 0; e < E; ++e) {
                const double chi = edge_chi2(&B, e);
                double pc[3];
                se3_map(&B.pose[obs_pose[e]], &B.pt[3 * obs_point[e]], pc);
                const int mono = obs_uvr[3 * e + 2] < 0;
                const float thr = mono ? 5.99146f : 7.81473f;
                /* huber < 0: a marker-corner edge -- its own container (:246-304), which the gate loop :324-343 does not visit */
                const int marker_edge = obs_huber && obs_huber[e] < 0;
                if (!marker_edge && ((double)thr < chi || !depth_ok(&B.intr[5 * obs_pose[e]], pc))) {
                    B.level[e] = 1;
                    st[5] += 1;
                }
                B.robust[e] = 0;
            }
            st[3] = optimize(&B, iters2, gain_thr, flag, off, items, trace ? trace + 2 * iters1 : NULL);
        }
        for (int e = 0; e < E; ++e) {
            const double chi = edge_chi2(&B, e); /* cached error: stale for level-1 edges, as in g2o */
            double pc[3];
            se3_map(&B.pose[obs_pose[e]], &B.pt[3 * obs_point[e]], pc);
            const int mono = obs_uvr[3 * e + 2] < 0;
            const float thr = mono ? 5.99146f : 7.81473f;
            const int marker_edge = obs_huber && obs_huber[e] < 0; /* nor does the outlier loop :354-375 */
            outlier_out[e] = (!marker_edge && ((double)thr < chi || !depth_ok(&B.intr[5 * obs_pose[e]], pc))) ? 1 : 0;
        }
        /* chi2 over the finally-active set */
        st[1] = active_robust_chi2(&B);
    }
    for (int p = 0; p < P; ++p) {
        double R[9];
        quat_to_R(B.pose[p].q, R);
        double* M = &pose_out[12 * p];
        for (int i = 0; i < 3; ++i) {
            M[4 * i] = R[3 * i];
            M[4 * i + 1] = R[3 * i + 1];
            M[4 * i + 2] = R[3 * i + 2];
            M[4 * i + 3] = B.pose[p].t[i];
        }
    }
    memcpy(points_out, B.pt, sizeof(double) * 3 * (size_t)L);
    if (stats) memcpy(stats, st, sizeof(st));
    free(off);
    free(items);
    free(B.pose);
    free(B.pt);
    free(B.level);
    free(B.robust);
    free(B.err);
    free(B.pose_slot);
    free(B.point_slot);
    return rc;
}


/* ---------------------------------------------------------------- pose_optimizer (motion-only BA)
 * optimize/pose_optimizer_g2o.cc:38-175; unary edges optimize/internal/se3/perspective_pose_opt_edge.h:68-110 (mono) and the
 * stereo variant; wrapper pose_opt_edge_wrapper.h (Huber delta = sqrt(chi_sq) by camera setup, level 0/1 = inlier/outlier).
 *   n observations of FIXED 3-D points: pos_w (n x 3), uvr (n x 3 f32, u_right < 0 => mono), inv_sigma_sq, huber (per obs)
 *   (num_trials_robust + num_trials) rounds of: initializeOptimization (level-0 edges) -> optimize(num_each_iter) -> chi-square
 *   re-classification of EVERY edge at the current pose; kernels are removed after round num_trials_robust.
 * No force-stop flag is installed by the reference, so g2o's terminate action keeps its own flag; SparseOptimizer::optimize
 * never sends the "iteration -1" reset, hence once the gain rule has fired the later rounds run zero LM iterations
 * (reset_flag_each_round = 0, the literal behaviour; 1 = reset per round, the behaviour if g2o did send that reset).
 * Returns num_valid (0 when fewer than 5 observations), pose_out 3x4 row-major, outlier[n]. */
/* errors of observation i at pose T (pose-only edges: *_pose_opt_edge.h computeError) and its chi2 = e^T (inv_sigma_sq I) e */
static void po_err(const se3q* T, const double* pos_w, const float* uvr, const double* intr, int i, double* err) {
    double pc[3];
    se3_map(T, &pos_w[3 * i], pc);
    double u = intr[0] * pc[0] / pc[2] + intr[2], v = intr[1] * pc[1] / pc[2] + intr[3];
    if (cam_is_equirect(intr)) equirect_project(intr, pc, &u, &v);
    err[3 * i] = (double)uvr[3 * i] - u;
    err[3 * i + 1] = (double)uvr[3 * i + 1] - v;
    err[3 * i + 2] = (uvr[3 * i + 2] < 0 || cam_is_equirect(intr)) ? 0.0 : (double)uvr[3 * i + 2] - (u - intr[4] / pc[2]);
}
static double po_chi(const double* err, const float* inv_sigma_sq, int i) {
    return (err[3 * i] * err[3 * i] + err[3 * i + 1] * err[3 * i + 1] + err[3 * i + 2] * err[3 * i + 2]) * (double)inv_sigma_sq[i];
}

/* One SparseOptimizer::optimize(num_each_iter) of the pose-only graph: Levenberg-Marquardt over the level-0 edges with their kernels,
 * the terminate rule behind every iteration (its _lastChi and the stop flag outlive the call, as the action object and g2o's flag do).
 * Returns the LM iterations run. */
static int po_optimize(se3q* Tp, int n, const double* pos_w, const float* uvr, const float* inv_sigma_sq, const float* huber_delta,
                       const double* intr, const uint8_t* level, const uint8_t* robust, int num_each_iter, double gain_thr, uint8_t* flagp,
                       double* last_chip, double* err) {
    se3q T = *Tp;
    uint8_t flag = *flagp;
    double last_chi = *last_chip;
    int total_iters = 0, nact = 0;
    for (int i = 0; i < n; ++i) nact += !level[i];
#define PO_ERR(i, Tq) po_err((Tq), pos_w, uvr, intr, (i), err);
#define PO_CHI(i) po_chi(err, inv_sigma_sq, (i))
    {
        int ok = 1;
        double lambda = 0, ni = 2;
        for (int it = 0; it < num_each_iter && !flag && ok && nact > 0; ++it) {
            double H[36] = {0}, b[6] = {0}, cur = 0;
            for (int i = 0; i < n; ++i) {
                if (level[i]) continue;
                PO_ERR(i, &T)
                double pc[3];
                se3_map(&T, &pos_w[3 * i], pc);
                const int stereo = !(uvr[3 * i + 2] < 0) && !cam_is_equirect(intr);
                double J[18]; /* the unary edges' pose block = the binary edges' (perspective_pose_opt_edge.h:75-98, :175-204; equirectangular_pose_opt_edge.h:70-118) */
                reproj_edge_jacobians(intr, pc, NULL, NULL, J);
                if (!stereo) memset(&J[12], 0, 6 * sizeof(double));
                const double chi = PO_CHI(i);
                double w = (double)inv_sigma_sq[i], rho[2] = {chi, 1.0};
                if (robust[i]) huber(chi, (double)huber_delta[i], rho);
                cur += robust[i] ? rho[0] : chi;
                w *= rho[1];
                for (int a = 0; a < 6; ++a) {
                    double s = 0;
                    for (int d = 0; d < 3; ++d) s += J[6 * d + a] * (-w * err[3 * i + d]);
                    b[a] += s;
                    for (int c = 0; c < 6; ++c) {
                        double h = 0;
                        for (int d = 0; d < 3; ++d) h += J[6 * d + a] * w * J[6 * d + c];
                        H[6 * a + c] += h;
                    }
                }
            }
            if (it == 0) {
                double md = 0;
                for (int a = 0; a < 6; ++a) md = fmax(md, fabs(H[7 * a]));
                lambda = 1e-5 * md;
                ni = 2;
            }
            double rho_ = 0;
            int qmax = 0;
            do {
                const se3q bak = T;
                double A[36], x6[6];
                memcpy(A, H, sizeof(A));
                for (int a = 0; a < 6; ++a) A[7 * a] += lambda;
                memcpy(x6, b, sizeof(x6));
                const int ok2 = chol_factor(A, 6) == 0;
                if (ok2) chol_solve(A, 6, x6);
                else memset(x6, 0, sizeof(x6));
                se3q ex;
                se3_exp(x6, &ex);
                se3_mul(&ex, &T, &T);
                double tmp = 0;
                for (int i = 0; i < n; ++i) {
                    if (level[i]) continue;
                    PO_ERR(i, &T)
                    const double chi = PO_CHI(i);
                    double r2[2] = {chi, 1.0};
                    if (robust[i]) huber(chi, (double)huber_delta[i], r2);
                    tmp += robust[i] ? r2[0] : chi;
                }
                if (!ok2) tmp = DBL_MAX;
                rho_ = cur - tmp;
                double scale = 1e-3;
                for (int a = 0; a < 6; ++a) scale += x6[a] * (lambda * x6[a] + b[a]);
                rho_ /= scale;
                if (rho_ > 0 && isfinite(tmp)) {
                    double alpha = 1. - pow((2 * rho_ - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    cur = tmp;
                }
                else {
                    lambda *= ni;
                    ni *= 2;
                    T = bak;
                    if (!isfinite(lambda)) break;
                }
                ++qmax;
            } while (rho_ < 0 && qmax < 10 && !flag);
            if (qmax == 10 || rho_ == 0 || !isfinite(lambda)) ok = 0;
            ++total_iters;
            if (terminate_rule(&last_chi, it, cur, gain_thr)) flag = 1;
        }
    }
#undef PO_ERR
#undef PO_CHI
    *Tp = T;
    *flagp = flag;
    *last_chip = last_chi;
    return total_iters;
}

int orc_pose_optimize(const double* pose_cw, int n, const double* pos_w, const float* uvr, const float* inv_sigma_sq,
                      const float* huber_delta, const double* intr, int num_trials_robust, int num_trials, int num_each_iter,
                      double gain_thr, int reset_flag_each_round, double* pose_out, uint8_t* outlier, double* stats) {
    se3q T;
    {
        const double R[9] = {pose_cw[0], pose_cw[1], pose_cw[2], pose_cw[4], pose_cw[5], pose_cw[6], pose_cw[8], pose_cw[9], pose_cw[10]};
        quat_from_R(R, T.q);
        T.t[0] = pose_cw[3];
        T.t[1] = pose_cw[7];
        T.t[2] = pose_cw[11];
        quat_normalize(&T);
    }
    memcpy(pose_out, pose_cw, sizeof(double) * 12);
    for (int i = 0; i < n; ++i) outlier[i] = 0;
    if (stats) memset(stats, 0, sizeof(double) * 4);
    if (n < 5) return 0;
    uint8_t* level = (uint8_t*)calloc(n, 1);
    uint8_t* robust = (uint8_t*)malloc(n);
    double* err = (double*)calloc(3 * (size_t)n, sizeof(double));
    for (int i = 0; i < n; ++i) robust[i] = (num_trials_robust != 0) && huber_delta[i] > 0;
    uint8_t flag = 0;
    double last_chi = 0;
    int num_bad = 0, total_iters = 0;
    for (int trial = 0; trial < num_trials_robust + num_trials; ++trial) {
        if (reset_flag_each_round) flag = 0;
        /* ---- optimizer.optimize(num_each_iter) */
        total_iters += po_optimize(&T, n, pos_w, uvr, inv_sigma_sq, huber_delta, intr, level, robust, num_each_iter, gain_thr, &flag, &last_chi, err);
        /* ---- re-classification at the current pose (:127-160) */
        num_bad = 0;
        for (int i = 0; i < n; ++i) {
            po_err(&T, pos_w, uvr, intr, i, err);
            const float thr = uvr[3 * i + 2] < 0 ? 5.99146f : 7.81473f;
            if ((double)thr < po_chi(err, inv_sigma_sq, i)) {
                outlier[i] = 1;
                level[i] = 1;
                ++num_bad;
            }
            else {
                outlier[i] = 0;
                level[i] = 0;
            }
            if (num_trials != 0 && trial + 1 == num_trials_robust) robust[i] = 0;
        }
        if (n - num_bad < 5) break;
    }
    double R[9];
    quat_to_R(T.q, R);
    for (int i = 0; i < 3; ++i) {
        pose_out[4 * i] = R[3 * i];
        pose_out[4 * i + 1] = R[3 * i + 1];
        pose_out[4 * i + 2] = R[3 * i + 2];
        pose_out[4 * i + 3] = T.t[i];
    }
    if (stats) {
        stats[0] = total_iters;
        stats[1] = num_bad;
    }
    free(level);
    free(robust);
    free(err);
    return n - num_bad;
}


/* ---------------------------------------------------------------- test hooks (oracle/ref_local pins these against the reference's edge
 * and vertex classes; the g2o stand-in's SE3Quat forwards to the first four) */
void orc_dbg_se3_from_Rt(const double* R_row_major, const double* t, double* q4, double* t3) {
    se3q T;
    quat_from_R(R_row_major, T.q);
    memcpy(T.t, t, sizeof(T.t));
    quat_normalize(&T);
    memcpy(q4, T.q, sizeof(T.q));
    memcpy(t3, T.t, sizeof(T.t));
}
void orc_dbg_se3_exp_mul(const double* upd6, const double* q4, const double* t3, double* q4_out, double* t3_out) { /* exp(upd) * T */
    se3q E, T, O;
    se3_exp(upd6, &E);
    memcpy(T.q, q4, sizeof(T.q));
    memcpy(T.t, t3, sizeof(T.t));
    se3_mul(&E, &T, &O);
    memcpy(q4_out, O.q, sizeof(O.q));
    memcpy(t3_out, O.t, sizeof(O.t));
}
void orc_dbg_se3_map(const double* q4, const double* t3, const double* p, double* out) {
    se3q T;
    memcpy(T.q, q4, sizeof(T.q));
    memcpy(T.t, t3, sizeof(T.t));
    se3_map(&T, p, out);
}
void orc_dbg_quat_to_R(const double* q4, double* R_row_major) { quat_to_R(q4, R_row_major); }
/* one reprojection edge exactly as build_system / edge_error see it: err[3], A[9], Bj[18]; returns D (2 or 3) */
int orc_dbg_reproj_edge(const double* q4, const double* t3, const double* point, const double* K5, const float* uvr, double* err, double* A, double* Bj) {
    se3q T;
    memcpy(T.q, q4, sizeof(T.q));
    memcpy(T.t, t3, sizeof(T.t));
    double pc[3], R[9], u, v;
    se3_map(&T, point, pc);
    quat_to_R(T.q, R);
    if (cam_is_equirect(K5)) equirect_project(K5, pc, &u, &v);
    else {
        u = K5[0] * pc[0] / pc[2] + K5[2];
        v = K5[1] * pc[1] / pc[2] + K5[3];
    }
    const int stereo = !(uvr[2] < 0) && !cam_is_equirect(K5);
    err[0] = (double)uvr[0] - u;
    err[1] = (double)uvr[1] - v;
    err[2] = stereo ? (double)uvr[2] - (u - K5[4] / pc[2]) : 0.0;
    reproj_edge_jacobians(K5, pc, R, A, Bj);
    return stereo ? 3 : 2;
}
/* the post-iteration stop rule over a scripted sequence of (iteration, chi2): stop[k] = 1 where the rule raises the flag */
void orc_dbg_terminate(int n, const int* iteration, const double* chi2, double gain_thr, uint8_t* stop, double* last_chi_out) {
    double last = 0.0;
    for (int k = 0; k < n; ++k) {
        stop[k] = (uint8_t)terminate_rule(&last, iteration[k], chi2[k], gain_thr);
        last_chi_out[k] = last;
    }
}
/* one optimize(num_each_iter) call of the pose-only graph on caller-held state (q4 / t3 pose, level / robust per observation, the stop flag
 * and _lastChi): what oracle/ref_local's g2o stand-in runs behind the reference's pose_optimizer_g2o.cc */
int orc_dbg_pose_lm(double* q4, double* t3, int n, const double* pos_w, const float* uvr, const float* inv_sigma_sq, const float* huber_delta,
                    const double* intr, const uint8_t* level, const uint8_t* robust, int num_each_iter, double gain_thr, uint8_t* flag,
                    double* last_chi) {
    se3q T;
    memcpy(T.q, q4, sizeof(T.q));
    memcpy(T.t, t3, sizeof(T.t));
    double* err = (double*)calloc(3 * (size_t)(n > 0 ? n : 1), sizeof(double));
    const int iters = po_optimize(&T, n, pos_w, uvr, inv_sigma_sq, huber_delta, intr, level, robust, num_each_iter, gain_thr, flag, last_chi, err);
    free(err);
    memcpy(q4, T.q, sizeof(T.q));
    memcpy(t3, T.t, sizeof(T.t));
    return iters;
}
/* one SparseOptimizer::optimize(iterations) of the BA graph on caller-held state (poses as q4 / t3, points, per-edge level / kernel flags,
 * cached errors): what oracle/ref_local's g2o stand-in runs behind the reference's local_bundle_adjuster_g2o.cc.  err (E x 3) goes in and
 * out: the errors g2o would have cached on the edges. */
int orc_dbg_ba_lm(int P, int L, int E, double* q4, double* t3, const uint8_t* pose_fixed, double* pts, const uint8_t* point_fixed,
                  const int32_t* obs_pose, const int32_t* obs_point, const float* obs_uvr, const float* obs_inv_sigma_sq, const float* obs_huber,
                  const uint8_t* level, const uint8_t* robust, const double* intr, int iterations, double gain_thr, uint8_t* stop, double* err) {
    ba_t B;
    memset(&B, 0, sizeof(B));
    B.P = P, B.L = L, B.E = E;
    B.pose_fixed = pose_fixed, B.point_fixed = point_fixed;
    B.obs_pose = obs_pose, B.obs_point = obs_point, B.obs_uvr = obs_uvr, B.obs_inv_sigma_sq = obs_inv_sigma_sq, B.obs_huber = obs_huber;
    B.intr = intr;
    B.pose = (se3q*)malloc(sizeof(se3q) * (P + 1));
    B.pt = pts;
    B.level = (uint8_t*)malloc(E + 1);
    B.robust = (uint8_t*)malloc(E + 1);
    B.err = err;
    B.pose_slot = (int*)malloc(sizeof(int) * (P + 1));
    B.point_slot = (int*)malloc(sizeof(int) * (L + 1));
    for (int p = 0; p < P; ++p) {
        memcpy(B.pose[p].q, q4 + 4 * p, sizeof(double) * 4);
        memcpy(B.pose[p].t, t3 + 3 * p, sizeof(double) * 3);
    }
    memcpy(B.level, level, E);
    memcpy(B.robust, robust, E);
    int* off = (int*)calloc(L + 2, sizeof(int));
    int* items = (int*)malloc(sizeof(int) * (E + 1));
    for (int e = 0; e < E; ++e) off[obs_point[e] + 1]++;
    for (int l = 0; l < L; ++l) off[l + 1] += off[l];
    int* fill = (int*)calloc(L + 1, sizeof(int));
    for (int e = 0; e < E; ++e) items[off[obs_point[e]] + fill[obs_point[e]]++] = e;
    free(fill);
    uint8_t aux = 0;
    const int it = optimize(&B, iterations, gain_thr, stop ? stop : &aux, off, items, NULL);
    for (int p = 0; p < P; ++p) {
        memcpy(q4 + 4 * p, B.pose[p].q, sizeof(double) * 4);
        memcpy(t3 + 3 * p, B.pose[p].t, sizeof(double) * 3);
    }
    free(off);
    free(items);
    free(B.pose);
    free(B.level);
    free(B.robust);
    free(B.pose_slot);
    free(B.point_slot);
    return it;
}
