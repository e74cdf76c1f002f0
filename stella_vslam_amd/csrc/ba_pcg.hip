// Preconditioned conjugate gradients on the block-sparse reduced camera system  S dp = g  of bundle adjustment
// (BlockSolver_6_3's reduced system: optimize/local_bundle_adjuster_g2o.cc:151-164, optimize/global_bundle_adjuster.cc:66-85;
// g2o solves it with a sparse Cholesky, LinearSolverEigen / LinearSolverCSparse; g2o's own alternative is LinearSolverPCG with
// the same block-Jacobi preconditioner used here).
//
// Why explicit blocks and not an implicit Schur product: S has one 6x6 block per covisible keyframe pair (6.6 k blocks = 1.9 MB at
// 500 keyframes), while S p through W Hll^-1 W^T would re-read every observation's 6x3 block on every iteration (E x 144 B =
// 173 MB per product at 1.2 M observations).  The blocks are built once per damping trial by k_ba_schur.
//
// One kernel launch per iteration (Chronopoulos-Gear form: both inner products of an iteration are taken at the same point,
// so an iteration needs ONE grid-wide dependency, and a dependent kernel boundary (~1.5 us) is the cheapest grid-wide
// dependency on this part -- cheaper than a device-scope barrier across 8 XCDs):
//   u = M^-1 r,  w = S u,  gamma = (r, u),  delta = (w, u)
//   beta = gamma / gamma_prev,  alpha = gamma / (delta - beta gamma / alpha_prev)
//   p = u + beta p,  s = w + beta s,  x += alpha p,  r -= alpha s
// A wave owns one block row.  w_new = S u_new needs u_new of the NEIGHBOUR rows, which other waves produce in the same launch;
// instead of a second launch every wave recomputes its neighbours' u_new = M_b^-1 (r_b - alpha (w_b + beta s_b)) from the
// previous iteration's r, w, s (double-buffered) -- ~72 extra multiply-adds per neighbour, no second dependency.
// Every reduction runs in a fixed order (per-workgroup partials summed identically by every wave): bit-reproducible, and all
// waves take the same convergence decision.
#include "svgpu_internal.h"
#include "ba_kernels.h"

namespace {

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// fixed-order sum of the per-workgroup partials {gamma, delta, rr, -}: identical in every wave of the grid
__device__ __forceinline__ void sum_parts(const double* __restrict__ parts, int nparts, int lane, double& g, double& d, double& rr) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = lane; i < nparts; i += 64) {
        a += parts[4 * i];
        b += parts[4 * i + 1];
        c += parts[4 * i + 2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    g = a;
    d = b;
    rr = c;
}

// block-Jacobi preconditioner: inverse of every diagonal block (LL^T, L^-1, L^-T L^-1), r0 = g, u0 = M^-1 r0, x = p = s = 0
__global__ __launch_bounds__(64) void k_pcg_init1(BaDev D) {
    if (D.ctl->phase != 1) return;
    const int a = blockIdx.x * 64 + threadIdx.x;
    if (a >= D.nP) return;
    const double* B = D.Sblk + (size_t)D.diag_blk[a] * 36;
    double L[36], X[36];
    bool bad = false;
    for (int j = 0; j < 6; ++j) {
        double d = B[7 * j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0.0)) {
            bad = true;
            d = 1.0;
        }
        d = sqrt(d);
        L[7 * j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double sum = B[6 * j + i];  // upper storage of the symmetric block: B[j][i] = B[i][j]
            for (int k = 0; k < j; ++k) sum -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = sum / d;
        }
    }
    for (int c = 0; c < 6; ++c)  // X = L^-1, column c
        for (int i = 0; i < 6; ++i) {
            double sum = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) sum -= L[6 * i + k] * X[6 * k + c];
            X[6 * i + c] = (i >= c) ? sum / L[7 * i] : 0.0;
        }
    double* M = D.pcg_Minv + (size_t)a * 36;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double sum = 0.0;
            for (int k = (i > j ? i : j); k < 6; ++k) sum += X[6 * k + i] * X[6 * k + j];
            M[6 * i + j] = sum;
        }
    if (bad) D.ctl->solve_failed = 1;  // a non-positive diagonal block: the reduced system is not positive definite
    const int n = D.n;
    double* r0 = D.pcg_rws;  // set 0: r | w | s
    double r[6];
    for (int i = 0; i < 6; ++i) r[i] = D.g[6 * a + i];
    for (int i = 0; i < 6; ++i) {
        double u = 0.0;
        for (int j = 0; j < 6; ++j) u += M[6 * i + j] * r[j];
        D.pcg_own[6 * a + i] = u;
        D.pcg_own[n + 6 * a + i] = 0.0;
        r0[6 * a + i] = r[i];
        r0[2 * n + 6 * a + i] = 0.0;
        D.dp[6 * a + i] = 0.0;
    }
}

// INIT: w0 = S u0 and the first partial sums.  Otherwise: iteration `it` (see the file header).
template <bool INIT>
__global__ __launch_bounds__(256) void k_pcg_iter(BaDev D, int it) {
    BaCtl& C = *D.ctl;
    const int dn = C.pcg_done;
    if (C.phase != 1 || (dn != 0 && dn <= it)) return;
    __shared__ double s_p[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = D.n, np = D.pcg_nparts;
    const int a = blockIdx.x * 4 + wave;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    double alpha = 0.0, beta = 0.0;
    const int rs = INIT ? 0 : (it & 1), ws = INIT ? 0 : ((it + 1) & 1);  // read / write buffer sets
    if (!INIT) {
        double gam, del, rr;
        sum_parts(D.pcg_parts + (size_t)rs * np * 4, np, lane, gam, del, rr);
        const double rr0 = it == 0 ? rr : C.pcg_rr0;
        if (it == 0 && leader) C.pcg_rr0 = rr;
        if (rr <= C.pcg_tol2 * rr0) {  // converged (also covers g = 0)
            if (leader) {
                C.pcg_done = it + 1;
                C.pcg_it = it;
                C.pcg_total_it += it;
                C.pcg_solves += 1;
            }
            return;
        }
        if (it >= C.pcg_max_it) {  // not converged: accept the iterate as an inexact step if the residual fell by 1e-6, else fail the trial
            if (leader) {
                C.pcg_done = it + 1;
                C.pcg_it = it;
                C.pcg_total_it += it;
                C.pcg_solves += 1;
                C.pcg_fail = 2;
                if (!(rr <= 1e-12 * rr0)) C.solve_failed = 1;
            }
            return;
        }
        double denom = del;
        if (it > 0) {
            const double gp = D.pcg_scal[4 * rs], ap = D.pcg_scal[4 * rs + 1];
            beta = gam / gp;
            denom = del - beta * gam / ap;
        }
        alpha = gam / denom;
        if (!(denom > 0.0) || !isfinite(alpha) || !isfinite(beta)) {  // breakdown: S is not positive definite (or NaN input)
            if (leader) {
                C.pcg_done = it + 1;
                C.pcg_it = it;
                C.pcg_fail = 1;
                C.solve_failed = 1;
            }
            return;
        }
        if (leader) {
            D.pcg_scal[4 * ws] = gam;
            D.pcg_scal[4 * ws + 1] = alpha;
        }
    }
    const double* r_old = D.pcg_rws + (size_t)rs * 3 * n;
    const double* w_old = r_old + n;
    const double* s_old = r_old + 2 * n;
    double* r_new = D.pcg_rws + (size_t)ws * 3 * n;
    double* w_new = r_new + n;
    double* s_new = r_new + 2 * n;
    const int c = lane / 6, i = lane - 6 * c;  // neighbour slot 0..9 (lanes 60..63 idle), component 0..5
    const bool row_ok = a < D.nP;
    // u_new of block row b, component i, on the 6 lanes of group c (all lanes execute: the shuffles are wave-wide)
    auto u_of = [&](int b, bool valid, double& rN, double& sN) -> double {
        if (INIT) {
            rN = valid ? r_old[6 * b + i] : 0.0;
            sN = 0.0;
            return valid ? D.pcg_own[6 * b + i] : 0.0;
        }
        const double rb = valid ? r_old[6 * b + i] : 0.0, wb = valid ? w_old[6 * b + i] : 0.0, sb = valid ? s_old[6 * b + i] : 0.0;
        sN = wb + beta * sb;
        rN = rb - alpha * sN;
        const double* M = D.pcg_Minv + (size_t)(valid ? b : 0) * 36 + 6 * i;
        double u = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) u += M[j] * shfl_d(rN, 6 * c + j);
        return valid ? u : 0.0;
    };
    double r_a, s_a;
    const double u_a = u_of(row_ok ? a : 0, row_ok && c < 10, r_a, s_a);  // own row (used on lanes 0..5)
    double y = 0.0;
    const int e0 = row_ok ? D.prow_off[a] : 0, e1 = row_ok ? D.prow_off[a + 1] : 0;
    for (int eb = e0; eb < e1; eb += 10) {
        const bool valid = c < 10 && eb + c < e1;
        int2 ent;
        ent.x = 0;
        ent.y = 0;
        if (valid) ent = D.prow_ent[eb + c];
        const int k = ent.x & 0x3fffffff, tr = (ent.x >> 30) & 1;
        double rN, sN;
        const double ub = u_of(ent.y, valid, rN, sN);
        const double* Bk = D.Sblk + (size_t)k * 36;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double sij = valid ? (tr ? Bk[6 * j + i] : Bk[6 * i + j]) : 0.0;
            y += sij * shfl_d(ub, 6 * c + j);
        }
    }
    double w_a = 0.0;
#pragma unroll
    for (int cc = 0; cc < 10; ++cc) w_a += shfl_d(y, 6 * cc + (lane < 6 ? lane : 0));  // neighbour groups in slot order
    double pg = 0.0, pd = 0.0, pr = 0.0;
    if (row_ok && lane < 6) {
        const int q = 6 * a + lane;
        if (!INIT) {
            const double u_prev = D.pcg_own[q];
            const double p_new = u_prev + beta * D.pcg_own[n + q];
            D.pcg_own[n + q] = p_new;
            D.dp[q] += alpha * p_new;
            D.pcg_own[q] = u_a;
            r_new[q] = r_a;
            s_new[q] = s_a;
        }
        w_new[q] = w_a;
        pg = r_a * u_a;
        pd = w_a * u_a;
        pr = r_a * r_a;
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        pg += __shfl_xor(pg, off, 64);
        pd += __shfl_xor(pd, off, 64);
        pr += __shfl_xor(pr, off, 64);
    }
    if (lane == 0) {
        s_p[wave][0] = pg;
        s_p[wave][1] = pd;
        s_p[wave][2] = pr;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double* out = D.pcg_parts + ((size_t)ws * np + blockIdx.x) * 4;
        out[threadIdx.x] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
    }
}

}  // namespace

void sv_pcg_init(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    if (D.nP <= 0) return;
    SvProfScope ps(ctx, s, "ba_solve");
    hipLaunchKernelGGL(k_pcg_init1, dim3((D.nP + 63) / 64), dim3(64), 0, s, D);
    hipLaunchKernelGGL(k_pcg_iter<true>, dim3(D.pcg_nparts), dim3(256), 0, s, D, 0);
}

// launches iterations first_it .. first_it + count - 1; converged problems fall through every remaining launch
void sv_pcg_iterate(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int first_it, int count) {
    if (D.nP <= 0) return;
    SvProfScope ps(ctx, s, "ba_solve");
    for (int k = 0; k < count; ++k) hipLaunchKernelGGL(k_pcg_iter<false>, dim3(D.pcg_nparts), dim3(256), 0, s, D, first_it + k);
}
