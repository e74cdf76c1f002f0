// Dense LL^T of the reduced camera system for local windows that outgrow one workgroup's LDS (n = 6 x free keyframes from 138: windows
// of 25 - 200 keyframes, whose block pattern is close to full -- every keyframe of a local window shares landmarks with most others).
// local_bundle_adjuster_g2o.cc:151-164 hands this system to a dense LL^T (Eigen); north_star asks for the matrix cores here.
//
// Right-looking, tiles of DT = 48 (8 keyframes), the matrix in global memory (n = 588: 2.8 MB, L2 resident), row n = the right-hand side
// riding along (z = L^-1 g arrives with the factorisation, as in the on-chip solvers).  Two launches per tile column k:
//   k_dt_panel   every workgroup factors the diagonal tile in LDS itself (48 columns, one barrier each: cheaper than a launch boundary between
//                "factor" and "solve") and inverts the factor (a column per thread), then finishes ONE 48-row tile of the panel below it as the
//                product A W^T, W = L_kk^-1, on the matrix cores; workgroup 0 stores L_kk^-T behind the matrix for the backward substitution.  (A substitution per row and thread -- the
//                first form -- took 57 us per launch: 48 scattered loads per thread and an LDS-bound dependent chain.)
//   k_dt_update  A_IJ -= L_Ik L_Jk^T for the tile pairs k < J <= I, one workgroup per pair: both operand tiles staged in LDS, nine 16 x 16
//                sub-tiles x twelve v_mfma_f64_16x16x4_f64 over the four waves, operands read in the instruction's own layouts.
// then k_dt_backward (one workgroup): L^T x = z by tile columns from the last -- x_K = L_KK^-T z_K with the stored inverse (a 48 x 48
// matrix-vector product, no sequential triangle), the update of the rows above as a coalesced matrix-vector product, z in LDS.  Every sum
// has a fixed order: run-to-run bit-identical.  A pivot that is not positive fails the damping trial (ctl.solve_failed), as a failed LL^T
// does in the reference; the later launches of the solve then return at once.
// Measured per launch at n = 228 (rocprofv3): panel 22.6 us, update 8.1 us, backward 25 us (5 tile columns); per damping trial 290 us against
// 467 us with the envelope factorisation (257 against 370 at n = 192).
// Launches: 2 nt + 2 (nt = tile columns): 12 at n = 228, 28 at n = 588 -- against a chain of n / 6 block columns at ~6 - 10 us each in the
// general envelope kernel, which is what AUTO took for these windows before (svgpu_ba.hip).
#include <algorithm>
#include "svgpu_internal.h"
#include "ba_kernels.h"

#define DT 48
static_assert(48 * 48 % 256 == 0, "the tile loads walk 48 x 48 entries in whole rounds of 256 threads");
#define DT_LD 49  // LDS pitch of a tile (odd: a column walk touches every bank)

namespace {

typedef double dt_v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double dt_rsqrt(double p) {  // 1 / sqrt(p): hardware seed, two Newton steps (as the on-chip solvers of ba_kernels.hip)
    double y = __builtin_amdgcn_rsq(p);
    y = y * fma(-0.5 * p * y, y, 1.5);
    y = y * fma(-0.5 * p * y, y, 1.5);
    return y;
}
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for the global loads in flight (ba_skyline.hip)
__device__ __forceinline__ void dt_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double dt_readlane(double v, int l) {  // l wave-uniform
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// The diagonal tile (rows / columns c0 .. c0 + DT - 1; an identity tail beyond the matrix) factored, and its inverse with it, in REGISTERS:
// the 256 threads are a 16 x 16 grid, thread (ti, tj) owns the entries (ti + 16 p, tj + 16 q) of the 96 x 48 array [A_kk ; I] -- the
// identity rows are "panel rows" like any other, so the column steps that factor A_kk leave L_kk^-T in their place (X L^T = I).  Per
// column: its owners put the column into LDS UNSCALED, one barrier, everybody scales by 1 / sqrt(d_jj) on the fly and updates its 18
// entries.  (The first form kept the tile in LDS and walked it with short loops: ~700 cycles of dependent LDS round trips per column,
// 34 us per tile, and a substitution per inverse column on 48 threads another 19 us.)
// Out: s_L = L_kk (lower triangle), s_U = L_kk^-T (upper triangle; s_U[m][j] = (L^-1)[j][m]).  Returns false when a pivot is not positive.
__device__ __forceinline__ bool dt_factor_invert(const double* __restrict__ A, int n, int c0, double (*s_L)[DT_LD], double (*s_U)[DT_LD], double (*s_col)[2 * DT],
                                                 int* s_flag) {
    const int tid = threadIdx.x, ld = n, ti = tid >> 4, tj = tid & 15;
    double a[6][3];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = ti + 16 * (p % 3), c = tj + 16 * q;
            if (p < 3) {  // (an unconditional load from a clamped address, masked afterwards: a conditional load is a branch, and the compiler waits
                          //  for each one before the next -- eight memory round trips in a row at the head of every panel launch)
                const bool in = c <= i && c0 + i < n && c0 + c < n;
                const double v = A[in ? (size_t)(c0 + i) * ld + c0 + c : (size_t)0];
                a[p][q] = in ? v : (i == c ? 1.0 : 0.0);
            }
            else a[p][q] = i == c ? 1.0 : 0.0;
        }
    if (tid == 0) *s_flag = 0;
    bool ok = true;
    // No predicates inside a step: an entry in a column <= j is final and was stored when its column was processed -- what its register
    // holds afterwards is never read; entries above the diagonal of the top tile are garbage from the start and only ever meet each other.
#pragma unroll
    for (int qj = 0; qj < 3; ++qj)  // (unrolled: the register index of the column's owners; the 16 columns of a group stay a loop)
#pragma unroll 1
    for (int tjj = 0; tjj < 16; ++tjj) {
        const int j = 16 * qj + tjj;
        double* const col = s_col[j & 1];
        if (tj == tjj) {
#pragma unroll
            for (int p = 0; p < 6; ++p) col[ti + 16 * p] = a[p][qj];
        }
        dt_sync_lds();
        const double djj = col[j];
        if (!(djj > 0.0)) ok = false;  // (uniform; the loop runs on with a harmless factor)
        const double inv = djj > 0.0 ? dt_rsqrt(djj) : 0.0;
        double li[6], lc[3];
#pragma unroll
        for (int p = 0; p < 6; ++p) li[p] = col[ti + 16 * p] * inv;
#pragma unroll
        for (int q = 0; q < 3; ++q) lc[q] = col[tj + 16 * q] * inv;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[p][q] = fma(-li[p], lc[q], a[p][q]);
        if (tj == tjj) {  // column j of L (rows >= j) and of L^-T (rows <= j; exact zeros below)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                s_L[ti + 16 * p][j] = ti + 16 * p >= j ? li[p] : 0.0;
                s_U[ti + 16 * p][j] = li[p + 3];
            }
        }
    }
    if (!ok) {
        if (tid == 0) *s_flag = 1;
        return false;
    }
    dt_sync_lds();
    return true;
}

// workgroup b: row tile I = k + b of the panel (b = 0: the rows of the diagonal tile's own row range that lie below the matrix -- only
// the right-hand side can be there -- and the store of the factored tile)
__global__ __launch_bounds__(256) void k_dt_panel(BaDev D, int k) {
    if (D.ctl->phase != 1 || D.ctl->solve_failed) return;
    __shared__ double s_L[DT][DT_LD], s_W[DT][DT_LD], s_a[DT][DT_LD];
    __shared__ double s_col[2][2 * DT];
    __shared__ int s_flag;
    const int n = D.n, ld = n, c0 = k * DT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* const A = D.S;
    const int w = min(DT, n - c0);  // columns of this panel inside the matrix = rows of the diagonal tile
    const int I = k + (int)blockIdx.x;
    // the panel rows of this workgroup are requested first: they land while the diagonal tile is factored
    double pa[DT * DT / 256];
#pragma unroll
    for (int it = 0; it < DT * DT / 256; ++it) {
        const int t = tid + 256 * it;
        const int i = t / DT, j = t - DT * i, ri = I * DT + i;
        const bool in = ri <= n && ri >= c0 + w && j < w;
        const double v = A[in ? (size_t)ri * ld + c0 + j : (size_t)0];
        pa[it] = in ? v : 0.0;
    }
    if (!dt_factor_invert(A, n, c0, s_L, s_W, s_col, &s_flag)) {
        if (blockIdx.x == 0 && tid == 0) D.ctl->solve_failed = 1;
        return;
    }
#pragma unroll
    for (int it = 0; it < DT * DT / 256; ++it) {  // (the panel rows were in flight all through the factorisation)
        const int t = tid + 256 * it;
        s_a[t / DT][t % DT] = pa[it];
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        // L_kk^-T goes behind the matrix: the backward substitution multiplies by it.  The factored tile itself is NOT stored: nobody reads it
        // again, and the other workgroups of this launch are still loading the unfactored tile from that very place.
#pragma unroll
        for (int it = 0; it < DT * DT / 256; ++it) {
            const int t = tid + 256 * it;
            A[(size_t)(n + 1) * ld + (size_t)k * (DT * DT) + t] = s_W[t / DT][t % DT];
        }
    }
    // the panel tile X = A L^-T on the matrix cores (s_W = L^-T: B[m][j] read as it lies)
    for (int st = wave; st < 9; st += 4) {
        const int si = st / 3, sj = st - 3 * si;
        dt_v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < DT / 4; ++kk) {
            const double a = s_a[16 * si + (lane & 15)][4 * kk + (lane >> 4)];
            const double b = s_W[4 * kk + (lane >> 4)][16 * sj + (lane & 15)];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
        const int col = 16 * sj + (lane & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rowg = I * DT + 16 * si + (lane >> 4) + 4 * reg;
            if (rowg <= n && rowg >= c0 + w && col < w) A[(size_t)rowg * ld + c0 + col] = c[reg];
        }
    }
}

// tile pair (I, J), k < J <= I (row tiles run up to the one that holds row n)
__global__ __launch_bounds__(256) void k_dt_update(BaDev D, int k, int nrt) {
    if (D.ctl->phase != 1 || D.ctl->solve_failed) return;
    __shared__ double s_a[DT][DT_LD], s_b[DT][DT_LD];
    const int n = D.n, ld = n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* const A = D.S;
    // pair index -> (I, J): p = (I - k - 1) (I - k) / 2 + (J - k - 1)
    int p = blockIdx.x, di = 0;
    while ((di + 1) * (di + 2) / 2 <= p) ++di;
    const int I = k + 1 + di, J = k + 1 + (p - di * (di + 1) / 2);
    if (I >= nrt) return;
    const int c0 = k * DT, w = min(DT, n - c0);
#pragma unroll
    for (int it = 0; it < DT * DT / 256; ++it) {
        const int t = tid + 256 * it;
        const int i = t / DT, j = t - DT * i;
        const int ri = I * DT + i, rj = J * DT + i;
        const bool ina = ri <= n && j < w, inb = rj < n && j < w;  // (b: columns of the result = rows of L_Jk, inside the matrix only)
        const double va = A[ina ? (size_t)ri * ld + c0 + j : (size_t)0], vb = A[inb ? (size_t)rj * ld + c0 + j : (size_t)0];
        s_a[i][j] = ina ? va : 0.0;
        s_b[i][j] = inb ? vb : 0.0;
    }
    __syncthreads();
    // nine 16 x 16 sub-tiles, wave w takes w, w + 4, w + 8
    for (int st = wave; st < 9; st += 4) {
        const int si = st / 3, sj = st - 3 * si;
        if (I == J && sj > si) continue;  // (above the diagonal of a diagonal tile: never read)
        dt_v4f64 c;
        const int col = J * DT + 16 * sj + (lane & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rowg = I * DT + 16 * si + (lane >> 4) + 4 * reg;
            const bool in = rowg <= n && col < n;
            const double v = A[in ? (size_t)rowg * ld + col : (size_t)0];
            c[reg] = in ? v : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < DT / 4; ++kk) {
            const double a = s_a[16 * si + (lane & 15)][4 * kk + (lane >> 4)];
            const double b = s_b[16 * sj + (lane & 15)][4 * kk + (lane >> 4)];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rowg = I * DT + 16 * si + (lane >> 4) + 4 * reg;
            if (rowg <= n && col < n) A[(size_t)rowg * ld + col] = c[reg];
        }
    }
}

// L^T x = z (z = row n), tile columns from the last; dp = x.  One workgroup of 256; z lives in LDS.  Per tile column: x_K = L_KK^-T z_K as a
// 48 x 48 matrix-vector product with the inverse the panel kernel left behind the matrix (no sequential triangle), then
// z_c -= sum_r L[c0 + r][c] x_r for the columns before it (consecutive threads read consecutive columns of a row).  The loads of that
// product and of the NEXT inverse tile are issued before x_K is formed, so a tile column costs one memory round trip.
#define DT_MAX_N 1536
__global__ __launch_bounds__(256) void k_dt_backward(BaDev D, int nct) {
    if (D.ctl->phase != 1) return;
    const int n = D.n, ld = n, tid = threadIdx.x;
    if (D.ctl->solve_failed) {
        for (int i = tid; i < n; i += 256) D.dp[i] = 0.0;
        return;
    }
    __shared__ double s_U[DT][DT_LD];
    __shared__ double s_x[DT];
    __shared__ double s_z[DT_MAX_N];
    const double* const A = D.S;
    const double* const Uinv = A + (size_t)(n + 1) * ld;
    for (int i = tid; i < n; i += 256) s_z[i] = A[(size_t)n * ld + i];
    double du[DT * DT / 256];
    auto load_u = [&](int K) {
#pragma unroll
        for (int it = 0; it < DT * DT / 256; ++it) du[it] = Uinv[(size_t)K * (DT * DT) + tid + 256 * it];
    };
    load_u(nct - 1);
    for (int K = nct - 1; K >= 0; --K) {
        const int c0 = K * DT, w = min(DT, n - c0);
        dt_sync_lds();  // (the previous column's readers of s_U / s_x are done; its updates of s_z are visible)
#pragma unroll
        for (int it = 0; it < DT * DT / 256; ++it) {
            const int t = tid + 256 * it;
            s_U[t / DT][t % DT] = du[it];
        }
        if (K > 0) load_u(K - 1);
        // first pass of the product (columns tid < c0): its 48 loads go out now
        double l[DT];
        const bool first = tid < c0;
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            const bool in = first && r < w;
            const double v = A[in ? (size_t)(c0 + r) * ld + tid : (size_t)0];
            l[r] = in ? v : 0.0;
        }
        dt_sync_lds();
        if (tid < DT) {  // x_j = sum_{m >= j} U[j][m] z_m  (U = L_KK^-T, upper triangular; the identity tail beyond the matrix contributes z = 0)
            double v = 0.0;
#pragma unroll 8
            for (int m = 0; m < DT; ++m) v = fma(s_U[tid][m], m < w ? s_z[c0 + m] : 0.0, v);
            s_x[tid] = v;
            if (tid < w) D.dp[c0 + tid] = v;
        }
        dt_sync_lds();
        if (first) {
            double v = s_z[tid];
#pragma unroll
            for (int r = 0; r < DT; ++r) v = fma(-l[r], s_x[r], v);
            s_z[tid] = v;
        }
        for (int c = tid + 256; c < c0; c += 256) {  // further passes (n > 304)
            double l2[DT];
#pragma unroll
            for (int r = 0; r < DT; ++r) {
                const double v = A[r < w ? (size_t)(c0 + r) * ld + c : (size_t)0];
                l2[r] = r < w ? v : 0.0;
            }
            double v = s_z[c];
#pragma unroll
            for (int r = 0; r < DT; ++r) v = fma(-l2[r], s_x[r], v);
            s_z[c] = v;
        }
    }
}

}  // namespace

// the reduced system is in D.S ((n + 1) x n row-major, lower triangle + the right-hand side as row n: k_ba_expand_dense)
void sv_ba_dense_tiled(hipStream_t s, const BaDev& D) {
    const int n = D.n;
    if (n <= 0) return;
    const int nct = (n + DT - 1) / DT, nrt = (n + 1 + DT - 1) / DT;  // tile columns; tile rows (row n included)
    for (int k = 0; k < nct; ++k) {
        hipLaunchKernelGGL(k_dt_panel, dim3(nrt - k), dim3(256), 0, s, D, k);  // row tiles k .. nrt - 1 (tile k: the rows below the matrix inside it, if any)
        const int m = nrt - (k + 1);  // tile rows below the diagonal tile
        if (m > 0) hipLaunchKernelGGL(k_dt_update, dim3(m * (m + 1) / 2), dim3(256), 0, s, D, k, nrt);
    }
    hipLaunchKernelGGL(k_dt_backward, dim3(1), dim3(256), 0, s, D, nct);
}
