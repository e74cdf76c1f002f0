// Direct solve of large reduced camera systems: block envelope ("skyline") Cholesky on the device.
//
// The reference factors the reduced system of global BA with a sparse Cholesky (g2o LinearSolverCSparse / Eigen,
// optimize/global_bundle_adjuster.cc:26-40); the oracle restates it as an envelope Cholesky.  This is the same method on the GPU,
// for systems beyond the on-chip dense solver: keyframe graphs are banded up to a few loop-closure rows once ordered (reverse
// Cuthill-McKee on the host), so the lower envelope of the 6x6 block matrix holds the whole fill -- config 5 (500 keyframes, a closed
// ring): a few MB, L2 resident.  One workgroup walks the block columns (right-looking LL^T):
//   pivot      L_jj = chol(D_jj), its inverse kept                      (one wave, lanes = rows)
//   column     L_ij = S_ij L_jj^-T for the rows i of column j            (one thread per entry, staged in LDS)
//   update     S_ik -= L_ij L_kj^T for all pairs of rows i >= k of j     (one thread per block pair, operands from LDS)
// then the forward / backward substitutions over the same column lists.  Every sum has a fixed order: bit-reproducible.
// A pivot that is not positive marks the damping trial as a solver failure, as g2o's solver returning false does.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "svgpu_internal.h"

#include "ba_kernels.h"

namespace {
#define SKY_THREADS 1024
#define SKY_MAXM 200  // rows below the diagonal of one column whose blocks are staged in LDS (200 x 288 B = 56 KB)

struct SkyDev {
    int nP = 0, NB = 0;
    const int* pos = nullptr;       // slot -> position in the elimination order
    const int* first = nullptr;     // position i -> first block column of row i's envelope
    const int* rowoff = nullptr;    // position i -> index of block (i, first[i]); nP + 1 entries
    const int* coloff = nullptr;    // position j -> start of its row list; nP + 1 entries
    const int* colrows = nullptr;   // rows i > j with first[i] <= j, ascending
    const int* colbase = nullptr;   // beside colrows: rowoff[i] - first[i] (block (i, k) of the envelope = base + k)
    const int* diag = nullptr;      // position j -> index of block (j, j)
    int max_m = 0;                  // widest column (rows below the diagonal)
    int ncr = 0;                    // entries of colrows / colbase
    int band = 0;                   // 1 = first[] is non-decreasing and max_m <= SKY_BAND_W: k_sky_band applies
    const int2* blkmap = nullptr;   // kept block k of the reduced system -> (envelope block index, 1 = store transposed)
    double* val = nullptr;          // envelope blocks, 36 doubles each, row-major
    double* dinv = nullptr;         // nP x 36: inverses of the diagonal factors (lower triangular)
    double* y = nullptr;            // n: right-hand side / solution in elimination order
    size_t nblocks = 0;
    int y_from = 0x7fffffff;        // positions >= y_from get no right-hand side from the assembly (the separator rows of the second half, below)
};

// Two-sided ("twisted") elimination of a banded system: the elimination order is cut into T | S | B with |S| = W block rows (no block
// couples T and B), one workgroup eliminates T top-down on the plan of [T, S], a second one B bottom-up on the plan of the REVERSED
// [B, S] -- the same kernel on a second set of arrays --, the Schur complements they leave on S are added, S is factored, and the
// substitutions run inwards / outwards the same way.  Two flags in global memory carry the hand-overs (value = the solve's epoch,
// negative = "my half failed"): f[0] second -> first: the S x S window and its share of the right-hand side of S; f[2] first -> second: x_S.  Same arithmetic per block as the one-sided kernel, half the dependent chain.
struct SkyTwist {
    int on = 0, m = 0, W = 0, nB = 0;  // first S position in the first plan; separator rows; eliminated columns of the second plan
    double* xch = nullptr;             // [W (W + 1) / 2 x 36: S x S blocks of the second plan's window][6 W: its y_S][6 W: x_S of the first plan]
    int* flags = nullptr;
};

__global__ __launch_bounds__(256) void k_sky_assemble(BaDev D, SkyDev K0, SkyDev K1) {  // blockIdx.y = plan (a block / slot a plan does not hold maps to -1)
    if (D.ctl->phase != 1) return;
    const SkyDev& K = blockIdx.y ? K1 : K0;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < (size_t)K.NB * 36) {
        const int k = (int)(t / 36), e = (int)(t - (size_t)k * 36), i = e / 6, j = e - 6 * i;
        const int2 m = K.blkmap[k];
        if (m.x >= 0) K.val[(size_t)m.x * 36 + (m.y ? j * 6 + i : e)] = D.Sblk[t];
    }
    if (t < (size_t)D.n) {
        const int a = (int)(t / 6), c = (int)(t - (size_t)a * 6);
        const int p = K.pos[a];
        if (p >= 0 && p < K.y_from) K.y[p * 6 + c] = D.g[t];
    }
}

// x in [0, m (m + 1) / 2) -> (p, q) with q <= p < m, row-major over the lower triangle
__device__ __forceinline__ void tri_index(int x, int& p, int& q) {
    p = (int)((sqrtf(8.0f * (float)x + 1.0f) - 1.0f) * 0.5f);
    while ((p + 1) * (p + 2) / 2 <= x) ++p;
    while (p * (p + 1) / 2 > x) --p;
    q = x - p * (p + 1) / 2;
}
__device__ __forceinline__ double lane_bcast(double v, int src) {  // v of lane `src` (a constant) as a wave-uniform value
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Ordering of LDS accesses between the lanes of ONE wave: the LDS unit executes a wave's DS instructions in issue order, so only the
// compiler has to be kept from moving them; unlike a fence this leaves the prefetched global loads in flight.
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// The 6 x 6 pivot of a column, on the first wave: L = chol(block at `src`, lower triangle), written to `dstL` (upper triangle zeroed), and
// L^-1 to `dstInv` (global, for the substitutions) and to s_Li (for the column).  Rows on lanes, entries exchanged with readlane, one
// rsqrt per pivot and no division; the inverse falls out of the same registers, column c on lane c.
__device__ __forceinline__ void sky_pivot(const double* src, double* dstL, double* dstInv, double* s_Li, int* s_fail, int tid) {
    const int r = min(tid, 5);
    double a[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = src[r * 6 + c];  // lane r < 6: row r (lanes >= 6 mirror row 5 and write nothing)
    double Lm[6][6], rd[6];  // the finished factor and its reciprocal diagonal, wave-uniform
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < c) v -= a[k] * Lm[c][k];  // a[k] = L[r][k] by now
        const double dcc = lane_bcast(v, c);
        bad = bad || !(dcc > 0.0);
        const double rs = rsqrt(dcc);
        rd[c] = rs;
        a[c] = v * rs;  // L[r][c]; on lane c: dcc / sqrt(dcc)
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) Lm[rr][c] = rr >= c ? lane_bcast(a[c], rr) : 0.0;
    }
    if (bad && tid == 0) *s_fail = 1;
    // column c of L^-1 on lane c: x_c = 1 / L_cc, x_r = -(sum_{k = c}^{r - 1} L_rk x_k) / L_rr
    const int cc = min(tid, 5);
    double x[6];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        double v = rr == cc ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < rr) v -= (k >= cc ? Lm[rr][k] * x[k] : 0.0);
        x[rr] = rr >= cc ? v * rd[rr] : 0.0;
    }
    if (tid < 6) {
#pragma unroll
        for (int c = 0; c < 6; ++c) dstL[tid * 6 + c] = c <= tid ? a[c] : 0.0;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            s_Li[rr * 6 + tid] = x[rr];
            dstInv[rr * 6 + tid] = x[rr];
        }
    }
}

// L z = g (column oriented), then L^T x = z, on the first wave alone: the right-hand side lives in LDS, the factor rows of the NEXT
// column are in flight while a column is processed, no workgroup barrier.
// Narrow columns (at most 16 rows below the diagonal -- every banded envelope): the factor rows of the next FOUR columns are in flight
// while a column is processed (a global load takes ~1 us seen from one wave, a column step ~0.2 us), and the backward sums are combined
// by a shuffle tree inside groups of eight lanes instead of sixty lane broadcasts.
// (columns jb .. je - 1 of the forward pass, columns jhi .. jlo of the backward pass: the two-sided kernel runs them in pieces)
__device__ __forceinline__ void sky_forward_narrow(const SkyDev& K, double* s_y, const int* s_coloff, const int* s_rows, const int* s_base, int lane, int jb, int je) {
    constexpr int PD = 4;
    {   // forward: lane rr < 6 holds row rr of Li_j; item k of a lane = (row (lane + 64 k) / 6 of the column, component (lane + 64 k) % 6)
        auto load = [&](int j, double (&li)[6], double (&bl)[2][6], int (&dst)[2]) {
            if (j >= je) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int rr = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) li[c] = K.dinv[(size_t)j * 36 + rr * 6 + c];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = lane + 64 * k;
                dst[k] = -1;
                if (t < m * 6) {
                    const int r = t / 6, a = t - 6 * r;
                    const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36 + a * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) bl[k][c] = Bl[c];
                    dst[k] = s_rows[c0 + r] * 6 + a;
                }
            }
        };
        double li[PD][6], bl[PD][2][6];
        int dst[PD][2];
#pragma unroll
        for (int d = 0; d < PD; ++d) load(jb + d, li[d], bl[d], dst[d]);
        for (int j0 = jb; j0 < je; j0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int j = j0 + d;
                if (j >= je) break;
                double v = 0.0;  // z[rr] on lane rr < 6
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    if (c <= min(lane, 5)) v += li[d][c] * s_y[j * 6 + c];
                double z[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) z[c] = lane_bcast(v, c);
                wave_lds_order();  // every lane has read y_j
                if (lane < 6) s_y[j * 6 + lane] = v;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (dst[d][k] >= 0) {
                        double u = bl[d][k][0] * z[0];
#pragma unroll
                        for (int c = 1; c < 6; ++c) u += bl[d][k][c] * z[c];
                        s_y[dst[d][k]] -= u;
                    }
                wave_lds_order();
                load(j + PD, li[d], bl[d], dst[d]);
            }
        }
    }
}
__device__ __forceinline__ void sky_backward_narrow(const SkyDev& K, double* s_y, const int* s_coloff, const int* s_rows, const int* s_base, int lane, int jhi, int jlo) {
    constexpr int PD = 4;
    {   // backward: w = z_j - sum_{i in rows(j)} L_ij^T x_i, lane = component * 8 + part (48 lanes), a lane's rows: part and part + 8
        const int a = min(lane >> 3, 5), part = lane & 7;
        auto load = [&](int j, double (&lc)[6], double (&bc)[2][6], int (&src)[2]) {
            if (j < jlo) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int ac = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) lc[c] = K.dinv[(size_t)j * 36 + c * 6 + ac];  // column `lane` of Li_j
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = part + 8 * k;
                src[k] = -1;
                if (lane < 48 && r < m) {
                    const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36;
#pragma unroll
                    for (int c = 0; c < 6; ++c) bc[k][c] = Bl[c * 6 + a];
                    src[k] = s_rows[c0 + r] * 6;
                }
            }
        };
        double lc[PD][6], bc[PD][2][6];
        int src[PD][2];
#pragma unroll
        for (int d = 0; d < PD; ++d) load(jhi - d, lc[d], bc[d], src[d]);
        for (int j0 = jhi; j0 >= jlo; j0 -= PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int j = j0 - d;
                if (j < jlo) break;
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (src[d][k] >= 0)
#pragma unroll
                        for (int c = 0; c < 6; ++c) v += bc[d][k][c] * s_y[src[d][k] + c];
                v += __shfl_xor(v, 1, 64);  // fixed tree inside the component's eight lanes
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                double w[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) w[c] = s_y[j * 6 + c] - lane_bcast(v, 8 * c);
                wave_lds_order();
                if (lane < 6) {  // x_j = Li_j^T w: component a = sum_{c >= a} Li[c][a] w[c]
                    double x = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        if (c >= lane) x += lc[d][c] * w[c];
                    s_y[j * 6 + lane] = x;
                }
                wave_lds_order();
                load(j - PD, lc[d], bc[d], src[d]);
            }
        }
    }
}
__device__ __forceinline__ void sky_substitute_narrow(const SkyDev& K, double* s_y, const int* s_coloff, const int* s_rows, const int* s_base, int lane) {
    sky_forward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, 0, K.nP);
    sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, K.nP - 1, 0);
}

__device__ __forceinline__ void sky_substitute(const SkyDev& K, double* s_y, const int* s_coloff, const int* s_rows, const int* s_base, int tid) {
    const int nP = K.nP;
        const int lane = tid;
        const bool pre = K.max_m * 6 <= 128;  // a lane owns at most two (row, component) items of a column: their factor rows are prefetched
        // forward.  Lane rr < 6 holds row rr of Li_j, item k of a lane = (row (lane + 64 k) / 6 of the column, component (lane + 64 k) % 6)
        auto fwd_load = [&](int j, double (&li)[6], double (&bl)[2][6], int (&dst)[2]) {
            if (j >= nP) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int rr = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) li[c] = K.dinv[(size_t)j * 36 + rr * 6 + c];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = lane + 64 * k;
                dst[k] = -1;
                if (t < m * 6) {
                    const int r = t / 6, a = t - 6 * r;
                    const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36 + a * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) bl[k][c] = Bl[c];
                    dst[k] = s_rows[c0 + r] * 6 + a;
                }
            }
        };
        double li[6], bl[2][6], li_n[6], bl_n[2][6];
        int dst[2], dst_n[2];
        if (pre) fwd_load(0, li, bl, dst);
        for (int j = 0; j < nP; ++j) {
            if (pre) fwd_load(j + 1, li_n, bl_n, dst_n);
            else fwd_load(j, li, bl, dst);
            double v = 0.0;  // z[rr] on lane rr < 6
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c <= min(lane, 5)) v += li[c] * s_y[j * 6 + c];
            double z[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) z[c] = lane_bcast(v, c);
            wave_lds_order();  // every lane has read y_j
            if (lane < 6) s_y[j * 6 + lane] = v;
            if (pre) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (dst[k] >= 0) {
                        double u = bl[k][0] * z[0];
#pragma unroll
                        for (int c = 1; c < 6; ++c) u += bl[k][c] * z[c];
                        s_y[dst[k]] -= u;
                    }
            }
            else {
                const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
                for (int t = lane; t < m * 6; t += 64) {
                    const int r = t / 6, a = t - 6 * r;
                    const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36 + a * 6;
                    double u = Bl[0] * z[0];
#pragma unroll
                    for (int c = 1; c < 6; ++c) u += Bl[c] * z[c];
                    s_y[s_rows[c0 + r] * 6 + a] -= u;
                }
            }
            wave_lds_order();
            if (pre) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    li[c] = li_n[c];
                    bl[0][c] = bl_n[0][c];
                    bl[1][c] = bl_n[1][c];
                }
                dst[0] = dst_n[0];
                dst[1] = dst_n[1];
            }
        }
        // backward.  w = z_j - sum_{i in rows(j)} L_ij^T x_i: 6 components x 10 parts on 60 lanes (lane = part * 6 + component), combined in
        // a fixed order; a lane's up to two rows (m <= 20) have their factor column prefetched.
        const int a = lane % 6, part = lane / 6;
        auto bwd_load = [&](int j, double (&lc)[6], double (&bc)[2][6], int (&src)[2]) {
            if (j < 0) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int ac = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) lc[c] = K.dinv[(size_t)j * 36 + c * 6 + ac];  // column `lane` of Li_j
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = part + 10 * k;
                src[k] = -1;
                if (part < 10 && r < m) {
                    const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36;
#pragma unroll
                    for (int c = 0; c < 6; ++c) bc[k][c] = Bl[c * 6 + a];
                    src[k] = s_rows[c0 + r] * 6;
                }
            }
        };
        const bool preb = K.max_m <= 20;
        double lc[6], bc[2][6], lc_n[6], bc_n[2][6];
        int src[2], src_n[2];
        if (preb) bwd_load(nP - 1, lc, bc, src);
        for (int j = nP - 1; j >= 0; --j) {
            double v = 0.0;
            if (preb) {
                bwd_load(j - 1, lc_n, bc_n, src_n);
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (src[k] >= 0)
#pragma unroll
                        for (int c = 0; c < 6; ++c) v += bc[k][c] * s_y[src[k] + c];
            }
            else {
                const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
                const int ac = min(lane, 5);
#pragma unroll
                for (int c = 0; c < 6; ++c) lc[c] = K.dinv[(size_t)j * 36 + c * 6 + ac];
                if (part < 10)
                    for (int r = part; r < m; r += 10) {
                        const double* Bl = K.val + (size_t)(s_base[c0 + r] + j) * 36;
                        const double* xi = s_y + s_rows[c0 + r] * 6;
#pragma unroll
                        for (int c = 0; c < 6; ++c) v += Bl[c * 6 + a] * xi[c];
                    }
            }
            double w[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double tot = 0.0;
#pragma unroll
                for (int pp = 0; pp < 10; ++pp) tot += lane_bcast(v, pp * 6 + c);
                w[c] = s_y[j * 6 + c] - tot;
            }
            wave_lds_order();
            if (lane < 6) {  // x_j = Li_j^T w: component a = sum_{c >= a} Li[c][a] w[c]
                double x = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    if (c >= lane) x += lc[c] * w[c];
                s_y[j * 6 + lane] = x;
            }
            wave_lds_order();
            if (preb) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    lc[c] = lc_n[c];
                    bc[0][c] = bc_n[0][c];
                    bc[1][c] = bc_n[1][c];
                }
                src[0] = src_n[0];
                src[1] = src_n[1];
            }
        }
    
}

// One workgroup (256 threads while no column has more than 21 rows, else 1024).  What a column costs is a chain of dependent memory
// round trips, so the chain is kept short: the index arrays live in LDS, every block of a column is addressed through one index
// (base[i] = rowoff[i] - first[i], block (i, k) = base[i] + k), the 6 x 6 pivot lives in the registers of one wave (rows on lanes,
// entries exchanged with readlane, one rsqrt per pivot and no division; the inverse factor falls out of the same registers, column c on
// lane c), and the two substitutions run on the first wave alone with the right-hand side in LDS and the factor rows of the NEXT column
// already in flight while a column is processed (no workgroup barrier).
// Measured at config 5 (499 block rows, 5 527 envelope blocks, widest column 11 rows): 3.2 ms per solve -- pivot + column 1.3, update 0.9,
// substitutions 1.0 -- i.e. ~6 us per column, every phase a global-memory round trip through L2 (~1 us each seen from one wave); the
// block-Jacobi PCG it replaces as the default took 3.4 ms per solve (~480 iterations) and up to 4 000 iterations when lambda is small.
// Next: the trailing window of a banded envelope kept in LDS.
__global__ __launch_bounds__(SKY_THREADS) void k_sky_factor_solve(BaDev D, SkyDev K) {
    if (D.ctl->phase != 1) return;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];  // [max_m x 36: the scaled column][n: right-hand side][index arrays]
    __shared__ double s_Li[36];
    __shared__ int s_fail;
    const int tid = threadIdx.x, nt = blockDim.x, nP = K.nP;
    double* const s_col = s_dyn;
    double* const s_y = s_dyn + (size_t)K.max_m * 36;
    int* const s_coloff = reinterpret_cast<int*>(s_y + D.n);
    int* const s_diag = s_coloff + (nP + 1);
    int* const s_rows = s_diag + nP;
    int* const s_base = s_rows + K.ncr;
    if (tid == 0) s_fail = 0;
    for (int t = tid; t < D.n; t += nt) s_y[t] = K.y[t];
    for (int t = tid; t <= nP; t += nt) s_coloff[t] = K.coloff[t];
    for (int t = tid; t < nP; t += nt) s_diag[t] = K.diag[t];
    for (int t = tid; t < K.ncr; t += nt) {
        s_rows[t] = K.colrows[t];
        s_base[t] = K.colbase[t];
    }
    __syncthreads();
    // ---------------------------------------------------------------- factorisation
    for (int j = 0; j < nP; ++j) {
        const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
        const int* rows = s_rows + c0;
        const int* base = s_base + c0;
        if (tid < 64) {
            double* Dj = K.val + (size_t)s_diag[j] * 36;
            sky_pivot(Dj, Dj, K.dinv + (size_t)j * 36, s_Li, &s_fail, tid);
        }
        __syncthreads();
        if (s_fail) break;
        // column: L_ij = S_ij L_jj^-T, entry (a, b) = sum_{c <= b} S_ij[a][c] Li[b][c]; staged in LDS (the reads of the old block finish first)
        for (int t = tid; t < m * 36; t += nt) {
            const int r = t / 36, e = t - r * 36, a = e / 6, b = e - 6 * a;
            const double* Bl = K.val + (size_t)(base[r] + j) * 36 + a * 6;
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c <= b) v += Bl[c] * s_Li[b * 6 + c];
            s_col[t] = v;
        }
        __syncthreads();
        for (int t = tid; t < m * 36; t += nt) {
            const int r = t / 36, e = t - r * 36;
            K.val[(size_t)(base[r] + j) * 36 + e] = s_col[t];
        }
        // update: one thread per pair of rows (p, q), q <= p: S_{ip, iq} -= L_p L_q^T
        const int npair = m * (m + 1) / 2;
        for (int x = tid; x < npair; x += nt) {
            int p, q;
            tri_index(x, p, q);
            double Lq[36];
#pragma unroll
            for (int e = 0; e < 36; ++e) Lq[e] = s_col[q * 36 + e];
            double* Dst = K.val + (size_t)(base[p] + rows[q]) * 36;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double La[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) La[c] = s_col[p * 36 + a * 6 + c];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    double v = La[0] * Lq[b * 6];
#pragma unroll
                    for (int c = 1; c < 6; ++c) v += La[c] * Lq[b * 6 + c];
                    Dst[a * 6 + b] -= v;
                }
            }
        }
        __syncthreads();
    }
    if (s_fail) {
        if (tid == 0) D.ctl->solve_failed = 1;
        for (int t = tid; t < D.n; t += nt) D.dp[t] = 0.0;
        return;
    }
    // ---------------------------------------------------------------- L z = g (column oriented), then L^T x = z: the first wave alone
    if (tid < 64) sky_substitute(K, s_y, s_coloff, s_rows, s_base, tid);
    __syncthreads();
    for (int t = tid; t < D.n; t += nt) {
        const int a = t / 6, c = t - 6 * a;
        D.dp[t] = s_y[K.pos[a] * 6 + c];
    }
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for the global loads in flight (the row that
// enters the window) nor for the global stores of the finished factor to be acknowledged.
__device__ __forceinline__ void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Banded envelopes (first[] non-decreasing, so the rows of column j are j + 1 .. j + m_j; bandwidth W <= SKY_BAND_W): the trailing
// (W + 1) x (W + 1) block window lives in LDS (slot (i mod (W + 1), k mod (W + 1))), so the pivot, the column and the update of a
// column never wait for global memory: the only global reads are the blocks of the row that enters the window (unmodified so far,
// requested before the pivot and stored behind the update), the only global writes the finished factor for the substitutions.
// Config 5 (499 columns, W = 11): 2.05 ms per solve against 3.2 ms for the general kernel below and 3.4 ms for the PCG -- by phase
// (runs with phases switched off): skeleton (3 barriers per column, window fill, memset + assembly) 0.43, pivots 0.35, columns 0.40,
// updates 0.53, substitutions 0.35 -- with 256 threads and before the pivot of column j + 1 moved into the shadow of column j's update.
#define SKY_BAND_W 16
#ifndef SKY_BAND_THREADS
#define SKY_BAND_THREADS 512  // 256: 1.81 ms per config-5 solve, 512: 1.59 (the update of a column fits one round), 1024: 2.41 (128 VGPRs: the substitutions spill)
#endif
// hand-over flags of the two-sided elimination (agent scope: the two workgroups sit on different compute units)
__device__ __forceinline__ void sky_post(int* flag, int value) {
    __threadfence();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int sky_wait(int* flag, int epoch) {  // +-epoch
    int v;
    while ((v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != epoch && v != -epoch) __builtin_amdgcn_s_sleep(4);
    return v;
}
__device__ __forceinline__ double sky_peek(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // past this unit's L1

__global__ __launch_bounds__(SKY_BAND_THREADS) void k_sky_band(BaDev D, SkyDev K0, SkyDev K1, SkyTwist T, int epoch) {
    if (D.ctl->phase != 1) return;
    const int who = blockIdx.x;  // 0: the whole system, or [T, S] of a two-sided elimination; 1: the reversed [B, S]
    const SkyDev& K = who ? K1 : K0;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];  // [window (W+1)^2 x 36][column W x 36][6 nP: right-hand side][index arrays]
    __shared__ double s_Li[2][36];  // inverse diagonal factors of columns j and j + 1 (the look-ahead pivot writes one while the other is in use)
    __shared__ int s_fail, s_other;
    __shared__ unsigned short s_pq[SKY_BAND_W * (SKY_BAND_W + 1) / 2];  // pair index -> p | q << 8 (q <= p), row-major over the lower triangle
    const int tid = threadIdx.x, nt = blockDim.x, nP = K.nP, W = K.max_m, Wn = W + 1, ny = 6 * nP;
    if (tid < SKY_BAND_W * (SKY_BAND_W + 1) / 2) {
        int p, q;
        tri_index(tid, p, q);
        s_pq[tid] = (unsigned short)(p | (q << 8));
    }
    double* const s_win = s_dyn;
    double* const s_col = s_win + (size_t)Wn * Wn * 36;
    double* const s_y = s_col + (size_t)W * 36;
    int* const s_coloff = reinterpret_cast<int*>(s_y + ny);
    int* const s_first = s_coloff + (nP + 1);
    int* const s_rbase = s_first + nP;   // rowoff[i] - first[i]: block (i, k) of the envelope = s_rbase[i] + k
    int* const s_rows = s_rbase + nP;    // for the substitution routine
    int* const s_base = s_rows + K.ncr;
    if (tid == 0) s_fail = 0, s_other = epoch;
    for (int t = tid; t < ny; t += nt) s_y[t] = K.y[t];
    for (int t = tid; t <= nP; t += nt) s_coloff[t] = K.coloff[t];
    for (int t = tid; t < nP; t += nt) {
        s_first[t] = K.first[t];
        s_rbase[t] = K.rowoff[t] - K.first[t];
    }
    for (int t = tid; t < K.ncr; t += nt) {
        s_rows[t] = K.colrows[t];
        s_base[t] = K.colbase[t];
    }
    __syncthreads();
    auto slot = [&](int i, int k) { return s_win + (size_t)((i % Wn) * Wn + (k % Wn)) * 36; };
    // rows 0 .. W of the assembled system
    for (int i = 0; i < min(Wn, nP); ++i) {
        const int f = s_first[i];
        for (int t = tid; t < (i - f + 1) * 36; t += nt) {
            const int k = f + t / 36, e = t % 36;
            slot(i, k)[e] = K.val[(size_t)(s_rbase[i] + k) * 36 + e];
        }
    }
    __syncthreads();
    // Column j: [scale] barrier [update of the other waves | first wave: the six rows of block (j + 1, j + 1), then the PIVOT of column
    // j + 1, which needs nothing else of this update] barrier.  Two barriers per column, the pivots off the critical path.
    auto update_item = [&](int j, int t) {  // S_{ip, iq} -= L_p L_q^T inside the window: item = (pair, row of the block)
        const int x = t / 6, a = t - 6 * x;
        const int pq = s_pq[x], p = pq & 255, q = pq >> 8;
        double La[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) La[c] = s_col[p * 36 + a * 6 + c];
        double* Dst = slot(j + 1 + p, j + 1 + q) + a * 6;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double v = La[0] * s_col[q * 36 + b * 6];
#pragma unroll
            for (int c = 1; c < 6; ++c) v += La[c] * s_col[q * 36 + b * 6 + c];
            Dst[b] -= v;
        }
    };
    // columns jb .. je - 1 (the pivot of column jb is in s_Li when it starts; the pivot of column je is NOT taken: it may wait for the other half)
    auto factor = [&](int jb, int je) {
        for (int j = jb; j < je; ++j) {
            if (s_fail) break;
            const int m = s_coloff[j + 1] - s_coloff[j];
            // the row that enters the window behind this column: requested now, stored in LDS behind the update
            const int inew = j + Wn;
            double pre[3];  // (W + 1) x 36 <= 612 entries over the workgroup's threads
            int npre = 0, fnew = 0;
            if (inew < nP) {
                fnew = max(s_first[inew], j + 1);
                npre = (inew - fnew + 1) * 36;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int t = tid + q * SKY_BAND_THREADS;
                    if (t < npre) pre[q] = K.val[(size_t)(s_rbase[inew] + fnew + t / 36) * 36 + t % 36];
                }
            }
            for (int t = tid; t < m * 36; t += nt) {  // L_ij = S_ij L_jj^-T
                const int r = t / 36, e = t - r * 36, a = e / 6, b = e - 6 * a;
                const double* Bl = slot(j + 1 + r, j) + a * 6;
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    if (c <= b) v += Bl[c] * s_Li[j & 1][b * 6 + c];
                s_col[t] = v;
            }
            block_sync_lds();
            for (int t = tid; t < m * 36; t += nt) K.val[(size_t)(s_rbase[j + 1 + t / 36] + j) * 36 + t % 36] = s_col[t];
            const int npair = m * (m + 1) / 2;
            if (tid < 64) {
                if (m > 0 && tid < 6) update_item(j, tid);  // pair (0, 0) = block (j + 1, j + 1)
                wave_lds_order();
                if (j + 1 < je)
                    sky_pivot(slot(j + 1, j + 1), K.val + (size_t)(s_rbase[j + 1] + j + 1) * 36, K.dinv + (size_t)(j + 1) * 36, s_Li[(j + 1) & 1], &s_fail, tid);
            }
            else {
                if (tid >= nt - 64) {  // the forward substitution of this column rides along on the last wave: z_j = L_jj^-1 y_j, y_i -= L_ij z_j
                    const int lane = tid - (nt - 64);  // (the sums in the order of sky_forward_narrow: the same bits)
                    const double* Li = s_Li[j & 1];
                    double z[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        double v = 0.0;
#pragma unroll
                        for (int c = 0; c < 6; ++c)
                            if (c <= r) v += Li[r * 6 + c] * s_y[j * 6 + c];
                        z[r] = v;
                    }
                    wave_lds_order();  // every lane has read y_j
                    if (lane < 6) s_y[j * 6 + lane] = lane == 0 ? z[0] : lane == 1 ? z[1] : lane == 2 ? z[2] : lane == 3 ? z[3] : lane == 4 ? z[4] : z[5];
                    for (int t = lane; t < m * 6; t += 64) {
                        const double* Bl = s_col + t * 6;  // row (t / 6, t % 6) of the scaled column: s_col[r * 36 + a * 6 + c]
                        double u = Bl[0] * z[0];
#pragma unroll
                        for (int c = 1; c < 6; ++c) u += Bl[c] * z[c];
                        s_y[(j + 1) * 6 + t] -= u;
                    }
                }
                for (int t = 6 + (tid - 64); t < npair * 6; t += nt - 64) update_item(j, t);
            }
            if (inew < nP) {  // row j's slots are free (its diagonal was read by the pivot of column j long ago)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int t = tid + q * SKY_BAND_THREADS;
                    if (t < npre) slot(inew, fnew + t / 36)[t % 36] = pre[q];
                }
            }
            block_sync_lds();
        }
    };
    const int nA = T.on ? (who ? T.nB : T.m) : nP;  // columns of the first stretch
    const int Ws = T.W, npairS = Ws * (Ws + 1) / 2;
    double* const xS2 = T.xch;                      // the second plan's S x S window blocks, its own (reversed) order
    double* const xY = T.xch + (size_t)npairS * 36; // its y_S
    double* const xX = xY + 6 * Ws;                 // x_S of the first plan
    if (tid < 64) sky_pivot(slot(0, 0), K.val + (size_t)s_rbase[0] * 36, K.dinv, s_Li[0], &s_fail, tid);
    block_sync_lds();
    factor(0, nA);
    __syncthreads();
    if (T.on && who == 1) {  // hand the Schur complement on S over (rows nB .. nB + W - 1 are the trailing rows of the window)
        for (int t = tid; t < npairS * 36; t += nt) {
            const int x = t / 36, e = t - 36 * x;
            int r, c;
            tri_index(x, r, c);
            xS2[t] = slot(nA + r, nA + c)[e];
        }
        for (int t = tid; t < 6 * Ws; t += nt) xY[t] = s_y[nA * 6 + t];  // ... and its share of the right-hand side of S (the forward pass rode along)
        __syncthreads();
        if (tid == 0) sky_post(T.flags + 0, s_fail ? -epoch : epoch);
    }
    if (T.on && who == 0) {
        if (tid == 0) {
            const int v = sky_wait(T.flags + 0, epoch);
            if (v < 0) s_other = v;
        }
        __syncthreads();
        if (s_other < 0 && tid == 0) s_fail = 1;
        __syncthreads();
        if (!s_fail) {  // block (i, k) of S here is the transposed block (k', i') there, i' = nP_total - 1 - i
            for (int t = tid; t < npairS * 36; t += nt) {
                const int x = t / 36, e = t - 36 * x, a = e / 6, b = e - 6 * a;
                int r, c;
                tri_index(x, r, c);
                const int k = nA + Ws - 1 - r, i = nA + Ws - 1 - c;
                slot(i, k)[b * 6 + a] += sky_peek(xS2 + t);
            }
            for (int t = tid; t < 6 * Ws; t += nt) s_y[(nA + t / 6) * 6 + t % 6] += sky_peek(xY + (Ws - 1 - t / 6) * 6 + t % 6);
            __syncthreads();
            if (tid < 64) sky_pivot(slot(nA, nA), K.val + (size_t)(s_rbase[nA] + nA) * 36, K.dinv + (size_t)nA * 36, s_Li[nA & 1], &s_fail, tid);
            block_sync_lds();
            factor(nA, nP);
            __syncthreads();
        }
    }
    // ---- backward substitutions on the first wave (the factor blocks written above are read back from global memory; the forward
    // pass went with the factorisation)
    if (tid < 64) {
        const int lane = tid;
        if (!T.on) {
            if (!s_fail) sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, nP - 1, 0);
        }
        else if (who == 1) {
            int v = 0;
            if (lane == 0) v = sky_wait(T.flags + 2, epoch);
            v = __builtin_amdgcn_readfirstlane(v);
            if (v < 0 && lane == 0) s_fail = 1;
            wave_lds_order();
            if (!s_fail) {  // x of position nB + q here is x of position m + W - 1 - q there
                for (int t = lane; t < 6 * Ws; t += 64) s_y[(nA + t / 6) * 6 + t % 6] = sky_peek(xX + (Ws - 1 - t / 6) * 6 + t % 6);
                wave_lds_order();
                sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, nA - 1, 0);
            }
        }
        else {
            if (!s_fail) {
                sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, nP - 1, nA);
                wave_lds_order();
                for (int t = lane; t < 6 * Ws; t += 64) xX[t] = s_y[nA * 6 + t];
            }
            if (lane == 0) sky_post(T.flags + 2, s_fail ? -epoch : epoch);
            if (!s_fail) sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, nA - 1, 0);
        }
    }
    __syncthreads();
    const int own_to = T.on && who == 1 ? nA : nP;  // the separator's unknowns are written by the first workgroup
    if (s_fail) {
        if (tid == 0) D.ctl->solve_failed = 1;
        for (int t = tid; t < D.n; t += nt) {
            const int pa = K.pos[t / 6];
            if (pa >= 0 && pa < own_to) D.dp[t] = 0.0;
        }
        return;
    }
    for (int t = tid; t < D.n; t += nt) {
        const int a = t / 6, c = t - 6 * a, pa = K.pos[a];
        if (pa >= 0 && pa < own_to) D.dp[t] = s_y[pa * 6 + c];
    }
}

struct SkyPlan {
    SkyDev dev[2];     // [1] only for the two-sided elimination
    SkyTwist twist;
    int epoch = 0;
    void* d_int = nullptr;
    size_t int_bytes = 0;
    void* d_val = nullptr;
    size_t val_bytes = 0, val_used = 0;
    bool usable = false;
};

// the index arrays of one plan on the host: elimination order `order` (position -> slot) over a subset of the slots
struct HostSky {
    std::vector<int> pos, first, rowoff, coloff, colrows, colbase, diag;
    std::vector<int2> blkmap;
    size_t nblocks = 0;
    int nP = 0, max_m = 0;
    bool band = false, ok = false;
};

// reverse Cuthill-McKee over the block graph (every component from a minimum-degree node of a far BFS level)
std::vector<int> rcm_order(int n, const std::vector<std::vector<int>>& adj) {
    std::vector<int> order;
    order.reserve(n);
    std::vector<char> seen(n, 0);
    std::vector<int> level(n);
    auto bfs = [&](int start, std::vector<int>& out) {
        out.clear();
        out.push_back(start);
        level[start] = 0;
        std::vector<char> mark(n, 0);
        mark[start] = 1;
        for (size_t h = 0; h < out.size(); ++h) {
            const int u = out[h];
            std::vector<int> nb;
            for (int v : adj[u])
                if (!mark[v] && !seen[v]) nb.push_back(v);
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
            for (int v : nb) {
                mark[v] = 1;
                level[v] = level[u] + 1;
                out.push_back(v);
            }
        }
    };
    std::vector<int> comp;
    for (int s0 = 0; s0 < n; ++s0) {
        if (seen[s0]) continue;
        int start = s0;
        bfs(start, comp);
        for (int rep = 0; rep < 2; ++rep) {  // pseudo-peripheral node: restart from a minimum-degree node of the last level
            int far = comp.back();
            for (int v : comp)
                if (level[v] == level[comp.back()] && adj[v].size() < adj[far].size()) far = v;
            if (far == start) break;
            start = far;
            bfs(start, comp);
        }
        for (int v : comp) {
            seen[v] = 1;
            order.push_back(v);
        }
    }
    std::reverse(order.begin(), order.end());
    return order;
}
}  // namespace

void sv_sky_release(svgpu_ctx* ctx) {
    SkyPlan* P = (SkyPlan*)ctx->ba_sky;
    if (!P) return;
    if (P->d_int) (void)hipFree(P->d_int);
    if (P->d_val) (void)hipFree(P->d_val);
    delete P;
    ctx->ba_sky = nullptr;
}

// Index arrays of the envelope over `order` (position -> slot; a subset of the nS slots).  Positions >= sep are separator rows of a
// two-sided elimination: dense among themselves after the elimination (their envelopes reach back to `sep`); with drop_sep their own
// blocks are not assembled into this plan (the other plan holds them).
static void host_sky(int nS, const std::vector<int>& order, const std::vector<std::vector<int>>& adj, const std::vector<int2>& blk_ab, int sep, bool drop_sep,
                     size_t max_bytes, HostSky& H) {
    const int nP = (int)order.size();
    H = HostSky();
    H.nP = nP;
    H.pos.assign(nS, -1);
    H.first.resize(nP);
    H.rowoff.resize(nP + 1);
    for (int i = 0; i < nP; ++i) H.pos[order[i]] = i;
    for (int i = 0; i < nP; ++i) {
        int f = i;
        for (int v : adj[order[i]])
            if (H.pos[v] >= 0) f = std::min(f, H.pos[v]);
        if (i >= sep) f = std::min(f, sep);
        H.first[i] = f;
    }
    // A non-decreasing first[] makes the rows of every column contiguous (the banded kernel); taking the suffix minimum only adds
    // blocks, so it is kept when the envelope grows by less than a third and stays narrow.
    {
        std::vector<int> fm(H.first);
        for (int i = nP - 2; i >= 0; --i) fm[i] = std::min(fm[i], fm[i + 1]);
        size_t n0 = 0, n1 = 0;
        int wmax = 0;
        for (int i = 0; i < nP; ++i) {
            n0 += (size_t)(i - H.first[i] + 1);
            n1 += (size_t)(i - fm[i] + 1);
            wmax = std::max(wmax, i - fm[i]);
        }
        if (wmax <= SKY_BAND_W && 3 * n1 <= 4 * n0 && !std::getenv("SVGPU_SKY_NO_BAND")) {
            H.first = fm;
            H.band = true;
        }
    }
    size_t nblocks = 0;
    for (int i = 0; i < nP; ++i) {
        H.rowoff[i] = (int)nblocks;
        nblocks += (size_t)(i - H.first[i] + 1);
        if (nblocks * 288 > max_bytes || nblocks > (size_t)1 << 30) return;
    }
    H.rowoff[nP] = (int)nblocks;
    H.nblocks = nblocks;
    H.coloff.assign(nP + 1, 0);
    for (int i = 0; i < nP; ++i)
        for (int j = H.first[i]; j < i; ++j) H.coloff[j + 1]++;
    for (int j = 0; j < nP; ++j) {
        H.max_m = std::max(H.max_m, H.coloff[j + 1]);
        H.coloff[j + 1] += H.coloff[j];
    }
    if (H.max_m > SKY_MAXM) return;
    H.colrows.resize(H.coloff[nP]);
    H.colbase.resize(H.coloff[nP]);
    H.diag.resize(nP);
    std::vector<int> fill(H.coloff.begin(), H.coloff.end() - 1);
    for (int i = 0; i < nP; ++i) {  // rows ascending
        H.diag[i] = H.rowoff[i] + i - H.first[i];
        for (int j = H.first[i]; j < i; ++j) {
            H.colbase[fill[j]] = H.rowoff[i] - H.first[i];
            H.colrows[fill[j]++] = i;
        }
    }
    // scaled column + right-hand side + index arrays must fit the LDS
    if ((size_t)H.max_m * 288 + (size_t)nP * 48 + 4 * ((size_t)2 * nP + 1 + 2 * H.colrows.size()) > 150 * 1024) return;
    H.blkmap.resize(blk_ab.size());
    for (size_t k = 0; k < blk_ab.size(); ++k) {
        const int pa = H.pos[blk_ab[k].x], pb = H.pos[blk_ab[k].y];
        int2 m;
        m.x = -1, m.y = 0;
        if (pa >= 0 && pb >= 0 && !(drop_sep && pa >= sep && pb >= sep)) {
            const int row = std::max(pa, pb), col = std::min(pa, pb);
            m.x = H.rowoff[row] + col - H.first[row];
            m.y = pa > pb || pa == pb ? 0 : 1;  // the kept block is S_ab; the envelope stores S_{row, col}: transposed when row = pos[b]
        }
        H.blkmap[k] = m;
    }
    H.ok = true;
}

// Plans the envelope factorisation of the reduced system whose kept upper blocks are blk_ab (a <= b, free-pose slots).  *usable = false
// (and nothing else changes) when the envelope would exceed max_bytes or a column has more than SKY_MAXM rows: the caller keeps the PCG.
int sv_sky_plan(svgpu_ctx* ctx, hipStream_t s, int nP, const std::vector<int2>& blk_ab, size_t max_bytes, bool* usable) {
    *usable = false;
    if (nP <= 0) return SVGPU_OK;
    std::vector<std::vector<int>> adj(nP);
    for (const int2& ab : blk_ab)
        if (ab.x != ab.y) {
            adj[ab.x].push_back(ab.y);
            adj[ab.y].push_back(ab.x);
        }
    const std::vector<int> order = rcm_order(nP, adj);  // position -> slot
    HostSky H[2];
    host_sky(nP, order, adj, blk_ab, nP, false, max_bytes, H[0]);
    if (!H[0].ok) return SVGPU_OK;
    // two-sided elimination of a long band: T | S | B with S = as many rows as the band is wide (then no block couples T and B)
    int nplans = 1, tw_m = 0, tw_W = 0, tw_nB = 0;
    if (H[0].band && H[0].max_m >= 1 && nP >= 8 * (H[0].max_m + 1) && !std::getenv("SVGPU_SKY_ONE_SIDED")) {
        const int W = H[0].max_m, m = (nP - W) / 2, nB = nP - m - W;
        std::vector<int> o0(order.begin(), order.begin() + m + W), o1(order.rbegin(), order.rbegin() + nB + W);
        HostSky A, B;
        host_sky(nP, o0, adj, blk_ab, m, false, max_bytes, A);
        host_sky(nP, o1, adj, blk_ab, nB, true, max_bytes, B);
        const int Ww = std::max(W, std::max(A.max_m, B.max_m));  // one window size, wide enough for all of S
        auto band_lds = [&](const HostSky& h) {
            return ((size_t)(Ww + 1) * (Ww + 1) * 36 + (size_t)Ww * 36 + (size_t)6 * h.nP) * sizeof(double) + 4 * ((size_t)3 * h.nP + 1 + 2 * h.colrows.size());
        };
        if (A.ok && B.ok && A.band && B.band && Ww <= SKY_BAND_W && std::max(band_lds(A), band_lds(B)) <= 150 * 1024) {
            A.max_m = B.max_m = Ww;
            H[0] = std::move(A);
            H[1] = std::move(B);
            nplans = 2, tw_m = m, tw_W = W, tw_nB = nB;
        }
    }
    SkyPlan* P = (SkyPlan*)ctx->ba_sky;
    if (!P) {
        P = new SkyPlan();
        ctx->ba_sky = P;
    }
    // one integer arena per call: for every plan pos | first | rowoff | coloff | colrows | colbase | diag | blkmap; then the flags
    size_t n_int = 16, n_val = 8;
    for (int q = 0; q < nplans; ++q) {
        n_int += (size_t)nP + (size_t)H[q].nP * 2 + (size_t)(H[q].nP + 1) * 2 + 2 * H[q].colrows.size() + H[q].blkmap.size() * 2 + 8;
        n_val += H[q].nblocks * 36 + (size_t)H[q].nP * 36 + (size_t)H[q].nP * 6;
    }
    n_val += (size_t)tw_W * (tw_W + 1) / 2 * 36 + 12 * (size_t)tw_W;
    if (n_int * 4 > P->int_bytes) {
        if (P->d_int) {
            SV_HIP(ctx, hipStreamSynchronize(s));
            SV_HIP(ctx, hipFree(P->d_int));
            P->d_int = nullptr;
            P->int_bytes = 0;
        }
        SV_HIP(ctx, hipMalloc(&P->d_int, n_int * 4 + n_int));
        P->int_bytes = n_int * 4 + n_int;
    }
    const size_t val_need = n_val * sizeof(double);
    if (val_need > P->val_bytes) {
        if (P->d_val) {
            SV_HIP(ctx, hipStreamSynchronize(s));
            SV_HIP(ctx, hipFree(P->d_val));
            P->d_val = nullptr;
            P->val_bytes = 0;
        }
        SV_HIP(ctx, hipMalloc(&P->d_val, val_need + val_need / 4));
        P->val_bytes = val_need + val_need / 4;
    }
    P->val_used = val_need;
    std::vector<int> host;
    host.reserve(n_int);
    auto put = [&](const int* p, size_t n) {
        if (host.size() & 1) host.push_back(0);  // (int2 alignment for whoever needs it)
        const size_t at = host.size();
        host.insert(host.end(), p, p + n);
        return at;
    };
    struct Off {
        size_t pos, first, rowoff, coloff, colrows, colbase, diag, blkmap;
    } off[2];
    for (int q = 0; q < nplans; ++q) {
        const HostSky& h = H[q];
        off[q].pos = put(h.pos.data(), h.pos.size());
        off[q].first = put(h.first.data(), h.nP);
        off[q].rowoff = put(h.rowoff.data(), h.nP + 1);
        off[q].coloff = put(h.coloff.data(), h.nP + 1);
        off[q].colrows = put(h.colrows.data(), h.colrows.size());
        off[q].colbase = put(h.colbase.data(), h.colbase.size());
        off[q].diag = put(h.diag.data(), h.nP);
        off[q].blkmap = put(reinterpret_cast<const int*>(h.blkmap.data()), h.blkmap.size() * 2);
    }
    const int zeros[4] = {0, 0, 0, 0};
    const size_t o_flags = put(zeros, 4);
    SV_HIP(ctx, hipMemcpyAsync(P->d_int, host.data(), host.size() * 4, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipStreamSynchronize(s));  // `host` is a pageable temporary
    int* di = (int*)P->d_int;
    double* dv = (double*)P->d_val;
    for (int q = 0; q < 2; ++q) P->dev[q] = SkyDev();
    for (int q = 0; q < nplans; ++q) {
        const HostSky& h = H[q];
        SkyDev& K = P->dev[q];
        K.nP = h.nP;
        K.NB = (int)blk_ab.size();
        K.pos = di + off[q].pos;
        K.first = di + off[q].first;
        K.rowoff = di + off[q].rowoff;
        K.coloff = di + off[q].coloff;
        K.colrows = di + off[q].colrows;
        K.colbase = di + off[q].colbase;
        K.diag = di + off[q].diag;
        K.max_m = h.max_m;
        K.ncr = (int)h.colrows.size();
        K.band = h.band ? 1 : 0;
        K.blkmap = reinterpret_cast<const int2*>(di + off[q].blkmap);
        K.val = dv;
        K.dinv = K.val + h.nblocks * 36;
        K.y = K.dinv + (size_t)h.nP * 36;
        K.nblocks = h.nblocks;
        dv = K.y + (size_t)h.nP * 6;
    }
    P->twist = SkyTwist();
    if (nplans == 2) {
        P->dev[1].y_from = tw_nB;  // the right-hand side of S goes into the first plan only
        P->twist.on = 1, P->twist.m = tw_m, P->twist.W = tw_W, P->twist.nB = tw_nB;
        P->twist.xch = dv;
        P->twist.flags = di + o_flags;
    }
    P->epoch = 0;
    P->usable = true;
    *usable = true;
    if (std::getenv("SVGPU_BA_TRACE"))
        std::fprintf(stderr, "[ba]     envelope plan: %d block rows, %zu blocks (%.1f MB), widest column %d rows%s%s\n", nP, H[0].nblocks + (nplans == 2 ? H[1].nblocks : 0),
                     (H[0].nblocks + (nplans == 2 ? H[1].nblocks : 0)) * 288.0 / 1048576.0, H[0].max_m, H[0].band ? ", banded" : "",
                     nplans == 2 ? ", two-sided elimination" : "");
    return SVGPU_OK;
}

void sv_sky_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    SkyPlan* P = (SkyPlan*)ctx->ba_sky;
    SvProfScope ps(ctx, s, "ba_solve");
    const SkyDev& K = P->dev[0];
    const int nplans = P->twist.on ? 2 : 1;
    // blocks of the envelope the reduced system does not fill must start at zero in every trial (and the right-hand side rows a plan is not given)
    (void)hipMemsetAsync(P->d_val, 0, P->val_used, s);
    const size_t items = std::max((size_t)K.NB * 36, (size_t)D.n);
    hipLaunchKernelGGL(k_sky_assemble, dim3((unsigned)((items + 255) / 256), nplans), dim3(256), 0, s, D, P->dev[0], P->dev[1]);
    if (K.band) {
        size_t lds = 0;
        for (int q = 0; q < nplans; ++q) {
            const SkyDev& k = P->dev[q];
            const size_t wn = (size_t)k.max_m + 1;
            lds = std::max(lds, (wn * wn * 36 + (size_t)k.max_m * 36 + (size_t)6 * k.nP) * sizeof(double) + 4 * ((size_t)3 * k.nP + 1 + 2 * (size_t)k.ncr));
        }
        if (lds <= 150 * 1024) {
            (void)sv_allow_dynamic_lds((const void*)k_sky_band, lds);
            hipLaunchKernelGGL(k_sky_band, dim3(nplans), dim3(SKY_BAND_THREADS), lds, s, D, P->dev[0], P->dev[1], P->twist, ++P->epoch);
            return;
        }
    }
    const size_t lds = ((size_t)K.max_m * 36 + (size_t)D.n) * sizeof(double) + 4 * ((size_t)2 * K.nP + 1 + 2 * (size_t)K.ncr);
    (void)sv_allow_dynamic_lds((const void*)k_sky_factor_solve, lds);
    hipLaunchKernelGGL(k_sky_factor_solve, dim3(1), dim3(K.max_m <= 21 ? 256 : SKY_THREADS), lds, s, D, K);
}
