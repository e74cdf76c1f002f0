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
    const int* order = nullptr;     // position -> slot (the plans of a segmented elimination only)
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

// Segmented elimination (the N-piece generalisation of the two-sided one, and the form in which the factorisation is DISTRIBUTED over the
// ranks of a sharded solve).  The ordered keyframe graph is cut by vertex separators; every connected piece between the cuts is an
// independent JOB: a plan over [the piece in an order that ends at its separators, its adjacent separator rows], of which only the
// piece's own columns are eliminated -- what it leaves on its separator rows (a dense Schur complement and a share of their right-hand
// side) goes to an exchange buffer.  The separator system (original separator blocks + every job's contribution, summed in job order)
// is factored and solved on its own plan, and every job finishes with the backward substitution of its columns.  One workgroup per
// job, jobs of different ranks on different GPUs: the exchange buffer and the solution are the only things that cross ranks, and since
// every rank contributes zeros outside its own jobs the all-reduce is an all-gather -- results do not depend on the world size.
struct SegJobDev {
    SkyDev K;      // plan of [piece, adjacent separator rows]
    int nC, ns;    // columns to eliminate; separator rows behind them
    double* xch;   // [ns (ns + 1) / 2 x 36: Schur complement on the separator rows, lower triangle row-major][6 ns: their right-hand side share]
};

// kept blocks / right-hand side of the reduced system -> the plans of a segmented elimination (ONE map over the whole value arena)
__global__ __launch_bounds__(256) void k_seg_assemble(BaDev D, const int2* __restrict__ blkmap, const int* __restrict__ yoff, double* __restrict__ arena, int NB) {
    if (D.ctl->phase != 1) return;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < (size_t)NB * 36) {
        const int k = (int)(t / 36), e = (int)(t - (size_t)k * 36), i = e / 6, j = e - 6 * i;
        const int2 m = blkmap[k];
        if (m.x >= 0) arena[(size_t)m.x * 36 + (m.y ? j * 6 + i : e)] = D.Sblk[t];
    }
    if (t < (size_t)D.n) {
        const int a = (int)(t / 6), c = (int)(t - (size_t)a * 6);
        const int o = yoff[a];
        if (o >= 0) arena[(size_t)o + c] = D.g[t];
    }
}

// separator system += the jobs' contributions, in job order (fixed order of the sums: bit-reproducible, and the same on every rank)
__global__ __launch_bounds__(256) void k_seg_gather(BaDev D, SkyDev K, const int* __restrict__ gb_off, const int* __restrict__ gb_src, const int* __restrict__ gy_off,
                                                     const int* __restrict__ gy_src, const double* __restrict__ xch) {
    if (D.ctl->phase != 1) return;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < K.nblocks * 36) {
        const int b = (int)(t / 36), e = (int)(t - (size_t)b * 36), i = e / 6, j = e - 6 * i;
        double v = K.val[t];
        for (int q = gb_off[b]; q < gb_off[b + 1]; ++q) {
            const int src = gb_src[q];
            v += xch[(size_t)(src & 0x3fffffff) * 36 + ((src >> 30) & 1 ? j * 6 + i : e)];
        }
        K.val[t] = v;
    }
    if (t < (size_t)K.nP * 6) {
        const int p = (int)(t / 6), c = (int)(t - (size_t)p * 6);
        double v = K.y[t];
        for (int q = gy_off[p]; q < gy_off[p + 1]; ++q) v += xch[(size_t)gy_src[q] + c];
        K.y[t] = v;
    }
}

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
// v of the lane a DPP control selects (a cross-lane move inside the VALU: ~10 cycles; __shfl_xor goes through the LDS crossbar, ~100 cycles,
// and queues behind the workgroup's LDS traffic -- three of them sat on the dependent chain of every backward-substitution step)
template <int CTRL>
__device__ __forceinline__ double dpp_swap(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
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
// (SV_LDS: LDS operands carry their address space.  Through generic pointers every LDS access of these kernels was a FLAT instruction --
//  639 of them in k_sky_band --, which travels the global-memory path: several times the latency of a ds_ access and a wait on the vector
//  memory counter, i.e. on the factor rows that are deliberately in flight.  The helpers are templates on the pointer types so that the
//  same code serves operands in LDS and in global memory.)
#define SV_LDS __attribute__((address_space(3)))
#define SV_GLB __attribute__((address_space(1)))
// (the plan's arrays are GLOBAL: in job mode the SkyDev comes out of memory, its pointers are generic to the compiler and every access
//  through them a FLAT instruction, which counts on the LDS counter too and so serialises the factor rows in flight with the LDS work)
struct SkyG {
    SV_GLB double* val;
    SV_GLB double* dinv;
    SV_GLB double* y;
    const SV_GLB int *coloff, *first, *rowoff, *colrows, *colbase, *pos, *order;
    __device__ __forceinline__ explicit SkyG(const SkyDev& K)
        : val((SV_GLB double*)K.val), dinv((SV_GLB double*)K.dinv), y((SV_GLB double*)K.y), coloff((const SV_GLB int*)K.coloff), first((const SV_GLB int*)K.first),
          rowoff((const SV_GLB int*)K.rowoff), colrows((const SV_GLB int*)K.colrows), colbase((const SV_GLB int*)K.colbase), pos((const SV_GLB int*)K.pos),
          order((const SV_GLB int*)K.order) {}
};
template <class DstP, class LiP, class FailP>
__device__ __forceinline__ void sky_pivot_rows(double (&a)[6], DstP dstL, DstP dstInv, LiP s_Li, FailP s_fail, int tid);
template <class SrcP, class DstP, class LiP, class FailP>
__device__ __forceinline__ void sky_pivot(SrcP src, DstP dstL, DstP dstInv, LiP s_Li, FailP s_fail, int tid) {
    const int r = min(tid, 5);
    double a[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = src[r * 6 + c];  // lane r < 6: row r (lanes >= 6 mirror row 5 and write nothing)
    sky_pivot_rows(a, dstL, dstInv, s_Li, s_fail, tid);
}
// (a: row min(lane, 5) of the block, in registers)
template <class DstP, class LiP, class FailP>
__device__ __forceinline__ void sky_pivot_rows(double (&a)[6], DstP dstL, DstP dstInv, LiP s_Li, FailP s_fail, int tid) {
    double Lm[6][6], rd[6];  // the finished factor and its reciprocal diagonal, wave-uniform
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < c) v = fma(-a[k], Lm[c][k], v);  // a[k] = L[r][k] by now (fma: the pivot is the dependent chain of the whole factorisation)
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
            if (k < rr) v = fma(-(k >= cc ? Lm[rr][k] : 0.0), x[k], v);
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
template <class YP, class IP>
__device__ __forceinline__ void sky_forward_narrow(const SkyDev& K, YP s_y, IP s_coloff, IP s_rows, IP s_base, int lane, int jb, int je) {
    constexpr int PD = 4;
    const SkyG Gk(K);
    {   // forward: lane rr < 6 holds row rr of Li_j; item k of a lane = (row (lane + 64 k) / 6 of the column, component (lane + 64 k) % 6)
        auto load = [&](int j, double (&li)[6], double (&bl)[2][6], int (&dst)[2]) {
            if (j >= je) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int rr = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) li[c] = Gk.dinv[(size_t)j * 36 + rr * 6 + c];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = lane + 64 * k;
                dst[k] = -1;
                if (t < m * 6) {
                    const int r = t / 6, a = t - 6 * r;
                    const SV_GLB double* Bl = Gk.val + (size_t)(s_base[c0 + r] + j) * 36 + a * 6;
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
template <class YP, class IP>
__device__ __forceinline__ void sky_backward_narrow(const SkyDev& K, YP s_y, IP s_coloff, IP s_rows, IP s_base, int lane, int jhi, int jlo) {
    constexpr int PD = 4;
    const SkyG Gk(K);
    {   // backward: w = z_j - sum_{i in rows(j)} L_ij^T x_i, lane = component * 8 + part (48 lanes), a lane's rows: part and part + 8
        const int a = min(lane >> 3, 5), part = lane & 7;
        auto load = [&](int j, double (&lc)[6], double (&bc)[2][6], int (&src)[2]) {
            if (j < jlo) return;
            const int c0 = s_coloff[j], m = s_coloff[j + 1] - c0;
            const int ac = min(lane, 5);
#pragma unroll
            for (int c = 0; c < 6; ++c) lc[c] = Gk.dinv[(size_t)j * 36 + c * 6 + ac];  // column `lane` of Li_j
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = part + 8 * k;
                src[k] = -1;
                if (lane < 48 && r < m) {
                    const SV_GLB double* Bl = Gk.val + (size_t)(s_base[c0 + r] + j) * 36;
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
                // z_j is read with the x's of the rows below (one LDS round trip per step, not two); the two items of a lane are summed apart
                // (two chains of six fused multiply-adds instead of one of twelve multiply + add pairs)
                double zj[6], vk[2] = {0.0, 0.0};
#pragma unroll
                for (int c = 0; c < 6; ++c) zj[c] = s_y[j * 6 + c];
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (src[d][k] >= 0)
#pragma unroll
                        for (int c = 0; c < 6; ++c) vk[k] = fma(bc[d][k][c], s_y[src[d][k] + c], vk[k]);
                double v = vk[0] + vk[1];
                v += dpp_swap<0xB1>(v);   // fixed tree inside the component's eight lanes: lane ^ 1 (quad_perm [1, 0, 3, 2]),
                v += dpp_swap<0x4E>(v);   // lane ^ 2 (quad_perm [2, 3, 0, 1]),
                v += dpp_swap<0x141>(v);  // the other quad (row_half_mirror: lane 7 - i; every lane of a quad holds the quad's sum by now)
                double w[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) w[c] = zj[c] - lane_bcast(v, 8 * c);
                wave_lds_order();
                if (lane < 6) {  // x_j = Li_j^T w: component a = sum_{c >= a} Li[c][a] w[c]
                    double x = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        if (c >= lane) x = fma(lc[d][c], w[c], x);
                    s_y[j * 6 + lane] = x;
                }
                wave_lds_order();
                load(j - PD, lc[d], bc[d], src[d]);
            }
        }
    }
}
template <class YP, class IP>
__device__ __forceinline__ void sky_substitute_narrow(const SkyDev& K, YP s_y, IP s_coloff, IP s_rows, IP s_base, int lane) {
    sky_forward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, 0, K.nP);
    sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, K.nP - 1, 0);
}

template <class YP, class IP>
__device__ __forceinline__ void sky_substitute(const SkyDev& K, YP s_y, IP s_coloff, IP s_rows, IP s_base, int tid) {
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
    const int ny = 6 * nP;  // (the plan may cover a subset of the slots: the separator system of a segmented elimination)
    int* const s_coloff = reinterpret_cast<int*>(s_y + ny);
    int* const s_diag = s_coloff + (nP + 1);
    int* const s_rows = s_diag + nP;
    int* const s_base = s_rows + K.ncr;
    if (tid == 0) s_fail = 0;
    for (int t = tid; t < ny; t += nt) s_y[t] = K.y[t];
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
        for (int t = tid; t < D.n; t += nt)
            if (K.pos[t / 6] >= 0) D.dp[t] = 0.0;
        return;
    }
    // ---------------------------------------------------------------- L z = g (column oriented), then L^T x = z: the first wave alone
    if (tid < 64) sky_substitute(K, s_y, s_coloff, s_rows, s_base, tid);
    __syncthreads();
    for (int t = tid; t < D.n; t += nt) {
        const int a = t / 6, c = t - 6 * a, pa = K.pos[a];
        if (pa >= 0) D.dp[t] = s_y[pa * 6 + c];  // (a plan over a subset of the slots -- the separator system of a segmented elimination -- writes its own only)
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
// (BOUNDED: the two workgroups of a two-sided elimination are launched side by side, not cooperatively -- should the partner never become
//  resident (an over-subscribed device) the wait gives up after ~0.5 s of polling and reports the partner as failed: the trial is then a
//  failed factorisation, which the LM loop answers with more damping, instead of a hang in the mapping thread)
#define SKY_WAIT_POLLS (1 << 21)
__device__ __forceinline__ int sky_wait(int* flag, int epoch) {  // +-epoch
    int v = 0;
    for (int polls = 0; polls < SKY_WAIT_POLLS; ++polls) {
        v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == epoch || v == -epoch) return v;
        __builtin_amdgcn_s_sleep(4);
    }
    return -epoch;
}
__device__ __forceinline__ double sky_peek(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // past this unit's L1

__global__ __launch_bounds__(SKY_BAND_THREADS) void k_sky_band(BaDev D, SkyDev K0, SkyDev K1, SkyTwist T_in, int epoch, const SegJobDev* __restrict__ jobs) {
    if (D.ctl->phase != 1) return;
    // jobs != null: workgroup b eliminates the nC columns of job b and exports what is left on its separator rows (the role of the
    // second workgroup of a two-sided elimination, without the hand-over flags: the separator solve and the backward pass are launches
    // of their own); else 0: the whole system, or [T, S] of a two-sided elimination; 1: the reversed [B, S]
    const int who = jobs ? 1 : (int)blockIdx.x;
    SkyDev Kj;
    SkyTwist T = T_in;
    if (jobs) {
        const SegJobDev J = jobs[blockIdx.x];
        Kj = J.K;
        T.on = 1, T.m = 0, T.W = J.ns, T.nB = J.nC, T.xch = J.xch;
    }
    const SkyDev& K = jobs ? Kj : (who ? K1 : K0);
    const SkyG Gk(K);  // the plan's arrays as global-address-space pointers
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
    SV_LDS double* const s_win = (SV_LDS double*)s_dyn;
    SV_LDS double* const s_col = s_win + Wn * Wn * 36;
    SV_LDS double* const s_y = s_col + W * 36;
    SV_LDS int* const s_coloff = (SV_LDS int*)(s_y + ny);
    SV_LDS int* const s_first = s_coloff + (nP + 1);
    SV_LDS int* const s_rbase = s_first + nP;   // rowoff[i] - first[i]: block (i, k) of the envelope = s_rbase[i] + k
    SV_LDS int* const s_rows = s_rbase + nP;    // for the substitution routine
    SV_LDS int* const s_base = s_rows + K.ncr;
    SV_LDS double* const s_Li0 = (SV_LDS double*)s_Li[0];
    SV_LDS double* const s_Li1 = (SV_LDS double*)s_Li[1];
    SV_LDS int* const s_failp = (SV_LDS int*)&s_fail;
    auto Li_of = [&](int j) { return (j & 1) ? s_Li1 : s_Li0; };
    if (tid == 0) s_fail = 0, s_other = epoch;
    for (int t = tid; t < ny; t += nt) s_y[t] = Gk.y[t];
    for (int t = tid; t <= nP; t += nt) s_coloff[t] = Gk.coloff[t];
    for (int t = tid; t < nP; t += nt) {
        s_first[t] = Gk.first[t];
        s_rbase[t] = Gk.rowoff[t] - Gk.first[t];
    }
    for (int t = tid; t < K.ncr; t += nt) {
        s_rows[t] = Gk.colrows[t];
        s_base[t] = Gk.colbase[t];
    }
    __syncthreads();
    auto slot = [&](int i, int k) { return s_win + ((i % Wn) * Wn + (k % Wn)) * 36; };  // (two integer divisions: set-up and hand-over code only)
    // inside the column loop the window slots are tracked incrementally -- jm = j mod (W + 1) is carried from column to column and a row
    // or column at distance d <= W + 1 from it needs one conditional subtraction -- so that no thread divides by a run-time value there
    auto wrap = [&](int x) { return x >= Wn ? x - Wn : x; };
    auto slot_at = [&](int rs, int cs) { return s_win + (rs * Wn + cs) * 36; };
    // rows 0 .. W of the assembled system
    for (int i = 0; i < min(Wn, nP); ++i) {
        const int f = s_first[i];
        for (int t = tid; t < (i - f + 1) * 36; t += nt) {
            const int k = f + t / 36, e = t % 36;
            slot(i, k)[e] = Gk.val[(size_t)(s_rbase[i] + k) * 36 + e];
        }
    }
    __syncthreads();
    // Column j: [scale] barrier [update of the other waves | first wave: the six rows of block (j + 1, j + 1), then the PIVOT of column
    // j + 1, which needs nothing else of this update] barrier.  Two barriers per column, the pivots off the critical path.
    // S_{ip, iq} -= L_p L_q^T inside the window; jm1 = slot of row / column j + 1.  Item = (pair, rows a0, a0 + 1 of the block): L_q is read
    // once per item, so a pair moves 180 doubles through LDS instead of the 324 of one-row items; every entry is one thread's sum in the
    // same order as before.
    auto update_rows2 = [&](int jm1, int x, int a0) {
        const int pq = s_pq[x], p = pq & 255, q = pq >> 8;
        // every operand is loaded BEFORE the first store: the compiler cannot tell the window from the column buffer (one LDS block), so a
        // store inside the loop made every later load wait for it -- six dependent round trips per item, 2 160 cycles for this phase
        double La[2][6], Lq[6][6], Dv[2][6];
        SV_LDS double* Dst = slot_at(wrap(jm1 + p), wrap(jm1 + q)) + a0 * 6;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) La[r][c] = s_col[p * 36 + (a0 + r) * 6 + c], Dv[r][c] = Dst[r * 6 + c];
#pragma unroll
        for (int bb = 0; bb < 6; ++bb)
#pragma unroll
            for (int c = 0; c < 6; ++c) Lq[bb][c] = s_col[q * 36 + bb * 6 + c];
#pragma unroll
        for (int bb = 0; bb < 6; ++bb)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                double v = La[r][0] * Lq[bb][0];  // explicit fma (the build runs with -ffp-contract=off): half the VALU work of this phase
#pragma unroll
                for (int c = 1; c < 6; ++c) v = fma(La[r][c], Lq[bb][c], v);
                Dv[r][bb] -= v;
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Dst[r * 6 + c] = Dv[r][c];
    };
    // columns jb .. je - 1 (the pivot of column jb is in s_Li when it starts; the pivot of column je is NOT taken: it may wait for the other half)
    auto factor = [&](int jb, int je) {
        int jm = jb % Wn;
        for (int j = jb; j < je; ++j, jm = wrap(jm + 1)) {
            if (s_fail) break;
            const int jm1 = wrap(jm + 1);
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
                    if (t < npre) pre[q] = Gk.val[(size_t)(s_rbase[inew] + fnew + t / 36) * 36 + t % 36];
                }
            }
            for (int t = tid; t < m * 36; t += nt) {  // L_ij = S_ij L_jj^-T
                const int r = t / 36, e = t - r * 36, a = e / 6, b = e - 6 * a;
                const SV_LDS double* Bl = slot_at(wrap(jm1 + r), jm) + a * 6;
                const SV_LDS double* Lj = Li_of(j);
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    if (c <= b) v += Bl[c] * Lj[b * 6 + c];
                s_col[t] = v;
            }
            block_sync_lds();
            // (the first wave has the pivot of the next column ahead of it: the other waves store the finished column)
            if (tid >= 64)
                for (int t = tid - 64; t < m * 36; t += nt - 64) Gk.val[(size_t)(s_rbase[j + 1 + t / 36] + j) * 36 + t % 36] = s_col[t];
            const int npair = m * (m + 1) / 2;
            if (tid < 64) {
                // Block (j + 1, j + 1) and the pivot of column j + 1 in ONE LDS round trip: lane r < 6 loads row r of the block, row r of
                // L_{j+1,j} and the whole of L_{j+1,j}, subtracts its six sums (update_item's, in its order) in registers and goes straight
                // into the pivot -- the updated block is not written back (nothing reads it again: the column scaling takes L_jj^-1).  Through
                // the window (update, store, reload) it was three dependent LDS accesses behind the other waves' update traffic.
                if (j + 1 < je) {
                    const int r = min(tid, 5);
                    double a[6], Lr[6], Lb[6][6];
                    const SV_LDS double* Dg = slot_at(jm1, jm1) + r * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) a[c] = Dg[c];
                    if (m > 0) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) Lr[c] = s_col[r * 6 + c];
#pragma unroll
                        for (int b = 0; b < 6; ++b)
#pragma unroll
                            for (int c = 0; c < 6; ++c) Lb[b][c] = s_col[b * 6 + c];
#pragma unroll
                        for (int b = 0; b < 6; ++b) {
                            double v = Lr[0] * Lb[b][0];
#pragma unroll
                            for (int c = 1; c < 6; ++c) v = fma(Lr[c], Lb[b][c], v);
                            a[b] -= v;
                        }
                    }
                    sky_pivot_rows(a, Gk.val + (s_rbase[j + 1] + j + 1) * 36, Gk.dinv + (j + 1) * 36, Li_of(j + 1), s_failp, tid);
                }
                else if (m > 0 && tid < 36) {  // last column of the stretch: block (j + 1, j + 1) stays in the window for what follows (hand-over or the next stretch)
                    const int a = tid / 6, b = tid - 6 * a;
                    double v = s_col[a * 6] * s_col[b * 6];
#pragma unroll
                    for (int c = 1; c < 6; ++c) v = fma(s_col[a * 6 + c], s_col[b * 6 + c], v);
                    slot_at(jm1, jm1)[tid] -= v;
                }
            }
            else {
                if (tid >= nt - 64) {  // the forward substitution of this column rides along on the last wave: z_j = L_jj^-1 y_j, y_i -= L_ij z_j
                    const int lane = tid - (nt - 64);  // (the sums in the order of sky_forward_narrow: the same bits)
                    const SV_LDS double* Li = Li_of(j);
                    // one LDS round trip for everything the step reads (y_j, L_jj^-1, this lane's rows of the scaled column and their y): the LDS
                    // pipe is busy with the update of the other waves, a dependent access costs several hundred cycles here
                    // (m <= SKY_BAND_W = 16: at most two items per lane)
                    static_assert(SKY_BAND_W * 6 <= 128, "two forward-substitution items per lane");
                    double Bv[2][6], yv[2], yj[6], Lv[21];
#pragma unroll
                    for (int c = 0; c < 6; ++c) yj[c] = s_y[j * 6 + c];
#pragma unroll
                    for (int r = 0, k = 0; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) Lv[k++] = Li[r * 6 + c];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = lane + 64 * k;
                        if (t < m * 6) {
#pragma unroll
                            for (int c = 0; c < 6; ++c) Bv[k][c] = s_col[t * 6 + c];  // row (t / 6, t % 6) of the scaled column: s_col[r * 36 + a * 6 + c]
                            yv[k] = s_y[(j + 1) * 6 + t];
                        }
                    }
                    double z[6];
#pragma unroll
                    for (int r = 0, k = 0; r < 6; ++r) {
                        double v = 0.0;
#pragma unroll
                        for (int c = 0; c <= r; ++c) v += Lv[k++] * yj[c];
                        z[r] = v;
                    }
                    if (lane < 6) s_y[j * 6 + lane] = lane == 0 ? z[0] : lane == 1 ? z[1] : lane == 2 ? z[2] : lane == 3 ? z[3] : lane == 4 ? z[4] : z[5];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = lane + 64 * k;
                        if (t < m * 6) {
                            double u = Bv[k][0] * z[0];
#pragma unroll
                            for (int c = 1; c < 6; ++c) u += Bv[k][c] * z[c];
                            s_y[(j + 1) * 6 + t] = yv[k] - u;
                        }
                    }
                }
                else  // the waves between: the update, two block rows per item (pair 0 went entry by entry on the first wave)
                    for (int t = 3 + (tid - 64); t < npair * 3; t += nt - 128) update_rows2(jm1, t / 3, 2 * (t - 3 * (t / 3)));
            }
            if (inew < nP) {  // row j's slots are free (its diagonal was read by the pivot of column j long ago)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int t = tid + q * SKY_BAND_THREADS;
                    if (t < npre) slot_at(jm, wrap(jm1 + (fnew - (j + 1)) + t / 36))[t % 36] = pre[q];  // row j + W + 1 takes the slots of row j
                }
            }
            block_sync_lds();
        }
    };
    const int nA = T.on ? (who ? T.nB : T.m) : nP;  // columns of the first stretch
    const int Ws = T.W, npairS = Ws * (Ws + 1) / 2;
    SV_GLB double* const xS2 = (SV_GLB double*)T.xch;   // the second plan's S x S window blocks, its own (reversed) order
    SV_GLB double* const xY = xS2 + (size_t)npairS * 36; // its y_S
    SV_GLB double* const xX = xY + 6 * Ws;              // x_S of the first plan
    if (tid < 64) sky_pivot(slot(0, 0), Gk.val + s_rbase[0] * 36, Gk.dinv, s_Li0, s_failp, tid);
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
        if (jobs) {  // a job ends here: z of its columns waits in global memory for k_seg_backward; a failed job contributes zeros
            if (s_fail) {
                for (int t = tid; t < npairS * 36 + 6 * Ws; t += nt) xS2[t] = 0.0;
                if (tid == 0) D.ctl->solve_failed = 1;
            }
            for (int t = tid; t < 6 * nA; t += nt) Gk.y[t] = s_fail ? 0.0 : s_y[t];
            return;
        }
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
                slot(i, k)[b * 6 + a] += sky_peek((const double*)(xS2 + t));
            }
            for (int t = tid; t < 6 * Ws; t += nt) s_y[(nA + t / 6) * 6 + t % 6] += sky_peek((const double*)(xY + (Ws - 1 - t / 6) * 6 + t % 6));
            __syncthreads();
            if (tid < 64) sky_pivot(slot(nA, nA), Gk.val + (s_rbase[nA] + nA) * 36, Gk.dinv + nA * 36, Li_of(nA), s_failp, tid);
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
                for (int t = lane; t < 6 * Ws; t += 64) s_y[(nA + t / 6) * 6 + t % 6] = sky_peek((const double*)(xX + (Ws - 1 - t / 6) * 6 + t % 6));
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
            const int pa = Gk.pos[t / 6];
            if (pa >= 0 && pa < own_to) ((SV_GLB double*)D.dp)[t] = 0.0;
        }
        return;
    }
    for (int t = tid; t < D.n; t += nt) {
        const int a = t / 6, c = t - 6 * a, pa = Gk.pos[a];
        if (pa >= 0 && pa < own_to) ((SV_GLB double*)D.dp)[t] = s_y[pa * 6 + c];
    }
}

// Backward substitution of the eliminated columns of every job, once the separator unknowns are known (D.dp of the separator slots):
// one wave per job, the right-hand side in LDS, the factor rows of the next four columns in flight (sky_backward_narrow).
__global__ __launch_bounds__(64) void k_seg_backward(BaDev D, const SegJobDev* __restrict__ jobs, double* __restrict__ out) {
    if (D.ctl->phase != 1) return;
    const SegJobDev J = jobs[blockIdx.x];
    const SkyDev& K = J.K;
    const SkyG Gk(K);
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    const int lane = threadIdx.x, nP = K.nP, nC = J.nC;
    SV_LDS double* const s_y = (SV_LDS double*)s_dyn;
    SV_LDS int* const s_coloff = (SV_LDS int*)(s_y + 6 * nP);
    SV_LDS int* const s_rows = s_coloff + (nP + 1);
    SV_LDS int* const s_base = s_rows + K.ncr;
    const bool failed = D.ctl->solve_failed != 0;  // (this rank's jobs or its separator solve: the trial is rejected anyway)
    for (int t = lane; t < 6 * nP; t += 64) s_y[t] = t < 6 * nC ? Gk.y[t] : ((const SV_GLB double*)D.dp)[Gk.order[t / 6] * 6 + t % 6];
    for (int t = lane; t <= nP; t += 64) s_coloff[t] = Gk.coloff[t];
    for (int t = lane; t < K.ncr; t += 64) {
        s_rows[t] = Gk.colrows[t];
        s_base[t] = Gk.colbase[t];
    }
    wave_lds_fence();
    if (!failed && nC > 0) sky_backward_narrow(K, s_y, s_coloff, s_rows, s_base, lane, nC - 1, 0);
    wave_lds_fence();
    for (int t = lane; t < 6 * nC; t += 64) ((SV_GLB double*)out)[Gk.order[t / 6] * 6 + t % 6] = failed ? 0.0 : s_y[t];
}
// out[separator slots] = D.dp (rank 0 of a sharded solve puts the separator unknowns into the exchanged solution vector)
__global__ __launch_bounds__(256) void k_seg_copy_sep(BaDev D, SkyDev K, double* __restrict__ out) {
    if (D.ctl->phase != 1) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 6 * K.nP) {
        const int a = K.order[t / 6];
        out[a * 6 + t % 6] = D.dp[a * 6 + t % 6];
    }
}

struct SkyPlan {
    SkyDev dev[2];     // [1] only for the two-sided elimination; [0] = the separator system of a segmented elimination
    SkyTwist twist;
    int epoch = 0;
    // segmented elimination (see SegJobDev): job descriptors of THIS rank (device array), the assembly / gather maps, the exchange buffers
    int seg = 0, seg_jobs_total = 0, seg_jobs_local = 0, seg_nsep = 0, seg_world = 1, seg_rank = 0;
    const SegJobDev* seg_jobs = nullptr;
    const int2* seg_blkmap = nullptr;
    const int *seg_yoff = nullptr, *seg_gb_off = nullptr, *seg_gb_src = nullptr, *seg_gy_off = nullptr, *seg_gy_src = nullptr;
    double *seg_arena = nullptr, *seg_xch = nullptr, *seg_dpx = nullptr;
    size_t seg_xch_doubles = 0, seg_lds_job = 0, seg_lds_back = 0;
    int seg_NB = 0, seg_n = 0, seg_cuts = 0, seg_longest = 0, plan_rows = 0, plan_width = 0;
    // roles of a segmented plan, host side (the keyframe-segment exchange of a sharded solve, svgpu_ba.hip): slot -> job (-1 = separator
    // row), job -> owning rank, the kept blocks whose two slots are both separator rows
    std::vector<int> seg_job_of, seg_owner, seg_sep_blocks;
    // the block pattern the plan was made for: an unchanged pattern (the same window / map optimised again) keeps plan and device arrays
    unsigned long long pattern_hash = 0;
    int pattern_nP = 0, pattern_rank = 0, pattern_world = 0, pattern_env = 0;
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

// -------------------------------------------------------------------------------------------------- segmented elimination: host planner
struct HostSeg {
    std::vector<HostSky> job;             // plan of [piece in elimination order, its adjacent separator rows]
    std::vector<std::vector<int>> order;  // per job: position -> slot
    std::vector<int> nC, ns, owner;       // eliminated columns, separator rows, owning rank
    HostSky sep;                          // the separator system
    std::vector<int> sep_order;           // position -> slot
    std::vector<size_t> xch_off;          // per job: its exchange block, in doubles (a multiple of 36)
    size_t xch_doubles = 0;
    std::vector<int> gb_off, gb_src, gy_off, gy_src;  // gather lists: separator block -> (exchange block | transposed << 30), separator row -> exchange offset
    int ncuts = 0, max_nC = 0, nsep_rows = 0;
    bool ok = false;
};
static size_t band_lds_bytes(const HostSky& h) {
    const size_t wn = (size_t)h.max_m + 1;
    return (wn * wn * 36 + (size_t)h.max_m * 36 + (size_t)6 * h.nP) * sizeof(double) + 4 * ((size_t)3 * h.nP + 1 + 2 * h.colrows.size());
}

// `ncuts` vertex separators in the RCM order (each a run of consecutive positions, as narrow as the band allows near its target), the
// connected pieces between them as jobs, the separator system with the fill the jobs leave on it.  G.ok only when every job fits the
// banded kernel (k_sky_band in job mode) and the separator system has a plan.
// cuts_only: stop once the pieces are known (G.max_nC, G.nsep_rows, the job count in G.nC.size()) -- what the cost model needs.
// ends_half: the first and the last stretch of positions are HALF as long as the ones between two cuts.  The RCM order of a loop (config 5:
// a ring of keyframes) walks both arcs away from its start at once, so a stretch between two cuts falls apart into two pieces (one per arc)
// of half its length, while the stretch before the first cut (the arc through the start vertex) and the one behind the last stay whole: with
// equal stretches those two jobs were twice as long as the other twelve (58 columns against 29) and set the time of the whole job launch.
static void plan_segments(int nP, const std::vector<int>& order, const std::vector<std::vector<int>>& adj, const std::vector<int2>& blk_ab, int ncuts, int world,
                          size_t max_bytes, HostSeg& G, bool cuts_only = false, bool ends_half = false) {
    G = HostSeg();
    if (ncuts < 1 || nP < 4) return;
    std::vector<int> pos0(nP);
    for (int i = 0; i < nP; ++i) pos0[order[i]] = i;
    // reach[c] = last position coupled to a position before c: a cut that starts at c has to extend that far
    std::vector<int> maxi_of_lo(nP, -1), reach(nP + 1, -1);
    for (int i = 0; i < nP; ++i) {
        int f = i;
        for (int v : adj[order[i]]) f = std::min(f, pos0[v]);
        maxi_of_lo[f] = std::max(maxi_of_lo[f], i);
    }
    for (int c = 1; c <= nP; ++c) reach[c] = std::max(reach[c - 1], maxi_of_lo[c - 1]);
    auto width = [&](int c) { return std::max(0, reach[c] - c + 1); };
    std::vector<char> is_sep(nP, 0);  // by slot
    int prev_end = 0;
    const int seg = ends_half ? nP / ncuts : nP / (ncuts + 1);
    for (int j = 1; j <= ncuts; ++j) {
        const int target = ends_half ? (int)((long long)(2 * j - 1) * nP / (2 * ncuts)) : (int)((long long)j * nP / (ncuts + 1)), win = std::max(1, seg / 4);
        int best = -1, bw = 1 << 30, boff = 1 << 30;
        for (int c = std::max(prev_end + 1, target - win); c <= std::min(nP - 2, target + win); ++c) {
            const int w = width(c);
            if (c + w >= nP) continue;  // nothing would be left behind it
            const int off = std::abs(c + w / 2 - target);
            if (w < bw || (w == bw && off < boff)) best = c, bw = w, boff = off;
        }
        if (best < 0) continue;
        for (int i = best; i < best + bw; ++i) is_sep[order[i]] = 1;
        prev_end = best + bw;
        ++G.ncuts;
    }
    if (G.ncuts == 0) return;
    // connected pieces of the graph without the separator vertices, members in RCM position order
    std::vector<int> comp(nP, -1);
    std::vector<std::vector<int>> members;
    for (int i = 0; i < nP; ++i) {
        const int s0 = order[i];
        if (is_sep[s0] || comp[s0] >= 0) continue;
        const int id = (int)members.size();
        members.emplace_back();
        std::vector<int> stack{s0};
        comp[s0] = id;
        while (!stack.empty()) {
            const int u = stack.back();
            stack.pop_back();
            members[id].push_back(u);
            for (int v : adj[u])
                if (!is_sep[v] && comp[v] < 0) {
                    comp[v] = id;
                    stack.push_back(v);
                }
        }
        std::sort(members[id].begin(), members[id].end(), [&](int a, int b) { return pos0[a] < pos0[b]; });
    }
    const int nj = (int)members.size();
    if (nj < 2) return;
    for (int u = 0; u < nP; ++u) G.nsep_rows += is_sep[u] ? 1 : 0;
    if (cuts_only) {
        G.nC.resize(nj);
        for (int q = 0; q < nj; ++q) {
            G.nC[q] = (int)members[q].size();
            G.max_nC = std::max(G.max_nC, G.nC[q]);
        }
        G.ok = true;
        return;
    }
    G.job.resize(nj);
    G.order.resize(nj);
    G.nC.resize(nj);
    G.ns.resize(nj);
    G.owner.assign(nj, 0);
    G.xch_off.resize(nj);
    std::vector<std::vector<int>> adj_sep(nP);  // separator graph: original couplings + the clique every job leaves on its separator rows
    for (int u = 0; u < nP; ++u)
        if (is_sep[u])
            for (int v : adj[u])
                if (is_sep[v]) adj_sep[u].push_back(v);
    std::vector<int> jpos(nP, -1);
    for (int q = 0; q < nj; ++q) {
        const std::vector<int>& M = members[q];
        const int n = (int)M.size();
        std::vector<int> A;
        for (int u : M)
            for (int v : adj[u])
                if (is_sep[v]) A.push_back(v);
        std::sort(A.begin(), A.end());
        A.erase(std::unique(A.begin(), A.end()), A.end());
        bool before = false, after = false;
        for (int v : A) {
            before = before || pos0[v] < pos0[M.front()];
            after = after || pos0[v] > pos0[M.back()];
        }
        // elimination order of the piece: it must END at its separators.  Separators on both sides: from the median of the piece
        // outwards, alternating (a chain folded in the middle: twice the bandwidth, no border rows); one side: towards it.
        std::vector<int>& ord = G.order[q];
        ord.reserve(n + A.size());
        if (before && after) {
            int l = (n - 1) / 2, r = l + 1;
            ord.push_back(M[l--]);
            while (l >= 0 || r < n) {
                if (r < n) ord.push_back(M[r++]);
                if (l >= 0) ord.push_back(M[l--]);
            }
        }
        else if (before)
            for (int k = n - 1; k >= 0; --k) ord.push_back(M[k]);
        else
            for (int k = 0; k < n; ++k) ord.push_back(M[k]);
        // separator rows behind it, those coupled deepest into the piece first (their envelopes then start in non-decreasing columns)
        for (int k = 0; k < n; ++k) jpos[ord[k]] = k;
        std::vector<std::pair<std::pair<int, int>, int>> key;
        for (int v : A) {
            int f = n;
            for (int w : adj[v])
                if (jpos[w] >= 0) f = std::min(f, jpos[w]);
            key.push_back({{f, pos0[v]}, v});
        }
        for (int k = 0; k < n; ++k) jpos[ord[k]] = -1;
        std::sort(key.begin(), key.end());
        for (const auto& kv : key) ord.push_back(kv.second);
        G.nC[q] = n;
        G.ns[q] = (int)A.size();
        G.max_nC = std::max(G.max_nC, n);
        host_sky(nP, ord, adj, blk_ab, n, true, max_bytes, G.job[q]);
        const HostSky& h = G.job[q];
        if (!h.ok || !h.band || h.max_m < 1 || h.max_m > SKY_BAND_W || (int)A.size() > h.max_m + 1 || band_lds_bytes(h) > 150 * 1024) return;
        for (size_t a = 0; a < A.size(); ++a)
            for (size_t b = 0; b < A.size(); ++b)
                if (a != b) adj_sep[A[a]].push_back(A[b]);
        G.xch_off[q] = G.xch_doubles;
        const size_t sz = (size_t)A.size() * (A.size() + 1) / 2 * 36 + 6 * A.size();
        G.xch_doubles += (sz + 35) / 36 * 36;
    }
    // the separator system in an RCM order of its OWN graph (the cuts of a loop are a ring of half-cuts: taken in the position order of
    // the original band every row would reach back over a whole cut, twice the width the folded ring has)
    {
        std::vector<int> sl, loc(nP, -1);
        for (int i = 0; i < nP; ++i)
            if (is_sep[order[i]]) {
                loc[order[i]] = (int)sl.size();
                sl.push_back(order[i]);
            }
        std::vector<std::vector<int>> a2(sl.size());
        for (size_t k = 0; k < sl.size(); ++k) {
            for (int v : adj_sep[sl[k]]) a2[k].push_back(loc[v]);
            std::sort(a2[k].begin(), a2[k].end());
            a2[k].erase(std::unique(a2[k].begin(), a2[k].end()), a2[k].end());
        }
        for (int k : rcm_order((int)sl.size(), a2)) G.sep_order.push_back(sl[k]);
    }
    const int nsep = (int)G.sep_order.size();
    if (nsep > 0) {
        host_sky(nP, G.sep_order, adj_sep, blk_ab, nsep, false, max_bytes, G.sep);
        if (!G.sep.ok) return;
        std::vector<std::vector<int>> gb(G.sep.nblocks), gy(nsep);
        for (int q = 0; q < nj; ++q) {
            const std::vector<int>& ord = G.order[q];
            const int nC = G.nC[q], ns = G.ns[q];
            const size_t b0 = G.xch_off[q] / 36, y0 = G.xch_off[q] + (size_t)ns * (ns + 1) / 2 * 36;
            for (int r = 0; r < ns; ++r) {
                const int pr = G.sep.pos[ord[nC + r]];
                gy[pr].push_back((int)(y0 + 6 * (size_t)r));
                for (int c = 0; c <= r; ++c) {
                    const int pc = G.sep.pos[ord[nC + c]];
                    const int row = std::max(pr, pc), col = std::min(pr, pc);
                    if (col < G.sep.first[row]) return;  // cannot happen: the clique is part of the separator graph
                    gb[G.sep.rowoff[row] + col - G.sep.first[row]].push_back((int)(b0 + (size_t)r * (r + 1) / 2 + c) | ((pr < pc ? 1 : 0) << 30));
                }
            }
        }
        G.gb_off.assign(1, 0);
        for (const auto& v : gb) {
            G.gb_src.insert(G.gb_src.end(), v.begin(), v.end());
            G.gb_off.push_back((int)G.gb_src.size());
        }
        G.gy_off.assign(1, 0);
        for (const auto& v : gy) {
            G.gy_src.insert(G.gy_src.end(), v.begin(), v.end());
            G.gy_off.push_back((int)G.gy_src.size());
        }
    }
    // owners: longest jobs first onto the least loaded rank (same on every rank: the plan is a function of the block pattern)
    std::vector<int> idx(nj);
    for (int q = 0; q < nj; ++q) idx[q] = q;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return G.nC[a] > G.nC[b]; });
    std::vector<long long> load(std::max(world, 1), 0);
    for (int q : idx) {
        int r = 0;
        for (int k = 1; k < (int)load.size(); ++k)
            if (load[k] < load[r]) r = k;
        G.owner[q] = r;
        load[r] += G.nC[q];
    }
    G.ok = true;
}

// Where everything of a segmented elimination lives inside ONE value arena (doubles; every block array starts at a multiple of 36), and the
// two maps of the assembly: kept block k -> (arena block, transposed), slot -> arena offset of its right-hand side entry.
struct SegLayout {
    std::vector<size_t> job_val, job_dinv, job_y;
    size_t sep_val = 0, sep_dinv = 0, sep_y = 0, xch = 0, dpx = 0, total = 0;
    std::vector<int2> blkmap;
    std::vector<int> yoff;
    bool ok = false;
};
static void seg_layout(int nP, const std::vector<int2>& blk_ab, const HostSeg& G, SegLayout& L) {
    L = SegLayout();
    const int nj = (int)G.job.size();
    size_t at = 0;
    auto take = [&](size_t doubles) {
        const size_t r = at;
        at += (doubles + 35) / 36 * 36;
        return r;
    };
    L.job_val.resize(nj), L.job_dinv.resize(nj), L.job_y.resize(nj);
    for (int q = 0; q < nj; ++q) {
        L.job_val[q] = take(G.job[q].nblocks * 36);
        L.job_dinv[q] = take((size_t)G.job[q].nP * 36);
        L.job_y[q] = take((size_t)G.job[q].nP * 6);
    }
    L.sep_val = take(G.sep.nblocks * 36);
    L.sep_dinv = take((size_t)G.sep.nP * 36);
    L.sep_y = take((size_t)G.sep.nP * 6);
    L.xch = take(G.xch_doubles);
    L.dpx = take((size_t)nP * 6);
    L.total = at;
    // which job holds a slot (separator slots: -1)
    std::vector<int> job_of(nP, -1);
    for (int q = 0; q < nj; ++q)
        for (int k = 0; k < G.nC[q]; ++k) job_of[G.order[q][k]] = q;
    L.yoff.assign(nP, -1);
    for (int a = 0; a < nP; ++a) {
        const int q = job_of[a];
        if (q >= 0) L.yoff[a] = (int)(L.job_y[q] + 6 * (size_t)G.job[q].pos[a]);
        else if (!G.sep.pos.empty() && G.sep.pos[a] >= 0) L.yoff[a] = (int)(L.sep_y + 6 * (size_t)G.sep.pos[a]);
        else return;  // a slot nobody owns
    }
    L.blkmap.resize(blk_ab.size());
    for (size_t k = 0; k < blk_ab.size(); ++k) {
        const int qa = job_of[blk_ab[k].x], qb = job_of[blk_ab[k].y];
        int2 m;
        m.x = -1, m.y = 0;
        if (qa < 0 && qb < 0) {
            m = G.sep.blkmap[k];
            if (m.x < 0) return;
            m.x += (int)(L.sep_val / 36);
        }
        else {
            const int q = qa >= 0 ? qa : qb;
            if (qa >= 0 && qb >= 0 && qa != qb) return;  // a block across two pieces: the cuts are not separators
            m = G.job[q].blkmap[k];
            if (m.x < 0) return;
            m.x += (int)(L.job_val[q] / 36);
        }
        L.blkmap[k] = m;
    }
    if (L.total >= ((size_t)1 << 31)) return;
    L.ok = true;
}

// ---- host arithmetic on a plan: used ONLY by svgpu_selftest_segmented_solve (the planner's self-test; the bundle adjusters never call it)
static bool host_chol6(const double* A, double* Lo, double* Li) {  // lower Cholesky factor and its inverse
    for (int i = 0; i < 36; ++i) Lo[i] = 0.0, Li[i] = 0.0;
    for (int c = 0; c < 6; ++c) {
        double d = A[c * 6 + c];
        for (int k = 0; k < c; ++k) d -= Lo[c * 6 + k] * Lo[c * 6 + k];
        if (!(d > 0.0)) return false;
        const double sd = std::sqrt(d);
        Lo[c * 6 + c] = sd;
        for (int r = c + 1; r < 6; ++r) {
            double v = A[r * 6 + c];
            for (int k = 0; k < c; ++k) v -= Lo[r * 6 + k] * Lo[c * 6 + k];
            Lo[r * 6 + c] = v / sd;
        }
    }
    for (int c = 0; c < 6; ++c)
        for (int r = c; r < 6; ++r) {
            double v = r == c ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) v -= Lo[r * 6 + k] * Li[k * 6 + c];
            Li[r * 6 + c] = v / Lo[r * 6 + r];
        }
    return true;
}
static bool host_env_eliminate(const HostSky& h, double* val, double* dinv, double* y, int ncol) {
    for (int j = 0; j < ncol; ++j) {
        double* Dj = val + (size_t)h.diag[j] * 36;
        double Lo[36], Li[36];
        if (!host_chol6(Dj, Lo, Li)) return false;
        for (int e = 0; e < 36; ++e) Dj[e] = Lo[e], dinv[(size_t)j * 36 + e] = Li[e];
        const int c0 = h.coloff[j], m = h.coloff[j + 1] - c0;
        for (int r = 0; r < m; ++r) {  // L_ij = S_ij L_jj^-T
            double* B = val + (size_t)(h.colbase[c0 + r] + j) * 36;
            double o[36];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    double v = 0.0;
                    for (int c = 0; c <= b; ++c) v += B[a * 6 + c] * Li[b * 6 + c];
                    o[a * 6 + b] = v;
                }
            for (int e = 0; e < 36; ++e) B[e] = o[e];
        }
        for (int p = 0; p < m; ++p)
            for (int q = 0; q <= p; ++q) {
                const double* Lp = val + (size_t)(h.colbase[c0 + p] + j) * 36;
                const double* Lq = val + (size_t)(h.colbase[c0 + q] + j) * 36;
                double* Dst = val + (size_t)(h.colbase[c0 + p] + h.colrows[c0 + q]) * 36;
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) {
                        double v = 0.0;
                        for (int c = 0; c < 6; ++c) v += Lp[a * 6 + c] * Lq[b * 6 + c];
                        Dst[a * 6 + b] -= v;
                    }
            }
        double z[6];
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
            for (int c = 0; c <= r; ++c) v += Li[r * 6 + c] * y[j * 6 + c];
            z[r] = v;
        }
        for (int r = 0; r < 6; ++r) y[j * 6 + r] = z[r];
        for (int r = 0; r < m; ++r) {
            const double* B = val + (size_t)(h.colbase[c0 + r] + j) * 36;
            const int i = h.colrows[c0 + r];
            for (int a = 0; a < 6; ++a) {
                double u = 0.0;
                for (int c = 0; c < 6; ++c) u += B[a * 6 + c] * z[c];
                y[i * 6 + a] -= u;
            }
        }
    }
    return true;
}
static void host_env_backward(const HostSky& h, const double* val, const double* dinv, double* y, int jhi) {
    for (int j = jhi; j >= 0; --j) {
        const int c0 = h.coloff[j], m = h.coloff[j + 1] - c0;
        double w[6];
        for (int a = 0; a < 6; ++a) w[a] = y[j * 6 + a];
        for (int r = 0; r < m; ++r) {
            const double* B = val + (size_t)(h.colbase[c0 + r] + j) * 36;
            const double* xi = y + (size_t)h.colrows[c0 + r] * 6;
            for (int a = 0; a < 6; ++a)
                for (int c = 0; c < 6; ++c) w[a] -= B[c * 6 + a] * xi[c];
        }
        const double* Li = dinv + (size_t)j * 36;
        for (int a = 0; a < 6; ++a) {
            double x = 0.0;
            for (int c = a; c < 6; ++c) x += Li[c * 6 + a] * w[c];
            y[j * 6 + a] = x;
        }
    }
}

// Plans the envelope factorisation of the reduced system whose kept upper blocks are blk_ab (a <= b, free-pose slots).  *usable = false
// (and nothing else changes) when the envelope would exceed max_bytes or a column has more than SKY_MAXM rows: the caller keeps the PCG.
// Uploads a segmented plan: every job's and the separator system's index arrays, the assembly / gather maps and the descriptors of the
// jobs THIS rank owns into the integer arena, and lays the value arena out (SegLayout).
// slot -> job (-1: separator row) and the kept blocks between two separator rows
static void seg_roles(int nP, const std::vector<int2>& blk_ab, const HostSeg& G, std::vector<int>& job_of, std::vector<int>& sep_blocks) {
    job_of.assign(nP, -1);
    for (size_t q = 0; q < G.job.size(); ++q)
        for (int k = 0; k < G.nC[q]; ++k) job_of[G.order[q][k]] = (int)q;
    sep_blocks.clear();
    for (size_t k = 0; k < blk_ab.size(); ++k)
        if (job_of[blk_ab[k].x] < 0 && job_of[blk_ab[k].y] < 0) sep_blocks.push_back((int)k);
}

static int seg_upload(svgpu_ctx* ctx, hipStream_t s, SkyPlan* P, int nP, const std::vector<int2>& blk_ab, const HostSeg& G, const SegLayout& LY, int rank, int world) {
    const int nj = (int)G.job.size(), nsep = (int)G.sep_order.size();
    std::vector<int> host;
    auto put = [&](const int* p, size_t n) {
        if (host.size() & 1) host.push_back(0);
        const size_t at = host.size();
        host.insert(host.end(), p, p + n);
        return at;
    };
    struct Off {
        size_t pos, first, rowoff, coloff, colrows, colbase, diag, order;
    };
    std::vector<Off> off(nj + 1);
    auto put_plan = [&](const HostSky& h, const std::vector<int>& ord, Off& o) {
        o.pos = put(h.pos.data(), h.pos.size());
        o.first = put(h.first.data(), h.nP);
        o.rowoff = put(h.rowoff.data(), h.nP + 1);
        o.coloff = put(h.coloff.data(), h.nP + 1);
        o.colrows = put(h.colrows.data(), h.colrows.size());
        o.colbase = put(h.colbase.data(), h.colbase.size());
        o.diag = put(h.diag.data(), h.nP);
        o.order = put(ord.data(), ord.size());
    };
    for (int q = 0; q < nj; ++q) put_plan(G.job[q], G.order[q], off[q]);
    if (nsep > 0) put_plan(G.sep, G.sep_order, off[nj]);
    const size_t o_blkmap = put(reinterpret_cast<const int*>(LY.blkmap.data()), LY.blkmap.size() * 2);
    const size_t o_yoff = put(LY.yoff.data(), LY.yoff.size());
    const int zero = 0;
    const size_t o_gb_off = nsep > 0 ? put(G.gb_off.data(), G.gb_off.size()) : put(&zero, 1);
    const size_t o_gb_src = nsep > 0 ? put(G.gb_src.data(), G.gb_src.size()) : put(&zero, 1);
    const size_t o_gy_off = nsep > 0 ? put(G.gy_off.data(), G.gy_off.size()) : put(&zero, 1);
    const size_t o_gy_src = nsep > 0 ? put(G.gy_src.data(), G.gy_src.size()) : put(&zero, 1);
    int n_local = 0;
    for (int q = 0; q < nj; ++q) n_local += G.owner[q] == rank;
    if (host.size() & 1) host.push_back(0);
    const size_t o_jobs = host.size();
    const size_t n_int = o_jobs + (size_t)n_local * (sizeof(SegJobDev) / 4) + 16;
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
    const size_t val_need = LY.total * sizeof(double);
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
    int* const di = (int*)P->d_int;
    double* const dv = (double*)P->d_val;
    auto dev_of = [&](const HostSky& h, const Off& o, size_t val, size_t dinv, size_t y) {
        SkyDev K;
        K.nP = h.nP;
        K.NB = (int)blk_ab.size();
        K.pos = di + o.pos;
        K.order = di + o.order;
        K.first = di + o.first;
        K.rowoff = di + o.rowoff;
        K.coloff = di + o.coloff;
        K.colrows = di + o.colrows;
        K.colbase = di + o.colbase;
        K.diag = di + o.diag;
        K.max_m = h.max_m;
        K.ncr = (int)h.colrows.size();
        K.band = h.band ? 1 : 0;
        K.val = dv + val;
        K.dinv = dv + dinv;
        K.y = dv + y;
        K.nblocks = h.nblocks;
        return K;
    };
    std::vector<SegJobDev> jobs;
    size_t lds_job = 0, lds_back = 0;
    for (int q = 0; q < nj; ++q) {
        if (G.owner[q] != rank) continue;
        SegJobDev J;
        J.K = dev_of(G.job[q], off[q], LY.job_val[q], LY.job_dinv[q], LY.job_y[q]);
        J.nC = G.nC[q];
        J.ns = G.ns[q];
        J.xch = dv + LY.xch + G.xch_off[q];
        jobs.push_back(J);
        lds_job = std::max(lds_job, band_lds_bytes(G.job[q]));
        lds_back = std::max(lds_back, (size_t)6 * G.job[q].nP * sizeof(double) + 4 * ((size_t)G.job[q].nP + 1 + 2 * G.job[q].colrows.size()));
    }
    if (!jobs.empty()) {
        host.resize(o_jobs + jobs.size() * (sizeof(SegJobDev) / 4));
        memcpy(host.data() + o_jobs, jobs.data(), jobs.size() * sizeof(SegJobDev));
    }
    SV_HIP(ctx, hipMemcpyAsync(P->d_int, host.data(), host.size() * 4, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipStreamSynchronize(s));  // `host` is a pageable temporary
    for (int q = 0; q < 2; ++q) P->dev[q] = SkyDev();
    if (nsep > 0) P->dev[0] = dev_of(G.sep, off[nj], LY.sep_val, LY.sep_dinv, LY.sep_y);
    P->twist = SkyTwist();
    P->seg = 1;
    P->seg_jobs_total = nj;
    P->seg_jobs_local = (int)jobs.size();
    P->seg_nsep = nsep;
    P->seg_world = world;
    P->seg_rank = rank;
    P->seg_jobs = reinterpret_cast<const SegJobDev*>(di + o_jobs);
    P->seg_blkmap = reinterpret_cast<const int2*>(di + o_blkmap);
    P->seg_yoff = di + o_yoff;
    P->seg_gb_off = di + o_gb_off;
    P->seg_gb_src = di + o_gb_src;
    P->seg_gy_off = di + o_gy_off;
    P->seg_gy_src = di + o_gy_src;
    P->seg_arena = dv;
    P->seg_xch = dv + LY.xch;
    P->seg_dpx = dv + LY.dpx;
    P->seg_xch_doubles = G.xch_doubles;
    P->seg_lds_job = lds_job;
    P->seg_lds_back = lds_back;
    P->seg_NB = (int)blk_ab.size();
    P->seg_n = 6 * nP;
    seg_roles(nP, blk_ab, G, P->seg_job_of, P->seg_sep_blocks);
    P->seg_owner = G.owner;
    P->val_used = LY.total * sizeof(double);
    P->epoch = 0;
    P->usable = true;
    return SVGPU_OK;
}

// The segmented plan sv_sky_plan would choose for this block pattern (shared with the self-test): candidates of 2..8 cuts, cheapest by a
// column-count model -- the longest job at ~3 us per column (jobs run side by side) + the separator system at ~3 us (banded kernel) or
// ~6.5 us (general kernel) per column; `want` > 0 forces that many cuts.
static bool choose_segments(int nP, const std::vector<int>& order, const std::vector<std::vector<int>>& adj, const std::vector<int2>& blk_ab, int want, int world,
                            size_t max_bytes, HostSeg& G, SegLayout& LY) {
    G = HostSeg();
    // candidates ranked by the column-count model on their cuts alone (cheap), then planned in full in that order until one holds
    std::vector<std::pair<double, int>> cand;  // (model cost, cuts * 2 + ends_half)
    for (int nc = want > 0 ? want : 2; nc <= (want > 0 ? want : 8); ++nc)
        for (int eh = 0; eh < 2; ++eh) {
            HostSeg c;
            plan_segments(nP, order, adj, blk_ab, nc, world, max_bytes, c, true, eh != 0);
            if (!c.ok || (int)c.nC.size() < world) continue;
            cand.push_back({3.0 * c.max_nC + 3.0 * c.nsep_rows + 15.0, 2 * nc + eh});
        }
    std::sort(cand.begin(), cand.end());
    double best = 1e30;
    for (const auto& cn : cand) {
        HostSeg c;
        plan_segments(nP, order, adj, blk_ab, cn.second >> 1, world, max_bytes, c, false, (cn.second & 1) != 0);
        if (!c.ok) continue;
        const bool sep_band = c.sep.nP == 0 || (c.sep.band && c.sep.max_m <= SKY_BAND_W && band_lds_bytes(c.sep) <= 150 * 1024);
        const double cost = 3.0 * c.max_nC + (sep_band ? 3.0 : 6.5) * c.sep.nP + 15.0;
        if (cost < best) {
            best = cost;
            G = std::move(c);
        }
        if (sep_band) break;  // the model's order already prefers it; a plan whose separator system needs the general kernel keeps looking
    }
    if (!G.ok) return false;
    if (want <= 0 && world == 1 && best > 0.8 * 3.0 * nP) return false;  // not worth three launches instead of one
    seg_layout(nP, blk_ab, G, LY);
    return LY.ok;
}

// The front half of sv_sky_plan, free of the device: keyframe graph, RCM order, the one-piece plan H0 and -- when the band is long enough
// and nothing switches it off -- the segmented plan.  Shared with the keyframe-segment partitioner of a sharded solve, which has to cut
// the graph exactly where the solve will.
static bool plan_front(int nP, const std::vector<int2>& blk_ab, int world, size_t max_bytes, std::vector<std::vector<int>>& adj, std::vector<int>& order,
                       HostSky& H0, HostSeg& G, SegLayout& LY) {
    adj.assign(nP, {});
    for (const int2& ab : blk_ab)
        if (ab.x != ab.y) {
            adj[ab.x].push_back(ab.y);
            adj[ab.y].push_back(ab.x);
        }
    order = rcm_order(nP, adj);
    host_sky(nP, order, adj, blk_ab, nP, false, max_bytes, H0);
    if (!H0.ok) return false;
    const char* ev = std::getenv("SVGPU_SKY_SEGMENTS");
    const int want = ev ? std::atoi(ev) : -1;
    if (want == 0 || std::getenv("SVGPU_SKY_ONE_SIDED") || !H0.band || H0.max_m < 1 || nP < 8 * (H0.max_m + 1)) return false;
    return choose_segments(nP, order, adj, blk_ab, want, world, max_bytes, G, LY);
}

// Keyframe-segment roles of a block pattern, host only: slot -> job (-1 = separator row) and job -> owning rank, as the segmented plan of a
// `world`-rank solve assigns them; false when that pattern gets no segmented plan.
bool sv_sky_partition_roles(int nP, const std::vector<int2>& blk_ab, int world, std::vector<int>& job_of, std::vector<int>& owner, int* ncuts, int* sep_blocks_n, long long* xch_doubles) {
    std::vector<std::vector<int>> adj;
    std::vector<int> order, sep_blocks;
    HostSky H0;
    HostSeg G;
    SegLayout LY;
    if (nP <= 0 || !plan_front(nP, blk_ab, world, (size_t)256 << 20, adj, order, H0, G, LY)) return false;
    seg_roles(nP, blk_ab, G, job_of, sep_blocks);
    owner = G.owner;
    if (ncuts) *ncuts = G.ncuts;
    if (sep_blocks_n) *sep_blocks_n = (int)sep_blocks.size();
    if (xch_doubles) *xch_doubles = (long long)G.xch_doubles;
    return true;
}

// roles of the context's current plan (null when it is not a segmented one)
bool sv_sky_current_roles(svgpu_ctx* ctx, const std::vector<int>** job_of, const std::vector<int>** owner, const std::vector<int>** sep_blocks) {
    const SkyPlan* P = (const SkyPlan*)ctx->ba_sky;
    if (!P || !P->usable || !P->seg) return false;
    *job_of = &P->seg_job_of, *owner = &P->seg_owner, *sep_blocks = &P->seg_sep_blocks;
    return true;
}

int sv_sky_plan(svgpu_ctx* ctx, hipStream_t s, int nP, const std::vector<int2>& blk_ab, size_t max_bytes, bool* usable, int rank, int world) {
    *usable = false;
    if (nP <= 0) return SVGPU_OK;
    unsigned long long hsh = 1469598103934665603ull;
    for (const int2& ab : blk_ab) {
        hsh = (hsh ^ (unsigned)ab.x) * 1099511628211ull;
        hsh = (hsh ^ (unsigned)ab.y) * 1099511628211ull;
    }
    const char* env_seg = std::getenv("SVGPU_SKY_SEGMENTS");
    const int env_code = (env_seg ? 1000 + std::atoi(env_seg) : 0) + (std::getenv("SVGPU_SKY_ONE_SIDED") ? 100000 : 0) + (std::getenv("SVGPU_SKY_NO_BAND") ? 200000 : 0);
    if (SkyPlan* Q = (SkyPlan*)ctx->ba_sky) {
        if (Q->usable && Q->pattern_hash == hsh && Q->pattern_nP == nP && Q->pattern_rank == rank && Q->pattern_world == world && Q->pattern_env == env_code
            && Q->seg_NB * (Q->seg ? 1 : 0) == (Q->seg ? (int)blk_ab.size() : 0) && (Q->seg || Q->dev[0].NB == (int)blk_ab.size())) {
            *usable = true;  // (the epoch of the hand-over flags keeps counting)
            return SVGPU_OK;
        }
        Q->usable = false;
    }
    struct Remember {  // records what the plan was made for when the function leaves with a usable plan
        svgpu_ctx* c;
        bool* ok;
        unsigned long long h;
        int nP, rank, world, env;
        ~Remember() {
            SkyPlan* Q = (SkyPlan*)c->ba_sky;
            if (Q && *ok) Q->pattern_hash = h, Q->pattern_nP = nP, Q->pattern_rank = rank, Q->pattern_world = world, Q->pattern_env = env;
        }
    } remember{ctx, usable, hsh, nP, rank, world, env_code};
    std::vector<std::vector<int>> adj;
    std::vector<int> order;  // position -> slot
    HostSky H[2];
    HostSeg G;
    SegLayout LY;
    const bool segmented = plan_front(nP, blk_ab, world, max_bytes, adj, order, H[0], G, LY);
    if (!H[0].ok) return SVGPU_OK;
    SkyPlan* P = (SkyPlan*)ctx->ba_sky;
    if (!P) {
        P = new SkyPlan();
        ctx->ba_sky = P;
    }
    P->seg = 0;
    // segmented elimination of a long band (the default there; SVGPU_SKY_SEGMENTS=<cuts> forces a cut count, 0 switches it off).  A sharded
    // solve distributes the jobs over its ranks -- every rank plans the same segments from the same (all-reduced) block pattern.
    if (segmented) {
        const int rs = seg_upload(ctx, s, P, nP, blk_ab, G, LY, rank, world);
        if (rs) return rs;
        P->seg_cuts = G.ncuts, P->seg_longest = G.max_nC, P->plan_rows = nP, P->plan_width = H[0].max_m;
        *usable = true;
        if (std::getenv("SVGPU_BA_TRACE")) {
            std::fprintf(stderr, "[ba]     envelope plan: %d block rows, segmented: %d cuts, %zu jobs (longest %d columns, %d on this rank), separator system %d rows%s\n", nP,
                         G.ncuts, G.job.size(), G.max_nC, P->seg_jobs_local, G.sep.nP, G.sep.band ? " (banded)" : "");
        }
        return SVGPU_OK;
    }
    // two-sided elimination of a long band: T | S | B with S = as many rows as the band is wide (then no block couples T and B)
    int nplans = 1, tw_m = 0, tw_W = 0, tw_nB = 0;
    // (the fallback when the segmented plan does not apply.  Its two workgroups wait for each other on flags, so both must be resident at
    //  once: any device with at least two compute units -- a device restricted below that keeps the one-sided sweep.)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    if (cus >= 2 && H[0].band && H[0].max_m >= 1 && nP >= 8 * (H[0].max_m + 1) && !std::getenv("SVGPU_SKY_ONE_SIDED")) {
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
    P->plan_rows = nP, P->plan_width = H[0].max_m;
    *usable = true;
    if (std::getenv("SVGPU_BA_TRACE"))
        std::fprintf(stderr, "[ba]     envelope plan: %d block rows, %zu blocks (%.1f MB), widest column %d rows%s%s\n", nP, H[0].nblocks + (nplans == 2 ? H[1].nblocks : 0),
                     (H[0].nblocks + (nplans == 2 ? H[1].nblocks : 0)) * 288.0 / 1048576.0, H[0].max_m, H[0].band ? ", banded" : "",
                     nplans == 2 ? ", two-sided elimination" : "");
    return SVGPU_OK;
}

int sv_sky_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    SkyPlan* P = (SkyPlan*)ctx->ba_sky;
    SvProfScope ps(ctx, s, "ba_solve");
    if (P->seg) {
        // segmented elimination: assemble -> the jobs of this rank, one workgroup each -> [exchange of what they leave on the separators] ->
        // separator system -> backward substitution of the jobs -> [exchange of the solution].  The exchanges are sums in which every rank
        // contributes zeros outside its own jobs: an all-gather through the all-reduce the solve already has.
        const bool multi = P->seg_world > 1 && ctx->ba_ar_fn;
        (void)hipMemsetAsync(P->seg_arena, 0, P->val_used, s);
        const size_t items = std::max((size_t)P->seg_NB * 36, (size_t)D.n);
        hipLaunchKernelGGL(k_seg_assemble, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, D, P->seg_blkmap, P->seg_yoff, P->seg_arena, P->seg_NB);
        if (P->seg_jobs_local > 0) {
            (void)sv_allow_dynamic_lds((const void*)k_sky_band, P->seg_lds_job);
            hipLaunchKernelGGL(k_sky_band, dim3(P->seg_jobs_local), dim3(SKY_BAND_THREADS), P->seg_lds_job, s, D, SkyDev(), SkyDev(), SkyTwist(), 0, P->seg_jobs);
        }
        if (multi && P->seg_xch_doubles > 0) {
            ctx->ba_xch[4] += 8 * (long long)P->seg_xch_doubles;
            ++ctx->ba_xch[7];
            if (ctx->ba_ar_fn(ctx->ba_ar_user, P->seg_xch, P->seg_xch_doubles, (void*)s) != 0) return sv_set_error(ctx, SVGPU_ERR_HIP, "all-reduce callback failed (separator exchange)");
        }
        if (P->seg_nsep > 0) {
            const SkyDev& K = P->dev[0];
            const size_t gitems = std::max(K.nblocks * 36, (size_t)K.nP * 6);
            hipLaunchKernelGGL(k_seg_gather, dim3((unsigned)((gitems + 255) / 256)), dim3(256), 0, s, D, K, P->seg_gb_off, P->seg_gb_src, P->seg_gy_off, P->seg_gy_src, P->seg_xch);
            const size_t wn = (size_t)K.max_m + 1;
            const size_t lds_b = (wn * wn * 36 + (size_t)K.max_m * 36 + (size_t)6 * K.nP) * sizeof(double) + 4 * ((size_t)3 * K.nP + 1 + 2 * (size_t)K.ncr);
            if (K.band && K.max_m <= SKY_BAND_W && lds_b <= 150 * 1024) {
                (void)sv_allow_dynamic_lds((const void*)k_sky_band, lds_b);
                hipLaunchKernelGGL(k_sky_band, dim3(1), dim3(SKY_BAND_THREADS), lds_b, s, D, K, SkyDev(), SkyTwist(), ++P->epoch, (const SegJobDev*)nullptr);
            }
            else {
                const size_t lds_g = ((size_t)K.max_m * 36 + (size_t)6 * K.nP) * sizeof(double) + 4 * ((size_t)2 * K.nP + 1 + 2 * (size_t)K.ncr);
                (void)sv_allow_dynamic_lds((const void*)k_sky_factor_solve, lds_g);
                hipLaunchKernelGGL(k_sky_factor_solve, dim3(1), dim3(K.max_m <= 21 ? 256 : SKY_THREADS), lds_g, s, D, K);
            }
        }
        double* const out = multi ? P->seg_dpx : D.dp;
        if (P->seg_jobs_local > 0) {
            (void)sv_allow_dynamic_lds((const void*)k_seg_backward, P->seg_lds_back);
            hipLaunchKernelGGL(k_seg_backward, dim3(P->seg_jobs_local), dim3(64), P->seg_lds_back, s, D, P->seg_jobs, out);
        }
        if (multi) {
            if (P->seg_rank == 0 && P->seg_nsep > 0)
                hipLaunchKernelGGL(k_seg_copy_sep, dim3((unsigned)((6 * P->dev[0].nP + 255) / 256)), dim3(256), 0, s, D, P->dev[0], P->seg_dpx);
            ctx->ba_xch[5] += 8 * (long long)P->seg_n;
            ++ctx->ba_xch[7];
            if (ctx->ba_ar_fn(ctx->ba_ar_user, P->seg_dpx, (size_t)P->seg_n, (void*)s) != 0) return sv_set_error(ctx, SVGPU_ERR_HIP, "all-reduce callback failed (solution exchange)");
            (void)hipMemcpyAsync(D.dp, P->seg_dpx, sizeof(double) * (size_t)P->seg_n, hipMemcpyDeviceToDevice, s);
        }
        return SVGPU_OK;
    }
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
            hipLaunchKernelGGL(k_sky_band, dim3(nplans), dim3(SKY_BAND_THREADS), lds, s, D, P->dev[0], P->dev[1], P->twist, ++P->epoch, (const SegJobDev*)nullptr);
            return SVGPU_OK;
        }
    }
    const size_t lds = ((size_t)K.max_m * 36 + (size_t)D.n) * sizeof(double) + 4 * ((size_t)2 * K.nP + 1 + 2 * (size_t)K.ncr);
    (void)sv_allow_dynamic_lds((const void*)k_sky_factor_solve, lds);
    hipLaunchKernelGGL(k_sky_factor_solve, dim3(1), dim3(K.max_m <= 21 ? 256 : SKY_THREADS), lds, s, D, K);
    return SVGPU_OK;
}

extern "C" int svgpu_ba_last_envelope_plan(svgpu_ctx* ctx, int* info) {
    if (!ctx || !info) return SVGPU_ERR_INVALID;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    const SkyPlan* P = (const SkyPlan*)ctx->ba_sky;
    if (!P || !P->usable) return SVGPU_OK;
    info[0] = P->seg ? 2 : (P->twist.on ? 1 : 0);
    info[1] = P->plan_rows, info[2] = P->plan_width;
    if (P->seg) info[3] = P->seg_cuts, info[4] = P->seg_jobs_total, info[5] = P->seg_jobs_local, info[6] = P->seg_nsep, info[7] = P->seg_longest;
    return SVGPU_OK;
}

// The planner's self-test (host arithmetic, no device needed; the bundle adjusters never call it): plans the segmented elimination of the
// block system exactly as sv_sky_plan does, then walks the SAME plan arrays, assembly maps and gather lists the kernels walk -- assemble,
// eliminate every job's columns, gather what they leave onto the separator system, solve it, substitute backwards -- and returns x.
// tests/test_sky_segments.py holds it against a dense solve: that pins the orders, the envelopes, the block maps and the transposition
// flags on the CPU, so that what is left to the GPU tests is the kernels' own mechanics.
extern "C" int svgpu_selftest_segmented_solve_rank(int nP, int NB, const int* blk_ab_in, const double* Sblk, const double* g, int cuts, int rank, int world,
                                                  svgpu_allreduce_fn allreduce, void* allreduce_user, double* x, int* info);
extern "C" int svgpu_selftest_segmented_solve(int nP, int NB, const int* blk_ab_in, const double* Sblk, const double* g, int cuts, int world, double* x, int* info) {
    return svgpu_selftest_segmented_solve_rank(nP, NB, blk_ab_in, Sblk, g, cuts, 0, world, nullptr, nullptr, x, info);
}
// One RANK of the distributed form (host arithmetic; `allreduce` sums a HOST buffer of doubles in place across the ranks -- the gloo test of
// tests/test_distributed_cpu.py): the rank eliminates only the jobs it owns, the exchange buffer and the solution cross ranks exactly as
// in sv_sky_solve (zeros outside the own jobs, rank 0 contributes the separator unknowns).  allreduce == NULL: every job is local.
extern "C" int svgpu_selftest_segmented_solve_rank(int nP, int NB, const int* blk_ab_in, const double* Sblk, const double* g, int cuts, int rank, int world,
                                                  svgpu_allreduce_fn allreduce, void* allreduce_user, double* x, int* info) {
    if (world < 1 || rank < 0 || rank >= world) return -1;
    if (nP <= 0 || NB <= 0 || !blk_ab_in || !Sblk || !g || !x || !info) return -1;
    std::vector<int2> blk_ab(NB);
    std::vector<std::vector<int>> adj(nP);
    for (int k = 0; k < NB; ++k) {
        blk_ab[k].x = blk_ab_in[2 * k], blk_ab[k].y = blk_ab_in[2 * k + 1];
        if (blk_ab[k].x < 0 || blk_ab[k].y >= nP || blk_ab[k].x > blk_ab[k].y) return -1;
        if (blk_ab[k].x != blk_ab[k].y) {
            adj[blk_ab[k].x].push_back(blk_ab[k].y);
            adj[blk_ab[k].y].push_back(blk_ab[k].x);
        }
    }
    const std::vector<int> order = rcm_order(nP, adj);
    HostSky H0;
    host_sky(nP, order, adj, blk_ab, nP, false, (size_t)256 << 20, H0);
    for (int k = 0; k < 8; ++k) info[k] = 0;
    if (!H0.ok) return 1;
    info[7] = H0.max_m;
    HostSeg G;
    SegLayout LY;
    if (!choose_segments(nP, order, adj, blk_ab, cuts, std::max(world, 1), (size_t)256 << 20, G, LY)) return 1;
    const int nj = (int)G.job.size(), nsep = (int)G.sep_order.size();
    info[0] = 1, info[1] = G.ncuts, info[2] = nj, info[3] = nsep, info[4] = G.max_nC, info[5] = G.sep.band ? 1 : 0;
    for (int q = 0; q < nj; ++q) info[6] = std::max(info[6], G.job[q].max_m);
    std::vector<double> A(LY.total, 0.0);
    for (int k = 0; k < NB; ++k) {  // k_seg_assemble
        const int2 m = LY.blkmap[k];
        for (int e = 0; e < 36; ++e) A[(size_t)m.x * 36 + (m.y ? (e % 6) * 6 + e / 6 : e)] = Sblk[(size_t)k * 36 + e];
    }
    for (int a = 0; a < nP; ++a)
        for (int c = 0; c < 6; ++c) A[(size_t)LY.yoff[a] + c] = g[a * 6 + c];
    int owned = 0;
    for (int q = 0; q < nj; ++q) {  // the jobs (k_sky_band in job mode): eliminate nC columns, export the trailing ns x ns blocks and y
        if (allreduce && G.owner[q] != rank) continue;
        ++owned;
        const HostSky& h = G.job[q];
        if (!host_env_eliminate(h, A.data() + LY.job_val[q], A.data() + LY.job_dinv[q], A.data() + LY.job_y[q], G.nC[q])) return 2;
        const int nC = G.nC[q], ns = G.ns[q];
        double* X = A.data() + LY.xch + G.xch_off[q];
        for (int r = 0; r < ns; ++r)
            for (int c = 0; c <= r; ++c) {
                const double* B = A.data() + LY.job_val[q] + (size_t)(h.rowoff[nC + r] + (nC + c) - h.first[nC + r]) * 36;
                for (int e = 0; e < 36; ++e) X[((size_t)r * (r + 1) / 2 + c) * 36 + e] = B[e];
            }
        if (ns > 0) {  // the diagonal blocks of the window are full symmetric blocks on the device; the host update above fills them the same way
            for (int r = 0; r < ns; ++r)
                for (int c = 0; c < 6; ++c) X[(size_t)ns * (ns + 1) / 2 * 36 + 6 * r + c] = A[LY.job_y[q] + (size_t)(nC + r) * 6 + c];
        }
    }
    if (allreduce && G.xch_doubles > 0 && allreduce(allreduce_user, A.data() + LY.xch, G.xch_doubles, nullptr) != 0) return -2;
    std::vector<double> sol((size_t)nP * 6, 0.0);
    if (nsep > 0) {  // k_seg_gather, then the separator solve
        for (size_t b = 0; b < G.sep.nblocks; ++b)
            for (int e = 0; e < 36; ++e) {
                double v = A[LY.sep_val + b * 36 + e];
                for (int qq = G.gb_off[b]; qq < G.gb_off[b + 1]; ++qq) {
                    const int src = G.gb_src[qq];
                    v += A[LY.xch + (size_t)(src & 0x3fffffff) * 36 + (((src >> 30) & 1) ? (e % 6) * 6 + e / 6 : e)];
                }
                A[LY.sep_val + b * 36 + e] = v;
            }
        for (int p = 0; p < nsep; ++p)
            for (int c = 0; c < 6; ++c) {
                double v = A[LY.sep_y + (size_t)p * 6 + c];
                for (int qq = G.gy_off[p]; qq < G.gy_off[p + 1]; ++qq) v += A[LY.xch + (size_t)G.gy_src[qq] + c];
                A[LY.sep_y + (size_t)p * 6 + c] = v;
            }
        if (!host_env_eliminate(G.sep, A.data() + LY.sep_val, A.data() + LY.sep_dinv, A.data() + LY.sep_y, nsep)) return 2;
        host_env_backward(G.sep, A.data() + LY.sep_val, A.data() + LY.sep_dinv, A.data() + LY.sep_y, nsep - 1);
        for (int p = 0; p < nsep; ++p)
            for (int c = 0; c < 6; ++c) sol[(size_t)G.sep_order[p] * 6 + c] = A[LY.sep_y + (size_t)p * 6 + c];
    }
    for (int q = 0; q < nj; ++q) {  // k_seg_backward
        if (allreduce && G.owner[q] != rank) continue;
        const HostSky& h = G.job[q];
        double* y = A.data() + LY.job_y[q];
        for (int r = 0; r < G.ns[q]; ++r)
            for (int c = 0; c < 6; ++c) y[(size_t)(G.nC[q] + r) * 6 + c] = sol[(size_t)G.order[q][G.nC[q] + r] * 6 + c];
        host_env_backward(h, A.data() + LY.job_val[q], A.data() + LY.job_dinv[q], y, G.nC[q] - 1);
        for (int k = 0; k < G.nC[q]; ++k)
            for (int c = 0; c < 6; ++c) sol[(size_t)G.order[q][k] * 6 + c] = y[(size_t)k * 6 + c];
    }
    if (allreduce) {  // the solution exchange: own columns + (rank 0) the separator unknowns, zeros elsewhere
        if (rank != 0)
            for (int p = 0; p < nsep; ++p)
                for (int c = 0; c < 6; ++c) sol[(size_t)G.sep_order[p] * 6 + c] = 0.0;
        if (allreduce(allreduce_user, sol.data(), sol.size(), nullptr) != 0) return -2;
    }
    info[5] = allreduce ? owned : info[5];  // (distributed form: the jobs this rank eliminated)
    for (size_t t = 0; t < sol.size(); ++t) x[t] = sol[t];
    return 0;
}
