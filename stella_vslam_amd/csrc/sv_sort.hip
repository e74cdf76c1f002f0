// Hand-written device sorts / scans (sv_sort.h).  No library sort is used anywhere in libsvgpu.
#include "sv_sort.h"
#include "svgpu_internal.h"

namespace {
// In-place exclusive scan of data[0 .. n) with the total written to data[n]: ONE workgroup, 16 consecutive elements per thread (a serial
// scan in registers), the 64 thread totals of a wave scanned with shuffles, the 16 wave totals by the first wave.  The loads of the next
// tile of 16 k elements are in flight while the current one is scanned, the wave totals are double-buffered and every thread carries the
// running total itself: two barriers per tile and no load latency on the critical path (a tile cost ~8 us with four barriers and the load
// in front of them: 106 us for the 200 k landmark counts of config 5; a Hillis-Steele scan over the 1024 totals in LDS took 142 us).
#define SCAN_PER_THREAD 16
__device__ __forceinline__ void scan_load_tile(const int* __restrict__ data, int i0, int n, int (&v)[SCAN_PER_THREAD]) {
    if (i0 + SCAN_PER_THREAD <= n) {
        const int4* p4 = reinterpret_cast<const int4*>(data + i0);  // i0 is a multiple of 16: aligned
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD / 4; ++k) {
            const int4 q = p4[k];
            v[4 * k] = q.x, v[4 * k + 1] = q.y, v[4 * k + 2] = q.z, v[4 * k + 3] = q.w;
        }
    }
    else {
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; ++k) v[k] = i0 + k < n ? data[i0 + k] : 0;
    }
}
__global__ __launch_bounds__(1024) void k_scan_i32(int* __restrict__ data, const int* __restrict__ n_dev, int n_host) {
    __shared__ int s_wave[2][17];  // [parity][exclusive prefix of the 16 wave totals | tile total]
    const int n = n_dev ? *n_dev : n_host, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TILE = 1024 * SCAN_PER_THREAD;
    int carry = 0, par = 0;
    int v[SCAN_PER_THREAD], vn[SCAN_PER_THREAD];
    if (n > 0) scan_load_tile(data, tid * SCAN_PER_THREAD, n, v);
    for (int base = 0; base < n; base += TILE, par ^= 1) {
        const int i0 = base + tid * SCAN_PER_THREAD;
        if (base + TILE < n) scan_load_tile(data, i0 + TILE, n, vn);  // disjoint from this tile's stores
        int sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; ++k) sum += v[k];
        int incl = sum;  // inclusive scan of the thread totals inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[par][wave] = incl;
        __syncthreads();
        if (wave == 0) {
            int w = lane < 16 ? s_wave[par][lane] : 0, wi = w;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int t = __shfl_up(wi, off, 64);
                if (lane >= off) wi += t;
            }
            if (lane < 16) s_wave[par][lane] = wi - w;  // exclusive prefix of the wave totals
            if (lane == 15) s_wave[par][16] = wi;        // the tile's total
        }
        __syncthreads();
        int run = carry + s_wave[par][wave] + incl - sum;
        carry += s_wave[par][16];
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; ++k) {
            if (i0 + k < n) data[i0 + k] = run;
            run += v[k];
        }
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; ++k) v[k] = vn[k];
    }
    if (tid == 0) data[n] = carry;
}

// ---- the same scan over many workgroups (n > SCAN_SINGLE_MAX): tile totals, a scan of the totals by the single-workgroup kernel above, then
// every tile scans itself behind its offset.  A tile is 4096 elements = four rows of 1024; thread t owns elements 4t .. 4t + 3 of every row,
// so that all loads and stores are 16 bytes per lane at a 16-byte lane pitch (the single-workgroup kernel walks 16 consecutive elements per
// thread: every wave-level access touches 64 cache lines, ~9 us per 16 k elements -- 111 us for the 200 k landmark counts of config 5 and
// 72 us for each histogram of its 4.2 M-pair radix passes; this takes ~15 us for either).
#define SCAN_SINGLE_MAX 16384
#define SCAN_TILE 4096
__device__ __forceinline__ void scan_tile_load(const int* __restrict__ data, int base, int n, int tid, int4 (&v)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = base + c * 1024 + tid * 4;
        if (i + 4 <= n) v[c] = *reinterpret_cast<const int4*>(data + i);
        else v[c] = make_int4(i < n ? data[i] : 0, i + 1 < n ? data[i + 1] : 0, i + 2 < n ? data[i + 2] : 0, 0);
    }
}
__global__ __launch_bounds__(256) void k_scan_tile_sums(const int* __restrict__ data, int n, int* __restrict__ tile_sum) {
    __shared__ int s_w[4];
    int4 v[4];
    scan_tile_load(data, blockIdx.x * SCAN_TILE, n, threadIdx.x, v);
    int t = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) t += v[c].x + v[c].y + v[c].z + v[c].w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_scan_tiles(int* __restrict__ data, int n, const int* __restrict__ tile_off, int ntiles) {
    __shared__ int s_w[4][4];  // [row][wave]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, base = blockIdx.x * SCAN_TILE;
    int4 v[4];
    scan_tile_load(data, base, n, tid, v);
    int sum[4], incl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) incl[c] = sum[c] = v[c].x + v[c].y + v[c].z + v[c].w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = __shfl_up(incl[c], off, 64);
            if (lane >= off) incl[c] += t;
        }
    if (lane == 63)
#pragma unroll
        for (int c = 0; c < 4; ++c) s_w[c][wave] = incl[c];
    __syncthreads();
    int run = tile_off[blockIdx.x];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int before = 0, row = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? s_w[c][w] : 0;
            row += s_w[c][w];
        }
        const int x0 = run + before + incl[c] - sum[c];
        const int4 o = make_int4(x0, x0 + v[c].x, x0 + v[c].x + v[c].y, x0 + v[c].x + v[c].y + v[c].z);
        const int i = base + c * 1024 + tid * 4;
        if (i + 4 <= n) *reinterpret_cast<int4*>(data + i) = o;
        else {
            if (i < n) data[i] = o.x;
            if (i + 1 < n) data[i + 1] = o.y;
            if (i + 2 < n) data[i + 2] = o.z;
        }
        run += row;
    }
    if (blockIdx.x == 0 && tid == 0) data[n] = tile_off[ntiles];  // the total
}

// ---- stable LSD radix sort of (key, value) pairs by 6-bit digits
#define RS_BITS 6
#define RS_BINS 64
#define RS_THREADS 256
#define RS_ITEMS 8                         // per thread: a block owns RS_THREADS * RS_ITEMS consecutive elements
#define RS_TILE (RS_THREADS * RS_ITEMS)

// hist[digit * nblk + block] = elements of the block with that digit
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const unsigned* __restrict__ keys, const int* __restrict__ n_dev, int n_host, int shift, int nblk, int* __restrict__ hist) {
    __shared__ int s_h[RS_BINS];
    const int n = n_dev ? *n_dev : n_host, tid = threadIdx.x, blk = blockIdx.x;
    if (tid < RS_BINS) s_h[tid] = 0;
    __syncthreads();
    const int base = blk * RS_TILE;
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int i = base + k * RS_THREADS + tid;
        if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & (RS_BINS - 1)], 1);
    }
    __syncthreads();
    if (tid < RS_BINS) hist[tid * nblk + blk] = s_h[tid];
}

// Scatter of one pass.  The block walks its elements in order (round r = elements base + r * 256 .. + 255, wave by wave, lane by lane);
// an element's position = scanned hist[digit][block] + the elements of the same digit before it in the block.  That in-block rank =
// (same digit in earlier rounds / waves: a prefix over the 32 (round, wave) counters per digit) + (same digit in lower lanes of its own
// wave: six ballots narrow the wave down to the lanes that share the digit).
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const unsigned* __restrict__ keys_in, const unsigned long long* __restrict__ vals_in, const int* __restrict__ n_dev,
                                                           int n_host, int shift, int nblk, const int* __restrict__ hist_scanned, unsigned* __restrict__ keys_out,
                                                           unsigned long long* __restrict__ vals_out) {
    __shared__ int s_cnt[RS_ITEMS * (RS_THREADS / 64)][RS_BINS];  // [round * 4 + wave][digit]
    __shared__ int s_base[RS_BINS];
    const int n = n_dev ? *n_dev : n_host, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
    for (int t = tid; t < RS_ITEMS * (RS_THREADS / 64) * RS_BINS; t += RS_THREADS) (&s_cnt[0][0])[t] = 0;
    if (tid < RS_BINS) s_base[tid] = hist_scanned[tid * nblk + blk];
    __syncthreads();
    const int base = blk * RS_TILE;
    unsigned key[RS_ITEMS];
    unsigned long long val[RS_ITEMS];
    int rank[RS_ITEMS];
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int i = base + r * RS_THREADS + tid;
        const bool live = i < n;
        key[r] = live ? keys_in[i] : 0u;
        val[r] = live ? vals_in[i] : 0ull;
        const unsigned d = (key[r] >> shift) & (RS_BINS - 1);
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        rank[r] = __popcll(same & below);
        if (live && rank[r] == 0) s_cnt[r * (RS_THREADS / 64) + wave][d] = __popcll(same);  // the first lane of every digit group of the wave
    }
    __syncthreads();
    if (tid < RS_BINS) {  // exclusive prefix over the (round, wave) slots of digit `tid`
        int run = 0;
        for (int q = 0; q < RS_ITEMS * (RS_THREADS / 64); ++q) {
            const int c = s_cnt[q][tid];
            s_cnt[q][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int i = base + r * RS_THREADS + tid;
        if (i < n) {
            const unsigned d = (key[r] >> shift) & (RS_BINS - 1);
            const int o = s_base[d] + s_cnt[r * (RS_THREADS / 64) + wave][d] + rank[r];
            keys_out[o] = key[r];
            vals_out[o] = val[r];
        }
    }
}


// (key, index) pairs of one small array: bitonic network over the 64-bit composites key << 32 | index in LDS
__global__ __launch_bounds__(1024) void k_sort_small(const unsigned* __restrict__ keys_in, const int* __restrict__ idx_in, int n, int npow2, unsigned* __restrict__ keys_out,
                                                    int* __restrict__ idx_out) {
    extern __shared__ unsigned long long s_v[];
    const int tid = threadIdx.x;
    for (int i = tid; i < npow2; i += 1024) s_v[i] = i < n ? ((unsigned long long)keys_in[i] << 32) | (unsigned)idx_in[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_v[i], b = s_v[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s_v[i] = b;
                        s_v[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 1024) {
        keys_out[i] = (unsigned)(s_v[i] >> 32);
        idx_out[i] = (int)(unsigned)s_v[i];
    }
}
}  // namespace

size_t sv_scan_scratch_ints(size_t n) { return n > SCAN_SINGLE_MAX ? (n + SCAN_TILE - 1) / SCAN_TILE + 8 : 0; }
void sv_scan_i32(hipStream_t s, int* data, int n, int* scratch) {
    if (n <= SCAN_SINGLE_MAX) {
        hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, data, (const int*)nullptr, n);
        return;
    }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, s, data, n, scratch);
    hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, scratch, (const int*)nullptr, ntiles);  // scratch[ntiles] = the total
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, s, data, n, scratch, ntiles);
}
int sv_sort_passes(int bits) { return (bits + RS_BITS - 1) / RS_BITS; }
static inline size_t hist_ints(size_t n) { return ((size_t)RS_BINS * ((n + RS_TILE - 1) / RS_TILE + 1) + 1 + 3) & ~size_t(3); }  // the scan's scratch follows, 16-byte aligned
size_t sv_sort_hist_ints(size_t n) { return hist_ints(n) + sv_scan_scratch_ints(hist_ints(n)); }
int sv_sort_pairs(hipStream_t s, unsigned* const keys[2], unsigned long long* const vals[2], int start, int n, int bits, int* hist, const int* n_dev) {
    int cur = start;
    if (n <= 0) return cur ^ (sv_sort_passes(bits) & 1);
    const int nblk = (n + RS_TILE - 1) / RS_TILE, passes = sv_sort_passes(bits);
    for (int pass = 0; pass < passes; ++pass, cur ^= 1) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(RS_THREADS), 0, s, keys[cur], n_dev, n, pass * RS_BITS, nblk, hist);
        sv_scan_i32(s, hist, RS_BINS * nblk, hist + hist_ints((size_t)n));
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(RS_THREADS), 0, s, keys[cur], vals[cur], n_dev, n, pass * RS_BITS, nblk, hist, keys[cur ^ 1], vals[cur ^ 1]);
    }
    return cur;
}
// false: the device does not grant the LDS this size needs (the caller takes the radix sort instead) -- a refused attribute must not turn
// into a launch that silently does not run and a merge-join over unsorted keys
bool sv_sort_small(hipStream_t s, const unsigned* keys_in, const int* idx_in, int n, unsigned* keys_out, int* idx_out) {
    if (n <= 0) return true;
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    const size_t lds = (size_t)npow2 * 8;
    if (lds > 48 * 1024 && sv_allow_dynamic_lds((const void*)k_sort_small, lds) != hipSuccess) return false;  // (per device, checked)
    hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), lds, s, keys_in, idx_in, n, npow2, keys_out, idx_out);
    return hipGetLastError() == hipSuccess;
}
