// What does the vector-memory front end (TA / TCP) charge for the address patterns of a patch fetch?  k_describe is bound by it
// (TCP_TOTAL_CACHE_ACCESSES per keypoint == cycles per keypoint per CU, round 6).  Every wave issues N loads of one pattern, 8 waves per
// SIMD-slot-filling workgroup set, all CUs busy; result = shader cycles per wave-instruction per CU (lower = cheaper), for
//   kind:  glds16 (global_load_lds_dwordx4), ld16 (global_load_dwordx4), ld8 (global_load_dwordx2)
//   pattern: lanes grouped G per row (G consecutive 16-byte / 8-byte pieces), rows `pitch` apart, first piece at byte offset `mis` from 64-byte alignment
// Source footprint per workgroup: `rows` rows revisited (L1-resident when small, L2-resident when large).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 256

template <int KIND>
__global__ __launch_bounds__(256) void k_pat(const uint8_t* src, size_t wg_stride, int pitch, int G, int mis, int rows_mask, long long* out, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) uint8_t s[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s[wave]);
    const int W = KIND == 2 ? 8 : 16;
    const uint8_t* base = src + (size_t)blockIdx.x * wg_stride + 64 + mis;
    const int row = lane / G, part = lane % G;
    const int rows_per = 64 / G;
    uint32_t acc = 0;
    long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (KIND == 0) {
        for (int it = 0; it < N_IT; ++it) {
            const int r = ((it * 4 + wave) * rows_per + row) & rows_mask;
            const uint8_t* g = base + (size_t)r * pitch + part * W;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    else {
        for (int it = 0; it < N_IT; it += 8) {  // eight plain loads in flight per wave (the compiler counts these itself)
            uint32_t t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = (((it + u) * 4 + wave) * rows_per + row) & rows_mask;
                const uint8_t* g = base + (size_t)r * pitch + part * W;
                if (KIND == 1) {
                    uint4 v;
                    __builtin_memcpy(&v, g, 16);
                    t[u] = v.x ^ v.y ^ v.z ^ v.w;
                }
                else {
                    uint2 v;
                    __builtin_memcpy(&v, g, 8);
                    t[u] = v.x ^ v.y;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= t[u];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (acc == 0x12345) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t wg_stride = 1 << 20;  // 1 MB of source per workgroup slot (blocks share slots modulo 512)
    const int nblk = 256 * 2;          // two 4-wave workgroups per CU
    uint8_t* d;
    long long* o;
    uint32_t* sink;
    if (hipMalloc(&d, wg_stride * nblk + (1 << 20)) != hipSuccess) return 1;
    hipMemset(d, 3, wg_stride * nblk + (1 << 20));
    hipMalloc(&o, nblk * 8);
    hipMalloc(&sink, 64);
    std::vector<long long> h(nblk);
    struct Case { const char* name; int kind, G, mis, pitch, rows; };
    const Case cases[] = {
        {"glds16 contiguous 1KB aligned", 0, 64, 0, 1024, 64},
        {"glds16 G=8 (128B/row) aligned", 0, 8, 0, 640, 64},
        {"glds16 G=4 (64B/row) aligned", 0, 4, 0, 640, 64},
        {"glds16 G=4 (64B/row) mis=16", 0, 4, 16, 640, 64},
        {"glds16 G=4 (64B/row) mis=5", 0, 4, 5, 640, 64},
        {"glds16 G=2 (32B/row) aligned", 0, 2, 0, 640, 64},
        {"glds16 G=2 (32B/row) mis=5", 0, 2, 5, 640, 64},
        {"glds16 G=2 (32B/row) mis=37", 0, 2, 37, 640, 64},
        {"glds16 G=3 (48B/row) aligned", 0, 3, 0, 640, 64},
        {"glds16 G=3 (48B/row) mis=5", 0, 3, 5, 640, 64},
        {"glds16 G=1 (16B/row) aligned", 0, 1, 0, 640, 64},
        {"glds16 G=1 (16B/row) mis=5", 0, 1, 5, 640, 64},
        {"ld16 G=2 aligned", 1, 2, 0, 640, 64},
        {"ld16 G=2 mis=5", 1, 2, 5, 640, 64},
        {"ld16 G=4 aligned", 1, 4, 0, 640, 64},
        {"ld8 G=4 (32B/row) aligned", 2, 4, 0, 640, 64},
        {"ld8 G=4 (32B/row) mis=5", 2, 4, 5, 640, 64},
        {"ld8 G=5 (40B/row) mis=5", 2, 5, 5, 640, 64},
        {"ld8 G=8 (64B/row) aligned", 2, 8, 0, 640, 64},
        {"glds16 G=2 mis=5 L2 (1024 rows)", 0, 2, 5, 640, 1024},
        {"glds16 G=4 aligned L2 (1024 rows)", 0, 4, 0, 640, 1024},
        {"ld8 G=4 mis=5 L2 (1024 rows)", 2, 4, 5, 640, 1024},
    };
    printf("{\"what\": \"shader cycles per wave-level load instruction per CU (8 waves per CU issuing, 256 loads each, <= 8 in flight per wave); rows L1-resident unless noted\", \"cases\": {");
    bool first = true;
    for (const Case& c : cases) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) hipEventRecord(e0, 0);
            if (c.kind == 0) hipLaunchKernelGGL(k_pat<0>, dim3(nblk), dim3(256), 0, 0, d, wg_stride, c.pitch, c.G, c.mis, c.rows - 1, o, sink);
            if (c.kind == 1) hipLaunchKernelGGL(k_pat<1>, dim3(nblk), dim3(256), 0, 0, d, wg_stride, c.pitch, c.G, c.mis, c.rows - 1, o, sink);
            if (c.kind == 2) hipLaunchKernelGGL(k_pat<2>, dim3(nblk), dim3(256), 0, 0, d, wg_stride, c.pitch, c.G, c.mis, c.rows - 1, o, sink);
            if (rep == 1) hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("\"error\": 1}}\n"); return 1; }
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), o, nblk * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (long long v : h) sum += (double)v;
        // per CU: 2 workgroups x 4 waves x N_IT instructions within (mean block time) -- s_memtime ticks at 100 MHz on this part: report raw ticks too
        const double ticks = sum / nblk;
        printf("%s\"%s\": {\"ticks_per_inst_per_cu\": %.2f, \"ns_per_inst_per_cu\": %.2f}", first ? "" : ", ", c.name, ticks / (8.0 * N_IT), ms * 1e6 / (8.0 * N_IT));
        first = false;
    }
    printf("}}\n");
    return 0;
}
