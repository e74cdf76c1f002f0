// Which counters does global_load_lds_dwordx4 tick on gfx950?  t(lgkmcnt(0)) - t(issue) vs t(vmcnt(0)) - t(issue), cold and warm source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint8_t* src, long long* out) {
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s;
    for (int rep = 0; rep < 4; ++rep) {
        const uint8_t* g = src + (size_t)rep * (1 << 20) * (rep < 2) + threadIdx.x * 16;  // reps 0,1: new lines (cold); 2,3: the rep-0 lines again (warm)
        uint32_t keep;
        long long t0, t1, t2;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (threadIdx.x == 0) {
            out[rep * 2] = t1 - t0;
            out[rep * 2 + 1] = t2 - t0;
        }
    }
}
int main() {
    uint8_t* d;
    long long* o;
    hipMalloc(&d, 4 << 20);
    hipMemset(d, 1, 4 << 20);
    hipMalloc(&o, 64);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    long long h[8];
    hipMemcpy(h, o, 64, hipMemcpyDeviceToHost);
    printf("{\"what\": \"cycles (s_memtime, 100 MHz ticks?) from issue of one global_load_lds_dwordx4 to lgkmcnt(0) / to vmcnt(0)\", \"cold\": [[%lld, %lld], [%lld, %lld]], \"warm\": [[%lld, %lld], [%lld, %lld]]}\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
