// micro-benchmark: throughput of v_bcnt_u32_b32 vs v_xor/v_add on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(uint32_t* out, int iters) {
    uint32_t a[8], acc[8];
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 2654435761u + k; acc[k] = k; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) acc[k] = __popc(a[k] ^ acc[k]) + acc[k];           // xor + bcnt(accumulate)
            else if (MODE == 1) acc[k] = (a[k] ^ acc[k]) + acc[k];             // xor + add
            else acc[k] = __popc(a[k]) + acc[k];                               // bcnt only
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double ops = (double)blocks * 256 * iters * 8;  // loop bodies (each = 2 or 1 VALU)
            if (rep) printf("mode %d: %.3f ms  -> %.2f G bodies/s, cycles per wave-body @2.4GHz: %.2f\n", mode, ms, ops / ms / 1e6,
                            ms * 1e-3 * 2.4e9 * 1024 / (ops / 64));
        }
    }
    return 0;
}
