// Does gfx950's LDS-DMA (global_load_lds_dword / _dwordx4) take source addresses that are not multiples of its width, and do
// ds_read_b32 / ds_read_b64 take unaligned LDS addresses?  (k_describe fetches 31 x 31 / 37 x 37 patches at arbitrary byte addresses.)
// One wave; lane i asks for `width` bytes at src + off + i * stride.  Output: one JSON line (profiles/r06_ubench_glds.json).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

template <int W>
__global__ void k_glds(const uint8_t* src, int off, int stride, uint8_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t s[64 * 16 + 64];
    for (int i = threadIdx.x; i < (int)sizeof(s); i += 64) s[i] = 0xEE;
    __syncthreads();
    const uint8_t* g = src + off + threadIdx.x * stride;
    auto gp = (const __attribute__((address_space(1))) void*)g;
    auto lp = (__attribute__((address_space(3))) void*)s;
    if constexpr (W == 16) __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
    else if constexpr (W == 4) __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
    else __builtin_amdgcn_global_load_lds(gp, lp, 1, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * W; i += 64) out[i] = s[i];
}

__global__ void k_ds_unaligned(int off, uint32_t* out32, unsigned long long* out64) {
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s + off + threadIdx.x * 9;
    uint32_t v32;
    unsigned long long v64;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    out32[threadIdx.x] = v32;
    out64[threadIdx.x] = v64;
}

#define CK(x)                                                         \
    do {                                                              \
        hipError_t e = (x);                                           \
        if (e != hipSuccess) {                                        \
            printf("{\"error\": \"%s\"}\n", hipGetErrorString(e));   \
            return 1;                                                 \
        }                                                             \
    } while (0)

template <int W>
static int run(const uint8_t* dsrc, const std::vector<uint8_t>& hsrc, int off, int stride, uint8_t* dout) {
    hipLaunchKernelGGL(k_glds<W>, dim3(1), dim3(64), 0, 0, dsrc, off, stride, dout);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint8_t> h(64 * W);
    hipMemcpy(h.data(), dout, h.size(), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < W; ++k) bad += h[l * W + k] != hsrc[off + l * stride + k];
    return bad;
}

int main() {
    std::vector<uint8_t> hsrc(1 << 16);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (uint8_t)((i * 2654435761u) >> 13);
    uint8_t *dsrc, *dout;
    CK(hipMalloc(&dsrc, hsrc.size()));
    CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dsrc, hsrc.data(), hsrc.size(), hipMemcpyHostToDevice));
    printf("{\"what\": \"mismatching bytes (of 64 x width) of one global_load_lds per source misalignment; -1 = fault\"");
    printf(", \"dwordx4_stride48\": [");
    for (int off = 0; off < 16; ++off) printf("%s%d", off ? ", " : "", run<16>(dsrc, hsrc, 256 + off, 48, dout));
    printf("], \"dwordx4_stride37\": [");
    for (int off = 0; off < 16; ++off) printf("%s%d", off ? ", " : "", run<16>(dsrc, hsrc, 256 + off, 37, dout));
    printf("], \"dword_stride4\": [");
    for (int off = 0; off < 4; ++off) printf("%s%d", off ? ", " : "", run<4>(dsrc, hsrc, 256 + off, 4, dout));
    printf("], \"dword_stride37\": [");
    for (int off = 0; off < 4; ++off) printf("%s%d", off ? ", " : "", run<4>(dsrc, hsrc, 256 + off, 37, dout));
    printf("], \"ubyte_stride1\": [%d]", run<1>(dsrc, hsrc, 259, 1, dout));
    uint32_t* d32;
    unsigned long long* d64;
    CK(hipMalloc(&d32, 256));
    CK(hipMalloc(&d64, 512));
    printf(", \"ds_read_unaligned_bad32_bad64\": [");
    for (int off = 0; off < 8; ++off) {
        hipLaunchKernelGGL(k_ds_unaligned, dim3(1), dim3(64), 0, 0, off, d32, d64);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("-1");
            break;
        }
        uint32_t h32[64];
        unsigned long long h64[64];
        hipMemcpy(h32, d32, 256, hipMemcpyDeviceToHost);
        hipMemcpy(h64, d64, 512, hipMemcpyDeviceToHost);
        int b32 = 0, b64 = 0;
        for (int l = 0; l < 64; ++l) {
            uint8_t e[8];
            for (int k = 0; k < 8; ++k) e[k] = (uint8_t)((off + l * 9 + k) * 7 + 3);
            uint32_t e32;
            unsigned long long e64;
            memcpy(&e32, e, 4);
            memcpy(&e64, e, 8);
            b32 += h32[l] != e32;
            b64 += h64[l] != e64;
        }
        printf("%s[%d, %d]", off ? ", " : "", b32, b64);
    }
    printf("]}\n");
    return 0;
}
