// Issue cost of the VALU instructions the front-end kernels are made of, in shader cycles per wave64 instruction per SIMD (gfx950).
// Each wave runs 8 independent dependency chains of one instruction (inline asm, so the opcode is exactly the one named) for
// ITERS x 64 instructions; W waves per SIMD (W = 1, 2, 4, 8) run side by side on every CU.  cycles per wave-instruction =
// elapsed shader cycles (s_memtime inside the kernel, max over the waves of a CU) / (instructions per wave x W): at W = 8 the
// dependency latency is hidden and the quotient is the SIMD's issue interval for that opcode.  Cross-check: the same quotient from
// HIP-event wall time x the shader clock reported by the device.  Output: JSON on stdout (committed as profiles/r02_ubench_valu.json).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define UNROLL 8  // x CHAINS = 64 instructions per loop trip

#define DEFINE_KERNEL(NAME, ASM, TYPE, CONSTR)                                                                  \
    __global__ void NAME(long long* out, int iters) {                                                            \
        TYPE acc[CHAINS], a = (TYPE)(threadIdx.x * 2654435761u + 12345u);                                        \
        for (int k = 0; k < CHAINS; ++k) acc[k] = (TYPE)(k + 1);                                                 \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                       \
        for (int i = 0; i < iters; ++i) {                                                                        \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                                 \
                _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) asm volatile(ASM : "+" CONSTR(acc[k]) : CONSTR(a)); \
            }                                                                                                    \
        }                                                                                                        \
        __builtin_amdgcn_s_waitcnt(0);                                                                           \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                       \
        TYPE r = 0;                                                                                              \
        for (int k = 0; k < CHAINS; ++k) r += acc[k];                                                            \
        if (r == (TYPE)0x5a5a5a5a) out[1 << 20] = 1;  /* keeps the chains alive */                               \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;           \
    }

DEFINE_KERNEL(k_add, "v_add_u32 %0, %1, %0", uint32_t, "v")
DEFINE_KERNEL(k_xor, "v_xor_b32 %0, %1, %0", uint32_t, "v")
DEFINE_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0", uint32_t, "v")
DEFINE_KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %1", uint32_t, "v")
DEFINE_KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %1, %0", uint32_t, "v")
DEFINE_KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %0, %1", uint32_t, "v")
DEFINE_KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %1, %0, %1", uint32_t, "v")
DEFINE_KERNEL(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %0, %0", uint32_t, "v")
DEFINE_KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %0, %0", uint32_t, "v")
DEFINE_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %1, %0", uint32_t, "v")
DEFINE_KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %0, %1", float, "v")
DEFINE_KERNEL(k_fma_f64, "v_fma_f64 %0, %1, %0, %1", double, "v")
DEFINE_KERNEL(k_add_f64, "v_add_f64 %0, %1, %0", double, "v")

struct Entry {
    const char* name;
    void (*fn)(long long*, int);
};

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk_mhz = prop.clockRate / 1000.0;
    long long* d;
    hipMalloc(&d, ((1 << 20) + 16) * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const Entry entries[] = {{"v_add_u32", k_add},           {"v_xor_b32", k_xor},       {"v_bcnt_u32_b32", k_bcnt},   {"v_perm_b32", k_perm},
                             {"v_pk_min_i16", k_pk_min_i16}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_mad_i32_i24", k_mad_i24}, {"v_dot2_u32_u16", k_dot2_u16},
                             {"v_dot4_u32_u8", k_dot4_u8},   {"v_mul_lo_u32", k_mul_lo},   {"v_fma_f32", k_fma_f32},     {"v_fma_f64", k_fma_f64},
                             {"v_add_f64", k_add_f64}};
    const int iters = 2048;
    const double insts = (double)iters * UNROLL * CHAINS;
    std::printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"reported_clock_mhz\": %.0f,\n", prop.gcnArchName, cus, clk_mhz);
    std::printf(" \"method\": \"8 independent chains per wave, %d instructions per wave, one workgroup of 256 x W threads per CU; cycles = max s_memtime delta over the waves / (instructions x W)\",\n", (int)insts);
    std::printf(" \"instructions\": {\n");
    for (size_t e = 0; e < sizeof(entries) / sizeof(entries[0]); ++e) {
        std::printf("  \"%s\": {", entries[e].name);
        for (int w = 1; w <= 8; w *= 2) {
            const int threads = 256 * w;  // w waves on each of the 4 SIMDs (a 2048-thread workgroup is not launchable: two workgroups of 1024 at w = 8)
            const int tpb = threads > 1024 ? 1024 : threads, blocks_per_cu = threads / tpb;
            const int blocks = cus * blocks_per_cu, waves = blocks * tpb / 64;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(entries[e].fn, dim3(blocks), dim3(tpb), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> t(waves);
            hipMemcpy(t.data(), d, waves * sizeof(long long), hipMemcpyDeviceToHost);
            long long mx = 0;
            for (long long v : t) mx = v > mx ? v : mx;
            std::printf("%s\"w%d\": {\"cycles_per_wave_inst_memtime\": %.3f, \"cycles_per_wave_inst_wall\": %.3f}", w == 1 ? "" : ", ", w, (double)mx / (insts * w),
                        ms * 1e-3 * clk_mhz * 1e6 / (insts * w));
        }
        std::printf("}%s\n", e + 1 < sizeof(entries) / sizeof(entries[0]) ? "," : "");
    }
    std::printf(" }\n}\n");
    return 0;
}
