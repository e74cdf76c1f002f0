// Latencies that bound the one-workgroup solvers of bundle adjustment (gfx950), in shader cycles (s_memtime), single workgroup:
//   dependent chains of v_fma_f64 / v_fma_f32 / v_rcp_f64 / v_rsq_f64 / v_readlane->v_fma / ds_read_b64 pointer chase / global (L2)
//   pointer chase, the cost of s_memtime itself, and __syncthreads() with 1..16 waves.
// One wave (or W waves for the barrier), N dependent operations, elapsed / N.  Output: JSON on stdout (profiles/r02_ubench_latency.json).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
#include <vector>

#define N_OPS 512

__global__ void k_fma64(long long* out, double a) {
    double x = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_OPS; ++i) asm volatile("v_fma_f64 %0, %1, %0, %1" : "+v"(x) : "v"(a));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_fma32(long long* out, float a) {
    float x = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_OPS; ++i) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(x) : "v"(a));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123f) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_rcp64(long long* out, double a) {
    double x = threadIdx.x + a;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_OPS; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_rsq64(long long* out, double a) {
    double x = threadIdx.x + a;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_OPS; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// accuracy of the hardware seeds: max relative error over a sweep
__global__ void k_seed_accuracy(double* out) {
    double worst_rcp = 0.0, worst_rsq = 0.0;
    for (int i = 0; i < 4096; ++i) {
        const double x = 1.0 + (threadIdx.x * 4096 + i) * (3.0 / (64.0 * 4096.0));
        const double r = __builtin_amdgcn_rcp(x), q = __builtin_amdgcn_rsq(x);
        worst_rcp = fmax(worst_rcp, fabs(r * x - 1.0));
        worst_rsq = fmax(worst_rsq, fabs(q * q * x - 1.0) * 0.5);
    }
    out[threadIdx.x] = worst_rcp;
    out[64 + threadIdx.x] = worst_rsq;
}
__global__ void k_readlane(long long* out, double a) {
    double x = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N_OPS; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
        x = fma(__hiloint2double(hi, lo), a, x);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_lds_chase(long long* out) {
    __shared__ long long s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    long long p = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N_OPS; ++i) p = s[p];
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (p == -1) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_lds_write_read(long long* out) {  // ds_write then ds_read of the value by another lane (same wave)
    __shared__ double s[64];
    double x = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N_OPS; ++i) {
        s[threadIdx.x] = x;
        x = s[(threadIdx.x + 1) & 63] + 1.0;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_global_chase(long long* out, const long long* chain) {
    long long p = threadIdx.x;
    for (int i = 0; i < 64; ++i) p = chain[p];  // warm the L2 / TLB
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N_OPS; ++i) p = chain[p];
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (p == -1) out[100] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_memtime(long long* out) {
    long long acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) acc += __builtin_amdgcn_s_memtime();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 1) out[100] = 1;
    if (threadIdx.x == 0) out[0] = (t1 - t0) * (N_OPS / 64);
}
__global__ void k_barrier(long long* out) {
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N_OPS; ++i) __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_barrier_lds(long long* out) {  // the usual pattern: LDS write, barrier, LDS read of another wave's value
    __shared__ double s[1024];
    double x = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < N_OPS; ++i) {
        s[threadIdx.x] = x;
        __syncthreads();
        x = s[(threadIdx.x + 64) % blockDim.x] + 1.0;
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 0.123) out[100] = 1;
    if (threadIdx.x == 0) out[0] = (t1 - t0) / 2;  // per barrier
}

int main() {
    long long *d, h = 0;
    hipMalloc(&d, 1024 * sizeof(long long));
    std::vector<long long> chain(1 << 16);
    for (size_t i = 0; i < chain.size(); ++i) chain[i] = (i * 4099 + 77) & (chain.size() - 1);
    long long* dc;
    hipMalloc(&dc, chain.size() * 8);
    hipMemcpy(dc, chain.data(), chain.size() * 8, hipMemcpyHostToDevice);
    auto get = [&]() {
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        return (double)h / N_OPS;
    };
    printf("{\n \"unit\": \"shader cycles (s_memtime) per dependent operation, one wave unless stated\",\n");
    for (int rep = 0; rep < 2; ++rep) {  // second pass = warm
        const bool p = rep == 1;
        hipLaunchKernelGGL(k_fma64, dim3(1), dim3(64), 0, 0, d, 1.0000001);
        double v = get();
        if (p) printf(" \"v_fma_f64_dependent\": %.1f,\n", v);
        hipLaunchKernelGGL(k_fma32, dim3(1), dim3(64), 0, 0, d, 1.0000001f);
        v = get();
        if (p) printf(" \"v_fma_f32_dependent\": %.1f,\n", v);
        hipLaunchKernelGGL(k_rcp64, dim3(1), dim3(64), 0, 0, d, 1.5);
        v = get();
        if (p) printf(" \"v_rcp_f64_dependent\": %.1f,\n", v);
        hipLaunchKernelGGL(k_rsq64, dim3(1), dim3(64), 0, 0, d, 1.5);
        v = get();
        if (p) printf(" \"v_rsq_f64_dependent\": %.1f,\n", v);
        hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, d, 1.0000001);
        v = get();
        if (p) printf(" \"readlane_x2_plus_fma_f64\": %.1f,\n", v);
        hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, d);
        v = get();
        if (p) printf(" \"ds_read_b64_pointer_chase\": %.1f,\n", v);
        hipLaunchKernelGGL(k_lds_write_read, dim3(1), dim3(64), 0, 0, d);
        v = get();
        if (p) printf(" \"ds_write_b64_then_read_plus_add\": %.1f,\n", v);
        hipLaunchKernelGGL(k_global_chase, dim3(1), dim3(64), 0, 0, d, dc);
        v = get();
        if (p) printf(" \"global_load_b64_pointer_chase_L2\": %.1f,\n", v);
        hipLaunchKernelGGL(k_memtime, dim3(1), dim3(64), 0, 0, d);
        v = get();
        if (p) printf(" \"s_memtime\": %.1f,\n", v);
        for (int w : {1, 2, 4, 9, 16}) {
            hipLaunchKernelGGL(k_barrier, dim3(1), dim3(64 * w), 0, 0, d);
            v = get();
            if (p) printf(" \"syncthreads_%d_waves\": %.1f,\n", w, v);
            hipLaunchKernelGGL(k_barrier_lds, dim3(1), dim3(64 * w), 0, 0, d);
            v = get();
            if (p) printf(" \"lds_write_syncthreads_read_%d_waves\": %.1f,\n", w, v);
        }
    }
    double* da;
    hipMalloc(&da, 128 * 8);
    hipLaunchKernelGGL(k_seed_accuracy, dim3(1), dim3(64), 0, 0, da);
    double acc[128];
    hipMemcpy(acc, da, sizeof(acc), hipMemcpyDeviceToHost);
    double wr = 0, wq = 0;
    for (int i = 0; i < 64; ++i) {
        wr = wr > acc[i] ? wr : acc[i];
        wq = wq > acc[64 + i] ? wq : acc[64 + i];
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf(" \"v_rcp_f64_max_rel_error\": %.3e,\n \"v_rsq_f64_max_rel_error\": %.3e,\n \"device_clock_mhz\": %.0f\n}\n", wr, wq, prop.clockRate / 1000.0);
    return 0;
}
