// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths used by the svgpu kernels (MI355X_MICROARCH.md,
// section HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel moves exactly BYTES (1 GiB, beyond the 256 MiB Infinity Cache) once.
//   build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
//   run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -o p -- ./pmc_calib ; same with WRITE_SIZE
//   then:  tools/pmc_traffic.py --calibrate out_f out_w profiles/r01_pmc_calibration.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t BYTES = size_t(1) << 30;

template <class T>
__global__ void k_calib_read(const T* __restrict__ src, size_t n, T* __restrict__ sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) {
        const T v = src[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        for (unsigned k = 0; k < sizeof(T); ++k) acc += b[k];
    }
    if (acc == 0xFFFFFFFFu) sink[0] = src[0];  // never true: keeps the loads alive
}
template <class T>
__global__ void k_calib_write(T* __restrict__ dst, size_t n, T v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = v;
}

int main() {
    void *a = nullptr, *sink = nullptr;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(a, 1, BYTES);
    hipDeviceSynchronize();
    const int grid = 256 * 32, block = 256;
    hipLaunchKernelGGL(k_calib_read<uint8_t>, dim3(grid), dim3(block), 0, 0, (const uint8_t*)a, BYTES, (uint8_t*)sink);
    hipLaunchKernelGGL(k_calib_read<uint32_t>, dim3(grid), dim3(block), 0, 0, (const uint32_t*)a, BYTES / 4, (uint32_t*)sink);
    hipLaunchKernelGGL(k_calib_read<uint2>, dim3(grid), dim3(block), 0, 0, (const uint2*)a, BYTES / 8, (uint2*)sink);
    hipLaunchKernelGGL(k_calib_read<uint4>, dim3(grid), dim3(block), 0, 0, (const uint4*)a, BYTES / 16, (uint4*)sink);
    hipLaunchKernelGGL(k_calib_write<uint8_t>, dim3(grid), dim3(block), 0, 0, (uint8_t*)a, BYTES, (uint8_t)3);
    hipLaunchKernelGGL(k_calib_write<uint32_t>, dim3(grid), dim3(block), 0, 0, (uint32_t*)a, BYTES / 4, 3u);
    hipLaunchKernelGGL(k_calib_write<uint4>, dim3(grid), dim3(block), 0, 0, (uint4*)a, BYTES / 16, make_uint4(3, 3, 3, 3));
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("moved %zu bytes per kernel\n", BYTES);
    return 0;
}
