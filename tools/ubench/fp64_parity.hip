// Debug aid: is device fp64 (+ - * / sqrt, no contraction) bit-identical to the host's?  Runs the same __host__ __device__
// undistortion routine on both sides and prints the first differing intermediate.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64_parity fp64_parity.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

struct Trace { double v[64]; int n; };

__host__ __device__ inline void undist(double fx_, double fy_, double cx_, double cy_, const double* kd, float px, float py, Trace& T) {
    const double fx = (double)(float)fx_, fy = (double)(float)fy_, cx = (double)(float)cx_, cy = (double)(float)cy_;
    const double k0 = (double)(float)kd[0], k1 = (double)(float)kd[1], k2 = (double)(float)kd[2], k3 = (double)(float)kd[3], k4 = (double)(float)kd[4];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double u = px, v = py;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    double error = 1.7976931348623157e308;
    T.n = 0;
    T.v[T.n++] = ifx; T.v[T.n++] = x0; T.v[T.n++] = y0;
    for (int j = 0; j < 20; ++j) {
        if (error < 1e-6) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
        if (T.n < 56) { T.v[T.n++] = r2; T.v[T.n++] = icdist; T.v[T.n++] = deltaX; T.v[T.n++] = x; T.v[T.n++] = y; }
        r2 = x * x + y * y;
        const double r4 = r2 * r2, r6 = r4 * r2, a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k0 * r2 + k1 * r4 + k4 * r6;
        const double xd = x * cdist + k2 * a1 + k3 * a2;
        const double yd = y * cdist + k2 * a3 + k3 * a1;
        const double xp = xd * fx + cx, yp = yd * fy + cy;
        error = sqrt((xp - u) * (xp - u) + (yp - v) * (yp - v));
        if (T.n < 56) { T.v[T.n++] = cdist; T.v[T.n++] = xd; T.v[T.n++] = error; }
    }
    T.v[T.n++] = fx * x + cx;
}

__global__ void k(double fx, double fy, double cx, double cy, const double* kd, const float* pts, int n, Trace* out) {
    int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) undist(fx, fy, cx, cy, kd, pts[2 * i], pts[2 * i + 1], out[i]);
}

int main() {
    const int n = 4096;
    std::vector<float> pts(2 * n);
    unsigned s = 12345;
    for (auto& p : pts) { s = s * 1664525u + 1013904223u; p = (s >> 8) * (752.0f / 16777216.0f); }
    double kd[5] = {-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0};
    float* dp; double* dk; Trace* dt;
    hipMalloc(&dp, pts.size() * 4); hipMalloc(&dk, 40); hipMalloc(&dt, n * sizeof(Trace));
    hipMemcpy(dp, pts.data(), pts.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dk, kd, 40, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, 458.654, 457.296, 367.215, 248.375, dk, dp, n, dt);
    std::vector<Trace> g(n);
    hipMemcpy(g.data(), dt, n * sizeof(Trace), hipMemcpyDeviceToHost);
    int bad = 0;
    const char* names[8] = {"r2", "icdist", "deltaX", "x", "y", "cdist", "xd", "error"};
    for (int i = 0; i < n; ++i) {
        Trace h;
        undist(458.654, 457.296, 367.215, 248.375, kd, pts[2 * i], pts[2 * i + 1], h);
        if (h.n != g[i].n) { if (bad++ < 5) printf("pt %d: trace length %d vs %d\n", i, h.n, g[i].n); continue; }
        for (int j = 0; j < h.n; ++j)
            if (memcmp(&h.v[j], &g[i].v[j], 8)) {
                if (bad++ < 10) printf("pt %d: first diff at slot %d (%s): host %.17g dev %.17g\n", i, j, j < 3 ? "init" : j == h.n - 1 ? "final" : names[(j - 3) % 8], h.v[j], g[i].v[j]);
                break;
            }
    }
    printf("%d of %d points differ\n", bad, n);
    return 0;
}
