// Micro-benchmark for a design question: how fast can 1.53 M window records be read in an order that is
// NOT the window order (statistics aggregated per emission-table row instead of per window)?
//   A: one 64-byte record per window, random order          B: four 16-byte pieces per window 1 KiB apart (lane-minor tiles)
//   C: the same 64-byte records in window order (coalesced)  W: the writer that fills the 98 MB array before every read pass
// hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_write(double2* __restrict__ R, int64_t n4) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n4) R[i] = make_double2(1.0 + (double) (i & 7) * 0.125, 0.5);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const double2* __restrict__ R, const int32_t* __restrict__ perm, int64_t n,
                                                double* __restrict__ out) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.0;
    if (i < n) {
        const int64_t w = MODE == 2 ? i : perm[i];   // MODE 0, 3: 64 B records in random order
        double2 v[4];
        if (MODE == 1) {
            const int64_t tile = w >> 8, lane = (w >> 2) & 63, j = w & 3;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = R[((tile * 4 + j) * 4 + q) * 64 + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = R[w * 4 + q];
        }
        const double f[4] = {v[0].x, v[0].y, v[1].x, v[1].y}, b[4] = {v[2].x, v[2].y, v[3].x, v[3].y};
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int s = 0; s < 4; s++) acc[p * 4 + s] = f[p] * 0.25 * b[s];
    }
    if (MODE == 3) {   // no group reduction: the loads alone
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) t += acc[k];
        if (i < n) out[i] = t;
        return;
    }
    // sum over the 16 lanes of a group
#pragma unroll
    for (int k = 0; k < 16; k++)
        for (int o = 8; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if ((threadIdx.x & 15) == 0 && i < n) {
        double* dst = out + (i >> 4) * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = acc[k];
    }
}

int main() {
    const int64_t n = 1527428;
    double2* R; int32_t* perm; double* out;
    hipMalloc(&R, (n + 256) * 64); hipMalloc(&perm, n * 4); hipMalloc(&out, (n + 16) * 8);
    std::vector<int32_t> h(n);
    std::iota(h.begin(), h.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(h.begin(), h.end(), rng);
    hipMemcpy(perm, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned gb = (unsigned) ((n + 255) / 256), wb = (unsigned) ((n * 4 + 255) / 256);
    const char* names[4] = {"A 64 B records, random order", "B 4 x 16 B pieces (lane-minor tiles), random order", "C 64 B records, window order", "D as A without the 16-lane reduction"};
    for (int mode = 0; mode < 4; mode++) {
        float sum = 0.f, wsum = 0.f;
        for (int it = 0; it < 12; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_write, dim3(wb), dim3(256), 0, 0, R, n * 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float wms; hipEventElapsedTime(&wms, e0, e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(gb), dim3(256), 0, 0, R, perm, n, out);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(gb), dim3(256), 0, 0, R, perm, n, out);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(gb), dim3(256), 0, 0, R, perm, n, out);
            if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(gb), dim3(256), 0, 0, R, perm, n, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2) { sum += ms; wsum += wms; }
        }
        printf("%-55s %.1f us   (writer %.1f us)\n", names[mode], sum / 10 * 1e3, wsum / 10 * 1e3);
    }
    return 0;
}
