#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k(int* p) { if (p) *p = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now(), t;
    int n = 0; hipGetDeviceCount(&n); t = now(); printf("hipGetDeviceCount %.1f ms\n", t - t0); t0 = t;
    hipSetDevice(0); t = now(); printf("hipSetDevice %.1f ms\n", t - t0); t0 = t;
    hipFree(nullptr); t = now(); printf("hipFree(nullptr) %.1f ms\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (int*) nullptr); hipDeviceSynchronize(); t = now(); printf("first launch + sync %.1f ms\n", t - t0); t0 = t;
    void *d = nullptr, *h = nullptr;
    hipMalloc(&d, 8 << 20); t = now(); printf("hipMalloc 8 MB %.1f ms\n", t - t0); t0 = t;
    hipHostMalloc(&h, 8 << 20); t = now(); printf("hipHostMalloc 8 MB %.1f ms\n", t - t0); t0 = t;
    hipMemcpy(d, h, 8 << 20, hipMemcpyHostToDevice); t = now(); printf("first H2D 8 MB %.1f ms\n", t - t0); t0 = t;
    hipMemcpy(d, h, 8 << 20, hipMemcpyHostToDevice); t = now(); printf("second H2D 8 MB %.1f ms\n", t - t0); t0 = t;
    hipMemcpy(h, d, 8 << 20, hipMemcpyDeviceToHost); t = now(); printf("first D2H 8 MB %.1f ms\n", t - t0); t0 = t;
    hipMemset(d, 0, 64); hipDeviceSynchronize(); t = now(); printf("memset %.1f ms\n", t - t0); t0 = t;
    hipHostFree(h); t = now(); printf("hipHostFree %.1f ms\n", t - t0); t0 = t;
    hipFree(d); t = now(); printf("hipFree %.1f ms\n", t - t0); t0 = t;
}
