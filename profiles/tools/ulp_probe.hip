// ulp_probe.hip — which device fp64 operations differ from the host's (glibc / IEEE) in the last bit?  (where do the
// last printed digits of --accelerate runs come from.)  For each operation: N arguments in the range the kernels use, device result vs
// host result, bitwise.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/tools/ulp_probe.hip -o /tmp/ulp_probe && /tmp/ulp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ void k_ops(int n, const double* a, const double* b, double* o_exp, double* o_log, double* o_sqrt, double* o_div, double* o_fma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_exp[i] = exp(a[i]);
    o_log[i] = log(b[i]);
    o_sqrt[i] = sqrt(b[i]);
    o_div[i] = a[i] / b[i];
    o_fma[i] = fma(a[i], b[i], a[i]);
}

int main() {
    const int n = 1 << 22;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> ea(-80.0, 0.0), lb(1e-6, 4.0);
    std::vector<double> a(n), b(n);
    for (int i = 0; i < n; i++) { a[i] = ea(rng); b[i] = lb(rng); }      // exp: -0.5 (x-mu)^2/var and -lambda x; log: products of <= 8 scales in [0.5,1)^8
    double *da, *db, *d[5];
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    for (auto& p : d) hipMalloc(&p, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ops, dim3(n / 256), dim3(256), 0, 0, n, da, db, d[0], d[1], d[2], d[3], d[4]);
    std::vector<double> r(n);
    const char* names[5] = {"exp", "log", "sqrt", "div", "fma"};
    for (int k = 0; k < 5; k++) {
        hipMemcpy(r.data(), d[k], n * 8, hipMemcpyDeviceToHost);
        long diff = 0, diff2 = 0;
        for (int i = 0; i < n; i++) {
            const double h = k == 0 ? std::exp(a[i]) : k == 1 ? std::log(b[i]) : k == 2 ? std::sqrt(b[i]) : k == 3 ? a[i] / b[i] : std::fma(a[i], b[i], a[i]);
            long long x, y;
            std::memcpy(&x, &h, 8); std::memcpy(&y, &r[i], 8);
            if (x != y) { diff++; if (llabs(x - y) > 1) diff2++; }
        }
        std::printf("%-5s device != host in %ld of %d arguments (%.4f %%), more than 1 ulp apart: %ld\n", names[k], diff, n, 100.0 * diff / n, diff2);
    }
    return 0;
}
