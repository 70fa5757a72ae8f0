// Measures the relative error of v_rcp_f64 with 0/1/2 Newton steps against IEEE division on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/rcp_accuracy.hip -o gpurun_out/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* e0, double* e1, double* e2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], ex = 1.0 / v;
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    e0[i] = fabs(r0 - ex) / ex; e1[i] = fabs(r1 - ex) / ex; e2[i] = fabs(r2 - ex) / ex;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = exp(u * 70.0); }
    double *x, *e0, *e1, *e2;
    hipMalloc(&x, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8); hipMalloc(&e2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, e0, e1, e2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), e0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), e1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), e2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; i++) { m0 = fmax(m0, a[i]); m1 = fmax(m1, b[i]); m2 = fmax(m2, c[i]); }
    printf("v_rcp_f64 max rel err: raw %.3e  1 NR %.3e  2 NR %.3e  (x in [1, e^70], %d samples)\n", m0, m1, m2, n);
    return 0;
}
