// Accuracy check of cpx::fast_log against the host's long-double logl: max error in ulps over random and edge inputs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I commpy_amd/csrc scripts/micro/fastlog_check.hip -o /tmp/flc && /tmp/flc
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "cpx_math.h"
__global__ void k(const double *a, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = cpx::fast_log(a[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), y(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; i++) {
        const int kind = i & 7;
        if (kind == 0) x[i] = 1.0 + (u(g) - 0.5) * 1e-3;                 // around 1 (cancellation zone)
        else if (kind == 1) x[i] = std::ldexp(0.5 + 0.5 * u(g), (int)(u(g) * 2040) - 1020);   // whole exponent range
        else if (kind == 2) x[i] = 4.9e-324 * (1 + (long long)(u(g) * 1e6));                  // denormals
        else if (kind == 3) x[i] = 0.70710678 + (u(g) - 0.5) * 1e-6;     // reduction boundary
        else x[i] = std::exp((u(g) - 0.5) * 40.0);
    }
    x[0] = 0.0; x[1] = INFINITY; x[2] = -1.0; x[3] = NAN; x[4] = 1.0; x[5] = -0.0;
    double *da, *dob;
    hipMalloc(&da, n * 8); hipMalloc(&dob, n * 8);
    hipMemcpy(da, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, dob, n);
    hipMemcpy(y.data(), dob, n * 8, hipMemcpyDeviceToHost);
    double worst = 0; int wi = -1; long bad = 0;
    for (int i = 6; i < n; i++) {
        const long double r = logl((long double)x[i]);
        const double rd = (double)r;
        double ulp = std::nextafter(std::fabs(rd), INFINITY) - std::fabs(rd);
        if (ulp == 0) ulp = 4.9e-324;
        const double err = (double)(fabsl((long double)y[i] - r) / ulp);
        if (err > worst) { worst = err; wi = i; }
        if (err > 1.0) bad++;
    }
    printf("special: log(0)=%g log(inf)=%g log(-1)=%g log(nan)=%g log(1)=%g log(-0)=%g\n", y[0], y[1], y[2], y[3], y[4], y[5]);
    printf("max error %.3f ulp at x=%.17g (got %.17g), %ld of %d above 1 ulp\n", worst, x[wi], y[wi], bad, n);
    return 0;
}
