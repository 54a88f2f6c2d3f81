// Does VGPR index mode (s_set_gpr_idx_on, GFX9) select a float64 operand pair by a wave-uniform scalar on gfx950, inside inline asm?
// (round 5: the table-driven Viterbi kernel wants "branch metric of code c_j" per butterfly with c_j in a scalar register and NO extra
// VALU instruction.)   hipcc --offload-arch=gfx950 -O3 gpr_idx_check.hip -o gpr_idx_check && ./gpr_idx_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *o, const double *in, const int *sel) {
    d4 T = {in[0], in[1], in[2], in[3]};
    const double a = in[4 + threadIdx.x];
    double r0, r1;
    const int ic = __builtin_amdgcn_readfirstlane(sel[blockIdx.x]) * 2, ix = ic ^ 6;
    asm volatile("s_set_gpr_idx_on %[ic], 1\n\t"
                 "s_nop 0\n\t"
                 "v_add_f64 %[r0], v[248:249], %[a]\n\t"
                 "s_set_gpr_idx_idx %[ix]\n\t"
                 "s_nop 0\n\t"
                 "v_add_f64 %[r1], v[248:249], %[a]\n\t"
                 "s_set_gpr_idx_off"
                 : [r0] "=&v"(r0), [r1] "=&v"(r1)
                 : [ic] "s"(ic), [ix] "s"(ix), [a] "v"(a), "{v[248:255]}"(T));
    o[(blockIdx.x * 64 + threadIdx.x) * 2] = r0;
    o[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = r1;
}
int main() {
    double h_in[4 + 64], *d_in, *d_o, h_o[4 * 64 * 2];
    int h_sel[4] = {0, 1, 2, 3}, *d_sel;
    const double bm[4] = {0.125, 10.5, 200.25, 3000.0625};
    for (int i = 0; i < 4; i++) h_in[i] = bm[i];
    for (int i = 0; i < 64; i++) h_in[4 + i] = 1e-3 * i;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_o, sizeof(h_o)); hipMalloc(&d_sel, sizeof(h_sel));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice); hipMemcpy(d_sel, h_sel, sizeof(h_sel), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d_o, d_in, d_sel);
    hipMemcpy(h_o, d_o, sizeof(h_o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 4; c++)
        for (int l = 0; l < 64; l++) {
            const double w0 = bm[c] + 1e-3 * l, w1 = bm[c ^ 3] + 1e-3 * l;
            if (h_o[(c * 64 + l) * 2] != w0 || h_o[(c * 64 + l) * 2 + 1] != w1) bad++;
        }
    printf("gpr index mode on float64 pairs: %d mismatches of 256 (c = 0..3 -> bm[c], bm[c ^ 3])\n", bad);
    return bad != 0;
}
