// Micro-benchmark: ds_bpermute_b32 throughput per CU for different source-lane patterns (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(const int *pat, int *out, int iters) {
    const int lane = threadIdx.x;
    int a0 = pat[lane] << 2, v0 = lane * 3 + 1, v1 = lane * 5 + 2, v2 = lane * 7 + 3, v3 = lane * 11 + 5;
    for (int i = 0; i < iters; i++) {
        v0 = __builtin_amdgcn_ds_bpermute(a0, v0);
        v1 = __builtin_amdgcn_ds_bpermute(a0, v1);
        v2 = __builtin_amdgcn_ds_bpermute(a0, v2);
        v3 = __builtin_amdgcn_ds_bpermute(a0, v3);
    }
    out[blockIdx.x * 64 + lane] = v0 ^ v1 ^ v2 ^ v3;
}

int main() {
    const int nblocks = 256 * 32, iters = 2000;
    int *d_pat, *d_out;
    CHECK(hipMalloc(&d_pat, 64 * 4));
    CHECK(hipMalloc(&d_out, nblocks * 64 * 4));
    struct { const char *name; int (*f)(int); } pats[] = {
        {"identity", [](int l) { return l; }},
        {"xor1", [](int l) { return l ^ 1; }},
        {"xor32", [](int l) { return l ^ 32; }},
        {"shuffle 2*(l&31)   [viterbi pred0]", [](int l) { return 2 * (l & 31); }},
        {"shuffle 2*(l&31)+1 [viterbi pred1]", [](int l) { return 2 * (l & 31) + 1; }},
        {"rotated: ((l&15)<<1)|(l>>5) -> src lane (j=0)", [](int l) { return ((l & 15) << 1) | (l >> 5); }},
        {"rotated pred1: 32 | ((l&15)<<1)|(l>>5)", [](int l) { return 32 | ((l & 15) << 1) | (l >> 5); }},
        {"all lane 0 (broadcast)", [](int l) { return 0; }},
        {"stride 32: (l*32)%64 + l/2%32", [](int l) { return ((l & 1) << 5) | (l >> 1); }},
    };
    for (auto &p : pats) {
        std::vector<int> h(64);
        for (int l = 0; l < 64; l++) h[l] = p.f(l);
        CHECK(hipMemcpy(d_pat, h.data(), 256, hipMemcpyHostToDevice));
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d_pat, d_out, iters);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d_pat, d_out, iters);
        hipEventRecord(b);
        CHECK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        double instr_per_cu = (double)nblocks / 256 * iters * 4;
        printf("%-52s %8.3f ms  -> %.2f LDS-cycles per bpermute per CU @2.4GHz\n", p.name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
    return 0;
}
