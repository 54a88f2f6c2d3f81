// Micro-probe (not part of the product): where does the dispatcher put the wavefronts of a grid that has about one
// wavefront per SIMD?  Every wave records HW_ID / XCC_ID, spins for ~200 us so that all of them are resident together,
// and the host prints how many SIMDs hold 0, 1, 2, ... waves.
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BIGREGS>
__global__ void probe(unsigned *out, long long spin) {
    extern __shared__ unsigned char lds[];
    if (BIGREGS) asm volatile("v_mov_b32 v250, 0" ::: "v250");     // forces a 256-VGPR allocation
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2;
        out[w] = hw; out[w + 1] = xcc;
        if (lds && threadIdx.x == 100000) lds[0] = 1;
    }
}

template <int BIGREGS>
void run(const char *name, int blocks, int threads, size_t lds) {
    const int waves = blocks * threads / 64;
    unsigned *d;
    CK(hipMalloc(&d, waves * 8));
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)probe<BIGREGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(probe<BIGREGS>, dim3(blocks), dim3(threads), lds, 0, d, 2000000LL);   // 0.8 ms at shader clock (20 ms if the counter runs at 100 MHz)
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(waves * 2);
    CK(hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost));
    std::map<unsigned long long, int> per_simd, per_cu;
    for (int w = 0; w < waves; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned long long cukey = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++;
        per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    printf("%-46s waves %5d | CUs used %3zu, SIMDs used %4zu | waves per SIMD:", name, waves, per_cu.size(), per_simd.size());
    for (auto &kv : hist_simd) printf(" %dx:%d", kv.first, kv.second);
    printf(" | waves per CU:");
    for (auto &kv : hist_cu) printf(" %dx:%d", kv.first, kv.second);
    printf("\n");
    CK(hipFree(d));
}

int main() {
    run<1>("1024 x 64 thr, 256 VGPR, no LDS", 1024, 64, 0);
    run<0>("1024 x 64 thr, few VGPR, no LDS", 1024, 64, 0);
    run<0>("1024 x 64 thr, few VGPR, 37 KB LDS", 1024, 64, 37 * 1024);
    run<1>("256 x 256 thr, 256 VGPR, no LDS", 256, 256, 0);
    run<0>("512 x 128 thr, few VGPR, 74 KB LDS", 512, 128, 74 * 1024);
    run<0>("256 x 256 thr, few VGPR, 148 KB LDS", 256, 256, 148 * 1024);
    run<0>("2048 x 64 thr, few VGPR, 37 KB LDS", 2048, 64, 37 * 1024);
    return 0;
}
