// FETCH_SIZE / WRITE_SIZE calibration (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
// Each kernel moves EXACTLY `bytes` bytes once, far beyond the caches (1 GiB), in one of the access patterns of the decoders:
//   rd16  16 B per lane, consecutive lanes (viterbi_cw_fused_kernel's LLR pairs, the demodulator's complex128 symbols)
//   rd8    8 B per lane, consecutive lanes (LDPC block rows, turbo stage kernels)
//   rd8seg 8 B per lane through raw buffer loads, eight lanes = one 64-byte segment, every 8-lane block in another row
//          (turbo_pass_kernel / map_decode_kernel: steps of a chunk x codewords of a pair)
//   rd4 / rd1  4 / 1 byte per lane
//   wr16 / wr8 / wr1  stores of 16 / 8 / 1 byte per lane, consecutive lanes;  wr8seg: 64-byte segments in rows like rd8seg
// Run under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE in its own pass); scripts/micro/fetch_calib.py turns the per-kernel
// counter values into factors  known bytes / reported bytes.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib scripts/micro/fetch_calib.hip && /tmp/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rd16(const double2 *p, int64_t n, double *sink) {
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double2 v = p[i]; a += v.x + v.y; }
    if (a == 1.2345e300) *sink = a;
}
__global__ __launch_bounds__(256) void rd8(const double *p, int64_t n, double *sink) {
    double a = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += p[i];
    if (a == 1.2345e300) *sink = a;
}
__global__ __launch_bounds__(256) void rd4(const float *p, int64_t n, double *sink) {
    float a = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += p[i];
    if (a == 1.2345e30f) *sink = a;
}
__global__ __launch_bounds__(256) void rd1(const uint8_t *p, int64_t n, double *sink) {
    unsigned a = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += p[i];
    if (a == 0xdeadbeefu) *sink = a;
}
// rows of `row` doubles; a wavefront reads chunk c (8 doubles) of 8 consecutive rows: lane = r * 8 + t
__global__ __launch_bounds__(256) void rd8seg(const double *p, int64_t nrows, int row, double *sink) {
    const int lane = threadIdx.x & 63, r = lane >> 3, t = lane & 7;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    double a = 0;
    for (int64_t g = wave; g < nrows / 8; g += nw) {
        const double *base = p + (g * 8 + r) * row;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, (unsigned)(row * 8), 0x00020000);
        for (int c = 0; c < row / 8; c++)
            a += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)((c * 8 + t) * 8), 0, 0));
    }
    if (a == 1.2345e300) *sink = a;
}
__global__ __launch_bounds__(256) void wr16(double2 *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(256) void wr8(double *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 3.0;
}
__global__ __launch_bounds__(256) void wr1(uint8_t *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 1;
}
__global__ __launch_bounds__(256) void wr8seg(double *p, int64_t nrows, int row) {
    const int lane = threadIdx.x & 63, r = lane >> 3, t = lane & 7;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t g = wave; g < nrows / 8; g += nw) {
        double *base = p + (g * 8 + r) * row;
        for (int c = 0; c < row / 8; c++) base[c * 8 + t] = 4.0;
    }
}

int main() {
    const int64_t bytes = (int64_t)1 << 30;
    void *buf = nullptr;
    double *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc((void **)&sink, 8));
    CK(hipMemset(buf, 0, bytes));
    const dim3 grid(256 * 8), block(256);
    const int row = 1024;                                 // doubles per row of the segment patterns (a codeword's 8 KB array)
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(rd16, grid, block, 0, 0, (const double2 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(rd8, grid, block, 0, 0, (const double *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(rd4, grid, block, 0, 0, (const float *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(rd1, grid, block, 0, 0, (const uint8_t *)buf, bytes, sink);
        hipLaunchKernelGGL(rd8seg, grid, block, 0, 0, (const double *)buf, bytes / 8 / row, row, sink);
        hipLaunchKernelGGL(wr16, grid, block, 0, 0, (double2 *)buf, bytes / 16);
        hipLaunchKernelGGL(wr8, grid, block, 0, 0, (double *)buf, bytes / 8);
        hipLaunchKernelGGL(wr1, grid, block, 0, 0, (uint8_t *)buf, bytes);
        hipLaunchKernelGGL(wr8seg, grid, block, 0, 0, (double *)buf, bytes / 8 / row, row);
    }
    CK(hipDeviceSynchronize());
    printf("fetch_calib: 9 kernels x 2, %lld bytes each\n", (long long)bytes);
    return 0;
}
