// PSK/QAM demodulation for gfx950.  Replaces the body of Modem.demodulate
// (/root/reference/commpy/modulation.py:100-141):
//   soft (:125-137): LLR(b) = log( sum_{m:(m>>b)&1} e^{-|y-c_m|^2/noise_var} / sum_{m:!(..)} ... ),
//                    written at index i*nb + nb-1-b, sums in increasing constellation index m;
//   hard (:121-123): first-minimum nearest point |y-c_m| -> MSB-first nb bits (int8).
// Element-wise map: one received symbol per lane, constellation (<= 256 points, 4 KiB) staged in LDS
// and read as wave-uniform broadcasts.  float64 like the reference; |.| is hypot like np.abs of a
// complex scalar.  Each symbol is read once (16 B) and nb outputs written once: HBM traffic equals
// the algorithmic bytes (16 + 8*nb per symbol); the naive float64 formula costs M exps per symbol.
#include "cpx_internal.h"

using namespace cpx;

namespace {

constexpr int DEMOD_BLOCK = 256;
constexpr int MAX_M = 256;
constexpr int MAX_NB = 8;

template <int NB>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                 const double2 *__restrict__ cst, int M,
                                                                 double noise_var, double *__restrict__ llr) {
    __shared__ double2 c_s[MAX_M];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = cst[m];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const double2 cur = y[i];
        double num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
        for (int m = 0; m < M; m++) {
            const double2 c = c_s[m];
            const double a = hypot(cur.x - c.x, cur.y - c.y);       // abs(current_symbol - symbol)
            const double e = exp((-(a * a)) / noise_var);             // exp((-abs(..)**2)/noise_var) (:134,136)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) llr[i * NB + NB - 1 - b] = log(num[b] / den[b]);   // (:137)
    }
}

__global__ __launch_bounds__(DEMOD_BLOCK) void demod_hard_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                 const double2 *__restrict__ cst, int M, int nb,
                                                                 int8_t *__restrict__ bits) {
    __shared__ double2 c_s[MAX_M];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = cst[m];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const double2 cur = y[i];
        int best = 0;
        double bd = hypot(cur.x - c_s[0].x, cur.y - c_s[0].y);
        for (int m = 1; m < M; m++) {                                 // abs(y - c[:, None]).argmin(0): first minimum (:122)
            const double a = hypot(cur.x - c_s[m].x, cur.y - c_s[m].y);
            if (a < bd) { bd = a; best = m; }
        }
        for (int b = 0; b < nb; b++) bits[i * nb + b] = (int8_t)((best >> (nb - 1 - b)) & 1);   // dec2bitarray (:123)
    }
}

unsigned grid_for(int64_t Ns) {
    int64_t blocks = (Ns + DEMOD_BLOCK - 1) / DEMOD_BLOCK;
    const int64_t cap = 256 * 16;   // 256 CUs x 16 resident blocks; grid-stride beyond that
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int cpx_modem_create(const double *constellation_re_im, int M, cpx_modem **out) {
    CPX_REQUIRE(constellation_re_im && out, CPX_EINVAL, "cpx_modem_create: null pointer");
    CPX_REQUIRE(M >= 2 && (M & (M - 1)) == 0, CPX_EINVAL, "Constellation length must be a power of 2.");
    CPX_REQUIRE(M <= MAX_M, CPX_ELIMIT, "cpx_modem_create: constellations above %d points are not supported", MAX_M);
    int rc = ensure_device();
    if (rc) return rc;
    cpx_modem *m = new cpx_modem;
    m->M = M;
    m->nbits = 0;
    while ((1 << m->nbits) < M) m->nbits++;
    (void)hipGetDevice(&m->device);
    CPX_HIP(hipMalloc((void **)&m->d_const, sizeof(double) * 2 * M));
    CPX_HIP(hipMemcpy(m->d_const, constellation_re_im, sizeof(double) * 2 * M, hipMemcpyHostToDevice));
    *out = m;
    return CPX_OK;
}

int cpx_modem_destroy(cpx_modem *m) {
    if (!m) return CPX_OK;
    (void)hipFree(m->d_const);
    delete m;
    return CPX_OK;
}

int cpx_demod_soft_dev(const cpx_modem *m, const double *d_y, int64_t Ns, double noise_var, double *d_llr, void *stream) {
    CPX_REQUIRE(m, CPX_EINVAL, "demod: null modem");
    CPX_REQUIRE(Ns >= 0, CPX_EINVAL, "demod: negative size");
    if (Ns == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const double2 *y = reinterpret_cast<const double2 *>(d_y);
    const double2 *c = reinterpret_cast<const double2 *>(m->d_const);
    dim3 grid(grid_for(Ns)), block(DEMOD_BLOCK);
    switch (m->nbits) {
#define CASE(NB) case NB: hipLaunchKernelGGL(demod_soft_kernel<NB>, grid, block, 0, st, y, Ns, c, m->M, noise_var, d_llr); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
    }
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_demod_hard_dev(const cpx_modem *m, const double *d_y, int64_t Ns, int8_t *d_bits, void *stream) {
    CPX_REQUIRE(m, CPX_EINVAL, "demod: null modem");
    CPX_REQUIRE(Ns >= 0, CPX_EINVAL, "demod: negative size");
    if (Ns == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    dim3 grid(grid_for(Ns)), block(DEMOD_BLOCK);
    hipLaunchKernelGGL(demod_hard_kernel, grid, block, 0, st, reinterpret_cast<const double2 *>(d_y), Ns,
                       reinterpret_cast<const double2 *>(m->d_const), m->M, m->nbits, d_bits);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_demod_soft(const cpx_modem *m, const double *y_re_im, int64_t Ns, double noise_var, double *llr) {
    CPX_REQUIRE(m && (y_re_im || Ns == 0) && (llr || Ns == 0), CPX_EINVAL, "demod: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (Ns == 0) return CPX_OK;
    DevBuf din, dout;
    const size_t out_bytes = sizeof(double) * (size_t)Ns * m->nbits;
    if ((rc = din.alloc(sizeof(double) * 2 * (size_t)Ns))) return rc;
    if ((rc = dout.alloc(out_bytes))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, y_re_im, sizeof(double) * 2 * (size_t)Ns, hipMemcpyHostToDevice, st));
    if ((rc = cpx_demod_soft_dev(m, din.as<double>(), Ns, noise_var, dout.as<double>(), st))) return rc;
    CPX_HIP(hipMemcpyAsync(llr, dout.p, out_bytes, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_demod_hard(const cpx_modem *m, const double *y_re_im, int64_t Ns, int8_t *bits) {
    CPX_REQUIRE(m && (y_re_im || Ns == 0) && (bits || Ns == 0), CPX_EINVAL, "demod: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (Ns == 0) return CPX_OK;
    DevBuf din, dout;
    const size_t out_bytes = (size_t)Ns * m->nbits;
    if ((rc = din.alloc(sizeof(double) * 2 * (size_t)Ns))) return rc;
    if ((rc = dout.alloc(out_bytes))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, y_re_im, sizeof(double) * 2 * (size_t)Ns, hipMemcpyHostToDevice, st));
    if ((rc = cpx_demod_hard_dev(m, din.as<double>(), Ns, dout.as<int8_t>(), st))) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dout.p, out_bytes, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

}  // extern "C"
