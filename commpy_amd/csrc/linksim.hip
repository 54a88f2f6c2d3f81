// Device-side link-simulation stages around the decoders ("next" rows of SURVEY 8f): the transmit
// chain and channel of the reference's Monte-Carlo loops, batched, so that a BER sweep never leaves HBM.
//   conv_encode        /root/reference/commpy/channelcoding/convcode.py:475-558 (table walk, both terminations)
//   puncturing         convcode.py:752-774  / depuncturing :777-804 (as index gathers built by the host)
//   Modem.modulate     /root/reference/commpy/modulation.py:79-98 (MSB-first label -> constellation point)
//   random messages    commpy/links.py:229 (np.random.choice((0,1), n)) -- Philox4x32-10 stream instead of MT19937
//   AWGN               commpy/channels.py:37-55 (noise = (randn + 1j*randn) * scale per component, or real)
//   BSC / BEC          commpy/channels.py:630-673 (one uniform draw per bit; flipped / erased to -1 where it is <= p) -- round 5
//   error counting     links.py:252-256 (per-chunk XOR popcount)
// Bit-exact stages (encode, (de)puncture, modulate, error count) are tested against the host mirror and
// the reference goldens; the random stages are statistical (different generator than the reference).
// All kernels are byte/element-wise streams: HBM bound, one codeword row or one element per lane.
#include "cpx_internal.h"
#include "cpx_rng.h"

using namespace cpx;

namespace {

constexpr int LS_BLOCK = 256;

unsigned ls_grid(int64_t n) {
    int64_t blocks = (n + LS_BLOCK - 1) / LS_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// ---- random message bits ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LS_BLOCK) void random_bits_kernel(uint8_t *__restrict__ bits, int64_t n, uint64_t seed,
                                                               uint64_t stream) {
    // one Philox call = 128 random bits -> 128 output bytes? keep it simple: 16 bytes (one bit each) per call uses
    // 16 of the 128 bits; throughput is irrelevant next to the decoders
    const int64_t n16 = (n + 15) / 16;
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * LS_BLOCK) {
        const Philox r = philox4x32_10((uint64_t)i, stream, seed);
        const uint32_t w = r.c[0];
        for (int j = 0; j < 16; j++) {
            const int64_t pos = i * 16 + j;
            if (pos < n) bits[pos] = (uint8_t)((w >> j) & 1u);
        }
    }
}

// ---- convolutional encoder: one codeword per lane, table walk (convcode.py:531-550) -----------------------------
// msg [B][nmsg] uint8 -> coded [B][nout] uint8.  term: 0 'cont', 1 'term'.  rsc: code_type == 'rsc'.
__global__ __launch_bounds__(LS_BLOCK) void conv_encode_kernel(const uint8_t *__restrict__ msg, int64_t B, int64_t nmsg,
                                                               const int32_t *__restrict__ next_state,
                                                               const int32_t *__restrict__ output, int k, int n, int S,
                                                               int I, int total_memory, int term, int rsc,
                                                               uint8_t *__restrict__ coded, int64_t nout) {
    extern __shared__ int32_t tabs[];
    int32_t *nx = tabs, *ot = tabs + S * I;
    for (int i = threadIdx.x; i < S * I; i += LS_BLOCK) { nx[i] = next_state[i]; ot[i] = output[i]; }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x;
    if (b >= B) return;
    const uint8_t *m = msg + b * nmsg;
    uint8_t *c = coded + b * nout;
    // number of input bits actually clocked in (:505-520): message, plus the zero tail for non-recursive 'term'
    int64_t ninb = nmsg;
    if (term && !rsc) ninb = nmsg + total_memory + total_memory % k;
    int state = 0;
    int64_t j = 0;
    for (int64_t i = 0; i < ninb / k; i++) {
        int cur = 0;
        for (int q = 0; q < k; q++) {
            const int64_t pos = i * k + q;
            cur = (cur << 1) | ((pos < nmsg) ? (m[pos] & 1) : 0);   // bitarray2dec: MSB first (:533)
        }
        const int o = ot[state * I + cur];
        for (int q = 0; q < n; q++)
            if (j * n + q < nout) c[j * n + q] = (uint8_t)((o >> (n - 1 - q)) & 1);   // dec2bitarray(o, n) (:535)
        state = nx[state * I + cur];
        j++;
    }
    if (rsc && term) {                                            // (:538-546): feed the register content back, LSB first
        const int tbits = state;                                  // term_bits = dec2bitarray(state, m)[::-1]
        for (int i = 0; i < total_memory; i++) {
            int cur = 0;
            for (int q = 0; q < k; q++) {
                const int pos = i * k + q;
                const int bit = (pos < total_memory) ? ((tbits >> pos) & 1) : 0;
                if (pos < total_memory) cur = (cur << 1) | bit;
            }
            const int o = ot[state * I + cur];
            for (int q = 0; q < n; q++)
                if (j * n + q < nout) c[j * n + q] = (uint8_t)((o >> (n - 1 - q)) & 1);
            state = nx[state * I + cur];
            j++;
        }
    }
    for (int64_t q = j * n; q < nout; q++) c[q] = 0;
}

// Feed-forward shift-register codes (k = 1, next state = (b << (m-1)) | (state >> 1), the trellises commpy builds
// for non-recursive codes): the state at step j is just the previous m message bits, so every trellis step is
// independent -- one thread per (codeword, step) instead of one per codeword (the sequential walk above took longer
// than the Viterbi decoder of the same batch in the link benchmark).
__global__ __launch_bounds__(LS_BLOCK) void conv_encode_ff_kernel(const uint8_t *__restrict__ msg, int64_t B, int64_t nmsg,
                                                                  const int32_t *__restrict__ output, int n, int m,
                                                                  int64_t nsteps, uint8_t *__restrict__ coded,
                                                                  int64_t nout) {
    const int64_t spc = (nout + n - 1) / n;                        // step slots per codeword (zero fill past nsteps)
    for (int64_t idx = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; idx < B * spc; idx += (int64_t)gridDim.x * LS_BLOCK) {
        const int64_t b = idx / spc, j = idx - b * spc;
        int o = 0;
        if (j < nsteps) {
            const uint8_t *mb = msg + b * nmsg;
            int state = 0;
            for (int i = 1; i <= m; i++) {                         // most recent bit = MSB of the state
                const int64_t pos = j - i;
                const int bit = (pos >= 0 && pos < nmsg) ? (mb[pos] & 1) : 0;
                state |= bit << (m - i);
            }
            const int cur = (j < nmsg) ? (mb[j] & 1) : 0;          // zero tail of 'term' (:505-520)
            o = output[state * 2 + cur];
        }
        uint8_t *c = coded + b * nout + j * n;
        for (int q = 0; q < n; q++)
            if (j * n + q < nout) c[q] = (uint8_t)((o >> (n - 1 - q)) & 1);   // dec2bitarray(o, n) (:535)
    }
}

// ---- row-wise gathers: puncturing (u8) and depuncturing (f64, -1 = punctured position -> 0.0) ----------------------
__global__ __launch_bounds__(LS_BLOCK) void gather_u8_kernel(const uint8_t *__restrict__ in, int64_t B, int64_t nin,
                                                             const int32_t *__restrict__ idx, int64_t nout,
                                                             uint8_t *__restrict__ out) {
    const int64_t total = B * nout;
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * LS_BLOCK) {
        const int64_t b = i / nout, j = i % nout;
        out[i] = in[b * nin + idx[j]];
    }
}

__global__ __launch_bounds__(LS_BLOCK) void gather_f64_kernel(const double *__restrict__ in, int64_t B, int64_t nin,
                                                              const int32_t *__restrict__ idx, int64_t nout,
                                                              double *__restrict__ out) {
    const int64_t total = B * nout;
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * LS_BLOCK) {
        const int64_t b = i / nout, j = i % nout;
        const int32_t src = idx[j];
        out[i] = (src >= 0) ? in[b * nin + src] : 0.0;
    }
}

// ---- modulate: nb bits (MSB first) -> constellation[label]  (modulation.py:93-96) -----------------------------------
__global__ __launch_bounds__(LS_BLOCK) void modulate_kernel(const uint8_t *__restrict__ bits, int64_t nsym, int nb,
                                                            const double2 *__restrict__ cst, int M,
                                                            double2 *__restrict__ sym) {
    __shared__ double2 c_s[256];
    const bool staged = M <= 256;                                 // larger tables (up to 65536 points) are gathered from HBM / L2
    if (staged) for (int m = threadIdx.x; m < M; m += LS_BLOCK) c_s[m] = cst[m];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < nsym; i += (int64_t)gridDim.x * LS_BLOCK) {
        int label = 0;
        for (int q = 0; q < nb; q++) label = (label << 1) | (bits[i * nb + q] & 1);
        sym[i] = staged ? c_s[label] : cst[label];
    }
}

// ---- AWGN: y = x + scale * (n_re + 1j * n_im), n ~ N(0,1) i.i.d. (Box-Muller on Philox) ------------------------------
__global__ __launch_bounds__(LS_BLOCK) void awgn_kernel(const double2 *__restrict__ x, int64_t n, double scale_re,
                                                        double scale_im, uint64_t seed, uint64_t stream,
                                                        double2 *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * LS_BLOCK) {
        const double2 v = awgn_add(x[i], (uint64_t)i, scale_re, scale_im, seed, stream);
        y[i] = v;
    }
}

// ---- binary symmetric / binary erasure channel (channels.py:630-673) ---------------------------------------------------
// One uniform draw per bit (53 bits, two per Philox call) compared with `<= p` like the reference's `random(n) <= p`:
// ERASE = false: out = in ^ hit (bsc, :652-673);  ERASE = true: out = hit ? -1 : in (bec, :630-649).  Either output may be null:
// int8 (the reference's integer bits, -1 = erasure) and / or float64 (what `viterbi_decode(..., 'hard')` is handed).
template <bool ERASE>
__global__ __launch_bounds__(LS_BLOCK) void binary_channel_kernel(const uint8_t *__restrict__ in, int64_t n, double p, uint64_t seed,
                                                                  uint64_t stream, int8_t *__restrict__ out_i8,
                                                                  double *__restrict__ out_f64) {
    const int64_t n2 = (n + 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * LS_BLOCK) {
        const Philox r = philox4x32_10((uint64_t)i, stream, seed);
        // u01 is uniform on (0, 1] in steps of 2^-53; `u - 2^-53 <= p` is the reference's [0, 1) draw compared with `<= p`
        const double u[2] = {u01(r.c[0], r.c[1]) - 1.0 / 9007199254740992.0, u01(r.c[2], r.c[3]) - 1.0 / 9007199254740992.0};
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t pos = 2 * i + j;
            if (pos >= n) break;
            const int b = in[pos] & 1;
            const bool hit = u[j] <= p;
            const int o = ERASE ? (hit ? -1 : b) : (b ^ (hit ? 1 : 0));
            if (out_i8) out_i8[pos] = (int8_t)o;
            if (out_f64) out_f64[pos] = (double)o;
        }
    }
}

// ---- error counting: errs[b][c] = popcount(msg[b, c*chunk:(c+1)*chunk] ^ dec[b, same]) (links.py:252-256) --------------
__global__ __launch_bounds__(LS_BLOCK) void count_errors_kernel(const uint8_t *__restrict__ msg, int64_t msg_stride,
                                                                const uint8_t *__restrict__ dec, int64_t dec_stride,
                                                                int64_t B, int64_t nchunks, int64_t chunk,
                                                                int32_t *__restrict__ errs) {
    // one wavefront per (block, chunk): coalesced byte reads, butterfly sum (a thread per chunk walked its bytes
    // alone and took 12 % of the link benchmark)
    const int lane = threadIdx.x & 63;
    const int64_t total = B * nchunks, nwaves = (int64_t)gridDim.x * (LS_BLOCK / 64);
    for (int64_t i = (int64_t)blockIdx.x * (LS_BLOCK / 64) + (threadIdx.x >> 6); i < total; i += nwaves) {
        const int64_t b = i / nchunks, c = i % nchunks;
        const uint8_t *m = msg + b * msg_stride + c * chunk, *d = dec + b * dec_stride + c * chunk;
        int32_t e = 0;
        for (int64_t q = lane; q < chunk; q += 64) e += (m[q] ^ d[q]) & 1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
        if (lane == 0) errs[i] = e;
    }
}

// y = a * x (e.g. the LLR sign flip between Modem.demodulate, log P1/P0, and ldpc_bp_decode, log P0/P1: quirk B6)
__global__ __launch_bounds__(LS_BLOCK) void scale_f64_kernel(const double *__restrict__ x, int64_t n, double a,
                                                             double *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * LS_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * LS_BLOCK) y[i] = a * x[i];
}

}  // namespace

extern "C" {

int cpx_scale_f64_dev(const double *d_x, int64_t n, double a, double *d_y, void *stream) {
    CPX_REQUIRE(n >= 0, CPX_EINVAL, "scale: negative size");
    if (n == 0) return CPX_OK;
    hipLaunchKernelGGL(scale_f64_kernel, dim3(ls_grid(n)), dim3(LS_BLOCK), 0, pick_stream(stream), d_x, n, a, d_y);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_random_bits_dev(uint8_t *d_bits, int64_t n, uint64_t seed, uint64_t stream_id, void *stream) {
    CPX_REQUIRE(n >= 0, CPX_EINVAL, "random_bits: negative size");
    if (n == 0) return CPX_OK;
    hipLaunchKernelGGL(random_bits_kernel, dim3(ls_grid((n + 15) / 16)), dim3(LS_BLOCK), 0, pick_stream(stream), d_bits, n,
                       seed, stream_id);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_conv_encode_batch_dev(const cpx_trellis *t, const uint8_t *d_msg, int64_t B, int64_t nmsg, int terminate, int rsc,
                              uint8_t *d_coded, int64_t nout, void *stream) {
    CPX_TRACE("cpx_conv_encode_batch_dev");
    CPX_REQUIRE(t, CPX_EINVAL, "conv_encode: null trellis");
    if (int rcd = check_handle_device(t->device, "conv_encode")) return rcd;
    CPX_REQUIRE(B >= 0 && nmsg >= 0 && nout >= 0, CPX_EINVAL, "conv_encode: negative size");
    if (B == 0 || nout == 0) return CPX_OK;
    int total_memory = 0;
    while ((1 << total_memory) < t->S) total_memory++;
    bool ff = (t->k == 1 && t->I == 2 && !rsc && total_memory >= 1);
    for (int s2 = 0; s2 < t->S && ff; s2++)
        for (int b2 = 0; b2 < 2; b2++)
            if (t->next_state[s2 * 2 + b2] != ((b2 << (total_memory - 1)) | (s2 >> 1))) ff = false;
    if (ff) {
        const int64_t nsteps = nmsg + (terminate ? total_memory : 0);
        const int64_t spc = (nout + t->n - 1) / t->n;
        hipLaunchKernelGGL(conv_encode_ff_kernel, dim3(ls_grid(B * spc)), dim3(LS_BLOCK), 0, pick_stream(stream), d_msg, B,
                           nmsg, t->d_out, t->n, total_memory, nsteps, d_coded, nout);
        CPX_HIP(hipGetLastError());
        return CPX_OK;
    }
    const size_t lds = sizeof(int32_t) * 2 * t->S * t->I;
    hipLaunchKernelGGL(conv_encode_kernel, dim3((unsigned)((B + LS_BLOCK - 1) / LS_BLOCK)), dim3(LS_BLOCK), lds,
                       pick_stream(stream), d_msg, B, nmsg, t->d_next, t->d_out, t->k, t->n, t->S, t->I, total_memory,
                       terminate ? 1 : 0, rsc ? 1 : 0, d_coded, nout);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_gather_u8_dev(const uint8_t *d_in, int64_t B, int64_t nin, const int32_t *d_idx, int64_t nout, uint8_t *d_out,
                      void *stream) {
    CPX_REQUIRE(B >= 0 && nin >= 0 && nout >= 0, CPX_EINVAL, "gather: negative size");
    if (B * nout == 0) return CPX_OK;
    hipLaunchKernelGGL(gather_u8_kernel, dim3(ls_grid(B * nout)), dim3(LS_BLOCK), 0, pick_stream(stream), d_in, B, nin, d_idx,
                       nout, d_out);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_gather_f64_dev(const double *d_in, int64_t B, int64_t nin, const int32_t *d_idx, int64_t nout, double *d_out,
                       void *stream) {
    CPX_REQUIRE(B >= 0 && nin >= 0 && nout >= 0, CPX_EINVAL, "gather: negative size");
    if (B * nout == 0) return CPX_OK;
    hipLaunchKernelGGL(gather_f64_kernel, dim3(ls_grid(B * nout)), dim3(LS_BLOCK), 0, pick_stream(stream), d_in, B, nin,
                       d_idx, nout, d_out);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_modulate_dev(const cpx_modem *m, const uint8_t *d_bits, int64_t nsym, double *d_sym_re_im, void *stream) {
    CPX_REQUIRE(m, CPX_EINVAL, "modulate: null modem");
    if (int rcd = check_handle_device(m->device, "modulate")) return rcd;
    CPX_REQUIRE(nsym >= 0, CPX_EINVAL, "modulate: negative size");
    if (nsym == 0) return CPX_OK;
    hipLaunchKernelGGL(modulate_kernel, dim3(ls_grid(nsym)), dim3(LS_BLOCK), 0, pick_stream(stream), d_bits, nsym, m->nbits,
                       reinterpret_cast<const double2 *>(m->d_const), m->M, reinterpret_cast<double2 *>(d_sym_re_im));
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_awgn_dev(const double *d_x_re_im, int64_t n, double scale_re, double scale_im, uint64_t seed, uint64_t stream_id,
                 double *d_y_re_im, void *stream) {
    CPX_REQUIRE(n >= 0, CPX_EINVAL, "awgn: negative size");
    if (n == 0) return CPX_OK;
    hipLaunchKernelGGL(awgn_kernel, dim3(ls_grid(n)), dim3(LS_BLOCK), 0, pick_stream(stream),
                       reinterpret_cast<const double2 *>(d_x_re_im), n, scale_re, scale_im, seed, stream_id,
                       reinterpret_cast<double2 *>(d_y_re_im));
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

static int binary_channel(bool erase, const uint8_t *d_bits, int64_t n, double p, uint64_t seed, uint64_t stream_id,
                          int8_t *d_out_i8, double *d_out_f64, void *stream) {
    const char *who = erase ? "bec" : "bsc";
    CPX_REQUIRE(n >= 0, CPX_EINVAL, "%s: negative size", who);
    CPX_REQUIRE(p >= 0.0 && p <= 1.0, CPX_EINVAL, "%s: probability %g outside [0, 1]", who, p);   // (also refuses NaN)
    if (n == 0) return CPX_OK;
    CPX_REQUIRE(d_bits && (d_out_i8 || d_out_f64), CPX_EINVAL, "%s: null pointer", who);
    const dim3 grid(ls_grid((n + 1) / 2)), block(LS_BLOCK);
    if (erase) hipLaunchKernelGGL(binary_channel_kernel<true>, grid, block, 0, pick_stream(stream), d_bits, n, p, seed, stream_id, d_out_i8, d_out_f64);
    else hipLaunchKernelGGL(binary_channel_kernel<false>, grid, block, 0, pick_stream(stream), d_bits, n, p, seed, stream_id, d_out_i8, d_out_f64);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_bsc_dev(const uint8_t *d_bits, int64_t n, double p_t, uint64_t seed, uint64_t stream_id, int8_t *d_out_i8,
                double *d_out_f64, void *stream) {
    return binary_channel(false, d_bits, n, p_t, seed, stream_id, d_out_i8, d_out_f64, stream);
}

int cpx_bec_dev(const uint8_t *d_bits, int64_t n, double p_e, uint64_t seed, uint64_t stream_id, int8_t *d_out_i8,
                double *d_out_f64, void *stream) {
    return binary_channel(true, d_bits, n, p_e, seed, stream_id, d_out_i8, d_out_f64, stream);
}

int cpx_count_errors_dev(const uint8_t *d_msg, int64_t msg_stride, const uint8_t *d_dec, int64_t dec_stride, int64_t B,
                         int64_t nchunks, int64_t chunk, int32_t *d_errs, void *stream) {
    CPX_REQUIRE(B >= 0 && nchunks >= 0 && chunk >= 0, CPX_EINVAL, "count_errors: negative size");
    if (B * nchunks == 0) return CPX_OK;
    hipLaunchKernelGGL(count_errors_kernel, dim3(ls_grid(B * nchunks * 64)), dim3(LS_BLOCK), 0, pick_stream(stream), d_msg,
                       msg_stride, d_dec, dec_stride, B, nchunks, chunk, d_errs);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

}  // extern "C"
