// Device-side channel encoders for the turbo and LDPC paths: the transmit side of configs 3 and 4,
// batched, so that non-trivial codewords are produced in HBM next to the decoders.
//   turbo_encode                    /root/reference/commpy/channelcoding/turbo.py:14-59
//   triang_ldpc_systematic_encode   /root/reference/commpy/channelcoding/ldpc.py:302-354 (parity = G.msg mod 2, :353)
// Both are bit-exact integer work, tested against the host mirrors (which the reference goldens pin).
//
// Turbo: a recursive encoder is a serial chain over the N message bits; a wave breaks it with a scan over
// *state maps*.  Lane l owns 16 consecutive trellis steps and folds them into the function "state before my
// segment -> state after it" (S <= 16 states x 4 bits packed in 64 bits); a 6-step inclusive scan composes the 64
// maps; every lane then knows its entry state and walks its 16 steps.  I/O is one 16-byte load and store per lane.
// Trellises with more than 16 states use the plain walk, one codeword per lane.
//
// LDPC: parity = G2 . msg over GF(2) with G2 (m x k) dense and bit-packed.  Lane = parity row (64 rows per wave, their
// packed words staged once in LDS, lane-major -> conflict free), codewords stream through as wave-uniform packed words
// (scalar loads), 8 codewords per LDS read: 1 VALU (v_bitop3_b32: acc ^ (msg & g)) per 32 message bits x 64 rows.
#include "cpx_internal.h"

using namespace cpx;

struct cpx_ldpc_encoder {
    int64_t m, k;
    int kw32, row_blocks;
    int device;
    uint32_t *d_gen = nullptr;   // [row_blocks][kw32][64]: word w of parity row rb*64+lane
};

namespace {

constexpr int EB = 256;          // block size (4 waves)
constexpr int SEG = 16;          // trellis steps per lane
constexpr int SUPER = 64 * SEG;  // steps per wave pass

// ---- packed state maps (entry s = bits 4s..4s+3) ----------------------------------------------------------------
// Map word: 32 bits hold the maps of up to 8 states (half the shift/mask work of the 64-bit form).
template <int S> struct MapWord { using type = uint64_t; };
template <> struct MapWord<4> { using type = uint32_t; };
template <> struct MapWord<8> { using type = uint32_t; };

template <int S, class M>
__device__ __forceinline__ M map_after(M first, M then) {         // s -> then[first[s]]
    M r = 0;
#pragma unroll
    for (int s = 0; s < S; s++) {
        const unsigned mid = (unsigned)(first >> (4 * s)) & 15u;
        r |= ((then >> (4 * mid)) & (M)15) << (4 * s);
    }
    return r;
}

template <class M>
__device__ __forceinline__ unsigned map_at(M map, unsigned s) { return (unsigned)(map >> (4 * s)) & 15u; }

__device__ __forceinline__ uint64_t shfl_up_map(uint64_t v, int d) {
    const unsigned lo = __shfl_up((unsigned)v, d, 64), hi = __shfl_up((unsigned)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shfl_up_map(uint32_t v, int d) { return __shfl_up(v, d, 64); }

__device__ __forceinline__ uint64_t lane63(uint64_t v) {
    return ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 63) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

struct TrellisBits {            // k = 1, n = 2, S <= 16
    uint64_t nx[2];             // packed next-state maps for input 0 / 1
    uint32_t hi[2], lo[2];      // bit s = MSB / LSB of output_table[s][input]
};

// One recursive-encoder pass of a wave over `nsteps` input bits.  in(pos) yields the input bit; emit(pos, hi, lo)
// receives the two output bits of the step (dec2bitarray(output, 2), convcode.py:535).
template <int S, class In, class Emit>
__device__ __forceinline__ void wave_encode(const TrellisBits &tb, int64_t nsteps, In in, Emit emit) {
    using M = typename MapWord<S>::type;
    const int lane = threadIdx.x & 63;
    const M nx[2] = {(M)tb.nx[0], (M)tb.nx[1]};
    unsigned state0 = 0;                                           // wave-uniform state at the start of the pass
    for (int64_t base = 0; base < nsteps; base += SUPER) {
        const int64_t p0 = base + (int64_t)lane * SEG;
        unsigned bits = in(p0);                                    // 16 input bits, bit j = step p0 + j
        const int nvalid = (p0 >= nsteps) ? 0 : (nsteps - p0 < SEG ? (int)(nsteps - p0) : SEG);
        M seg = (M)0xFEDCBA9876543210ull;                          // identity
#pragma unroll
        for (int j = 0; j < SEG; j++)
            if (j < nvalid) seg = map_after<S, M>(seg, nx[(bits >> j) & 1u]);
        M incl = seg;                                              // incl = seg_lane o ... o seg_0
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const M prev = shfl_up_map(incl, d);
            if (lane >= d) incl = map_after<S, M>(prev, incl);
        }
        const M excl = shfl_up_map(incl, 1);
        unsigned st = (lane == 0) ? state0 : map_at(excl, state0);
        unsigned ohi = 0, olo = 0;
#pragma unroll
        for (int j = 0; j < SEG; j++) {
            const unsigned b = (bits >> j) & 1u;
            ohi |= ((tb.hi[b] >> st) & 1u) << j;
            olo |= ((tb.lo[b] >> st) & 1u) << j;
            st = map_at(nx[b], st);
        }
        emit(p0, nvalid, ohi, olo);
        state0 = map_at(lane63(incl), state0);
    }
}

// 16 bytes (one bit each) <-> a 16-bit mask
__device__ __forceinline__ unsigned pack16(const uint8_t *p, int nvalid) {
    unsigned r = 0;
    if (nvalid == SEG && ((uintptr_t)p & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) r |= ((w[q] >> (8 * j)) & 1u) << (4 * q + j);
    } else {
        for (int j = 0; j < nvalid; j++) r |= (unsigned)(p[j] & 1) << j;
    }
    return r;
}

__device__ __forceinline__ void unpack16(uint8_t *p, int nvalid, unsigned bits) {
    if (nvalid == SEG && ((uintptr_t)p & 15) == 0) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[q] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) w[q] |= ((bits >> (4 * q + j)) & 1u) << (8 * j);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int j = 0; j < nvalid; j++) p[j] = (uint8_t)((bits >> j) & 1u);
    }
}

// turbo.py:47-57 for one codeword per wave.  sys/p1 [B][N], p2 [B][np2] (zero beyond N: turbo.py:47 passes 'rsc' as
// the *termination*, so conv_encode runs no tail steps and leaves the tail of its output zero).
template <int S>
__global__ __launch_bounds__(EB) void turbo_encode_wave_kernel(TrellisBits t1, TrellisBits t2,
                                                               const uint8_t *__restrict__ msg, int64_t B, int64_t N,
                                                               const int32_t *__restrict__ perm, uint8_t *__restrict__ sys,
                                                               uint8_t *__restrict__ p1, uint8_t *__restrict__ p2,
                                                               int64_t np2) {
    extern __shared__ uint8_t rows[];                              // [4 waves][N16] systematic row, for the interleaver
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t N16 = (N + 15) & ~(int64_t)15;
    uint8_t *row = rows + (int64_t)wave * N16;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
        const uint8_t *m = msg + b * N;
        uint8_t *s = sys + b * N, *q1 = p1 + b * N, *q2 = p2 + b * np2;
        wave_encode<S>(
            t1, N, [&](int64_t p0) { return pack16(m + p0, p0 >= N ? 0 : (N - p0 < SEG ? (int)(N - p0) : SEG)); },
            [&](int64_t p0, int nvalid, unsigned ohi, unsigned olo) {
                unpack16(s + p0, nvalid, ohi);                     // stream[::2]  (turbo.py:48)
                unpack16(q1 + p0, nvalid, olo);                    // stream[1::2] (turbo.py:49)
                for (int j = 0; j < nvalid; j++) row[p0 + j] = (uint8_t)((ohi >> j) & 1u);
            });
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // LDS row complete (same wave wrote it)
        __builtin_amdgcn_wave_barrier();
        wave_encode<S>(
            t2, N,
            [&](int64_t p0) {                                      // interleaver.interlv(sys_stream) (turbo.py:51)
                unsigned r = 0;
                if (p0 + SEG <= N) {                               // perm + p0 is 64-byte aligned (hipMalloc base)
                    const int4 *pp = reinterpret_cast<const int4 *>(perm + p0);
#pragma unroll
                    for (int q = 0; q < SEG / 4; q++) {
                        const int4 ix = pp[q];
                        r |= ((unsigned)row[ix.x] << (4 * q)) | ((unsigned)row[ix.y] << (4 * q + 1)) |
                             ((unsigned)row[ix.z] << (4 * q + 2)) | ((unsigned)row[ix.w] << (4 * q + 3));
                    }
                } else {
                    for (int j = 0; j < SEG; j++)
                        if (p0 + j < N) r |= (unsigned)row[perm[p0 + j]] << j;
                }
                return r;
            },
            [&](int64_t p0, int nvalid, unsigned, unsigned olo) { unpack16(q2 + p0, nvalid, olo); });  // puncture [[0,1]]
        for (int64_t t = N + lane; t < np2; t += 64) q2[t] = 0;
        __builtin_amdgcn_wave_barrier();
    }
}

// Any table (S <= 128): one codeword per lane, plain walk.
__global__ __launch_bounds__(EB) void turbo_encode_seq_kernel(const int32_t *__restrict__ nx1, const int32_t *__restrict__ ot1,
                                                              int S1, const int32_t *__restrict__ nx2,
                                                              const int32_t *__restrict__ ot2, int S2,
                                                              const uint8_t *__restrict__ msg, int64_t B, int64_t N,
                                                              const int32_t *__restrict__ perm, uint8_t *__restrict__ sys,
                                                              uint8_t *__restrict__ p1, uint8_t *__restrict__ p2,
                                                              int64_t np2) {
    extern __shared__ int32_t tabs[];
    int32_t *a_nx = tabs, *a_ot = a_nx + S1 * 2, *b_nx = a_ot + S1 * 2, *b_ot = b_nx + S2 * 2;
    for (int i = threadIdx.x; i < S1 * 2; i += EB) { a_nx[i] = nx1[i]; a_ot[i] = ot1[i]; }
    for (int i = threadIdx.x; i < S2 * 2; i += EB) { b_nx[i] = nx2[i]; b_ot[i] = ot2[i]; }
    __syncthreads();
    for (int64_t b = (int64_t)blockIdx.x * EB + threadIdx.x; b < B; b += (int64_t)gridDim.x * EB) {
        const uint8_t *m = msg + b * N;
        uint8_t *s = sys + b * N, *q1 = p1 + b * N, *q2 = p2 + b * np2;
        int st = 0;
        for (int64_t t = 0; t < N; t++) {
            const int i = st * 2 + (m[t] & 1);
            const int o = a_ot[i];
            s[t] = (uint8_t)((o >> 1) & 1);
            q1[t] = (uint8_t)(o & 1);
            st = a_nx[i];
        }
        st = 0;
        for (int64_t t = 0; t < N; t++) {
            const int i = st * 2 + (s[perm[t]] & 1);
            q2[t] = (uint8_t)(b_ot[i] & 1);
            st = b_nx[i];
        }
        for (int64_t t = N; t < np2; t++) q2[t] = 0;
    }
}

int trellis_bits(const cpx_trellis *t, TrellisBits *tb) {
    tb->nx[0] = tb->nx[1] = 0;
    tb->hi[0] = tb->hi[1] = tb->lo[0] = tb->lo[1] = 0;
    for (int s = 0; s < t->S; s++)
        for (int b = 0; b < 2; b++) {
            const int nx = t->next_state[s * 2 + b], o = t->output[s * 2 + b];
            if (nx < 0 || nx >= t->S) return 0;
            tb->nx[b] |= (uint64_t)nx << (4 * s);
            tb->hi[b] |= (uint32_t)((o >> 1) & 1) << s;
            tb->lo[b] |= (uint32_t)(o & 1) << s;
        }
    return 1;
}

// ---- LDPC systematic encoder ------------------------------------------------------------------------------------
// pass 1: one wave per codeword: copy the systematic part into the codeword row and bit-pack the message into
// packed[(c >> 3)][w][c & 7] (32 message bits per word; 8 codewords interleaved so that pass 2 fetches one word of 8
// codewords with a single 32-byte scalar load).
__global__ __launch_bounds__(EB) void ldpc_pack_kernel(const uint8_t *__restrict__ msg, int64_t B, int64_t k, int kw32,
                                                       uint32_t *__restrict__ packed, uint8_t *__restrict__ code,
                                                       int64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < B; c += (int64_t)gridDim.x * 4) {
        const uint8_t *m = msg + c * k;
        uint8_t *cw = code + c * n;
        uint32_t *pk = packed + ((c >> 3) * kw32) * 8 + (c & 7);
        for (int64_t j = 0; j * 64 < k; j++) {
            const int64_t pos = j * 64 + lane;
            const unsigned bit = (pos < k) ? (m[pos] & 1u) : 0u;
            if (pos < k) cw[pos] = (uint8_t)bit;
            const uint64_t bal = __ballot(bit != 0);
            if (lane == 0) {
                pk[(2 * j) * 8] = (uint32_t)bal;
                if (2 * j + 1 < kw32) pk[(2 * j + 1) * 8] = (uint32_t)(bal >> 32);
            }
        }
    }
}

// pass 2: block = one block of 64 parity rows (its packed generator rows in LDS), waves stream over groups of 8
// codewords.  parity bit = popcount(G2[row] & msg) & 1  (ldpc.py:353: generator_matrix.dot(message_bits) % 2).
__global__ __launch_bounds__(EB) void ldpc_parity_kernel(const uint32_t *__restrict__ gen, int kw32, int64_t m,
                                                         const uint32_t *__restrict__ packed, int64_t B, int64_t k,
                                                         uint8_t *__restrict__ code, int64_t n) {
    extern __shared__ uint32_t grow[];                             // [kw32][64]
    const int rb = blockIdx.x;
    const uint32_t *g = gen + (int64_t)rb * kw32 * 64;
    for (int i = threadIdx.x; i < kw32 * 64; i += EB) grow[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row = (int64_t)rb * 64 + lane;
    const int64_t groups = (B + 7) >> 3;
    for (int64_t grp = (int64_t)blockIdx.y * 4 + wave; grp < groups; grp += (int64_t)gridDim.y * 4) {
        const uint32_t *pm = packed + grp * kw32 * 8;              // wave-uniform
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < kw32; w++) {
            const uint32_t gw = grow[w * 64 + lane];
#pragma unroll
            for (int c = 0; c < 8; c++)                            // acc ^= msg & g in one v_bitop3_b32 (table 0x6c)
                acc[c] = __builtin_amdgcn_bitop3_b32(pm[w * 8 + c], acc[c], gw, 0x6c);
        }
        if (row < m) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int64_t cw = grp * 8 + c;
                if (cw < B) code[cw * n + k + row] = (uint8_t)(__popc(acc[c]) & 1);
            }
        }
    }
}

}  // namespace

extern "C" {

int cpx_turbo_encode_batch_dev(const cpx_trellis *t1, const cpx_trellis *t2, const uint8_t *d_msg, int64_t B, int64_t N,
                               const int32_t *d_perm, uint8_t *d_sys, uint8_t *d_p1, uint8_t *d_p2, int64_t np2,
                               int mode, void *stream) {
    CPX_TRACE("cpx_turbo_encode_batch_dev");
    CPX_REQUIRE(t1 && t2, CPX_EINVAL, "turbo_encode: null trellis");
    if (int rcd = check_handle_device(t1->device, "turbo_encode")) return rcd;
    if (int rcd = check_handle_device(t2->device, "turbo_encode")) return rcd;
    CPX_REQUIRE(t1->k == 1 && t1->n == 2 && t2->k == 1 && t2->n == 2, CPX_EINVAL,
                "turbo_encode: component codes must be rate 1/2 (k = 1, n = 2)");
    CPX_REQUIRE(B >= 0 && N >= 0 && np2 >= N, CPX_EINVAL, "turbo_encode: bad sizes");
    CPX_REQUIRE(mode >= 0 && mode <= 2, CPX_EINVAL, "turbo_encode: mode must be 0 (auto), 1 (walk) or 2 (scan)");
    CPX_REQUIRE(((uintptr_t)d_perm & 15) == 0, CPX_EINVAL, "turbo_encode: perm must be 16-byte aligned");
    if (B == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const int Smax = t1->S > t2->S ? t1->S : t2->S;
    const int64_t N16 = (N + 15) & ~(int64_t)15;
    const bool scan_ok = Smax <= 16 && 4 * N16 <= 64 * 1024;
    CPX_REQUIRE(mode != 2 || scan_ok, CPX_EINVAL, "turbo_encode: scan kernel needs <= 16 states and N <= 16384");
    if (scan_ok && mode != 1) {
        TrellisBits b1, b2;
        CPX_REQUIRE(trellis_bits(t1, &b1) && trellis_bits(t2, &b2), CPX_EINVAL, "turbo_encode: bad next-state table");
        int64_t blocks = (B + 3) / 4;
        if (blocks > 256 * 16) blocks = 256 * 16;
        const size_t lds = (size_t)(4 * N16);
#define CPX_TE_LAUNCH(SS)                                                                                              \
    hipLaunchKernelGGL(turbo_encode_wave_kernel<SS>, dim3((unsigned)blocks), dim3(EB), lds, st, b1, b2, d_msg, B, N,   \
                       d_perm, d_sys, d_p1, d_p2, np2)
        if (Smax <= 4) CPX_TE_LAUNCH(4);
        else if (Smax <= 8) CPX_TE_LAUNCH(8);
        else CPX_TE_LAUNCH(16);
#undef CPX_TE_LAUNCH
    } else {
        int64_t blocks = (B + EB - 1) / EB;
        if (blocks > 256 * 16) blocks = 256 * 16;
        const size_t lds = sizeof(int32_t) * 4 * (size_t)(t1->S + t2->S);
        hipLaunchKernelGGL(turbo_encode_seq_kernel, dim3((unsigned)blocks), dim3(EB), lds, st, t1->d_next, t1->d_out, t1->S,
                           t2->d_next, t2->d_out, t2->S, d_msg, B, N, d_perm, d_sys, d_p1, d_p2, np2);
    }
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_ldpc_encoder_create(const uint8_t *gen_bits, int64_t m, int64_t k, cpx_ldpc_encoder **out) {
    CPX_REQUIRE(gen_bits && out, CPX_EINVAL, "ldpc_encoder_create: null pointer");
    CPX_REQUIRE(m >= 1 && k >= 1, CPX_EINVAL, "ldpc_encoder_create: empty generator");
    CPX_REQUIRE(k <= 8192, CPX_EINVAL, "ldpc_encoder_create: k = %lld exceeds the 8192-bit limit of the LDS-staged rows",
                (long long)k);
    int rc = ensure_device();
    if (rc) return rc;
    cpx_ldpc_encoder *e = new cpx_ldpc_encoder;
    e->m = m; e->k = k;
    e->kw32 = (int)((k + 31) / 32);
    if (e->kw32 & 1) e->kw32++;                                     // the packer writes word pairs
    e->row_blocks = (int)((m + 63) / 64);
    CPX_HIP(hipGetDevice(&e->device));
    std::vector<uint32_t> h((size_t)e->row_blocks * e->kw32 * 64, 0u);
    for (int64_t r = 0; r < m; r++) {
        uint32_t *dst = h.data() + (size_t)(r / 64) * e->kw32 * 64 + (r % 64);
        const uint8_t *src = gen_bits + r * k;
        for (int64_t j = 0; j < k; j++)
            if (src[j] & 1) dst[(j >> 5) * 64] |= 1u << (j & 31);
    }
    hipError_t er = hipMalloc((void **)&e->d_gen, h.size() * sizeof(uint32_t));
    if (er != hipSuccess) { delete e; set_error("ldpc_encoder_create: hipMalloc failed: %s", hipGetErrorString(er)); return CPX_ENOMEM; }
    er = hipMemcpy(e->d_gen, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (er != hipSuccess) { (void)hipFree(e->d_gen); delete e; set_error("ldpc_encoder_create: upload failed: %s", hipGetErrorString(er)); return CPX_EHIP; }
    *out = e;
    return CPX_OK;
}

int cpx_ldpc_encoder_destroy(cpx_ldpc_encoder *e) {
    if (!e) return CPX_OK;
    if (e->d_gen) (void)hipFree(e->d_gen);
    delete e;
    return CPX_OK;
}

int cpx_ldpc_encode_batch_dev(const cpx_ldpc_encoder *e, const uint8_t *d_msg, int64_t B, uint8_t *d_code, void *stream) {
    CPX_TRACE("cpx_ldpc_encode_batch_dev");
    cpx::IssueGuard issue_guard;
    CPX_REQUIRE(e, CPX_EINVAL, "ldpc_encode: null encoder");
    if (int rcd = check_handle_device(e->device, "ldpc_encode")) return rcd;
    CPX_REQUIRE(B >= 0, CPX_EINVAL, "ldpc_encode: negative batch");
    if (B == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const int64_t n = e->k + e->m, groups = (B + 7) / 8;
    void *ws = nullptr;
    int rc = workspace(st, 2, (size_t)groups * e->kw32 * 8 * sizeof(uint32_t), &ws);
    if (rc) return rc;
    uint32_t *packed = static_cast<uint32_t *>(ws);
    int64_t pblocks = (B + 3) / 4;
    if (pblocks > 256 * 32) pblocks = 256 * 32;
    hipLaunchKernelGGL(ldpc_pack_kernel, dim3((unsigned)pblocks), dim3(EB), 0, st, d_msg, B, e->k, e->kw32, packed, d_code, n);
    CPX_HIP(hipGetLastError());
    int64_t gy = (groups + 3) / 4;
    const int64_t cap = (256 * 8 + e->row_blocks - 1) / e->row_blocks;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    const size_t lds = (size_t)e->kw32 * 64 * sizeof(uint32_t);
    hipLaunchKernelGGL(ldpc_parity_kernel, dim3((unsigned)e->row_blocks, (unsigned)gy), dim3(EB), lds, st, e->d_gen, e->kw32,
                       e->m, packed, B, e->k, d_code, n);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

}  // extern "C"
