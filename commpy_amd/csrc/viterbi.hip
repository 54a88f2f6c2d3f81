// Viterbi decoder for gfx950 -- one wavefront per codeword (S = 64), or 64/S codewords per
// wavefront for smaller trellises.  Replaces the body of the reference's
//   viterbi_decode / _acs_traceback / _compute_branch_metrics
//   (/root/reference/commpy/channelcoding/convcode.py:661-749, :590-657, :575-587)
// with the decision rule of SURVEY Appendix A.1 (verified against the live reference):
//   forward ACS over t = 1..T with float64 path metrics, first-minimum tie rule in np.where order;
//   best[t] = first-argmin state;  bit(s) of step s = survivor symbol at step s of the path traced
//   back from best[min(s + tb - 2, T)].
//
// Mapping (wave64):
//   lane = g*S + s : codeword slot g (G = 64/S slots per wave), trellis state s.
//   * path metric of state s lives in one VGPR pair of lane s; predecessors are fetched with
//     wavefront shuffles (ds_bpermute), no LDS round trip for the ACS recursion;
//   * branch metrics: per chunk of CH = S steps, lane (g,i) loads the n received values of step
//     t_base+i (coalesced), evaluates the reference's per-bit metrics once and writes the 2^n
//     codeword metrics of that step to LDS; the ACS lanes read them back by codeword index;
//   * survivor decisions: one 64-bit ballot word per step (two for I = 4) in an LDS ring of
//     RS >= CH + tb - 2 steps, plus the per-step first-argmin state (1 byte per codeword slot);
//   * sliding traceback after every chunk, lane-parallel over output steps (lane (g,i) owns
//     output step next_out + i of codeword g), walking tb-2 decision words of the ring.
// HBM traffic is exactly the algorithmic one: each received value is read once, each decoded
// bit written once.  float64 throughout (the reference's arithmetic); compiled with
// -ffp-contract=off so sums round like NumPy's.
#include "cpx_internal.h"

using namespace cpx;

namespace {

struct VitParams {
    const double *coded;   // [B][len]
    uint8_t *bits;         // [B][L]
    const int32_t *pred_state, *pred_input, *pred_code;  // [S][I]
    int64_t B, len, L, T, Lk;
    int k, n, lgS, I, NC, type, tb, RS;
};

__device__ __forceinline__ double shfl_f64(double v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

// Per-bit metrics of one received value (convcode.py:575-587): m0 = cost of code bit 0, m1 = of bit 1.
__device__ __forceinline__ void bit_metrics(int type, double r, double &m0, double &m1) {
    if (type == CPX_VIT_HARD) {
        long long ri = (long long)r;            // r_codeword.astype(int) (:580)
        m0 = (double)(ri ^ 0ll);                // hamming_dist = sum of xor (utilities.py:130)
        m1 = (double)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        double nll0 = log(exp(r) + 1.0);        // :582
        m0 = nll0;
        m1 = nll0 - r;                          // :583
    } else {
        double d0 = r - (-1.0), d1 = r - 1.0;   // i_codeword_array = 2*c - 1 (:586), euclid_dist utilities.py:152
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

template <int I_T>
__global__ __launch_bounds__(64) void viterbi_wave_kernel(VitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PL = (I_T == 2) ? 1 : 2;          // decision bit planes
    const int lane = threadIdx.x;
    const int lgS = p.lgS, S = 1 << lgS, G = 64 >> lgS, CH = S;
    const int NC = p.NC, n = p.n, k = p.k, RM = p.RS - 1;
    const int g = lane >> lgS, s = lane & (S - 1);

    double *bm = reinterpret_cast<double *>(smem);                              // [64][NC]
    unsigned long long *dring = reinterpret_cast<unsigned long long *>(bm + 64 * NC);  // [RS][PL]
    unsigned short *ptab = reinterpret_cast<unsigned short *>(dring + (size_t)p.RS * PL);  // [S*I]
    unsigned char *bring = reinterpret_cast<unsigned char *>(ptab + S * I_T);   // [RS][G]

    const int64_t cw = (int64_t)blockIdx.x * G + g;
    const bool valid_cw = cw < p.B;
    const double *x = p.coded + (valid_cw ? cw : 0) * p.len;

    int predlane[I_T], pcode[I_T];
#pragma unroll
    for (int j = 0; j < I_T; j++) {
        predlane[j] = (g << lgS) + p.pred_state[s * I_T + j];
        pcode[j] = p.pred_code[s * I_T + j];
    }
    for (int idx = lane; idx < S * I_T; idx += 64)
        ptab[idx] = (unsigned short)(p.pred_state[idx] | (p.pred_input[idx] << 8));

    double pm = (s == 0) ? 0.0 : __builtin_huge_val();      // path_metrics[:,0] = inf, [0][0] = 0 (:705-706)
    const int gshift = g << lgS;
    const unsigned long long gmask = (S == 64) ? ~0ull : ((1ull << S) - 1ull);
    int64_t next_out = 1;                                    // first output step not yet finalised

    for (int64_t t_base = 1; t_base <= p.T; t_base += CH) {
        // ---------------- branch-metric table of this chunk ----------------
        {
            const int64_t t = t_base + s;                    // lane (g, s) prepares step t of codeword g
            double m0[CPX_MAX_N], m1[CPX_MAX_N];
            const bool have = valid_cw && (t <= p.Lk) && (t <= p.T);   // t > L//k -> padding (:722-734)
            for (int j = 0; j < n; j++) {
                double r = (p.type == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;
                if (have) {
                    r = x[(t - 1) * n + j];
                    if (p.type == CPX_VIT_SOFT) r = fmin(fmax(r, -500.0), 500.0);   // coded_bits.clip(-500, 500) (:719)
                }
                bit_metrics(p.type, r, m0[j], m1[j]);
            }
            double *row = bm + lane * NC;
            for (int c = 0; c < NC; c++) {
                double acc = 0.0;                            // NumPy add.reduce, n < 8: sequential from 0
                for (int j = 0; j < n; j++) acc += ((c >> (n - 1 - j)) & 1) ? m1[j] : m0[j];   // MSB-first bits (:622)
                row[c] = acc;
            }
        }
        __syncthreads();

        // ---------------- forward add-compare-select ----------------
        const int nsteps = (int)((p.T - t_base + 1 < CH) ? (p.T - t_base + 1) : CH);
        for (int i = 0; i < nsteps; i++) {
            const int64_t t = t_base + i;
            const double *row = bm + ((g << lgS) + i) * NC;
            double best = shfl_f64(pm, predlane[0]) + row[pcode[0]];      // pmetrics[0] (:629)
            int jb = 0;
#pragma unroll
            for (int j = 1; j < I_T; j++) {
                double c = shfl_f64(pm, predlane[j]) + row[pcode[j]];
                if (c < best) { best = c; jb = j; }                        // first minimum wins (:633-642)
            }
            pm = best;
            const unsigned long long w0 = __ballot(jb & 1);
            unsigned long long w1 = 0;
            if (PL == 2) w1 = __ballot(jb & 2);
            // first-argmin state of this step (:645): wave min-reduce inside the S-lane group
            double mn = pm;
            for (int off = 1; off < S; off <<= 1) {
                double o = shfl_xor_f64(mn, off);
                mn = (o < mn) ? o : mn;
            }
            const unsigned long long eq = __ballot(pm == mn);
            const unsigned long long grp = (eq >> gshift) & gmask;
            const int bst = grp ? (__ffsll((long long)grp) - 1) : 0;
            const int slot = (int)(t & RM);
            if (lane == 0) {
                dring[slot * PL] = w0;
                if (PL == 2) dring[slot * PL + 1] = w1;
            }
            if (s == 0) bring[slot * G + g] = (unsigned char)bst;
        }
        __syncthreads();

        // ---------------- sliding traceback ----------------
        const int64_t t_done = t_base + nsteps - 1;
        const int64_t s_hi = (t_done >= p.T) ? p.T : (t_done - p.tb + 2);
        while (next_out <= s_hi) {
            const int64_t so = next_out + s;                 // output step owned by this lane
            if (so <= s_hi && valid_cw) {
                int64_t t0 = so + p.tb - 2;
                if (t0 > p.T) t0 = p.T;
                int st = bring[(int)(t0 & RM) * G + g];
                for (int64_t tt = t0; tt > so; --tt) {
                    const int slot = (int)(tt & RM);
                    int j = (int)((dring[slot * PL] >> (gshift + st)) & 1ull);
                    if (PL == 2) j |= (int)((dring[slot * PL + 1] >> (gshift + st)) & 1ull) << 1;
                    st = ptab[st * I_T + j] & 0xff;          // paths[current_state, j] (:651)
                }
                const int slot = (int)(so & RM);
                int j = (int)((dring[slot * PL] >> (gshift + st)) & 1ull);
                if (PL == 2) j |= (int)((dring[slot * PL + 1] >> (gshift + st)) & 1ull) << 1;
                const int sym = ptab[st * I_T + j] >> 8;     // decoded_symbols[current_state, j] (:650)
                for (int b = 0; b < k; b++) {
                    const int64_t pos = (so - 1) * k + b;
                    if (pos < p.L) p.bits[cw * p.L + pos] = (uint8_t)((sym >> (k - 1 - b)) & 1);   // dec2bitarray(sym, k) (:652)
                }
            }
            next_out = (next_out + CH <= s_hi + 1) ? next_out + CH : s_hi + 1;
        }
        __syncthreads();
    }
}

int next_pow2(int v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}

}  // namespace

extern "C" {

int cpx_trellis_create(int k, int n, int n_states, int n_inputs, const int32_t *next_state_table,
                       const int32_t *output_table, cpx_trellis **out) {
    CPX_REQUIRE(out && next_state_table && output_table, CPX_EINVAL, "cpx_trellis_create: null pointer");
    CPX_REQUIRE(k >= 1 && n >= 1 && n_inputs == (1 << k), CPX_EINVAL, "cpx_trellis_create: n_inputs must be 2^k");
    CPX_REQUIRE(n_states >= 1 && (n_states & (n_states - 1)) == 0, CPX_EINVAL,
                "cpx_trellis_create: n_states must be a power of two");
    CPX_REQUIRE(n_states <= CPX_MAX_STATES, CPX_ELIMIT, "cpx_trellis_create: at most %d states supported", CPX_MAX_STATES);
    CPX_REQUIRE(n <= 16, CPX_ELIMIT, "cpx_trellis_create: n <= 16");
    int rc = ensure_device();
    if (rc) return rc;
    const int S = n_states, I = n_inputs;
    for (int i = 0; i < S * I; i++) {
        CPX_REQUIRE(next_state_table[i] >= 0 && next_state_table[i] < S, CPX_EINVAL, "next_state_table entry out of range");
        CPX_REQUIRE(output_table[i] >= 0 && output_table[i] < (1 << n), CPX_EINVAL, "output_table entry out of range");
    }
    cpx_trellis *t = new cpx_trellis;
    t->k = k; t->n = n; t->S = S; t->I = I;
    t->next_state.assign(next_state_table, next_state_table + S * I);
    t->output.assign(output_table, output_table + S * I);
    // predecessor lists in np.where (row-major) order: convcode.py:561-572
    t->pred_state.assign(S * I, -1); t->pred_input.assign(S * I, -1); t->pred_code.assign(S * I, 0);
    bool regular = true;
    std::vector<int> cnt(S, 0);
    for (int ps = 0; ps < S && regular; ps++)
        for (int i = 0; i < I; i++) {
            int ns = next_state_table[ps * I + i];
            if (cnt[ns] >= I) { regular = false; break; }
            t->pred_state[ns * I + cnt[ns]] = ps;
            t->pred_input[ns * I + cnt[ns]] = i;
            t->pred_code[ns * I + cnt[ns]] = output_table[ps * I + i];
            cnt[ns]++;
        }
    for (int s2 = 0; s2 < S; s2++) if (cnt[s2] != I) regular = false;
    if (!regular) {
        delete t;
        set_error("cpx_trellis_create: every state needs exactly %d incoming branches (the reference indexes "
                  "pmetrics[number_inputs], convcode.py:604-629)", I);
        return CPX_EINVAL;
    }
    hipGetDevice(&t->device);
    size_t bytes = sizeof(int32_t) * S * I;
    int32_t **dst[5] = {&t->d_next, &t->d_out, &t->d_pred_state, &t->d_pred_input, &t->d_pred_code};
    const int32_t *src[5] = {t->next_state.data(), t->output.data(), t->pred_state.data(), t->pred_input.data(),
                             t->pred_code.data()};
    for (int i = 0; i < 5; i++) {
        CPX_HIP(hipMalloc((void **)dst[i], bytes));
        CPX_HIP(hipMemcpy(*dst[i], src[i], bytes, hipMemcpyHostToDevice));
    }
    *out = t;
    return CPX_OK;
}

int cpx_trellis_destroy(cpx_trellis *t) {
    if (!t) return CPX_OK;
    (void)hipFree(t->d_next); (void)hipFree(t->d_out);
    (void)hipFree(t->d_pred_state); (void)hipFree(t->d_pred_input); (void)hipFree(t->d_pred_code);
    delete t;
    return CPX_OK;
}

int cpx_viterbi_decode_batch_dev(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L,
                                 int64_t n_steps, int tb_depth, int decoding_type, uint8_t *d_bits, void *stream) {
    CPX_REQUIRE(t, CPX_EINVAL, "viterbi: null trellis");
    CPX_REQUIRE(decoding_type >= 0 && decoding_type <= 2, CPX_EINVAL,
                "The available decoding types are \"hard\", \"soft\" and \"unquantized");
    CPX_REQUIRE(B >= 0 && len >= 0 && L >= 0, CPX_EINVAL, "viterbi: negative size");
    CPX_REQUIRE(tb_depth >= 2, CPX_EINVAL, "viterbi: tb_depth must be >= 2");
    CPX_REQUIRE((L / t->k) * (int64_t)t->n <= len, CPX_EINVAL, "viterbi: L inconsistent with len");
    CPX_REQUIRE(t->I == 2 || t->I == 4, CPX_ELIMIT, "viterbi: trellis with %d inputs per step not supported (k <= 2)", t->I);
    CPX_REQUIRE(t->n <= CPX_MAX_N, CPX_ELIMIT, "viterbi: n = %d > %d not supported", t->n, CPX_MAX_N);
    CPX_REQUIRE(t->S <= 64, CPX_ELIMIT, "viterbi: %d states not supported yet (<= 64)", t->S);
    if (B == 0 || L == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    if (n_steps <= 0 || n_steps * t->k < L) CPX_HIP(hipMemsetAsync(d_bits, 0, (size_t)(B * L), st));
    if (n_steps <= 0) return CPX_OK;

    VitParams p;
    p.coded = d_coded; p.bits = d_bits;
    p.pred_state = t->d_pred_state; p.pred_input = t->d_pred_input; p.pred_code = t->d_pred_code;
    p.B = B; p.len = len; p.L = L; p.T = n_steps; p.Lk = L / t->k;
    p.k = t->k; p.n = t->n; p.I = t->I; p.NC = 1 << t->n; p.type = decoding_type; p.tb = tb_depth;
    int lgS = 0;
    while ((1 << lgS) < t->S) lgS++;
    p.lgS = lgS;
    const int S = t->S, G = 64 / S, CH = S, PL = (t->I == 2) ? 1 : 2;
    p.RS = next_pow2(CH + tb_depth);
    size_t lds = sizeof(double) * 64 * p.NC + sizeof(unsigned long long) * p.RS * PL + sizeof(unsigned short) * S * t->I +
                 (size_t)p.RS * G;
    CPX_REQUIRE(lds <= 64 * 1024, CPX_ELIMIT, "viterbi: tb_depth %d needs %zu B of LDS (> 64 KiB)", tb_depth, lds);
    const int64_t nblocks = (B + G - 1) / G;
    CPX_REQUIRE(nblocks < (1ll << 31), CPX_ELIMIT, "viterbi: batch too large");
    dim3 grid((unsigned)nblocks), block(64);
    if (t->I == 2) hipLaunchKernelGGL(viterbi_wave_kernel<2>, grid, block, lds, st, p);
    else hipLaunchKernelGGL(viterbi_wave_kernel<4>, grid, block, lds, st, p);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_viterbi_decode_batch(const cpx_trellis *t, const double *coded, int64_t B, int64_t len, int64_t L,
                             int64_t n_steps, int tb_depth, int decoding_type, uint8_t *bits) {
    CPX_REQUIRE(t && (coded || B * len == 0) && (bits || B * L == 0), CPX_EINVAL, "viterbi: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || L == 0) return CPX_OK;
    DevBuf din, dout;
    if ((rc = din.alloc(sizeof(double) * (size_t)(B * len)))) return rc;
    if ((rc = dout.alloc((size_t)(B * L)))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, coded, sizeof(double) * (size_t)(B * len), hipMemcpyHostToDevice, st));
    rc = cpx_viterbi_decode_batch_dev(t, din.as<double>(), B, len, L, n_steps, tb_depth, decoding_type,
                                      dout.as<uint8_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dout.p, (size_t)(B * L), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

}  // extern "C"
