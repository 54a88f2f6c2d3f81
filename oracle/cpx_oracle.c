/*
 * cpx_oracle.c -- CPU restatement ("oracle") of the CommPy 0.8.0 decoding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the CPU baseline, never as the thing that is shipped or measured as the GPU path.
 *
 * Each function restates, line by line and in the reference's own evaluation order, one function
 * of the reference (veeresht/CommPy @ /root/reference, pure Python/NumPy/SciPy).  Citations are
 * file:line into /root/reference/.  Parity status: PINNED -- tests/test_oracle_golden.py checks
 * every function below against fixtures generated from the live reference
 * (tests/golden/make_golden.py) and against the reference's own golden tables
 * (commpy/channelcoding/tests/test_convcode.py:23-111, commpy/tests/test_utilities.py:12-13).
 *
 * Build: see oracle/Makefile (gcc -O2 -fno-fast-math -ffp-contract=off; no FMA contraction so
 * that sums/products round exactly like NumPy's scalar loops).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#undef I /* complex.h macro; `I` is the number of trellis inputs below */
#define CI _Complex_I

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-2)

/* ---------------------------------------------------------------------------------------------
 * NumPy float64 add.reduce order (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum):
 * n < 8 sequential; n <= 128 eight strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7));
 * larger n split recursively.  Probed in this container: [1e16,1,1,1].sum()==1e16 (sequential),
 * [1e16,1,1,1,1,1,1,1].sum()-1e16==6 (8 accumulators).  The reference relies on it at
 * convcode.py:584 (.sum() of n values), turbo.py:110-111,155-156 (normalisation over S states).
 * ------------------------------------------------------------------------------------------- */
static double np_pairwise_sum(const double *a, int64_t n, int64_t stride)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2, stride) + np_pairwise_sum(a + n2 * stride, n - n2, stride);
    }
}

double orc_np_sum(const double *a, int64_t n, int64_t stride) { return np_pairwise_sum(a, n, stride); }

/* utilities.py:59-86 decimal2bitarray: MSB first; writes result[bit_width - pox - 1] with Python
 * negative-index wrap-around when the value needs more than bit_width bits (quirk B1). */
int orc_dec2bitarray(int64_t number, int bit_width, int8_t *result)
{
    memset(result, 0, (size_t)bit_width);
    int64_t i = 1;
    int pox = 0;
    while (i <= number) {
        if (i & number) {
            int idx = bit_width - pox - 1;
            if (idx < 0) idx += bit_width; /* Python negative index */
            if (idx < 0 || idx >= bit_width) return ORC_EINVAL; /* IndexError in the reference */
            result[idx] = 1;
        }
        i <<= 1;
        pox += 1;
    }
    return ORC_OK;
}

/* =============================================================================================
 * Viterbi -- convcode.py:561-749 (_where_c, _compute_branch_metrics, _acs_traceback, viterbi_decode)
 * Literal sliding-window restatement: survivors `paths`/`decoded_symbols` of width tb_depth, a
 * full traceback at every step t >= tb_depth-1, later tracebacks overwriting earlier ones.
 * type: 0 hard, 1 soft, 2 unquantized.
 * ===========================================================================================*/
static double branch_metric(int type, const double *r, int64_t codeword, int n)
{
    /* convcode.py:575-587 */
    double acc = 0.;
    double tmp[64];
    for (int j = 0; j < n; j++) {
        int c = (int)((codeword >> (n - 1 - j)) & 1); /* dec2bitarray(i_codeword, n): MSB first (:622) */
        if (type == 0) {
            int64_t ri = (int64_t)r[j];                 /* r_codeword.astype(int) (:580) */
            tmp[j] = (double)(ri ^ c);                   /* hamming_dist: xor, sum (utilities.py:130) */
        } else if (type == 1) {
            double nll0 = log(exp(r[j]) + 1);            /* :582 */
            double nll1 = nll0 - r[j];                   /* :583 */
            tmp[j] = c ? nll1 : nll0;                    /* :584 */
        } else {
            double d = r[j] - (double)(2 * c - 1);       /* :586-587, utilities.py:152 */
            tmp[j] = d * d;
        }
    }
    if (type == 0) { /* integer sum */
        for (int j = 0; j < n; j++) acc += tmp[j];
        return acc;
    }
    return np_pairwise_sum(tmp, n, 1);
}

int orc_viterbi_decode(const double *coded_in, int64_t len, int k, int n, int total_memory, int S, int I,
                       const int32_t *next_state, const int32_t *output, int tb_depth, int type,
                       int64_t *decoded_out /* [L] */, int64_t *L_out)
{
    if (n > 64 || k < 1 || S < 1 || type < 0 || type > 2) return ORC_EINVAL;
    double rate = (double)k / (double)n;                                  /* :694 */
    int64_t L = (int64_t)((double)len * rate);                            /* :698 */
    if (L_out) *L_out = L;
    if (tb_depth <= 0) tb_depth = (int)((5 * total_memory < L) ? 5 * total_memory : L); /* :701-702 */
    if (tb_depth < 2) return ORC_EINVAL;
    int64_t nsteps_end = (int64_t)((double)(L + total_memory) / (double)k); /* range(1, int((L+m)/k)) :721 */

    double *pm = malloc(sizeof(double) * 2 * S);
    int64_t *paths = calloc((size_t)S * tb_depth, sizeof(int64_t));
    int64_t *dsym = calloc((size_t)S * tb_depth, sizeof(int64_t));
    int64_t nbits = (int64_t)ceil((double)(L + tb_depth) / (double)k) * k;  /* :711 */
    int64_t *dbits = calloc((size_t)nbits + 1, sizeof(int64_t));
    double *coded = malloc(sizeof(double) * (size_t)(len > 0 ? len : 1));
    double *pmetrics = malloc(sizeof(double) * I);
    int32_t *idx_state = malloc(sizeof(int32_t) * S * I), *idx_input = malloc(sizeof(int32_t) * S * I);
    double rpad[64];
    if (!pm || !paths || !dsym || !dbits || !coded || !pmetrics || !idx_state || !idx_input) return ORC_ENOMEM;
    for (int s = 0; s < S; s++) { pm[2 * s] = INFINITY; pm[2 * s + 1] = INFINITY; }  /* :705 */
    pm[0] = 0;                                                               /* :706 */
    for (int64_t i = 0; i < len; i++) {
        double v = coded_in[i];
        if (type == 1) v = v < -500 ? -500 : (v > 500 ? 500 : v);            /* :718-719 */
        coded[i] = v;
    }
    int tb_count = 1;
    int64_t count = 0;
    int rc = ORC_OK;
    int ncode = 1 << n;
    if (n > 16) return ORC_EINVAL;
    double *bmtab = malloc(sizeof(double) * ncode);
    int32_t *pred_state = idx_state, *pred_input = idx_input;
    for (int state = 0; state < S; state++) {
        int found = 0;
        for (int p = 0; p < S; p++)                          /* np.where order: row-major (:563-565) */
            for (int i = 0; i < I; i++)
                if (next_state[p * I + i] == state) {
                    if (found >= I) { rc = ORC_EINVAL; goto done; }   /* pmetrics[i] IndexError in the reference */
                    pred_state[(int64_t)state * I + found] = p;
                    pred_input[(int64_t)state * I + found] = i;
                    found++;
                }
        if (found != I) { rc = ORC_EINVAL; goto done; }       /* reference would read np.empty garbage */
    }
    for (int i = 0; i < S * I; i++) if (output[i] < 0 || output[i] >= ncode) { rc = ORC_EINVAL; goto done; }

    for (int64_t t = 1; t < nsteps_end; t++) {
        const double *r;
        if (t <= L / k) {
            r = coded + (t - 1) * n;                                          /* :723-724 */
        } else {
            for (int j = 0; j < n; j++) rpad[j] = (type == 2) ? -1. : 0.;     /* :726-734 */
            r = rpad;
        }
        /* ---- _acs_traceback (:590-657) ---- */
        /* branch metrics are a pure function of (r, codeword): evaluate each of the 2^n codewords once
         * per step (the reference recomputes / lru-caches the same values per branch, :625). */
        for (int c = 0; c < ncode; c++) bmtab[c] = branch_metric(type, r, c, n);
        for (int state = 0; state < S; state++) {
            /* _where_c result (row-major scan of next_state_table == state, :561-572) precomputed below */
            const int32_t *ps = pred_state + (int64_t)state * I, *pi = pred_input + (int64_t)state * I;
            for (int i = 0; i < I; i++)
                pmetrics[i] = pm[2 * ps[i]] + bmtab[output[ps[i] * I + pi[i]]];  /* :629 */
            int min_idx = 0;                                                   /* argmin: first minimum (:637) */
            for (int i = 1; i < I; i++) if (pmetrics[i] < pmetrics[min_idx]) min_idx = i;
            pm[2 * state + 1] = pmetrics[min_idx];                            /* :633 */
            paths[(int64_t)state * tb_depth + tb_count] = ps[min_idx];        /* :638 */
            dsym[(int64_t)state * tb_depth + tb_count] = pi[min_idx];         /* :642 */
        }
        if (t >= tb_depth - 1) {                                               /* :644 */
            int cur = 0;
            for (int s = 1; s < S; s++) if (pm[2 * s + 1] < pm[2 * cur + 1]) cur = s;   /* :645 argmin */
            for (int j = tb_depth - 1; j >= 1; j--) {                         /* :648 */
                int64_t sym = dsym[(int64_t)cur * tb_depth + j];
                int64_t prev = paths[(int64_t)cur * tb_depth + j];
                int64_t base = t - tb_depth + 1 + (int64_t)(j - 1) * k + count;      /* :653 */
                for (int b = 0; b < k; b++)
                    if (base + b >= 0 && base + b < nbits) dbits[base + b] = (sym >> (k - 1 - b)) & 1;
                cur = (int)prev;
            }
            for (int s = 0; s < S; s++) {                                      /* :656-657 */
                memmove(paths + (int64_t)s * tb_depth, paths + (int64_t)s * tb_depth + 1, sizeof(int64_t) * (tb_depth - 1));
                memmove(dsym + (int64_t)s * tb_depth, dsym + (int64_t)s * tb_depth + 1, sizeof(int64_t) * (tb_depth - 1));
            }
        }
        /* ---- back in viterbi_decode (:740-747) ---- */
        if (t >= tb_depth - 1) { tb_count = tb_depth - 1; count = count + k - 1; }
        else tb_count = tb_count + 1;
        for (int s = 0; s < S; s++) pm[2 * s] = pm[2 * s + 1];
    }
    for (int64_t i = 0; i < L; i++) decoded_out[i] = dbits[i];                /* :749 */
done:
    free(bmtab); free(pm); free(paths); free(dsym); free(dbits); free(coded); free(pmetrics); free(idx_state); free(idx_input);
    return rc;
}

/* =============================================================================================
 * BCJR / MAP -- turbo.py:62-251  (rate-1/2 trellis, probability domain, per-step normalisation)
 * mode: 1 'decode' (hard bits written), 0 'compute' (bits stay 0)
 * ===========================================================================================*/
int orc_map_decode(const double *sys, const double *par, int64_t N, int n, int S, int I,
                   const int32_t *next_state, const int32_t *output, double noise_variance,
                   const double *L_int, int mode, double *L_ext, int64_t *bits)
{
    if (I != 2 || n < 2) return ORC_EINVAL;
    double *f = calloc((size_t)S * 2, sizeof(double));                 /* f_state_metrics [S,2]  :220 */
    double *b = calloc((size_t)S * (N + 1), sizeof(double));           /* b_state_metrics [S,N+1] :224 */
    double *bp = calloc((size_t)I * S * (N + 1), sizeof(double));      /* branch_probs [I,S,N+1]  :228 */
    double *pri = malloc(sizeof(double) * 2 * (size_t)(N > 0 ? N : 1)); /* priors [2,N] :238 */
    if (!f || !b || !bp || !pri) return ORC_ENOMEM;
    f[0] = 1;                                                           /* :221 */
    for (int s = 0; s < S; s++) b[(int64_t)s * (N + 1) + N] = 1;        /* :225 */
    for (int64_t t = 0; t < N; t++) {
        pri[t] = 1 / (1 + exp(L_int[t]));                               /* :239 */
        pri[N + t] = 1 - pri[t];                                        /* :240 */
    }
    /* _backward_recursion :78-111 */
    for (int64_t rt = N; rt >= 1; rt--) {
        for (int cs = 0; cs < S; cs++)
            for (int ci = 0; ci < I; ci++) {
                int ns = next_state[cs * I + ci];
                int code = output[cs * I + ci];
                int msg_bit = (code >> (n - 1)) & 1;                    /* codeword_array[0] :98 */
                int parity_bit = (code >> (n - 2)) & 1;                 /* codeword_array[1] :97 */
                double x = sys[rt - 1] - (double)(2 * msg_bit - 1);     /* :69-72 */
                double y = par[rt - 1] - (double)(2 * parity_bit - 1);
                double p = exp(-(x * x + y * y) / (2 * noise_variance)); /* :74 */
                bp[((int64_t)ci * S + cs) * (N + 1) + rt - 1] = p;      /* :105 */
                b[(int64_t)cs * (N + 1) + rt - 1] +=
                    (b[(int64_t)ns * (N + 1) + rt] * p * pri[(int64_t)ci * N + rt - 1]);   /* :106-108 */
            }
        double sum = np_pairwise_sum(b + rt - 1, S, N + 1);              /* :110-111 (strided column) */
        for (int s = 0; s < S; s++) b[(int64_t)s * (N + 1) + rt - 1] /= sum;
    }
    /* _forward_recursion_decoding :114-158 */
    double app[2];
    for (int64_t t = 1; t <= N; t++) {
        app[0] = app[1] = 0;
        for (int cs = 0; cs < S; cs++)
            for (int ci = 0; ci < I; ci++) {
                int ns = next_state[cs * I + ci];
                double p = bp[((int64_t)ci * S + cs) * (N + 1) + t - 1];
                f[2 * ns + 1] += (f[2 * cs] * p * pri[(int64_t)ci * N + t - 1]);          /* :136-138 */
                app[ci] += (f[2 * cs] * p * b[(int64_t)ns * (N + 1) + t]);                /* :141-143 */
            }
        double lappr = L_int[t - 1] + log(app[1] / app[0]);              /* :145 */
        L_ext[t - 1] = lappr;
        if (mode == 1) bits[t - 1] = lappr > 0 ? 1 : 0; else bits[t - 1] = 0;           /* :148-152 */
        double sum = np_pairwise_sum(f + 1, S, 2);                       /* :155 */
        for (int s = 0; s < S; s++) { f[2 * s] = f[2 * s + 1] / sum; f[2 * s + 1] = 0.0; } /* :155-158 */
    }
    free(f); free(b); free(bp); free(pri);
    return ORC_OK;
}

/* turbo.py:254-333.  perm = interleaver.p_array (interlv: out = in[p]; deinterlv: out[p[i]] = in[i],
 * interleavers.py:13-47). */
int orc_turbo_decode(const double *sys, const double *p1, const double *p2, int64_t N, int n, int S, int I,
                     const int32_t *next_state, const int32_t *output, double noise_variance, int n_iter,
                     const int64_t *perm, const double *L_int_or_null, int64_t *decoded)
{
    double *L1 = malloc(sizeof(double) * N), *Le = malloc(sizeof(double) * N), *L2in = malloc(sizeof(double) * N);
    double *sysi = malloc(sizeof(double) * N), *L2 = malloc(sizeof(double) * N);
    int64_t *bits = calloc(N, sizeof(int64_t));
    if (!L1 || !Le || !L2in || !sysi || !L2 || !bits) return ORC_ENOMEM;
    for (int64_t i = 0; i < N; i++) L1[i] = L_int_or_null ? L_int_or_null[i] : 0.;       /* :305-308 */
    for (int64_t i = 0; i < N; i++) sysi[i] = sys[perm[i]];                               /* :310 */
    int rc = ORC_OK;
    for (int it = 0; it < n_iter; it++) {
        rc = orc_map_decode(sys, p1, N, n, S, I, next_state, output, noise_variance, L1, 0, Le, bits);   /* :315 */
        if (rc) break;
        for (int64_t i = 0; i < N; i++) Le[i] = Le[i] - L1[i];                            /* :318 */
        for (int64_t i = 0; i < N; i++) L2in[i] = Le[perm[i]];                            /* :319 */
        int mode = (it == n_iter - 1) ? 1 : 0;                                            /* :320-323 */
        rc = orc_map_decode(sysi, p2, N, n, S, I, next_state, output, noise_variance, L2in, mode, L2, bits); /* :326 */
        if (rc) break;
        for (int64_t i = 0; i < N; i++) L1[perm[i]] = L2[i] - L2in[i];                    /* :328-329 */
    }
    for (int64_t i = 0; i < N; i++) decoded[perm[i]] = bits[i];                           /* :331 */
    free(L1); free(Le); free(L2in); free(sysi); free(L2); free(bits);
    return rc;
}

/* =============================================================================================
 * LDPC belief propagation -- ldpc.py:144-254.
 * The sparse matrix is given as an edge list sorted by (check, variable) = the row-major COO order
 * SciPy holds `message_matrix` in (probed: coo.multiply(dense) returns row-major COO), so that
 * `sum(1)` accumulates a row's entries in increasing variable order and `sum(0)` a column's
 * entries in increasing check order (coo_matvec walks the entries in storage order).
 * alg: 0 SPA, 1 MSA.  llr is clipped IN PLACE (:186).  Outputs are block-major [n_blocks][n_v]
 * (the caller applies the reference's final order='F' reshape, :251-253).
 * ===========================================================================================*/
static double clipd(double v, double lo, double hi)
{
    if (isnan(v)) return v;               /* np.clip propagates NaN */
    return v < lo ? lo : (v > hi ? hi : v);
}

int orc_ldpc_bp_decode(double *llr, int64_t n_blocks, int n_v, int n_c, int64_t n_edges,
                       const int32_t *edge_c, const int32_t *edge_v, int alg, int n_iters,
                       int8_t *dec_word, double *out_llrs, int32_t *iters_done)
{
    if (alg != 0 && alg != 1) return ORC_EINVAL;
    int64_t total = n_blocks * n_v;
    for (int64_t i = 0; i < total; i++) llr[i] = clipd(llr[i], -500, 500);              /* :186 */
    for (int64_t i = 0; i < total; i++) { dec_word[i] = (int8_t)(signbit(llr[i]) ? 1 : 0); out_llrs[i] = llr[i]; } /* :193-194 */
    double *M = malloc(sizeof(double) * n_edges), *msum = malloc(sizeof(double) * n_v);
    double complex *lsum = malloc(sizeof(double complex) * n_c);
    double *prod = malloc(sizeof(double) * n_c);
    int *syn = malloc(sizeof(int) * n_c);
    int64_t *row_ptr = calloc(n_c + 1, sizeof(int64_t));
    if (!M || !msum || !lsum || !prod || !syn || !row_ptr) return ORC_ENOMEM;
    for (int64_t e = 0; e < n_edges; e++) row_ptr[edge_c[e] + 1]++;
    for (int c = 0; c < n_c; c++) row_ptr[c + 1] += row_ptr[c];

    for (int64_t blk = 0; blk < n_blocks; blk++) {                                      /* :197 */
        double *l = llr + blk * n_v;
        int8_t *dw = dec_word + blk * n_v;
        double *ol = out_llrs + blk * n_v;
        for (int64_t e = 0; e < n_edges; e++) M[e] = 1.0 * l[edge_v[e]];                /* :199 */
        int it;
        for (it = 0; it < n_iters; it++) {                                              /* :202 */
            memset(syn, 0, sizeof(int) * n_c);
            for (int64_t e = 0; e < n_edges; e++) syn[edge_c[e]] += dw[edge_v[e]];
            int all_even = 1;
            for (int c = 0; c < n_c; c++) if (syn[c] % 2 != 0) { all_even = 0; break; }
            if (all_even) break;                                                        /* :205-206 */
            if (alg == 0) {
                /* NaN signs (an LLR of 0 is an expected input, :214): NumPy's tanh and its complex log2 / exp2 product return a POSITIVE
                 * NaN for a NaN argument (glibc keeps the argument's sign); inf * 0 generates x86's negative NaN in both; pinned by
                 * tests/golden/abnormal.npz (spaz_*) */
                for (int64_t e = 0; e < n_edges; e++) { M[e] *= .5; M[e] = tanh(M[e]); if (isnan(M[e])) M[e] = NAN; } /* :210-211 */
                for (int c = 0; c < n_c; c++) lsum[c] = 0;
                for (int64_t e = 0; e < n_edges; e++) {                                  /* :217-218 */
                    double complex lg = clog(M[e] + 0.0 * CI);     /* numpy nc_log2: clog then *LOG2E on both parts */
                    lg = (creal(lg) * 1.442695040888963407359924681001892137) +
                         (cimag(lg) * 1.442695040888963407359924681001892137) * CI;
                    lsum[edge_c[e]] += lg;
                }
                for (int c = 0; c < n_c; c++) {                                          /* :219 np.exp2(...).real */
                    double complex a = (creal(lsum[c]) * 0.693147180559945309417232121458176568) +
                                       (cimag(lsum[c]) * 0.693147180559945309417232121458176568) * CI;
                    prod[c] = creal(cexp(a));
                    if (isnan(prod[c])) prod[c] = NAN;
                }
                for (int64_t e = 0; e < n_edges; e++) {
                    double v = 1 / M[e];                                                 /* :222 */
                    v = v * prod[edge_c[e]];                                             /* :223 */
                    v = clipd(v, -1, 1);                                                 /* :224 */
                    v = atanh(v);                                                        /* :225 */
                    v *= 2;                                                              /* :226 */
                    M[e] = clipd(v, -500, 500);                                          /* :227 */
                }
            } else {
                for (int c = 0; c < n_c; c++) {                                          /* :231-238 */
                    int64_t b0 = row_ptr[c], b1 = row_ptr[c + 1];
                    int64_t deg = b1 - b0;
                    double row[4096];
                    if (deg > 4096) return ORC_EINVAL;
                    for (int64_t j = 0; j < deg; j++) row[j] = M[b0 + j];
                    for (int64_t j = 0; j < deg; j++) {
                        double sp = 1.0, mn = INFINITY;
                        for (int64_t q = 0; q < deg; q++) {
                            if (q == j) continue;
                            double v = row[q];
                            double sg = (v > 0) - (v < 0);                               /* np.sign */
                            if (isnan(v)) sg = v;
                            sp *= sg;
                            double av = fabs(v);
                            if (av < mn || isnan(av)) mn = av;
                        }
                        M[b0 + j] = sp * mn;
                    }
                }
            }
            for (int v = 0; v < n_v; v++) msum[v] = 0;                                   /* :243 */
            for (int64_t e = 0; e < n_edges; e++) msum[edge_v[e]] += M[e];
            for (int64_t e = 0; e < n_edges; e++) {                                      /* :244-245 */
                double m = M[e] * -1;
                m += 1.0 * (msum[edge_v[e]] + l[edge_v[e]]);
                M[e] = m;
            }
            for (int v = 0; v < n_v; v++) {                                              /* :247-248 */
                ol[v] = msum[v] + l[v];
                dw[v] = (int8_t)(signbit(ol[v]) ? 1 : 0);
            }
        }
        if (iters_done) iters_done[blk] = it;
    }
    free(M); free(msum); free(lsum); free(prod); free(syn); free(row_ptr);
    return ORC_OK;
}

/* =============================================================================================
 * Modem.demodulate -- modulation.py:100-141
 * ===========================================================================================*/
int orc_demod_soft(const double *y_re_im, int64_t nsym, const double *const_re_im, int M, int nbits,
                   double noise_var, double *llr /* [nsym*nbits] */)
{
    for (int64_t i = 0; i < nsym; i++) {
        double complex cur = y_re_im[2 * i] + y_re_im[2 * i + 1] * CI;
        for (int bit_index = 0; bit_index < nbits; bit_index++) {
            double num = 0, den = 0;
            for (int m = 0; m < M; m++) {
                double complex sym = const_re_im[2 * m] + const_re_im[2 * m + 1] * CI;
                double a = cabs(cur - sym);                       /* abs(current_symbol - symbol) */
                double e = exp((-(a * a)) / noise_var);           /* (-abs(..)**2)/noise_var (:134,136) */
                if ((m >> bit_index) & 1) num += e; else den += e;
            }
            llr[i * nbits + nbits - 1 - bit_index] = log(num / den);   /* :137 */
        }
    }
    return ORC_OK;
}

int orc_demod_hard(const double *y_re_im, int64_t nsym, const double *const_re_im, int M, int nbits,
                   int8_t *bits /* [nsym*nbits] */)
{
    for (int64_t i = 0; i < nsym; i++) {
        double complex cur = y_re_im[2 * i] + y_re_im[2 * i + 1] * CI;
        int best = 0;
        double bd = INFINITY;
        for (int m = 0; m < M; m++) {                              /* abs(y - c[:,None]).argmin(0) (:122) */
            double complex sym = const_re_im[2 * m] + const_re_im[2 * m + 1] * CI;
            double a = cabs(cur - sym);
            if (m == 0 || a < bd) { bd = a; best = m; }
        }
        for (int b = 0; b < nbits; b++) bits[i * nbits + b] = (int8_t)((best >> (nbits - 1 - b)) & 1); /* :123 */
    }
    return ORC_OK;
}

/* Batch convenience for the CPU baseline timing: B codewords back to back through orc_viterbi_decode
 * (one foreign call per thread keeps the Python GIL out of the timed region). */
int orc_viterbi_decode_batch(const double *coded, int64_t B, int64_t len, int k, int n, int total_memory, int S, int I,
                             const int32_t *next_state, const int32_t *output, int tb_depth, int type,
                             int64_t *decoded /* [B][L] */, int64_t L)
{
    for (int64_t b = 0; b < B; b++) {
        int rc = orc_viterbi_decode(coded + b * len, len, k, n, total_memory, S, I, next_state, output, tb_depth, type,
                                    decoded + b * L, NULL);
        if (rc) return rc;
    }
    return ORC_OK;
}
