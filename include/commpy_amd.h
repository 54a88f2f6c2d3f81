/*
 * commpy_amd.h -- C-ABI of libcommpy_amd.so, the MI355X (gfx950) decoding engine.
 *
 * The reference (veeresht/CommPy 0.8.0, pure Python) has no FFI/plugin layer: its boundary for the
 * hot path is the set of Python callables exported by commpy/channelcoding/__init__.py:65-71 and
 * commpy/modulation.py:35-36.  Each entry point below replaces the BODY of one of those callables;
 * the Python mirror in commpy_amd/ keeps the reference signatures and reaches these symbols with
 * ctypes (see INTEGRATION.md for the stub a CommPy maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, <0 on error (CPX_E*); cpx_last_error() gives a
 *     thread-local message.  The Python layer maps errors to the reference's exception types.
 *   - "host" entry points take caller-owned host buffers (C-contiguous NumPy memory), copy
 *     H2D/D2H themselves on the library's stream and return after completion.
 *   - "_dev" entry points take DEVICE pointers and a hipStream_t (as void*; NULL = the library's
 *     own stream), enqueue the kernels and return immediately (asynchronous); used by bench.py and
 *     the multi-GPU path where inputs are already resident in HBM.
 *   - handles (cpx_trellis/cpx_ldpc/cpx_modem) own small device-side tables; they belong to the
 *     device that was current at creation and are freed by the matching *_destroy.
 *   - no CPU fallback exists: without a usable HIP device every compute entry point fails.
 */
#ifndef COMMPY_AMD_H
#define COMMPY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is ALL it exports */
#pragma GCC visibility push(default)

#define CPX_OK 0
#define CPX_EINVAL (-1)   /* bad argument (Python: ValueError) */
#define CPX_EHIP (-2)     /* HIP runtime error (Python: RuntimeError) */
#define CPX_ENOMEM (-3)   /* allocation failure */
#define CPX_ENODEV (-4)   /* no usable gfx950 device */
#define CPX_ELIMIT (-5)   /* argument exceeds a documented engine limit */

typedef struct cpx_trellis cpx_trellis;
typedef struct cpx_ldpc cpx_ldpc;
typedef struct cpx_modem cpx_modem;

/* ---- runtime ---------------------------------------------------------------------------------*/
const char *cpx_last_error(void);
int cpx_version(void);
/* "full:<sha16>;viterbi:<sha16>" -- digests of the sources this library was compiled from (commpy_amd/build.py).  Measurement
 * files (profiles/...pmc.json) carry the id of the library that produced them; bench.py only quotes their counters when the
 * "viterbi" part equals the loaded library's. */
const char *cpx_build_id(void);
int cpx_device_count(int *n);
int cpx_set_device(int device);
int cpx_get_device(int *device);
/* Name of the (dominant) kernel the last decoder call of the calling thread launched, e.g.
 * "viterbi_cw_fused_kernel<6,0155,0117,soft,28>": benchmarks and tests report what really ran.  Decoders with a "detect and redo"
 * path append "; redo: n of N ..." -- n is written by the redo kernel itself, so that suffix is only valid once the issuing stream
 * has been synchronised (the host-buffer entry points return synchronised; after a *_dev call, cpx_stream_sync first). */
int cpx_last_kernel(char *name, int cap);
/* Precision mode of the process ("fp64-parity default vs fp32-fast"), initial value from the environment variable
 * CPX_PRECISION.  "fp64-parity" (default): every kernel computes in float64 in the reference's operation order -- the mode all
 * parity claims are made in.  "fp32-fast": kernels that have a float32 variant use it -- the fused codeword-per-lane Viterbi
 * kernel (float32 path metrics, correlation branch metric; measured mismatch rate in DESIGN.md 4.1) and the LDS-resident LDPC
 * kernels (float32 messages, hardware exp2 / log2 / reciprocal for sum-product; same decoded words and frame error rate on the
 * measured workloads) and soft demodulation (float32 log-sum-exp, |dLLR| <= 2e-5 + 4e-6 |LLR|, finite where the
 * reference's sums underflow) -- NOT bit-exact and NOT under the 1e-5 LLR criterion; every other kernel is unaffected.
 * cpx_last_kernel shows "f32" in the name when a variant ran. */
int cpx_set_precision(const char *mode);
int cpx_get_precision(void);   /* 0 fp64-parity, 1 fp32-fast */
int cpx_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes);
int cpx_malloc(void **dptr, size_t bytes);
int cpx_free(void *dptr);
/* cpx_memset / cpx_memcpy_h2d / cpx_memcpy_d2h are synchronous AND ordered with the library stream: they run after every `_dev`
 * call that was given stream = NULL has finished (work on a caller's own stream needs cpx_stream_sync first). */
int cpx_memset(void *dptr, int value, size_t bytes);
int cpx_memcpy_h2d(void *dst, const void *src, size_t bytes);
int cpx_memcpy_d2h(void *dst, const void *src, size_t bytes);
int cpx_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream);
int cpx_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream);
int cpx_memcpy_d2d_async(void *dst, const void *src, size_t bytes, void *stream);
int cpx_stream_create(void **stream);       /* a non-blocking hipStream_t on the current device */
int cpx_stream_destroy(void *stream);
int cpx_stream_sync(void *stream);          /* NULL = the library's stream */
int cpx_release_workspace(void);            /* free the per-stream scratch arenas the decoders keep between calls */
void *cpx_default_stream(void);             /* the library's own hipStream_t */
/* HIP-event timer on a stream (bench.py times the kernels on the stream they are launched on) */
int cpx_timer_create(void **timer);
int cpx_timer_start(void *timer, void *stream);
int cpx_timer_stop(void *timer, void *stream);
int cpx_timer_elapsed_ms(void *timer, float *ms);   /* synchronises on the stop event */
int cpx_timer_destroy(void *timer);
/* Shader-clock probe (round 5; measurement support, no reference counterpart): cpx_sclk_probe_start launches one mostly sleeping
 * wavefront on a stream of its own for `spin_ms` milliseconds of the constant-rate device clock; cpx_sclk_probe_read waits for it
 * and returns the average shader clock (MHz) over the interval it really covered (`interval_ms`) -- the clock the kernels on the
 * OTHER streams ran at meanwhile -- and frees the probe.  bench.py records it next to its per-launch times.  Every outstanding
 * probe owns one of 16 result slots of its device (a 17th start: CPX_ELIMIT); cpx_sclk_probe_destroy gives back a probe that is never
 * read (it waits for the probe's wavefront first). */
int cpx_sclk_probe_start(void **probe, double spin_ms);
int cpx_sclk_probe_read(void *probe, double *sclk_mhz, double *interval_ms);
int cpx_sclk_probe_destroy(void *probe);

/* ---- convolutional codes: Viterbi ---------------------------------------------------------------
 * cpx_trellis_create: device copy of the code description built by the host Trellis class.
 *   Replaces nothing by itself; it carries Trellis.next_state_table / output_table
 *   (reference commpy/channelcoding/convcode.py:117-255) to the device.  Tables are [S][I]
 *   row-major int32.  Predecessor lists are derived in np.where order (convcode.py:561-572), which
 *   defines the ACS tie-break.  Limits: S = 2^m <= 65536, I = 2^k <= 256, n <= 16; every state needs exactly I incoming branches
 *   (the reference indexes pmetrics[number_inputs], convcode.py:604-629).  The specialised kernels serve S <= 128, k <= 2, n <= 6;
 *   the general kernels (viterbi_generic.hip, bcjr_exact.hip) the rest.
 */
int cpx_trellis_create(int k, int n, int n_states, int n_inputs, const int32_t *next_state_table,
                       const int32_t *output_table, cpx_trellis **out);
int cpx_trellis_destroy(cpx_trellis *t);

/* decoding_type */
#define CPX_VIT_HARD 0
#define CPX_VIT_SOFT 1
#define CPX_VIT_UNQUANTIZED 2

/* cpx_viterbi_decode_batch replaces the body of
 *   viterbi_decode(coded_bits, trellis, tb_depth, decoding_type)   convcode.py:661-749
 * (with _acs_traceback :590-657, _compute_branch_metrics :575-587) for B independent codewords.
 *   coded      [B][len] float64 (hard: 0/1 values; soft: LLR log P1/P0, clipped to +-500 inside;
 *              unquantized: real symbols)
 *   L          number of decoded bits per codeword = int(len*k/n)  (convcode.py:698)
 *   n_steps    trellis steps actually run = int((L+total_memory)/k) - 1  (convcode.py:721)
 *   tb_depth   traceback depth (>= 2), the caller resolves the default min(5*m, L)
 *   bits       [B][L] uint8 decoded bits, tail included (convcode.py:749)
 * Decision rule: the bit(s) of step s come from the survivor of the
 * first-minimum state at step min(s+tb_depth-2, n_steps), ties -> lowest index.
 * Kernel selection is internal and does not change a single output bit: batches of >= 0.45 * (SIMDs of the device) * 64
 * codewords of a rate-1/2 shift-register code of 4 .. 64 states run one codeword per lane (csrc/viterbi_cw.hip): a single fused
 * kernel with the generators compiled in -- K = 7 (133,171), (171,133) in both polynomial formats and Wifi80211's (5,43) for
 * tb_depth <= 48; K = 3 (5,7) and K = 5 (23,35) at their default depth -- or, for any other pair whose generators both tap the
 * input and the oldest register bit, with a run-time code table (default depth 5 * memory); an add-compare-select + a traceback kernel with a 9 B per codeword-step device workspace beyond that;
 * every other trellis of up to 128 states, k <= 2, n <= 6 runs one trellis state per lane (csrc/viterbi.hip); the rest of the
 * reference's argument domain (up to 65536 states, k <= 8, n <= 16, any traceback depth) one workgroup per codeword
 * (csrc/viterbi_generic.hip: slow, complete).  A NaN among 'soft' inputs is handled as the reference
 * handles it (convcode.py:719 lets it through the clip): flagged codewords are decoded again by a NaN-exact instantiation.
 * cpx_viterbi_set_path (or the environment variable CPX_VITERBI_PATH = wave | cw | cw! | cw2 | cw2! | general at load time)
 * overrides the choice (tests, benchmarks); cpx_last_kernel reports which kernel ran.
 */
int cpx_viterbi_decode_batch(const cpx_trellis *t, const double *coded, int64_t B, int64_t len,
                             int64_t L, int64_t n_steps, int tb_depth, int decoding_type, uint8_t *bits);
int cpx_viterbi_decode_batch_dev(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len,
                                 int64_t L, int64_t n_steps, int tb_depth, int decoding_type,
                                 uint8_t *d_bits, void *stream);
/* Kernel-path override for tests and benchmarks (initial value: environment variable CPX_VITERBI_PATH):
 * NULL / "" / "auto" = automatic, "wave", "cw", "cw!", "cw2", "cw2!" as described above, "general" = the general kernel. */
int cpx_viterbi_set_path(const char *mode);
/* Per-pair code objects for the codeword-per-lane kernels (round 6).  A rate-1/2 code of full constraint length that is not one of
 * the built-in generator pairs runs the TABLE-DRIVEN fused kernel (branch metric selected by VGPR index mode: 1.78 ms on the config-2
 * geometry where a built-in pair takes 1.55).  The same source file compiled with the pair as template arguments
 *     hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -std=c++17 -ffp-contract=off \
 *           -DCPX_VIT_SPEC_LG=<lg> -DCPX_VIT_SPEC_G0=<g0>u -DCPX_VIT_SPEC_G1=<g1>u -c commpy_amd/csrc/viterbi_cw.hip
 * gives a code object with the six fused kernels of THAT pair; commpy_amd/jit.py builds and caches it by (generators, build id).
 *   cpx_trellis_viterbi_spec_query   *lg = log2(states) and the generators in the kernel template's convention when the trellis would
 *                                    gain from such an object (else *lg = 0: built-in pair, other structure, or already attached)
 *   cpx_trellis_attach_viterbi_code  loads the image (hipModuleLoadData) and looks its kernels up BY THIS TRELLIS'S GENERATORS: an
 *                                    image of another pair or of other sources is refused (CPX_EINVAL), the trellis stays as it was;
 *                                    afterwards viterbi_decode batches that take the codeword-per-lane path launch the module's
 *                                    kernels (cpx_last_kernel: "... (code object of this pair)"); results are bit-identical
 *   cpx_trellis_detach_viterbi_code  back to the table-driven kernel (tests) */
int cpx_trellis_viterbi_spec_query(const cpx_trellis *t, int *lg, unsigned *g0, unsigned *g1);
int cpx_trellis_attach_viterbi_code(cpx_trellis *t, const void *image, size_t bytes);
int cpx_trellis_detach_viterbi_code(cpx_trellis *t);
int cpx_trellis_has_viterbi_code(const cpx_trellis *t);   /* 1: a code object is attached */
/* Fused hard demodulation + hard-decision Viterbi: replaces the pair
 *   bits = modem.demodulate(y, 'hard')            commpy/modulation.py:121-123
 *   viterbi_decode(bits, trellis, tb_depth, 'hard')  commpy/channelcoding/convcode.py:578-580, 661-749
 * for B codewords of nsym symbols each (y [B][nsym][2] float64 = complex128).  The kernel takes the hard decisions
 * itself while it prepares the branch metrics: the int8 bits are never written to (or re-read from) HBM -- 16 B in per
 * symbol instead of 16 B in + nb B out + 8 nb B in.  Output identical to the two calls.  len = nsym * bits-per-symbol
 * takes the place of len(coded_bits); trellises above 64 states return CPX_ELIMIT (use the two calls). */
int cpx_demod_hard_viterbi_batch(const cpx_modem *m, const cpx_trellis *t, const double *y_re_im, int64_t B,
                                 int64_t nsym, int64_t L, int64_t n_steps, int tb_depth, uint8_t *bits);
int cpx_demod_hard_viterbi_batch_dev(const cpx_modem *m, const cpx_trellis *t, const double *d_y_re_im, int64_t B,
                                     int64_t nsym, int64_t L, int64_t n_steps, int tb_depth, uint8_t *d_bits,
                                     void *stream);
/* Same as cpx_viterbi_decode_batch with the result widened to the reference's return type (`decoded_bits` is an
 * int array, convcode.py:711/749): bits64 [B][L] int64.  The compact bits cross PCIe, host threads widen them
 * straight into the caller's array: a single-threaded astype of 67 M bits costs more than decoding them. */
int cpx_viterbi_decode_batch_i64(const cpx_trellis *t, const double *coded, int64_t B, int64_t len,
                                 int64_t L, int64_t n_steps, int tb_depth, int decoding_type, int64_t *bits64);

/* ---- turbo codes: BCJR / MAP ---------------------------------------------------------------------
 * cpx_map_decode_batch replaces map_decode(sys, non_sys, trellis, noise_variance, L_int, mode)
 *   commpy/channelcoding/turbo.py:163-251 (+ _backward_recursion :78-111,
 *   _forward_recursion_decoding :114-158, _compute_branch_prob :62-76) for B codewords.
 *   sys, par, L_int [B][N] float64; L_ext [B][N] float64 (= L_int + log(app1/app0), turbo.py:145);
 *   bits [B][N] uint8 (all zero unless want_bits, i.e. mode == 'decode').
 * cpx_turbo_decode_batch replaces turbo_decode(...) turbo.py:254-333: n_iter x (MAP1, interleave,
 *   MAP2, de-interleave); perm = interleaver.p_array (interleavers.py:13-47), shared by the batch;
 *   L_int may be NULL (zeros).  bits [B][N] uint8, already de-interleaved.
 * Limits: k = 1 (I == 2, like the reference's priors[2]), n >= 2; N < 2^24 (map), N < 2^21 (turbo).  2 .. 16 states run the
 *   wave-pair kernels of bcjr.hip; larger trellises the literal kernel of bcjr_exact.hip alone (one codeword per lane, scratch
 *   (N + 1) * S doubles per lane: CPX_ELIMIT only beyond 4 GB for 64 lanes).
 * Inputs for which the reference's absolute-scale recursion underflows (turbo.py:62-76, :238-240: symbol amplitudes of 5 - 20 at
 *   sigma^2 <= 0.1, priors of e^-200) or that are not finite give what the reference gives -- NaN / +-inf LLRs, their decisions --:
 *   the fast kernels flag such codewords and a literal absolute-scale kernel decodes them again (blocks up to the scratch limit of
 *   that path).
 */
int cpx_map_decode_batch(const cpx_trellis *t, const double *sys, const double *par, const double *L_int,
                         int64_t B, int64_t N, double noise_variance, int want_bits, double *L_ext,
                         uint8_t *bits);
int cpx_map_decode_batch_dev(const cpx_trellis *t, const double *d_sys, const double *d_par,
                             const double *d_L_int, int64_t B, int64_t N, double noise_variance,
                             int want_bits, double *d_L_ext, uint8_t *d_bits, void *stream);
int cpx_turbo_decode_batch(const cpx_trellis *t, const double *sys, const double *p1, const double *p2,
                           const double *L_int_or_null, const int32_t *perm, int64_t B, int64_t N,
                           double noise_variance, int n_iter, uint8_t *bits);
int cpx_turbo_decode_batch_dev(const cpx_trellis *t, const double *d_sys, const double *d_p1,
                               const double *d_p2, const double *d_L_int_or_null, const int32_t *d_perm,
                               int64_t B, int64_t N, double noise_variance, int n_iter, uint8_t *d_bits,
                               void *stream);

/* ---- LDPC belief propagation ---------------------------------------------------------------------
 * cpx_ldpc_create: device copy of the Tanner graph produced by get_ldpc_code_params
 *   (commpy/channelcoding/ldpc.py:51-141): the edge list sorted by (check, variable) -- the
 *   row-major order SciPy keeps `message_matrix` in, which fixes the summation orders of
 *   ldpc.py:217-219 and :243.
 * cpx_ldpc_bp_decode_batch replaces ldpc_bp_decode(llr_vec, params, alg, n_iters) ldpc.py:144-254.
 *   llr       [B][n_v] float64, positive = bit 0 (ldpc.py:193); CLIPPED IN PLACE to +-500 (:186)
 *   alg       0 'SPA', 1 'MSA'
 *   dec_word  [n_v][B] int8 and out_llrs [n_v][B] float64: one block per COLUMN, the reference's
 *             output layout (ldpc.py:251-253)
 *   iters_done[B] int32 executed iterations per block (early exit ldpc.py:205-206), may be NULL
 * Limits (CPX_ELIMIT): n_v, n_c < 2^24.  Checks of up to 32 edges run the LDS-resident / tiled kernels (a row lives in registers
 * / one 32-bit sign mask); a code with a larger check is decoded by the literal kernel (ldpc_exact_kernel) alone.
 */
#define CPX_LDPC_SPA 0
#define CPX_LDPC_MSA 1
int cpx_ldpc_create(int n_vnodes, int n_cnodes, int64_t n_edges, const int32_t *edge_check,
                    const int32_t *edge_var, cpx_ldpc **out);
int cpx_ldpc_destroy(cpx_ldpc *c);
/* Compiled design ("blob"): the device tables of a Tanner graph -- sorted edge list, row / column
 * pointers, the variable-major view and the padded node tables the passes read -- in one position-independent,
 * checksummed byte string, so that a design file (ldpc.py:51-141 / write_ldpc_params :257-299) is compiled ONCE and the
 * result cached under the file's hash instead of being re-derived per code object.
 *   cpx_ldpc_blob_build        host only (no device needed); blob == NULL queries the size into *need;
 *                              blob must be 8-byte aligned
 *   cpx_ldpc_blob_info         validates a blob (magic, sizes, checksum, every index in range) and returns its dimensions
 *   cpx_ldpc_create_from_blob  validates, then uploads; cpx_ldpc_create == blob_build + create_from_blob */
int cpx_ldpc_blob_build(int n_vnodes, int n_cnodes, int64_t n_edges, const int32_t *edge_check, const int32_t *edge_var,
                        void *blob, size_t cap, size_t *need);
int cpx_ldpc_blob_info(const void *blob, size_t nbytes, int *n_vnodes, int *n_cnodes, int64_t *n_edges,
                       int *max_cnode_deg, int *max_vnode_deg);
int cpx_ldpc_create_from_blob(const void *blob, size_t nbytes, cpx_ldpc **out);
int cpx_ldpc_bp_decode_batch(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters,
                             int8_t *dec_word, double *out_llrs, int32_t *iters_done);
int cpx_ldpc_bp_decode_batch_dev(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters,
                                 int8_t *d_dec_word, double *d_out_llrs, int32_t *d_iters_done,
                                 void *stream);
/* The same decode with BLOCK-MAJOR outputs: dec_word [B][n_v] int8 and out_llrs [B][n_v] float64, one block per ROW.  This
 * is the memory the reference's own results live in: ldpc.py:251-253 returns `x.reshape(-1, n_blocks, order='F')`, an
 * F-ordered (n_v, n_blocks) VIEW of a buffer in which every block is contiguous -- so the Python layer wraps these arrays as
 * `dec.T` / `out.T` and returns objects with the reference's shape, dtype, values AND strides, and no transposition pass exists
 * anywhere (the [n_v][B] entry points above cost one: a staging buffer written, read back and written again).  HBM traffic of a
 * block is then SURVEY 8d's resident model exactly: llr in, out_llrs and dec_word out, 17 n_v bytes. */
int cpx_ldpc_bp_decode_batch_bm(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters,
                                int8_t *dec_word, double *out_llrs, int32_t *iters_done);
int cpx_ldpc_bp_decode_batch_bm_dev(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters,
                                    int8_t *d_dec_word, double *d_out_llrs, int32_t *d_iters_done,
                                    void *stream);
/* Two implementations: the LDS-resident path (csrc/ldpc_resident.hip: the decoder state of a block lives in the LDS of one
 * compute unit, one persistent launch, blocks retire and are replaced individually) whenever that state fits, else the tiled
 * HBM path (csrc/ldpc.hip).  Min-sum: the same arithmetic, identical results.  Sum-product: the tiled path and "resident-log"
 * share the log-domain row (identical results); the default resident kernel keeps the state as likelihood ratios -- no exp / log
 * inside an iteration, same dec_word, iteration counts and out_llrs contract, blocks it cannot carry (a NaN, an
 * iteration saturated in more than half of its rows) decoded again in place with the log-domain row (tests/test_ldpc_resident_gpu.py).
 * cpx_ldpc_set_path("auto" | "tiled" | "resident" | "resident-log") forces one (initial value: environment variable
 * CPX_LDPC_PATH); the "resident" modes fail with CPX_EINVAL instead of falling back.  cpx_last_kernel names what ran. */
int cpx_ldpc_set_path(const char *mode);

/* ---- PSK/QAM demodulation ------------------------------------------------------------------------
 * cpx_modem_create: device copy of Modem.constellation (commpy/modulation.py:68-77,159-172),
 *   M = 2^nbits complex points as [M][2] float64 (re, im), index = Gray-reordered symbol label.
 * cpx_demod_soft replaces Modem.demodulate(y, 'soft', noise_var) modulation.py:125-137:
 *   llr[i*nb + nb-1-b] = log( sum_{m:(m>>b)&1} e^{-|y_i-c_m|^2/noise_var} / sum_{m:!..} ... ),
 *   positive = bit 1, sums in increasing m.
 * cpx_demod_hard replaces Modem.demodulate(y, 'hard') modulation.py:121-123: first-min nearest
 *   point, MSB-first int8 bits.
 *   y [Ns][2] float64 (complex128 memory), llr [Ns*nb] float64, bits [Ns*nb] int8.
 */
int cpx_modem_create(const double *constellation_re_im, int M, cpx_modem **out);
int cpx_modem_destroy(cpx_modem *m);
/* Implementation choice of the soft demodulator for tests and A/B runs (initial value: environment variable CPX_DEMOD):
 * NULL / "auto": square QAM of 64 points and more takes the geometric-progression form (equally spaced Gray-labelled levels:
 * TWO exponentials and one division per axis, every other level by two multiplications) with table-driven exp / log (32-entry
 * tables in LDS); PSK and arbitrary constellations take the table-driven point-by-point kernel (round 6).  Both store a wave's LLRs
 * as one contiguous run of 16-byte stores: an output pointer that is only 8-byte aligned is served by the literal kernel instead.
 * "libm": the same forms with the library's exp / log (generic constellations: the literal kernel); "plain": one exponential per
 * level everywhere.
 * All are within 1e-5 of modulation.py:125-137 (measured: 1e-13); symbols near the underflow range are decided point by
 * point in the reference's order either way. */
int cpx_demod_set_path(const char *mode);
int cpx_demod_soft(const cpx_modem *m, const double *y_re_im, int64_t Ns, double noise_var, double *llr);
int cpx_demod_soft_dev(const cpx_modem *m, const double *d_y_re_im, int64_t Ns, double noise_var,
                       double *d_llr, void *stream);
/* the same with every LLR multiplied by `scale` on its way out: scale = -1 is the sign flip between Modem.demodulate (log P1/P0)
 * and ldpc_bp_decode (log P0/P1), test_ldpc.py:53-54, without a second pass over the LLRs; scale = 1 is cpx_demod_soft_dev */
int cpx_demod_soft_scaled_dev(const cpx_modem *m, const double *d_y_re_im, int64_t Ns, double noise_var, double scale,
                              double *d_llr, void *stream);
int cpx_demod_hard(const cpx_modem *m, const double *y_re_im, int64_t Ns, int8_t *bits);
int cpx_demod_hard_dev(const cpx_modem *m, const double *d_y_re_im, int64_t Ns, int8_t *d_bits,
                       void *stream);

/* ---- link-simulation stages around the decoders ("next" rows) ---------------------------
 * Device-resident (all pointers are device pointers, asynchronous on `stream`), so that a Monte-Carlo
 * BER sweep (commpy/links.py:155-267, commpy/wifi80211.py:132-216) never leaves HBM.
 *   cpx_random_bits_dev       message bits (links.py:229); Philox4x32-10 counter stream (seed, stream_id)
 *   cpx_conv_encode_batch_dev conv_encode(msg, trellis, termination) convcode.py:475-558 for B rows:
 *                             msg [B][nmsg] uint8 -> coded [B][nout] uint8 (nout as the reference computes
 *                             number_outbits; rsc = code_type == 'rsc'; terminate = (termination == 'term') for
 *                             recursive codes -- the only case whose tail is clocked, convcode.py:538 -- and
 *                             (termination != 'cont') otherwise; positions past the clocked steps are zero)
 *   cpx_gather_u8_dev         out[b][j] = in[b][idx[j]]: puncturing (convcode.py:752-774) with the kept
 *                             positions as idx
 *   cpx_gather_f64_dev        out[b][j] = idx[j] >= 0 ? in[b][idx[j]] : 0: depuncturing (convcode.py:777-804)
 *   cpx_modulate_dev          Modem.modulate modulation.py:79-98: nb bits MSB-first -> constellation point
 *   cpx_awgn_dev              y = x + scale_re*n_re + 1j*scale_im*n_im, n ~ N(0,1) (channels.py:37-55)
 *   cpx_bsc_dev / cpx_bec_dev bsc(input_bits, p_t) commpy/channels.py:652-673 / bec(input_bits, p_e) :630-649: one uniform
 *                             draw per bit (Philox stream (seed, stream_id)), flipped / erased to -1 where the draw is
 *                             <= p.  Outputs: int8 (the reference's integer bits) and / or float64 (what
 *                             viterbi_decode(..., 'hard') takes, BASELINE config 1); either may be NULL.  p outside
 *                             [0, 1] (or NaN): CPX_EINVAL
 *   cpx_count_errors_dev      errs[b][c] = sum(msg[b, chunk c] ^ dec[b, chunk c]) (links.py:252-256)
 *   cpx_scale_f64_dev         y = a*x, e.g. the sign flip between Modem.demodulate (log P1/P0) and ldpc_bp_decode
 *                             (log P0/P1), test_ldpc.py:53-54
 */
int cpx_random_bits_dev(uint8_t *d_bits, int64_t n, uint64_t seed, uint64_t stream_id, void *stream);
int cpx_conv_encode_batch_dev(const cpx_trellis *t, const uint8_t *d_msg, int64_t B, int64_t nmsg, int terminate,
                              int rsc, uint8_t *d_coded, int64_t nout, void *stream);
int cpx_gather_u8_dev(const uint8_t *d_in, int64_t B, int64_t nin, const int32_t *d_idx, int64_t nout,
                      uint8_t *d_out, void *stream);
int cpx_gather_f64_dev(const double *d_in, int64_t B, int64_t nin, const int32_t *d_idx, int64_t nout,
                       double *d_out, void *stream);
int cpx_modulate_dev(const cpx_modem *m, const uint8_t *d_bits, int64_t nsym, double *d_sym_re_im, void *stream);
int cpx_awgn_dev(const double *d_x_re_im, int64_t n, double scale_re, double scale_im, uint64_t seed,
                 uint64_t stream_id, double *d_y_re_im, void *stream);
int cpx_scale_f64_dev(const double *d_x, int64_t n, double a, double *d_y, void *stream);
int cpx_bsc_dev(const uint8_t *d_bits, int64_t n, double p_t, uint64_t seed, uint64_t stream_id, int8_t *d_out_i8,
                double *d_out_f64, void *stream);
int cpx_bec_dev(const uint8_t *d_bits, int64_t n, double p_e, uint64_t seed, uint64_t stream_id, int8_t *d_out_i8,
                double *d_out_f64, void *stream);
int cpx_count_errors_dev(const uint8_t *d_msg, int64_t msg_stride, const uint8_t *d_dec, int64_t dec_stride,
                         int64_t B, int64_t nchunks, int64_t chunk, int32_t *d_errs, void *stream);

/* ---- the stages of one Monte-Carlo point in front of the decoder as ONE kernel (round 6; same SURVEY 8f rows) ----------------------
 * random bits -> conv_encode ('cont') -> puncturing -> Modem.modulate -> AWGN -> Modem.demodulate('soft') -> depuncturing
 * (links.py:229-250, wifi80211.py:178-206, convcode.py:475-558, :752-804, modulation.py:79-141, channels.py:37-55) for T
 * transmissions of `nbits` message bits, without the five intermediate arrays: every lane produces one symbol's LLRs from its index
 * and the counter-based streams (seed, stream_bits) / (seed, stream_noise).  The results equal those of the staged calls
 * cpx_random_bits_dev(d_msg, T*nbits, seed, stream_bits) ... cpx_gather_f64_dev bit for bit.
 *   cpx_link_front_create   HOST tables: keep_idx[ntx] = coded position of transmitted bit t (NULL: no puncturing), pos_idx[ntx] =
 *                           decoder-input position of transmitted bit t among nde (NULL: t; the others are set to 0.0), both
 *                           increasing.  CPX_ELIMIT when the combination is not what the kernel is built for (feed-forward k = 1
 *                           trellis of <= 64 states, square QAM of 4..256 points with equally spaced Gray levels, a symbol depending on
 *                           <= 49 message bits): the caller keeps the staged calls.  The handle refers to `t` and `m`: destroy it first.
 *   cpx_link_front_run_dev  d_msg [T][nbits] uint8, d_llr [T][nde] float64 (LLR * llr_scale), d_rx_re_im optional [T][ntx/nb][2]
 *                           noisy symbols (NULL: not stored).  scale_re / scale_im as cpx_awgn_dev.  CPX_ELIMIT in the non-default
 *                           demodulator modes (cpx_demod_set_path, fp32-fast) and when 1 / noise_var is not a normal number.
 */
typedef struct cpx_link_front cpx_link_front;
int cpx_link_front_create(const cpx_trellis *t, const cpx_modem *m, int64_t nbits, const int32_t *keep_idx, int64_t ntx,
                          const int32_t *pos_idx, int64_t nde, cpx_link_front **out);
int cpx_link_front_destroy(cpx_link_front *lf);
int cpx_link_front_run_dev(const cpx_link_front *lf, int64_t T, double noise_var, double scale_re, double scale_im,
                           double llr_scale, uint64_t seed, uint64_t stream_bits, uint64_t stream_noise, uint8_t *d_msg,
                           double *d_llr, double *d_rx_re_im, void *stream);

/* ---- channel encoders on the device ("next" rows) --------------------------------
 * Device pointers, asynchronous on `stream`; bit-exact integer work.
 *   cpx_turbo_encode_batch_dev  turbo_encode(msg, trellis1, trellis2, interleaver) commpy/channelcoding/turbo.py:14-59
 *       for B rows: msg [B][N] uint8 -> sys [B][N], p1 [B][N], p2 [B][np2] (np2 >= N; entries past N are 0:
 *       turbo.py:47,53 pass 'rsc' as the termination argument, so conv_encode (convcode.py:538) clocks no
 *       tail and leaves the tail of its output zero; the reference returns len(p2) = 2(N+m2)-m2).
 *       Component codes must be rate 1/2 (the [::2] / [1::2] split of turbo.py:48-49).  perm = the
 *       interleaver's p_array (interleavers.py:45: out[i] = in[p[i]]), int32 [N], 16-byte aligned.
 *       mode: 0 auto, 1 one-codeword-per-lane walk, 2 wave-per-codeword state-map scan (<= 16 states).
 *   cpx_ldpc_encoder_create     packs a GF(2) generator: gen_bits [m][k] uint8 (0/1), parity = gen . msg mod 2 --
 *       `generator_matrix` of build_matrix (ldpc.py:44-48) reduced mod 2; k <= 8192.
 *   cpx_ldpc_encode_batch_dev   triang_ldpc_systematic_encode (ldpc.py:302-354) for B blocks: msg [B][k] uint8 ->
 *       code [B][k+m] uint8, systematic part first (:354); row b is column b of the reference's result.
 */
typedef struct cpx_ldpc_encoder cpx_ldpc_encoder;
int cpx_turbo_encode_batch_dev(const cpx_trellis *t1, const cpx_trellis *t2, const uint8_t *d_msg, int64_t B, int64_t N,
                               const int32_t *d_perm, uint8_t *d_sys, uint8_t *d_p1, uint8_t *d_p2, int64_t np2,
                               int mode, void *stream);
int cpx_ldpc_encoder_create(const uint8_t *gen_bits, int64_t m, int64_t k, cpx_ldpc_encoder **out);
int cpx_ldpc_encoder_destroy(cpx_ldpc_encoder *e);
int cpx_ldpc_encode_batch_dev(const cpx_ldpc_encoder *e, const uint8_t *d_msg, int64_t B, uint8_t *d_code,
                              void *stream);

/* ---- multi-GPU: RCCL collectives of the sharded decode ------------------------------------
 * The path shards by codeword (contiguous blocks of B/G codewords per GPU, tables replicated) and has no exchange
 * step inside a decoder.  Two collectives exist around it:
 *   all-gather of decoded bits (uint8)   -- every GPU ends with the whole [B][L] result, the array the reference
 *                                           returns from viterbi_decode / ldpc_bp_decode (convcode.py:749, ldpc.py:251-254)
 *   all-reduce (sum / max) of int64 / float64 counters -- the error and bit counters of a Monte-Carlo sweep,
 *                                           commpy/links.py:252-260
 * A communicator is formed either by ONE process for several devices (cpx_comm_init_all = ncclCommInitAll; every
 * collective then takes one buffer per local device, in the order of `devices`, issued inside one ncclGroup) or by one
 * process per GPU (cpx_comm_unique_id on rank 0, the 128-byte id handed to the others by the launcher plumbing,
 * cpx_comm_init_rank on every rank with its device current).  librccl.so.1 is dlopen()ed at the first communicator.
 * d_send / d_recv / streams are arrays of nlocal pointers (nlocal = 1 for init_rank); streams == NULL or a NULL entry =
 * the library's stream of that device.  All calls are asynchronous on those streams.
 *   cpx_comm_allgather_u8: d_recv[i] holds nranks * bytes_per_rank bytes, rank r's block at r * bytes_per_rank
 *                          (ragged shards: the caller pads to the largest shard, commpy_amd/parallel.py)
 *   op: 0 = sum, 1 = max */
typedef struct cpx_comm cpx_comm;
int cpx_comm_unique_id(void *id128);
int cpx_comm_init_rank(const void *id128, int nranks, int rank, cpx_comm **out);
int cpx_comm_init_all(const int *devices, int ndev, cpx_comm **out);
int cpx_comm_info(const cpx_comm *c, int *nranks, int *nlocal, int *first_rank);
int cpx_comm_destroy(cpx_comm *c);
int cpx_comm_allgather_u8(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank,
                          void *const *streams);
int cpx_comm_allreduce_i64(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t count, int op,
                           void *const *streams);
int cpx_comm_allreduce_f64(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t count, int op,
                           void *const *streams);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* COMMPY_AMD_H */
