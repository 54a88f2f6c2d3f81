// Internal helpers shared by the HIP translation units of libcommpy_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "commpy_amd.h"

namespace cpx {

void set_error(const char *fmt, ...);
hipStream_t lib_stream();   // lazily created per-process stream of the current device
// name of the (dominant) kernel the last decoder call of this thread launched -- read back by cpx_last_kernel(), so
// that benchmarks and tests report what really ran instead of re-deriving the dispatch rules
void note_kernel(const char *fmt, ...);
const char *last_kernel_name();
// redo counter of "detect and redo" paths: `dev` = two device words {count, workgroups done}, zero between launches; `host` = a pinned
// host word.  The redo kernel adds its flagged items to dev[0] and calls redo_finish() at its end: the last workgroup publishes the
// count to *host and zeroes the device words.  note_redo() AFTER the final note_kernel() makes cpx_last_kernel append
// "redo: <host word> of <total> <what>" (pointers null: no counter available)
struct RedoCounter { unsigned *dev, *host; };
RedoCounter redo_counter(hipStream_t st);
#ifdef __HIPCC__
// collective over the workgroup (one __syncthreads); `mine` = this workgroup's contribution, added by thread 0
__device__ __forceinline__ void redo_finish(RedoCounter rc, unsigned mine) {
    __syncthreads();
    if (threadIdx.x == 0 && rc.dev) {
        if (mine) atomicAdd(&rc.dev[0], mine);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned prev = atomicAdd(&rc.dev[1], 1u);
        if (prev == gridDim.x - 1) {                              // the last workgroup of the launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const unsigned total = atomicExch(&rc.dev[0], 0u);
            atomicExch(&rc.dev[1], 0u);
            __hip_atomic_store(rc.host, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif
void note_redo(long long total, const char *what);
// CPX_EINVAL unless the current device is the one the handle's tables were created on
int check_handle_device(int handle_device, const char *what);
int ensure_device();        // CPX_OK if a HIP device is usable
int device_cus();           // compute units of the current device (256 on MI355X)
// workgroups of `fn` (block size `threads`, no dynamic LDS) resident on the whole device at once -- the size of a
// persistent grid; cached per kernel
int resident_blocks(const void *fn, int threads);
// Scratch arena keyed by (device, stream, slot): grown with hipMalloc on demand, reused by later calls and
// released by cpx_release_workspace().  Kernels of one stream serialise, so one arena per stream is safe.
// (Stream-ordered hipMallocAsync/hipFreeAsync was measured to hand out memory that is still in use on this
//  stack -- 45/120 corrupted LDPC decodes -- so the engine never uses it.)
int workspace(hipStream_t stream, int slot, size_t bytes, void **out, bool *fresh = nullptr);   // fresh: the block was (re)allocated by this call

#define CPX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            cpx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return CPX_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define CPX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            cpx::set_error(__VA_ARGS__);        \
            return (code);                      \
        }                                       \
    } while (0)

// codeword-per-lane Viterbi path (viterbi_cw.hip): true when it handled the call (*rc = status)
// nanflags ('soft' only, else null): [B] bytes, set to 1 for every codeword that received a NaN (viterbi.hip re-decodes those)
bool viterbi_codeword_path(const ::cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L, int64_t T,
                           int tb, int type, uint8_t *d_bits, uint8_t *nanflags, hipStream_t st, int *rc);

// LDS-resident LDPC path (ldpc_resident.hip): true when it handled the call (*rc = status)
int ldpc_resident_tables(::cpx_ldpc *c, const int32_t *row_ptr, const int32_t *row_pad, const int32_t *col_ptr,
                         const int32_t *col_pad_cj);
void ldpc_resident_free(::cpx_ldpc *c);
// nanflags (min-sum only, else null): [B] bytes, written for every block: 1 = a NaN among its LLRs (ldpc.hip decodes it again)
// block_major: d_dec / d_out are [B][n_v] (one block per row) instead of [n_v][B]
int ldpc_forced_path();   // cpx_ldpc_set_path / CPX_LDPC_PATH: 0 auto, 1 'tiled', 2 'resident', 3 'resident-log' (forced modes never substitute another kernel)
bool ldpc_resident_path(const ::cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec, double *d_out,
                        int block_major, int32_t *d_iters, int *d_clipped, uint8_t *nanflags, hipStream_t st, int *rc);

// absolute-scale BCJR / turbo redo path (bcjr_exact.hip): decodes the codewords whose flag byte is set, overwriting the outputs
bool bcjr_exact_supported(int S, int64_t N, int turbo);
int bcjr_exact_map(const ::cpx_trellis *t, const double *sys, const double *par, const double *Lin, int64_t B, int64_t N, double nv2,
                   int want_bits, double *Lout, uint8_t *bits, const uint8_t *flags, hipStream_t st);
int bcjr_exact_turbo(const ::cpx_trellis *t, const double *sys, const double *p1, const double *p2, const double *Lint_or_null,
                     const int32_t *perm, int64_t B, int64_t N, double nv2, int n_iter, uint8_t *bits, const uint8_t *flags,
                     hipStream_t st);

// CPX_LDPC_SPA=exact: sum-product check rows always by the exact-order sequence (ldpc_dev.h)
bool ldpc_spa_exact();

// per-device issue lock for entry points that take scratch-arena memory (runtime.hip)
struct IssueGuard {
    IssueGuard();
    ~IssueGuard();
    IssueGuard(const IssueGuard &) = delete;
    IssueGuard &operator=(const IssueGuard &) = delete;
    int dev;
};

// blocking download into pageable host memory through pinned staging + host threads (runtime.hip)
int d2h_pageable(void *dst, const void *d_src, size_t bytes, hipStream_t st);
void issue_lock(int dev, bool lock);     // cpx_release_workspace takes all of them

// roctx range for the lifetime of the object when CPX_TRACE=1 (runtime.hip); a no-op otherwise
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
    bool on;
};
bool trace_enabled();
#define CPX_TRACE_CAT2(a, b) a##b
#define CPX_TRACE_CAT(a, b) CPX_TRACE_CAT2(a, b)
#define CPX_TRACE(name) cpx::TraceRange CPX_TRACE_CAT(cpx_trace_, __LINE__)(name)

// precision mode (cpx_set_precision / CPX_PRECISION): false = fp64-parity (default), true = fp32-fast
bool precision_fast();

void viterbi_lean_ring(bool on);   // thread-local: the next 64-state rounds of this thread take the unmirrored ring where that flavour exists
void viterbi_prefer_cw(bool on);   // thread-local: the next dispatches of this thread take the codeword path whatever the batch size
int viterbi_path_flags();   // bit 0 wave only, bit 1 codeword path forced, bit 2 strict, bit 3 two-kernel form, bit 4 general kernel
// the general Viterbi kernel (viterbi_generic.hip): any trellis cpx_trellis_create accepts, any traceback depth
int viterbi_generic(const ::cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L, int64_t T, int tb, int type,
                    uint8_t *d_bits, hipStream_t st);

inline hipStream_t pick_stream(void *s) { return s ? reinterpret_cast<hipStream_t>(s) : lib_stream(); }

// RAII device buffer for the host-buffer entry points.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); p = nullptr; return CPX_ENOMEM; }
        return CPX_OK;
    }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// a block of the scratch arena (not owned: released by cpx_release_workspace)
struct ArenaBuf {
    void *p = nullptr;
    template <class T> T *as() { return static_cast<T *>(p); }
};

}  // namespace cpx

// ---- handles -----------------------------------------------------------------------------------
#define CPX_MAX_STATES 65536   // the specialised kernels stop at 128; viterbi_generic.hip / bcjr_exact.hip serve the rest
#define CPX_MAX_INPUTS 4
#define CPX_MAX_N 6

struct cpx_trellis {
    __attribute__((visibility("hidden"))) ~cpx_trellis() = default;   // (the handle types are declared in the public header, i.e. with
                                                                      //  default visibility: keep their implicit members out of the ABI)
    int k, n, S, I;
    int device;
    // host copies
    std::vector<int32_t> next_state, output;
    std::vector<int32_t> pred_state, pred_input, pred_code;  // [S][I] in np.where order
    // device tables (int32)
    int32_t *d_next = nullptr, *d_out = nullptr;             // [S][I]
    int32_t *d_pred_state = nullptr, *d_pred_input = nullptr, *d_pred_code = nullptr;  // [S][I]
    // the fused codeword-per-lane Viterbi kernels compiled for THIS code's generators (cpx_trellis_attach_viterbi_code, round 6):
    // [decoding type][run-time hop count]; null module: none attached (built-in pair or table-driven kernel)
    hipModule_t spec_mod = nullptr;
    hipFunction_t spec_fn[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    int spec_lg = 0;
    unsigned spec_g0 = 0, spec_g1 = 0;
};

struct cpx_ldpc {
    int n_v, n_c;
    int64_t n_edges;
    int device;
    int max_cdeg, max_vdeg;
    // check-major edge list (sorted by check, then variable)
    int32_t *d_edge_var = nullptr;    // [E] variable of edge e
    int32_t *d_row_ptr = nullptr;     // [n_c+1]
    // variable-major view: for each variable its edges in increasing check order
    int32_t *d_col_ptr = nullptr;     // [n_v+1]
    int32_t *d_col_edge = nullptr;    // [E] edge ids
    int32_t *d_col_cj = nullptr;      // [E] (check << 5) | position of the edge in its check's row (min-sum records)
    // padded copies (row/column stride cpad/vpad): their address depends on the node index only, so the
    // scalar loads of a work item issue together with row_ptr/col_ptr instead of after them
    int cpad = 0, vpad = 0;
    int32_t *d_row_pad = nullptr;     // [n_c][cpad] variable of the j-th edge of check c
    int32_t *d_col_pad_edge = nullptr;  // [n_v][vpad] edge id of the q-th edge of variable v
    int32_t *d_col_pad_cj = nullptr;    // [n_v][vpad] (check << 5) | position
    // LDS-resident path (ldpc_resident.hip): tables of pre-scaled LDS byte offsets, null when the state of one block
    // does not fit the LDS of a compute unit
    int32_t *d_res_row_deg = nullptr;   // [n_c] check degree
    int32_t *d_res_row_q = nullptr;     // [n_c][cpad] offset of Q[variable]; padding -> a +inf slot
    int32_t *d_res_col_r = nullptr;     // [n_v][vpad] offset of R[check][position], increasing check; padding -> a +0.0 slot
    int32_t *d_res_row_q32 = nullptr;   // the same two tables with 4-byte offsets (fp32-fast mode)
    int32_t *d_res_col_r32 = nullptr;
    int32_t *d_res_vgrp = nullptr;      // [ceil(n_v/64)] chunks of four entries per group of 64 variables
};

struct cpx_modem {
    int M, nbits;
    int device;
    double *d_const = nullptr;  // [M][2]
    // axis-separable square constellations (QAMModem): label = (a << nbits/2) | b, point = xs[a] + 1j*ys[b]
    bool separable = false;
    double *d_axes = nullptr;   // [2][sqrt(M)]: xs then ys
    // ... whose levels are equally spaced and labelled in reflected Gray order (QAMModem): level j = axes[0] + j * gp_step has
    // label j ^ (j >> 1) -- the soft demodulator then needs four exp per axis (demod.hip, GP)
    bool gp = false;
    double gp_step[2] = {0.0, 0.0};
};
