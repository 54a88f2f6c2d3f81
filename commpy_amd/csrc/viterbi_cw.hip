// Viterbi decoder, "codeword per lane" path for large batches of the standard rate-1/2 feed-forward codes.
// Same contract and decision rule as viterbi.hip (/root/reference/commpy/channelcoding/convcode.py:661-749,
// :590-657, :575-587) and the same float64 operations in the same order: bit-identical output.
//
// viterbi.hip maps one trellis STATE to a lane: every step pays a cross-lane exchange of the path metrics, an
// all-lane float64 minimum (first-argmin rule) and a ballot -- 26 VALU instructions per trellis step per codeword,
// VALU-issue bound.  When the batch gives every SIMD of the chip a wavefront of 64 codewords, a lane can own a
// whole CODEWORD instead:
//   * the S path metrics live in the lane's registers and are updated IN PLACE.  The radix-2 butterfly of a
//     shift-register trellis reads states (2j, 2j+1) and writes (j, j+S/2); leaving the results where the inputs
//     were rotates the logical->physical register map by one bit per step, so log2(S) unrolled steps return to the
//     identity: 2S VGPRs hold the metrics and the add-compare-select of all states is straight-line code with NO
//     cross-lane traffic (2 v_add_f64, v_cmp_lt_f64, v_min_f64 or 2 v_cndmask, v_addc per state);
//   * WHICH of the four branch metrics a branch uses depends on the generator polynomials; they are template
//     parameters (the branch code is a constexpr function), instantiated for the standard codes below and checked
//     against the trellis tables the caller built -- any other trellis takes the state-per-lane kernels;
//   * the first-argmin state of a step is an in-lane v_min_f64 tree + an in-lane first-equal scan;
//   * traceback, two forms:
//       - fused (viterbi_cw_fused_kernel, default traceback depth 5 m): the decision words stay in an LDS ring per
//         wave, the walk of step t-1 is pipelined through the arithmetic of step t, decoded bits leave through an
//         LDS tile -- one kernel, HBM traffic = LLRs in + bits out;
//       - two kernels (any depth <= 48): decision words and first-argmin states go to a workspace in HBM,
//         [group of 64 codewords][step][lane] so that every store is one coalesced line per wave (9 B per
//         codeword-step), and viterbi_cw_tb_kernel runs the sliding traceback: one workgroup per group stages a
//         window of 64 + tb - 2 steps in LDS (row stride 65 words: the lanes of a wave walk consecutive rows of one
//         column without bank conflicts) and every wave traces two codewords at a time, lane = output step.
// Measured on MI355X for BASELINE config 2 (B = 65536, K = 7, soft): see DESIGN.md 4.1.
#include "cpx_internal.h"
#include "cpx_math.h"
#include "viterbi_cw_asm.h"
#ifndef CPX_GEN_GB
#define CPX_GEN_GB 2      // (rounds 3 / 4: butterflies per LDS fetch group of the table-driven kernel; unused since the index-mode selection of round 5)
#endif

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

using namespace cpx;

namespace {

template <int LGS, unsigned G0, unsigned G1>
struct SrCode {
    static constexpr int S = 1 << LGS;
    static constexpr int parity(unsigned v) { return __builtin_popcount(v) & 1; }
    // 2-bit output (MSB = first generator) of the branch into state s from its j-th predecessor:
    // register = [input bit | predecessor state], convcode.py:166-175 (generator MSB taps the input)
    static constexpr int code(int s, int j) {
        const unsigned p = (unsigned)(((s << 1) & (S - 1)) | j), b = (unsigned)(s >> (LGS - 1));
        const unsigned reg = (b << LGS) | p;
        return (parity(reg & G0) << 1) | parity(reg & G1);
    }
};

template <int LGS>
constexpr int rotl(int s, int r) {
    r %= LGS;
    return r == 0 ? s : (((s << r) | (s >> (LGS - r))) & ((1 << LGS) - 1));
}

__device__ __forceinline__ double vmin(double a, double b) {      // one v_min_f64 (fmin adds canonicalising v_max)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One add-compare-select decision: d = (y < x); result = d ? y : x (first minimum wins, convcode.py:633-642);
// acc = 2*acc + d shifts the decision bit in.  Two forms:
//  * acs_min: v_cmp + v_addc + v_min_f64 -- equal to the select whenever neither operand is NaN, which holds for
//    'hard' (integer metrics) and 'soft' (the clip maps a NaN input to -500, metrics are >= 0 or +inf, only added);
//  * acs_select: v_cmp + 2 v_cndmask + v_addc, the select of viterbi.hip itself -- used for 'unquantized', where a NaN
//    input reaches the metrics and v_min_f64's NaN rule (return the other operand) would differ.
__device__ __forceinline__ double acs_min(unsigned &acc, double x, double y) {
    double r;
    asm("v_cmp_lt_f64 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\tv_min_f64 %0, %2, %3"
        : "=&v"(r), "+v"(acc) : "v"(x), "v"(y) : "vcc");
    return r;
}

__device__ __forceinline__ double acs_select(unsigned &acc, double x, double y) {
    int rlo, rhi;
    asm("v_cmp_lt_f64 vcc, %4, %3\n\tv_cndmask_b32 %0, %5, %7, vcc\n\tv_cndmask_b32 %1, %6, %8, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %2, vcc"
        : "=&v"(rlo), "=&v"(rhi), "+v"(acc)
        : "v"(x), "v"(y), "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(__double2loint(y)), "v"(__double2hiint(y))
        : "vcc");
    return __hiloint2double(rhi, rlo);
}

struct CwParams {
    const double *coded;          // [B][len]
    unsigned long long *dec;      // [groups][T][64]  decision word of step t of the lane's codeword
    unsigned char *best;          // [groups][T][64]  first-argmin state
    uint8_t *bits;                // [B][L]
    uint8_t *nanflags;            // 'soft': one byte per item of the redo launch (64 / S consecutive codewords; one codeword for 64 states),
                                  // 1 = an item's codeword received a NaN (viterbi.hip decodes the item again, NaN-exact)
    int64_t B, len, L, T, Lk, Tp;   // Tp: T rounded up to whole groups of log2(S) steps (row count of dec/best)
    int type, tb;
    unsigned goff[4];             // table-driven kernel (G0 = G1 = 0): per butterfly j the REGISTER OFFSET 2 c_j of the branch metric of its code
                                  // c_j (the 2-bit code of the branch 2j -> j, input 0), packed eight 4-bit fields per word (round 5: four scalar
                                  // registers; thirty-two separate ones spilled to VGPR lanes)
};

// 'soft': the reference's clip lets a NaN through and the codeword's metrics are NaN from that step on (convcode.py:719,
// :633-645).  The kernels here only DETECT that -- one unordered compare of the step's two received values, OR-ed into a
// lane mask on the scalar unit -- and go on with the NaN clipped to -500; flagged codewords are decoded again by the
// NaN-exact state-per-lane kernel (viterbi.hip, "NaN among 'soft' inputs").
__device__ __forceinline__ void nan_or(unsigned long long &mask, double r0, double r1) {
#ifdef CPX_AB_NO_NAN_DETECT                                         // A/B builds only (experiments/README.md): what the detection costs
    (void)mask; (void)r0; (void)r1;
    return;
#endif
    asm("v_cmp_u_f64 vcc, %1, %2\n\ts_or_b64 %0, %0, vcc" : "+s"(mask) : "v"(r0), "v"(r1) : "vcc", "scc");
}

// Per-bit metrics of one received value (convcode.py:575-587) -- identical to viterbi.hip
__device__ __forceinline__ void bit_metrics_cw(int type, double r, double &m0, double &m1) {
    if (type == CPX_VIT_HARD) {
        long long ri = (long long)r;
        m0 = (double)(ri ^ 0ll);
        m1 = (double)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        double nll0 = fast_log<false>(exp(r) + 1.0);   // r is clipped to +-500: the argument is finite and >= 1
        m0 = nll0;
        m1 = nll0 - r;
    } else {
        double d0 = r - (-1.0), d1 = r - 1.0;
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

// One traceback hop: the decision bit of state st is bit 63 - st of the step's word, i.e. the top bit of (w << st)
// (v_lshlrev_b64 uses the low 6 bits of st only); st' = (st << 1) | bit in one v_alignbit_b32.  The high bits of st are
// never masked on the way -- only st mod S is meaningful (for S < 64 the shift amount is masked explicitly).
template <int LGS>
__device__ __forceinline__ unsigned tb_hop(unsigned long long w, unsigned st) {
    const unsigned hi = (unsigned)((w << (st & ((1u << LGS) - 1u))) >> 32);       // LGS = 6: the mask is the hardware's own
    return __builtin_amdgcn_alignbit(st, hi, 31);
}

// One trellis step with the logical->physical map rotated by R: physical register rotl(s, R) holds state s before the
// step and rotl(s, R + 1) after it.
// The hot loop is kept free of branches (decoding type as a template parameter, unconditional loads and stores): with
// control flow between the prefetch and its use the compiler waits for vmcnt(0) right after issuing the loads and the
// memory latency of every group is exposed (measured: 3.26 ms instead of 1.75 ms for BASELINE config 2).
struct NoHook {
    template <int P> __device__ __forceinline__ void at() const {}
};

// `hook.at<P>()` is called at four fixed points of the step (after the branch metrics, after each half of the butterflies,
// after the minimum tree): the fused kernel slots the traceback of the previous step in there.
// G0 = G1 = 0: TABLE-DRIVEN codes -- any shift-register code whose two generators both tap the input bit and the oldest register bit
// (every rate-1/2 code of full constraint length).  The four branches of butterfly j then carry the codes c_j, c_j ^ 3, c_j ^ 3, c_j,
// so one 2-bit number per butterfly describes the code; it arrives packed in scalar registers (goff) and selects the branch metric by
// VGPR index mode (round 5, see the GEN branch below; rounds 3 / 4: an LDS table [4][64 lanes] of pairs (metric of c, metric of c ^ 3) and
// every butterfly reads its pair from there: 32 16-byte LDS reads + 32 address adds per step more than the compiled-in codes.
template <int LGS, unsigned G0, unsigned G1, int TYPE, int R, class Hook = NoHook>
__device__ __forceinline__ void cw_step(double (&pm)[1 << LGS], double r0, double r1, unsigned long long &word, int &best,
                                        const Hook &hook = Hook(), unsigned char *bml = nullptr, const unsigned *gidx = nullptr) {
    constexpr int type = TYPE;
    using C = SrCode<LGS, G0, G1>;
    constexpr bool GEN = G0 == 0 && G1 == 0;
    constexpr int S = 1 << LGS, H = S / 2;
    if (type == CPX_VIT_SOFT) {                                    // coded_bits.clip(-500, 500) (:719)
        r0 = fmin(fmax(r0, -500.0), 500.0);
        r1 = fmin(fmax(r1, -500.0), 500.0);
    }
    double m00, m01, m10, m11;
    bit_metrics_cw(type, r0, m00, m01);
    bit_metrics_cw(type, r1, m10, m11);
    double bmv[4];                                                 // NumPy add.reduce, n < 8: sequential from 0
    bmv[0] = (0.0 + m00) + m10; bmv[1] = (0.0 + m00) + m11;
    bmv[2] = (0.0 + m01) + m10; bmv[3] = (0.0 + m01) + m11;
    unsigned da = 0, db = 0;                                       // decisions of states 0..H-1 / H..S-1
    hook.template at<0>();
    auto butterfly = [&](int j, double m_a0, double m_a1, double m_b0, double m_b1) {
        const int x = rotl<LGS>(2 * j, R), y = rotl<LGS>(2 * j + 1, R);
        const double a = pm[x], b = pm[y];                         // metrics of the predecessors 2j, 2j+1
        const double a0 = a + m_a0, a1 = b + m_a1;                 // into state j      (:629)
        const double b0 = a + m_b0, b1 = b + m_b1;                 // into state j + S/2
        if (TYPE == CPX_VIT_UNQUANTIZED) {
            pm[x] = acs_select(da, a0, a1);                        // state j     now lives in register rotl(j, R+1) = x
            pm[y] = acs_select(db, b0, b1);                        // state j+S/2 now lives in register y
        } else {
            pm[x] = acs_min(da, a0, a1);
            pm[y] = acs_min(db, b0, b1);
        }
    };
    if constexpr (GEN) {
        // Round 5: the metric of "code c_j of butterfly j" is selected by VGPR INDEX MODE (s_set_gpr_idx_on, GFX9): the step's four
        // branch metrics sit in eight consecutive registers, the wave-uniform code arrives in a scalar register, and the four additions of
        // a butterfly read `v[base + 2 c]` / `v[base + 2 (c ^ 3)]` as their first operand -- the scalar unit, idle in this kernel, does the
        // selecting and the vector unit executes exactly the instructions of a compiled-in code.  (Rounds 3 / 4 parked the metrics as
        // (c, c ^ 3) pairs in an LDS table and read 16 bytes per butterfly: 32 LDS reads, their address adds and waits per step and
        // 64 - 98 values spilled to AGPRs: 1.94 ms on the config-2 geometry against 1.55 ms for compiled-in generators.)  The table is
        // pinned to v[248:255] because an index-mode operand must be named in the asm text; the mode is on for the four additions only
        // (it would redirect the first operand of EVERY vector instruction), and scripts/micro/gpr_idx_check.hip checks the addressing.
        typedef double cpx_d4 __attribute__((ext_vector_type(4)));
        // two tables: the metrics in code order and in REVERSED order -- code c ^ 3 = 3 - c sits at offset 2 c of the reversed one, so ONE
        // index serves all four additions of a butterfly.  With one wave per SIMD every instruction, scalar or not, costs an issue slot:
        // the first version (one table; s_bfe_u32 + on + s_xor_b32 + idx + off + three s_nop per butterfly) took 2.21 ms where the LDS
        // table took 1.94.  Now the mode is switched on once per step with index 0 -- harmless for every other vector instruction --
        // and a butterfly costs two scalar instructions: its index in, index 0 back.
        const cpx_d4 tab = {bmv[0], bmv[1], bmv[2], bmv[3]}, rtab = {bmv[3], bmv[2], bmv[1], bmv[0]};
        // (every statement that sets the index names m0 as clobbered -- s_set_gpr_idx_on / _idx write M0[7:0] -- and
        //  tests/test_isa_guards.py scans the built code object: between `on` and `off` nothing else may write or consume M0.)
        asm volatile("s_set_gpr_idx_on 0, 1" ::: "memory", "m0");
        auto acs_pair = [&](int j, double a0, double a1, double b0, double b1) {
            const int x = rotl<LGS>(2 * j, R), y = rotl<LGS>(2 * j + 1, R);
            if (TYPE == CPX_VIT_UNQUANTIZED) {
                pm[x] = acs_select(da, a0, a1);                    // state j     now lives in register rotl(j, R+1) = x
                pm[y] = acs_select(db, b0, b1);                    // state j+S/2 now lives in register y
            } else {
                pm[x] = acs_min(da, a0, a1);
                pm[y] = acs_min(db, b0, b1);
            }
        };
        if constexpr (H >= 4 && TYPE != CPX_VIT_UNQUANTIZED) {
            // two butterflies per statement, additions AND add-compare-selects (acs_min's three instructions, four times): three scalar
            // instructions per pair.  (Separate statements cost a wait state each: the compiler pads every boundary between two inline-asm
            // blocks with an s_nop -- 2.2 per butterfly when the selects were statements of their own.)
#pragma unroll
            for (int j = 0; j < H; j += 2) {
                if (j == H / 2) hook.template at<1>();          // (index 0 while it runs)
                const int x0 = rotl<LGS>(2 * j, R), y0 = rotl<LGS>(2 * j + 1, R), x1 = rotl<LGS>(2 * j + 2, R), y1 = rotl<LGS>(2 * j + 3, R);
                const double a = pm[x0], b = pm[y0], c = pm[x1], d = pm[y1];      // predecessors 2j, 2j+1 | 2j+2, 2j+3
                double a0, a1, b0, b1, c0, c1, d0, d1;             // (the four minima overwrite a0, b0, c0, d0)
                asm volatile("s_set_gpr_idx_idx %[i0]\n\t"
                             "v_add_f64 %[a0], v[248:249], %[a]\n\t"   // into state j:       predecessor 2j   + code c      (:629)
                             "v_add_f64 %[b1], v[248:249], %[b]\n\t"   // into state j + S/2: predecessor 2j+1 + code c
                             "v_add_f64 %[a1], v[240:241], %[b]\n\t"   // into state j:       predecessor 2j+1 + code c ^ 3
                             "v_add_f64 %[b0], v[240:241], %[a]\n\t"   // into state j + S/2: predecessor 2j   + code c ^ 3
                             "s_set_gpr_idx_idx %[i1]\n\t"
                             "v_add_f64 %[c0], v[248:249], %[c]\n\t"
                             "v_add_f64 %[d1], v[248:249], %[d]\n\t"
                             "v_add_f64 %[c1], v[240:241], %[d]\n\t"
                             "v_add_f64 %[d0], v[240:241], %[c]\n\t"
                             "s_set_gpr_idx_idx 0\n\t"
                             "v_cmp_lt_f64 vcc, %[a1], %[a0]\n\tv_addc_co_u32 %[da], vcc, %[da], %[da], vcc\n\tv_min_f64 %[a0], %[a0], %[a1]\n\t"
                             "v_cmp_lt_f64 vcc, %[b1], %[b0]\n\tv_addc_co_u32 %[db], vcc, %[db], %[db], vcc\n\tv_min_f64 %[b0], %[b0], %[b1]\n\t"
                             "v_cmp_lt_f64 vcc, %[c1], %[c0]\n\tv_addc_co_u32 %[da], vcc, %[da], %[da], vcc\n\tv_min_f64 %[c0], %[c0], %[c1]\n\t"
                             "v_cmp_lt_f64 vcc, %[d1], %[d0]\n\tv_addc_co_u32 %[db], vcc, %[db], %[db], vcc\n\tv_min_f64 %[d0], %[d0], %[d1]"
                             : [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1), [c0] "=&v"(c0), [c1] "=&v"(c1),
                               [d0] "=&v"(d0), [d1] "=&v"(d1), [da] "+v"(da), [db] "+v"(db)
                             : [i0] "s"(gidx[j]), [i1] "s"(gidx[j + 1]), [a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d),
                               "{v[248:255]}"(tab), "{v[240:247]}"(rtab)
                             : "vcc", "m0");
                pm[x0] = a0; pm[y0] = b0;                          // state j now lives in register rotl(j, R+1) = x0, state j+S/2 in y0
                pm[x1] = c0; pm[y1] = d0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < H; j++) {
                if (j == H / 2) hook.template at<1>();
                const double a = pm[rotl<LGS>(2 * j, R)], b = pm[rotl<LGS>(2 * j + 1, R)];
                double a0, a1, b0, b1;
                asm volatile("s_set_gpr_idx_idx %[ic]\n\t"
                             "v_add_f64 %[a0], v[248:249], %[a]\n\t"
                             "v_add_f64 %[b1], v[248:249], %[b]\n\t"
                             "v_add_f64 %[a1], v[240:241], %[b]\n\t"
                             "v_add_f64 %[b0], v[240:241], %[a]\n\t"
                             "s_set_gpr_idx_idx 0"
                             : [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1)
                             : [ic] "s"(gidx[j]), [a] "v"(a), [b] "v"(b), "{v[248:255]}"(tab), "{v[240:247]}"(rtab)
                             : "m0");
                acs_pair(j, a0, a1, b0, b1);
            }
        }
        asm volatile("s_set_gpr_idx_off" ::: "memory", "m0");
    } else {
#pragma unroll
        for (int j = 0; j < H; j++) {
            if (j == H / 2) hook.template at<1>();
            butterfly(j, bmv[C::code(j, 0)], bmv[C::code(j, 1)], bmv[C::code(j + H, 0)], bmv[C::code(j + H, 1)]);
        }
    }
    hook.template at<2>();
    // first-argmin state (:645): the minimum (v_min_f64 tree, as viterbi.hip's cross-lane tree) and the first state equal to it
    double m0 = pm[0], m1 = pm[1 % S], m2 = pm[2 % S], m3 = pm[3 % S];
    if constexpr (S == 64) {
        // 60 v_min_f64 as three statements of four interleaved chains (viterbi_cw_asm.h: one statement per instruction cost an
        // s_nop per dependent pair and serialised the chains)
#define CPX_V4(a) "v"(pm[a]), "v"(pm[(a) + 1]), "v"(pm[(a) + 2]), "v"(pm[(a) + 3])
        asm(CPX_MIN_BLOCK24 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
            : CPX_V4(4), CPX_V4(8), CPX_V4(12), CPX_V4(16), CPX_V4(20), CPX_V4(24));
        asm(CPX_MIN_BLOCK24 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
            : CPX_V4(28), CPX_V4(32), CPX_V4(36), CPX_V4(40), CPX_V4(44), CPX_V4(48));
        asm(CPX_MIN_BLOCK12 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : CPX_V4(52), CPX_V4(56), CPX_V4(60));
#undef CPX_V4
    } else {
#pragma unroll
        for (int s = 4; s < S; s += 4) {
            m0 = vmin(m0, pm[s]); m1 = vmin(m1, pm[s + 1]); m2 = vmin(m2, pm[s + 2]); m3 = vmin(m3, pm[s + 3]);
        }
    }
    const double mn = vmin(vmin(m0, m1), vmin(m2, m3));
    hook.template at<3>();
    int bst = 0;
    if constexpr (S == 64) {
        // first state equal to the minimum, scanned downwards (a later, lower state overwrites): per four states four compares
        // into four SGPR pairs, then the four selects with inline-constant state numbers -- three instructions always sit
        // between a float64 compare and the select that reads its mask (back to back the pair needs two wait states).  Three
        // statements of 24 / 24 / 16 states (30 operands is the limit of one).
        unsigned long long k0, k1, k2, k3;
#define CPX_S4(a) "v"(pm[rotl<LGS>((a), R + 1)]), "v"(pm[rotl<LGS>((a) - 1, R + 1)]), "v"(pm[rotl<LGS>((a) - 2, R + 1)]), "v"(pm[rotl<LGS>((a) - 3, R + 1)])
        asm(CPX_SCAN_BLOCK0 : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
            : "v"(mn), CPX_S4(63), CPX_S4(59), CPX_S4(55), CPX_S4(51), CPX_S4(47), CPX_S4(43));
        asm(CPX_SCAN_BLOCK1 : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
            : "v"(mn), CPX_S4(39), CPX_S4(35), CPX_S4(31), CPX_S4(27), CPX_S4(23), CPX_S4(19));
        asm(CPX_SCAN_BLOCK2 : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
            : "v"(mn), CPX_S4(15), CPX_S4(11), CPX_S4(7), CPX_S4(3));
#undef CPX_S4
    } else if constexpr (S % 4 == 0) {
#pragma unroll
        for (int s = S - 4; s >= 0; s -= 4) {
            unsigned long long k0, k1, k2, k3;
            asm("v_cmp_eq_f64 %1, %5, %9\n\tv_cmp_eq_f64 %2, %6, %9\n\tv_cmp_eq_f64 %3, %7, %9\n\tv_cmp_eq_f64 %4, %8, %9\n\t"
                "v_cndmask_b32 %0, %0, %10, %1\n\tv_cndmask_b32 %0, %0, %11, %2\n\tv_cndmask_b32 %0, %0, %12, %3\n\t"
                "v_cndmask_b32 %0, %0, %13, %4"
                : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
                : "v"(pm[rotl<LGS>(s + 3, R + 1)]), "v"(pm[rotl<LGS>(s + 2, R + 1)]), "v"(pm[rotl<LGS>(s + 1, R + 1)]),
                  "v"(pm[rotl<LGS>(s, R + 1)]), "v"(mn), "n"(s + 3), "n"(s + 2), "n"(s + 1), "n"(s));   // inline constants
        }
    } else {
#pragma unroll
        for (int s = S - 1; s >= 0; s--) bst = (pm[rotl<LGS>(s, R + 1)] == mn) ? s : bst;
    }
    // shifting the decisions in in increasing state order leaves state j of a half at bit H-1-j of its word; placing the
    // lower half on top puts state s at bit 63 - s of the 64-bit word: the traceback reads it as the top bit of (w << s)
    word = ((unsigned long long)da << (64 - H)) | ((unsigned long long)db << (64 - S));
    best = bst;
}

// ---- float32 path metrics: the "fp32-fast" mode (cpx_set_precision) -- NOT the parity mode ------------------
// 'soft' and 'unquantized' metrics in float32; the soft branch metric is the correlation form (the term common to all
// branches of a step is dropped: no exp, no log), and near-ties between path metrics may resolve differently (measured
// mismatch rate: DESIGN.md 4.1).  'hard' metrics of 0/1 inputs are Hamming distances <= 2 T, exact in float32
// for T < 2^22: identical bits.
// Same structure as cw_step: in-place butterflies, decision words, first-argmin state; the minimum tree uses v_min3_f32.
__device__ __forceinline__ float acs_min_f32(unsigned &acc, float x, float y) {
    float r;
    asm("v_cmp_lt_f32 vcc, %3, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\tv_min_f32 %0, %2, %3"
        : "=&v"(r), "+v"(acc) : "v"(x), "v"(y) : "vcc");
    return r;
}
__device__ __forceinline__ float vmin3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void bit_metrics_f32(int type, double r, float &m0, float &m1) {
    if (type == CPX_VIT_HARD) {
        const long long ri = (long long)r;
        m0 = (float)(ri ^ 0ll);
        m1 = (float)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        // the reference's metrics are m0 = log(e^r + 1), m1 = m0 - r: every branch of a step carries the same
        // m0(r0) + m0(r1), which cannot change a comparison between paths of equal length -- the fast mode drops it
        // (correlation metric, SURVEY Appendix C: 0 flipped bits of 263 680 against the reference) and with it the exp and
        // the log of every received value
        m0 = 0.0f;
        m1 = -(float)r;
    } else {
        const float x = (float)r, d0 = x + 1.0f, d1 = x - 1.0f;
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

template <int LGS, unsigned G0, unsigned G1, int TYPE, int R, class Hook = NoHook>
__device__ __forceinline__ void cw_step(float (&pm)[1 << LGS], double r0, double r1, unsigned long long &word, int &best,
                                        const Hook &hook = Hook(), unsigned char * = nullptr, const unsigned * = nullptr) {
    using C = SrCode<LGS, G0, G1>;
    constexpr int S = 1 << LGS, H = S / 2;
    if (TYPE == CPX_VIT_SOFT) {                                    // coded_bits.clip(-500, 500) (:719)
        r0 = fmin(fmax(r0, -500.0), 500.0);
        r1 = fmin(fmax(r1, -500.0), 500.0);
    }
    float m00, m01, m10, m11;
    bit_metrics_f32(TYPE, r0, m00, m01);
    bit_metrics_f32(TYPE, r1, m10, m11);
    float bmv[4];
    bmv[0] = (0.0f + m00) + m10; bmv[1] = (0.0f + m00) + m11;
    bmv[2] = (0.0f + m01) + m10; bmv[3] = (0.0f + m01) + m11;
    unsigned da = 0, db = 0;
    hook.template at<0>();
#pragma unroll
    for (int j = 0; j < H; j++) {
        if (j == H / 2) hook.template at<1>();
        const int x = rotl<LGS>(2 * j, R), y = rotl<LGS>(2 * j + 1, R);
        const float a = pm[x], b = pm[y];
        const float a0 = a + bmv[C::code(j, 0)], a1 = b + bmv[C::code(j, 1)];
        const float b0 = a + bmv[C::code(j + H, 0)], b1 = b + bmv[C::code(j + H, 1)];
        pm[x] = acs_min_f32(da, a0, a1);
        pm[y] = acs_min_f32(db, b0, b1);
    }
    hook.template at<2>();
    float mn;
    if constexpr (S >= 8 && S % 8 == 0) {
        float m0 = pm[0], m1 = pm[1], m2 = pm[2], m3 = pm[3];
#pragma unroll
        for (int s = 4; s + 8 <= S; s += 8) {
            m0 = vmin3_f32(m0, pm[s], pm[s + 1]); m1 = vmin3_f32(m1, pm[s + 2], pm[s + 3]);
            m2 = vmin3_f32(m2, pm[s + 4], pm[s + 5]); m3 = vmin3_f32(m3, pm[s + 6], pm[s + 7]);
        }
        m0 = vmin3_f32(m0, pm[S - 4], pm[S - 3]);
        m1 = vmin3_f32(m1, pm[S - 2], pm[S - 1]);
        mn = fminf(vmin3_f32(m0, m1, m2), m3);
    } else {
        mn = pm[0];
#pragma unroll
        for (int s = 1; s < S; s++) mn = fminf(mn, pm[s]);
    }
    hook.template at<3>();
    int bst = 0;
    if constexpr (S % 4 == 0) {
#pragma unroll
        for (int s = S - 4; s >= 0; s -= 4) {
            unsigned long long k0, k1, k2, k3;
            asm("v_cmp_eq_f32 %1, %5, %9\n\tv_cmp_eq_f32 %2, %6, %9\n\tv_cmp_eq_f32 %3, %7, %9\n\tv_cmp_eq_f32 %4, %8, %9\n\t"
                "v_cndmask_b32 %0, %0, %10, %1\n\tv_cndmask_b32 %0, %0, %11, %2\n\tv_cndmask_b32 %0, %0, %12, %3\n\t"
                "v_cndmask_b32 %0, %0, %13, %4"
                : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
                : "v"(pm[rotl<LGS>(s + 3, R + 1)]), "v"(pm[rotl<LGS>(s + 2, R + 1)]), "v"(pm[rotl<LGS>(s + 1, R + 1)]),
                  "v"(pm[rotl<LGS>(s, R + 1)]), "v"(mn), "n"(s + 3), "n"(s + 2), "n"(s + 1), "n"(s));
        }
    } else {
#pragma unroll
        for (int s = S - 1; s >= 0; s--) bst = (pm[rotl<LGS>(s, R + 1)] == mn) ? s : bst;
    }
    word = ((unsigned long long)da << (64 - H)) | ((unsigned long long)db << (64 - S));
    best = bst;
}

// Workgroups of four wavefronts (one group of 64 codewords each): the four waves of a workgroup land on the four SIMDs
// of one CU, so 256 workgroups put exactly one wave on every SIMD of the chip.  (With single-wave workgroups the
// dispatcher packed two waves per SIMD on half of the CUs for B = 65536: 2.98 ms instead of 1.75 ms.)
constexpr int ACS_WAVES = 4;

template <int LGS, unsigned G0, unsigned G1, int TYPE>
__global__ __launch_bounds__(64 * ACS_WAVES) void viterbi_cw_acs_kernel(CwParams p) {
    constexpr int S = 1 << LGS;
    const int lane = threadIdx.x & 63;
    const int64_t grp = (int64_t)blockIdx.x * ACS_WAVES + (threadIdx.x >> 6);
    const int64_t cw = grp * 64 + lane;
    if (grp * 64 >= p.B) return;                                                   // whole wave beyond the batch
    const bool valid = cw < p.B;
    const double *x = p.coded + (valid ? cw : 0) * p.len;
    unsigned long long *dec = p.dec + grp * p.Tp * 64 + lane;                      // rows of steps T+1..Tp: never read
    unsigned char *best = p.best + grp * p.Tp * 64 + lane;

    double pm[S];
#pragma unroll
    for (int s = 0; s < S; s++) pm[s] = (s == 0) ? 0.0 : __builtin_huge_val();   // path_metrics[:,0] = inf, [0][0] = 0 (:705-706)

    const double pad = (TYPE == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;                // t > L//k -> padding (:722-734)
    // step indices are 32-bit (the dispatcher guarantees T < 2^31): the scalar unit has no signed 64-bit compare, and
    // 64-bit step arithmetic put ~20 VALU instructions per step of purely wave-uniform work into the hot loop
    const int T = (int)p.T, tmax = (int)((p.Lk < p.T) ? p.Lk : p.T);              // tmax: last step with received values (>= 1)
    auto load = [&](int t) {                                                      // always a valid address; padded below
        const int tc = (t < tmax) ? t : tmax;
        return *reinterpret_cast<const double2 *>(x + (int64_t)(tc - 1) * 2);
    };
    double2 cur[LGS], nxt[LGS];
#pragma unroll
    for (int u = 0; u < LGS; u++) cur[u] = load(1 + u);
    unsigned long long nanmask = 0;                                               // 'soft': lanes that received a NaN
    // groups of LGS steps; a partial last group simply runs on (steps > T see padding, their rows are never read)
    for (int t = 1; t <= T; t += LGS) {
#pragma unroll
        for (int u = 0; u < LGS; u++) nxt[u] = load(t + LGS + u);                // prefetch: lands during this group
#pragma unroll
        for (int u = 0; u < LGS; u++) {
            const bool have = t + u <= tmax;
            cur[u].x = have ? cur[u].x : pad;
            cur[u].y = have ? cur[u].y : pad;
        }
        unsigned long long *d = dec + (int64_t)(t - 1) * 64;
        unsigned char *b = best + (int64_t)(t - 1) * 64;
        auto one = [&](auto rtag, const double2 &v) {
            constexpr int R = decltype(rtag)::value;
            unsigned long long word;
            int bst;
            if constexpr (TYPE == CPX_VIT_SOFT) nan_or(nanmask, v.x, v.y);
            cw_step<LGS, G0, G1, TYPE, R>(pm, v.x, v.y, word, bst);
            d[R * 64] = word;
            b[R * 64] = (unsigned char)bst;
        };
        if constexpr (LGS >= 1) one(std::integral_constant<int, 0>{}, cur[0]);
        if constexpr (LGS >= 2) one(std::integral_constant<int, 1>{}, cur[1 % LGS]);
        if constexpr (LGS >= 3) one(std::integral_constant<int, 2>{}, cur[2 % LGS]);
        if constexpr (LGS >= 4) one(std::integral_constant<int, 3>{}, cur[3 % LGS]);
        if constexpr (LGS >= 5) one(std::integral_constant<int, 4>{}, cur[4 % LGS]);
        if constexpr (LGS >= 6) one(std::integral_constant<int, 5>{}, cur[5 % LGS]);
#pragma unroll
        for (int u = 0; u < LGS; u++) cur[u] = nxt[u];
    }
    // One flag per ITEM of the redo kernel = 64 / S consecutive codewords (viterbi_dispatch reads them that way); the two-kernel form
    // is only instantiated for 64 states, where an item is one codeword and this per-codeword store IS the per-item one.
    static_assert(LGS == 6, "viterbi_cw_acs_kernel writes per-codeword NaN flags: valid for 64-state trellises only (items of one codeword)");
    if constexpr (TYPE == CPX_VIT_SOFT)
        if (valid) p.nanflags[cw] = (uint8_t)((nanmask >> lane) & 1ull);
}

// ---- fused variant: add-compare-select AND sliding traceback in one kernel, no workspace in HBM -------------------------
// For the default traceback depth of the K = 7 code (tb_depth = 5 m = 30, H = tb - 2 = 28 hops) the decision words never
// leave the CU: every wave keeps the last 32 words of its 64 codewords in an LDS ring [slot][lane] (conflict-free, each
// lane reads only its own column) and, right after the add-compare-select of step t, walks the H hops back from the
// step's first-argmin state -- 2 VALU instructions per hop (tb_hop), LDS addresses that do not depend on the states, the
// whole walk straight-line code that the scheduler interleaves with the next step's arithmetic.  The ring is stored
// twice, 32 slots apart, so that the H + 1 words of a walk sit at constant offsets below one base address.  The decoded
// bit of step t - H goes to a staging tile in LDS and the tile is written out every 96 steps, one coalesced 64-byte
// store per codeword and half-tile.  HBM traffic is the algorithmic one again (the two-kernel path moves 9 B per
// codeword-step through a workspace and back).
constexpr int FR_RING = 32, FR_CHUNK = 96, FR_OBPAD = 100;     // ring slots (mirrored), steps per flush (a multiple of LGS), tile row bytes
// Traceback depths above the default (tb_depth 31 .. 48 for K = 7) run the same kernel on a ring of FR_RING_DEEP slots that is
// NOT mirrored -- the same LDS per wave (64 slots once instead of 32 twice) --: a walk's words then wrap around the ring and
// every hop forms its own LDS address, ((q - h) mod 64) * 512 + lane * 8: one more vector add per hop (+6 % per step with 46
// hops).  Round 2 sent these depths to the two-kernel form (1.66 x the time: decision words through HBM and back).
constexpr int FR_RING_DEEP = 64;

// slots of one wave: the ring (twice if mirrored) + one dummy slot per copy for steps > T
template <int RING, bool MIR> constexpr int fused_slots() { return MIR ? 2 * RING + 2 : RING + 1; }
template <int RING, bool MIR, bool GEN = false>
constexpr size_t fused_wave_lds() {
    // ring + dummy slot(s), staging tile
    return (size_t)fused_slots<RING, MIR>() * 64 * 8 + (size_t)64 * FR_OBPAD;   // (GEN: rounds 3 / 4 kept an LDS table of 4 KB here)
}

// The traceback walk of one step, cut into four batches of hops that cw_step's hook runs between the phases of the NEXT
// step: the LDS reads of a batch are issued one phase before their hops, so their latency hides behind the
// add-compare-select arithmetic (left to itself the compiler emits the walk as 14 read -> wait -> 2-hop rounds at the end
// of the step: +0.5 ms).  sched_barrier pins the batches where they are put.
// RT: the number of hops is a run-time value hr <= H (traceback depths below the default): hops h >= hr leave the state alone
// (one v_cndmask each, +4 % per step); their LDS reads still go to valid, older ring slots.
template <int LGS, int H, bool RT = false, int RING = FR_RING, bool MIR = true>
struct WalkHook {
    static constexpr int NB = 4, PER = (H + NB - 1) / NB;
    // MIR: ring pointer of the walked step t': word of step t' - h at pw[(RING - h) * 64].  Not mirrored: pw = this lane's
    // column, q = ring slot of step t' (wave-uniform), word of step t' - h at pw[((q - h) mod RING) * 64].
    const unsigned long long *pw;
    int q;
    mutable unsigned st;                    // state of the walk
    int hr;                                 // hops to execute (RT only)
    mutable unsigned long long buf[PER];
    __device__ __forceinline__ unsigned long long word(int h) const {
        return MIR ? pw[(RING - h) * 64] : pw[((q - h) & (RING - 1)) * 64];
    }
    template <int B> __device__ __forceinline__ void issue() const {
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (B * PER + i < H) buf[i] = word(B * PER + i);
    }
    template <int B> __device__ __forceinline__ void hops() const {
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (B * PER + i < H) {
                const unsigned nx = tb_hop<LGS>(buf[i], st);
                st = (!RT || B * PER + i < hr) ? nx : st;
            }
    }
    template <int P> __device__ __forceinline__ void at() const {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) {
            issue<0>();
        } else {
            hops<P - 1>();
            issue<P>();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void finish() const {
        __builtin_amdgcn_sched_barrier(0);
        hops<NB - 1>();
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int LGS, unsigned G0, unsigned G1, int TYPE, int HT, bool RT = false, class F = double, int RING = FR_RING, bool MIR = true>
__global__ __launch_bounds__(64 * ACS_WAVES) void viterbi_cw_fused_kernel(CwParams p) {
    constexpr int S = 1 << LGS, FR_GROUPS = FR_CHUNK / LGS, CHUNK = FR_GROUPS * LGS;
    static_assert(HT >= 0 && HT + 1 <= RING && (RING & (RING - 1)) == 0 && CHUNK + 1 <= FR_OBPAD, "ring / tile too small");
    const int H = RT ? p.tb - 2 : HT;                                              // hops of a walk = tb_depth - 2
    constexpr bool GEN = G0 == 0 && G1 == 0;                                        // table-driven code (cw_step)
#ifdef CPX_VIT_SPEC_LG
    // per-pair code object (see the end of the file): launched through hipModuleLaunchKernel, whose functions have no
    // "dynamic LDS above 64 KiB" attribute to raise -- the ring and the tiles are a static array of the same size instead
    __shared__ __attribute__((aligned(16))) unsigned char smem[ACS_WAVES * fused_wave_lds<RING, MIR, GEN>()];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t grp = (int64_t)blockIdx.x * ACS_WAVES + wv;
    const int64_t cw = grp * 64 + lane;
    if (grp * 64 >= p.B) return;                                                   // whole wave beyond the batch
    if constexpr (LGS == 6 && !MIR && !GEN && RING == FR_RING) __builtin_amdgcn_s_setprio(3);   // launch_fused_lean: another kernel's waves share the SIMD
    const bool valid = cw < p.B;
    const double *x = p.coded + (valid ? cw : 0) * p.len;
    static_assert(!GEN || (!MIR && std::is_same<F, double>::value), "table-driven codes: unmirrored ring, float64");
    unsigned long long *ring = reinterpret_cast<unsigned long long *>(smem + (size_t)wv * fused_wave_lds<RING, MIR, GEN>());
    unsigned char *obuf = reinterpret_cast<unsigned char *>(ring + fused_slots<RING, MIR>() * 64);
    unsigned long long *mycol = ring + lane;                                       // slot s of this lane's codeword: mycol[s * 64]
    unsigned char *myrow = obuf + lane * FR_OBPAD;

    F pm[S];
#pragma unroll
    for (int s = 0; s < S; s++) pm[s] = (s == 0) ? (F)0.0 : (F)__builtin_huge_val();   // path_metrics[:,0] = inf, [0][0] = 0 (:705-706)
    // table-driven codes: the register offset 2 c_j of every butterfly, unpacked once into scalar registers (cw_step, GEN)
    unsigned gidx[GEN ? S / 2 : 1];
#pragma unroll
    for (int j = 0; j < (GEN ? S / 2 : 1); j++) gidx[j] = GEN ? __builtin_amdgcn_ubfe(p.goff[j >> 3], 4 * (j & 7), 4) : 0u;

    const double pad = (TYPE == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;                // t > L//k -> padding (:722-734)
    const int T = (int)p.T, tmax = (int)((p.Lk < p.T) ? p.Lk : p.T);              // 32-bit step indices, see the ACS kernel
    auto load = [&](int t) {                                                      // always a valid address; padded at use
        const int tc = (t < tmax) ? t : tmax;
        return *reinterpret_cast<const double2 *>(x + (int64_t)(tc - 1) * 2);
    };
    double2 cur[LGS];
#pragma unroll
    for (int u = 0; u < LGS; u++) cur[u] = load(1 + u);
    int best_T = 0;                                                               // first-argmin state of step T
    unsigned long long nanmask = 0;                                               // 'soft': lanes that received a NaN
    WalkHook<LGS, HT, RT, RING, MIR> walk;                                             // walk of the previous step (step 0: a dummy)
    walk.pw = mycol;
    walk.q = 0;
    walk.st = 0;
    walk.hr = H;

    // writes the decoded bits staged in tile entries 0 .. n-1: entry li holds the result of the walk of step tc0 + li - 1,
    // i.e. output step tc0 + li - 1 - H
    auto flush = [&](int tc0, int n) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // the tile rows are complete (same wave wrote them)
        const int64_t cwb = grp * 64;
        const int ncw = (int)((p.B - cwb < 64) ? (p.B - cwb) : 64);
        // eight codewords per round: their LDS reads are all in flight before the first store waits for one (one codeword
        // per round put an LDS round trip in front of every store: ~19 k cycles per 96 steps, 5 % of the kernel)
        for (int c0 = 0; c0 < 64; c0 += 8) {
            unsigned char v[8][2];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int li = lane + 64 * h;
                    v[u][h] = (li < n) ? obuf[(c0 + u) * FR_OBPAD + li] : (unsigned char)0;
                }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int li = lane + 64 * h, so = tc0 + li - 1 - H;
                    if (c0 + u < ncw && li < n && so >= 1 && so <= T - H && so - 1 < p.L)
                        p.bits[(cwb + c0 + u) * p.L + so - 1] = v[u][h];
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // tile read before the next chunk overwrites it
    };

    for (int tc0 = 1; tc0 <= T; tc0 += CHUNK) {
        const int left = (T - tc0) / LGS + 1;                                      // groups that start at a step <= T
        const int ngroups = left < FR_GROUPS ? left : FR_GROUPS;
        for (int g = 0; g < ngroups; g++) {
            const int t = tc0 + g * LGS;
            // prefetch of the NEXT group, all of it at the top of this one: a whole group (~4700 instructions) lies between
            // a load and the vmcnt(0) the compiler puts at the loop head.  (Reloading cur[R] inside step R saved 12
            // registers but left the youngest load only one step of lead: rocprofv3 showed 14.5 % of the wave cycles
            // parked in s_waitcnt.)
            double2 nxt[LGS];
#pragma unroll
            for (int u = 0; u < LGS; u++) nxt[u] = load(t + LGS + u);
            auto one = [&](auto rtag) {
                constexpr int R = decltype(rtag)::value;
                const int tt = t + R;
                const bool have = tt <= tmax;
                const double r0 = have ? cur[R].x : pad, r1 = have ? cur[R].y : pad;
                unsigned long long word;
                int bst;
                if constexpr (TYPE == CPX_VIT_SOFT) nan_or(nanmask, cur[R].x, cur[R].y);   // (steps > tmax re-read step tmax)
                cw_step<LGS, G0, G1, TYPE, R>(pm, r0, r1, word, bst, walk, nullptr, gidx);   // + hops 0 .. 3/4 H of the walk of step tt - 1
                walk.finish();
                myrow[g * LGS + R] = (unsigned char)((walk.st >> (LGS - 1)) & 1u);   // input bit of the branch into the state at step tt - 1 - H
                // ring slot of step tt and (mirrored ring) its copy RING slots above; the (at most LGS - 1) steps > T of the last group
                // write to two dummy slots instead: the ring must keep the words of steps T-H+1 .. T for the final walk
                const bool live = tt <= T;
                const int q = tt & (RING - 1);
                unsigned long long *wb = mycol + q * 64;
                if constexpr (MIR) {
                    unsigned long long *w0 = live ? wb : mycol + (2 * RING) * 64;
                    unsigned long long *w1 = live ? wb + RING * 64 : mycol + (2 * RING + 1) * 64;
                    *w0 = word;
                    *w1 = word;
                    walk.pw = wb;                                                 // next: the walk of this step
                } else {
                    unsigned long long *w0 = live ? wb : mycol + RING * 64;
                    *w0 = word;
                    walk.q = q;
                }
                best_T = (tt == T) ? bst : best_T;
                walk.st = (unsigned)bst;
            };
            if constexpr (LGS >= 1) one(std::integral_constant<int, 0>{});
            if constexpr (LGS >= 2) one(std::integral_constant<int, 1 % LGS>{});
            if constexpr (LGS >= 3) one(std::integral_constant<int, 2 % LGS>{});
            if constexpr (LGS >= 4) one(std::integral_constant<int, 3 % LGS>{});
            if constexpr (LGS >= 5) one(std::integral_constant<int, 4 % LGS>{});
            if constexpr (LGS >= 6) one(std::integral_constant<int, 5 % LGS>{});
#pragma unroll
            for (int u = 0; u < LGS; u++) cur[u] = nxt[u];
        }
        int n = ngroups * LGS;
        if (tc0 + CHUNK > T) {                                                     // last chunk: finish the pending walk (of its last step)
            unsigned st = walk.st;
            for (int h = 0; h < H; h++) st = tb_hop<LGS>(walk.word(h), st);
            myrow[n] = (unsigned char)((st >> (LGS - 1)) & 1u);
            n++;
        }
        flush(tc0, n);
    }
    if constexpr (TYPE == CPX_VIT_SOFT)
    {   // one flag per ITEM of the state-per-lane redo kernel (viterbi.hip): 64 / S consecutive codewords (one codeword for S = 64)
            constexpr int CPI = 64 / S;
            const int64_t item = grp * (64 / CPI) + lane;
            if (lane < 64 / CPI && item * CPI < p.B)
                p.nanflags[item] = (uint8_t)(((nanmask >> (lane * CPI)) & (CPI == 64 ? ~0ull : ((1ull << CPI) - 1ull))) != 0);
        }
    // the last H output steps: one walk from best[T]; the state before hop h is the state of step T - h
    if (valid) {
        const int qT = T & (RING - 1);
        unsigned st = (unsigned)best_T;
        for (int h = 0; h < H; h++) {
            const int so = T - h;
            if (so < 1) break;
            if (so - 1 < p.L) p.bits[cw * p.L + so - 1] = (uint8_t)((st >> (LGS - 1)) & 1u);
            st = tb_hop<LGS>(mycol[((qT - h) & (RING - 1)) * 64], st);
        }
    }
}

// Sliding traceback: bit of output step s = input bit of the state at step s on the path traced back from
// best[min(s + tb - 2, T)] (the rule of viterbi.hip).  One workgroup per group of 64 codewords; per pass of 64 output
// steps the rows of steps base .. base + 63 + tb - 2 are staged in LDS; wave w traces codewords w, w+8, ... two at a
// time (two independent dependent-load chains), lane i = output step base + i.
constexpr int TB_THREADS = 512, TB_WAVES = TB_THREADS / 64, TB_STRIDE = 65;

template <int LGS>
__global__ __launch_bounds__(TB_THREADS) void viterbi_cw_tb_kernel(CwParams p) {
    (void)LGS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.tb - 2, rows = 64 + H;
    unsigned long long *win = reinterpret_cast<unsigned long long *>(smem);        // [rows][65]
    unsigned char *bwin = reinterpret_cast<unsigned char *>(win + (size_t)rows * TB_STRIDE);   // [rows][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t grp = blockIdx.x;
    const unsigned long long *dec = p.dec + grp * p.Tp * 64;
    const unsigned char *best = p.best + grp * p.Tp * 64;
    for (int64_t base = 1; base <= p.T; base += 64) {
        int64_t need = base + 63 + H;
        if (need > p.T) need = p.T;
        __syncthreads();                                                          // readers of the previous window are done
        for (int64_t tt = base + w; tt <= need; tt += TB_WAVES) {                 // row of step tt -> window row tt - base
            win[(tt - base) * TB_STRIDE + lane] = dec[(tt - 1) * 64 + lane];
            bwin[(tt - base) * 64 + lane] = best[(tt - 1) * 64 + lane];
        }
        __syncthreads();
        const int64_t so = base + lane;                                           // output step of this lane
        const bool full = base + 63 + H <= p.T;                                   // every lane of the pass walks all H hops
        int64_t t0 = so + H;
        if (t0 > p.T) t0 = p.T;
        for (int c0 = w; c0 < 64; c0 += 2 * TB_WAVES) {
            const int c1 = c0 + TB_WAVES;
            unsigned st0 = 0, st1 = 0;
            if (so <= p.T) {
                st0 = bwin[(t0 - base) * 64 + c0];
                st1 = bwin[(t0 - base) * 64 + c1];
            }
            // hop h uses the decision word of step so + H - h (clipped lanes wait until that step is <= T)
            int idx = (lane + H) * TB_STRIDE;                                     // window row of step so + H
            if (full) {
                // the addresses do not depend on the states: unrolled, the reads of four hops are in flight together
#pragma unroll 4
                for (int h = 0; h < H; h++) {
                    st0 = tb_hop<LGS>(win[idx + c0], st0);
                    st1 = tb_hop<LGS>(win[idx + c1], st1);
                    idx -= TB_STRIDE;
                }
            } else {
                for (int h = 0; h < H; h++) {
                    const bool go = so + H - h <= p.T;                            // rows past `need` hold stale words: skipped
                    const unsigned n0 = tb_hop<LGS>(win[idx + c0], st0), n1 = tb_hop<LGS>(win[idx + c1], st1);
                    st0 = go ? n0 : st0;
                    st1 = go ? n1 : st1;
                    idx -= TB_STRIDE;
                }
            }
            const int64_t pos = so - 1;
            if (so <= p.T && pos < p.L) {
                const int64_t cwa = grp * 64 + c0, cwb = grp * 64 + c1;
                if (cwa < p.B) p.bits[cwa * p.L + pos] = (uint8_t)((st0 >> (LGS - 1)) & 1u);   // input bit of the branch into st
                if (cwb < p.B) p.bits[cwb * p.L + pos] = (uint8_t)((st1 >> (LGS - 1)) & 1u);
            }
        }
    }
}

template <int LGS, unsigned G0, unsigned G1>
bool tables_match(const cpx_trellis *t) {
    using C = SrCode<LGS, G0, G1>;
    if (t->S != (1 << LGS) || t->I != 2 || t->k != 1 || t->n != 2) return false;
    for (int s = 0; s < t->S; s++)
        for (int j = 0; j < 2; j++) {
            if (t->pred_state[s * 2 + j] != (((s << 1) & (t->S - 1)) | j)) return false;
            if (t->pred_input[s * 2 + j] != (s >> (LGS - 1))) return false;
            if (t->pred_code[s * 2 + j] != C::code(s, j)) return false;
        }
    return true;
}

// Table-driven codes (cw_step, G0 = G1 = 0): a 2^LGS-state shift-register trellis of rate 1/2 whose butterflies have the form
// (c, c ^ 3, c ^ 3, c) -- both generators tap the input and the oldest register bit.  Fills goff: field j = 2 c_j.
template <int LGS>
bool generic_match(const cpx_trellis *t, unsigned (&goff)[4]) {
    static_assert((1 << LGS) / 2 <= 32, "goff holds 32 butterflies");
    const int S = 1 << LGS, H = S / 2;
    for (unsigned &g : goff) g = 0u;
    if (t->S != S || t->I != 2 || t->k != 1 || t->n != 2) return false;
    for (int s = 0; s < S; s++)
        for (int j = 0; j < 2; j++) {
            if (t->pred_state[s * 2 + j] != (((s << 1) & (S - 1)) | j)) return false;
            if (t->pred_input[s * 2 + j] != (s >> (LGS - 1))) return false;
        }
    for (int j = 0; j < H; j++) {
        const int c = t->pred_code[j * 2 + 0];
        if (c < 0 || c > 3) return false;
        if (t->pred_code[j * 2 + 1] != (c ^ 3) || t->pred_code[(j + H) * 2 + 0] != (c ^ 3) || t->pred_code[(j + H) * 2 + 1] != c) return false;
        goff[j >> 3] |= (2u * (unsigned)c) << (4 * (j & 7));
    }

    return true;
}

template <int LGS, int TYPE, bool RT>
int launch_fused_generic_typed(const CwParams &p, hipStream_t st) {
    constexpr int RING = 5 * LGS - 1 <= 16 ? 16 : 32;              // a ring cut to the depth (see launch_fused_small_typed)
    auto *fn = viterbi_cw_fused_kernel<LGS, 0u, 0u, TYPE, 5 * LGS - 2, RT, double, RING, false>;
    const size_t lds = ACS_WAVES * fused_wave_lds<RING, false, true>();
    static bool raised[64] = {};
    static std::mutex raised_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(raised_mu);
    if (!raised[dev]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        raised[dev] = true;
    }
    const unsigned groups = (unsigned)((p.B + 63) / 64), blocks = (groups + ACS_WAVES - 1) / ACS_WAVES;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * ACS_WAVES), lds, st, p);
    return 1;
}

// the default depth runs the compile-time walk; smaller depths the same kernel with a run-time hop count (round 4)
template <int LGS>
int launch_fused_generic(const CwParams &p, hipStream_t st) {
    const bool rt = p.tb != 5 * LGS;
    if (p.type == CPX_VIT_HARD) return rt ? launch_fused_generic_typed<LGS, CPX_VIT_HARD, true>(p, st) : launch_fused_generic_typed<LGS, CPX_VIT_HARD, false>(p, st);
    if (p.type == CPX_VIT_SOFT) return rt ? launch_fused_generic_typed<LGS, CPX_VIT_SOFT, true>(p, st) : launch_fused_generic_typed<LGS, CPX_VIT_SOFT, false>(p, st);
    return rt ? launch_fused_generic_typed<LGS, CPX_VIT_UNQUANTIZED, true>(p, st) : launch_fused_generic_typed<LGS, CPX_VIT_UNQUANTIZED, false>(p, st);
}

template <int LGS, unsigned G0, unsigned G1>
void launch(const CwParams &p, size_t tb_lds, hipStream_t st) {
    const unsigned groups = (unsigned)((p.B + 63) / 64), ablocks = (groups + ACS_WAVES - 1) / ACS_WAVES;
    const dim3 ab(64 * ACS_WAVES);
    if (p.type == CPX_VIT_HARD) hipLaunchKernelGGL((viterbi_cw_acs_kernel<LGS, G0, G1, CPX_VIT_HARD>), dim3(ablocks), ab, 0, st, p);
    else if (p.type == CPX_VIT_SOFT) hipLaunchKernelGGL((viterbi_cw_acs_kernel<LGS, G0, G1, CPX_VIT_SOFT>), dim3(ablocks), ab, 0, st, p);
    else hipLaunchKernelGGL((viterbi_cw_acs_kernel<LGS, G0, G1, CPX_VIT_UNQUANTIZED>), dim3(ablocks), ab, 0, st, p);
    hipLaunchKernelGGL((viterbi_cw_tb_kernel<LGS>), dim3(groups), dim3(TB_THREADS), tb_lds, st, p);
}

// fused kernel: the walk is compiled for the reference's default traceback depth, tb_depth = 5 * total_memory
// (convcode.py:701); every smaller depth runs it with a run-time hop count
template <int LGS> constexpr int fused_tb() { return 5 * LGS; }

// ... and up to this depth on the deep ring (run-time hop count, float64 metrics only)
template <int LGS> constexpr int fused_tb_deep() { return 8 * LGS; }

template <int LGS, unsigned G0, unsigned G1, int TYPE, bool RT, class F, bool DEEP = false>
int launch_fused_typed(const CwParams &p, hipStream_t st) {
    constexpr int RING = DEEP ? FR_RING_DEEP : FR_RING;
    auto *fn = viterbi_cw_fused_kernel<LGS, G0, G1, TYPE, (DEEP ? fused_tb_deep<LGS>() : fused_tb<LGS>()) - 2, RT, F, RING, !DEEP>;
    const size_t lds = ACS_WAVES * fused_wave_lds<RING, !DEEP>();
    static bool raised[64] = {};                                 // > 64 KiB of dynamic LDS is opt-in, once per kernel and device
    static std::mutex raised_mu;                                 // host threads may launch concurrently (ctypes drops the GIL)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(raised_mu);
    if (!raised[dev]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();                             // a device with less LDS: the two-kernel form is used
            return 0;
        }
        raised[dev] = true;
    }
    const unsigned groups = (unsigned)((p.B + 63) / 64), blocks = (groups + ACS_WAVES - 1) / ACS_WAVES;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * ACS_WAVES), lds, st, p);
    return 1;
}

// 64 states on the 32-slot ring stored ONCE ('soft', default depth, float64): 93 KB of LDS per workgroup instead of 157, one more
// instruction per traceback hop.  Used for the round(s) of a batch that ALSO has a remainder on the state-per-lane kernels
// (viterbi_dispatch, round 6): with the mirrored ring a round's workgroup holds the CU's whole LDS and the remainder's workgroups are
// not placed until it retires (profiles/r06_viterbi_remainder_overlap_ab.txt); with this one they run beside it.
template <int LGS, unsigned G0, unsigned G1>
int launch_fused_lean(const CwParams &p, hipStream_t st) {
    auto *fn = viterbi_cw_fused_kernel<LGS, G0, G1, CPX_VIT_SOFT, fused_tb<LGS>() - 2, false, double, FR_RING, false>;
    const size_t lds = ACS_WAVES * fused_wave_lds<FR_RING, false>();
    static bool raised[64] = {};
    static std::mutex raised_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(raised_mu);
    if (!raised[dev]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        raised[dev] = true;
    }
    const unsigned groups = (unsigned)((p.B + 63) / 64), blocks = (groups + ACS_WAVES - 1) / ACS_WAVES;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * ACS_WAVES), lds, st, p);
    return 1;
}

// Small trellises (4 and 16 states) at their default depth: the same fused kernel on a 16- or 32-slot unmirrored ring -- 14.6 /
// 22.8 KB of LDS per wave instead of 39, so that two workgroups fit a CU (the one-wave-per-SIMD structure of the 64-state kernel
// leaves a 4-state step, ~60 instructions, waiting for its own latencies: 0.77 ms on BASELINE config 1 where the state-per-lane
// kernel takes 0.56; with the small ring 0.42 ms).
template <int LGS, unsigned G0, unsigned G1, int TYPE, bool RT>
int launch_fused_small_typed(const CwParams &p, hipStream_t st) {
    constexpr int RING = fused_tb<LGS>() - 1 <= 16 ? 16 : 32;
    auto *fn = viterbi_cw_fused_kernel<LGS, G0, G1, TYPE, fused_tb<LGS>() - 2, RT, double, RING, false>;
    const size_t lds = ACS_WAVES * fused_wave_lds<RING, false>();
    if (lds > 64 * 1024) {                                        // (32-slot ring: 91 KB) dynamic LDS above 64 KiB is opt-in, once per kernel and device
        static bool raised[64] = {};
        static std::mutex raised_mu;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
        std::lock_guard<std::mutex> lk(raised_mu);
        if (!raised[dev]) {
            if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            raised[dev] = true;
        }
    }
    const unsigned groups = (unsigned)((p.B + 63) / 64), blocks = (groups + ACS_WAVES - 1) / ACS_WAVES;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * ACS_WAVES), lds, st, p);
    return 1;
}
template <int LGS, unsigned G0, unsigned G1>
int launch_fused_small(const CwParams &p, hipStream_t st) {
    const bool rt = p.tb != fused_tb<LGS>();                       // below the default depth: run-time hop count (round 4)
    if (p.type == CPX_VIT_HARD) return rt ? launch_fused_small_typed<LGS, G0, G1, CPX_VIT_HARD, true>(p, st) : launch_fused_small_typed<LGS, G0, G1, CPX_VIT_HARD, false>(p, st);
    if (p.type == CPX_VIT_SOFT) return rt ? launch_fused_small_typed<LGS, G0, G1, CPX_VIT_SOFT, true>(p, st) : launch_fused_small_typed<LGS, G0, G1, CPX_VIT_SOFT, false>(p, st);
    return rt ? launch_fused_small_typed<LGS, G0, G1, CPX_VIT_UNQUANTIZED, true>(p, st) : launch_fused_small_typed<LGS, G0, G1, CPX_VIT_UNQUANTIZED, false>(p, st);
}

// the default depth runs the compile-time walk; smaller depths the same kernel with a run-time hop count
template <int LGS, unsigned G0, unsigned G1, class F>
int launch_fused(const CwParams &p, hipStream_t st) {
    if (p.tb > fused_tb<LGS>()) {
        if (p.type == CPX_VIT_HARD) return launch_fused_typed<LGS, G0, G1, CPX_VIT_HARD, true, double, true>(p, st);
        if (p.type == CPX_VIT_SOFT) return launch_fused_typed<LGS, G0, G1, CPX_VIT_SOFT, true, double, true>(p, st);
        return launch_fused_typed<LGS, G0, G1, CPX_VIT_UNQUANTIZED, true, double, true>(p, st);
    }
    const bool rt = p.tb != fused_tb<LGS>();
    if (p.type == CPX_VIT_HARD) return rt ? launch_fused_typed<LGS, G0, G1, CPX_VIT_HARD, true, F>(p, st) : launch_fused_typed<LGS, G0, G1, CPX_VIT_HARD, false, F>(p, st);
    if (p.type == CPX_VIT_SOFT) return rt ? launch_fused_typed<LGS, G0, G1, CPX_VIT_SOFT, true, F>(p, st) : launch_fused_typed<LGS, G0, G1, CPX_VIT_SOFT, false, F>(p, st);
    return rt ? launch_fused_typed<LGS, G0, G1, CPX_VIT_UNQUANTIZED, true, F>(p, st) : launch_fused_typed<LGS, G0, G1, CPX_VIT_UNQUANTIZED, false, F>(p, st);
}

}  // namespace

#ifdef CPX_VIT_SPEC_LG
// ---- a code object for ONE generator pair (round 6) ------------------------------------------------------------------------------
// Any rate-1/2 code of full constraint length runs the table-driven flavour above (branch metric selected by VGPR index mode: 1.78 ms
// on the config-2 geometry where a compiled-in pair takes 1.55).  Compiled with
//     hipcc --offload-arch=gfx950 --cuda-device-only -DCPX_VIT_SPEC_LG=6 -DCPX_VIT_SPEC_G0=<g0> -DCPX_VIT_SPEC_G1=<g1> -c viterbi_cw.hip
// this file yields a code object that holds nothing but the six fused kernels of THAT pair (three decoding types x {default depth,
// run-time hop count}), generators as template arguments like the built-in pairs; commpy_amd/jit.py builds and caches it,
// cpx_trellis_attach_viterbi_code loads it (hipModuleLoadData) and viterbi_codeword_path launches it in place of the table-driven
// kernel.  Generators in the template's convention ("MSB taps the input", see CPX_TRY below).  64 states: the mirrored 32-slot ring of
// launch_fused_typed; fewer: the small unmirrored ring of launch_fused_small_typed.
namespace {
constexpr int SPEC_LG = CPX_VIT_SPEC_LG;
constexpr int SPEC_RING = SPEC_LG == 6 ? FR_RING : (5 * SPEC_LG - 1 <= 16 ? 16 : 32);
constexpr bool SPEC_MIR = SPEC_LG == 6;
#define CPX_SPEC(TYPE, RT)                                                                                                  \
    template __global__ void viterbi_cw_fused_kernel<SPEC_LG, CPX_VIT_SPEC_G0, CPX_VIT_SPEC_G1, TYPE, 5 * SPEC_LG - 2, RT, double, \
                                                     SPEC_RING, SPEC_MIR>(CwParams);
CPX_SPEC(CPX_VIT_HARD, false) CPX_SPEC(CPX_VIT_HARD, true) CPX_SPEC(CPX_VIT_SOFT, false) CPX_SPEC(CPX_VIT_SOFT, true)
CPX_SPEC(CPX_VIT_UNQUANTIZED, false) CPX_SPEC(CPX_VIT_UNQUANTIZED, true)
#undef CPX_SPEC
}  // namespace
#else

namespace cpx {

// Kernel-path override (tests, benchmarks): bit 0 = "wave" (state-per-lane kernels only), bit 1 = forced codeword path,
// bit 2 = strict ("!": fail instead of falling back), bit 3 = two-kernel form even where the fused kernel applies,
// bit 4 = "general" (the slow-but-complete kernel of viterbi_generic.hip whatever the trellis).
// Initialised once from the environment variable CPX_VITERBI_PATH, changed at run time through cpx_viterbi_set_path().
static std::atomic<int> g_vit_path{-1};

static int parse_path(const char *e) {
    if (!e || !e[0]) return 0;
    if (e[0] == 'w') return 1;
    if (e[0] == 'g') return 16;                      // "general": viterbi_generic.hip for every trellis
    if (e[0] != 'c') return 0;
    return 2 | (strchr(e, '!') ? 4 : 0) | (strchr(e, '2') ? 8 : 0);
}

int viterbi_path_flags() {
    int v = g_vit_path.load(std::memory_order_relaxed);
    if (v < 0) {
        static std::once_flag once;
        std::call_once(once, [] { g_vit_path.store(parse_path(getenv("CPX_VITERBI_PATH")), std::memory_order_relaxed); });
        v = g_vit_path.load(std::memory_order_relaxed);
    }
    return v;
}

// Set by the host-buffer pipeline (viterbi.hip) around its per-chunk calls: a chunk of a large batch takes the codeword path
// whatever its own size (a round costs the same however full it is and hides behind the next chunk's upload; the fused
// kernel -- and with it the precision mode -- then serves the host API exactly as it serves the device API).
static thread_local bool tl_prefer_cw = false;
void viterbi_prefer_cw(bool on) { tl_prefer_cw = on; }
// Set by viterbi_dispatch around the round(s) of a batch whose remainder it runs beside them: the 64-state built-in pairs then take
// the unmirrored ring (launch_fused_lean) where that flavour exists ('soft', default depth, float64).
static thread_local bool tl_lean_ring = false;
void viterbi_lean_ring(bool on) { tl_lean_ring = on; }

static const char *type_name(int type) { return type == CPX_VIT_HARD ? "hard" : type == CPX_VIT_SOFT ? "soft" : "unquantized"; }

// Returns true when the call was handled here (*rc = status); false -> the caller uses the state-per-lane kernels.
bool viterbi_codeword_path(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L, int64_t T,
                           int tb, int type, uint8_t *d_bits, uint8_t *nanflags, hipStream_t st, int *rc) {
    *rc = CPX_OK;
    // path override (cpx_viterbi_set_path / CPX_VITERBI_PATH): "wave" = state-per-lane kernels; "cw" = this path whatever
    // the batch size; "cw!" = fail instead of falling back; a '2' anywhere ("cw2", "cw2!") = the two-kernel form even
    // where the fused kernel applies
    const int pf = viterbi_path_flags();
    if (pf & 1) return false;
    const bool forced = (pf & 2) || tl_prefer_cw, strict = pf & 4, two_kernels = pf & 8;
    auto reject = [&](const char *why) {
        if (!strict) return false;
        set_error("viterbi (codeword path): %s", why);
        *rc = CPX_ELIMIT;
        return true;
    };
    // one wavefront of 64 codewords per SIMD needs 65536 codewords on 256 CUs; below 45 % of that the wave kernels win
    // (their time is proportional to the batch, a round of this path costs the same however full it is: break-even 0.44)
    if (!forced && 20 * B < 9 * (int64_t)device_cus() * 4 * 64) return false;
    if (t->I != 2 || t->k != 1 || t->n != 2 || T < 1) return reject("needs a rate-1/2, k = 1 trellis");
    if ((len & 1) || ((uintptr_t)d_coded & 15)) return reject("rows must be 16-byte aligned");
    const size_t tb_lds = (size_t)(64 + tb - 2) * (TB_STRIDE * 8 + 64);
    if (tb_lds > 64 * 1024) return reject("traceback window exceeds 64 KiB of LDS");
    const int64_t groups = (B + 63) / 64;
    if (groups >= (1ll << 31) || T >= (1ll << 31) - 64) return reject("batch too large / block too long");
    CwParams p;
    p.coded = d_coded; p.bits = d_bits; p.B = B; p.len = len; p.L = L; p.T = T; p.Lk = L;   // k = 1
    p.type = type; p.tb = tb; p.nanflags = nanflags;
    if (type == CPX_VIT_SOFT && !nanflags) return reject("'soft' needs the NaN flag array");
    // float32 path metrics when the caller selected "fp32-fast" (cpx_set_precision) -- fused kernel only.  ('hard' metrics
    // of 0/1 inputs are Hamming distances <= 2 T, exact in float32: there the fast mode returns identical bits.)
    const bool f32 = precision_fast() && T < (1ll << 22);
#define CPX_TRY(LG, GA, GB)                                                                                         \
    if (tables_match<LG, GA, GB>(t)) {                                                                              \
        if (tl_lean_ring && type == CPX_VIT_SOFT && tb == fused_tb<LG>() && !two_kernels && !f32 && launch_fused_lean<LG, GA, GB>(p, st)) { \
            if (hipGetLastError() != hipSuccess) { set_error("viterbi (fused codeword path): launch failed"); *rc = CPX_EHIP; } \
            note_kernel("viterbi_cw_fused_kernel<%d,0%o,0%o,%s,%d,ring stored once>", LG, GA, GB, type_name(type), fused_tb<LG>() - 2); \
            return true;                                                                                            \
        }                                                                                                           \
        if (tb >= 2 && tb <= fused_tb_deep<LG>() && !two_kernels &&                                                 \
            (f32 ? launch_fused<LG, GA, GB, float>(p, st) : launch_fused<LG, GA, GB, double>(p, st))) {                 \
            const bool deep = tb > fused_tb<LG>();                                                                  \
            if (hipGetLastError() != hipSuccess) { set_error("viterbi (fused codeword path): launch failed"); *rc = CPX_EHIP; } \
            note_kernel("viterbi_cw_fused_kernel<%d,0%o,0%o,%s,%d%s%s>", LG, GA, GB, type_name(type),                    \
                        (deep ? fused_tb_deep<LG>() : fused_tb<LG>()) - 2,                                            \
                        deep ? ",runtime hops,64-slot ring" : tb == fused_tb<LG>() ? "" : ",runtime hops",              \
                        (f32 && !deep) ? ",f32" : "");                                                              \
            return true;                                                                                            \
        }                                                                                                           \
        void *w0 = nullptr, *w1 = nullptr;                                                                          \
        p.Tp = (T + LG - 1) / LG * LG;                                                                              \
        if ((*rc = workspace(st, 0, sizeof(unsigned long long) * (size_t)(groups * p.Tp * 64), &w0))) return true;  \
        if ((*rc = workspace(st, 1, (size_t)(groups * p.Tp * 64), &w1))) return true;                               \
        p.dec = static_cast<unsigned long long *>(w0);                                                              \
        p.best = static_cast<unsigned char *>(w1);                                                                  \
        launch<LG, GA, GB>(p, tb_lds, st);                                                                          \
        if (hipGetLastError() != hipSuccess) { set_error("viterbi (codeword path): launch failed"); *rc = CPX_EHIP; } \
        note_kernel("viterbi_cw_acs_kernel<%d,0%o,0%o,%s> + viterbi_cw_tb_kernel<%d>", LG, GA, GB, type_name(type), LG); \
        return true;                                                                                                \
    }
    // Template generators are in "MSB taps the input" order.  commpy's default polynomial_format='MSB' makes the
    // LEAST significant bit of the octal number the D^0 tap (convcode.py:211-222), i.e. the bit-reversed number here:
    // (133,171) -> (155,117).
    CPX_TRY(6, 0155u, 0117u)      // K = 7 (133,171), commpy default format: 802.11 / BASELINE configs 2 and 5
    CPX_TRY(6, 0117u, 0155u)      // K = 7 (171,133)
    CPX_TRY(6, 0133u, 0171u)      // K = 7 (133,171) written with polynomial_format='LSB' (convcode.py:216-222)
    CPX_TRY(6, 0171u, 0133u)      // K = 7 (171,133), 'LSB'
    // Wifi80211 as shipped: its generators are written in DECIMAL, (133, 171), and dec2bitarray wraps them to (5, 43)
    // (wifi80211.py:49, utilities.py:81-85) -- 0000101 / 0101011, bit-reversed 0120 / 0152.  The reference's
    // own link simulation (BASELINE config 5 with default arguments) decodes this 64-state code.
    CPX_TRY(6, 0120u, 0152u)
#undef CPX_TRY
    // K = 3 (5,7) -- BASELINE config 1 -- and K = 5 (23,35) up to their default depths: the small-ring flavour of the fused kernel
#define CPX_TRY_SMALL(LG, GA, GB)                                                                                       \
    if (tb >= 2 && tb <= fused_tb<LG>() && !two_kernels && !f32 && tables_match<LG, GA, GB>(t) && launch_fused_small<LG, GA, GB>(p, st)) { \
        if (hipGetLastError() != hipSuccess) { set_error("viterbi (small fused codeword path): launch failed"); *rc = CPX_EHIP; } \
        note_kernel("viterbi_cw_fused_kernel<%d,0%o,0%o,%s,%d,small ring%s>", LG, GA, GB, type_name(type), fused_tb<LG>() - 2,   \
                    tb == fused_tb<LG>() ? "" : ",runtime hops");                                                       \
        return true;                                                                                                    \
    }
    CPX_TRY_SMALL(2, 05u, 07u)
    CPX_TRY_SMALL(4, 031u, 027u)
#undef CPX_TRY_SMALL
    // a code object compiled for this code's generators (cpx_trellis_attach_viterbi_code): the built-in pairs' kernel, from a module
    if (t->spec_mod && tb >= 2 && tb <= 5 * t->spec_lg && !two_kernels && !f32) {
        const bool rt = tb != 5 * t->spec_lg;
        hipFunction_t fn = t->spec_fn[type][rt ? 1 : 0];
        const unsigned groups32 = (unsigned)groups, blocks = (groups32 + ACS_WAVES - 1) / ACS_WAVES;
        void *args[] = {(void *)&p};
        if (hipModuleLaunchKernel(fn, blocks, 1, 1, 64 * ACS_WAVES, 1, 1, 0, st, args, nullptr) != hipSuccess) {
            (void)hipGetLastError();
            set_error("viterbi (per-pair code object): launch failed");
            *rc = CPX_EHIP;
            return true;
        }
        note_kernel("viterbi_cw_fused_kernel<%d,0%o,0%o,%s,%d%s> (code object of this pair)", t->spec_lg, t->spec_g0, t->spec_g1,
                    type_name(type), 5 * t->spec_lg - 2, rt ? ",runtime hops" : "");
        return true;
    }
    // any other rate-1/2 shift-register code of full constraint length (4 .. 64 states), up to its default traceback depth: the
    // table-driven fused kernel (deeper windows, and the fp32-fast mode, go to the state-per-lane kernels)
#define CPX_TRY_TABLE(LG)                                                                                               \
    if (tb >= 2 && tb <= fused_tb<LG>() && !two_kernels && !f32 && generic_match<LG>(t, p.goff) && launch_fused_generic<LG>(p, st)) { \
        if (hipGetLastError() != hipSuccess) { set_error("viterbi (table-driven codeword path): launch failed"); *rc = CPX_EHIP; } \
        note_kernel("viterbi_cw_fused_kernel<%d,table-driven,%s,%d%s>", LG, type_name(type), fused_tb<LG>() - 2,              \
                    tb == fused_tb<LG>() ? "" : ",runtime hops");                                                       \
        return true;                                                                                                    \
    }
    CPX_TRY_TABLE(6) CPX_TRY_TABLE(5) CPX_TRY_TABLE(4) CPX_TRY_TABLE(3) CPX_TRY_TABLE(2)
#undef CPX_TRY_TABLE
    return reject("no instantiation for this trellis");
}

}  // namespace cpx

extern "C" int cpx_viterbi_set_path(const char *mode) {
    if (mode && mode[0] && mode[0] != 'w' && mode[0] != 'c' && mode[0] != 'g' && strcmp(mode, "auto") != 0) {
        cpx::set_error("cpx_viterbi_set_path: unknown mode '%s' (auto | wave | cw | cw! | cw2 | cw2! | general)", mode);
        return CPX_EINVAL;
    }
    cpx::g_vit_path.store((mode && strcmp(mode, "auto") != 0) ? cpx::parse_path(mode) : 0, std::memory_order_relaxed);
    return CPX_OK;
}

// ---- per-pair code objects (round 6): query and attach -------------------------------------------------------------------------
namespace {
// generators of a rate-1/2, k = 1 shift-register trellis in the kernel template's convention (bit lg = the input tap), or false
bool runtime_generators(const cpx_trellis *t, int &lg, unsigned &g0, unsigned &g1) {
    lg = 0;
    while ((1 << lg) < t->S) lg++;
    if (t->I != 2 || t->k != 1 || t->n != 2 || (1 << lg) != t->S || lg < 2 || lg > 6) return false;
    const int S = t->S;
    auto code_of = [&](unsigned reg) -> int {                    // reg = [input bit | predecessor state]
        const int pred = (int)(reg & (unsigned)(S - 1)), b = (int)(reg >> lg);
        const int s = (pred >> 1) | (b << (lg - 1)), j = pred & 1;
        return t->pred_code[s * 2 + j];
    };
    g0 = g1 = 0;
    for (int i = 0; i <= lg; i++) {                               // a linear code: the output of a unit register is the generators' bit i
        const int c = code_of(1u << i);
        if (c < 0 || c > 3) return false;
        g0 |= (unsigned)((c >> 1) & 1) << i;
        g1 |= (unsigned)(c & 1) << i;
    }
    for (int s = 0; s < S; s++)
        for (int j = 0; j < 2; j++) {
            if (t->pred_state[s * 2 + j] != (((s << 1) & (S - 1)) | j) || t->pred_input[s * 2 + j] != (s >> (lg - 1))) return false;
            const unsigned reg = ((unsigned)(s >> (lg - 1)) << lg) | (unsigned)(((s << 1) & (S - 1)) | j);
            const int want = ((__builtin_popcount(reg & g0) & 1) << 1) | (__builtin_popcount(reg & g1) & 1);
            if (t->pred_code[s * 2 + j] != want) return false;
        }
    return true;
}
bool builtin_pair(const cpx_trellis *t) {
    return tables_match<6, 0155u, 0117u>(t) || tables_match<6, 0117u, 0155u>(t) || tables_match<6, 0133u, 0171u>(t) ||
           tables_match<6, 0171u, 0133u>(t) || tables_match<6, 0120u, 0152u>(t) || tables_match<2, 05u, 07u>(t) ||
           tables_match<4, 031u, 027u>(t);
}
}  // namespace

extern "C" int cpx_trellis_viterbi_spec_query(const cpx_trellis *t, int *lg, unsigned *g0, unsigned *g1) {
    CPX_REQUIRE(t && lg && g0 && g1, CPX_EINVAL, "cpx_trellis_viterbi_spec_query: null pointer");
    *lg = 0; *g0 = 0; *g1 = 0;
    int l = 0;
    unsigned a = 0, b = 0;
    unsigned goff[4];
    // what the table-driven kernel serves today and a compiled pair would serve faster: full-constraint-length codes that are not built in
    const bool table = (t->S == 64 && generic_match<6>(t, goff)) || (t->S == 32 && generic_match<5>(t, goff)) ||
                       (t->S == 16 && generic_match<4>(t, goff)) || (t->S == 8 && generic_match<3>(t, goff)) ||
                       (t->S == 4 && generic_match<2>(t, goff));
    if (!table || builtin_pair(t) || t->spec_mod || !runtime_generators(t, l, a, b)) return CPX_OK;
    *lg = l; *g0 = a; *g1 = b;
    return CPX_OK;
}

extern "C" int cpx_trellis_attach_viterbi_code(cpx_trellis *t, const void *image, size_t bytes) {
    CPX_REQUIRE(t && image && bytes >= 64, CPX_EINVAL, "cpx_trellis_attach_viterbi_code: null / empty image");
    if (int rcd = cpx::check_handle_device(t->device, "attach_viterbi_code")) return rcd;
    CPX_REQUIRE(!t->spec_mod, CPX_EINVAL, "cpx_trellis_attach_viterbi_code: the trellis already has a code object");
    int lg = 0;
    unsigned g0 = 0, g1 = 0;
    CPX_REQUIRE(runtime_generators(t, lg, g0, g1), CPX_EINVAL,
                "cpx_trellis_attach_viterbi_code: not a rate-1/2 shift-register code of 4 .. 64 states");
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, image) != hipSuccess) {
        (void)hipGetLastError();
        cpx::set_error("cpx_trellis_attach_viterbi_code: hipModuleLoadData refused the image");
        return CPX_EHIP;
    }
    // the kernels are looked up BY THE GENERATORS OF THIS TRELLIS: an image compiled for another pair has no such symbols and is refused
    const int ring = lg == 6 ? FR_RING : (5 * lg - 1 <= 16 ? 16 : 32);
    hipFunction_t fn[3][2];
    for (int type = 0; type < 3; type++)
        for (int rt = 0; rt < 2; rt++) {
            char name[200];
            snprintf(name, sizeof(name), "_ZN12_GLOBAL__N_123viterbi_cw_fused_kernelILi%dELj%uELj%uELi%dELi%dELb%dEdLi%dELb%dEEEvNS_8CwParamsE",
                     lg, g0, g1, type, 5 * lg - 2, rt, ring, lg == 6 ? 1 : 0);
            if (hipModuleGetFunction(&fn[type][rt], mod, name) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipModuleUnload(mod);
                cpx::set_error("cpx_trellis_attach_viterbi_code: the image has no kernel %s (compiled for other generators, or from other sources)", name);
                return CPX_EINVAL;
            }
        }
    memcpy(t->spec_fn, fn, sizeof(fn));
    t->spec_lg = lg; t->spec_g0 = g0; t->spec_g1 = g1;
    t->spec_mod = mod;                                            // last: the dispatcher tests this field (attach while another thread
    return CPX_OK;                                                //  decodes with the same handle is still the caller's to serialise)
}

extern "C" int cpx_trellis_has_viterbi_code(const cpx_trellis *t) { return (t && t->spec_mod) ? 1 : 0; }

extern "C" int cpx_trellis_detach_viterbi_code(cpx_trellis *t) {
    CPX_REQUIRE(t, CPX_EINVAL, "cpx_trellis_detach_viterbi_code: null trellis");
    if (t->spec_mod) {
        (void)hipDeviceSynchronize();                             // nothing may still run from the module
        (void)hipModuleUnload(t->spec_mod);
        t->spec_mod = nullptr;
        t->spec_lg = 0;
    }
    return CPX_OK;
}
#endif  // CPX_VIT_SPEC_LG
