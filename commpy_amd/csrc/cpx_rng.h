// Counter-based random numbers of the device link-simulation stages, shared by linksim.hip (one stage per kernel) and demod.hip
// (link_front_kernel: the same stages fused into the soft demodulator).  Philox4x32-10 (Salmon et al., SC'11); the streams are
// defined by (seed, stream id, element index), so a fused kernel reproduces exactly what the staged kernels draw.
//   random messages    /root/reference/commpy/links.py:229 (np.random.choice((0,1), n)) -- Philox stream instead of MT19937
//   AWGN               /root/reference/commpy/channels.py:37-55 (noise = (randn + 1j*randn) * scale per component)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace cpx {

struct Philox {
    uint32_t c[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox{{c0, c1, c2, c3}};
}

// uniform in (0, 1] with 53 bits
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
    const uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return ((double)m + 1.0) * (1.0 / 9007199254740992.0);
}

// 16 message bits per counter value: bit j of the low word of philox(i, stream, seed) is message bit 16 i + j (random_bits_kernel)
__device__ __forceinline__ uint32_t message_bits16(uint64_t i, uint64_t stream, uint64_t seed) {
    return philox4x32_10(i, stream, seed).c[0] & 0xFFFFu;
}

// y = x + (scale_re n_re, scale_im n_im), n ~ N(0,1) i.i.d.: Box-Muller on the four words of philox(i, stream, seed) (awgn_kernel)
__device__ __forceinline__ double2 awgn_add(double2 v, uint64_t i, double scale_re, double scale_im, uint64_t seed, uint64_t stream) {
    const Philox r = philox4x32_10(i, stream, seed);
    const double u1 = u01(r.c[0], r.c[1]), u2 = u01(r.c[2], r.c[3]);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);                                   // (no reduction of a large argument: u2 is in (0, 1])
    v.x += scale_re * rad * cs;
    v.y += scale_im * rad * sn;
    return v;
}

}  // namespace cpx
