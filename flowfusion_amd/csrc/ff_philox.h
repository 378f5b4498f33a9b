// ff_philox.h -- counter-based normal deviates shared by the integrator kernels and ff_aux.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ff {

// Counter-based normal deviates for the Euler-Maruyama rows (include/flowfusion_amd.h, "In-kernel noise"):
// Philox4x32-10 (Salmon et al., SC'11) keyed by the caller's seed, counter = (global sample index lo/hi,
// noise index, dimension / 4); its four words make four normals by two Box-Muller pairs.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1)
{
    const float u1 = __builtin_fmaf((float)(a >> 8), 0x1p-24f, 0x1p-25f);        // (0, 1)
    const float u2 = (float)(b >> 8) * 0x1p-24f;                                   // [0, 1): a turn
    const float rad = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    z0 = rad * __builtin_amdgcn_cosf(u2);                                          // v_cos/v_sin take turns
    z1 = rad * __builtin_amdgcn_sinf(u2);
}

// index of the noise slab reserved for the PRIOR draw of a sample (ff_normal_fill / distributed.py): the
// Euler-Maruyama rows count their slabs from rng_noise_base upwards and never reach it
constexpr uint32_t kPriorNoiseIndex = 0xFFFFFFFFu;

} // namespace ff
