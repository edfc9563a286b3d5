// Philox4x32-10 counter-based generator (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator
// torch.cuda uses as well) for the path's random draws: the reparameterisation noise eps ~ N(0, 1) of z_dist.rsample()
// (imagevae/mnist_vae.py:79, measurevae/measure_vae.py:116) and the keep-masks of nn.Dropout / nn.GRU's dropout
// (imagevae/mnist_vae.py:16-47, measurevae/encoder.py:27-34).  A draw is a pure function of
//     key = seed (64 bit),  counter = (element index (64 bit), call offset (32 bit), device step (32 bit)),
// so a kernel generates exactly the values it needs where it needs them, a backward pass can REGENERATE a mask instead of
// reading it back, and nothing depends on launch geometry.  oracle/philox.py restates it in numpy (known-answer vectors of
// the Random123 distribution + the float mapping below) for the tests.
#pragma once
#include "common.h"

namespace arvae {

struct PhiloxKey {
    uint32_t k0, k1;
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, PhiloxKey k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.k0, lo1, hi0 ^ c.w ^ k.k1, lo0);
        k.k0 += 0x9E3779B9u;
        k.k1 += 0xBB67AE85u;
    }
    return c;
}

// what the library's draws are keyed by; `dev_step` (may be null) is a device counter a captured graph advances itself
struct RngStream {
    uint64_t seed;
    uint32_t offset;             // call index inside the step (host side)
    const uint32_t *dev_step;    // optional device word added to the fourth counter word (0 when null)
    uint32_t step;               // host step number
};

__device__ __forceinline__ uint4 rng_block(const RngStream &s, uint64_t index) {
    const uint32_t c3 = s.step + (s.dev_step != nullptr ? s.dev_step[0] : 0u);
    return philox4x32_10(make_uint4((uint32_t)index, (uint32_t)(index >> 32), s.offset, c3),
                         PhiloxKey{(uint32_t)s.seed, (uint32_t)(s.seed >> 32)});
}

// u in (0, 1]: (x + 1) * 2^-32 evaluated in float (x = 2^32 - 1 rounds to 1.0, x = 0 gives 2^-32)
__device__ __forceinline__ float rng_unit(uint32_t x) { return ((float)x + 1.0f) * 2.3283064365386963e-10f; }

// one standard normal per element: Box-Muller on the first two words of the element's own block
__device__ __forceinline__ float rng_normal(const RngStream &s, uint64_t index) {
    const uint4 b = rng_block(s, index);
    const float r = sqrtf(-2.0f * logf(rng_unit(b.x)));
    return r * cosf(6.283185307179586f * rng_unit(b.y));
}

}  // namespace arvae
