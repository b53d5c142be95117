// ses_rng.h -- counter-based Gaussian noise for offspring perturbation.
//
// Integer generator: rocRAND's device-side Philox4x32-10 engine (rocrand_init / rocrand4).
// The engine's (seed, subsequence, offset) triple is used as a pure counter:
//     key          = seed
//     subsequence  = (stream tag << 56) | generation        -> counter words 2,3
//     offset       = 4 * ((row << 32) | column)             -> counter words 0 (column), 1 (row)
// so one rocrand4() call returns the 4 words of Philox(counter = {column, row, gen_lo, tag|gen_hi}).
// The words are turned into normals by a Box-Muller built from ses_math.h, hence bit-reproducible on
// the CPU (oracle/ses_oracle.c restates it with the published Random123 Philox).
// Replaces np.random.normal at offspring_strategies.py:57,173,320 (reference).
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>

#include "ses_math.h"

namespace ses {

constexpr uint64_t TAG_PARAM_NOISE = 0ull;
constexpr uint64_t TAG_ENV_INIT = 1ull;

__device__ __forceinline__ uint4 philox_words(uint64_t seed, uint64_t tag, uint64_t gen, uint32_t row, uint32_t col)
{
    const unsigned long long subseq = (tag << 56) | (gen & 0x00FFFFFFFFFFFFFFull);
    const unsigned long long offset = 4ull * (((unsigned long long)row << 32) | (unsigned long long)col);
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subseq, offset, &st);
    return rocrand4(&st);
}

__device__ __forceinline__ float u32_to_unit(uint32_t r)
{
    return fma_((float)r, 0x1.0p-32f, 0x1.0p-33f);  // (r + 0.5) / 2^32, one rounding
}

__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float &z0, float &z1)
{
    const float u = u32_to_unit(r0);
    const float ang = fma_((float)r1, 0x1.921fb6p-30f, 0x1.921fb6p-31f);
    const float rad = __builtin_sqrtf(-2.0f * log_(u));
    float s, c;
    sincos_(ang, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

// the four N(0,1) draws of (seed, gen, offspring row, parameter quad)
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t gen, uint32_t row, uint32_t quad, float z[4])
{
    const uint4 r = philox_words(seed, TAG_PARAM_NOISE, gen, row, quad);
    box_muller(r.x, r.y, z[0], z[1]);
    box_muller(r.z, r.w, z[2], z[3]);
}

}  // namespace ses
