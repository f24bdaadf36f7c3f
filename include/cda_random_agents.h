/*
 * cda_random_agents.h - the random agent of the reference as a counter-based sampler.
 *
 * The reference's random-agent driver (CDA_rand.py:40-85) and its fixed opponents (RandomRLModule,
 * train/model/model_handler.py:38-53) draw every action component uniformly from its space:
 *   category ~ U{0..8}, price ~ U{0..9}, price_offset ~ U{0..2}, size_mean ~ U[-1,1) f32, size_sigma ~ U[0,1) f32.
 * The reference takes those draws from gymnasium's space sampler, whose stream is a property of that library's
 * version, not of the env; SURVEY §8(a) row H pins the LAW only.  Here the action of (seed, global market index,
 * step, agent) is a pure function - three rounds of the splitmix64 finaliser - so the device kernel
 * (cda_run_random), a host loop and the CPU oracle all see the same stream without sharing any state.
 * Plain C, usable from host and device code.
 */
#ifndef CDA_RANDOM_AGENTS_H
#define CDA_RANDOM_AGENTS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CDA_RA_FN __host__ __device__ static inline
#else
#define CDA_RA_FN static inline
#endif

CDA_RA_FN uint64_t cda_ra_mix(uint64_t z) {              /* splitmix64 (Steele, Lea, Flood 2014) */
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

CDA_RA_FN void cda_random_action(uint64_t seed, uint64_t market, uint32_t step, uint32_t agent,
                                 int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset) {
    const uint64_t h0 = cda_ra_mix(seed + market * 0xd1342543de82ef95ULL);
    const uint64_t w0 = cda_ra_mix(h0 + (((uint64_t)step << 32) | (uint64_t)agent));
    const uint64_t w1 = cda_ra_mix(w0);
    *category = (int32_t)(((w0 & 0xffffffffULL) * 9ULL) >> 32);
    *price = (int32_t)(((w0 >> 32) * 10ULL) >> 32);
    *price_offset = (int32_t)(((w1 & 0xffffffffULL) * 3ULL) >> 32);
    *size_mean = (float)((w1 >> 32) & 0xffffffULL) * (1.0f / 8388608.0f) - 1.0f;      /* 24 bits: exact in f32 */
    *size_sigma = (float)(w1 >> 40) * (1.0f / 16777216.0f);
}

#endif
