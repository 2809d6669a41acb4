/* TEST INFRASTRUCTURE ONLY (oracle): never linked into the product library.
 *
 * Random-number contract shared by the reference harness (oracle/ref_harness)
 * and the CPU restatement (oracle/bdpt_oracle.cpp).
 *
 * The reference draws every random number from one process-global
 * std::mt19937 through std::uniform_real_distribution<double>(0,1)
 * (src/main.h:28-32).  libstdc++ turns two 32-bit engine words g1, g2 into
 *     r = (g1 + g2 * 2^32) / 2^64 ,   r >= 1  ->  nextafter(1, 0)
 * (std::generate_canonical<double, 53>).  That word->double formula is kept
 * bit for bit.  What changes is where the words come from: each
 * (pixel, sample) owns an independent counter-based stream so that a sample's
 * numbers do not depend on which thread / GPU lane rendered the other samples.
 * The draw ORDER inside one sample is the reference's.
 *
 * Two engines:
 *   ORNG_COUNTER  - stream key = f(seed, pixel, sample); draw n = mix64(key + (n+1)*GOLDEN)
 *                   (the product's HIP kernels implement exactly this)
 *   ORNG_MT19937  - one serial default-seeded (5489) Mersenne twister, used only
 *                   to pin the restatement against the UNSHIMMED reference.
 */
#ifndef ORACLE_RNG_H
#define ORACLE_RNG_H

#include <stdint.h>
#include <math.h>

#define ORNG_GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t orng_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* key of the stream owned by sample `sample` of pixel `pixel` (pixel = j*W + i of the FULL image) */
static inline uint64_t orng_stream_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t k = orng_mix64(seed + ORNG_GOLDEN * (pixel + 1));
    return orng_mix64(k ^ (0xD1B54A32D192ED03ULL * (sample + 1)));
}

/* libstdc++ generate_canonical<double,53> applied to the two halves of one 64-bit word */
static inline double orng_words_to_double(uint32_t g1, uint32_t g2) {
    double sum = (double)g1 + (double)g2 * 4294967296.0;
    double r = sum / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}

typedef struct {
    uint64_t key;
    uint64_t counter;   /* number of doubles drawn from this stream so far */
} orng_stream;

static inline double orng_stream_next(orng_stream* s) {
    uint64_t x = orng_mix64(s->key + (s->counter + 1) * ORNG_GOLDEN);
    s->counter++;
    return orng_words_to_double((uint32_t)(x & 0xFFFFFFFFu), (uint32_t)(x >> 32));
}

/* ---- MT19937 (32-bit), default seed 5489, for the serial pin ---- */
typedef struct {
    uint32_t mt[624];
    int idx;
} orng_mt19937;

static inline void orng_mt_seed(orng_mt19937* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static inline uint32_t orng_mt_next(orng_mt19937* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7FFFFFFFu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908B0DFu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= (y >> 18);
    return y;
}

static inline double orng_mt_next_double(orng_mt19937* g) {
    uint32_t g1 = orng_mt_next(g);
    uint32_t g2 = orng_mt_next(g);
    return orng_words_to_double(g1, g2);
}

#endif
