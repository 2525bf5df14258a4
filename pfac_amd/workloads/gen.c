/*
 * gen.c -- deterministic synthetic input streams for the BASELINE.json
 * configurations (bench / test infrastructure; not part of the match path).
 *
 * All randomness is splitmix64 seeded by the caller, so a stream is a pure
 * function of (seed, parameters) and a prefix of a longer stream with the same
 * seed is identical to the shorter one (needed to build slice overlaps for the
 * multi-GPU configuration without generating the neighbour slice).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* uniform random bytes (configuration C2 input) */
void wl_fill_random(uint64_t seed, uint8_t *out, uint64_t n)
{
    uint64_t s = seed;
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t v = splitmix64(&s);
        memcpy(out + i, &v, 8);
    }
    if (i < n) {
        uint64_t v = splitmix64(&s);
        memcpy(out + i, &v, (size_t)(n - i));
    }
}

/*
 * Concatenate pseudo-randomly chosen records of a pool until n bytes are
 * written (the last record is truncated).  A 0x00 byte inside a record is a
 * placeholder that is replaced by a fresh random byte from `alphabet`, so the
 * stream is not periodic in the pool.  Used for the HTTP-like stream (C3/C4)
 * and the adversarial near-miss stream (C5).
 */
void wl_fill_from_pool(uint64_t seed, const uint8_t *pool, const uint64_t *offsets, uint64_t count,
                       const uint8_t *alphabet, uint32_t alphabet_len, uint8_t *out, uint64_t n)
{
    uint64_t s = seed;
    uint64_t w = 0;
    uint64_t bits = 0;
    int have = 0;
    while (w < n) {
        const uint64_t r = splitmix64(&s) % count;
        const uint8_t *rec = pool + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len > n - w) len = n - w;
        for (uint64_t i = 0; i < len; i++) {
            uint8_t b = rec[i];
            if (b == 0) {
                if (have == 0) { bits = splitmix64(&s); have = 4; }
                b = alphabet[(uint32_t)(bits & 0xFFFF) % alphabet_len];
                bits >>= 16;
                have--;
            }
            out[w + i] = b;
        }
        w += len;
    }
}

/* FNV-1a 64 over a byte range (to pin generator output in tests) */
uint64_t wl_fnv1a(const uint8_t *p, uint64_t n)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}


static uint64_t pow_u64(uint64_t b, uint64_t e)
{
    uint64_t r = 1;
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return r;
}

/* FNV-1a 64 over the little-endian int32 vector of n elements that is zero everywhere except
 * v[pos[i]] = ids[i] (pos ascending): a zero byte only multiplies the state by the FNV prime, so a run
 * of zeros is one modular power.  Lets a test pin a 4 GiB result vector from its sparse form. */
uint64_t wl_fnv1a_sparse_i32(const int64_t *pos, const int32_t *ids, uint64_t count, uint64_t n)
{
    const uint64_t P = 0x100000001b3ULL;
    uint64_t h = 0xcbf29ce484222325ULL, prev = 0;
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t p = (uint64_t)pos[i];
        h *= pow_u64(P, 4 * (p - prev));
        uint32_t v = (uint32_t)ids[i];
        for (int j = 0; j < 4; j++) { h ^= (v >> (8 * j)) & 0xFF; h *= P; }
        prev = p + 1;
    }
    return h * pow_u64(P, 4 * (n - prev));
}
