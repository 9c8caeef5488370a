// Client side of the boundary: secret keys, bootstrapping / key-switching key material, LWE
// encryption and decryption of booleans.  Host CPU code, as in the reference, where
// tfhe::boolean::gen_keys() and ClientKey::{encrypt,decrypt} run on the client
// (/root/reference/tfhe-aes/src/main.rs:64, /root/reference/base/src/primitive.rs:63,71,85).
//
// Randomness: ChaCha20, one stream per key component so generation parallelises deterministically.  seed == 0 (the
// production setting, what gen_keys() of the reference does) keys the cipher with 256 bits of OS entropy and fails
// when the OS has none to give.  A non-zero 64-bit seed expands to the key through splitmix64: reproducible streams for
// tests and benchmarks, NOT secret (64 bits of entropy at most).
#include <sys/random.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/tfhe_aes_hip.h"

namespace {

struct ChaCha {
    uint32_t key[8];
    uint64_t stream = 0, counter = 0;
    uint32_t buf[16];
    int pos = 16;

    static uint64_t mix(uint64_t &s) {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    ChaCha(const uint32_t (&k)[8], uint64_t stream_id) : stream(stream_id) { std::memcpy(key, k, sizeof key); }
    // test / benchmark keys: the 64-bit seed stretched by splitmix64
    static void key_from_seed(uint64_t seed, uint32_t (&k)[8]) {
        uint64_t s = seed;
        for (int i = 0; i < 4; i++) {
            uint64_t v = mix(s);
            k[2 * i] = (uint32_t)v;
            k[2 * i + 1] = (uint32_t)(v >> 32);
        }
    }
    static inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    static inline void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) in[4 + i] = key[i];
        in[12] = (uint32_t)counter;
        in[13] = (uint32_t)(counter >> 32);
        in[14] = (uint32_t)stream;
        in[15] = (uint32_t)(stream >> 32);
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int r = 0; r < 10; r++) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf[i] = x[i] + in[i];
        counter++;
        pos = 0;
    }
    inline uint32_t u32() {
        if (pos == 16) refill();
        return buf[pos++];
    }
    inline double unit() {  // uniform in (0, 1)
        uint64_t v = ((uint64_t)u32() << 32) | u32();
        return ((double)(v >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
    // two torus noise samples round(N(0,1) * std * 2^32) by Box-Muller
    inline void gauss2(double std_dev, uint32_t &e0, uint32_t &e1) {
        double u1 = unit(), u2 = unit();
        double rad = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925286766559 * u2;
        double s = std_dev * 4294967296.0;
        e0 = (uint32_t)(int64_t)std::llrint(rad * std::cos(th) * s);
        e1 = (uint32_t)(int64_t)std::llrint(rad * std::sin(th) * s);
    }
    void gauss_fill(double std_dev, uint32_t *out, uint32_t count) {
        uint32_t i = 0, a, b;
        for (; i + 1 < count; i += 2) gauss2(std_dev, out[i], out[i + 1]);
        if (i < count) {
            gauss2(std_dev, a, b);
            out[i] = a;
        }
    }
};

// 256 bits from the OS (getrandom, else /dev/urandom); false when neither delivers
bool os_key(uint32_t (&k)[8]) {
    size_t got = 0;
    unsigned char *dst = reinterpret_cast<unsigned char *>(k);
    while (got < sizeof k) {
        ssize_t r = getrandom(dst + got, sizeof k - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got == sizeof k) return true;
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (!f) return false;
    const bool ok = std::fread(k, sizeof k, 1, f) == 1;
    std::fclose(f);
    return ok;
}
bool make_key(uint64_t seed, uint32_t (&k)[8]) {
    if (seed == 0) return os_key(k);
    ChaCha::key_from_seed(seed, k);
    return true;
}

// body += mask * key (negacyclic, binary key)
void addmul_binary(uint32_t *body, const uint32_t *mask, const uint32_t *key, uint32_t N) {
    for (uint32_t j = 0; j < N; j++) {
        if (!key[j]) continue;
        for (uint32_t b = 0; b + j < N; b++) body[j + b] += mask[b];
        for (uint32_t b = N - j; b < N; b++) body[j + b - N] -= mask[b];
    }
}

bool params_ok(const tfa_params *p) {
    return p && p->lwe_dimension && p->glwe_dimension && p->polynomial_size && p->pbs_level && p->ks_level && p->pbs_base_log * p->pbs_level <= 32 &&
           p->ks_base_log * p->ks_level <= 32 && p->ks_level <= 64;
}

}  // namespace

extern "C" int tfa_client_keygen(const tfa_params *p, uint64_t seed, uint32_t *lwe_sk, uint32_t *glwe_sk, uint32_t *bsk, uint32_t *ksk) {
    if (!params_ok(p) || !lwe_sk || !glwe_sk || !bsk || !ksk) return TFA_EINVAL;
    uint32_t seed_key[8];
    if (!make_key(seed, seed_key)) return TFA_ESTATE;  // no OS entropy: never fall back to a fixed key
    const uint32_t n = p->lwe_dimension, k = p->glwe_dimension, N = p->polynomial_size, l = p->pbs_level, k1 = k + 1;
    {
        ChaCha r(seed_key, 1);
        for (uint32_t i = 0; i < n; i++) lwe_sk[i] = r.u32() & 1u;
    }
    {
        ChaCha r(seed_key, 2);
        for (uint32_t i = 0; i < k * N; i++) glwe_sk[i] = r.u32() & 1u;
    }
    // bootstrapping key: GGSW(s_i), rows = GLWE encryptions of zero plus s_i * q / beta^j on polynomial `row`
    const uint64_t ggsw_words = (uint64_t)l * k1 * k1 * N;
#pragma omp parallel for schedule(dynamic, 8)
    for (uint32_t i = 0; i < n; i++) {
        ChaCha masks(seed_key, (3ull << 32) | i), noise(seed_key, (4ull << 32) | i);
        uint32_t *ggsw = bsk + (uint64_t)i * ggsw_words;
        for (uint32_t j = 0; j < l; j++)
            for (uint32_t row = 0; row < k1; row++) {
                uint32_t *glwe = ggsw + ((uint64_t)(j * k1 + row) * k1) * N;
                for (uint32_t w = 0; w < k * N; w++) glwe[w] = masks.u32();
                uint32_t *body = glwe + (uint64_t)k * N;
                noise.gauss_fill(p->glwe_noise_std, body, N);
                for (uint32_t c = 0; c < k; c++) addmul_binary(body, glwe + (uint64_t)c * N, glwe_sk + (uint64_t)c * N, N);
                if (lwe_sk[i]) glwe[(uint64_t)row * N] += 1u << (32 - p->pbs_base_log * (j + 1));
            }
    }
    // key-switching key: KSK[i][j] = LWE_s(s'_i * q / beta_ks^(j+1))
    const uint32_t lk = p->ks_level, big = k * N;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < big; i++) {
        ChaCha masks(seed_key, (5ull << 32) | i), noise(seed_key, (6ull << 32) | i);
        uint32_t e[64];
        noise.gauss_fill(p->lwe_noise_std, e, lk);
        for (uint32_t j = 0; j < lk; j++) {
            uint32_t *ct = ksk + ((uint64_t)i * lk + j) * (n + 1);
            uint32_t b = e[j];
            for (uint32_t w = 0; w < n; w++) {
                ct[w] = masks.u32();
                b += ct[w] * lwe_sk[w];
            }
            if (glwe_sk[i]) b += 1u << (32 - p->ks_base_log * (j + 1));
            ct[n] = b;
        }
    }
    return TFA_OK;
}

extern "C" int tfa_client_encrypt(const tfa_params *p, const uint32_t *lwe_sk, uint64_t seed, const uint8_t *bits, uint32_t count, uint32_t *lwe) {
    if (!params_ok(p) || !lwe_sk || (!bits && count) || (!lwe && count)) return TFA_EINVAL;
    uint32_t seed_key[8];
    if (!make_key(seed, seed_key)) return TFA_ESTATE;
    const uint32_t n = p->lwe_dimension;
    ChaCha masks(seed_key, 7ull << 32), noise(seed_key, (7ull << 32) ^ 1ull);
    for (uint32_t g = 0; g < count; g++) {
        uint32_t *ct = lwe + (uint64_t)g * (n + 1), e0, e1;
        noise.gauss2(p->lwe_noise_std, e0, e1);
        uint32_t b = e0 + (bits[g] ? (1u << 29) : (7u << 29));  // true = +1/8, false = -1/8
        for (uint32_t w = 0; w < n; w++) {
            ct[w] = masks.u32();
            b += ct[w] * lwe_sk[w];
        }
        ct[n] = b;
    }
    return TFA_OK;
}

extern "C" int tfa_client_decrypt(const tfa_params *p, const uint32_t *lwe_sk, const uint32_t *lwe, uint32_t count, uint8_t *bits) {
    if (!params_ok(p) || !lwe_sk || (!lwe && count) || (!bits && count)) return TFA_EINVAL;
    const uint32_t n = p->lwe_dimension;
    for (uint32_t g = 0; g < count; g++) {
        const uint32_t *ct = lwe + (uint64_t)g * (n + 1);
        uint32_t phase = ct[n];
        for (uint32_t w = 0; w < n; w++) phase -= ct[w] * lwe_sk[w];
        bits[g] = phase < (1u << 31);
    }
    return TFA_OK;
}
