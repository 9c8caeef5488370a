/*
 * CPU oracle (test infrastructure, never linked into the product): see tfhe_oracle.h for scope
 * and parity status.  Restates the algorithm every `server_key.{and,or,xor,mux}` call of the
 * reference executes (call sites /root/reference/base/src/boolean_tree.rs:644-654,
 * /root/reference/base/src/primitive.rs:96-145,349-362) inside tfhe 0.11.0 @ 4e2db929
 * (Cargo.lock:700-703): gate linear combination -> modulus switch -> blind rotation
 * (GGSW x GLWE external products) -> sample extract -> LWE keyswitch, with the
 * tfhe::boolean DEFAULT_PARAMETERS (bootstrap-then-keyswitch, u32 torus).
 */
#include "tfhe_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ parameters */
void orc_default_params(orc_params *p) {
    /* tfhe::boolean::parameters::DEFAULT_PARAMETERS as used by gen_keys()
       (/root/reference/tfhe-aes/src/main.rs:64) */
    p->lwe_dimension = 805;
    p->glwe_dimension = 3;
    p->polynomial_size = 512;
    p->pbs_base_log = 10;
    p->pbs_level = 2;
    p->ks_base_log = 3;
    p->ks_level = 5;
    p->lwe_noise_std = 5.8615896642671336e-06;
    p->glwe_noise_std = 9.315272083503367e-10;
}
uint64_t orc_bsk_words(const orc_params *p) {
    uint64_t k1 = p->glwe_dimension + 1;
    return (uint64_t)p->lwe_dimension * p->pbs_level * k1 * k1 * p->polynomial_size;
}
uint64_t orc_ksk_words(const orc_params *p) {
    return (uint64_t)p->glwe_dimension * p->polynomial_size * p->ks_level * (p->lwe_dimension + 1);
}
uint64_t orc_bsk_fourier_doubles(const orc_params *p) { return orc_bsk_words(p); /* N/2 complex = N doubles per poly */ }
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ ChaCha20 CSPRNG */
typedef struct {
    uint32_t key[8];
    uint64_t stream, counter;
    uint32_t buf[16];
    int pos;
} rng_t;
static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rng_init(rng_t *r, uint64_t seed, uint64_t stream) {
    uint64_t s = seed;
    for (int i = 0; i < 4; i++) {
        uint64_t v = splitmix64(&s);
        r->key[2 * i] = (uint32_t)v;
        r->key[2 * i + 1] = (uint32_t)(v >> 32);
    }
    r->stream = stream;
    r->counter = 0;
    r->pos = 16;
}
#define ROTL(a, b) (((a) << (b)) | ((a) >> (32 - (b))))
#define QR(a, b, c, d) a += b, d ^= a, d = ROTL(d, 16), c += d, b ^= c, b = ROTL(b, 12), a += b, d ^= a, d = ROTL(d, 8), c += d, b ^= c, b = ROTL(b, 7)
static void rng_refill(rng_t *r) {
    uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) in[4 + i] = r->key[i];
    in[12] = (uint32_t)r->counter;
    in[13] = (uint32_t)(r->counter >> 32);
    in[14] = (uint32_t)r->stream;
    in[15] = (uint32_t)(r->stream >> 32);
    uint32_t x[16];
    memcpy(x, in, sizeof x);
    for (int i = 0; i < 10; i++) {
        QR(x[0], x[4], x[8], x[12]);
        QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]);
        QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]);
        QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);
        QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) r->buf[i] = x[i] + in[i];
    r->counter++;
    r->pos = 0;
}
static inline uint32_t rng_u32(rng_t *r) {
    if (r->pos == 16) rng_refill(r);
    return r->buf[r->pos++];
}
static inline double rng_unit(rng_t *r) { /* (0,1) */
    uint64_t v = ((uint64_t)rng_u32(r) << 32) | rng_u32(r);
    return ((double)(v >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
/* torus noise: round(N(0,1) * std * 2^32), Box-Muller, two samples per call */
static void rng_gauss2(rng_t *r, double std, uint32_t *e0, uint32_t *e1) {
    double u1 = rng_unit(r), u2 = rng_unit(r);
    double rad = sqrt(-2.0 * log(u1)), th = 6.283185307179586476925286766559 * u2;
    double s = std * 4294967296.0;
    *e0 = (uint32_t)(int64_t)llrint(rad * cos(th) * s);
    *e1 = (uint32_t)(int64_t)llrint(rad * sin(th) * s);
}
static void rng_gauss_fill(rng_t *r, double std, uint32_t *out, uint32_t count) {
    uint32_t i = 0, a, b;
    for (; i + 1 < count; i += 2) rng_gauss2(r, std, &out[i], &out[i + 1]);
    if (i < count) {
        rng_gauss2(r, std, &a, &b);
        out[i] = a;
    }
}

/* ------------------------------------------------------------------ small helpers */
uint32_t orc_modswitch(uint32_t x, uint32_t log2_two_n) {
    /* closest multiple of q/2N, as an index mod 2N */
    uint32_t v = x >> (32 - log2_two_n - 1);
    v = (v + 1) >> 1;
    return v & ((1u << log2_two_n) - 1);
}
void orc_decompose(uint32_t x, uint32_t base_log, uint32_t level, int32_t *digits) {
    /* signed closest-representable decomposition; digits[j-1] pairs with scale 2^(32 - base_log*j) */
    uint32_t rep = base_log * level, non_rep = 32 - rep;
    uint32_t state = ((x >> (non_rep - 1)) + 1) >> 1;
    if (rep < 32) state &= (1u << rep) - 1;
    uint32_t mask = (1u << base_log) - 1;
    for (uint32_t lv = level; lv >= 1; lv--) {
        uint32_t res = state & mask;
        state >>= base_log;
        uint32_t carry = (((res - 1u) | state) & res) >> (base_log - 1);
        state += carry;
        digits[lv - 1] = (int32_t)res - (int32_t)(carry << base_log);
    }
}
uint32_t orc_phase(const uint32_t *sk, uint32_t dim, const uint32_t *ct) {
    uint32_t acc = ct[dim];
    for (uint32_t i = 0; i < dim; i++) acc -= ct[i] * sk[i];
    return acc;
}
static uint32_t ilog2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v) l++;
    return l;
}

/* negacyclic helpers on uint32 polynomials */
static void poly_mul_monomial(const uint32_t *in, uint32_t N, uint32_t a /* [0,2N) */, uint32_t *out) {
    uint32_t neg = 0;
    if (a >= N) {
        a -= N;
        neg = 1;
    }
    for (uint32_t j = 0; j < N; j++) {
        uint32_t v = (j >= a) ? in[j - a] : (uint32_t)(0u - in[j + N - a]);
        out[j] = neg ? (uint32_t)(0u - v) : v;
    }
}
/* out += a * s, s binary, negacyclic */
static void poly_addmul_binary(uint32_t *out, const uint32_t *a, const uint32_t *s, uint32_t N) {
    for (uint32_t j = 0; j < N; j++) {
        if (!s[j]) continue;
        for (uint32_t b = 0; b < N - j; b++) out[j + b] += a[b];
        for (uint32_t b = N - j; b < N; b++) out[j + b - N] -= a[b];
    }
}

/* ------------------------------------------------------------------ key generation, encryption */
void orc_keygen(const orc_params *p, uint64_t seed, uint32_t *lwe_sk, uint32_t *glwe_sk, uint32_t *bsk, uint32_t *ksk) {
    const uint32_t n = p->lwe_dimension, k = p->glwe_dimension, N = p->polynomial_size, l = p->pbs_level, k1 = k + 1;
    rng_t r;
    rng_init(&r, seed, 1);
    for (uint32_t i = 0; i < n; i++) lwe_sk[i] = rng_u32(&r) & 1u;
    rng_init(&r, seed, 2);
    for (uint32_t i = 0; i < k * N; i++) glwe_sk[i] = rng_u32(&r) & 1u;

    /* BSK_i = GGSW(s_i): row (level j, r) is a GLWE encryption of zero with s_i * 2^(32 - beta*j)
       added to polynomial r (mask r for r<k, body for r==k), constant coefficient. */
    const uint64_t ggsw_words = (uint64_t)l * k1 * k1 * N;
#pragma omp parallel for schedule(dynamic, 8)
    for (uint32_t i = 0; i < n; i++) {
        rng_t rm, re;
        rng_init(&rm, seed, (3ull << 32) | i);
        rng_init(&re, seed, (4ull << 32) | i);
        uint32_t *g = bsk + (uint64_t)i * ggsw_words;
        for (uint32_t j = 0; j < l; j++)
            for (uint32_t row = 0; row < k1; row++) {
                uint32_t *glwe = g + ((uint64_t)(j * k1 + row) * k1) * N;
                for (uint32_t w = 0; w < k * N; w++) glwe[w] = rng_u32(&rm);
                uint32_t *body = glwe + (uint64_t)k * N;
                rng_gauss_fill(&re, p->glwe_noise_std, body, N);
                for (uint32_t c = 0; c < k; c++) poly_addmul_binary(body, glwe + (uint64_t)c * N, glwe_sk + (uint64_t)c * N, N);
                if (lwe_sk[i]) glwe[(uint64_t)row * N] += 1u << (32 - p->pbs_base_log * (j + 1));
            }
    }
    /* KSK[i][j] = LWE_s( s'_i * 2^(32 - beta_ks*(j+1)) ), s' = flattened GLWE key */
    const uint32_t lk = p->ks_level, kN = k * N;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < kN; i++) {
        rng_t rm, re;
        rng_init(&rm, seed, (5ull << 32) | i);
        rng_init(&re, seed, (6ull << 32) | i);
        uint32_t noise[64];
        rng_gauss_fill(&re, p->lwe_noise_std, noise, lk);
        for (uint32_t j = 0; j < lk; j++) {
            uint32_t *ct = ksk + ((uint64_t)i * lk + j) * (n + 1);
            uint32_t b = noise[j];
            for (uint32_t w = 0; w < n; w++) {
                ct[w] = rng_u32(&rm);
                b += ct[w] * lwe_sk[w];
            }
            if (glwe_sk[i]) b += 1u << (32 - p->ks_base_log * (j + 1));
            ct[n] = b;
        }
    }
}

void orc_encrypt_bits(const orc_params *p, const uint32_t *lwe_sk, uint64_t seed, uint64_t stream, const uint8_t *bits,
                      uint32_t count, uint32_t *cts) {
    const uint32_t n = p->lwe_dimension;
    rng_t rm, re;
    rng_init(&rm, seed, (7ull << 32) ^ (stream << 1));
    rng_init(&re, seed, (7ull << 32) ^ (stream << 1) ^ 1);
    for (uint32_t g = 0; g < count; g++) {
        uint32_t *ct = cts + (uint64_t)g * (n + 1), e0, e1;
        rng_gauss2(&re, p->lwe_noise_std, &e0, &e1);
        uint32_t b = e0 + (bits[g] ? (1u << 29) : (7u << 29));
        for (uint32_t w = 0; w < n; w++) {
            ct[w] = rng_u32(&rm);
            b += ct[w] * lwe_sk[w];
        }
        ct[n] = b;
    }
}
void orc_decrypt_bits(const orc_params *p, const uint32_t *lwe_sk, const uint32_t *cts, uint32_t count, uint8_t *bits) {
    const uint32_t n = p->lwe_dimension;
    for (uint32_t g = 0; g < count; g++) bits[g] = orc_phase(lwe_sk, n, cts + (uint64_t)g * (n + 1)) < (1u << 31);
}

/* ------------------------------------------------------------------ gate pre-combination */
void orc_gate_precombine(const orc_params *p, int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *u1,
                         uint32_t *u2) {
    const uint32_t n = p->lwe_dimension, T = 1u << 29;
    switch (op) {
    case ORC_AND:
        for (uint32_t i = 0; i <= n; i++) u1[i] = a[i] + b[i];
        u1[n] -= T;
        break;
    case ORC_NAND:
        for (uint32_t i = 0; i <= n; i++) u1[i] = 0u - (a[i] + b[i]);
        u1[n] += T;
        break;
    case ORC_OR:
        for (uint32_t i = 0; i <= n; i++) u1[i] = a[i] + b[i];
        u1[n] += T;
        break;
    case ORC_NOR:
        for (uint32_t i = 0; i <= n; i++) u1[i] = 0u - (a[i] + b[i]);
        u1[n] -= T;
        break;
    case ORC_XOR:
        for (uint32_t i = 0; i <= n; i++) u1[i] = 2u * (a[i] + b[i]);
        u1[n] += 2u * T;
        break;
    case ORC_XNOR:
        for (uint32_t i = 0; i <= n; i++) u1[i] = 0u - 2u * (a[i] + b[i]);
        u1[n] -= 2u * T;
        break;
    case ORC_MUX: /* a = condition, b = then, c = else */
        for (uint32_t i = 0; i <= n; i++) {
            u1[i] = a[i] + b[i];
            u2[i] = c[i] - a[i];
        }
        u1[n] -= T;
        u2[n] -= T;
        break;
    default:
        break;
    }
}

/* ------------------------------------------------------------------ accumulator ops */
void orc_acc_init(const orc_params *p, uint32_t b_tilde, uint32_t *acc) {
    const uint32_t k = p->glwe_dimension, N = p->polynomial_size;
    memset(acc, 0, sizeof(uint32_t) * (size_t)k * N);
    uint32_t tv[4096];
    for (uint32_t j = 0; j < N; j++) tv[j] = 1u << 29;
    poly_mul_monomial(tv, N, (2 * N - b_tilde) % (2 * N), acc + (size_t)k * N);
}
void orc_rotate_sub(const orc_params *p, const uint32_t *acc, uint32_t a, uint32_t *d) {
    const uint32_t k1 = p->glwe_dimension + 1, N = p->polynomial_size;
    for (uint32_t r = 0; r < k1; r++) {
        poly_mul_monomial(acc + (size_t)r * N, N, a, d + (size_t)r * N);
        for (uint32_t j = 0; j < N; j++) d[(size_t)r * N + j] -= acc[(size_t)r * N + j];
    }
}
void orc_external_product_exact(const orc_params *p, const uint32_t *ggsw, const uint32_t *d, uint32_t *out) {
    const uint32_t k1 = p->glwe_dimension + 1, N = p->polynomial_size, l = p->pbs_level;
    int32_t dig[8];
    int32_t *digs = (int32_t *)malloc(sizeof(int32_t) * (size_t)l * k1 * N);
    for (uint32_t r = 0; r < k1; r++)
        for (uint32_t j = 0; j < N; j++) {
            orc_decompose(d[(size_t)r * N + j], p->pbs_base_log, l, dig);
            for (uint32_t lv = 0; lv < l; lv++) digs[((size_t)lv * k1 + r) * N + j] = dig[lv];
        }
    for (uint32_t lv = 0; lv < l; lv++)
        for (uint32_t r = 0; r < k1; r++) {
            const int32_t *dp = digs + ((size_t)lv * k1 + r) * N;
            for (uint32_t c = 0; c < k1; c++) {
                const uint32_t *g = ggsw + (((size_t)lv * k1 + r) * k1 + c) * N;
                uint32_t *o = out + (size_t)c * N;
                for (uint32_t a = 0; a < N; a++) {
                    uint32_t dv = (uint32_t)dp[a];
                    if (!dv) continue;
                    for (uint32_t b = 0; b < N - a; b++) o[a + b] += dv * g[b];
                    for (uint32_t b = N - a; b < N; b++) o[a + b - N] -= dv * g[b];
                }
            }
        }
    free(digs);
}
void orc_sample_extract(const orc_params *p, const uint32_t *acc, uint32_t *big) {
    const uint32_t k = p->glwe_dimension, N = p->polynomial_size;
    for (uint32_t r = 0; r < k; r++) {
        const uint32_t *a = acc + (size_t)r * N;
        uint32_t *o = big + (size_t)r * N;
        o[0] = a[0];
        for (uint32_t j = 1; j < N; j++) o[j] = 0u - a[N - j];
    }
    big[(size_t)k * N] = acc[(size_t)k * N];
}
void orc_keyswitch(const orc_params *p, const uint32_t *ksk, const uint32_t *big, uint32_t *out) {
    const uint32_t n = p->lwe_dimension, kN = p->glwe_dimension * p->polynomial_size, lk = p->ks_level;
    int32_t dig[32];
    memset(out, 0, sizeof(uint32_t) * n);
    out[n] = big[kN];
    for (uint32_t i = 0; i < kN; i++) {
        orc_decompose(big[i], p->ks_base_log, lk, dig);
        for (uint32_t j = 0; j < lk; j++) {
            uint32_t dv = (uint32_t)dig[j];
            if (!dv) continue;
            const uint32_t *row = ksk + ((size_t)i * lk + j) * (n + 1);
            for (uint32_t w = 0; w <= n; w++) out[w] -= dv * row[w];
        }
    }
}

/* ------------------------------------------------------------------ fp64 negacyclic FFT engine */
typedef struct {
    uint32_t N, M;
    double *tw_re, *tw_im; /* twist e^{i pi n / N}, n < M */
    double *w_re, *w_im;   /* e^{2 pi i j / M}, j < M/2 */
} fft_plan;
static fft_plan g_plan = {0, 0, 0, 0, 0, 0};
static const fft_plan *get_plan(uint32_t N) {
#pragma omp critical(orc_plan)
    {
        if (g_plan.N != N) {
            uint32_t M = N / 2;
            g_plan.tw_re = (double *)malloc(sizeof(double) * M);
            g_plan.tw_im = (double *)malloc(sizeof(double) * M);
            g_plan.w_re = (double *)malloc(sizeof(double) * M);
            g_plan.w_im = (double *)malloc(sizeof(double) * M);
            for (uint32_t i = 0; i < M; i++) {
                double a = 3.14159265358979323846264338327950288 * (double)i / (double)N;
                g_plan.tw_re[i] = cos(a);
                g_plan.tw_im[i] = sin(a);
                double b = 6.283185307179586476925286766559 * (double)i / (double)M;
                g_plan.w_re[i] = cos(b);
                g_plan.w_im[i] = sin(b);
            }
            g_plan.M = M;
            g_plan.N = N;
        }
    }
    return &g_plan;
}
/* in-place decimation-in-frequency, output in bit-reversed order, kernel e^{+2 pi i nk/M} */
static void fft_dif(const fft_plan *pl, double *re, double *im) {
    const uint32_t M = pl->M;
    for (uint32_t len = M / 2, stride = 1; len >= 1; len >>= 1, stride <<= 1)
        for (uint32_t i = 0; i < M; i += 2 * len)
            for (uint32_t j = 0; j < len; j++) {
                double ur = re[i + j], ui = im[i + j], vr = re[i + j + len], vi = im[i + j + len];
                double wr = pl->w_re[j * stride], wi = pl->w_im[j * stride];
                re[i + j] = ur + vr;
                im[i + j] = ui + vi;
                double dr = ur - vr, di = ui - vi;
                re[i + j + len] = dr * wr - di * wi;
                im[i + j + len] = dr * wi + di * wr;
            }
}
/* in-place decimation-in-time taking bit-reversed input, kernel e^{-2 pi i nk/M}, unscaled */
static void fft_dit_inv(const fft_plan *pl, double *re, double *im) {
    const uint32_t M = pl->M;
    for (uint32_t len = 1, stride = M / 2; len < M; len <<= 1, stride >>= 1)
        for (uint32_t i = 0; i < M; i += 2 * len)
            for (uint32_t j = 0; j < len; j++) {
                double wr = pl->w_re[j * stride], wi = -pl->w_im[j * stride];
                double xr = re[i + j + len], xi = im[i + j + len];
                double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                double ur = re[i + j], ui = im[i + j];
                re[i + j] = ur + vr;
                im[i + j] = ui + vi;
                re[i + j + len] = ur - vr;
                im[i + j + len] = ui - vi;
            }
}
static void fwd_negacyclic(const fft_plan *pl, const double *poly /*[N]*/, double *re, double *im) {
    const uint32_t M = pl->M;
    for (uint32_t i = 0; i < M; i++) {
        double a = poly[i], b = poly[i + M];
        re[i] = a * pl->tw_re[i] - b * pl->tw_im[i];
        im[i] = a * pl->tw_im[i] + b * pl->tw_re[i];
    }
    fft_dif(pl, re, im);
}
void orc_bsk_to_fourier(const orc_params *p, const uint32_t *bsk, double *bsk_f) {
    const uint32_t N = p->polynomial_size, M = N / 2;
    const fft_plan *pl = get_plan(N);
    const uint64_t polys = orc_bsk_words(p) / N;
#pragma omp parallel for schedule(static)
    for (uint64_t q = 0; q < polys; q++) {
        double tmp[4096], re[2048], im[2048];
        for (uint32_t j = 0; j < N; j++) tmp[j] = (double)(int32_t)bsk[q * N + j];
        fwd_negacyclic(pl, tmp, re, im);
        double *o = bsk_f + q * N;
        for (uint32_t j = 0; j < M; j++) { /* fold the 1/M of the inverse transform into the key */
            o[j] = re[j] / (double)M;
            o[M + j] = im[j] / (double)M;
        }
    }
}
void orc_external_product_fft(const orc_params *p, const double *ggsw_f, const uint32_t *d, uint32_t *out) {
    const uint32_t k1 = p->glwe_dimension + 1, N = p->polynomial_size, M = N / 2, l = p->pbs_level;
    const fft_plan *pl = get_plan(N);
    double acc_re[8][2048], acc_im[8][2048];
    double tmp[4096], fr[2048], fi[2048];
    int32_t dig[8];
    for (uint32_t c = 0; c < k1; c++) {
        memset(acc_re[c], 0, sizeof(double) * M);
        memset(acc_im[c], 0, sizeof(double) * M);
    }
    for (uint32_t lv = 0; lv < l; lv++)
        for (uint32_t r = 0; r < k1; r++) {
            for (uint32_t j = 0; j < N; j++) {
                orc_decompose(d[(size_t)r * N + j], p->pbs_base_log, l, dig);
                tmp[j] = (double)dig[lv];
            }
            fwd_negacyclic(pl, tmp, fr, fi);
            for (uint32_t c = 0; c < k1; c++) {
                const double *g = ggsw_f + (((size_t)lv * k1 + r) * k1 + c) * N;
                double *ar = acc_re[c], *ai = acc_im[c];
                for (uint32_t j = 0; j < M; j++) {
                    ar[j] += fr[j] * g[j] - fi[j] * g[M + j];
                    ai[j] += fr[j] * g[M + j] + fi[j] * g[j];
                }
            }
        }
    for (uint32_t c = 0; c < k1; c++) {
        fft_dit_inv(pl, acc_re[c], acc_im[c]);
        uint32_t *o = out + (size_t)c * N;
        for (uint32_t j = 0; j < M; j++) {
            /* untwist by e^{-i pi j/N}; real part -> coeff j, imaginary part -> coeff j+M */
            double xr = acc_re[c][j] * pl->tw_re[j] + acc_im[c][j] * pl->tw_im[j];
            double xi = acc_im[c][j] * pl->tw_re[j] - acc_re[c][j] * pl->tw_im[j];
            xr -= 4294967296.0 * rint(xr * (1.0 / 4294967296.0));
            xi -= 4294967296.0 * rint(xi * (1.0 / 4294967296.0));
            o[j] += (uint32_t)(int64_t)llrint(xr);
            o[j + M] += (uint32_t)(int64_t)llrint(xi);
        }
    }
}

/* ------------------------------------------------------------------ bootstrap */
void orc_blind_rotate(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *lwe_in, uint32_t steps,
                      uint32_t *acc) {
    const uint32_t n = p->lwe_dimension, k1 = p->glwe_dimension + 1, N = p->polynomial_size;
    const uint32_t lg = ilog2(2 * N);
    const uint64_t ggsw_words = (uint64_t)p->pbs_level * k1 * k1 * N;
    uint32_t *d = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)k1 * N);
    orc_acc_init(p, orc_modswitch(lwe_in[n], lg), acc);
    for (uint32_t i = 0; i < steps && i < n; i++) {
        uint32_t a = orc_modswitch(lwe_in[i], lg);
        if (a == 0) continue; /* d would be identically zero */
        orc_rotate_sub(p, acc, a, d);
        if (bsk_f)
            orc_external_product_fft(p, bsk_f + i * ggsw_words, d, acc);
        else
            orc_external_product_exact(p, bsk + i * ggsw_words, d, acc);
    }
    free(d);
}
void orc_pbs(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *lwe_in, uint32_t *big) {
    const uint32_t k1 = p->glwe_dimension + 1, N = p->polynomial_size;
    uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)k1 * N);
    orc_blind_rotate(p, bsk, bsk_f, lwe_in, p->lwe_dimension, acc);
    orc_sample_extract(p, acc, big);
    free(acc);
}

/* ------------------------------------------------------------------ gates */
void orc_gates(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *ksk, const uint8_t *ops,
               const uint32_t *in0, const uint32_t *in1, const uint32_t *in2, uint32_t count, uint32_t *out) {
    const uint32_t n = p->lwe_dimension, kN = p->glwe_dimension * p->polynomial_size;
    const size_t w = n + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t g = 0; g < count; g++) {
        const uint32_t *a = in0 + g * w, *b = in1 ? in1 + g * w : 0, *c = in2 ? in2 + g * w : 0;
        uint32_t *o = out + g * w;
        int op = ops[g];
        if (op == ORC_NOT) {
            for (uint32_t i = 0; i <= n; i++) o[i] = 0u - a[i];
            continue;
        }
        if (op == ORC_COPY) {
            memcpy(o, a, sizeof(uint32_t) * w);
            continue;
        }
        uint32_t *u1 = (uint32_t *)malloc(sizeof(uint32_t) * w), *u2 = (uint32_t *)malloc(sizeof(uint32_t) * w);
        uint32_t *big1 = (uint32_t *)malloc(sizeof(uint32_t) * (kN + 1)), *big2 = (uint32_t *)malloc(sizeof(uint32_t) * (kN + 1));
        orc_gate_precombine(p, op, a, b, c, u1, u2);
        orc_pbs(p, bsk, bsk_f, u1, big1);
        if (op == ORC_MUX) {
            orc_pbs(p, bsk, bsk_f, u2, big2);
            for (uint32_t i = 0; i <= kN; i++) big1[i] += big2[i];
            big1[kN] += 1u << 29;
        }
        orc_keyswitch(p, ksk, big1, o);
        free(u1);
        free(u2);
        free(big1);
        free(big2);
    }
}
