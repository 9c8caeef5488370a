/*
 * CPU oracle for the TFHE boolean-gate path used by tfhe-aes.  TEST INFRASTRUCTURE ONLY:
 * nothing in the shipped library may include, link or call this file; only tests/, the
 * smoke check and the cpu_baseline leg of bench.py load it.
 *
 * PARITY STATUS: "ciphertext-level parity unpinned".  The arithmetic restated here lives in
 * the un-vendored dependency tfhe 0.11.0 (zama-ai/tfhe-rs @ 4e2db929, pinned by
 * /root/reference/Cargo.lock:700-703; call sites base/src/boolean_tree.rs:644-654,
 * base/src/primitive.rs:96-145,349-362).  The reference holds no ciphertext, key or torus-level
 * fixture, so this oracle is pinned only at the level the reference's own tests pin the path:
 * decrypted gate results and the FIPS-197 byte vectors.
 *
 * All torus arithmetic is wrapping uint32 (q = 2^32).  Two polynomial-product engines:
 *   exact : schoolbook negacyclic convolution in integers (ground truth, deterministic)
 *   fft   : fp64 negacyclic FFT (the algorithm the reference CPU path runs; timed as cpu_baseline)
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t lwe_dimension;   /* n   */
    uint32_t glwe_dimension;  /* k   */
    uint32_t polynomial_size; /* N   */
    uint32_t pbs_base_log, pbs_level;
    uint32_t ks_base_log, ks_level;
    double lwe_noise_std, glwe_noise_std; /* as a fraction of the torus */
} orc_params;

/* gate opcodes shared with the C-ABI of the engine */
enum { ORC_AND = 0, ORC_NAND = 1, ORC_OR = 2, ORC_NOR = 3, ORC_XOR = 4, ORC_XNOR = 5, ORC_MUX = 6, ORC_NOT = 7, ORC_COPY = 8 };

void orc_default_params(orc_params *p);

/* sizes in uint32 words */
uint64_t orc_bsk_words(const orc_params *p); /* n*l*(k+1)*(k+1)*N */
uint64_t orc_ksk_words(const orc_params *p); /* kN*l_ks*(n+1)     */

/* Deterministic key generation from a 64-bit seed (ChaCha20 streams). */
void orc_keygen(const orc_params *p, uint64_t seed, uint32_t *lwe_sk /*[n]*/, uint32_t *glwe_sk /*[kN]*/,
                uint32_t *bsk /*[n][l][k+1][k+1][N]*/, uint32_t *ksk /*[kN][l_ks][n+1]*/);

/* LWE encryption of booleans under the small key; stream selects an independent noise stream. */
void orc_encrypt_bits(const orc_params *p, const uint32_t *lwe_sk, uint64_t seed, uint64_t stream,
                      const uint8_t *bits, uint32_t count, uint32_t *cts /*[count][n+1]*/);
uint32_t orc_phase(const uint32_t *sk, uint32_t dim, const uint32_t *ct);
void orc_decrypt_bits(const orc_params *p, const uint32_t *lwe_sk, const uint32_t *cts, uint32_t count, uint8_t *bits);

/* building blocks */
uint32_t orc_modswitch(uint32_t x, uint32_t log2_two_n);
void orc_decompose(uint32_t x, uint32_t base_log, uint32_t level, int32_t *digits /*[level], index 0 = level 1*/);
void orc_gate_precombine(const orc_params *p, int op, const uint32_t *in0, const uint32_t *in1, const uint32_t *in2,
                         uint32_t *u1, uint32_t *u2 /* MUX only */);
/* acc[k+1][N]: init test vector and rotate by -b~ */
void orc_acc_init(const orc_params *p, uint32_t b_tilde, uint32_t *acc);
/* d = acc * X^a - acc */
void orc_rotate_sub(const orc_params *p, const uint32_t *acc, uint32_t a, uint32_t *d);
/* out += GGSW (std domain, [l][k+1][k+1][N]) external-product d, exact integers */
void orc_external_product_exact(const orc_params *p, const uint32_t *ggsw, const uint32_t *d, uint32_t *out);
void orc_sample_extract(const orc_params *p, const uint32_t *acc, uint32_t *big_lwe /*[kN+1]*/);
void orc_keyswitch(const orc_params *p, const uint32_t *ksk, const uint32_t *big_lwe, uint32_t *out /*[n+1]*/);

/* Fourier-domain BSK for the fft engine: n*l*(k+1)*(k+1)*(N/2) complex doubles (re,im interleaved) */
uint64_t orc_bsk_fourier_doubles(const orc_params *p);
void orc_bsk_to_fourier(const orc_params *p, const uint32_t *bsk, double *bsk_f);
void orc_external_product_fft(const orc_params *p, const double *ggsw_f, const uint32_t *d, uint32_t *out);

/* Programmable bootstrap with the constant +1/8 test vector; bsk_f may be NULL -> exact engine. */
void orc_pbs(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *lwe_in, uint32_t *big_lwe_out);
/* blind rotation only, stopped after `steps` iterations (steps==n: full) -> acc[k+1][N] */
void orc_blind_rotate(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *lwe_in,
                      uint32_t steps, uint32_t *acc);

/* Full gates on batches (OpenMP over gates).  in*: [count][n+1]; in2 only for MUX (in0=cond,in1=then,in2=else). */
void orc_gates(const orc_params *p, const uint32_t *bsk, const double *bsk_f, const uint32_t *ksk, const uint8_t *ops,
               const uint32_t *in0, const uint32_t *in1, const uint32_t *in2, uint32_t count, uint32_t *out);

int orc_num_threads(void);
#ifdef __cplusplus
}
#endif
#endif
