/*
 * tfhe_aes_hip.h - C ABI of the MI355X batched TFHE boolean-gate engine.
 *
 * The reference (pjanthony2001/tfhe-aes) has no FFI: its seam is the tfhe::boolean gate API that
 * `base` calls.  Each entry point below names the reference call site it replaces.  A binding only
 * needs these declarations (plain pointers and sizes; no C++ or torch types cross the boundary).
 *
 * Conventions: every function returns 0 on success or a negative TFA_E* code and never throws or
 * aborts; tfa_last_error() gives the message of the last failure on that context.  The caller owns
 * all host buffers; the context owns all device memory.  One submitting thread per context, one
 * context per GPU.  Ciphertexts live in a device-resident slot pool; only slot indices cross the
 * boundary per stage.  Trivial ciphertexts (tfhe-rs `Ciphertext::Trivial`) never reach the ABI:
 * the host layer resolves them, as tfhe-rs does.
 */
#ifndef TFHE_AES_HIP_H
#define TFHE_AES_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TFA_OK 0
#define TFA_EINVAL -1   /* bad argument / unsupported parameter set */
#define TFA_EHIP -2     /* a HIP runtime call failed */
#define TFA_ESTATE -3   /* call order violated (e.g. submit before keys) */
#define TFA_ENOMEM -4

/* tfhe::boolean::BooleanParameters (DEFAULT_PARAMETERS of gen_keys(),
 * /root/reference/tfhe-aes/src/main.rs:64).  Runtime struct; the kernels are specialised for
 * (k=3, N=512, pbs 2^10 x 2) and the context refuses anything else. */
typedef struct tfa_params {
    uint32_t lwe_dimension, glwe_dimension, polynomial_size;
    uint32_t pbs_base_log, pbs_level, ks_base_log, ks_level;
    double lwe_noise_std, glwe_noise_std;
} tfa_params;

/* Gate opcodes: tfhe::boolean ServerKey::{and,nand,or,nor,xor,xnor,mux,not} plus a slot copy
 * (the `BooleanExpr::Operand(_) => clone` arm of Runnable::run,
 * /root/reference/base/src/boolean_tree.rs:644-654). */
enum tfa_op { TFA_AND = 0, TFA_NAND = 1, TFA_OR = 2, TFA_NOR = 3, TFA_XOR = 4, TFA_XNOR = 5, TFA_MUX = 6, TFA_NOT = 7, TFA_COPY = 8 };

/* One gate of a stage.  in0/in1 are the operands (MUX: in0 = condition, in1 = then, in2 = else);
 * bit i of `neg` negates operand i first (ServerKey::not is a free LWE negation, so it is folded
 * into the consumer instead of costing a slot).  All inputs must come from earlier stages. */
typedef struct tfa_gate {
    uint32_t op;
    uint32_t in0, in1, in2;
    uint32_t out;
    uint32_t neg;
} tfa_gate;

typedef struct tfa_ctx tfa_ctx;

typedef struct tfa_stats {
    uint64_t pbs_count, ks_count, gate_count, stage_count;
    double pbs_kernel_ms, ks_kernel_ms; /* accumulated device time, only while profiling is on */
    uint64_t pbs_launches, ks_launches;
} tfa_stats;

void tfa_default_params(tfa_params *p);

/* ServerKey construction, device side (replaces the ServerKey half of gen_keys(), main.rs:64). */
int tfa_ctx_create(const tfa_params *p, int hip_device, tfa_ctx **out);
int tfa_ctx_destroy(tfa_ctx *ctx);
/* bsk: standard-domain GGSW list [n][l][k+1][k+1][N] u32; ksk: [kN][l_ks][n+1] u32.  The device
 * converts the BSK to its Fourier layout. */
int tfa_keys_upload(tfa_ctx *ctx, const uint32_t *bsk, const uint32_t *ksk);

/* Ciphertext pool: n_slots LWE ciphertexts of n+1 words (host view), resident in HBM. */
int tfa_pool_reserve(tfa_ctx *ctx, uint32_t n_slots);
int tfa_pool_write(tfa_ctx *ctx, const uint32_t *slots, uint32_t count, const uint32_t *lwe /*[count][n+1]*/);
int tfa_pool_read(tfa_ctx *ctx, const uint32_t *slots, uint32_t count, uint32_t *lwe /*[count][n+1]*/);
/* The same read into DEVICE memory of the context's GPU: `d_out` receives [count][n+1] dense rows, complete on return, and
 * can be handed to a collective as it is (the gather of output ciphertexts across GPUs; the reference's CTR mode leaves
 * that step to the caller, /root/reference/modes/src/ctr.rs:8-10).  negate[i] != 0 writes the LWE negation of slot i
 * (ServerKey::not on the way out); negate may be NULL. */
int tfa_pool_read_device(tfa_ctx *ctx, const uint32_t *slots, const uint8_t *negate, uint32_t count, void *d_out);
/* Device address/stride (in u32 words) of the pool, for zero-copy collectives. */
int tfa_pool_device_ptr(tfa_ctx *ctx, void **dptr, uint32_t *stride_words, uint32_t *n_slots);

/* The per-stage batch submitter: replaces the rayon pool of FHEByte::sub_byte
 * (/root/reference/base/src/primitive.rs:250-265, Runnable::run boolean_tree.rs:644-654) and the
 * per-bit loops of FHEByte::{xor,and,or}_in_place / mul_x_gf2_in_place (primitive.rs:96-133,
 * 349-362).  Asynchronous and ordered on the context's stream.  Every gate reads the pool as it was
 * before the stage: a bootstrapped gate may write a slot that it or any other gate of the stage reads
 * (the *_assign forms); a NOT/COPY may write its own input slot.  Rejected with TFA_EINVAL, nothing launched: two
 * gates of a stage writing the same slot, and a NOT/COPY writing a slot that another NOT/COPY of the stage reads. */
int tfa_stage_submit(tfa_ctx *ctx, const tfa_gate *gates, uint32_t count);
int tfa_sync(tfa_ctx *ctx);

/* tfa_stage_submit for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc once set-up is
 * over).  No counterpart in the reference.  Same gates, same results, same ordering on the context's stream,
 * but the stage's job lists and big-LWE intermediates live in a static device array of fixed size instead of the
 * context's grow-only buffers, and the call performs no hipMalloc / hipFree / hipHostMalloc.  It is not a graph-capture
 * entry point: it copies the translated job lists from a pinned host buffer on the context's own stream and records an
 * event there.  There is one such host buffer, so a call first waits until the previous call's copy has executed (host and
 * device run at most one call apart; tfa_stage_submit has a ring of four).  Limits: at most
 * TFA_NOALLOC_MAX_GATES gates and TFA_NOALLOC_MAX_PBS bootstraps (a MUX counts two) per call; larger stages return
 * TFA_EINVAL.  Each GPU holds one copy of the array (TFA_NOALLOC_SCRATCH_BYTES) until the process ends; all contexts
 * on a GPU share it and it keeps the previous call's contents, so calls that overlap on one GPU (two contexts, two
 * threads) race: serialise them.  The pool must already hold every slot the gates name (tfa_pool_reserve). */
#define TFA_NOALLOC_MAX_GATES 4096
#define TFA_NOALLOC_MAX_PBS 4096
#define TFA_NOALLOC_SCRATCH_BYTES (32u << 20)
int tfa_stage_submit_noalloc(tfa_ctx *ctx, const tfa_gate *gates, uint32_t count);

/* Building blocks on host buffers (tfhe-rs core_crypto programmable bootstrap / keyswitch as
 * invoked inside every ServerKey gate).  `steps` < n stops the blind rotation early and returns
 * the raw accumulator [count][k+1][N] instead of the extracted LWE. */
int tfa_bootstrap_batch(tfa_ctx *ctx, const uint32_t *lwe_in /*[count][n+1]*/, uint32_t count, uint32_t *big_out /*[count][kN+1]*/);
int tfa_blind_rotate_batch(tfa_ctx *ctx, const uint32_t *lwe_in, uint32_t count, uint32_t steps, uint32_t *acc_out);
int tfa_keyswitch_batch(tfa_ctx *ctx, const uint32_t *big_in /*[count][kN+1]*/, uint32_t count, uint32_t *lwe_out /*[count][n+1]*/);

int tfa_set_profiling(tfa_ctx *ctx, int on); /* HIP events around each kernel on the context stream */
int tfa_get_stats(tfa_ctx *ctx, tfa_stats *out);
int tfa_reset_stats(tfa_ctx *ctx);
/* The timed launches behind tfa_stats, one record each, in submission order since the last tfa_reset_stats (only while
 * profiling is on): what the reference prints with Instant::now() around a mode call
 * (/root/reference/tfhe-aes/src/main.rs:154-163), resolved per stage.  kind 0 = the bootstraps of a stage (units =
 * bootstraps; mapping = bootstraps per workgroup of the kernel that ran them: 8 or 4 = one wave each, 1 or 2 = one
 * bootstrap per workgroup with that many workgroups per CU, 8 | tail << 8 = whole rounds of the 8-per-workgroup kernel
 * plus a tail), kind 1 = its keyswitches.  Synchronises the stream; writes min(capacity, *count) records, *count = all. */
typedef struct tfa_launch_record {
    uint32_t units, kind, mapping;
    float ms;
} tfa_launch_record;
int tfa_profile_launches(tfa_ctx *ctx, tfa_launch_record *out, uint32_t capacity, uint32_t *count);
const char *tfa_last_error(tfa_ctx *ctx);

/* Client side (ClientKey::{encrypt,decrypt} and the key-material half of gen_keys();
 * /root/reference/base/src/primitive.rs:63,71,85, main.rs:64).  Host CPU, as in the reference.
 * seed == 0 (production) keys the generator with 256 bits of OS entropy and returns TFA_ESTATE if the OS has none;
 * a non-zero seed gives reproducible, NON-SECRET streams (64 bits of entropy) for tests and benchmarks. */
int tfa_client_keygen(const tfa_params *p, uint64_t seed, uint32_t *lwe_sk /*[n]*/, uint32_t *glwe_sk /*[kN]*/,
                      uint32_t *bsk, uint32_t *ksk);
int tfa_client_encrypt(const tfa_params *p, const uint32_t *lwe_sk, uint64_t seed, const uint8_t *bits, uint32_t count,
                       uint32_t *lwe /*[count][n+1]*/);
int tfa_client_decrypt(const tfa_params *p, const uint32_t *lwe_sk, const uint32_t *lwe, uint32_t count, uint8_t *bits);
uint64_t tfa_bsk_words(const tfa_params *p);
uint64_t tfa_ksk_words(const tfa_params *p);

#ifdef __cplusplus
}
#endif
#endif
