// Engine behind the C ABI in include/tfhe_aes_hip.h: context, key upload, ciphertext pool and
// the per-stage batch submitter.  One context per GPU, one submitting thread per context.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/tfhe_aes_hip.h"
#include "ks_kernel.h"
#include "pbs_kernel.h"
#include "pbs_asm_kernel.h"
#include "pbs_lat_kernel.h"
#include "pbs_lat_asm_kernel.h"

using namespace tfa;

namespace {
constexpr size_t kAsmKeySlackBytes = 131072;  // one blind-rotation step of Fourier key
constexpr uint32_t kPoolStridePad = 4;   // pool rows padded to a multiple of 4 words (16 B)
constexpr size_t kMaxLaunchRecords = 1u << 20;  // tfa_profile_launches keeps at most this many records between resets
constexpr int kStagingRing = 4;

struct EventPair {
    hipEvent_t a, b;
    int kind;  // 0 = pbs, 1 = ks
    uint32_t units = 0, mapping = 0;  // bootstraps / keyswitches of the launch; bootstrap mapping (tfa_launch_record)
};
}  // namespace

struct tfa_ctx {
    tfa_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // keys
    c64 *d_bsk_f = nullptr;
    uint32_t *d_ksk = nullptr;
    c64 *d_tw = nullptr;
    int8_t *d_ksk8 = nullptr;  // keyswitching key as signed byte planes for the int8 matrix cores (default parameter shape only)
    c64 *d_bsk_ls = nullptr;   // TFA_LANESUM: copy of the Fourier key in the lane-local-sum layout of the hand-scheduled wide loops
    c64 *d_bsk_lat = nullptr;  // second copy of the Fourier key in the latency kernel's layout
    c64 *d_tw_lat = nullptr;
    uint32_t ksk_stride = 0;
    bool have_keys = false;
    // pool
    uint32_t *d_pool = nullptr;
    uint32_t pool_slots = 0, pool_stride = 0;
    // per-stage scratch (grow-only)
    uint32_t *d_big = nullptr;
    uint32_t big_cap = 0, big_stride = 0;
    unsigned char *d_jobs = nullptr;
    size_t jobs_cap = 0;
    unsigned char *h_stage[kStagingRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t h_stage_cap[kStagingRing] = {0, 0, 0, 0};
    hipEvent_t h_stage_ev[kStagingRing] = {nullptr, nullptr, nullptr, nullptr};
    int ring = 0;
    uint32_t *d_io = nullptr;  // staging for pool read/write and the host-buffer entry points
    size_t io_cap = 0;
    // stats
    tfa_stats stats{};
    bool profiling = false;
    uint32_t n_cus = 256;
    bool ks_valu = false;  // TFA_KS_VALU=1: keep the keyswitch on the vector ALUs (the generic-parameter path; for comparison)
    bool trace = false;   // TFA_TRACE=1: one stderr line per submitted stage
    uint32_t ks_mfma_min = 1;    // fewest gates of a stage that go to the matrix-core keyswitch (TFA_KS_MFMA_MIN, experiments)
    int ks_split = 0;     // TFA_KS_SPLIT=n (1..24): pins the matrix-core keyswitch's split of the input coefficients (tests)
    bool lat_asm = true;  // TFA_LAT_ASM=0: the one-bootstrap-per-workgroup mapping runs the compiler-scheduled kernel
    bool pbs_asm = true;  // TFA_PBS_ASM=0: the one-bootstrap-per-wave mappings run the compiler-scheduled kernels instead of the hand-scheduled loop
    // time of one round of workgroups over all CUs per mapping, microseconds: [0] 8 bootstraps per workgroup, [1] 4 per
    // workgroup, [2] one bootstrap per workgroup and one workgroup per CU, [3] the same with two workgroups per CU.
    // Measured on this device after the key upload (calibrate_rounds); the defaults are an MI355X at 2.4 GHz.
    double round_us[4] = {9800, 6600, 2400, 4100};
    int force_waves = 0;  // 0 = choose per launch; 1 / 2 (latency kernel, workgroups per CU), 4 or 8 pins the kernel variant (TFA_PBS_WAVES, for experiments)
    std::vector<EventPair> pending;
    std::vector<tfa_launch_record> launches;  // one record per timed bootstrap / keyswitch launch since the last reset (profiling only)
    // per-stage validation marks, sized with the pool: slot_mark[s] = stamp of the stage that last wrote s,
    // lin_reader[s] = (stamp << 32) | index + 1 of the NOT/COPY gate reading s (all ones: several)
    std::vector<uint32_t> slot_mark;
    std::vector<uint64_t> lin_reader;
    uint32_t mark_stamp = 0;
    // tfa_stage_submit_noalloc: pinned translation buffer of fixed size and the event guarding its reuse
    unsigned char *h_noalloc = nullptr;
    hipEvent_t h_noalloc_ev = nullptr;
};

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
            return TFA_EHIP;                                                                                   \
        }                                                                                                      \
    } while (0)

static int fail(tfa_ctx *ctx, int code, const std::string &msg) {
    ctx->err = msg;
    return code;
}
static int noalloc_setup(tfa_ctx *ctx);
static int calibrate_rounds(tfa_ctx *ctx);

static int ensure_io(tfa_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->io_cap) return TFA_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_io) HIPCHK(ctx, hipFree(ctx->d_io));
    ctx->d_io = nullptr;
    ctx->io_cap = 0;
    size_t cap = bytes + bytes / 4;
    HIPCHK(ctx, hipMalloc(&ctx->d_io, cap));
    ctx->io_cap = cap;
    return TFA_OK;
}
static int ensure_big(tfa_ctx *ctx, uint32_t rows) {
    if (rows <= ctx->big_cap) return TFA_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_big) HIPCHK(ctx, hipFree(ctx->d_big));
    ctx->d_big = nullptr;
    ctx->big_cap = 0;
    uint32_t cap = rows + rows / 4 + 64;
    HIPCHK(ctx, hipMalloc(&ctx->d_big, (size_t)cap * ctx->big_stride * sizeof(uint32_t)));
    ctx->big_cap = cap;
    return TFA_OK;
}
static int ensure_jobs(tfa_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->jobs_cap) return TFA_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_jobs) HIPCHK(ctx, hipFree(ctx->d_jobs));
    ctx->d_jobs = nullptr;
    ctx->jobs_cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&ctx->d_jobs, cap));
    ctx->jobs_cap = cap;
    return TFA_OK;
}
// pinned staging ring slot, safe to overwrite once its previous upload has executed
static int staging(tfa_ctx *ctx, size_t bytes, unsigned char **out) {
    int s = ctx->ring;
    ctx->ring = (ctx->ring + 1) % kStagingRing;
    if (ctx->h_stage_ev[s]) HIPCHK(ctx, hipEventSynchronize(ctx->h_stage_ev[s]));
    if (bytes > ctx->h_stage_cap[s]) {
        if (ctx->h_stage[s]) HIPCHK(ctx, hipHostFree(ctx->h_stage[s]));
        ctx->h_stage[s] = nullptr;
        ctx->h_stage_cap[s] = 0;
        size_t cap = bytes + bytes / 4 + 4096;
        HIPCHK(ctx, hipHostMalloc(&ctx->h_stage[s], cap, hipHostMallocDefault));
        ctx->h_stage_cap[s] = cap;
    }
    if (!ctx->h_stage_ev[s]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->h_stage_ev[s], hipEventDisableTiming));
    *out = ctx->h_stage[s];
    return s;
}

static int drain_events(tfa_ctx *ctx) {
    for (auto &ep : ctx->pending) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(ep.b));
        HIPCHK(ctx, hipEventElapsedTime(&ms, ep.a, ep.b));
        if (ep.kind == 0) {
            ctx->stats.pbs_kernel_ms += ms;
            ctx->stats.pbs_launches++;
        } else {
            ctx->stats.ks_kernel_ms += ms;
            ctx->stats.ks_launches++;
        }
        if (ctx->launches.size() < kMaxLaunchRecords) ctx->launches.push_back(tfa_launch_record{ep.units, (uint32_t)ep.kind, ep.mapping, ms});
        hipEventDestroy(ep.a);
        hipEventDestroy(ep.b);
    }
    ctx->pending.clear();
    return TFA_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" void tfa_default_params(tfa_params *p) {
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
extern "C" uint64_t tfa_bsk_words(const tfa_params *p) {
    uint64_t k1 = p->glwe_dimension + 1;
    return (uint64_t)p->lwe_dimension * p->pbs_level * k1 * k1 * p->polynomial_size;
}
extern "C" uint64_t tfa_ksk_words(const tfa_params *p) {
    return (uint64_t)p->glwe_dimension * p->polynomial_size * p->ks_level * (p->lwe_dimension + 1);
}

extern "C" int tfa_ctx_create(const tfa_params *p, int hip_device, tfa_ctx **out) {
    if (!p || !out) return TFA_EINVAL;
    *out = nullptr;
    if (p->glwe_dimension != 3 || p->polynomial_size != 512 || p->pbs_base_log != 10 || p->pbs_level != 2) return TFA_EINVAL;
    // the vector keyswitch maps its n + 1 output columns onto 13 waves x 62 lanes = 806 columns of one workgroup
    if (p->lwe_dimension < 1 || p->lwe_dimension + 1 > 13 * kKsLanesUsed || p->ks_level < 1 || p->ks_level > kKsMaxLevels ||
        p->ks_base_log * p->ks_level > 31 || p->ks_base_log < 1)
        return TFA_EINVAL;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || hip_device < 0 || hip_device >= n_dev) return TFA_EHIP;
    tfa_ctx *ctx = new (std::nothrow) tfa_ctx();
    if (!ctx) return TFA_ENOMEM;
    ctx->p = *p;
    ctx->device = hip_device;
    ctx->pool_stride = (p->lwe_dimension + 1 + kPoolStridePad - 1) / kPoolStridePad * kPoolStridePad;
    ctx->ksk_stride = ctx->pool_stride;
    ctx->big_stride = (p->glwe_dimension * p->polynomial_size + 1 + 3) / 4 * 4;
    *out = ctx;  // handed out even on failure below so the caller can read the message and destroy it
    HIPCHK(ctx, hipSetDevice(hip_device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, hip_device));
    if (prop.multiProcessorCount > 0) ctx->n_cus = (uint32_t)prop.multiProcessorCount;
    if (const char *t = getenv("TFA_TRACE")) ctx->trace = atoi(t) != 0;
    if (const char *t = getenv("TFA_KS_VALU")) ctx->ks_valu = atoi(t) != 0;
    if (const char *t = getenv("TFA_PBS_ASM")) ctx->pbs_asm = atoi(t) != 0;
    if (const char *t = getenv("TFA_LAT_ASM")) ctx->lat_asm = atoi(t) != 0;
    if (const char *t = getenv("TFA_KS_MFMA_MIN")) ctx->ks_mfma_min = (uint32_t)atoi(t);
    if (const char *t = getenv("TFA_KS_SPLIT")) ctx->ks_split = atoi(t) < 1 ? 0 : (atoi(t) > 24 ? 24 : atoi(t));
    if (const char *w = getenv("TFA_PBS_WAVES")) ctx->force_waves = (atoi(w) == 8 || atoi(w) == 4 || atoi(w) == 2 || atoi(w) == 1) ? atoi(w) : 0;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    // per-lane twiddles e^{i pi t (1 + 4 r)/512}, [r][t]
    std::vector<c64> tw(256);
    for (int r = 0; r < 16; r++)
        for (int t = 0; t < 16; t++) {
            long double a = 3.14159265358979323846264338327950288L * (long double)(t * (1 + 4 * r)) / 512.0L;
            tw[r * 16 + t] = c64{(double)cosl(a), (double)sinl(a)};
        }
    HIPCHK(ctx, hipMalloc(&ctx->d_tw, sizeof(c64) * 256));
    HIPCHK(ctx, hipMemcpy(ctx->d_tw, tw.data(), sizeof(c64) * 256, hipMemcpyHostToDevice));
    // latency kernel: twist e^{i pi n/512} then e^{2 pi i k/256}
    std::vector<c64> twl(512);
    for (int k = 0; k < 256; k++) {
        const long double pi = 3.14159265358979323846264338327950288L;
        twl[k] = c64{(double)cosl(pi * k / 512.0L), (double)sinl(pi * k / 512.0L)};
        twl[256 + k] = c64{(double)cosl(2.0L * pi * k / 256.0L), (double)sinl(2.0L * pi * k / 256.0L)};
    }
    HIPCHK(ctx, hipMalloc(&ctx->d_tw_lat, sizeof(c64) * 512));
    HIPCHK(ctx, hipMemcpy(ctx->d_tw_lat, twl.data(), sizeof(c64) * 512, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_lat_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LatGeom<1>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_lat_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LatGeom<2>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_lat_asm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LatAsmGeom<1>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_lat_asm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LatAsmGeom<2>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bsk_convert_lat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLatConvertLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, PbsGeom<4>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, PbsGeom<8>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_asm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, AsmGeom<8>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_asm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, AsmGeom<8>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_asm_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, AsmGeom<4>::kLdsBytes));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bsk_convert_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kConvertLdsBytes));
    return noalloc_setup(ctx);
}

extern "C" int tfa_ctx_destroy(tfa_ctx *ctx) {
    if (!ctx) return TFA_EINVAL;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto &ep : ctx->pending) {
        hipEventDestroy(ep.a);
        hipEventDestroy(ep.b);
    }
    hipFree(ctx->d_bsk_f);
    hipFree(ctx->d_ksk);
    hipFree(ctx->d_tw);
    hipFree(ctx->d_ksk8);
    hipFree(ctx->d_bsk_lat);
    hipFree(ctx->d_bsk_ls);
    hipFree(ctx->d_tw_lat);
    hipFree(ctx->d_pool);
    hipFree(ctx->d_big);
    hipFree(ctx->d_jobs);
    hipFree(ctx->d_io);
    for (int s = 0; s < kStagingRing; s++) {
        if (ctx->h_stage[s]) hipHostFree(ctx->h_stage[s]);
        if (ctx->h_stage_ev[s]) hipEventDestroy(ctx->h_stage_ev[s]);
    }
    if (ctx->h_noalloc) hipHostFree(ctx->h_noalloc);
    if (ctx->h_noalloc_ev) hipEventDestroy(ctx->h_noalloc_ev);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return TFA_OK;
}

extern "C" const char *tfa_last_error(tfa_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int tfa_keys_upload(tfa_ctx *ctx, const uint32_t *bsk, const uint32_t *ksk) {
    if (!ctx || !bsk || !ksk) return TFA_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const tfa_params &p = ctx->p;
    const uint64_t bsk_words = tfa_bsk_words(&p);
    const uint32_t n = p.lwe_dimension, kN_ = p.glwe_dimension * p.polynomial_size;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // N/2 complex per polynomial, plus one step of slack: the hand-scheduled loop prefetches key chunks without an end-of-key test
    if (!ctx->d_bsk_f) {
        HIPCHK(ctx, hipMalloc(&ctx->d_bsk_f, bsk_words / 2 * sizeof(c64) + kAsmKeySlackBytes));
        HIPCHK(ctx, hipMemset(reinterpret_cast<unsigned char *>(ctx->d_bsk_f) + bsk_words / 2 * sizeof(c64), 0, kAsmKeySlackBytes));
    }
    if (TFA_LANESUM && !ctx->d_bsk_ls) {
        HIPCHK(ctx, hipMalloc(&ctx->d_bsk_ls, bsk_words / 2 * sizeof(c64) + kAsmKeySlackBytes));
        HIPCHK(ctx, hipMemset(reinterpret_cast<unsigned char *>(ctx->d_bsk_ls) + bsk_words / 2 * sizeof(c64), 0, kAsmKeySlackBytes));
    }
    if (!ctx->d_bsk_lat) {  // plus one step of slack: the hand-scheduled loop loads the key one step ahead without an end-of-key test
        HIPCHK(ctx, hipMalloc(&ctx->d_bsk_lat, bsk_words / 2 * sizeof(c64) + kLatKeySlackBytes));
        HIPCHK(ctx, hipMemset(reinterpret_cast<unsigned char *>(ctx->d_bsk_lat) + bsk_words / 2 * sizeof(c64), 0, kLatKeySlackBytes));
    }
    if (!ctx->d_ksk) HIPCHK(ctx, hipMalloc(&ctx->d_ksk, (size_t)kN_ * p.ks_level * ctx->ksk_stride * sizeof(uint32_t)));
    // KSK rows padded to the pool stride
    HIPCHK(ctx, hipMemset(ctx->d_ksk, 0, (size_t)kN_ * p.ks_level * ctx->ksk_stride * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemcpy2D(ctx->d_ksk, ctx->ksk_stride * sizeof(uint32_t), ksk, (n + 1) * sizeof(uint32_t), (n + 1) * sizeof(uint32_t),
                            (size_t)kN_ * p.ks_level, hipMemcpyHostToDevice));
    // byte planes of the KSK for the matrix-core keyswitch
    if (p.ks_level == 5 && p.ks_base_log <= 8 && kN_ % kKsmCoefs == 0) {
        const uint32_t n_kb = kN_ / kKsmCoefs, n_tiles = (n + 1 + kKsmCols - 1) / kKsmCols;
        const size_t frags = (size_t)n_tiles * n_kb * 5 * 4 * 4 * 64;
        if (!ctx->d_ksk8) HIPCHK(ctx, hipMalloc(&ctx->d_ksk8, frags * 16));
        hipLaunchKernelGGL(ksk_to_planes_kernel<5>, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ksk, ctx->ksk_stride, n, n_kb,
                           n_tiles, ctx->d_ksk8);
        HIPCHK(ctx, hipGetLastError());
    }
    // BSK: upload the standard domain, transform on the device
    uint32_t *d_std = nullptr;
    HIPCHK(ctx, hipMalloc(&d_std, bsk_words * sizeof(uint32_t)));
    hipError_t e = hipMemcpy(d_std, bsk, bsk_words * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t rows = n * p.pbs_level * (p.glwe_dimension + 1);
        hipLaunchKernelGGL(bsk_convert_kernel, dim3((rows + kConvertWaves - 1) / kConvertWaves), dim3(256), kConvertLdsBytes, ctx->stream, d_std,
                           ctx->d_bsk_f, ctx->d_tw, rows);
        e = hipGetLastError();
        if (e == hipSuccess && TFA_LANESUM) {
            const size_t total = (size_t)n * 128 * 64;
            hipLaunchKernelGGL(bsk_lanesum_layout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_bsk_f, ctx->d_bsk_ls, n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bsk_convert_lat_kernel, dim3(rows), dim3(256), kLatConvertLdsBytes, ctx->stream, d_std, ctx->d_bsk_lat, ctx->d_tw_lat, rows);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(d_std);
    if (e != hipSuccess) return fail(ctx, TFA_EHIP, std::string("bsk upload/convert: ") + hipGetErrorString(e));
    ctx->have_keys = true;
    return calibrate_rounds(ctx);
}

// ---------------------------------------------------------------------------------------------
extern "C" int tfa_pool_reserve(tfa_ctx *ctx, uint32_t n_slots) {
    if (!ctx || n_slots == 0) return TFA_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_slots <= ctx->pool_slots) return TFA_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t *np_ = nullptr;
    size_t bytes = (size_t)n_slots * ctx->pool_stride * sizeof(uint32_t);
    if (hipMalloc(&np_, bytes) != hipSuccess) return fail(ctx, TFA_ENOMEM, "pool allocation failed");
    hipError_t e = hipMemset(np_, 0, bytes);
    if (e == hipSuccess && ctx->d_pool)
        e = hipMemcpy(np_, ctx->d_pool, (size_t)ctx->pool_slots * ctx->pool_stride * sizeof(uint32_t), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
        hipFree(np_);
        return fail(ctx, TFA_EHIP, std::string("pool grow: ") + hipGetErrorString(e));
    }
    if (ctx->d_pool) hipFree(ctx->d_pool);
    ctx->d_pool = np_;
    ctx->pool_slots = n_slots;
    ctx->slot_mark.resize(n_slots, 0u);
    ctx->lin_reader.resize(n_slots, 0u);
    return TFA_OK;
}

static int check_slots(tfa_ctx *ctx, const uint32_t *slots, uint32_t count) {
    for (uint32_t i = 0; i < count; i++)
        if (slots[i] >= ctx->pool_slots) return fail(ctx, TFA_EINVAL, "slot index out of range");
    return TFA_OK;
}

extern "C" int tfa_pool_write(tfa_ctx *ctx, const uint32_t *slots, uint32_t count, const uint32_t *lwe) {
    if (!ctx || !slots || !lwe) return TFA_EINVAL;
    if (count == 0) return TFA_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, slots, count);
    if (rc) return rc;
    const uint32_t words = ctx->p.lwe_dimension + 1;
    size_t data_bytes = (size_t)count * words * sizeof(uint32_t), slot_bytes = (size_t)count * sizeof(uint32_t);
    size_t slot_off = (data_bytes + 15) / 16 * 16;
    rc = ensure_io(ctx, slot_off + slot_bytes);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(ctx->d_io);
    HIPCHK(ctx, hipMemcpyAsync(base, lwe, data_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(base + slot_off, slots, slot_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pool_scatter_kernel, dim3(count), dim3(256), 0, ctx->stream, ctx->d_pool, ctx->pool_stride,
                       reinterpret_cast<const uint32_t *>(base + slot_off), reinterpret_cast<const uint32_t *>(base), words);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host buffers may be reused on return
    return TFA_OK;
}

extern "C" int tfa_pool_read(tfa_ctx *ctx, const uint32_t *slots, uint32_t count, uint32_t *lwe) {
    if (!ctx || !slots || !lwe) return TFA_EINVAL;
    if (count == 0) return TFA_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, slots, count);
    if (rc) return rc;
    const uint32_t words = ctx->p.lwe_dimension + 1;
    size_t data_bytes = (size_t)count * words * sizeof(uint32_t), slot_bytes = (size_t)count * sizeof(uint32_t);
    size_t slot_off = (data_bytes + 15) / 16 * 16;
    rc = ensure_io(ctx, slot_off + slot_bytes);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(ctx->d_io);
    HIPCHK(ctx, hipMemcpyAsync(base + slot_off, slots, slot_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pool_gather_kernel, dim3(count), dim3(256), 0, ctx->stream, ctx->d_pool, ctx->pool_stride,
                       reinterpret_cast<const uint32_t *>(base + slot_off), reinterpret_cast<uint32_t *>(base), words);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(lwe, base, data_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TFA_OK;
}

extern "C" int tfa_pool_read_device(tfa_ctx *ctx, const uint32_t *slots, const uint8_t *negate, uint32_t count, void *d_out) {
    if (!ctx || !slots || !d_out) return TFA_EINVAL;
    if (count == 0) return TFA_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_slots(ctx, slots, count);
    if (rc) return rc;
    if (ctx->pool_slots > 0x80000000u) return fail(ctx, TFA_EINVAL, "pool too large for tagged slot indices");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_out) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != ctx->device) {
        (void)hipGetLastError();
        return fail(ctx, TFA_EINVAL, "tfa_pool_read_device: destination is not device memory of the context's GPU");
    }
    const size_t slot_bytes = (size_t)count * sizeof(uint32_t);
    rc = ensure_io(ctx, slot_bytes);
    if (rc) return rc;
    std::vector<uint32_t> tagged(slots, slots + count);
    if (negate)
        for (uint32_t i = 0; i < count; i++)
            if (negate[i]) tagged[i] |= 0x80000000u;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_io, tagged.data(), slot_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pool_export_kernel, dim3(count), dim3(256), 0, ctx->stream, ctx->d_pool, ctx->pool_stride, ctx->d_io,
                       static_cast<uint32_t *>(d_out), ctx->p.lwe_dimension + 1);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the rows are complete (and `tagged` may go) on return
    return TFA_OK;
}

extern "C" int tfa_pool_device_ptr(tfa_ctx *ctx, void **dptr, uint32_t *stride_words, uint32_t *n_slots) {
    if (!ctx) return TFA_EINVAL;
    if (dptr) *dptr = ctx->d_pool;
    if (stride_words) *stride_words = ctx->pool_stride;
    if (n_slots) *n_slots = ctx->pool_slots;
    return TFA_OK;
}

// ---------------------------------------------------------------------------------------------
// One launch of a bootstrap mapping: 8 / 4 = bootstraps per workgroup (one wave each), 1 / 2 = one bootstrap spread over a
// workgroup's four waves with that many workgroups per CU.
static void launch_variant(tfa_ctx *ctx, int variant, const uint32_t *in_base, uint32_t in_stride, const PbsJob *jobs, uint32_t count, uint32_t steps,
                           uint32_t *big, uint32_t *acc, bool calibration = false) {
    const c64 *asm_key = TFA_LANESUM ? ctx->d_bsk_ls : ctx->d_bsk_f;  // the hand-scheduled wide loops' key layout
    if (variant == 8 && ctx->pbs_asm && calibration)
        hipLaunchKernelGGL(pbs_asm_kernel<1>, dim3((count + 7) / 8), dim3(512), AsmGeom<8>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           asm_key, ctx->d_tw, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 8 && ctx->pbs_asm)
        hipLaunchKernelGGL(pbs_asm_kernel<0>, dim3((count + 7) / 8), dim3(512), AsmGeom<8>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           asm_key, ctx->d_tw, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 8)
        hipLaunchKernelGGL(pbs_kernel<8>, dim3((count + 7) / 8), dim3(512), PbsGeom<8>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_f, ctx->d_tw, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 4 && ctx->pbs_asm)
        hipLaunchKernelGGL((pbs_asm_kernel<0, 4>), dim3((count + 3) / 4), dim3(256), AsmGeom<4>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           asm_key, ctx->d_tw, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 4)
        hipLaunchKernelGGL(pbs_kernel<4>, dim3((count + 3) / 4), dim3(256), PbsGeom<4>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_f, ctx->d_tw, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 2 && ctx->lat_asm)
        hipLaunchKernelGGL(pbs_lat_asm_kernel<2>, dim3(count), dim3(256), LatAsmGeom<2>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_lat, ctx->d_tw_lat, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (variant == 2)
        hipLaunchKernelGGL(pbs_lat_kernel<2>, dim3(count), dim3(256), LatGeom<2>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_lat, ctx->d_tw_lat, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else if (ctx->lat_asm)
        hipLaunchKernelGGL(pbs_lat_asm_kernel<1>, dim3(count), dim3(256), LatAsmGeom<1>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_lat, ctx->d_tw_lat, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
    else
        hipLaunchKernelGGL(pbs_lat_kernel<1>, dim3(count), dim3(256), LatGeom<1>::kLdsBytes, ctx->stream, in_base, in_stride, jobs, count,
                           ctx->d_bsk_lat, ctx->d_tw_lat, ctx->p.lwe_dimension, steps, big, ctx->big_stride, acc);
}

// Round times of the four mappings on this device: each mapping runs one full round of workgroups for two step counts
// (a quarter and three quarters of the blind rotation's length, so both lie inside what the kernels execute), and the line
// through the two timings gives the time of a complete blind rotation (slope) plus the launch's fixed part.  A table belongs
// to one (device, LWE dimension, kernel selection) and is kept for the life of the process; a context with another key
// length or with TFA_PBS_ASM=0 / TFA_LAT_ASM=0 measures its own.  The table is a scheduling heuristic: when the timings make no
// sense (non-positive slope) or the rotation is too short to time, the built-in table stays and the key upload succeeds.
// TFA_NO_CALIBRATE=1 keeps the built-in table.
namespace {
struct RoundTable {
    int device;
    uint32_t lwe_dimension;
    bool pbs_asm, lat_asm;
    double us[4];
};
std::vector<RoundTable> g_round_tables;
std::mutex g_round_tables_mutex;
}  // namespace

static int calibrate_rounds(tfa_ctx *ctx) {
    if (getenv("TFA_NO_CALIBRATE")) return TFA_OK;
    const uint32_t n = ctx->p.lwe_dimension;
    if (n < 64) {  // too few steps for a slope that stands out of the launch overhead
        if (ctx->trace) fprintf(stderr, "tfa round times: lwe_dimension %u too short to calibrate, built-in table kept\n", n);
        return TFA_OK;
    }
    std::lock_guard<std::mutex> lock(g_round_tables_mutex);
    const RoundTable *found = nullptr;
    for (const auto &t : g_round_tables)
        if (t.device == ctx->device && t.lwe_dimension == n && t.pbs_asm == ctx->pbs_asm && t.lat_asm == ctx->lat_asm) found = &t;
    if (!found) {
        RoundTable tab{ctx->device, n, ctx->pbs_asm, ctx->lat_asm, {0, 0, 0, 0}};
        const uint32_t cus = ctx->n_cus, max_jobs = 8 * cus;
        uint32_t *d_in = nullptr, *d_big = nullptr;
        PbsJob *d_jobs = nullptr;
        HIPCHK(ctx, hipMalloc(&d_in, 2 * (size_t)ctx->pool_stride * sizeof(uint32_t)));
        hipError_t e = hipMalloc(&d_big, (size_t)max_jobs * ctx->big_stride * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&d_jobs, max_jobs * sizeof(PbsJob));
        std::vector<PbsJob> jobs(max_jobs, PbsJob{0, 1, 1, 1, 1u << 29, {0, 0, 0}});
        std::vector<uint32_t> in(2 * (size_t)ctx->pool_stride);
        for (size_t i = 0; i < in.size(); i++) in[i] = (uint32_t)(i * 2654435761u);  // any mask: the time does not depend on the data
        if (e == hipSuccess) e = hipMemcpy(d_in, in.data(), in.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(PbsJob), hipMemcpyHostToDevice);
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        for (auto &x : ev)
            if (e == hipSuccess) e = hipEventCreate(&x);
        const int variants[4] = {8, 4, 1, 2};
        const uint32_t widths[4] = {8 * cus, 4 * cus, cus, 2 * cus}, s0 = n / 4, s1 = n - n / 4;
        bool sane = true;
        // the shader clock takes a few milliseconds of load to come up from idle: one complete round first
        if (e == hipSuccess) launch_variant(ctx, 8, d_in, ctx->pool_stride, d_jobs, widths[0], n, d_big, nullptr, true);
        for (int v = 0; v < 4 && e == hipSuccess; v++) {
            launch_variant(ctx, variants[v], d_in, ctx->pool_stride, d_jobs, widths[v], s0, d_big, nullptr, true);  // warm-up: code and key in the caches
            hipEventRecord(ev[0], ctx->stream);
            launch_variant(ctx, variants[v], d_in, ctx->pool_stride, d_jobs, widths[v], s0, d_big, nullptr, true);
            hipEventRecord(ev[1], ctx->stream);
            launch_variant(ctx, variants[v], d_in, ctx->pool_stride, d_jobs, widths[v], s1, d_big, nullptr, true);
            hipEventRecord(ev[2], ctx->stream);
            e = hipEventSynchronize(ev[2]);
            if (e == hipSuccess) e = hipGetLastError();
            float ms0 = 0, ms1 = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms0, ev[0], ev[1]);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms1, ev[1], ev[2]);
            const double per_step = (ms1 - ms0) / (double)(s1 - s0), fixed = ms0 - per_step * s0;
            tab.us[v] = 1000.0 * (per_step * n + (fixed > 0 ? fixed : 0));
            if (!(per_step > 0)) sane = false;
        }
        for (auto &x : ev)
            if (x) hipEventDestroy(x);
        hipFree(d_in);
        hipFree(d_big);
        hipFree(d_jobs);
        if (e != hipSuccess) return fail(ctx, TFA_EHIP, std::string("round-time calibration: ") + hipGetErrorString(e));
        if (!sane) {  // a timing anomaly must not fail the key upload: keep the defaults for this context, measure again next time
            if (ctx->trace) fprintf(stderr, "tfa round times: calibration gave a non-positive slope, built-in table kept\n");
            return TFA_OK;
        }
        g_round_tables.push_back(tab);
        found = &g_round_tables.back();
    }
    for (int v = 0; v < 4; v++) ctx->round_us[v] = found->us[v];
    if (ctx->trace)
        fprintf(stderr, "tfa round times (us): 8/wg %.0f, 4/wg %.0f, 1 wg/CU %.0f, 2 wg/CU %.0f\n", found->us[0], found->us[1], found->us[2], found->us[3]);
    return TFA_OK;
}

static int launch_pbs(tfa_ctx *ctx, const uint32_t *in_base, uint32_t in_stride, const PbsJob *d_jobs, uint32_t n_jobs, uint32_t steps,
                      uint32_t *acc_out, uint32_t *big_base, bool timed = true) {
    if (n_jobs == 0) return TFA_OK;
    EventPair ep{nullptr, nullptr, 0};
    const bool profiling = ctx->profiling && timed;
    if (profiling) {
        HIPCHK(ctx, hipEventCreate(&ep.a));
        HIPCHK(ctx, hipEventCreate(&ep.b));
        HIPCHK(ctx, hipEventRecord(ep.a, ctx->stream));
    }
    // Four mappings, picked by which finishes soonest on the device's CUs, from the round times in ctx->round_us.
    // A batch that fills whole rounds of the 8-per-workgroup kernel and leaves a short tail runs the tail separately.
    const uint64_t cus = ctx->n_cus;
    const double r8 = ctx->round_us[0], r4 = ctx->round_us[1], rl1 = ctx->round_us[2], rl2 = ctx->round_us[3];
    auto best_single = [&](uint64_t P, int &variant) -> double {
        const double t8 = (double)((P + 8 * cus - 1) / (8 * cus)) * r8, t4 = (double)((P + 4 * cus - 1) / (4 * cus)) * r4;
        const uint64_t lat_rest = P % (2 * cus);
        const double t_lat = (double)(P / (2 * cus)) * rl2 + (lat_rest == 0 ? 0.0 : lat_rest <= cus ? rl1 : rl2);
        variant = t_lat <= t4 && t_lat <= t8 ? (P > cus ? 2 : 1) : (t8 < t4 ? 8 : 4);
        return variant == 8 ? t8 : variant == 4 ? t4 : t_lat;
    };
    auto launch = [&](int variant, uint32_t first, uint32_t count) {
        launch_variant(ctx, variant, in_base, in_stride, d_jobs + first, count, steps, big_base + (size_t)first * ctx->big_stride,
                       acc_out ? acc_out + (size_t)first * kK1 * kN : nullptr);
    };
    int variant = 0, tail_variant = 0;
    const double t_single = best_single(n_jobs, variant);
    const uint32_t full = (uint32_t)((n_jobs / (8 * cus)) * (8 * cus));
    ep.units = n_jobs;
    if (ctx->force_waves) {
        ep.mapping = (uint32_t)ctx->force_waves;
        launch(ctx->force_waves, 0, n_jobs);
    } else if (full > 0 && full < n_jobs && (double)(full / (8 * cus)) * r8 + best_single(n_jobs - full, tail_variant) < t_single) {
        ep.mapping = 8u | ((uint32_t)tail_variant << 8);
        launch(8, 0, full);
        HIPCHK(ctx, hipGetLastError());
        launch(tail_variant, full, n_jobs - full);
    } else {
        ep.mapping = (uint32_t)variant;
        launch(variant, 0, n_jobs);
    }
    HIPCHK(ctx, hipGetLastError());
    if (profiling) {
        HIPCHK(ctx, hipEventRecord(ep.b, ctx->stream));
        ctx->pending.push_back(ep);
    }
    ctx->stats.pbs_count += n_jobs;
    return TFA_OK;
}

static int launch_ks(tfa_ctx *ctx, const KsJob *d_jobs, uint32_t n_jobs, uint32_t *out_base, uint32_t out_stride, const uint32_t *big_base,
                     bool timed = true) {
    if (n_jobs == 0) return TFA_OK;
    const tfa_params &p = ctx->p;
    EventPair ep{nullptr, nullptr, 1, n_jobs, 0};
    const bool profiling = ctx->profiling && timed;
    if (profiling) {
        HIPCHK(ctx, hipEventCreate(&ep.a));
        HIPCHK(ctx, hipEventCreate(&ep.b));
        HIPCHK(ctx, hipEventRecord(ep.a, ctx->stream));
    }
    const uint32_t big_dim = p.glwe_dimension * p.polynomial_size;
    // With the default parameter set every stage goes to the int8 matrix cores (65 536 gates: 3.7 ms against 29 ms for the
    // vector kernel; 128 gates: 37 us against 111 us; 16 gates: 22 us against 39 us); the vector kernels serve other
    // parameter sets and TFA_KS_VALU=1.
    if (ctx->d_ksk8 && !ctx->ks_valu && n_jobs >= ctx->ks_mfma_min) {
        // A workgroup walks its blocks of 64 input coefficients one after the other, about 9 us each: alone on its CU it is
        // latency-bound.  While the gate tiles are fewer than about 1.5 workgroups per CU the blocks are split over several
        // workgroups that add their partial sums with integer atomics (measured best: 128 gates 24-fold, 37 us against 223 us
        // unsplit and 111 us for the vector kernel; 512 gates 8-fold, 59 us; 2 048 gates 2-fold, 147 us; from 3 840 gates unsplit).
        const uint32_t tiles = (n_jobs + kKsmGates - 1) / kKsmGates, col_tiles = (p.lwe_dimension + 1 + kKsmCols - 1) / kKsmCols;
        const uint32_t n_kb = big_dim / kKsmCoefs;
        uint32_t split = (3 * ctx->n_cus / 2 + tiles * col_tiles - 1) / (tiles * col_tiles);
        if (split > n_kb) split = n_kb;
        if (ctx->ks_split >= 1) split = (uint32_t)ctx->ks_split;
        if (split > 1) {
            hipLaunchKernelGGL(keyswitch_init_kernel, dim3(n_jobs), dim3(256), 0, ctx->stream, big_base, ctx->big_stride, d_jobs, n_jobs,
                               p.lwe_dimension, big_dim, out_base, out_stride);
            HIPCHK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(keyswitch_mfma_kernel<5>, dim3(tiles, col_tiles, split), dim3(256), 0,
                           ctx->stream, big_base, ctx->big_stride, d_jobs, n_jobs, ctx->d_ksk8, p.lwe_dimension, big_dim, p.ks_base_log, out_base,
                           out_stride);
        HIPCHK(ctx, hipGetLastError());
        if (profiling) {
            HIPCHK(ctx, hipEventRecord(ep.b, ctx->stream));
            ctx->pending.push_back(ep);
        }
        ctx->stats.ks_count += n_jobs;
        return TFA_OK;
    }
#define KS_LAUNCH(TG, UNROLL, SPLIT)                                                                                                         \
    hipLaunchKernelGGL((keyswitch_kernel<TG, UNROLL>), dim3((n_jobs + TG - 1) / TG, SPLIT), dim3(832), kKsChunk *p.ks_level *TG * sizeof(int32_t), \
                       ctx->stream, big_base, ctx->big_stride, d_jobs, n_jobs, ctx->d_ksk, ctx->ksk_stride, p.lwe_dimension, big_dim,            \
                       p.ks_base_log, p.ks_level, out_base, out_stride)
    if (n_jobs >= 32 * 256) {
        KS_LAUNCH(32, 2, 1);
    } else if (n_jobs >= 8 * 256) {
        KS_LAUNCH(8, 8, 1);
    } else {
        // narrow stage: small gate tiles, and the input coefficients split over enough workgroups to occupy the CUs
        const uint32_t tiles = (n_jobs + 1) / 2, chunks = (big_dim + kKsChunk - 1) / kKsChunk;
        uint32_t split = (ctx->n_cus + tiles - 1) / tiles;
        if (split > chunks) split = chunks;
        if (split > 1) {
            hipLaunchKernelGGL(keyswitch_init_kernel, dim3(n_jobs), dim3(256), 0, ctx->stream, big_base, ctx->big_stride, d_jobs, n_jobs,
                               p.lwe_dimension, big_dim, out_base, out_stride);
            HIPCHK(ctx, hipGetLastError());
        }
        KS_LAUNCH(2, 16, split);
    }
#undef KS_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    if (profiling) {
        HIPCHK(ctx, hipEventRecord(ep.b, ctx->stream));
        ctx->pending.push_back(ep);
    }
    ctx->stats.ks_count += n_jobs;
    return TFA_OK;
}

// Gates of a stage -> bootstrap, keyswitch and linear jobs, written into `h` (sized by stage_layout).
struct StageLayout {
    uint32_t n_pbs = 0, n_ks = 0, n_lin = 0;
    size_t off_ks = 0, off_lin = 0, total = 0;
};

// Checks one stage and counts its jobs.  Rejected: unknown opcodes, slots outside the pool, two gates writing one slot,
// and a NOT/COPY writing a slot that another NOT/COPY of the stage reads (the linear jobs run concurrently; every
// bootstrapped gate reads the pool as it was before the stage, so those may overlap freely).
static int stage_layout(tfa_ctx *ctx, const tfa_gate *gates, uint32_t count, StageLayout *out) {
    StageLayout L;
    if (ctx->slot_mark.size() < ctx->pool_slots) return fail(ctx, TFA_ESTATE, "slot marks not sized (tfa_pool_reserve)");
    if (ctx->mark_stamp >= 0x7FFFFFF0u) {  // stamps are about to wrap: clear once
        std::fill(ctx->slot_mark.begin(), ctx->slot_mark.end(), 0u);
        std::fill(ctx->lin_reader.begin(), ctx->lin_reader.end(), 0u);
        ctx->mark_stamp = 0;
    }
    const uint32_t stamp = ++ctx->mark_stamp;
    bool any_lin = false;
    for (uint32_t i = 0; i < count; i++) {
        const tfa_gate &g = gates[i];
        if (g.op > TFA_COPY) return fail(ctx, TFA_EINVAL, "unknown gate opcode");
        if (g.in0 >= ctx->pool_slots || g.out >= ctx->pool_slots) return fail(ctx, TFA_EINVAL, "gate slot out of range");
        if (g.op <= TFA_MUX && g.in1 >= ctx->pool_slots) return fail(ctx, TFA_EINVAL, "gate slot out of range");
        if (g.op == TFA_MUX && g.in2 >= ctx->pool_slots) return fail(ctx, TFA_EINVAL, "gate slot out of range");
        if (ctx->slot_mark[g.out] == stamp) return fail(ctx, TFA_EINVAL, "two gates of one stage write slot " + std::to_string(g.out));
        ctx->slot_mark[g.out] = stamp;
        if (g.op == TFA_MUX) {
            L.n_pbs += 2;
            L.n_ks++;
        } else if (g.op <= TFA_XNOR) {
            L.n_pbs++;
            L.n_ks++;
        } else {
            L.n_lin++;
            any_lin = true;
            // remember who reads in0 among the linear gates: gate index + 1 under the stamp, or "several"
            uint64_t &r = ctx->lin_reader[g.in0];
            r = ((uint32_t)(r >> 32) == stamp) ? (((uint64_t)stamp << 32) | 0xFFFFFFFFu) : (((uint64_t)stamp << 32) | (i + 1));
        }
    }
    if (any_lin)
        for (uint32_t i = 0; i < count; i++) {
            const tfa_gate &g = gates[i];
            if (g.op < TFA_NOT) continue;
            const uint64_t r = ctx->lin_reader[g.out];
            if ((uint32_t)(r >> 32) == stamp && (uint32_t)r != i + 1)
                return fail(ctx, TFA_EINVAL, "a NOT/COPY writes slot " + std::to_string(g.out) + ", which another NOT/COPY of the stage reads");
        }
    L.off_ks = (sizeof(PbsJob) * L.n_pbs + 15) / 16 * 16;
    L.off_lin = L.off_ks + (sizeof(KsJob) * L.n_ks + 15) / 16 * 16;
    L.total = L.off_lin + sizeof(LinJob) * L.n_lin;
    *out = L;
    return TFA_OK;
}

static void stage_translate(const tfa_gate *gates, uint32_t count, const StageLayout &L, unsigned char *h) {
    PbsJob *pj = reinterpret_cast<PbsJob *>(h);
    KsJob *kj = reinterpret_cast<KsJob *>(h + L.off_ks);
    LinJob *lj = reinterpret_cast<LinJob *>(h + L.off_lin);
    const uint32_t T = 1u << 29;
    uint32_t ip = 0, ik = 0, il = 0;
    for (uint32_t i = 0; i < count; i++) {
        const tfa_gate &g = gates[i];
        const int32_t s0 = (g.neg & 1) ? -1 : 1, s1 = (g.neg & 2) ? -1 : 1, s2 = (g.neg & 4) ? -1 : 1;
        PbsJob j{};
        switch (g.op) {
        case TFA_AND: j = PbsJob{g.in0, g.in1, s0, s1, 0u - T, {0, 0, 0}}; break;
        case TFA_NAND: j = PbsJob{g.in0, g.in1, -s0, -s1, T, {0, 0, 0}}; break;
        case TFA_OR: j = PbsJob{g.in0, g.in1, s0, s1, T, {0, 0, 0}}; break;
        case TFA_NOR: j = PbsJob{g.in0, g.in1, -s0, -s1, 0u - T, {0, 0, 0}}; break;
        case TFA_XOR: j = PbsJob{g.in0, g.in1, 2 * s0, 2 * s1, 2u * T, {0, 0, 0}}; break;
        case TFA_XNOR: j = PbsJob{g.in0, g.in1, -2 * s0, -2 * s1, 0u - 2u * T, {0, 0, 0}}; break;
        case TFA_MUX:
            pj[ip] = PbsJob{g.in0, g.in1, s0, s1, 0u - T, {0, 0, 0}};    // cond AND then
            pj[ip + 1] = PbsJob{g.in0, g.in2, -s0, s2, 0u - T, {0, 0, 0}};  // (NOT cond) AND else
            kj[ik++] = KsJob{ip, ip + 1, T, g.out};
            ip += 2;
            continue;
        case TFA_NOT: lj[il++] = LinJob{g.in0, g.out, -s0, 0}; continue;
        default: lj[il++] = LinJob{g.in0, g.out, s0, 0}; continue;
        }
        pj[ip] = j;
        kj[ik++] = KsJob{ip, kNoSlot, 0u, g.out};
        ip++;
    }
}

// The three launches of a stage on job lists at `d_jobs` and big-LWE rows at `d_big`.
static int stage_launch(tfa_ctx *ctx, const StageLayout &L, unsigned char *d_jobs, uint32_t *d_big, bool timed) {
    int rc = launch_pbs(ctx, ctx->d_pool, ctx->pool_stride, reinterpret_cast<const PbsJob *>(d_jobs), L.n_pbs, ctx->p.lwe_dimension, nullptr, d_big, timed);
    if (rc) return rc;
    // NOT/COPY before the keyswitch, the only writer of bootstrapped results: every gate of the stage, linear ones
    // included, reads the pool as it was before the stage, so a gate may write the slot it (or another gate) reads.
    if (L.n_lin) {
        hipLaunchKernelGGL(lin_kernel, dim3(L.n_lin), dim3(256), 0, ctx->stream, ctx->d_pool, ctx->pool_stride,
                           reinterpret_cast<const LinJob *>(d_jobs + L.off_lin), L.n_lin, ctx->p.lwe_dimension);
        HIPCHK(ctx, hipGetLastError());
    }
    return launch_ks(ctx, reinterpret_cast<const KsJob *>(d_jobs + L.off_ks), L.n_ks, ctx->d_pool, ctx->pool_stride, d_big, timed);
}

extern "C" int tfa_stage_submit(tfa_ctx *ctx, const tfa_gate *gates, uint32_t count) {
    if (!ctx || (!gates && count)) return TFA_EINVAL;
    if (count == 0) return TFA_OK;
    if (!ctx->have_keys) return fail(ctx, TFA_ESTATE, "tfa_stage_submit before tfa_keys_upload");
    if (!ctx->d_pool) return fail(ctx, TFA_ESTATE, "tfa_stage_submit before tfa_pool_reserve");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    StageLayout L;
    int rc = stage_layout(ctx, gates, count, &L);
    if (rc) return rc;
    unsigned char *h = nullptr;
    int slot = staging(ctx, L.total, &h);
    if (slot < 0) return slot;
    stage_translate(gates, count, L, h);
    rc = ensure_jobs(ctx, L.total);
    if (rc) return rc;
    if (ctx->trace) fprintf(stderr, "tfa stage: gates=%u pbs=%u ks=%u lin=%u\n", count, L.n_pbs, L.n_ks, L.n_lin);
    rc = ensure_big(ctx, L.n_pbs);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, h, L.total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->h_stage_ev[slot], ctx->stream));
    rc = stage_launch(ctx, L, ctx->d_jobs, ctx->d_big, true);
    if (rc) return rc;
    ctx->stats.gate_count += count;
    ctx->stats.stage_count++;
    return TFA_OK;
}

// ---- the entry point that allocates nothing -----------------------------------------------------
// Job lists and big-LWE rows of tfa_stage_submit_noalloc live in this file-scope device array instead of the
// context's grow-only buffers.  Every GPU holds its own copy at its own address (looked up once per device in
// tfa_ctx_create); all calls on a GPU share it, so calls that overlap on one GPU race.
static __device__ __align__(256) unsigned char scratch[TFA_NOALLOC_SCRATCH_BYTES];
namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices];  // device address of `scratch` per GPU; the array's name itself is only a host-side handle
constexpr size_t kNoAllocJobBytes = (sizeof(PbsJob) * TFA_NOALLOC_MAX_PBS + 15) / 16 * 16 + (sizeof(KsJob) * TFA_NOALLOC_MAX_GATES + 15) / 16 * 16 +
                                    sizeof(LinJob) * TFA_NOALLOC_MAX_GATES;
constexpr size_t kNoAllocBigOffset = (kNoAllocJobBytes + 255) / 256 * 256;
constexpr size_t kBigStrideWords = (size_t)(kK1 - 1) * kN + 4;  // big LWE rows: k N + 1 words padded to 16 bytes
static_assert(kNoAllocBigOffset + (size_t)TFA_NOALLOC_MAX_PBS * kBigStrideWords * sizeof(uint32_t) <= TFA_NOALLOC_SCRATCH_BYTES,
              "static scratch too small for its stated limits");
}  // namespace

static int noalloc_setup(tfa_ctx *ctx) {  // tfa_ctx_create, with the context's device current
    if (ctx->device >= kMaxDevices) return TFA_OK;  // the no-alloc entry point then reports the device as unsupported
    if (!g_scratch_ptr[ctx->device]) {
        void *ptr = nullptr;
        HIPCHK(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[ctx->device] = static_cast<unsigned char *>(ptr);
    }
    // translation buffer for the job lists: pinned, fixed size, owned by the context
    HIPCHK(ctx, hipHostMalloc(&ctx->h_noalloc, kNoAllocJobBytes, hipHostMallocDefault));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->h_noalloc_ev, hipEventDisableTiming));
    return TFA_OK;
}

extern "C" int tfa_stage_submit_noalloc(tfa_ctx *ctx, const tfa_gate *gates, uint32_t count) {
    if (!ctx || (!gates && count)) return TFA_EINVAL;
    if (count == 0) return TFA_OK;
    if (!ctx->have_keys) return fail(ctx, TFA_ESTATE, "tfa_stage_submit_noalloc before tfa_keys_upload");
    if (!ctx->d_pool) return fail(ctx, TFA_ESTATE, "tfa_stage_submit_noalloc before tfa_pool_reserve");
    if (ctx->device >= kMaxDevices || !g_scratch_ptr[ctx->device] || !ctx->h_noalloc)
        return fail(ctx, TFA_ESTATE, "static scratch not set up for this device");
    if (count > TFA_NOALLOC_MAX_GATES)
        return fail(ctx, TFA_EINVAL, "tfa_stage_submit_noalloc: " + std::to_string(count) + " gates exceed the static scratch (max " +
                                         std::to_string(TFA_NOALLOC_MAX_GATES) + " gates, " + std::to_string(TFA_NOALLOC_MAX_PBS) + " bootstraps)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    StageLayout L;
    int rc = stage_layout(ctx, gates, count, &L);
    if (rc) return rc;
    if (L.n_pbs > TFA_NOALLOC_MAX_PBS)
        return fail(ctx, TFA_EINVAL, "tfa_stage_submit_noalloc: " + std::to_string(L.n_pbs) + " bootstraps exceed the static scratch (max " +
                                         std::to_string(TFA_NOALLOC_MAX_PBS) + ")");
    if (ctx->big_stride != kBigStrideWords) return fail(ctx, TFA_ESTATE, "static scratch laid out for another parameter set");
    unsigned char *dev = g_scratch_ptr[ctx->device];
    HIPCHK(ctx, hipEventSynchronize(ctx->h_noalloc_ev));  // the previous call's upload has left the translation buffer
    stage_translate(gates, count, L, ctx->h_noalloc);
    HIPCHK(ctx, hipMemcpyAsync(dev, ctx->h_noalloc, L.total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->h_noalloc_ev, ctx->stream));
    rc = stage_launch(ctx, L, dev, reinterpret_cast<uint32_t *>(dev + kNoAllocBigOffset), false);
    if (rc) return rc;
    ctx->stats.gate_count += count;
    ctx->stats.stage_count++;
    return TFA_OK;
}

extern "C" int tfa_sync(tfa_ctx *ctx) {
    if (!ctx) return TFA_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return drain_events(ctx);
}

// ---------------------------------------------------------------------------------------------
static int bootstrap_common(tfa_ctx *ctx, const uint32_t *lwe_in, uint32_t count, uint32_t steps, uint32_t *acc_host, uint32_t *big_host) {
    if (!ctx || !lwe_in || count == 0) return TFA_EINVAL;
    if (!ctx->have_keys) return fail(ctx, TFA_ESTATE, "bootstrap before tfa_keys_upload");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const tfa_params &p = ctx->p;
    const uint32_t words = p.lwe_dimension + 1, kN_ = p.glwe_dimension * p.polynomial_size;
    const size_t in_bytes = (size_t)count * words * sizeof(uint32_t);
    const size_t acc_off = (in_bytes + 255) / 256 * 256;
    const size_t acc_bytes = acc_host ? (size_t)count * (kN_ + p.polynomial_size) * sizeof(uint32_t) : 0;
    int rc = ensure_io(ctx, acc_off + acc_bytes);
    if (rc) return rc;
    rc = ensure_big(ctx, count);
    if (rc) return rc;
    std::vector<PbsJob> jobs(count);
    for (uint32_t i = 0; i < count; i++) jobs[i] = PbsJob{i, i, 1, 0, 0u, {0, 0, 0}};
    rc = ensure_jobs(ctx, sizeof(PbsJob) * count);
    if (rc) return rc;
    unsigned char *io = reinterpret_cast<unsigned char *>(ctx->d_io);
    HIPCHK(ctx, hipMemcpyAsync(io, lwe_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, jobs.data(), sizeof(PbsJob) * count, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    rc = launch_pbs(ctx, reinterpret_cast<const uint32_t *>(io), words, reinterpret_cast<const PbsJob *>(ctx->d_jobs), count, steps,
                    acc_host ? reinterpret_cast<uint32_t *>(io + acc_off) : nullptr, ctx->d_big);
    if (rc) return rc;
    if (acc_host)
        HIPCHK(ctx, hipMemcpyAsync(acc_host, io + acc_off, acc_bytes, hipMemcpyDeviceToHost, ctx->stream));
    else
        HIPCHK(ctx, hipMemcpy2DAsync(big_host, (kN_ + 1) * sizeof(uint32_t), ctx->d_big, ctx->big_stride * sizeof(uint32_t),
                                     (kN_ + 1) * sizeof(uint32_t), count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return drain_events(ctx);
}

extern "C" int tfa_bootstrap_batch(tfa_ctx *ctx, const uint32_t *lwe_in, uint32_t count, uint32_t *big_out) {
    if (!big_out) return TFA_EINVAL;
    return bootstrap_common(ctx, lwe_in, count, ctx ? ctx->p.lwe_dimension : 0, nullptr, big_out);
}
extern "C" int tfa_blind_rotate_batch(tfa_ctx *ctx, const uint32_t *lwe_in, uint32_t count, uint32_t steps, uint32_t *acc_out) {
    if (!acc_out) return TFA_EINVAL;
    return bootstrap_common(ctx, lwe_in, count, steps, acc_out, nullptr);
}

extern "C" int tfa_keyswitch_batch(tfa_ctx *ctx, const uint32_t *big_in, uint32_t count, uint32_t *lwe_out) {
    if (!ctx || !big_in || !lwe_out || count == 0) return TFA_EINVAL;
    if (!ctx->have_keys) return fail(ctx, TFA_ESTATE, "keyswitch before tfa_keys_upload");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const tfa_params &p = ctx->p;
    const uint32_t words = p.lwe_dimension + 1, kN_ = p.glwe_dimension * p.polynomial_size;
    int rc = ensure_big(ctx, count);
    if (rc) return rc;
    rc = ensure_io(ctx, (size_t)count * ctx->pool_stride * sizeof(uint32_t));
    if (rc) return rc;
    std::vector<KsJob> jobs(count);
    for (uint32_t i = 0; i < count; i++) jobs[i] = KsJob{i, kNoSlot, 0u, i};
    rc = ensure_jobs(ctx, sizeof(KsJob) * count);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_big, ctx->big_stride * sizeof(uint32_t), big_in, (kN_ + 1) * sizeof(uint32_t), (kN_ + 1) * sizeof(uint32_t),
                                 count, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, jobs.data(), sizeof(KsJob) * count, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    rc = launch_ks(ctx, reinterpret_cast<const KsJob *>(ctx->d_jobs), count, ctx->d_io, ctx->pool_stride, ctx->d_big);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(lwe_out, words * sizeof(uint32_t), ctx->d_io, ctx->pool_stride * sizeof(uint32_t), words * sizeof(uint32_t), count,
                                 hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return drain_events(ctx);
}

extern "C" int tfa_set_profiling(tfa_ctx *ctx, int on) {
    if (!ctx) return TFA_EINVAL;
    ctx->profiling = on != 0;
    return TFA_OK;
}
extern "C" int tfa_get_stats(tfa_ctx *ctx, tfa_stats *out) {
    if (!ctx || !out) return TFA_EINVAL;
    *out = ctx->stats;
    return TFA_OK;
}
extern "C" int tfa_reset_stats(tfa_ctx *ctx) {
    if (!ctx) return TFA_EINVAL;
    int rc = tfa_sync(ctx);
    ctx->stats = tfa_stats{};
    ctx->launches.clear();
    return rc;
}
extern "C" int tfa_profile_launches(tfa_ctx *ctx, tfa_launch_record *out, uint32_t capacity, uint32_t *count) {
    if (!ctx || !count || (capacity && !out)) return TFA_EINVAL;
    int rc = tfa_sync(ctx);  // drains the pending events into the records
    if (rc) return rc;
    const uint32_t have = (uint32_t)ctx->launches.size(), n = have < capacity ? have : capacity;
    for (uint32_t i = 0; i < n; i++) out[i] = ctx->launches[i];
    *count = have;
    return TFA_OK;
}

#ifdef TFA_STAMPS
// diagnostic builds only: read and clear the per-phase cycle counters of pbs_kernel
extern "C" int tfa_debug_read_stamps(tfa_ctx *ctx, unsigned long long *out /*[32]*/) {
    if (!ctx || !out) return TFA_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32));
    unsigned long long zero[32] = {0};
    HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero));
    return TFA_OK;
}
#endif
