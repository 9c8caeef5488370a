// LWE keyswitch (big key, dimension kN -> small key, dimension n) for a batch of gates, plus the
// merge of the two bootstraps of a MUX.  Pure wrapping u32 arithmetic: results are bit-exact.
//
//   out = (0, ..., 0, b) - sum_{i < kN} sum_{j < l_ks} digit_j(a_i) * KSK[i][j]
//
// As a batch this is the integer GEMM [gates x kN*l_ks] * [kN*l_ks x (n+1)].  A workgroup owns a
// tile of TG gates and all n+1 output columns (13 waves x 62 lanes = 806 columns), so each input
// coefficient is decomposed once; the key-switching key streams through once per tile and every
// word loaded feeds TG multiply-accumulates.  Follows tfhe-rs keyswitch_lwe_ciphertext as run by
// every ServerKey gate (call sites /root/reference/base/src/boolean_tree.rs:644-654).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfa {

constexpr int kKsChunk = 64;       // input coefficients decomposed per LDS refill
constexpr int kKsMaxLevels = 8;
constexpr int kKsLanesUsed = 62;   // columns per wave
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

// out_slot <- KS(big[a] (+ big[b]) + (0,..,0,cst))
struct KsJob {
    uint32_t a, b;
    uint32_t cst;
    uint32_t out;
};

// TG gates per workgroup; UNROLL key rows in flight per lane.  gridDim.y > 1 splits the kN input
// coefficients over workgroups (narrow stages: more CUs share the key stream); the partial sums are
// then combined with integer atomics, which keeps the result bit-exact whatever the order.  In that
// mode the output rows must have been initialised by keyswitch_init_kernel.
template <int TG, int UNROLL>
__global__ void __launch_bounds__(832)
    keyswitch_kernel(const uint32_t *__restrict__ big, uint32_t big_stride, const KsJob *__restrict__ jobs, uint32_t n_jobs,
                     const uint32_t *__restrict__ ksk, uint32_t ksk_stride, uint32_t n, uint32_t big_dim, uint32_t base_log, uint32_t levels,
                     uint32_t *__restrict__ out_base, uint32_t out_stride) {
    extern __shared__ __align__(16) int32_t digs[];  // [kKsChunk * levels][TG]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t col = wave * kKsLanesUsed + lane;
    const bool col_ok = lane < kKsLanesUsed && col <= n;
    const uint32_t g0 = blockIdx.x * TG;
    const bool split = gridDim.y > 1;
    const uint32_t chunks = (big_dim + kKsChunk - 1) / kKsChunk;
    const uint32_t c_begin = (uint32_t)((uint64_t)chunks * blockIdx.y / gridDim.y), c_end = (uint32_t)((uint64_t)chunks * (blockIdx.y + 1) / gridDim.y);
    uint32_t acc[TG];
#pragma unroll
    for (int g = 0; g < TG; g++) acc[g] = 0;

    const uint32_t rep = base_log * levels, non_rep = 32 - rep, mask = (1u << base_log) - 1u;
    for (uint32_t cidx = c_begin; cidx < c_end; cidx++) {
        const uint32_t i0 = cidx * kKsChunk;
        __syncthreads();
        // decompose big[gate][i0 .. i0+chunk) for the TG gates of the tile
        for (uint32_t p = threadIdx.x; p < TG * kKsChunk; p += blockDim.x) {
            uint32_t gl = p / kKsChunk, il = p % kKsChunk, gi = g0 + gl, i = i0 + il;
            uint32_t x = 0;
            if (gi < n_jobs && i < big_dim) {
                KsJob jb = jobs[gi];
                x = big[(size_t)jb.a * big_stride + i];
                if (jb.b != kNoSlot) x += big[(size_t)jb.b * big_stride + i];
            }
            uint32_t state = ((x >> (non_rep - 1)) + 1u) >> 1;
            state &= (1u << rep) - 1u;
            for (uint32_t lv = levels; lv >= 1; lv--) {
                uint32_t res = state & mask;
                state >>= base_log;
                uint32_t carry = (((res - 1u) | state) & res) >> (base_log - 1);
                state += carry;
                digs[(il * levels + (lv - 1)) * TG + gl] = (int32_t)res - (int32_t)(carry << base_log);
            }
        }
        __syncthreads();
        if (col_ok) {
            const uint32_t rows = (i0 + kKsChunk <= big_dim ? kKsChunk : big_dim - i0) * levels;
            const uint32_t *kp = ksk + (size_t)i0 * levels * ksk_stride + col;
            uint32_t r = 0;
            for (; r + UNROLL <= rows; r += UNROLL) {
                uint32_t x[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) x[u] = kp[(size_t)(r + u) * ksk_stride];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    const int32_t *d = digs + (r + u) * TG;
#pragma unroll
                    for (int g = 0; g < TG; g++) acc[g] -= (uint32_t)d[g] * x[u];
                }
            }
            for (; r < rows; r++) {
                uint32_t x = kp[(size_t)r * ksk_stride];
                const int32_t *d = digs + r * TG;
#pragma unroll
                for (int g = 0; g < TG; g++) acc[g] -= (uint32_t)d[g] * x;
            }
        }
    }
    if (col_ok) {
#pragma unroll
        for (int g = 0; g < TG; g++) {
            uint32_t gi = g0 + g;
            if (gi < n_jobs) {
                KsJob jb = jobs[gi];
                uint32_t *dst = out_base + (size_t)jb.out * out_stride + col;
                if (split) {
                    atomicAdd(dst, acc[g]);
                } else {
                    uint32_t v = acc[g];
                    if (col == n) {
                        v += big[(size_t)jb.a * big_stride + big_dim] + jb.cst;
                        if (jb.b != kNoSlot) v += big[(size_t)jb.b * big_stride + big_dim];
                    }
                    *dst = v;
                }
            }
        }
    }
}

// Output rows for the split mode: (0, ..., 0, body)
__global__ void keyswitch_init_kernel(const uint32_t *__restrict__ big, uint32_t big_stride, const KsJob *__restrict__ jobs, uint32_t n_jobs, uint32_t n,
                                      uint32_t big_dim, uint32_t *__restrict__ out_base, uint32_t out_stride) {
    const uint32_t gi = blockIdx.x;
    if (gi >= n_jobs) return;
    const KsJob jb = jobs[gi];
    for (uint32_t col = threadIdx.x; col <= n; col += blockDim.x) {
        uint32_t v = 0;
        if (col == n) {
            v = big[(size_t)jb.a * big_stride + big_dim] + jb.cst;
            if (jb.b != kNoSlot) v += big[(size_t)jb.b * big_stride + big_dim];
        }
        out_base[(size_t)jb.out * out_stride + col] = v;
    }
}

// ---- keyswitch on the int8 matrix cores ------------------------------------------------------------
// The keyswitch of a batch is an integer matrix product: [gates x (kN * levels)] signed digits (|d| <= 2^(base_log-1))
// times the [kN * levels x (n+1)] key, mod 2^32.  Each 32-bit key word is recoded once, at upload, into four
// signed bytes (word = sum_b byte_b * 2^(8b) mod 2^32); the four byte planes are multiplied separately by
// v_mfma_i32_16x16x64_i8 into int32 accumulators (|sum| <= 7680 * 4 * 128 < 2^22, no overflow) and recombined
// with shifts at the end, which is exact.
// Workgroup = 4 waves = 128 gates x 64 output columns; wave w owns columns 16w..16w+15 (all four planes), as
// 8 row tiles of 16 gates.  K runs over blocks of 64 input coefficients: the block's digits are decomposed
// into LDS once ([level][gate][64 coefficients] int8, rows padded to 80 bytes: conflict-free 16-byte reads),
// then one MFMA K-step per level.  Key layout ksk8: [column tile][K block][level][wave][plane][lane][16 bytes],
// byte j of lane l = coefficient 64 kb + 16 (l >> 4) + j, column 16 w + (l & 15): the same (lane, byte) -> k
// assignment as the digit fragments, so the hardware's k order inside a step does not matter.
constexpr int kKsmGates = 128, kKsmCols = 64, kKsmCoefs = 64, kKsmRowBytes = 80;
typedef int ksm_v4i __attribute__((ext_vector_type(4)));

template <int LEVELS>
__global__ void __launch_bounds__(256, 2)
    keyswitch_mfma_kernel(const uint32_t *__restrict__ big, uint32_t big_stride, const KsJob *__restrict__ jobs, uint32_t n_jobs,
                          const int8_t *__restrict__ ksk8, uint32_t n, uint32_t big_dim, uint32_t base_log, uint32_t *__restrict__ out_base,
                          uint32_t out_stride) {
    __shared__ __align__(16) int8_t digs[LEVELS * kKsmGates * kKsmRowBytes];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g0 = blockIdx.x * kKsmGates, ntile = blockIdx.y;
    const uint32_t n_kb = big_dim / kKsmCoefs;
    const uint32_t rep = base_log * LEVELS, non_rep = 32 - rep, mask = (1u << base_log) - 1u;
    ksm_v4i acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[m][b] = ksm_v4i{0, 0, 0, 0};

    // this thread's decomposition tasks: gate (task >> 4), coefficients 4 * (task & 15) .. + 3 of the block
    uint32_t task_a[8], task_b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t gi = g0 + ((threadIdx.x + 256 * r) >> 4);
        task_a[r] = task_b[r] = kNoSlot;
        if (gi < n_jobs) {
            task_a[r] = jobs[gi].a;
            task_b[r] = jobs[gi].b;
        }
    }
    // gridDim.z > 1: the input coefficients are split over that many workgroups, which add their partial sums into rows
    // initialised by keyswitch_init_kernel (integer atomics: the order does not matter, the result is exact)
    const bool split = gridDim.z > 1;
    const uint32_t kb_begin = (uint32_t)((uint64_t)n_kb * blockIdx.z / gridDim.z), kb_end = (uint32_t)((uint64_t)n_kb * (blockIdx.z + 1) / gridDim.z);
    for (uint32_t kb = kb_begin; kb < kb_end; kb++) {
        __syncthreads();  // the previous block's digit reads are done
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t task = threadIdx.x + 256 * r, gl = task >> 4, quad = task & 15, i0 = kb * kKsmCoefs + quad * 4;
            uint4 x = uint4{0, 0, 0, 0};
            if (task_a[r] != kNoSlot) {
                x = *reinterpret_cast<const uint4 *>(big + (size_t)task_a[r] * big_stride + i0);
                if (task_b[r] != kNoSlot) {
                    const uint4 y = *reinterpret_cast<const uint4 *>(big + (size_t)task_b[r] * big_stride + i0);
                    x.x += y.x;
                    x.y += y.y;
                    x.z += y.z;
                    x.w += y.w;
                }
            }
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
            uint32_t packed[LEVELS];
#pragma unroll
            for (int lv = 0; lv < LEVELS; lv++) packed[lv] = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint32_t state = ((xs[c] >> (non_rep - 1)) + 1u) >> 1;
                state &= (1u << rep) - 1u;
#pragma unroll
                for (int lv = LEVELS; lv >= 1; lv--) {
                    const uint32_t res = state & mask;
                    state >>= base_log;
                    const uint32_t carry = (((res - 1u) | state) & res) >> (base_log - 1);
                    state += carry;
                    const uint32_t digit = res - (carry << base_log);  // two's complement
                    packed[lv - 1] |= (digit & 0xFFu) << (8 * c);
                }
            }
#pragma unroll
            for (int lv = 0; lv < LEVELS; lv++)
                *reinterpret_cast<uint32_t *>(digs + (lv * kKsmGates + gl) * kKsmRowBytes + quad * 4) = packed[lv];
        }
        __syncthreads();
#pragma unroll
        for (int lv = 0; lv < LEVELS; lv++) {
            const ksm_v4i *kp = reinterpret_cast<const ksm_v4i *>(ksk8) + ((((size_t)ntile * n_kb + kb) * LEVELS + lv) * 4 + wave) * 4 * 64 + lane;
            ksm_v4i bq[4];
#pragma unroll
            for (int b = 0; b < 4; b++) bq[b] = kp[b * 64];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const ksm_v4i a = *reinterpret_cast<const ksm_v4i *>(digs + (lv * kKsmGates + 16 * m + (lane & 15)) * kKsmRowBytes + (lane >> 4) * 16);
#pragma unroll
                for (int b = 0; b < 4; b++) acc[m][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq[b], acc[m][b], 0, 0, 0);
            }
        }
    }
    // accumulator tile: column = lane & 15, row = 4 * (lane >> 4) + register
    const uint32_t col = ntile * kKsmCols + wave * 16 + (lane & 15);
    if (col > n) return;
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t gi = g0 + 16 * m + 4 * (lane >> 4) + r;
            if (gi >= n_jobs) continue;
            const KsJob jb = jobs[gi];
            uint32_t v = 0u - ((uint32_t)acc[m][0][r] + ((uint32_t)acc[m][1][r] << 8) + ((uint32_t)acc[m][2][r] << 16) + ((uint32_t)acc[m][3][r] << 24));
            if (split) {
                atomicAdd(out_base + (size_t)jb.out * out_stride + col, v);
                continue;
            }
            if (col == n) {
                v += big[(size_t)jb.a * big_stride + big_dim] + jb.cst;
                if (jb.b != kNoSlot) v += big[(size_t)jb.b * big_stride + big_dim];
            }
            out_base[(size_t)jb.out * out_stride + col] = v;
        }
}

// u32 key [kN * levels][ksk_stride] -> signed byte planes in keyswitch_mfma_kernel's layout; one thread per 16-byte fragment
template <int LEVELS>
__global__ void ksk_to_planes_kernel(const uint32_t *__restrict__ ksk, uint32_t ksk_stride, uint32_t n, uint32_t n_kb, uint32_t n_tiles,
                                     int8_t *__restrict__ ksk8) {
    const size_t frag = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // ((((ntile * n_kb + kb) * LEVELS + lv) * 4 + wave) * 4 + plane) * 64 + lane
    const size_t total = (size_t)n_tiles * n_kb * LEVELS * 4 * 4 * 64;
    if (frag >= total) return;
    const uint32_t lane = frag & 63, plane = (frag >> 6) & 3, wave = (frag >> 8) & 3;
    size_t rest = frag >> 10;
    const uint32_t lv = rest % LEVELS;
    rest /= LEVELS;
    const uint32_t kb = rest % n_kb, ntile = rest / n_kb;
    const uint32_t col = ntile * kKsmCols + wave * 16 + (lane & 15);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t coef = kb * kKsmCoefs + (lane >> 4) * 16 + j;
        uint32_t x = col <= n ? ksk[((size_t)coef * LEVELS + lv) * ksk_stride + col] : 0u;
        uint32_t byte = 0;
        for (uint32_t b = 0; b <= plane; b++) {  // signed-digit recoding, base 256
            byte = x & 0xFFu;
            x = (x - (uint32_t)(int32_t)(int8_t)byte) >> 8;
        }
        w[j >> 2] |= byte << (8 * (j & 3));
    }
    reinterpret_cast<uint4 *>(ksk8)[frag] = uint4{w[0], w[1], w[2], w[3]};
}

// dst = coef * src on whole ciphertexts (NOT: coef = -1, COPY: coef = +1)
struct LinJob {
    uint32_t src, dst;
    int32_t coef;
    uint32_t pad;
};
__global__ void lin_kernel(uint32_t *__restrict__ pool, uint32_t stride, const LinJob *__restrict__ jobs, uint32_t n_jobs, uint32_t n) {
    uint32_t j = blockIdx.x;
    if (j >= n_jobs) return;
    LinJob jb = jobs[j];
    for (uint32_t w = threadIdx.x; w <= n; w += blockDim.x) pool[(size_t)jb.dst * stride + w] = (uint32_t)jb.coef * pool[(size_t)jb.src * stride + w];
}

// pool[slots[s]][0..n] <-> staging[s][0..n]
__global__ void pool_scatter_kernel(uint32_t *__restrict__ pool, uint32_t stride, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ staging,
                                    uint32_t words) {
    uint32_t s = blockIdx.x;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) pool[(size_t)slots[s] * stride + w] = staging[(size_t)s * words + w];
}
__global__ void pool_gather_kernel(const uint32_t *__restrict__ pool, uint32_t stride, const uint32_t *__restrict__ slots, uint32_t *__restrict__ staging,
                                   uint32_t words) {
    uint32_t s = blockIdx.x;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) staging[(size_t)s * words + w] = pool[(size_t)slots[s] * stride + w];
}

// out[s][0..n] = +-pool[slot][0..n]; bit 31 of slots[s] asks for the LWE negation (ServerKey::not), rows of `out` are dense
__global__ void pool_export_kernel(const uint32_t *__restrict__ pool, uint32_t stride, const uint32_t *__restrict__ slots, uint32_t *__restrict__ out,
                                   uint32_t words) {
    const uint32_t s = blockIdx.x, tagged = slots[s], slot = tagged & 0x7FFFFFFFu;
    const uint32_t coef = (tagged >> 31) ? 0xFFFFFFFFu : 1u;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) out[(size_t)s * words + w] = coef * pool[(size_t)slot * stride + w];
}

}  // namespace tfa
