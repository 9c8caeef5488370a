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

}  // namespace tfa
