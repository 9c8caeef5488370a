// Latency variant of the programmable bootstrap for narrow stages: one bootstrap per workgroup, spread
// over its four waves, so that a stage with at most one bootstrap per CU finishes in about a quarter of
// the time the one-wave-per-bootstrap kernel needs.  Same arithmetic as pbs_kernel.h (k = 3, N = 512,
// (beta, l) = (2^10, 2)), different mapping:
//
//   wave w owns polynomial w of the GLWE accumulator; lane L holds the 8 coefficients j = L + 64 m.
//   The size-256 complex transform of a polynomial runs over the 64 lanes as four radix-4 stages
//   (decimation in frequency, in place), 4 points per lane, with three exchanges through wave-private LDS;
//   the inverse walks the same stages backwards (decimation in time), so the frequency order never has to
//   be undone.  The waves publish the spectra of both levels of their polynomial in a double-buffered LDS
//   exchange (one workgroup barrier per step); wave w then accumulates output polynomial w from all
//   eight spectra and column w of the step's Fourier GGSW, which it reads straight from global memory
//   into registers one step ahead (every key word is used by exactly one lane of the workgroup).
//
// Key layout bsk_lat (c64): [i][level][row p][column c][q][L], (q, L) = register slot and lane that hold
// the frequency after the forward transform; 1/256 of the inverse is folded into the key.
#pragma once
#include "pbs_kernel.h"

namespace tfa {

#ifndef TFA_LAT_READ_ALL
#define TFA_LAT_READ_ALL 0  // 1: one workgroup per CU reads all 24 foreign spectrum words before the first product (328 registers; measured 5 % slower)
#endif
#ifndef TFA_LAT_SPREAD_KEY
#define TFA_LAT_SPREAD_KEY 1
#endif
#ifndef TFA_LAT_TOUCH
#define TFA_LAT_TOUCH 1
#endif
constexpr uint32_t kLatTouchAhead = 3;                  // steps between the L2 warm-up of a step's key and its use
constexpr int kLatFftElems = 320;                       // 256 points, one c64 of padding after every 4
constexpr int kLatFftBytes = kLatFftElems * 16;         // 5120
constexpr int kLatWaveBytes = 2 * kLatFftBytes;         // one transform buffer per level; the first doubles as rotation buffer
constexpr int kLatRedOffset = 4 * kLatWaveBytes;        // 40960
constexpr int kLatSpecElems = 4 * 2 * 4 * 64;           // spectra exchange [polynomial][level][q][L], 32 KiB
// PER_CU = 1: one workgroup per CU, the exchange alternates between two buffers (one barrier per step);
// PER_CU = 2: two workgroups share a CU and hide each other's waits, one exchange buffer and a second
// barrier per step so that both fit in the LDS
template <int PER_CU>
struct LatGeom {
    static constexpr int kSpecBuffers = PER_CU == 1 ? 2 : 1;
    static constexpr int kAOffset = kLatRedOffset + kSpecBuffers * kLatSpecElems * 16;
    static constexpr int kLdsBytes = kAOffset + kABytes;  // 108160 / 75392
};
static_assert(LatGeom<1>::kLdsBytes <= 163840 && 2 * LatGeom<2>::kLdsBytes <= 163840, "LDS budget");
static_assert(kN * 4 <= kLatFftBytes, "rotation buffer must fit in a transform buffer");
constexpr int kLatConvertLdsBytes = 4 * kLatFftBytes;

// tw_lat: [0, 256) twist e^{i pi n / 512}; [256, 512) e^{2 pi i k / 256}
struct LatTwiddles {
    double twr[4], twi[4];        // twist of point n = L + 64 q
    double sr[3][3], si[3][3];    // stage s, output r + 1: e^{2 pi i o (r + 1) / span_s}, o = L mod (span_s / 4)
};
__device__ __forceinline__ void lat_load_twiddles(LatTwiddles &T, const c64 *__restrict__ tw_lat, int L) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        c64 v = tw_lat[L + 64 * q];
        T.twr[q] = v.x;
        T.twi[q] = v.y;
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int quarter = 64 >> (2 * s), o = L & (quarter - 1), scale = 1 << (2 * s);  // W_span = W_256^(4^s)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            c64 v = tw_lat[256 + ((o * (r + 1) * scale) & 255)];
            T.sr[s][r] = v.x;
            T.si[s][r] = v.y;
        }
    }
}

// padded position of point `slot` of lane L's butterfly in stage s: element p sits at p + p / 4, which is
// linear in the slot (stride 5 * quarter / 4, or 1 in the last stage), so a lane needs one base per stage
__device__ __forceinline__ int lat_pos(int s, int L, int slot) {
    const int quarter = 64 >> (2 * s);
    const int p0 = (L / quarter) * (4 * quarter) + (L & (quarter - 1));
    return p0 + (p0 >> 2) + slot * (s < 3 ? 5 * quarter / 4 : 1);
}

// hand-over between stage `s_from` (results written to its positions) and stage `s_to` (operands read from its positions)
template <int NL>
__device__ __forceinline__ void lat_exchange(double (&zr)[NL][4], double (&zi)[NL][4], c64 *fb, int s_from, int s_to, int L) {
#ifdef TFA_EXP_LAT_NOEXCH  // timing experiment: no hand-over between stages (results are garbage)
    return;
#endif
#pragma unroll
    for (int lv = 0; lv < NL; lv++)
#pragma unroll
        for (int r = 0; r < 4; r++) fb[lv * kLatFftElems + lat_pos(s_from, L, r)] = c64{zr[lv][r], zi[lv][r]};
    wave_lds_fence();
#pragma unroll
    for (int lv = 0; lv < NL; lv++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            c64 v = fb[lv * kLatFftElems + lat_pos(s_to, L, q)];
            zr[lv][q] = v.x;
            zi[lv][q] = v.y;
        }
    wave_lds_fence();
}

// one level's stage: butterfly, twiddles, and (all but the last stage) the hand-over to the next stage's layout
template <int NL>
__device__ __forceinline__ void lat_stage(double (&zr)[NL][4], double (&zi)[NL][4], int lv, int s, const LatTwiddles &T, c64 *fb, int L) {
    dft4<1>(zr[lv][0], zi[lv][0], zr[lv][1], zi[lv][1], zr[lv][2], zi[lv][2], zr[lv][3], zi[lv][3]);
    if (s == 3) return;
#pragma unroll
    for (int r = 0; r < 3; r++) cmulc<1>(zr[lv][r + 1], zi[lv][r + 1], T.sr[s][r], T.si[s][r]);
#ifdef TFA_EXP_LAT_NOEXCH
    return;
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) fb[lv * kLatFftElems + lat_pos(s, L, r)] = c64{zr[lv][r], zi[lv][r]};
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        c64 v = fb[lv * kLatFftElems + lat_pos(s + 1, L, q)];
        zr[lv][q] = v.x;
        zi[lv][q] = v.y;
    }
    wave_lds_fence();
}
// in: z[lv][q] = point n = L + 64 q of NL polynomials (folded, untwisted); out: z[lv][q] = frequency slot (q, L).
// The levels alternate stage by stage, each in its own buffer, so one level's LDS round trip runs under the other's butterfly.
template <int NL>
__device__ __forceinline__ void lat_forward(double (&zr)[NL][4], double (&zi)[NL][4], const LatTwiddles &T, c64 *fb, int L) {
#pragma unroll
    for (int lv = 0; lv < NL; lv++)
#pragma unroll
        for (int q = 0; q < 4; q++) cmulc<1>(zr[lv][q], zi[lv][q], T.twr[q], T.twi[q]);
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int lv = 0; lv < NL; lv++) lat_stage<NL>(zr, zi, lv, s, T, fb, L);
}

// exact mirror of lat_forward (times 256).  between(s) runs before stage s: independent work (the next step's key
// reads) that the compiler may not move across the hand-overs' fences by itself, placed where the wave would
// otherwise only wait for the LDS.
template <class Between>
__device__ __forceinline__ void lat_inverse(double (&zr)[1][4], double (&zi)[1][4], const LatTwiddles &T, c64 *fb, int L, Between &&between) {
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        between(s);
        if (s < 3) {
#pragma unroll
            for (int r = 0; r < 3; r++) cmulc<-1>(zr[0][r + 1], zi[0][r + 1], T.sr[s][r], T.si[s][r]);
        }
        dft4<-1>(zr[0][0], zi[0][0], zr[0][1], zi[0][1], zr[0][2], zi[0][2], zr[0][3], zi[0][3]);
        if (s > 0) lat_exchange<1>(zr, zi, fb, s, s - 1, L);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) cmulc<-1>(zr[0][q], zi[0][q], T.twr[q], T.twi[q]);
}

__device__ __forceinline__ size_t bsk_lat_index(uint32_t i, int lv, int p, int c, int q) {
    return (((((size_t)i * kLevels + lv) * kK1 + p) * kK1 + c) * 4 + q) * 64;
}

template <int PER_CU>
__global__ void __launch_bounds__(256, PER_CU)
    pbs_lat_kernel(const uint32_t *__restrict__ in_base, uint32_t in_stride, const PbsJob *__restrict__ jobs, uint32_t n_jobs,
                   const c64 *__restrict__ bsk_lat, const c64 *__restrict__ tw_lat, uint32_t n, uint32_t steps, uint32_t *__restrict__ big_out,
                   uint32_t big_stride, uint32_t *__restrict__ acc_out) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = threadIdx.x & 63;
    const uint32_t job = blockIdx.x;
    if (job >= n_jobs) return;  // whole workgroup
    c64 *fb = reinterpret_cast<c64 *>(lds + w * kLatWaveBytes);
    uint32_t *rot = reinterpret_cast<uint32_t *>(fb);
    c64 *red = reinterpret_cast<c64 *>(lds + kLatRedOffset);
    uint16_t *a_tilde = reinterpret_cast<uint16_t *>(lds + LatGeom<PER_CU>::kAOffset);

    // gate linear combination + modulus switch
    const PbsJob jb = jobs[job];
    const uint32_t *pa = in_base + (size_t)jb.a * in_stride, *pb = in_base + (size_t)jb.b * in_stride;
    for (uint32_t idx = threadIdx.x; idx <= n; idx += 256) {
        uint32_t u = (uint32_t)jb.ca * pa[idx] + (uint32_t)jb.cb * pb[idx];
        if (idx == n) u += jb.cst;
        a_tilde[idx] = (uint16_t)modswitch_2n(u);
    }
    LatTwiddles T;
    lat_load_twiddles(T, tw_lat, L);
    __syncthreads();
    const uint32_t b_tilde = a_tilde[n];

    // accumulator = X^{-b~} * (0, 0, 0, 2^29 * (1 + X + ... + X^{N-1}))
    uint32_t acc[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        uint32_t jj = (uint32_t)(64 * m + L) + b_tilde;  // < 3N
        acc[m] = (w == 3) ? (((jj >> 9) & 1u) ? 0xE0000000u : 0x20000000u) : 0u;
    }

    const uint32_t n_steps = steps < n ? steps : n;
    // key words of the first step: [level][polynomial p][q], column w
    c64 key[2][4][4];
    if (n_steps > 0) {
#pragma unroll
        for (int lv = 0; lv < 2; lv++)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int q = 0; q < 4; q++) key[lv][c][q] = bsk_lat[bsk_lat_index(0, lv, c, w, q) + L];
    }

    // L2 warm-up: every workgroup of an XCD (workgroups go to the XCDs round-robin) touches its share of the 1024
    // 128-byte lines of the key of step i + kLatTouchAhead, so the key reads one step ahead hit the XCD's L2
    const uint32_t touch_wgs = (gridDim.x + 7) / 8, touch_line = blockIdx.x / 8 + threadIdx.x * touch_wgs;
    uint32_t touched = 0;
    for (uint32_t i = 0; i < n_steps; i++) {
        const int a = __builtin_amdgcn_readfirstlane((int)a_tilde[i]);
#if TFA_LAT_TOUCH
        {
            asm volatile("" ::"v"(touched));  // the previous step's touch has returned by now (keeps the load alive and counted)
            if (touch_line < 1024 && i + kLatTouchAhead < n_steps)
                touched = reinterpret_cast<const volatile uint32_t *>(bsk_lat + (size_t)(i + kLatTouchAhead) * 8192)[touch_line * 32];
        }
#endif
        // d = acc * X^a - acc through the rotation buffer, then signed decomposition
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < 8; m++) rot[64 * m + L] = acc[m];
        wave_lds_fence();
        double zr[2][4], zi[2][4];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            int jj = 64 * m + L - a;  // (-1024, 512]
            const uint32_t v = rot[jj & 511];
            int d1, d2;
            rotated_digits(v, jj, acc[m], d1, d2);
            if (m < 4) {
                zr[0][m] = (double)d1;
                zr[1][m] = (double)d2;
            } else {
                zi[0][m - 4] = (double)d1;
                zi[1][m - 4] = (double)d2;
            }
        }
        wave_lds_fence();
        lat_forward<2>(zr, zi, T, fb, L);

        // publish both spectra.  With two buffers they alternate with the step parity: a wave can only write
        // step i + 2's spectra after the barrier of step i + 1, which every wave passes after it has read step
        // i's.  With one buffer a second barrier below separates this step's reads from the next step's writes.
        c64 *sp = red + (LatGeom<PER_CU>::kSpecBuffers == 2 ? (i & 1) * kLatSpecElems : 0);
#pragma unroll
        for (int lv = 0; lv < 2; lv++)
#pragma unroll
            for (int q = 0; q < 4; q++) sp[((w * 2 + lv) * 4 + q) * 64 + L] = c64{zr[lv][q], zi[lv][q]};
        // output polynomial w = sum over polynomials p and levels of spectrum[p][lv] * key[lv][p][w];
        // the own polynomial's term needs no exchange and covers the barrier wait
        double our[1][4], oui[1][4];
#pragma unroll
        for (int q = 0; q < 4; q++) our[0][q] = oui[0][q] = 0.0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p != w) continue;
#pragma unroll
            for (int lv = 0; lv < 2; lv++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const c64 b = key[lv][p][q];
                    our[0][q] += zr[lv][q] * b.x - zi[lv][q] * b.y;
                    oui[0][q] += zr[lv][q] * b.y + zi[lv][q] * b.x;
                }
        }
#ifndef TFA_EXP_NOBARRIER
        __syncthreads();
#endif
#if TFA_LAT_READ_ALL
        if constexpr (PER_CU == 1) {
            // one wave per SIMD may use 512 registers: all 24 reads of the other polynomials' spectra in flight at once
            c64 fa[3][8];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int p = (w + 1 + k) & 3;
#pragma unroll
                for (int e = 0; e < 8; e++) fa[k][e] = sp[(p * 8 + e) * 64 + L];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int pp = 0; pp < 4; pp++) {
                    if (pp != ((w + 1 + k) & 3)) continue;
#pragma unroll
                    for (int lv = 0; lv < 2; lv++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const c64 b = key[lv][pp][q];
                            our[0][q] += fa[k][lv * 4 + q].x * b.x - fa[k][lv * 4 + q].y * b.y;
                            oui[0][q] += fa[k][lv * 4 + q].x * b.y + fa[k][lv * 4 + q].y * b.x;
                        }
                }
            }
        } else
#endif
        {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p == w) continue;
            c64 f[8];  // all eight reads of a polynomial in flight before the first product
#pragma unroll
#ifdef TFA_EXP_LAT_NOSPEC  // timing experiment: other polynomials' spectra from registers (results are garbage)
            for (int e = 0; e < 8; e++) f[e] = c64{zr[e >> 2][e & 3] + p, zi[e >> 2][e & 3] - p};
#else
            for (int e = 0; e < 8; e++) f[e] = sp[(p * 8 + e) * 64 + L];
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int lv = 0; lv < 2; lv++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const c64 b = key[lv][p][q];
                    our[0][q] += f[lv * 4 + q].x * b.x - f[lv * 4 + q].y * b.y;
                    oui[0][q] += f[lv * 4 + q].x * b.y + f[lv * 4 + q].y * b.x;
                }
        }
        }
#ifndef TFA_EXP_NOBARRIER
        if (LatGeom<PER_CU>::kSpecBuffers == 1) __syncthreads();
#endif
        // next step's key words replace the ones just consumed; they have almost a whole step to arrive
#ifdef TFA_EXP_NODMA  // timing experiment: the first step's key words are reused (results are garbage)
        const bool more = false;
#else
        const bool more = i + 1 < n_steps;
#endif
#if TFA_LAT_SPREAD_KEY
        // a quarter of the next step's key words before each stage of the inverse transform
        lat_inverse(our, oui, T, fb, L, [&](int s) __attribute__((always_inline)) {
            if (more) {
#pragma unroll
                for (int lv = 0; lv < 2; lv++)
#pragma unroll
                    for (int q = 0; q < 4; q++) key[lv][3 - s][q] = bsk_lat[bsk_lat_index(i + 1, lv, 3 - s, w, q) + L];
            }
        });
#else
        if (more) {
#pragma unroll
            for (int lv = 0; lv < 2; lv++)
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = 0; q < 4; q++) key[lv][p][q] = bsk_lat[bsk_lat_index(i + 1, lv, p, w, q) + L];
        }
        lat_inverse(our, oui, T, fb, L, [](int) {});
#endif
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[q] += round_torus(our[0][q]);
            acc[q + 4] += round_torus(oui[0][q]);
        }
    }

    if (acc_out) {  // raw accumulator [job][polynomial][N]
#pragma unroll
        for (int m = 0; m < 8; m++) acc_out[((size_t)job * kK1 + w) * kN + 64 * m + L] = acc[m];
        return;
    }
    // sample extraction of the constant coefficient: mask block r = (A_r[0], -A_r[N-1], ..., -A_r[1]), body B[0]
    uint32_t *out = big_out + (size_t)job * big_stride;
    if (w < 3) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            int j = 64 * m + L;
            out[w * kN + ((kN - j) & (kN - 1))] = j == 0 ? acc[m] : 0u - acc[m];
        }
    } else if (L == 0) {
        out[3 * kN] = acc[0];
    }
}

// Standard-domain GGSW rows -> bsk_lat.  One workgroup per GGSW row (i, level, p): wave c transforms column c.
__global__ void __launch_bounds__(256, 1)
    bsk_convert_lat_kernel(const uint32_t *__restrict__ bsk, c64 *__restrict__ bsk_lat, const c64 *__restrict__ tw_lat, uint32_t n_rows) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int c = threadIdx.x >> 6, L = threadIdx.x & 63;
    const uint32_t row = blockIdx.x;  // (i * l + lv) * 4 + p
    if (row >= n_rows) return;
    c64 *fb = reinterpret_cast<c64 *>(lds + c * kLatFftBytes);
    LatTwiddles T;
    lat_load_twiddles(T, tw_lat, L);
    const uint32_t *src = bsk + ((size_t)row * kK1 + c) * kN;
    double zr[1][4], zi[1][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        zr[0][q] = (double)(int32_t)src[L + 64 * q];
        zi[0][q] = (double)(int32_t)src[L + 64 * q + 256];
    }
    lat_forward<1>(zr, zi, T, fb, L);
#pragma unroll
    for (int q = 0; q < 4; q++) bsk_lat[(((size_t)row * kK1 + c) * 4 + q) * 64 + L] = c64{zr[0][q] * (1.0 / 256.0), zi[0][q] * (1.0 / 256.0)};
}

}  // namespace tfa
