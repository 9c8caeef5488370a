// Programmable bootstrap (gate linear combination + modulus switch + blind rotation + sample
// extract) for gfx950, specialised for k = 3, N = 512, (beta, l) = (2^10, 2).
//
// Mapping: one 64-lane wavefront owns one bootstrap for all n blind-rotation steps.  Lane = 4 t + g: the sixteen lanes
// with the same g = lane & 3 own polynomial g of the GLWE accumulator; lane t = lane >> 2 holds the 32 coefficients
// j == t (mod 16) in registers.  A negacyclic size-512 transform is a size-256 complex FFT computed as 16 x 16: a
// register-resident DFT16 per lane, one 16 x 16 transpose through wave-private LDS, a second register DFT16.  The
// Fourier domain multiply-accumulate with the bootstrapping key sums over the four polynomials inside each quad of
// neighbouring lanes with DPP quad permutes.  The waves of a workgroup advance in lock step and share the
// bootstrapping key through an LDS ring filled by LDS-DMA, one barrier per 16 KiB chunk.
//
// What the kernel computes follows tfhe-rs 0.11 (blind_rotate_assign / add_external_product_assign,
// the engine under every ServerKey gate called at /root/reference/base/src/boolean_tree.rs:644-654).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace tfa {

// Diagnostic builds (-DTFA_STAMPS) accumulate s_memtime deltas per phase for wave 0 of workgroup 0
// into g_stamps; the shipped library compiles the stamps away.
#ifdef TFA_STAMPS
__device__ unsigned long long g_stamps[32];
#define TFA_STAMP_DECL unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(); const bool stamp_on = blockIdx.x == 0 && threadIdx.x == 0;
#define TFA_STAMP(idx)                                                        \
    do {                                                                      \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();               \
        if (stamp_on) g_stamps[idx] += now_ - stamp_t0;                       \
        stamp_t0 = now_;                                                      \
    } while (0)
#else
#define TFA_STAMP_DECL
#define TFA_STAMP(idx)
#endif

constexpr int kN = 512;          // polynomial size
constexpr int kK1 = 4;           // glwe_dimension + 1
constexpr int kLevels = 2;       // pbs_level
constexpr int kBaseLog = 10;     // pbs_base_log
// Lane mapping.  Quad mode (default): lane = 4 t + g, the four polynomials' lanes for one t are neighbours, so the sum
// over the polynomials in the key products runs on DPP quad permutes (4.5 clk per moved dword at two waves per SIMD,
// in place) instead of the half / row swap instructions (8.6 clk per instruction).  Row mode (TFA_QUAD_LANES=0): lane = 16 g + t.
#ifndef TFA_EXP_QUAD_SWAPS
#define TFA_EXP_QUAD_SWAPS 0  // timing/pressure experiment: quad lane mapping with the swap-based reduction (results are garbage)
#endif
#ifndef TFA_QUAD_LANES
#define TFA_QUAD_LANES 1
#endif
constexpr bool kQuad = TFA_QUAD_LANES != 0;
__device__ __forceinline__ int lane_g(int lane) { return kQuad ? lane & 3 : lane >> 4; }
__device__ __forceinline__ int lane_t(int lane) { return kQuad ? lane >> 2 : lane & 15; }
// u32 words per polynomial in the rotation buffer: the 32 lanes of an LDS access group (8 values of t, 4 of g in quad
// mode) must fall on 32 different banks: stride = 8 mod 32 (quad) or 16 mod 32 (row)
constexpr int kRotStride = kQuad ? 520 : 528;
// Transpose buffer, one real plane of one polynomial per group: lane t stores element r at (row r, column t) and loads
// its row t.  Row mode pads the rows to 18 doubles (16-byte aligned rows, conflict-free ds_read_b128; 17 with 8-byte
// reads measured 2 % slower).  Quad mode needs no row padding but a swizzle: column c of row R sits at c ^ 2 (R >> 2),
// groups 260 doubles apart; found by exhaustive search over paddings and swizzles to be conflict-free for the 8-byte
// stores (16 consecutive lanes = 4 values of t x 4 groups) and for the 16-byte loads (lane groups of ds_read_b128).
#ifndef TFA_QUAD_SWIZZLE
#define TFA_QUAD_SWIZZLE 1  // 0: plain padded rows in quad mode (18 doubles, groups 292 apart): 2-way conflicts on the loads, fewer address registers
#endif
constexpr bool kSwz = kQuad && TFA_QUAD_SWIZZLE;
constexpr int kTrRow = kSwz ? 16 : 18;
constexpr int kTrGroup = kSwz ? 260 : kQuad ? 292 : 16 * kTrRow;
__device__ __forceinline__ int tr_store_index(int r, int t) { return kSwz ? r * 16 + (t ^ (2 * (r >> 2))) : r * kTrRow + t; }  // element r of lane t
__device__ __forceinline__ int tr_load_index(int t, int r) { return kSwz ? t * 16 + (r ^ (2 * (t >> 2))) : t * kTrRow + r; }   // r even: r and r + 1 adjacent
constexpr int kWaveLdsBytes = (4 * kTrGroup * 8 > 4 * kRotStride * 4) ? 4 * kTrGroup * 8 : 4 * kRotStride * 4;  // transpose buffer, aliased by the rotation buffer
constexpr int kABytes = 1664;                    // modulus-switched mask, u16[<=832]
constexpr int kWavePrivBytes = kWaveLdsBytes + kABytes;
constexpr int kTwBytes = 4096;                   // per-lane twiddle table shared by all waves
// Bootstrapping-key ring: the Fourier GGSW of one blind-rotation step (128 KiB) streams through LDS
// in 8 chunks of 16 KiB (two frequency groups k2 each), filled by LDS-DMA and shared by all waves.
#ifndef TFA_RING8_ROWS
#define TFA_RING8_ROWS 16
#endif
#ifndef TFA_PBS8_BOTH
#define TFA_PBS8_BOTH 1  // 0 builds the older 8-wave variant that reduces each level on its own (15 % slower)
#endif
// Build switches of the bootstrap kernels.  The defaults are the measured best; the other settings are kept so that
// the measurements can be repeated (tools/build_exp.sh builds a library per setting, tools/run_exp.sh times them).
#ifndef TFA_PIN_GROUP4
#define TFA_PIN_GROUP4 1  // one wave per SIMD: the sums of a frequency group are pinned after the reduction (0: 7 % slower, 486 registers)
#endif
#ifndef TFA_PIN_GROUP
#define TFA_PIN_GROUP 1  // two waves per SIMD: the same pin; without it two groups' partial sums overlap and 21 registers spill
#endif
#ifndef TFA_MAC_STAGGER
#define TFA_MAC_STAGGER 1  // two waves per SIMD: key columns in two pairs with counted waits (0: one wait per group, 6 % slower in quad mode)
#endif
#ifndef TFA_INV_TW_PREFETCH
#define TFA_INV_TW_PREFETCH 0  // 8 waves: 1 = the inverse transform's twiddles are read before its first DFT16 (measured 4 % slower)
#endif
#ifndef TFA_TR_FENCES
#define TFA_TR_FENCES 1  // 1 = compiler fences around every plane of a transpose; 0 = one fence per transpose (the planes' accesses may alias, which orders them)
#endif
#ifndef TFA_FFT_PAIR
#define TFA_FFT_PAIR 0  // 8 waves: 1 = the two forward transforms of a step share their twiddle reads (fewer LDS reads, measured 2.5 % slower)
#endif
#ifndef TFA_PIN_SPECTRA
#define TFA_PIN_SPECTRA 1  // the forward transforms' outputs are pinned before the products (0: the compiler evaluates them lazily and spills)
#endif
#ifndef TFA_PBS8_PREFETCH
#define TFA_PBS8_PREFETCH 1  // 8 waves: key words of the next frequency group are read while the previous group is reduced
#endif
// WAVES bootstraps per workgroup, PER_CU workgroups per CU.  (8, 1) and (4, 2) put two waves on every SIMD and run the
// 256-register code path ("dense"); (4, 1) keeps one wave per SIMD and spends up to 512 registers on prefetching.
// (4, 2) - two independent workgroups per CU, each with its own 32 KiB ring and a barrier over four waves only - was
// measured at the same rate as (8, 1) (171 k against 172 k bootstraps/s) and is not instantiated by the engine.
template <int WAVES, int PER_CU = 1>
struct PbsGeom {
    static constexpr int kWaves = WAVES;
    static constexpr bool kDense = WAVES * PER_CU == 8;
    // dense, split mode: each level is transformed, multiplied and reduced on its own; otherwise both levels
    // are transformed first and multiplied and reduced together (half the row reductions)
    static constexpr bool kSplitLevels = kDense && !TFA_PBS8_BOTH;
    static constexpr int kChunkRows = kDense ? TFA_RING8_ROWS : 16;  // 1 KiB key rows per chunk
    static constexpr int kChunkBytes = kChunkRows * 1024;
    static constexpr int kChunksPerStep = 128 / kChunkRows;
    static constexpr int kSlots = kDense ? 64 / (kChunkRows * PER_CU) : ((163840 - WAVES * (2 * kWaveLdsBytes + kABytes) - kTwBytes) / 16384 >= 5 ? 5 : 4);
    static constexpr int kRowsPerWave = kChunkRows / WAVES;  // DMA rows each wave issues per chunk
    // one wave per SIMD has no partner wave to hide its LDS round trips, so it gets a second
    // transpose buffer and overlaps two transposes with each other and with arithmetic
    static constexpr int kTrBuffers = kDense ? 1 : 2;
    static constexpr int kWaveBytes = kTrBuffers * kWaveLdsBytes + kABytes;
    static constexpr int kTwOffset = WAVES * kWaveBytes;
    static constexpr int kRingOffset = kTwOffset + kTwBytes;
    static constexpr int kLdsBytes = kRingOffset + kSlots * kChunkBytes;
    static_assert(kSlots >= 2, "ring needs two slots");
    static_assert(kLdsBytes * PER_CU <= 163840, "LDS budget");
};
constexpr int kConvertWaves = 4;
constexpr int kConvertLdsBytes = kConvertWaves * kWavePrivBytes;
static_assert(4 * kRotStride * 4 <= kWaveLdsBytes, "rotation buffer must fit in the transpose buffer");

// One bootstrap: u = ca*in[a] + cb*in[b] + (0,..,0,cst), all wrapping u32.
struct PbsJob {
    uint32_t a, b;
    int32_t ca, cb;
    uint32_t cst;
    uint32_t pad[3];
};

struct c64 {
    double x, y;
};

// Orders this wave's LDS accesses for the compiler (the LDS executes one wave's instructions in
// order, so no hardware wait is needed); arithmetic may still be scheduled across it.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A fixed fp64 value written into a scalar register pair right where it is used.  Left to itself the compiler loads
// every constant of the step loop once, ahead of the loop; the folded transforms have more of them than there are scalar
// registers, and the surplus is parked in vector-register lanes and fetched back with v_readlane at every use.
#ifndef TFA_SCONST
#define TFA_SCONST 0
#endif
template <uint64_t BITS>
__device__ __forceinline__ double sconst() {
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((uint32_t)BITS), "i"((uint32_t)(BITS >> 32)));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
#if TFA_SCONST
#define TFA_K(x) sconst<__builtin_bit_cast(uint64_t, (double)(x))>()
#else
#define TFA_K(x) (x)
#endif

// ---- register DFT16 --------------------------------------------------------------------------
// S = +1: kernel e^{+2 pi i nk/16}; S = -1: its conjugate.
template <int S>
__device__ __forceinline__ void dft4(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2, double &r3, double &i3) {
    double t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    double t2r = r1 + r3, t2i = i1 + i3, dr = r1 - r3, di = i1 - i3;
    double t3r = S > 0 ? -di : di, t3i = S > 0 ? dr : -dr;
    r0 = t0r + t2r;
    i0 = t0i + t2i;
    r2 = t0r - t2r;
    i2 = t0i - t2i;
    r1 = t1r + t3r;
    i1 = t1i + t3i;
    r3 = t1r - t3r;
    i3 = t1i - t3i;
}
template <int S>
__device__ __forceinline__ void cmulc(double &r, double &i, double c, double s) {  // (r + i i) * (c + i S s)
    double nr = r * c - (S > 0 ? i * s : -(i * s));
    double ni = (S > 0 ? r * s : -(r * s)) + i * c;
    r = nr;
    i = ni;
}
template <int S>
__device__ __forceinline__ void dft16(double (&re)[16], double (&im)[16]) {
    constexpr double C1 = 0.9238795325112867, S1 = 0.3826834323650898, R = 0.7071067811865476;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4<S>(re[n2], im[n2], re[4 + n2], im[4 + n2], re[8 + n2], im[8 + n2], re[12 + n2], im[12 + n2]);
    // position 4*k1 + n2 now holds u[n2][k1]; twiddle by W16^(n2*k1)
    cmulc<S>(re[5], im[5], C1, S1);    // e = 1
    cmulc<S>(re[6], im[6], R, R);      // e = 2
    cmulc<S>(re[7], im[7], S1, C1);    // e = 3
    cmulc<S>(re[9], im[9], R, R);      // e = 2
    {                                  // e = 4: times S*i
        double r = re[10], i = im[10];
        re[10] = S > 0 ? -i : i;
        im[10] = S > 0 ? r : -r;
    }
    cmulc<S>(re[11], im[11], -R, R);    // e = 6
    cmulc<S>(re[13], im[13], S1, C1);   // e = 3
    cmulc<S>(re[14], im[14], -R, R);    // e = 6
    cmulc<S>(re[15], im[15], -C1, -S1); // e = 9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
        dft4<S>(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2], re[4 * k1 + 3], im[4 * k1 + 3]);
    // position 4*k1 + k2 holds X[k1 + 4*k2]: undo the digit reversal (register renaming only)
    double tr[16], ti[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        tr[k] = re[4 * (k & 3) + (k >> 2)];
        ti[k] = im[4 * (k & 3) + (k >> 2)];
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        re[k] = tr[k];
        im[k] = ti[k];
    }
}

// Forces the 16 values to exist in registers at this point of the program (an empty instruction that "modifies" them):
// keeps the compiler from evaluating the tail of a transform lazily inside the code that follows, which stretches the
// live ranges of the transform's inputs over that code.
__device__ __forceinline__ void pin16(double (&a)[16]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    asm volatile("" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
// e^{i pi n/32}, n = 0..15: the lane-independent part of the negacyclic twist
__device__ constexpr double kC32[16] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355,
                                        0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196023,
                                        0.4713967368259978, 0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077};
__device__ constexpr double kS32[16] = {0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898, 0.47139673682599764,
                                        0.5555702330196022, 0.6343932841636455, 0.7071067811865475, 0.773010453362737, 0.8314696123025452,
                                        0.8819212643483549, 0.9238795325112867, 0.9569403357322089, 0.9807852804032304, 0.9951847266721968};

// ---- register DFT16 with the twiddle scalings folded into the butterflies' fused multiply-adds ------------------
// A twiddle w = c (1 + i S tau) costs two instructions (the "1 + i S tau" part, rot_tan) instead of four: the real
// scale c rides on the multiply-add that consumes the value.
template <int S>
__device__ __forceinline__ void rot_tan(double &r, double &i, double tau) {  // (r + i i) * (1 + i S tau)
    const double st = S > 0 ? tau : -tau;
    const double nr = __builtin_fma(-st, i, r), ni = __builtin_fma(st, r, i);
    r = nr;
    i = ni;
}
// DFT4 of x0, c1 u1, c2 u2, c3 u3 (x0 unscaled), results in place; q = c3 / c1
template <int S>
__device__ __forceinline__ void dft4_scaled(double &r0, double &i0, double &r1, double &i1, double c1, double &r2, double &i2, double c2, double &r3,
                                            double &i3, double q) {
    const double sc = S > 0 ? c1 : -c1;
    const double t0r = __builtin_fma(c2, r2, r0), t0i = __builtin_fma(c2, i2, i0), t1r = __builtin_fma(-c2, r2, r0), t1i = __builtin_fma(-c2, i2, i0);
    const double t2r = __builtin_fma(q, r3, r1), t2i = __builtin_fma(q, i3, i1), dr = __builtin_fma(-q, r3, r1), di = __builtin_fma(-q, i3, i1);
    r0 = __builtin_fma(c1, t2r, t0r);
    i0 = __builtin_fma(c1, t2i, t0i);
    r2 = __builtin_fma(-c1, t2r, t0r);
    i2 = __builtin_fma(-c1, t2i, t0i);
    r1 = __builtin_fma(-sc, di, t1r);  // t1 + S i c1 d
    i1 = __builtin_fma(sc, dr, t1i);
    r3 = __builtin_fma(sc, di, t1r);
    i3 = __builtin_fma(-sc, dr, t1i);
}
// the same with x0 = c0 u0 scaled as well
template <int S>
__device__ __forceinline__ void dft4_allscaled(double &r0, double &i0, double c0, double &r1, double &i1, double c1, double &r2, double &i2, double c2,
                                               double &r3, double &i3, double q) {
    r0 *= c0;
    i0 *= c0;
    dft4_scaled<S>(r0, i0, r1, i1, c1, r2, i2, c2, r3, i3, q);
}
// TWIST (forward transforms only, S = +1): the inputs are first multiplied by e^{i pi n / 32}
template <int S, bool TWIST>
__device__ __forceinline__ void dft16m(double (&re)[16], double (&im)[16]) {
    constexpr double C1 = 0.9238795325112867, S1 = 0.3826834323650898, R = 0.7071067811865476;
    constexpr double T1 = S1 / C1, T3 = C1 / S1;  // tan(pi/8), tan(3 pi/8)
    static_assert(!TWIST || S > 0, "the twist belongs to the forward transform");
    static_for<4>([&](auto N2) __attribute__((always_inline)) {
        constexpr int n2 = N2.value;
        if constexpr (!TWIST) {
            dft4<S>(re[n2], im[n2], re[4 + n2], im[4 + n2], re[8 + n2], im[8 + n2], re[12 + n2], im[12 + n2]);
        } else {
            static_for<4>([&](auto A) __attribute__((always_inline)) {
                constexpr int n = n2 + 4 * A.value;
                if constexpr (n > 0) rot_tan<1>(re[n], im[n], TFA_K(kS32[n] / kC32[n]));
            });
            if constexpr (n2 == 0)
                dft4_scaled<1>(re[0], im[0], re[4], im[4], TFA_K(kC32[4]), re[8], im[8], TFA_K(kC32[8]), re[12], im[12], TFA_K(kC32[12] / kC32[4]));
            else
                dft4_allscaled<1>(re[n2], im[n2], TFA_K(kC32[n2]), re[4 + n2], im[4 + n2], TFA_K(kC32[4 + n2]), re[8 + n2], im[8 + n2],
                                  TFA_K(kC32[8 + n2]), re[12 + n2], im[12 + n2], TFA_K(kC32[12 + n2] / kC32[4 + n2]));
        }
    });
    // position 4*k1 + n2 now holds u[n2][k1]; the twiddles W16^(n2*k1) = c (1 + i S tau) go into the second layer
    dft4<S>(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
    // k1 = 1: e = 1, 2, 3
    rot_tan<S>(re[5], im[5], TFA_K(T1));
    rot_tan<S>(re[6], im[6], 1.0);
    rot_tan<S>(re[7], im[7], TFA_K(T3));
    dft4_scaled<S>(re[4], im[4], re[5], im[5], TFA_K(C1), re[6], im[6], TFA_K(R), re[7], im[7], TFA_K(S1 / C1));
    // k1 = 2: e = 2, 4 (times S i: a rotation by a quarter turn, no arithmetic), 6
    rot_tan<S>(re[9], im[9], 1.0);
    {
        const double r = re[10], i = im[10];
        re[10] = S > 0 ? -i : i;
        im[10] = S > 0 ? r : -r;
    }
    rot_tan<S>(re[11], im[11], -1.0);
    dft4_scaled<S>(re[8], im[8], re[9], im[9], TFA_K(R), re[10], im[10], 1.0, re[11], im[11], -1.0);
    // k1 = 3: e = 3, 6, 9
    rot_tan<S>(re[13], im[13], TFA_K(T3));
    rot_tan<S>(re[14], im[14], -1.0);
    rot_tan<S>(re[15], im[15], TFA_K(T1));
    dft4_scaled<S>(re[12], im[12], re[13], im[13], TFA_K(S1), re[14], im[14], TFA_K(-R), re[15], im[15], TFA_K(-C1 / S1));
    // position 4*k1 + k2 holds X[k1 + 4*k2]: undo the digit reversal (register renaming only)
    double tr[16], ti[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        tr[k] = re[4 * (k & 3) + (k >> 2)];
        ti[k] = im[4 * (k & 3) + (k >> 2)];
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        re[k] = tr[k];
        im[k] = ti[k];
    }
}
// Which DFT16s use the folded form, as a bit mask: 1 = first of a forward transform (after the twist), 2 = second of a
// forward transform, 4 = both of an inverse transform, 8 = the twist folded into the first forward DFT16 as well.
// All forms give the same values to rounding; measured on the 8-wave kernel the folded forms issue up to 120 fewer fp64
// instructions per step but make the compiler serialise the LDS reads of the step (one read, one wait), which costs
// more than the instructions save (mask 7: 167 k against 187 k bootstraps/s; mask 15 also spills 170 scalar registers).
// With the constants written into scalar registers where they are used (TFA_SCONST=1) nothing spills and the kernel
// needs 234 vector registers, and it is still slower (masks 8 / 7 / 15: 182 k / 178 k / 176 k against 188 k).  The
// multiply-adds themselves issue as fast as additions (4.2-4.4 clocks either way in tools/probe.hip), so the loss is in
// the schedule of the longer dependent chains, not in the instruction count.
#ifndef TFA_FMA_DFT
#define TFA_FMA_DFT 0
#endif
__device__ __forceinline__ void dft16_twisted(double (&re)[16], double (&im)[16]) {  // first DFT16 of a forward transform, twist included
    if constexpr ((TFA_FMA_DFT & 8) != 0) {
        dft16m<1, true>(re, im);
    } else {
#pragma unroll
        for (int n1 = 1; n1 < 16; n1++) cmulc<1>(re[n1], im[n1], kC32[n1], kS32[n1]);
        if constexpr ((TFA_FMA_DFT & 1) != 0)
            dft16m<1, false>(re, im);
        else
            dft16<1>(re, im);
    }
}
template <int S>
__device__ __forceinline__ void dft16_plain(double (&re)[16], double (&im)[16]) {  // the other DFT16s
    if constexpr ((TFA_FMA_DFT & (S > 0 ? 2 : 4)) != 0)
        dft16m<S, false>(re, im);
    else
        dft16<S>(re, im);
}

// 16 x 16 transpose inside each row of 16 lanes through the wave-private LDS buffer, one real
// plane at a time: lane t writes element r to [g][r][t] and reads back [g][t][r].
__device__ __forceinline__ void transpose16_plane(double (&v)[16], double *tr, int g, int t) {
    double *base = tr + g * kTrGroup;
#if TFA_TR_FENCES
    wave_lds_fence();  // earlier reads of the aliased buffer are done
#endif
#ifndef TFA_EXP_TR_NOWRITE  // timing experiments: one direction of the exchange only (results are garbage)
#pragma unroll
    for (int r = 0; r < 16; r++) base[tr_store_index(r, t)] = v[r];
#else
    {  // keeps the values alive at the price of integer operations and a single store
        uint32_t x = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) x ^= __builtin_bit_cast(uint2, v[r]).x ^ __builtin_bit_cast(uint2, v[r]).y;
        reinterpret_cast<uint32_t *>(base)[t] = x;
    }
#endif
#if TFA_TR_FENCES
    wave_lds_fence();
#endif
#ifndef TFA_EXP_TR_NOREAD
    {  // rows are 16-byte aligned: eight 16-byte reads
        typedef double v2d_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const v2d_ x = *reinterpret_cast<const v2d_ *>(base + tr_load_index(t, r));
            v[r] = x.x;
            v[r + 1] = x.y;
        }
    }
#endif
}
template <bool INVERSE = false>
__device__ __forceinline__ void transpose16(double (&re)[16], double (&im)[16], double *tr, int g, int t) {
#ifdef TFA_EXP_NOTRANSPOSE  // timing experiment: no exchange between the two DFT16 (results are garbage)
    return;
#endif
#ifdef TFA_EXP_NOTR_FWD  // ... in the forward / in the inverse transforms only
    if (!INVERSE) return;
#endif
#ifdef TFA_EXP_NOTR_INV
    if (INVERSE) return;
#endif
#if !TFA_TR_FENCES
    wave_lds_fence();  // orders the exchange after earlier accesses to the aliased rotation buffer (other element type)
#endif
    transpose16_plane(re, tr, g, t);
    transpose16_plane(im, tr, g, t);
}

// Forward negacyclic transform of one polynomial per row.  Input: the lane's 32 integer
// coefficients (re = coefficient 16*n1 + t, im = coefficient 16*n1 + t + 256).  Output: lane k1 = t
// holds frequency bins k1 + 16*k2, k2 = 0..15.
__device__ __forceinline__ void fft_forward(double (&re)[16], double (&im)[16], const c64 *twl, double *tr, int g, int t) {
    dft16_twisted(re, im);
#ifdef TFA_EXP_NOTWIDDLE  // timing experiment: one twiddle read per transform instead of sixteen (results are garbage)
    const c64 w0 = twl[t];
#endif
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
#ifdef TFA_EXP_NOTWIDDLE
        c64 w = w0;
#else
        c64 w = twl[k1 * 16 + t];
#endif
        cmulc<1>(re[k1], im[k1], w.x, w.y);
    }
    transpose16(re, im, tr, g, t);
    dft16_plain<1>(re, im);
}
// Inverse of fft_forward without the 1/256 (folded into the Fourier key).
__device__ __forceinline__ void fft_inverse(double (&re)[16], double (&im)[16], const c64 *twl, double *tr, int g, int t) {
    dft16_plain<-1>(re, im);
#ifdef TFA_EXP_NOTWIDDLE
    const c64 w0 = twl[t];
#endif
    transpose16<true>(re, im, tr, g, t);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
#ifdef TFA_EXP_NOTWIDDLE
        c64 w = w0;
#else
        c64 w = twl[k1 * 16 + t];
#endif
        cmulc<-1>(re[k1], im[k1], w.x, w.y);
    }
    dft16_plain<-1>(re, im);
#pragma unroll
    for (int n1 = 1; n1 < 16; n1++) cmulc<-1>(re[n1], im[n1], kC32[n1], kS32[n1]);
}

// Two forward transforms on one transpose buffer: every twiddle is read once for both, and the two independent
// dependency chains interleave.  Used by the two-waves-per-SIMD variant.
__device__ __forceinline__ void fft_forward_pair(double (&ar)[16], double (&ai)[16], double (&br)[16], double (&bi)[16], const c64 *twl, double *tr,
                                                 int g, int t) {
    dft16_twisted(ar, ai);
    dft16_twisted(br, bi);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(ar[k1], ai[k1], w.x, w.y);
        cmulc<1>(br[k1], bi[k1], w.x, w.y);
    }
    transpose16(ar, ai, tr, g, t);
    transpose16(br, bi, tr, g, t);
    dft16_plain<1>(ar, ai);
    dft16_plain<1>(br, bi);
}
// Two forward transforms on one transpose buffer, ordered so that each one's exchange travels behind the other's
// arithmetic: twiddles and exchange of a, twiddles of b (cover for a), exchange of b, second DFT16 of a (cover for b).
__device__ __forceinline__ void fft_forward_pair_ordered(double (&ar)[16], double (&ai)[16], double (&br)[16], double (&bi)[16], const c64 *twl,
                                                         double *tr, int g, int t) {
    dft16_twisted(ar, ai);
    dft16_twisted(br, bi);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(ar[k1], ai[k1], w.x, w.y);
    }
    transpose16(ar, ai, tr, g, t);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(br[k1], bi[k1], w.x, w.y);
    }
    pin16(br), pin16(bi);  // b's twiddle products are done (they are a's cover) before b's exchange starts
    transpose16(br, bi, tr, g, t);
    dft16_plain<1>(ar, ai);
    pin16(ar), pin16(ai);
    dft16_plain<1>(br, bi);
}
// Two forward transforms on one transpose buffer where each exchange travels behind arithmetic that needs no LDS data:
// a's exchange behind b's twist and first DFT16 (constants only), b's exchange behind a's second DFT16.  (The LDS
// returns a wave's reads in order, so a cover that itself waits for LDS reads - b's twiddles - would expose a's exchange.)
__device__ __forceinline__ void fft_forward_pair_cover(double (&ar)[16], double (&ai)[16], double (&br)[16], double (&bi)[16], const c64 *twl,
                                                       double *tr, int g, int t) {
    dft16_twisted(ar, ai);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(ar[k1], ai[k1], w.x, w.y);
    }
    transpose16(ar, ai, tr, g, t);
    dft16_twisted(br, bi);
    pin16(br), pin16(bi);  // b's first DFT16 is done (it is a's cover) before b's twiddles are read
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(br[k1], bi[k1], w.x, w.y);
    }
    transpose16(br, bi, tr, g, t);
    dft16_plain<1>(ar, ai);
    pin16(ar), pin16(ai);
    dft16_plain<1>(br, bi);
}
// Two independent forward transforms interleaved (two transpose buffers): the LDS round trips of one
// overlap the arithmetic of the other.  Used by the one-wave-per-SIMD variant.
__device__ __forceinline__ void fft_forward2(double (&ar)[16], double (&ai)[16], double (&br)[16], double (&bi)[16], const c64 *twl, double *trA,
                                             double *trB, int g, int t) {
    dft16_twisted(ar, ai);
    dft16_twisted(br, bi);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(ar[k1], ai[k1], w.x, w.y);
        cmulc<1>(br[k1], bi[k1], w.x, w.y);
    }
    double *pa = trA + g * kTrGroup, *pb = trB + g * kTrGroup;
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) pa[tr_store_index(r, t)] = ar[r];
#pragma unroll
    for (int r = 0; r < 16; r++) pb[tr_store_index(r, t)] = br[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) ar[r] = pa[tr_load_index(t, r & ~1) + (r & 1)];
#pragma unroll
    for (int r = 0; r < 16; r++) br[r] = pb[tr_load_index(t, r & ~1) + (r & 1)];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) pa[tr_store_index(r, t)] = ai[r];
#pragma unroll
    for (int r = 0; r < 16; r++) pb[tr_store_index(r, t)] = bi[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) ai[r] = pa[tr_load_index(t, r & ~1) + (r & 1)];
#pragma unroll
    for (int r = 0; r < 16; r++) bi[r] = pb[tr_load_index(t, r & ~1) + (r & 1)];
    dft16_plain<1>(ar, ai);
    dft16_plain<1>(br, bi);
}
// Inverse transform with the real and imaginary planes in separate buffers (no wait between planes).
__device__ __forceinline__ void fft_inverse2buf(double (&re)[16], double (&im)[16], const c64 *twl, double *trA, double *trB, int g, int t) {
    dft16_plain<-1>(re, im);
    double *pa = trA + g * kTrGroup, *pb = trB + g * kTrGroup;
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) pa[tr_store_index(r, t)] = re[r];
#pragma unroll
    for (int r = 0; r < 16; r++) pb[tr_store_index(r, t)] = im[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) re[r] = pa[tr_load_index(t, r & ~1) + (r & 1)];
#pragma unroll
    for (int r = 0; r < 16; r++) im[r] = pb[tr_load_index(t, r & ~1) + (r & 1)];
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<-1>(re[k1], im[k1], w.x, w.y);
    }
    dft16_plain<-1>(re, im);
#pragma unroll
    for (int n1 = 1; n1 < 16; n1++) cmulc<-1>(re[n1], im[n1], kC32[n1], kS32[n1]);
}

// ---- cross-row reduction ---------------------------------------------------------------------
// (Measured alternative: routing these exchanges through the LDS crossbar with ds_swizzle /
// ds_bpermute frees the vector pipe but runs 12-40 % slower: the exchanges sit on the dependency
// chain of every frequency group and at most 15 LDS operations can be in flight per wave.)
// x[lanes 32..63] <-> y[lanes 0..31]
__device__ __forceinline__ void swap_halves(double &x, double &y) {
    uint2 a = __builtin_bit_cast(uint2, x), b = __builtin_bit_cast(uint2, y);
    auto lo = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    x = __builtin_bit_cast(double, uint2{lo[0], hi[0]});
    y = __builtin_bit_cast(double, uint2{lo[1], hi[1]});
}
// x[odd rows] <-> y[even rows]
__device__ __forceinline__ void swap_rows(double &x, double &y) {
    uint2 a = __builtin_bit_cast(uint2, x), b = __builtin_bit_cast(uint2, y);
    auto lo = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    x = __builtin_bit_cast(double, uint2{lo[0], hi[0]});
    y = __builtin_bit_cast(double, uint2{lo[1], hi[1]});
}
// Sum over the four polynomials of the partial products, scattered so that polynomial g's lanes end with output g.
// Row mode: each row g holds partial sums p[c] for outputs c = 0..3.  Quad mode: lane g's p[j] belongs to output
// j ^ g (the key stores its columns in that order per lane), which makes the exchange the same in every lane:
// two values from the neighbour at distance 2, then one from the neighbour at distance 1.
// (the permuted value is built in place: old = source ties the destination to the source's registers)
__device__ __forceinline__ double quad_from(double x, int ctrl_is_xor1) {
    uint2 a = __builtin_bit_cast(uint2, x);
    if (ctrl_is_xor1) {
        a.x = (uint32_t)__builtin_amdgcn_update_dpp((int)a.x, (int)a.x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        a.y = (uint32_t)__builtin_amdgcn_update_dpp((int)a.y, (int)a.y, 0xB1, 0xF, 0xF, false);
    } else {
        a.x = (uint32_t)__builtin_amdgcn_update_dpp((int)a.x, (int)a.x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
        a.y = (uint32_t)__builtin_amdgcn_update_dpp((int)a.y, (int)a.y, 0x4E, 0xF, 0xF, false);
    }
    return __builtin_bit_cast(double, a);
}
__device__ __forceinline__ double reduce_scatter_rows(double p0, double p1, double p2, double p3) {
    if constexpr (kQuad && !TFA_EXP_QUAD_SWAPS) {
        const double sa = p0 + quad_from(p2, 0), sb = p1 + quad_from(p3, 0);
        return sa + quad_from(sb, 1);
    } else {
        swap_halves(p0, p2);
        swap_halves(p1, p3);
        double sa = p0 + p2, sb = p1 + p3;
        swap_rows(sa, sb);
        return sa + sb;
    }
}

// ---- pinned LDS reads of key words ------------------------------------------------------------
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_offset(const void *p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }  // low half of a flat LDS address
// Reads the 16-byte key word BYTES past this lane's slot address.  dep0 / dep1 are not used by the instruction: they
// make it wait, in program order, for the values whose computation released the destination registers.  The compiler
// does not count these reads: key_wait() must come before the first use of their results.
template <int BYTES>
__device__ __forceinline__ v2d key_read_at(uint32_t addr, double dep0, double dep1) {
#ifdef TFA_EXP_NOKEYREAD  // timing experiment: no LDS read (results are garbage)
    return v2d{dep0 + 1.0, dep1 - 1.0};
#else
    v2d v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(BYTES), "v"(dep0), "v"(dep1));
    return v;
#endif
}
__device__ __forceinline__ void key_wait(v2d &a, v2d &b, v2d &c, v2d &d, v2d &e, v2d &f, v2d &g, v2d &h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
// the four words named have arrived when at most LATER reads issued after them are still travelling
template <int LATER>
__device__ __forceinline__ void key_wait_four(v2d &a, v2d &b, v2d &c, v2d &d) {
    if constexpr (LATER == 4)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// ---- exchange with pinned reads ------------------------------------------------------------------------------
// The 16 x 16 exchange of one real plane with the row reads issued as inline asm right behind the stores: the compiler
// would sink plain loads towards their first use and then wait for them with nothing left to cover the round trip.  It
// does not count these reads: exchange_wait() must come before the first use of the values.  rd[w] = LDS byte address of
// this lane's row word holding elements 2w, 2w+1 (tr_load_index), loop invariant.
struct ExchangeAddr {
    uint32_t rd[8];
};
__device__ __forceinline__ ExchangeAddr exchange_addr(const double *tr, int g, int t) {
    ExchangeAddr a;
#pragma unroll
    for (int w = 0; w < 8; w++) a.rd[w] = lds_offset(tr + g * kTrGroup + tr_load_index(t, 2 * w));
    return a;
}
__device__ __forceinline__ v2d lds_read_b128_ordered(uint32_t addr) {  // after every earlier store of this wave, before every later one
    v2d v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// `tie` is a value the covering arithmetic starts from: making the block "modify" it keeps the compiler from starting the
// cover before the reads are on their way.
__device__ __forceinline__ void exchange_issue_plane(double (&v)[16], double *tr, int g, int t, const ExchangeAddr &a, double &tie) {
    double *base = tr + g * kTrGroup;
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) base[tr_store_index(r, t)] = v[r];
    wave_lds_fence();
    v2d x0, x1, x2, x3, x4, x5, x6, x7;
    asm volatile(
        "ds_read_b128 %0, %9\n\tds_read_b128 %1, %10\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %12\n\t"
        "ds_read_b128 %4, %13\n\tds_read_b128 %5, %14\n\tds_read_b128 %6, %15\n\tds_read_b128 %7, %16"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "+v"(tie)
        : "v"(a.rd[0]), "v"(a.rd[1]), "v"(a.rd[2]), "v"(a.rd[3]), "v"(a.rd[4]), "v"(a.rd[5]), "v"(a.rd[6]), "v"(a.rd[7])
        : "memory");
    v[0] = x0.x, v[1] = x0.y, v[2] = x1.x, v[3] = x1.y, v[4] = x2.x, v[5] = x2.y, v[6] = x3.x, v[7] = x3.y;
    v[8] = x4.x, v[9] = x4.y, v[10] = x5.x, v[11] = x5.y, v[12] = x6.x, v[13] = x6.y, v[14] = x7.x, v[15] = x7.y;
}
__device__ __forceinline__ void exchange_wait(double (&a)[16], double (&b)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]),
                 "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]));
}
// Two forward transforms, exchanges issued early and covered by LDS-free arithmetic of the other transform.
__device__ __forceinline__ void fft_forward_pair_pinned(double (&ar)[16], double (&ai)[16], double (&br)[16], double (&bi)[16], const c64 *twl,
                                                        double *tr, int g, int t, const ExchangeAddr &xa) {
    dft16_twisted(ar, ai);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(ar[k1], ai[k1], w.x, w.y);
    }
    exchange_issue_plane(ar, tr, g, t, xa, br[1]);
    exchange_issue_plane(ai, tr, g, t, xa, bi[1]);
    dft16_twisted(br, bi);  // cover: constants only
    pin16(br), pin16(bi);
    exchange_wait(ar, ai);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        c64 w = twl[k1 * 16 + t];
        cmulc<1>(br[k1], bi[k1], w.x, w.y);
    }
    exchange_issue_plane(br, tr, g, t, xa, ar[0]);
    exchange_issue_plane(bi, tr, g, t, xa, ai[0]);
    dft16_plain<1>(ar, ai);  // cover
    pin16(ar), pin16(ai);
    exchange_wait(br, bi);
    dft16_plain<1>(br, bi);
}

// Inverse transform whose sixteen twiddles are read (pinned, uncounted by the compiler) before the first register
// DFT16, behind the values dep0 / dep1, and so arrive long before the twiddle products need them.
__device__ __forceinline__ void fft_inverse_pf(double (&re)[16], double (&im)[16], const c64 *twl, double *tr, int g, int t) {
    v2d w[16];
    const uint32_t waddr = lds_offset(twl + t);
    static_for<16>([&](auto K) __attribute__((always_inline)) { w[K.value] = key_read_at<K.value * 256>(waddr, re[15], im[15]); });
    dft16_plain<-1>(re, im);
    transpose16<true>(re, im, tr, g, t);
    key_wait(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    key_wait(w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) cmulc<-1>(re[k1], im[k1], w[k1].x, w[k1].y);
    dft16_plain<-1>(re, im);
#pragma unroll
    for (int n1 = 1; n1 < 16; n1++) cmulc<-1>(re[n1], im[n1], kC32[n1], kS32[n1]);
}

// ---- integer helpers -------------------------------------------------------------------------
__device__ __forceinline__ uint32_t modswitch_2n(uint32_t x) {  // closest multiple of q/1024, index mod 1024
    return (((x >> 21) + 1u) >> 1) & 1023u;
}
// Signed decomposition, base B = 2^10, 2 levels, of the closest multiple of q / 2^20: d1 pairs with q / 2^10
// (key level 0), d2 with q / 2^20 (key level 1).  tfhe-rs extracts digits from the least significant end:
// res = state mod B, state = state / B, carry = res > B/2 || (res == B/2 && bit 9 of state), digit = res - carry * B.
// Written on the whole word: the carry is bit 10 of res + 511 + bit 19 of the state.
__device__ __forceinline__ uint32_t decompose_closest(uint32_t x) { return (((x >> 11) + 1u) >> 1) & 0xFFFFFu; }
__device__ __forceinline__ void decompose_low_of(uint32_t s, int &d2, uint32_t &state) {  // s = decompose_closest(x)
    const uint32_t u = s + 511u + (s >> 19);
    d2 = (int)(s - (u & ~1023u));
    state = u >> 10;  // <= 1024
}
// second digit: the state above it is 0 or 1 (bit 9 clear), and 1 only with res == 0
__device__ __forceinline__ int decompose_high(uint32_t state) { return (int)(state - ((state + 511u) & ~1023u)); }
__device__ __forceinline__ void decompose2(uint32_t x, int &d1, int &d2) {
    uint32_t state;
    decompose_low_of(decompose_closest(x), d2, state);
    d1 = decompose_high(state);
}
// Both digits of sign * v - acc, sign = -1 where bit 9 of the rotated index jj is set (X^N = -1), written for the fewest
// integer instructions: y = x + 2^11 carries the rounding of decompose_closest, so s = y >> 12 and its bit 19 is bit 31
// of y; w = u + 511 * 2^10 turns "state - ((state + 511) & ~1023)" into a bit field minus 511.  Same digits as
// decompose2(sign * v - acc), 18 instead of 22 instructions per coefficient with the rotation and the conversions.
__device__ __forceinline__ void rotated_digits(uint32_t v, int jj, uint32_t acc, int &d1, int &d2) {
    const uint32_t neg = (uint32_t)((jj << 22) >> 31);           // 0 or ~0
    const uint32_t y = (v ^ neg) - (acc + neg - 2048u);         // (v ^ neg) - neg = sign * v
    const uint32_t s = y >> 12, b19 = y >> 31;
    const uint32_t u = s + 511u + b19;
    d2 = (int)(s - (u & ~1023u));
    d1 = (int)(((u + (511u << 10)) >> 10) & 1023u) - 511;
}
// Nearest integer mod 2^32 of an inverse-transform output.  |x| < 2^51 always holds here: x is a sum of
// 4096 products of a digit (|d| <= 2^9) and a key word (|w| < 2^31), whose standard deviation is about
// 2^44.  Adding 1.5 * 2^52 then leaves round-to-nearest-even(x) as a two's complement integer in the
// mantissa, and the low word is the torus value.
__device__ __forceinline__ uint32_t round_torus(double x) {
    x += 6755399441055744.0;
    return __builtin_bit_cast(uint2, x).x;
}

// bsk_f layout (c64): [i][level][p][c][row g][t]: one (level, p, c) row is 1 KiB = the 64 lanes of a wave.  Position p
// holds frequency group k2 = kperm(p) = 0, 4, 8, 12, 1, 5, ...: the order in which the last radix-4 layer of the forward
// transform produces the groups and the first layer of the inverse transform consumes them, so the products stream
// from one transform into the other without the whole spectrum being live.
#ifndef TFA_K2_PERM
#define TFA_K2_PERM 1
#endif
__host__ __device__ constexpr int kperm(int p) { return TFA_K2_PERM ? 4 * (p & 3) + (p >> 2) : p; }  // an involution
__device__ __forceinline__ size_t bsk_index(uint32_t i, int lv, int p, int c) { return ((((size_t)i * kLevels + lv) * 16 + p) * kK1 + c) * 64; }

// Ring chunks.  A chunk is 16 key rows of 1 KiB.  With 4 waves a chunk holds both levels of two
// frequency groups (the two levels are multiplied and reduced together); with 8 waves it holds one
// level of four frequency groups (each level is reduced on its own, which halves the live registers).
template <class G>
__device__ __forceinline__ size_t chunk_row_index(uint32_t chunk, int row) {
    const uint32_t i = chunk / G::kChunksPerStep, q = chunk % G::kChunksPerStep;
    if constexpr (G::kSplitLevels) {
        constexpr int kPerLevel = G::kChunksPerStep / 2, kGroups = G::kChunkRows / 4;  // chunks per level, frequency groups per chunk
        return bsk_index(i, q / kPerLevel, kGroups * (q % kPerLevel) + (row >> 2), row & 3);
    } else {
        return bsk_index(i, (row >> 2) & 1, (G::kChunkRows / 8) * q + (row >> 3), row & 3);
    }
}
// LDS-DMA of chunk `chunk` (global index over all steps) into its ring slot, 16 / WAVES rows per wave.
template <class G>
__device__ __forceinline__ void ring_issue(const c64 *bsk_f, uint32_t chunk, unsigned char *ring, int wave, int lane) {
#ifdef TFA_EXP_NODMA  // timing experiment: no key traffic at all (results are garbage)
    return;
#endif
    unsigned char *dst = ring + (chunk % G::kSlots) * G::kChunkBytes;
#pragma unroll
    for (int j = 0; j < G::kRowsPerWave; j++) {
        const int row = wave * G::kRowsPerWave + j;
        // wave-uniform row pointer plus a 32-bit lane offset: lets the load use its scalar-base form instead of a
        // 64-bit per-lane address that would have to stay in (or be reloaded into) a register pair
        const char *row_ptr = reinterpret_cast<const char *>(bsk_f + chunk_row_index<G>(chunk, row));
        auto gsrc = (const __attribute__((address_space(1))) void *)(reinterpret_cast<uintptr_t>(row_ptr + (uint32_t)(lane * 16)));
        auto ldst = (__attribute__((address_space(3))) void *)(reinterpret_cast<uintptr_t>(dst + row * 1024));
        __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 0, 0);
    }
}
// Chunk hand-over, in three steps so diagnostic builds can time them apart:
//   ring_wait_own : this wave's DMA rows of chunk `ch` have landed (and, with 4 waves, its LDS reads of the
//                   chunk whose slot is about to be refilled have returned)
//   s_barrier     : all rows of `ch` are visible and every wave is done with chunk ch-1
//   ring_refill   : the slot freed by ch-1 is refilled with chunk ch-1+kSlots
template <class G>
__device__ __forceinline__ void ring_wait_own(uint32_t ch, uint32_t n_chunks) {
    if constexpr (!G::kDense) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // at most the rows of the kSlots-2 younger chunks stay in flight
    if (ch + (G::kSlots - 2) < n_chunks) {
        constexpr int kInFlight = (G::kSlots - 2) * G::kRowsPerWave;
        static_assert(kInFlight < 16, "vmcnt immediate below assumes the low four bits");
        __builtin_amdgcn_s_waitcnt(0x0F70 | kInFlight);  // vmcnt(kInFlight), expcnt and lgkmcnt untouched
        asm volatile("" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <class G>
__device__ __forceinline__ const c64 *ring_refill(const c64 *bsk_f, uint32_t ch, uint32_t n_chunks, unsigned char *ring, int wave, int lane) {
    if (ch + (G::kSlots - 1) < n_chunks) ring_issue<G>(bsk_f, ch + (G::kSlots - 1), ring, wave, lane);
    return reinterpret_cast<const c64 *>(ring + (ch % G::kSlots) * G::kChunkBytes) + lane;
}
#ifdef TFA_EXP_NOBARRIER  // timing experiment: waves run free (results are garbage)
#define TFA_RING_BARRIER() ((void)0)
#else
#define TFA_RING_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#define TFA_RING_ACQUIRE(slot_var, ch_expr)                                     \
    const c64 *slot_var;                                                        \
    {                                                                           \
        const uint32_t ch_ = (ch_expr);                                         \
        TFA_STAMP(8);                                                           \
        ring_wait_own<G>(ch_, n_chunks);                                    \
        TFA_STAMP(9);                                                           \
        TFA_RING_BARRIER();                                                     \
        TFA_STAMP(10);                                                          \
        slot_var = ring_refill<G>(bsk_f, ch_, n_chunks, ring, wave, lane);  \
        TFA_STAMP(11);                                                          \
    }

// tw: [r][t] c64, e^{i pi t (1 + 4 r)/512}
//
// Workgroup = WAVES waves = WAVES bootstraps that walk the blind rotation in lock step so that they
// share the bootstrapping key: step i's Fourier GGSW streams through the LDS ring once per workgroup.
// WAVES = 8 puts two waves on every SIMD (throughput: one hides the other's LDS and barrier waits);
// WAVES = 4 keeps one wave per SIMD (lowest latency per bootstrap, used for narrow stages).
// Chunk schedule (global chunk index ch = 8*i + q): before consuming ch every wave waits for its own
// DMA rows of ch, then one barrier makes all rows of ch visible and proves every wave is done with
// ch-1, whose slot is refilled with ch-1+kSlots right after the barrier.
template <int WAVES, int PER_CU = 1>
__global__ void __launch_bounds__(WAVES * 64, WAVES * PER_CU / 4)
    pbs_kernel(const uint32_t *__restrict__ in_base, uint32_t in_stride, const PbsJob *__restrict__ jobs, uint32_t n_jobs,
               const c64 *__restrict__ bsk_f, const c64 *__restrict__ tw, uint32_t n, uint32_t steps, uint32_t *__restrict__ big_out,
               uint32_t big_stride, uint32_t *__restrict__ acc_out) {
    using G = PbsGeom<WAVES, PER_CU>;
    extern __shared__ __align__(16) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep it (and all ring addressing) scalar
    const int lane = threadIdx.x & 63, g = lane_g(lane), t = lane_t(lane);
    const uint32_t job_raw = blockIdx.x * WAVES + wave;
    const bool live = job_raw < n_jobs;                 // idle waves of the last workgroup still run the
    const uint32_t job = live ? job_raw : n_jobs - 1;   // loop (barriers, DMA) on a duplicate job, write nothing
    unsigned char *wl = lds + wave * G::kWaveBytes;
    unsigned char *ring = lds + G::kRingOffset;
    c64 *twl = reinterpret_cast<c64 *>(lds + G::kTwOffset);
    double *tr = reinterpret_cast<double *>(wl);
    double *tr2 = reinterpret_cast<double *>(wl + (G::kTrBuffers - 1) * kWaveLdsBytes);  // second transpose buffer (4 waves)
    uint32_t *rot = reinterpret_cast<uint32_t *>(wl);
    uint16_t *a_tilde = reinterpret_cast<uint16_t *>(wl + G::kTrBuffers * kWaveLdsBytes);

    // gate linear combination + modulus switch
    const PbsJob jb = jobs[job];
    const uint32_t *pa = in_base + (size_t)jb.a * in_stride, *pb = in_base + (size_t)jb.b * in_stride;
    for (uint32_t idx = lane; idx <= n; idx += 64) {
        uint32_t u = (uint32_t)jb.ca * pa[idx] + (uint32_t)jb.cb * pb[idx];
        if (idx == n) u += jb.cst;
        a_tilde[idx] = (uint16_t)modswitch_2n(u);
    }
    if (threadIdx.x < 256) twl[threadIdx.x] = tw[threadIdx.x];
    __syncthreads();
    const uint32_t b_tilde = a_tilde[n];
#if TFA_FFT_PAIR == 4
    const ExchangeAddr xaddr = exchange_addr(tr, g, t);
#endif

    // accumulator = X^{-b~} * (0, 0, 0, 2^29 * (1 + X + ... + X^{N-1}))
    uint32_t acc[32];
#pragma unroll
    for (int m = 0; m < 32; m++) {
        uint32_t jj = (uint32_t)(16 * m + t) + b_tilde;  // < 3N
        acc[m] = (g == 3) ? (((jj >> 9) & 1u) ? 0xE0000000u : 0x20000000u) : 0u;
    }

    const uint32_t n_steps = steps < n ? steps : n;
    const uint32_t n_chunks = n_steps * G::kChunksPerStep;
    // every ordinary load above has been consumed; from here on the only vector-memory operations in
    // flight are the ring's LDS-DMA rows, so vmcnt counts them exactly
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t ch = 0; ch < G::kSlots - 1 && ch < n_chunks; ch++) ring_issue<G>(bsk_f, ch, ring, wave, lane);

#ifdef TFA_EXP_SKEW  // timing experiment (with NODMA + NOBARRIER): the second wave of every SIMD starts about half a step late
    if (G::kDense && WAVES == 8 && wave >= 4) {
        double xr[16], xi[16];
#pragma unroll
        for (int q = 0; q < 16; q++) xr[q] = (double)acc[q], xi[q] = (double)acc[q + 16];
        for (int rep = 0; rep < TFA_EXP_SKEW; rep++) fft_forward(xr, xi, twl, tr, g, t);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] += round_torus(xr[q]) + round_torus(xi[q]);
    }
#endif
#ifdef TFA_EXP_PRIO  // timing experiment: the second wave of every SIMD issues with priority
    if (WAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(TFA_EXP_PRIO);
#endif
    TFA_STAMP_DECL
    for (uint32_t i = 0; i < n_steps; i++) {
        TFA_STAMP(0);
        const int a = __builtin_amdgcn_readfirstlane((int)a_tilde[i]);
        // d = acc * X^a - acc through the rotation buffer, then signed decomposition (a == 0 gives d = 0)
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < 32; m++) rot[g * kRotStride + 16 * m + t] = acc[m];
        wave_lds_fence();
        uint32_t dig[32];                           // 8 waves: level-1 digit in the low half, level-2 digit in the high half
        double f1r[16], f1i[16], f2r[16], f2i[16];  // 4 waves: both levels' digits
#pragma unroll
        for (int m = 0; m < 32; m++) {
            int jj = 16 * m + t - a;  // (-1024, 512]
            const uint32_t v = rot[g * kRotStride + (jj & 511)];
            int d1, d2;
            rotated_digits(v, jj, acc[m], d1, d2);
            if constexpr (G::kSplitLevels) {
                dig[m] = ((uint32_t)d1 & 0xFFFFu) | ((uint32_t)d2 << 16);
            } else {
                (m < 16 ? f1r[m & 15] : f1i[m & 15]) = (double)d1;
                (m < 16 ? f2r[m & 15] : f2i[m & 15]) = (double)d2;
            }
        }
        TFA_STAMP(1);  // rotate + decompose
        // Fourier-domain multiply-accumulate: row g contributes polynomial g of both levels to all
        // four output polynomials; the sum over rows lands in row c for output polynomial c.
        double our[16], oui[16];
        if constexpr (G::kSplitLevels) {
#pragma unroll
            for (int q = 0; q < 16; q++) our[q] = oui[q] = 0.0;
            // a real (not unrolled) loop over the two levels keeps one level's transform live at a time
#pragma nounroll
            for (int lv = 0; lv < 2; lv++) {
                const int sh = lv ? 0 : 16;
                double fr[16], fi[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    fr[q] = (double)((int)(dig[q] << sh) >> 16);
                    fi[q] = (double)((int)(dig[q + 16] << sh) >> 16);
                }
                TFA_STAMP(2);
                fft_forward(fr, fi, twl, tr, g, t);
                TFA_STAMP(3);  // forward transform
                constexpr int kPerLevel = G::kChunksPerStep / 2, kGroups = G::kChunkRows / 4;
#pragma unroll
                for (int qq = 0; qq < kPerLevel; qq++) {
                    TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep + lv * kPerLevel + qq)
                    TFA_STAMP(4);  // ring wait + barrier + refill issue
#pragma unroll
                    for (int kq = 0; kq < kGroups; kq++) {
                        const int k2 = kperm(kGroups * qq + kq);
                        double pr[4], pi[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            c64 b = slot[(kq * 4 + c) * 64];
                            pr[c] = fr[k2] * b.x - fi[k2] * b.y;
                            pi[c] = fr[k2] * b.y + fi[k2] * b.x;
                        }
                        our[k2] += reduce_scatter_rows(pr[0], pr[1], pr[2], pr[3]);
                        oui[k2] += reduce_scatter_rows(pi[0], pi[1], pi[2], pi[3]);
                    }
                    TFA_STAMP(5);  // key products + row reduction of one chunk
                }
            }
        } else if constexpr (G::kDense && TFA_PBS8_PREFETCH) {
            // Both levels transformed one after the other.  Key words are read from the ring one frequency group ahead,
            // each pair of words (column c, both levels) into the registers that the products of column c have just
            // released, so the reads travel behind the remaining products and the row reduction and cost no extra
            // registers.  The reads are inline asm pinned after those products by (unused) operands; the compiler does
            // not count them, so every group starts with an explicit wait.
#if TFA_FFT_PAIR == 4
            fft_forward_pair_pinned(f1r, f1i, f2r, f2i, twl, tr, g, t, xaddr);
#elif TFA_FFT_PAIR == 3
            fft_forward_pair_cover(f1r, f1i, f2r, f2i, twl, tr, g, t);
#elif TFA_FFT_PAIR == 2
            fft_forward_pair_ordered(f1r, f1i, f2r, f2i, twl, tr, g, t);
#elif TFA_FFT_PAIR
            fft_forward_pair(f1r, f1i, f2r, f2i, twl, tr, g, t);
#else
            fft_forward(f1r, f1i, twl, tr, g, t);
            fft_forward(f2r, f2i, twl, tr, g, t);
#endif
#if TFA_PIN_SPECTRA
            pin16(f1r), pin16(f1i), pin16(f2r), pin16(f2i);
#endif
            TFA_STAMP(3);
            constexpr int kPerChunk = G::kChunkRows / 8;  // frequency groups per chunk
            v2d kb[8];                                    // [level * 4 + column]
            uint32_t kaddr;
#if TFA_MAC_STAGGER
            // Columns in two pairs: the key registers of a pair are refilled with the next group's words as soon as the
            // pair's products are done, and each pair waits only for its own four words (the LDS returns reads in
            // order), so every read has the other pair's products and the row reduction to travel behind.
            {
                TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep)
                kaddr = lds_offset(slot);
                static_for<4>([&](auto C) __attribute__((always_inline)) {
                    kb[C.value] = key_read_at<C.value * 1024>(kaddr, 0.0, 0.0);
                    kb[4 + C.value] = key_read_at<(4 + C.value) * 1024>(kaddr, 0.0, 0.0);
                });
            }
            static_for<16>([&](auto POS) __attribute__((always_inline)) {
                constexpr int pos = POS.value, k2 = kperm(pos), kk = (pos + 1) % kPerChunk;
                constexpr bool kMore = pos + 1 < 16, kOpens = kMore && kk == 0;
                if constexpr (kOpens) {  // next group opens the next chunk: this wave must hold all of the current one first
                    key_wait(kb[0], kb[1], kb[2], kb[3], kb[4], kb[5], kb[6], kb[7]);
                    TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep + (pos + 1) / kPerChunk)
                    kaddr = lds_offset(slot);
                } else {
                    key_wait_four<4>(kb[0], kb[4], kb[1], kb[5]);
                }
                double pr[4], pi[4];
                static_for<2>([&](auto H) __attribute__((always_inline)) {
                    constexpr int h = H.value;
                    if constexpr (h == 1 && !kOpens) key_wait_four<kMore ? 4 : 0>(kb[2], kb[6], kb[3], kb[7]);
                    static_for<2>([&](auto C) __attribute__((always_inline)) {
                        constexpr int c = 2 * h + C.value;
                        const v2d b0 = kb[c], b1 = kb[4 + c];
                        pr[c] = f1r[k2] * b0.x - f1i[k2] * b0.y + f2r[k2] * b1.x - f2i[k2] * b1.y;
                        pi[c] = f1r[k2] * b0.y + f1i[k2] * b0.x + f2r[k2] * b1.y + f2i[k2] * b1.x;
                    });
                    if constexpr (kMore) {
                        static_for<2>([&](auto C) __attribute__((always_inline)) {
                            constexpr int c = 2 * h + C.value;
                            kb[c] = key_read_at<(kk * 8 + c) * 1024>(kaddr, pr[2 * h], pi[2 * h + 1]);
                            kb[4 + c] = key_read_at<(kk * 8 + 4 + c) * 1024>(kaddr, pr[2 * h + 1], pi[2 * h]);
                        });
                    }
                    // the second pair's products start behind the first pair's reads
                    if constexpr (h == 0) asm volatile("" : "+v"(f1r[k2]), "+v"(f1i[k2]), "+v"(f2r[k2]), "+v"(f2i[k2]));
                });
                our[k2] = reduce_scatter_rows(pr[0], pr[1], pr[2], pr[3]);
                oui[k2] = reduce_scatter_rows(pi[0], pi[1], pi[2], pi[3]);
#if TFA_PIN_GROUP
                // the group's sums exist before the next group's waits: keeps two groups' partial sums from being live together
                asm volatile("" : "+v"(our[k2]), "+v"(oui[k2]));
#endif
            });
#else
            {
                TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep)
                kaddr = lds_offset(slot);
                static_for<8>([&](auto R) __attribute__((always_inline)) { kb[R.value] = key_read_at<R.value * 1024>(kaddr, 0.0, 0.0); });
            }
            static_for<16>([&](auto POS) __attribute__((always_inline)) {
                constexpr int pos = POS.value, k2 = kperm(pos), kk = (pos + 1) % kPerChunk;
                key_wait(kb[0], kb[1], kb[2], kb[3], kb[4], kb[5], kb[6], kb[7]);
                if constexpr (pos + 1 < 16 && kk == 0) {  // next group opens the next chunk; this wave holds all it needs of the current one
                    TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep + (pos + 1) / kPerChunk)
                    kaddr = lds_offset(slot);
                }
                // (Measured and rejected: products level by level with the first level's registers refilled behind the
                // second level's products and separate waits per level: 7 % slower, the shorter chains leave less to interleave.)
                double pr[4], pi[4];
                static_for<4>([&](auto C) __attribute__((always_inline)) {
                    constexpr int c = C.value;
                    const v2d b0 = kb[c], b1 = kb[4 + c];
                    pr[c] = f1r[k2] * b0.x - f1i[k2] * b0.y + f2r[k2] * b1.x - f2i[k2] * b1.y;
                    pi[c] = f1r[k2] * b0.y + f1i[k2] * b0.x + f2r[k2] * b1.y + f2i[k2] * b1.x;
                    if constexpr (pos + 1 < 16) {
                        kb[c] = key_read_at<(kk * 8 + c) * 1024>(kaddr, pr[c], pi[c]);
                        kb[4 + c] = key_read_at<(kk * 8 + 4 + c) * 1024>(kaddr, pr[c], pi[c]);
                    }
                });
                our[k2] = reduce_scatter_rows(pr[0], pr[1], pr[2], pr[3]);
                oui[k2] = reduce_scatter_rows(pi[0], pi[1], pi[2], pi[3]);
            });
#endif
            TFA_STAMP(5);
        } else if constexpr (G::kDense) {
            // both levels transformed one after the other, key words read from the ring as they are used
            fft_forward(f1r, f1i, twl, tr, g, t);
            fft_forward(f2r, f2i, twl, tr, g, t);
#if TFA_PIN_SPECTRA
            pin16(f1r), pin16(f1i), pin16(f2r), pin16(f2i);
#endif
            TFA_STAMP(3);
#pragma unroll
            for (int q = 0; q < G::kChunksPerStep; q++) {
                TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep + q)
                TFA_STAMP(4);
#pragma unroll
                for (int kk = 0; kk < G::kChunkRows / 8; kk++) {
                    const int k2 = kperm((G::kChunkRows / 8) * q + kk);
                    double pr[4], pi[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
#ifdef TFA_EXP_NOKEYREAD  // timing experiment: key words from registers (results are garbage)
                        const c64 b0 = c64{f1r[(k2 + c + 1) & 15], f2i[(k2 + c + 2) & 15]}, b1 = c64{f2r[(k2 + c + 3) & 15], f1i[(k2 + c + 4) & 15]};
#else
                        const c64 b0 = slot[((kk * 2 + 0) * 4 + c) * 64], b1 = slot[((kk * 2 + 1) * 4 + c) * 64];
#endif
                        pr[c] = f1r[k2] * b0.x - f1i[k2] * b0.y + f2r[k2] * b1.x - f2i[k2] * b1.y;
                        pi[c] = f1r[k2] * b0.y + f1i[k2] * b0.x + f2r[k2] * b1.y + f2i[k2] * b1.x;
                    }
                    our[k2] = reduce_scatter_rows(pr[0], pr[1], pr[2], pr[3]);
                    oui[k2] = reduce_scatter_rows(pi[0], pi[1], pi[2], pi[3]);
                }
                TFA_STAMP(5);
            }
        } else {
            fft_forward2(f1r, f1i, f2r, f2i, twl, tr, tr2, g, t);
#if TFA_PIN_SPECTRA
            pin16(f1r), pin16(f1i), pin16(f2r), pin16(f2i);
#endif
            TFA_STAMP(3);
            // key words of a chunk are read into registers one barrier ahead of their use, so the LDS read
            // latency and the barrier hide behind the products of the previous chunk
            c64 kb[2][16];
            {
                TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep)
#pragma unroll
                for (int r = 0; r < 16; r++) kb[0][r] = slot[r * 64];
            }
#pragma unroll
            for (int q = 0; q < G::kChunksPerStep; q++) {
                if (q + 1 < G::kChunksPerStep) {
                    TFA_RING_ACQUIRE(slot, i * G::kChunksPerStep + q + 1)
#pragma unroll
                    for (int r = 0; r < 16; r++) kb[(q + 1) & 1][r] = slot[r * 64];
                }
                TFA_STAMP(4);
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int k2 = kperm(2 * q + kk);
                    double pr[4], pi[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const c64 b0 = kb[q & 1][(kk * 2 + 0) * 4 + c], b1 = kb[q & 1][(kk * 2 + 1) * 4 + c];
                        pr[c] = f1r[k2] * b0.x - f1i[k2] * b0.y + f2r[k2] * b1.x - f2i[k2] * b1.y;
                        pi[c] = f1r[k2] * b0.y + f1i[k2] * b0.x + f2r[k2] * b1.y + f2i[k2] * b1.x;
                    }
                    our[k2] = reduce_scatter_rows(pr[0], pr[1], pr[2], pr[3]);
                    oui[k2] = reduce_scatter_rows(pi[0], pi[1], pi[2], pi[3]);
#if TFA_PIN_GROUP4
                    asm volatile("" : "+v"(our[k2]), "+v"(oui[k2]));
#endif
                }
                TFA_STAMP(5);
            }
        }
        if constexpr (!G::kDense)
            fft_inverse2buf(our, oui, twl, tr, tr2, g, t);
        else if constexpr (TFA_INV_TW_PREFETCH && !G::kSplitLevels)
            fft_inverse_pf(our, oui, twl, tr, g, t);
        else
            fft_inverse(our, oui, twl, tr, g, t);
        TFA_STAMP(6);  // inverse transform
#pragma unroll
        for (int q = 0; q < 16; q++) {
            acc[q] += round_torus(our[q]);
            acc[q + 16] += round_torus(oui[q]);
        }
        TFA_STAMP(7);  // rounding + accumulate
    }

    if (!live) return;
    if (acc_out) {  // raw accumulator [job][row][N]
#pragma unroll
        for (int m = 0; m < 32; m++) acc_out[((size_t)job * kK1 + g) * kN + 16 * m + t] = acc[m];
        return;
    }
    // sample extraction of the constant coefficient: mask block r = (A_r[0], -A_r[N-1], ..., -A_r[1]), body B[0]
    uint32_t *out = big_out + (size_t)job * big_stride;
    if (g < 3) {
#pragma unroll
        for (int m = 0; m < 32; m++) {
            int j = 16 * m + t;
            out[g * kN + ((kN - j) & (kN - 1))] = j == 0 ? acc[m] : 0u - acc[m];
        }
    } else if (t == 0) {
        out[3 * kN] = acc[0];
    }
}

// Standard-domain GGSW rows -> Fourier layout.  One wave per GGSW row (i, level, r): its four
// polynomials (columns c) are transformed together, row c of the wave taking column c.
__global__ void __launch_bounds__(256, 1)
    bsk_convert_kernel(const uint32_t *__restrict__ bsk, c64 *__restrict__ bsk_f, const c64 *__restrict__ tw, uint32_t n_rows) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane_g(lane), t = lane_t(lane);
    const uint32_t row = blockIdx.x * kConvertWaves + wave;  // (i * l + lv) * 4 + r
    if (row >= n_rows) return;
    double *tr = reinterpret_cast<double *>(lds + wave * kWavePrivBytes);
    double re[16], im[16];
    const uint32_t *src = bsk + ((size_t)row * kK1 + g) * kN;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        re[q] = (double)(int32_t)src[16 * q + t];
        im[q] = (double)(int32_t)src[16 * q + t + 256];
    }
    fft_forward(re, im, tw, tr, g, t);
    const uint32_t r = row & 3, lv = (row >> 2) % kLevels, i = (row >> 2) / kLevels;
    // destination: the lane of the bootstrap kernel that owns input polynomial r and frequency lane t; in quad mode that
    // lane keeps column g in slot g ^ r (see reduce_scatter_rows)
    const int dst_lane = kQuad ? 4 * t + (int)r : (int)r * 16 + t, slot = kQuad ? (g ^ (int)r) : g;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) bsk_f[bsk_index(i, lv, kperm(k2), slot) + dst_lane] = c64{re[k2] * (1.0 / 256.0), im[k2] * (1.0 / 256.0)};
}

}  // namespace tfa
