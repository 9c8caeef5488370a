// Hardware probe: permlane swap semantics and fp64 VALU issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_swap(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + lane] = q[0]; out[192 + lane] = q[1];
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(double* out, int iters) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double c = 1.0000001, d = 1e-9;
  unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // fma f64
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 = fma(x0, c, d); x1 = fma(x1, c, d); x2 = fma(x2, c, d); x3 = fma(x3, c, d);
        x4 = fma(x4, c, d); x5 = fma(x5, c, d); x6 = fma(x6, c, d); x7 = fma(x7, c, d);
      }
    } else if (MODE == 1) {  // add f64
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 += d; x1 += d; x2 += d; x3 += d; x4 += d; x5 += d; x6 += d; x7 += d;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    } else if (MODE == 2) {  // permlane32 swap
#pragma unroll
      for (int j = 0; j < 32; j++) {
        auto r = __builtin_amdgcn_permlane32_swap(u0, u1, false, false); u0 = r[0]; u1 = r[1];
        auto q = __builtin_amdgcn_permlane16_swap(u2, u3, false, false); u2 = q[0]; u3 = q[1];
      }
    } else if (MODE == 3) {  // mul f64
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 *= c; x1 *= c; x2 *= c; x3 *= c; x4 *= c; x5 *= c; x6 *= c; x7 *= c;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    } else if (MODE == 5) {  // fma f64 interleaved with int adds
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 = fma(x0, c, d); u0 += u1; x1 = fma(x1, c, d); u1 ^= u2; x2 = fma(x2, c, d); u2 += u3; x3 = fma(x3, c, d); u3 ^= u0;
        x4 = fma(x4, c, d); u0 += u2; x5 = fma(x5, c, d); u1 ^= u3; x6 = fma(x6, c, d); u2 += u0; x7 = fma(x7, c, d); u3 ^= u1;
      }
    } else if (MODE == 6) {  // v_dot4c_i32_i8
#pragma unroll
      for (int j = 0; j < 16; j++) {
        u0 = __builtin_amdgcn_sdot4(u1, u2, u0, false); u1 = __builtin_amdgcn_sdot4(u2, u3, u1, false);
        u2 = __builtin_amdgcn_sdot4(u3, u0, u2, false); u3 = __builtin_amdgcn_sdot4(u0, u1, u3, false);
      }
    } else if (MODE == 7) {  // v_mad_u32_u24 / plain add
#pragma unroll
      for (int j = 0; j < 16; j++) { u0 += u1; u1 ^= u2; u2 += u3; u3 ^= u0; }
    } else if (MODE == 8) {  // fma f64 with three vector operands
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 = fma(x1, x2, x0); x1 = fma(x2, x3, x1); x2 = fma(x3, x4, x2); x3 = fma(x4, x5, x3);
        x4 = fma(x5, x6, x4); x5 = fma(x6, x7, x5); x6 = fma(x7, x0, x6); x7 = fma(x0, x1, x7);
      }
    } else if (MODE == 9) {  // add f64 with two vector operands
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 += x4; x1 += x5; x2 += x6; x3 += x7; x4 += x1; x5 += x2; x6 += x3; x7 += x0;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    } else if (MODE == 10) {  // fma f64: scalar constant times vector plus vector (the folded butterfly form)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 = fma(c, x4, x0); x1 = fma(c, x5, x1); x2 = fma(c, x6, x2); x3 = fma(c, x7, x3);
        x4 = fma(c, x1, x4); x5 = fma(c, x2, x5); x6 = fma(c, x3, x6); x7 = fma(c, x0, x7);
      }
    } else if (MODE == 11) {  // DPP quad permute moves
#pragma unroll
      for (int j = 0; j < 16; j++) {
        u0 = (unsigned)__builtin_amdgcn_update_dpp((int)u0, (int)u0, 0x4E, 0xF, 0xF, false); u1 = (unsigned)__builtin_amdgcn_update_dpp((int)u1, (int)u1, 0xB1, 0xF, 0xF, false);
        u2 = (unsigned)__builtin_amdgcn_update_dpp((int)u2, (int)u2, 0x4E, 0xF, 0xF, false); u3 = (unsigned)__builtin_amdgcn_update_dpp((int)u3, (int)u3, 0xB1, 0xF, 0xF, false);
      }
    } else if (MODE == 12) {  // v_cvt_f64_i32
#pragma unroll
      for (int j = 0; j < 8; j++) {
        asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        x0 = (double)(int)u0; x1 = (double)(int)u1; x2 = (double)(int)u2; x3 = (double)(int)u3;
        x4 = (double)(int)(u0 ^ 1); x5 = (double)(int)(u1 ^ 1); x6 = (double)(int)(u2 ^ 1); x7 = (double)(int)(u3 ^ 1);
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    } else if (MODE == 4) {  // v_mul_lo_u32
#pragma unroll
      for (int j = 0; j < 16; j++) { u0 = u0 * u1 + 1; u1 = u1 * u2 + 1; u2 = u2 * u3 + 1; u3 = u3 * u0 + 1; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(u0 ^ u1 ^ u2 ^ u3);
}

template <int MODE>
int rate(const char* name, double ops_per_iter_per_lane, int blocks, int threads) {
  double* d; CK(hipMalloc(&d, sizeof(double) * blocks * threads));
  int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double inst = ops_per_iter_per_lane * iters;                 // per wave instruction count
  double waves_per_simd = (double)blocks * threads / 64 / (256 * 4);
  double clk = ms * 1e-3 * 2.4e9 / (inst * waves_per_simd);
  printf("%-14s blocks=%d threads=%d  %.3f ms  -> %.2f clk per wave-instr per SIMD (at 2.4GHz), %.2f Tinst-lanes/s\n", name, blocks, threads, ms,
         clk, inst * blocks * threads / (ms * 1e-3) / 1e12);
  CK(hipFree(d));
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz lds/block=%zu regs/block=%d L2=%d\n", prop.name, prop.multiProcessorCount, prop.clockRate,
         prop.sharedMemPerBlock, prop.regsPerBlock, prop.l2CacheSize);
  unsigned* d; CK(hipMalloc(&d, 256 * 4));
  hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned> h(256); CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
  const char* names[4] = {"p32.vdst", "p32.src0", "p16.vdst", "p16.src0"};
  for (int r = 0; r < 4; r++) { printf("%s:", names[r]); for (int i = 0; i < 64; i++) printf(" %u", h[r * 64 + i]); printf("\n"); }
  for (int mult : {1, 2, 4}) {
    int th = 256, blocks = 256 * mult;   // mult waves per SIMD
    printf("--- %d wave(s) per SIMD\n", mult);
    rate<0>("fma_f64", 64, blocks, th); rate<1>("add_f64", 64, blocks, th); rate<3>("mul_f64", 64, blocks, th);
    rate<2>("permlane_swap", 64, blocks, th); rate<4>("mul_lo_u32", 64, blocks, th); rate<5>("mix_fma_int", 128, blocks, th);
    rate<6>("dot4_i8", 64, blocks, th); rate<7>("int_add_xor", 64, blocks, th);
    rate<8>("fma_f64_vvv", 64, blocks, th); rate<9>("add_f64_vv", 64, blocks, th); rate<10>("fma_f64_svv", 64, blocks, th); rate<11>("dpp_quad_mov", 64, blocks, th);
    rate<12>("cvt_f64_i32+xor/2", 64, blocks, th);
  }
  return 0;
}
