// Probe: does the fp64 MFMA pipe run concurrently with fp64 vector arithmetic on gfx950?
// Waves with even index run an MFMA stream, odd ones a vector FMA stream (2 waves per SIMD), then each alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// mode bit0: waves on "slot A" run mfma; bit1: waves on "slot B" run fma.  512 threads = 8 waves = 2 per SIMD; wave w sits on SIMD w%4.
__global__ void __launch_bounds__(512) k(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool slotA = wave < 4;
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double c = 1.0000001, d = 1e-9;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  d4 big0 = {0, 0, 0, 0}, big1 = {0, 0, 0, 0};
  if (slotA && (mode & 1)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, x1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, x3, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x4, x5, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x6, x7, acc3, 0, 0, 0);
      }
    }
  } else if (slotA && (mode & 4)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        big0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, big0, 0, 0, 0);
        big1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x3, big1, 0, 0, 0);
      }
    }
  } else if (!slotA && (mode & 2)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x0 = fma(x0, c, d); x1 = fma(x1, c, d); x2 = fma(x2, c, d); x3 = fma(x3, c, d);
        x4 = fma(x4, c, d); x5 = fma(x5, c, d); x6 = fma(x6, c, d); x7 = fma(x7, c, d);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc0 + acc1 + acc2 + acc3 + big0[0] + big0[3] + big1[1];
}

int run(const char* name, int mode, double mfma_per_iter, double fma_per_iter) {
  double* d; const int blocks = 256, threads = 512, iters = 20000;
  CK(hipMalloc(&d, sizeof(double) * blocks * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 100, mode);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double clk = ms * 1e-3 * 2.4e9;
  printf("%-26s %.3f ms", name, ms);
  if (mfma_per_iter > 0) printf("  %.1f clk per mfma", clk / (mfma_per_iter * iters));
  if (fma_per_iter > 0) printf("  %.2f clk per vector fma", clk / (fma_per_iter * iters));
  printf("\n");
  CK(hipFree(d));
  return 0;
}
int main() {
  run("mfma 4x4x4 alone", 1, 64, 0);
  run("vector fma alone", 2, 0, 64);
  run("mfma 4x4x4 + vector fma", 3, 64, 64);
  run("mfma 16x16x4 alone", 4, 16, 0);
  run("mfma 16x16x4 + vector fma", 6, 16, 64);
  return 0;
}
