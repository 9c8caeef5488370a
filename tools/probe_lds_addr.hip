// Does the cost of a ds_read_b128 depend on where in the 160 KiB LDS it reads?  256 FMAs + 32 reads per body, two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int NF, int NL>
__global__ void __launch_bounds__(512, 2) k(double* out, int iters, unsigned base, unsigned lane_stride) {
  extern __shared__ double buf[];
  for (int i = threadIdx.x; i < 20480; i += 512) buf[i] = 1.0 + i * 1e-9;
  __syncthreads();
  unsigned addr = base + (threadIdx.x & 63) * lane_stride;
  double x = threadIdx.x * 1e-9;
  asm volatile("v_mov_b32 v80, %1\n\t"
               "v_mov_b32 v0, %2\n\tv_mov_b32 v1, %3\n\tv_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %3\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\t"
               "v_mov_b32 v8, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\tv_mov_b32 v14, %2\n\tv_mov_b32 v15, %3\n\t"
               "v_mov_b32 v16, %2\n\tv_mov_b32 v17, %3\n\tv_mov_b32 v18, %2\n\tv_mov_b32 v19, %3\n\t"
               "L_loop%=:\n\t"
               ".set i, 0\n\t"
               ".rept 32\n\t"
               ".if i < %c4\n\t"
               "v_fma_f64 v[0:1], v[2:3], v[16:17], v[0:1]\n\tv_fma_f64 v[2:3], v[4:5], v[18:19], v[2:3]\n\tv_fma_f64 v[4:5], v[6:7], v[16:17], v[4:5]\n\tv_fma_f64 v[6:7], v[8:9], v[18:19], v[6:7]\n\t"
               "v_fma_f64 v[8:9], v[10:11], v[16:17], v[8:9]\n\tv_fma_f64 v[10:11], v[12:13], v[18:19], v[10:11]\n\tv_fma_f64 v[12:13], v[14:15], v[16:17], v[12:13]\n\tv_fma_f64 v[14:15], v[0:1], v[18:19], v[14:15]\n\t"
               ".endif\n\t"
               ".if i < %c5\n\t"
               "ds_read_b128 v[96:99], v80 offset:(i & 7) * 1024\n\t"
               ".endif\n\t"
               ".set i, i + 1\n\t"
               ".endr\n\t"
               "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 L_loop%=\n\t"
               "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %2, v0\n\tv_mov_b32 %3, v1\n\t"
               : "+s"(iters), "+v"(addr), "+v"(((unsigned*)&x)[0]), "+v"(((unsigned*)&x)[1])
               : "n"(NF / 8), "n"(NL)
               : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v80", "v96", "v97", "v98", "v99", "memory", "scc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + buf[threadIdx.x];
}
template <int NF, int NL>
int run(const char* name, unsigned base, unsigned lane_stride, double* d) {
  auto fn = k<NF, NL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), 163840, 0, d, 100, base, lane_stride);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), 163840, 0, d, iters, base, lane_stride);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-10s base=%6u lane_stride=%2u  %8.1f clk per body\n", name, base, lane_stride, ms * 1e-3 * 2.4e9 / iters);
  return 0;
}
int main() {
  double* d; CK(hipMalloc(&d, sizeof(double) * 256 * 512));
  for (int rep = 0; rep < 2; rep++) {
    run<256, 0>("fma only", 0, 16, d);
    for (unsigned base : {0u, 32768u, 65536u, 81920u, 98304u, 131072u, 147456u}) run<256, 32>("fma+rd128", base, 16, d);
    run<256, 32>("fma+bcast4", 0, 4, d);      // four lanes per 16-byte word (addresses lane * 4 rounded by the hardware? no: distinct, overlapping)
    for (unsigned base : {0u, 81920u, 131072u}) run<0, 32>("rd128 only", base, 16, d);
  }
  return 0;
}
