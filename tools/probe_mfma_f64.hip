// Probe: fp64 MFMA issue rate and C/D layout on gfx950. Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void rate_kernel(double* out, int iters) {
  d4 acc0 = {0,0,0,0}, acc1 = {0,0,0,0}, acc2 = {0,0,0,0}, acc3 = {0,0,0,0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  d4 s = acc0 + acc1 + acc2 + acc3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void layout_kernel(const double* A, const double* B, double* C) {
  // A is 16x4 row-major, B is 4x16 row-major, C is 16x16 row-major.
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[((l >> 4) + 4 * j) * 16 + (l & 15)] = acc[j];
}

__global__ void fma_rate_kernel(double* out, int iters) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0+4, x5=x0+5, x6=x0+6, x7=x0+7;
  double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, m, c); x1 = fma(x1, m, c); x2 = fma(x2, m, c); x3 = fma(x3, m, c);
    x4 = fma(x4, m, c); x5 = fma(x5, m, c); x6 = fma(x6, m, c); x7 = fma(x7, m, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0+x1+x2+x3+x4+x5+x6+x7;
}

int main() {
  double *out; hipMalloc(&out, 1024 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpb : {256, 512}) {
    int iters = 20000; int blocks = 256 * (wpb == 256 ? 2 : 1);
    rate_kernel<<<blocks, wpb>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<<<blocks, wpb>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (wpb / 64) * iters * 4 * 2048.0;
    printf("mfma_f64 16x16x4: blocks=%d threads=%d  %.3f ms  %.2f TFLOP/s\n", blocks, wpb, ms, flops / ms * 1e-9);
  }
  {
    int iters = 20000, blocks = 1024, wpb = 256;
    fma_rate_kernel<<<blocks, wpb>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_rate_kernel<<<blocks, wpb>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * wpb * iters * 8 * 2.0;
    printf("v_fma_f64: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
  }
  // layout
  std::vector<double> A(64), B(64), C(256), Cr(256, 0.0);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + i; B[i] = 100 + 3 * i + (i % 5); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Cr[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dC; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (C[i] != Cr[i]) ++bad;
  printf("layout check: %d mismatches of 256\n", bad);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s CUs=%d clock=%d kHz mem=%.1f GB\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9);
  return 0;
}
