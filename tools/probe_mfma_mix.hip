// Probe: does fp64 VALU work or LDS traffic steal fp64 MFMA throughput on gfx950?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i * 1e-3;
  __syncthreads();
  d4 acc[9];
  for (int j = 0; j < 9; ++j) acc[j] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  double m0 = 1.0 + threadIdx.x * 1e-9, m1 = m0, m2 = m0, m3 = m0;
  float f0 = 1.0f + threadIdx.x * 1e-6f, f1 = f0;
  int li = threadIdx.x & 63;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      if (MODE == 1) { m0 *= 1.0000001; m1 *= 0.9999999; }
      if (MODE == 2) { f0 *= 1.0000001f; f1 *= 0.9999999f; }
      if (MODE == 3) { m0 += lds[(li + j * 67 + i) & 2047]; m1 += lds[(li * 3 + j + i) & 2047]; }
      if (MODE == 4) { m0 *= 1.0000001; m1 *= 0.9999999; m2 *= 1.0000002; m3 *= 0.9999998; }
    }
  }
  double s = 0;
  for (int j = 0; j < 9; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + m0 + m1 + m2 + m3 + f0 + f1;
}

template <int MODE>
void run(const char* name, double* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000, blocks = 512;
  k<MODE><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 9 * 2048.0;
  printf("%-34s %.3f ms  %.2f TFLOP/s (MFMA flops only)\n", name, ms, flops / ms * 1e-9);
}

int main() {
  double* out; hipMalloc(&out, 512 * 256 * 8);
  run<0>("mfma only", out);
  run<1>("mfma + 2 v_mul_f64 each", out);
  run<4>("mfma + 4 v_mul_f64 each", out);
  run<2>("mfma + 2 v_mul_f32 each", out);
  run<3>("mfma + 2 ds_read_b64+add_f64 each", out);
  return 0;
}
