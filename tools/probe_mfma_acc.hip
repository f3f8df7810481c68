// Probe: fp64 MFMA throughput vs number of independent accumulators and waves per SIMD.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(double* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 36000 / NACC;
  k<NACC><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("nacc=%2d blocks=%4d  %.3f ms  %.2f TFLOP/s\n", NACC, blocks, ms, flops / ms * 1e-9);
}

int main() {
  double* out; hipMalloc(&out, 2048 * 256 * 8);
  for (int blocks : {256, 512, 1024}) {
    run<1>(out, blocks); run<2>(out, blocks); run<3>(out, blocks); run<4>(out, blocks); run<6>(out, blocks);
    run<8>(out, blocks); run<9>(out, blocks); run<12>(out, blocks); run<16>(out, blocks);
  }
  return 0;
}
