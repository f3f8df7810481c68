// Probe: latencies of the instructions the sub-panel factorisation is made of (gfx950, one wave): dependent
// v_fma_f64 / v_mul_f64 / v_rcp_f64 chains, v_readlane -> VALU use, LDS write -> broadcast read -> use.
#include <hip/hip_runtime.h>
#include <cstdio>
__shared__ double buf[2][64];
template <int K>
__global__ __launch_bounds__(256, 2) void k(double* out, long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  double x = 1.0 + lane * 1e-3, y = 0.999, z = 1e-9, acc = 0.0;
  double ind[8];
  for (int j = 0; j < 8; ++j) ind[j] = 1.0 + j;
  long long total = 0;
  for (int it = 0; it < iters; ++it) {
    const long long t0 = clock64();
    if (K == 0) {          // 64 dependent fma
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    } else if (K == 1) {   // 64 independent fma (8 chains of 8)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(ind[q]) : "v"(y), "v"(z));
    } else if (K == 2) {   // 16 dependent rcp (raw)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
    } else if (K == 3) {   // 16 x (readlane pair -> fma)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
        x = fma(x, __hiloint2double(hi, lo), z);
      }
    } else if (K == 4) {   // 16 x (LDS write -> broadcast read -> fma)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        buf[j & 1][lane] = x;
        asm volatile("" ::: "memory");
        x = fma(x, buf[j & 1][j], z);
      }
    } else if (K == 5) {   // 64 dependent mul
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    } else if (K == 6) {   // 64 dependent v_add_f64
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(z));
    } else if (K == 7) {   // 64 dependent fp32 fma
      float xf = (float)x, yf = 0.999f, zf = 1e-9f;
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf) : "v"(yf), "v"(zf));
      x = xf;
    }
    total += clock64() - t0;
    acc += x;
    for (int j = 0; j < 8; ++j) acc += ind[j];
    x = 1.0 + lane * 1e-3 + acc * 1e-300;
  }
  if (lane == 0) clk[0] = total;
  out[lane] = acc;
}
template <int K>
static void run(double* dout, long long* dclk, int n, const char* what) {
  const int iters = 1000;
  hipLaunchKernelGGL(k<K>, dim3(1), dim3(256), 0, 0, dout, dclk, 10);
  hipLaunchKernelGGL(k<K>, dim3(1), dim3(256), 0, 0, dout, dclk, iters);
  long long c = 0;
  hipMemcpy(&c, dclk, sizeof(c), hipMemcpyDeviceToHost);
  printf("K%d  %7.1f clocks per item  %s\n", K, (double)c / iters / n, what);
}
int main() {
  double* dout;
  long long* dclk;
  hipMalloc(&dout, 64 * 8);
  hipMalloc(&dclk, 8);
  run<0>(dout, dclk, 64, "dependent v_fma_f64");
  run<1>(dout, dclk, 64, "independent v_fma_f64 (8 chains)");
  run<2>(dout, dclk, 16, "dependent v_rcp_f64");
  run<3>(dout, dclk, 16, "2 v_readlane -> v_fma_f64, dependent");
  run<4>(dout, dclk, 16, "LDS write -> broadcast read -> v_fma_f64, dependent");
  run<5>(dout, dclk, 64, "dependent v_mul_f64");
  run<6>(dout, dclk, 64, "dependent v_add_f64");
  run<7>(dout, dclk, 64, "dependent v_fma_f32");
  return 0;
}
