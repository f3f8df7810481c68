// Probe: can fp64 VALU / LDS work overlap fp64 MFMA on gfx950, within one wave and across two waves
// of a SIMD?  Exact instruction streams via inline asm.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define MUL(M) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(M) : "v"(c))
#define ADD32(I) asm volatile("v_add_u32 %0, %0, %1" : "+v"(I) : "v"(one))
#define DSR(D, ADDR) asm volatile("ds_read_b64 %0, %1" : "=v"(D) : "v"(ADDR))

// MODE 0: 9 MFMA.  1: 9 MFMA then 18 v_mul_f64.  2: interleaved MFMA, 2 mul.  3: 9 MFMA then 18 v_add_u32.
// 4: interleaved MFMA + 2 v_add_u32.  5: 9 MFMA then 14 ds_read + wait.  6: interleaved ds_read (wait at end).
// 7: only 18 v_mul_f64 (no MFMA).
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0;
  __syncthreads();
  d4 acc[9];
  for (int j = 0; j < 9; ++j) acc[j] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3, c = 1.0000001;
  double m[18];
  for (int j = 0; j < 18; ++j) m[j] = 1.0 + j;
  int iv[18];
  for (int j = 0; j < 18; ++j) iv[j] = j;
  int one = 1;
  double dv[14];
  unsigned addr = (threadIdx.x & 63) * 8;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 5) {
#pragma unroll
      for (int j = 0; j < 9; ++j) MFMA(acc[j]);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 18; ++j) MUL(m[j]);
      }
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 18; ++j) ADD32(iv[j]);
      }
      if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < 14; ++j) DSR(dv[j], addr + j * 520);
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
    } else if (MODE == 2 || MODE == 4 || MODE == 6) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        MFMA(acc[j]);
        if (MODE == 2) { MUL(m[2 * j]); MUL(m[2 * j + 1]); }
        if (MODE == 4) { ADD32(iv[2 * j]); ADD32(iv[2 * j + 1]); }
        if (MODE == 6) { if (j < 7) { DSR(dv[2 * j], addr + j * 1040); DSR(dv[2 * j + 1], addr + j * 1040 + 520); } }
      }
      if (MODE == 6) asm volatile("s_waitcnt lgkmcnt(0)");
    } else {
#pragma unroll
      for (int j = 0; j < 18; ++j) MUL(m[j]);
    }
  }
  double s = 0;
  for (int j = 0; j < 9; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int j = 0; j < 18; ++j) s += m[j] + iv[j];
  if (MODE == 5 || MODE == 6) for (int j = 0; j < 14; ++j) s += dv[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4000;
  k<MODE><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / iters / (blocks / 256);
  printf("%-40s waves/SIMD=%d  %.3f ms  %.0f cycles@2.4GHz per iteration per wave-slot\n", name, blocks / 256, ms, cyc);
}

int main() {
  double* out; hipMalloc(&out, 1024 * 256 * 8);
  for (int blocks : {256, 512}) {
    run<0>("9 mfma", out, blocks);
    run<7>("18 v_mul_f64 only", out, blocks);
    run<1>("9 mfma then 18 v_mul_f64", out, blocks);
    run<2>("9 x (mfma, 2 v_mul_f64)", out, blocks);
    run<3>("9 mfma then 18 v_add_u32", out, blocks);
    run<4>("9 x (mfma, 2 v_add_u32)", out, blocks);
    run<5>("9 mfma then 14 ds_read_b64 + wait", out, blocks);
    run<6>("9 x (mfma, 2 ds_read_b64) + wait", out, blocks);
  }
  return 0;
}
