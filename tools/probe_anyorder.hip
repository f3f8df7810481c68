// Probe: can a "chain" kernel (few workgroups, 130 KiB of LDS each, ~30 us) run beside a wide kernel
// (thousands of small workgroups, ~40 us) (a) through the any-order launch flag on ONE stream,
// (b) on two streams joined by events?  And what does a cross-stream dependency cost?
// Diagnostic only; build: hipcc --offload-arch=gfx950 -O3 -o probe_anyorder probe_anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void chain_kernel(double* out, int spin) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a = lds[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 123.0) out[blockIdx.x] = a;
}
__global__ __launch_bounds__(256, 4) void wide_kernel(double* out, int spin) {
  __shared__ double lds[2048];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a = lds[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 123.0) out[blockIdx.x] = a;
}
__global__ void tiny_kernel(double* out) { if (out[0] == 123.0) out[1] = 1.0; }

int main() {
  double* out;
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(out, 0, 1 << 20));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const size_t big = 130 * 1024;
  CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big));
  int spinA = 6000, spinB = 700;
  const int nA = 97, nB = 4096, reps = 200;
  void* argsA[] = {&out, &spinA};
  void* argsB[] = {&out, &spinB};
  auto run = [&](int mode) -> double {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) {          // chain only
        CK(hipExtLaunchKernel((const void*)chain_kernel, dim3(nA), dim3(256), argsA, big, s1, nullptr, nullptr, 0));
      } else if (mode == 1) {   // wide only
        CK(hipExtLaunchKernel((const void*)wide_kernel, dim3(nB), dim3(256), argsB, 0, s1, nullptr, nullptr, 0));
      } else if (mode == 2) {   // both, in order
        CK(hipExtLaunchKernel((const void*)chain_kernel, dim3(nA), dim3(256), argsA, big, s1, nullptr, nullptr, 0));
        CK(hipExtLaunchKernel((const void*)wide_kernel, dim3(nB), dim3(256), argsB, 0, s1, nullptr, nullptr, 0));
      } else if (mode == 3) {   // both, the wide one any-order
        CK(hipExtLaunchKernel((const void*)chain_kernel, dim3(nA), dim3(256), argsA, big, s1, nullptr, nullptr, 0));
        CK(hipExtLaunchKernel((const void*)wide_kernel, dim3(nB), dim3(256), argsB, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch));
      } else if (mode == 4) {   // two streams, fork + join by events
        CK(hipEventRecord(e1, s1));
        CK(hipStreamWaitEvent(s2, e1, 0));
        CK(hipExtLaunchKernel((const void*)chain_kernel, dim3(nA), dim3(256), argsA, big, s2, nullptr, nullptr, 0));
        CK(hipExtLaunchKernel((const void*)wide_kernel, dim3(nB), dim3(256), argsB, 0, s1, nullptr, nullptr, 0));
        CK(hipEventRecord(e2, s2));
        CK(hipStreamWaitEvent(s1, e2, 0));
      } else if (mode == 5) {   // tiny kernels in order on one stream
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s1, out);
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s1, out);
      } else if (mode == 6) {   // tiny kernels ping-pong between two streams through events
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s1, out);
        CK(hipEventRecord(e1, s1));
        CK(hipStreamWaitEvent(s2, e1, 0));
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s2, out);
        CK(hipEventRecord(e2, s2));
        CK(hipStreamWaitEvent(s1, e2, 0));
      }
    }
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
  };
  const char* names[] = {"chain only", "wide only", "chain + wide in order", "chain + wide any-order", "two streams fork/join",
                         "2 tiny in order", "2 tiny across streams"};
  for (int pass = 0; pass < 2; ++pass)
    for (int m = 0; m < 7; ++m) {
      double us = run(m);
      if (pass) printf("%-28s %8.2f us per repetition\n", names[m], us);
    }
  return 0;
}
