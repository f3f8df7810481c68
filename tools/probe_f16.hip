// Probe: what bounds the 16-column sub-panel factorisation of the single-launch Cholesky (dag_f16 in
// csrc/ttk.hip)?  One wave factorises the first 16 columns of a 64 x 64 SPD tile in LDS, over and over;
// variants take pieces of the loop away or reorder it.  Prints shader clocks per call.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHB 64
__shared__ double T0[CHB][CHB + 1];
__shared__ double T[CHB][CHB + 1];
__shared__ double piv[16];
__shared__ double colb[2][CHB];
__shared__ double rsd[CHB];
__shared__ double colT[16][CHB];

__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_f64(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(y, fma(-d, y, 1.0), y);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double rcp_short(double d) {
  const double y0 = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, y0, 1.0);
  const double a = fma(y0, e, y0);
  return fma(a, e * e, a);
}
__device__ __forceinline__ void rsqrt_sqrt(double d, double& rs, double& sq) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  double s = d * y;
  s = fma(0.5 * y, fma(-s, s, d), s);
  rs = y;
  sq = s;
}

// V: 0 as in the library; 1 pivots kept in registers (no LDS write of the pivot per column); 2 = 1 + short
// reciprocal; 3 = 1 with every pivot-row entry through lane reads (no LDS broadcast); 4 = 1 + LDS entries applied one
// column late; 5 = 1 + pivot chain on uniform values; 6 only load / scale / store; 7 = 1 with a constant reciprocal
// (no chain: what the instruction stream alone costs); 8 = 1 without the final scaling part.
template <int V>
__device__ __forceinline__ void f16(int c0, int lane) {
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = T[lane][c0 + c];
  double pv[16];
  if (V == 6) {
#pragma unroll
    for (int c = 0; c < 16; ++c) pv[c] = readlane_f64(a[c], c0 + c);
  } else if (V == 9 || V == 10) {
    // Blocks of 4 columns.  Before a block is eliminated, the columns before it are applied to its four columns
    // in one go (their multipliers t_k stay in registers, the pivot-row entries come from the per-column LDS
    // copies: all reads of a block in flight together, one exposed LDS latency per block instead of per column);
    // inside the block the three, two, one pivot-row entries come through lane reads.
    double tk[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b > 0) {
        double e[4][12];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4 * b; ++k) e[j][k] = colT[k][c0 + 4 * b + j];
#pragma unroll
        for (int k = 0; k < 4 * b; ++k)      // every column receives the earlier columns in their order
#pragma unroll
          for (int j = 0; j < 4; ++j) a[4 * b + j] = fma(tk[k], e[j][k], a[4 * b + j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * b + j;
        const double d = readlane_f64(a[c], c0 + c);
        const double rc = V == 10 ? rcp_short(d) : rcp_f64(d);
        pv[c] = d;
        colT[c][lane] = a[c];
        const double t = -a[c] * rc;
        tk[c] = t;
#pragma unroll
        for (int jj = j + 1; jj < 4; ++jj) a[4 * b + jj] = fma(t, readlane_f64(a[c], c0 + 4 * b + jj), a[4 * b + jj]);
      }
      asm volatile("" ::: "memory");
    }
  } else if (V == 5) {
    double d = readlane_f64(a[0], c0);
    double rc = rcp_f64(d);
    double xn = readlane_f64(a[1], c0 + 1), yn = readlane_f64(a[0], c0 + 1);
    double y2 = yn * yn;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      pv[c] = d;
      colb[c & 1][lane] = a[c];
      asm volatile("" ::: "memory");
      double dn = d, rn = rc;
      if (c + 1 < 16) {
        dn = fma(-y2, rc, xn);
        rn = rcp_f64(dn);
      }
      const double t = -a[c] * rc;
      if (c + 1 < 16) a[c + 1] = fma(t, readlane_f64(a[c], c0 + c + 1), a[c + 1]);
      if (c + 2 < 16) {
        a[c + 2] = fma(t, readlane_f64(a[c], c0 + c + 2), a[c + 2]);
        yn = readlane_f64(a[c + 1], c0 + c + 2);
        xn = readlane_f64(a[c + 2], c0 + c + 2);
        y2 = yn * yn;
      }
#pragma unroll
      for (int cc = c + 3; cc < 16; ++cc) a[cc] = fma(t, colb[c & 1][c0 + cc], a[cc]);
      d = dn;
      rc = rn;
    }
  } else if (V == 4) {
    double d = readlane_f64(a[0], c0);
    double rc = rcp_f64(d);
    double tp = 0.0, rp[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      pv[c] = d;
      colb[c & 1][lane] = a[c];
      asm volatile("" ::: "memory");
      const double t = -a[c] * rc;
      if (c + 1 < 16) {
        a[c + 1] = fma(t, readlane_f64(a[c], c0 + c + 1), a[c + 1]);
        d = readlane_f64(a[c + 1], c0 + c + 1);
        rc = rcp_f64(d);
      }
      if (c >= 1) {
#pragma unroll
        for (int cc = c + 2; cc < 16; ++cc) a[cc] = fma(tp, rp[cc], a[cc]);
      }
      if (c + 2 < 16) a[c + 2] = fma(t, readlane_f64(a[c], c0 + c + 2), a[c + 2]);
#pragma unroll
      for (int cc = c + 3; cc < 16; ++cc) rp[cc] = colb[c & 1][c0 + cc];
      tp = t;
    }
  } else {
    double d = readlane_f64(a[0], c0);
    double rc = V == 2 ? rcp_short(d) : rcp_f64(d);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (V == 0) {
        if (lane == 0) piv[c] = d;
      } else {
        pv[c] = d;
      }
      if (V != 3) colb[c & 1][lane] = a[c];
      asm volatile("" ::: "memory");
      const double t = -a[c] * rc;
      if (c + 1 < 16) {
        a[c + 1] = fma(t, readlane_f64(a[c], c0 + c + 1), a[c + 1]);
        d = readlane_f64(a[c + 1], c0 + c + 1);
        if (V != 7) rc = V == 2 ? rcp_short(d) : rcp_f64(d);
      }
#pragma unroll
      for (int cc = c + 2; cc < 16; ++cc)
        a[cc] = fma(t, V == 3 ? readlane_f64(a[c], c0 + cc) : colb[c & 1][c0 + cc], a[cc]);
    }
  }
  if (V != 0 && V < 9) {
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) piv[c] = pv[c];
    }
  }
  if (V != 8) {
    double dv = 1.0;
    if (V >= 9) {     // pivot c to the lanes c mod 16 by selects: no trip through LDS
#pragma unroll
      for (int c = 0; c < 16; ++c) dv = (lane & 15) == c ? pv[c] : dv;
    } else {
      dv = piv[lane & 15];
    }
    double rsv, sqv;
    rsqrt_sqrt(dv, rsv, sqv);
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] *= readlane_f64(rsv, c);
    if (lane < 16) rsd[c0 + lane] = rsv;
  }
  if (lane >= c0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) T[lane][c0 + c] = a[c];
  }
}

template <int V>
__global__ __launch_bounds__(256, 2) void k(const double* A, double* out, long long* clk, int iters, int c0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < CHB * CHB; i += 256) T0[i / CHB][i % CHB] = A[i];
  __syncthreads();
  long long total = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < CHB * CHB; i += 256) T[i / CHB][i % CHB] = T0[i / CHB][i % CHB];
    __syncthreads();
    if (wave == 0) {
      const long long t0 = clock64();
      f16<V>(c0, lane);
      total += clock64() - t0;
    }
    __syncthreads();
  }
  if (tid == 0) clk[0] = total;
  for (int i = tid; i < CHB * CHB; i += 256) out[i] = T[i / CHB][i % CHB];
}

template <int V>
static void run(const double* dA, double* dout, long long* dclk, const std::vector<double>& ref, const char* what) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, dA, dout, dclk, 10, 0);
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, dA, dout, dclk, iters, 0);
  long long c = 0;
  std::vector<double> o(CHB * CHB);
  hipMemcpy(&c, dclk, sizeof(c), hipMemcpyDeviceToHost);
  hipMemcpy(o.data(), dout, sizeof(double) * CHB * CHB, hipMemcpyDeviceToHost);
  double err = 0.0;
  if (!ref.empty())
    for (int r = 0; r < CHB; ++r)
      for (int cc = 0; cc < 16; ++cc) {
        const double e = o[r * CHB + cc] - ref[r * CHB + cc];
        if (r >= cc && (e > err || -e > err)) err = e > 0 ? e : -e;
      }
  printf("V%d  %8.1f clocks per call  (max diff to V0 on the factor columns %.2e)  %s\n", V, (double)c / iters, err, what);
}

int main() {
  std::vector<double> A(CHB * CHB), M(CHB * 80);
  srand(1);
  for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < CHB; ++i)
    for (int j = 0; j < CHB; ++j) {
      double s = i == j ? 0.5 : 0.0;
      for (int q = 0; q < 80; ++q) s += M[i * 80 + q] * M[j * 80 + q] / 80.0;
      A[i * CHB + j] = s;
    }
  double *dA, *dout;
  long long* dclk;
  hipMalloc(&dA, sizeof(double) * CHB * CHB);
  hipMalloc(&dout, sizeof(double) * CHB * CHB);
  hipMalloc(&dclk, sizeof(long long));
  hipMemcpy(dA, A.data(), sizeof(double) * CHB * CHB, hipMemcpyHostToDevice);
  std::vector<double> none, ref(CHB * CHB);
  run<0>(dA, dout, dclk, none, "as in the library");
  hipMemcpy(ref.data(), dout, sizeof(double) * CHB * CHB, hipMemcpyDeviceToHost);
  run<1>(dA, dout, dclk, ref, "pivots in registers");
  run<2>(dA, dout, dclk, ref, "+ short reciprocal");
  run<3>(dA, dout, dclk, ref, "lane reads only, no LDS broadcast");
  run<4>(dA, dout, dclk, ref, "LDS entries applied one column late");
  run<5>(dA, dout, dclk, ref, "pivot chain on uniform values");
  run<6>(dA, dout, dclk, none, "load / scale / store only");
  run<7>(dA, dout, dclk, none, "constant reciprocal (no chain)");
  run<8>(dA, dout, dclk, none, "pivots in registers, no final scaling");
  run<9>(dA, dout, dclk, ref, "blocks of 4 columns, left-looking between blocks, pivots by selects");
  run<10>(dA, dout, dclk, ref, "the same + short reciprocal");
  return 0;
}
