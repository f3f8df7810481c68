// ttk.hip -- hand-written gfx950 (CDNA4) kernels for the tensor-train DMRG core update
// and the C ABI declared in include/ttk.h.
//
// All matrix-shaped work runs on v_mfma_f64_16x16x4_f64 (wave64).  Operand lane map
// (verified on hardware with asymmetric integer data): lane l holds A[i = l&15][k = l>>4]
// and B[k = l>>4][j = l&15]; result register t of lane l is D[row = (l>>4) + 4t][col = l&15].
//
// The operands of every product here are row-wise Khatri-Rao products of small
// per-sample factor tables (left environment, feature vector, right environment), so
// they are generated in registers from LDS-staged tables instead of being read from HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ttk.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

static inline int ttk_launch_status() { return hipGetLastError() == hipSuccess ? TTK_OK : TTK_ELAUNCH; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------
// Environment step as a GEMM with an on-the-fly left operand.
//   rows  = (s,c) pairs (M = S*C), contraction index k = (e_idx, p_idx), columns n.
//   LEFT : k = l*p + pp,  Z[row,k] = E[row,l] * X[s,pp],  B[k][n] = core[k*n_out + n]
//   RIGHT: k = pp*e + r,  Z[row,k] = X[s,pp] * E[row,r],  B[k][n] = core[n*K + k]
// One wave owns 16 rows and up to 64 output columns; the core is staged through LDS in
// K-chunks; the wave's E and X rows sit transposed in LDS ([feature][row]) so that the 16
// lanes of a k-group read 16 consecutive doubles.
// ---------------------------------------------------------------------------------------
template <bool LEFT>
__global__ __launch_bounds__(256) void env_step_kernel(const double* __restrict__ E, const double* __restrict__ X,
                                                        const double* __restrict__ core, double* __restrict__ out,
                                                        int64_t M, int C, int e, int p, int n, int KC) {
  extern __shared__ double smem[];
  const int K = e * p;
  const int nb0 = blockIdx.y * 64;
  const int nloc = min(64, n - nb0);
  const int npad = (nloc + 15) & ~15;
  const int nt = npad >> 4;
  double* Bs = smem;
  double* Es = Bs + (size_t)KC * 64;
  double* Xs = Es + 4 * e * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  double* Ew = Es + wave * e * 16;
  double* Xw = Xs + wave * p * 16;
  for (int idx = lane; idx < 16 * e; idx += 64) {
    const int ii = idx & 15, l = idx >> 4;
    const int64_t row = row0 + ii;
    Ew[l * 16 + ii] = row < M ? E[row * e + l] : 0.0;
  }
  for (int idx = lane; idx < 16 * p; idx += 64) {
    const int ii = idx & 15, pp = idx >> 4;
    const int64_t row = row0 + ii;
    Xw[pp * 16 + ii] = row < M ? X[(row / C) * p + pp] : 0.0;
  }
  d4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};

  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < KC * npad; idx += 256) {
      int k, nn;
      double v = 0.0;
      if (LEFT) {
        k = idx / npad;
        nn = idx - k * npad;
        const int kk = kc0 + k;
        if (kk < K && nn < nloc) v = core[(int64_t)kk * n + nb0 + nn];
      } else {
        nn = idx / KC;
        k = idx - nn * KC;
        const int kk = kc0 + k;
        if (kk < K && nn < nloc) v = core[(int64_t)(nb0 + nn) * K + kk];
      }
      Bs[k * 64 + nn] = v;
    }
    __syncthreads();
    const int kend = min(KC, K - kc0);
    int kk = kc0 + kq;
    int ei, pi;
    if (LEFT) {
      ei = kk / p;
      pi = kk - ei * p;
    } else {
      pi = kk / e;
      ei = kk - pi * e;
    }
    for (int k0 = 0; k0 < kend; k0 += 4) {
      double a = 0.0;
      if (kk < K) a = Ew[ei * 16 + i] * Xw[pi * 16 + i];
      const double* brow = Bs + (k0 + kq) * 64 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < nt) acc[j] = MFMA_F64(a, brow[j * 16], acc[j]);
      }
      kk += 4;
      if (LEFT) {
        pi += 4;
        while (pi >= p) {
          pi -= p;
          ++ei;
        }
      } else {
        ei += 4;
        while (ei >= e) {
          ei -= e;
          ++pi;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < nt) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t row = row0 + kq + 4 * t;
        const int col = j * 16 + i;
        if (row < M && col < nloc) out[row * n + nb0 + col] = acc[j][t];
      }
    }
  }
}

static int env_step_launch(bool left, const double* E, const double* X, const double* core, double* out, int64_t M,
                           int C, int e, int p, int n, hipStream_t st) {
  const int K = e * p;
  int KC = round_up(K, 4);
  const int kc_max = (48 * 1024) / (64 * 8);  // 96 rows of 64 doubles = 48 KiB
  if (KC > kc_max) KC = kc_max;
  const size_t lds = ((size_t)KC * 64 + 4 * 16 * (size_t)(e + p)) * sizeof(double);
  if (lds > 150 * 1024) return TTK_EINVAL;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((n + 63) / 64));
  if (left)
    hipLaunchKernelGGL(env_step_kernel<true>, grid, dim3(256), lds, st, E, X, core, out, M, C, e, p, n, KC);
  else
    hipLaunchKernelGGL(env_step_kernel<false>, grid, dim3(256), lds, st, E, X, core, out, M, C, e, p, n, KC);
  return ttk_launch_status();
}

// ---------------------------------------------------------------------------------------
// y[s,c] = sum_r L[s,c,r] R[s,r]; 8 lanes per output row, shuffle reduction.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void predict_kernel(const double* __restrict__ L, const double* __restrict__ R,
                                                       double* __restrict__ y, int64_t M, int C, int r) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  double acc = 0.0;
  if (row < M) {
    const double* lp = L + row * r;
    const double* rp = R + (row / C) * r;
    for (int k = sub; k < r; k += 8) acc += lp[k] * rp[k];
  }
  acc += __shfl_xor(acc, 4, 8);
  acc += __shfl_xor(acc, 2, 8);
  acc += __shfl_xor(acc, 1, 8);
  if (row < M && sub == 0) y[row] = acc;
}

// ---------------------------------------------------------------------------------------
// Losses: one thread per sample.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_kernel(int kind, double w, const double* __restrict__ yp,
                                                    const double* __restrict__ y, double* __restrict__ loss,
                                                    double* __restrict__ g, double* __restrict__ H, int64_t S, int C) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const double* x = yp + s * C;
  if (kind == 0) {  // Bregman divergence of psi(x) = sum x^2
    const double* t = y + s * C;
    double px = 0.0, py = 0.0, pr = 0.0;
    for (int c = 0; c < C; ++c) {
      px += x[c] * x[c];
      py += t[c] * t[c];
      pr += (2.0 * t[c]) * (x[c] - t[c]);
      g[s * C + c] = 2.0 * x[c] - 2.0 * t[c];
    }
    loss[s] = px - py - pr;
    for (int c = 0; c < C * C; ++c) H[s * C * C + c] = 2.0;
  } else if (kind == 1) {  // element-wise squared error
    const double* t = y + s * C;
    for (int c = 0; c < C; ++c) {
      const double d = x[c] - t[c];
      loss[s * C + c] = d * d;
      g[s * C + c] = 2.0 * d;
      for (int d2 = 0; d2 < C; ++d2) H[(s * C + c) * C + d2] = (c == d2) ? 2.0 : 0.0;
    }
  } else {  // cross entropy on logits [w x, 0]; y one-hot over C+1 classes
    const double* t = y + s * (C + 1);
    double zmax = 0.0;
    int label = 0;
    double tbest = t[0];
    for (int c = 0; c < C; ++c) zmax = fmax(zmax, w * x[c]);
    for (int c = 1; c <= C; ++c)
      if (t[c] > tbest) {
        tbest = t[c];
        label = c;
      }
    double se = exp(0.0 - zmax);
    for (int c = 0; c < C; ++c) se += exp(w * x[c] - zmax);
    const double lse = zmax + log(se);
    const double zl = label < C ? w * x[label] : 0.0;
    loss[s] = lse - zl;
    for (int c = 0; c < C; ++c) {
      const double pc = exp(w * x[c] - lse);
      g[s * C + c] = w * (pc - t[c]);
      for (int d2 = 0; d2 < C; ++d2) {
        const double pd = exp(w * x[d2] - lse);
        H[(s * C + c) * C + d2] = w * w * ((c == d2 ? pc : 0.0) - pc * pd);
      }
    }
  }
}

// mean over mini-batches of per-batch means; one block, fixed summation order.
__global__ __launch_bounds__(1024) void batch_mean_kernel(const double* __restrict__ v, int64_t S, int nps,
                                                           int64_t batch, double* __restrict__ out) {
  __shared__ double wsum[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t nb = (S + batch - 1) / batch;
  double mine = 0.0;
  for (int64_t b = wave; b < nb; b += 16) {
    const int64_t s0 = b * batch, s1 = min(S, s0 + batch);
    const int64_t cnt = (s1 - s0) * nps;
    const double* base = v + s0 * nps;
    double acc = 0.0;
    for (int64_t k = lane; k < cnt; k += 64) acc += base[k];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    mine += acc / (double)cnt;
  }
  if (lane == 0) wsum[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    out[0] = t / (double)nb;
  }
}

// HL[s,d,l] = sum_c H[s,c,d] L[s,c,l]
__global__ __launch_bounds__(256) void hess_left_kernel(const double* __restrict__ H, const double* __restrict__ L,
                                                         double* __restrict__ HL, int64_t S, int C, int rl) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * C * rl) return;
  const int l = (int)(idx % rl);
  const int64_t sd = idx / rl;
  const int d = (int)(sd % C);
  const int64_t s = sd / C;
  double acc = 0.0;
  for (int c = 0; c < C; ++c) acc += H[(s * C + c) * C + d] * L[(s * C + c) * rl + l];
  HL[idx] = acc;
}

// ---------------------------------------------------------------------------------------
// Gram matrix in symmetry-compressed form.
//   G[a][b][c] = sum_s P[s,a] * XX[s,b] * RR[s,c]
//   P[s,a=(l,m)]  = sum_d HL[s,d,l] L[s,d,m]      (l<=m if the Hessian is symmetric)
//   XX[s,b=(p,q)] = X[s,p] X[s,q]  (p<=q),   RR[s,c=(r,t)] = R[s,r] R[s,t]  (r<=t)
// A is invariant under l<->m, p<->q, r<->t separately, so G holds every distinct entry
// once: for rl=rr=32, p=6 that is 528*21*528 numbers instead of 6144^2 (6.4x fewer FLOPs).
// GEMM view: rows a, columns (b,c), contraction over samples.  One wave owns a 48x48
// (a,c) tile of one b (3x3 MFMA tiles, 9 independent accumulators); the 4 waves of a
// workgroup take 4 consecutive tiles and share the LDS-staged factor rows of a sample
// chunk.  blockIdx.x % nslabs selects the sample slab, so workgroups that share an XCD
// (blockIdx % 8) stream the same slab through that XCD's L2.  Each (tile, slab) result
// goes to its own slab of `partial`; a second kernel sums slabs in fixed order.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_decode(int a, int n, int& i, int& j) {
  int row = 0, len = n;
  while (a >= len) {
    a -= len;
    --len;
    ++row;
  }
  i = row;
  j = row + a;
}
__host__ __device__ __forceinline__ int tri_index(int i, int j, int n) {  // requires i <= j
  return i * n - (i * (i - 1)) / 2 + (j - i);
}

struct GramDims {
  int MP, MPpad, NX, NR, NRpad, ntm, ntn, ntiles, ngroups, nslabs, CH;
  int64_t slab_len, gsize;
};

static GramDims gram_dims(int64_t S, int C, int rl, int p, int rr, int sym) {
  GramDims d;
  d.MP = sym ? rl * (rl + 1) / 2 : rl * rl;
  d.MPpad = round_up(d.MP, 48);
  d.NX = p * (p + 1) / 2;
  d.NR = rr * (rr + 1) / 2;
  d.NRpad = round_up(d.NR, 48);
  d.ntm = d.MPpad / 48;
  d.ntn = d.NRpad / 48;
  d.ntiles = d.NX * d.ntm * d.ntn;
  d.ngroups = (d.ntiles + 3) / 4;
  d.gsize = (int64_t)d.MPpad * d.NX * d.NRpad;
  int want = (1024 + d.ngroups - 1) / d.ngroups;
  int64_t cap = S / 512;
  if (cap < 1) cap = 1;
  if (want > cap) want = (int)cap;
  if (want > 64) want = 64;
  if (want >= 5 || S >= 8 * 2048) want = round_up(want < 8 ? 8 : want, 8);
  d.nslabs = want < 1 ? 1 : want;
  d.slab_len = round_up64((S + d.nslabs - 1) / d.nslabs, 4);
  // chunk of samples staged in LDS: (2*C*rl + p + rr) doubles per sample, <= 56 KiB
  const int width = 2 * C * rl + p + rr;
  int ch = (56 * 1024) / (width * 8);
  ch = ch / 4 * 4;
  if (ch > 64) ch = 64;
  d.CH = ch;
  return d;
}

template <bool C1>
__global__ __launch_bounds__(256, 2) void gram_kernel(const double* __restrict__ L, const double* __restrict__ HL,
                                                       const double* __restrict__ X, const double* __restrict__ R,
                                                       int64_t S, int C, int rl, int p, int rr, int sym, GramDims gd,
                                                       double* __restrict__ partial) {
  extern __shared__ double smem[];
  const int CW = C * rl;
  const int CH = gd.CH;
  double* Ls = smem;
  double* HLs = Ls + (size_t)CH * CW;
  double* Xs = HLs + (size_t)CH * CW;
  double* Rs = Xs + (size_t)CH * p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int slab = blockIdx.x % gd.nslabs;
  const int group = blockIdx.x / gd.nslabs;
  const int tile = group * 4 + wave;
  const bool active = tile < gd.ntiles;
  int b = 0, tm = 0, tn = 0;
  if (active) {
    b = tile / (gd.ntm * gd.ntn);
    const int rem = tile - b * (gd.ntm * gd.ntn);
    tm = rem / gd.ntn;
    tn = rem - tm * gd.ntn;
  }
  int pb, qb;
  tri_decode(b, p, pb, qb);
  int la[3], ma[3], rc[3], tc[3];
  double am[3], cm[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int a = tm * 48 + u * 16 + i;
    am[u] = a < gd.MP ? 1.0 : 0.0;
    const int aa = a < gd.MP ? a : 0;
    if (sym) {
      tri_decode(aa, rl, la[u], ma[u]);
    } else {
      la[u] = aa / rl;
      ma[u] = aa - la[u] * rl;
    }
    const int c = tn * 48 + u * 16 + i;
    cm[u] = c < gd.NR ? 1.0 : 0.0;
    tri_decode(c < gd.NR ? c : 0, rr, rc[u], tc[u]);
  }
  d4 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

  const int64_t s_begin = (int64_t)slab * gd.slab_len;
  int64_t s_end = s_begin + gd.slab_len;
  if (s_end > S) s_end = S;
  for (int64_t c0 = s_begin; c0 < s_end; c0 += CH) {
    const int64_t nvalid = min((int64_t)CH, s_end - c0);
    __syncthreads();
    {
      const int64_t nLe = nvalid * CW;
      const double* gL = L + c0 * CW;
      const double* gH = HL + c0 * CW;
      for (int idx = threadIdx.x; idx < CH * CW; idx += 256) {
        const bool ok = idx < nLe;
        Ls[idx] = ok ? gL[idx] : 0.0;
        HLs[idx] = ok ? gH[idx] : 0.0;
      }
      const double* gX = X + c0 * p;
      for (int idx = threadIdx.x; idx < CH * p; idx += 256) Xs[idx] = idx < nvalid * p ? gX[idx] : 0.0;
      const double* gR = R + c0 * rr;
      for (int idx = threadIdx.x; idx < CH * rr; idx += 256) Rs[idx] = idx < nvalid * rr ? gR[idx] : 0.0;
    }
    __syncthreads();
    if (active) {
      const int steps = (int)((nvalid + 3) >> 2);
      for (int ks = 0; ks < steps; ++ks) {
        const int s = ks * 4 + kq;
        const double* Lrow = Ls + s * CW;
        const double* Hrow = HLs + s * CW;
        const double* Xrow = Xs + s * p;
        const double* Rrow = Rs + s * rr;
        const double xx = Xrow[pb] * Xrow[qb];
        double aop[3], bop[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          double pa;
          if (C1) {
            pa = Hrow[la[u]] * Lrow[ma[u]];
          } else {
            pa = 0.0;
            for (int d = 0; d < C; ++d) pa += Hrow[d * rl + la[u]] * Lrow[d * rl + ma[u]];
          }
          aop[u] = pa * xx * am[u];
          bop[u] = Rrow[rc[u]] * Rrow[tc[u]] * cm[u];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v) acc[u][v] = MFMA_F64(aop[u], bop[v], acc[u][v]);
      }
    }
  }
  if (active) {
    double* dst = partial + (int64_t)slab * gd.gsize;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int a = tm * 48 + u * 16 + kq + 4 * t;
          const int c = tn * 48 + v * 16 + i;
          dst[((int64_t)a * gd.NX + b) * gd.NRpad + c] = acc[u][v][t];
        }
  }
}

// out[i] = sum_k partial[k*stride + i], fixed order
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const double* __restrict__ partial, int nslabs,
                                                            int64_t stride, int64_t n, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  double acc = 0.0;
  for (int k = 0; k < nslabs; ++k) acc += partial[(int64_t)k * stride + idx];
  out[idx] = acc;
}

__global__ __launch_bounds__(1024) void gram_scale_kernel(const double* __restrict__ G, int rl, int p, int rr, int sym,
                                                           int NX, int NRpad, double* __restrict__ scale) {
  __shared__ double wsum[16];
  const int D = rl * p * rr;
  double acc = 0.0;
  for (int idx = threadIdx.x; idx < D; idx += 1024) {
    const int r = idx % rr;
    const int lp = idx / rr;
    const int pp = lp % p;
    const int l = lp / p;
    const int a = sym ? tri_index(l, l, rl) : l * rl + l;
    const int b = tri_index(pp, pp, p);
    const int c = tri_index(r, r, rr);
    acc += fabs(G[((int64_t)a * NX + b) * NRpad + c]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    t /= (double)D;
    scale[0] = (t == 0.0) ? 1.0 : t;
  }
}

// A[row=(l,p,r)][col=(m,q,t)] = G[a(l,m)][b(p,q)][c(r,t)] / scale + ridge * (row == col)
__global__ __launch_bounds__(256) void gram_expand_kernel(const double* __restrict__ G, int rl, int p, int rr, int sym,
                                                           int NX, int NRpad, const double* __restrict__ scale,
                                                           double ridge, double* __restrict__ A) {
  const int D = rl * p * rr;
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= D) return;
  const int r = row % rr, lp = row / rr, pp = lp % p, l = lp / p;
  const int t = col % rr, mq = col / rr, q = mq % p, m = mq / p;
  const int a = sym ? tri_index(min(l, m), max(l, m), rl) : l * rl + m;
  const int b = tri_index(min(pp, q), max(pp, q), p);
  const int c = tri_index(min(r, t), max(r, t), rr);
  double v = G[((int64_t)a * NX + b) * NRpad + c];
  if (scale) v = v / scale[0];
  if (row == col) v += ridge;
  A[(int64_t)row * D + col] = v;
}

// ---------------------------------------------------------------------------------------
// Right-hand side / J^T u:  b[l,(p,r)] = sum_s GL[s,l] * X[s,p] R[s,r],  GL[s,l] = sum_c g[s,c] L[s,c,l].
// GEMM view: rows l (rl), columns (p,r), contraction over the samples of one slab.
// A workgroup owns a slab; each wave holds up to 6 output tiles at a time.
// ---------------------------------------------------------------------------------------
struct RhsDims {
  int mt, ntl, T, nslabs, CH, passes;
  int64_t slab_len;
};

static RhsDims rhs_dims(int64_t S, int C, int rl, int p, int rr) {
  RhsDims d;
  d.mt = (rl + 15) / 16;
  d.ntl = (p * rr + 15) / 16;
  d.T = d.mt * d.ntl;
  d.passes = (d.T + 23) / 24;
  int64_t ns = (S + 127) / 128;
  if (ns > 512) ns = 512;
  if (ns < 1) ns = 1;
  d.nslabs = (int)ns;
  d.slab_len = round_up64((S + d.nslabs - 1) / d.nslabs, 4);
  const int width = rl + p + rr;
  int ch = (48 * 1024) / (width * 8);
  ch = ch / 4 * 4;
  if (ch > 128) ch = 128;
  d.CH = ch;
  return d;
}

__global__ __launch_bounds__(256) void rhs_kernel(const double* __restrict__ L, const double* __restrict__ X,
                                                   const double* __restrict__ R, const double* __restrict__ g,
                                                   int64_t S, int C, int rl, int p, int rr, RhsDims rd,
                                                   double* __restrict__ partial) {
  extern __shared__ double smem[];
  const int CH = rd.CH;
  const int N = p * rr;
  double* GLs = smem;
  double* Xs = GLs + (size_t)CH * rl;
  double* Rs = Xs + (size_t)CH * p;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int slab = blockIdx.x;
  const int64_t s_begin = (int64_t)slab * rd.slab_len;
  int64_t s_end = s_begin + rd.slab_len;
  if (s_end > S) s_end = S;
  double* dst = partial + (int64_t)slab * rl * N;
  for (int pass = 0; pass < rd.passes; ++pass) {
    int lrow[6], xo[6], ro[6];
    double msk[6];
    bool tv[6];
    d4 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int t = pass * 24 + j * 4 + wave;
      tv[j] = t < rd.T;
      const int tmi = tv[j] ? t / rd.ntl : 0;
      const int tni = tv[j] ? t - tmi * rd.ntl : 0;
      const int l = tmi * 16 + i;
      const int n = tni * 16 + i;
      lrow[j] = l < rl ? l : 0;
      const int nn = n < N ? n : 0;
      xo[j] = nn / rr;
      ro[j] = nn - xo[j] * rr;
      msk[j] = (n < N ? 1.0 : 0.0);
      if (l >= rl) lrow[j] = -1;
      acc[j] = d4{0.0, 0.0, 0.0, 0.0};
    }
    for (int64_t c0 = s_begin; c0 < s_end; c0 += CH) {
      const int64_t nvalid = min((int64_t)CH, s_end - c0);
      __syncthreads();
      for (int idx = threadIdx.x; idx < CH * rl; idx += 256) {
        const int sl = idx / rl, l = idx - sl * rl;
        double v = 0.0;
        if (sl < nvalid) {
          const int64_t s = c0 + sl;
          for (int c = 0; c < C; ++c) v += g[s * C + c] * L[(s * C + c) * rl + l];
        }
        GLs[idx] = v;
      }
      const double* gX = X + c0 * p;
      for (int idx = threadIdx.x; idx < CH * p; idx += 256) Xs[idx] = idx < nvalid * p ? gX[idx] : 0.0;
      const double* gR = R + c0 * rr;
      for (int idx = threadIdx.x; idx < CH * rr; idx += 256) Rs[idx] = idx < nvalid * rr ? gR[idx] : 0.0;
      __syncthreads();
      const int steps = (int)((nvalid + 3) >> 2);
      for (int ks = 0; ks < steps; ++ks) {
        const int s = ks * 4 + kq;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (tv[j]) {
            const double a = lrow[j] >= 0 ? GLs[s * rl + lrow[j]] : 0.0;
            const double bb = Xs[s * p + xo[j]] * Rs[s * rr + ro[j]] * msk[j];
            acc[j] = MFMA_F64(a, bb, acc[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (tv[j]) {
        const int t = pass * 24 + j * 4 + wave;
        const int tmi = t / rd.ntl, tni = t - tmi * rd.ntl;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int l = tmi * 16 + kq + 4 * u;
          const int n = tni * 16 + i;
          if (l < rl && n < N) dst[(int64_t)l * N + n] = acc[j][u];
        }
      }
    }
  }
}

// u[s,c] = sum_r tmp[s,c,r] R[s,r];  hu[s,d] = sum_c H[s,c,d] u[s,c]
__global__ __launch_bounds__(256) void jvp_finish_kernel(const double* __restrict__ tmp, const double* __restrict__ R,
                                                          const double* __restrict__ H, double* __restrict__ hu,
                                                          int64_t S, int C, int rr) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const double* rp = R + s * rr;
  for (int d = 0; d < C; ++d) hu[s * C + d] = 0.0;
  for (int c = 0; c < C; ++c) {
    const double* tp = tmp + (s * C + c) * rr;
    double u = 0.0;
    for (int k = 0; k < rr; ++k) u += tp[k] * rp[k];
    for (int d = 0; d < C; ++d) hu[s * C + d] += H[(s * C + c) * C + d] * u;
  }
}

__global__ __launch_bounds__(256) void solve_rhs_kernel(const double* __restrict__ b, const double* __restrict__ w,
                                                         const double* __restrict__ scale, double ridge,
                                                         double* __restrict__ out, int64_t D) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= D) return;
  double v = scale ? b[idx] / scale[0] : b[idx];
  if (w) v += ridge * w[idx];
  out[idx] = -v;
}

__global__ __launch_bounds__(1024) void dot_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                    double* __restrict__ out, int64_t n) {
  __shared__ double wsum[16];
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += 1024) acc += x[k] * y[k];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    out[0] = t;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const double* __restrict__ a, double sign,
                                                    const double* __restrict__ x, double* __restrict__ y, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < n) y[idx] += sign * a[0] * x[idx];
}

// Event pairs around the Gram kernel (diagnostic; armed by ttk_gram_timing).
static const int kTimingPool = 512;
static hipEvent_t g_ev_start[kTimingPool], g_ev_stop[kTimingPool];
static bool g_ev_created = false, g_ev_armed = false;
static int g_ev_count = 0;

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

int ttk_gram_timing(int enable) {
  if (enable && !g_ev_created) {
    for (int k = 0; k < kTimingPool; ++k) {
      if (hipEventCreate(&g_ev_start[k]) != hipSuccess || hipEventCreate(&g_ev_stop[k]) != hipSuccess) return TTK_ELAUNCH;
    }
    g_ev_created = true;
  }
  g_ev_armed = enable != 0;
  g_ev_count = 0;
  return TTK_OK;
}

int ttk_gram_timing_read(double* ms_out, int max) {
  if (!ms_out || max < 0) return TTK_EINVAL;
  int n = 0;
  for (int k = 0; k < g_ev_count && n < max; ++k) {
    float ms = 0.f;
    if (hipEventSynchronize(g_ev_stop[k]) != hipSuccess) return TTK_ELAUNCH;
    if (hipEventElapsedTime(&ms, g_ev_start[k], g_ev_stop[k]) != hipSuccess) return TTK_ELAUNCH;
    ms_out[n++] = (double)ms;
  }
  g_ev_count = 0;
  return n;
}

int ttk_version(void) { return 100; }

int ttk_env_step_left_f64(const double* E, const double* X, const double* core, double* out, int64_t S, int C, int rl,
                          int p, int rr, void* stream) {
  if (!E || !X || !core || !out || S < 0 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  return env_step_launch(true, E, X, core, out, S * C, C, rl, p, rr, (hipStream_t)stream);
}

int ttk_env_step_right_f64(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                           int rr, void* stream) {
  if (!R || !X || !core || !out || S < 0 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  return env_step_launch(false, R, X, core, out, S, 1, rr, p, rl, (hipStream_t)stream);
}

int ttk_predict_f64(const double* L, const double* R, double* y, int64_t S, int C, int r, void* stream) {
  if (!L || !R || !y || S < 0 || C < 1 || r < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  const int64_t M = S * C;
  hipLaunchKernelGGL(predict_kernel, dim3((unsigned)((M * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, R, y,
                     M, C, r);
  return ttk_launch_status();
}

int ttk_loss_f64(int kind, double w, const double* y_pred, const double* y, double* loss, double* g, double* H,
                 int64_t S, int C, void* stream) {
  if (kind < 0 || kind > 2 || !y_pred || !y || !loss || !g || !H || S < 0 || C < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  hipLaunchKernelGGL(loss_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind, w, y_pred,
                     y, loss, g, H, S, C);
  return ttk_launch_status();
}

int ttk_batch_mean_f64(const double* v, int64_t S, int n_per_sample, int64_t batch, double* out, void* stream) {
  if (!v || !out || S < 1 || n_per_sample < 1 || batch < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, S, n_per_sample, batch, out);
  return ttk_launch_status();
}

int ttk_hess_left_f64(const double* H, const double* L, double* HL, int64_t S, int C, int rl, void* stream) {
  if (!H || !L || !HL || S < 0 || C < 1 || rl < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  const int64_t n = S * C * rl;
  hipLaunchKernelGGL(hess_left_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, L, HL,
                     S, C, rl);
  return ttk_launch_status();
}

int64_t ttk_gram_ws_bytes(int64_t S, int C, int rl, int p, int rr, int sym_left) {
  if (S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const GramDims d = gram_dims(S, C, rl, p, rr, sym_left);
  if (d.CH < 4) return TTK_EINVAL;
  return (int64_t)(1 + (d.nslabs > 1 ? d.nslabs : 0)) * d.gsize * (int64_t)sizeof(double);
}

int ttk_gram_f64(const double* L, const double* HL, const double* X, const double* R, int64_t S, int C, int rl, int p,
                 int rr, int sym_left, void* ws, int64_t ws_bytes, void* stream) {
  if (!L || !HL || !X || !R || !ws || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const GramDims d = gram_dims(S, C, rl, p, rr, sym_left);
  if (d.CH < 4) return TTK_EINVAL;
  if (ws_bytes < ttk_gram_ws_bytes(S, C, rl, p, rr, sym_left)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* G = (double*)ws;
  double* partial = d.nslabs > 1 ? G + d.gsize : G;
  const size_t lds = (size_t)d.CH * (2 * C * rl + p + rr) * sizeof(double);
  const unsigned grid = (unsigned)d.ngroups * (unsigned)d.nslabs;
  const bool timed = g_ev_armed && g_ev_count < kTimingPool;
  if (timed) hipEventRecord(g_ev_start[g_ev_count], st);
  if (C == 1)
    hipLaunchKernelGGL(gram_kernel<true>, dim3(grid), dim3(256), lds, st, L, HL, X, R, S, C, rl, p, rr, sym_left, d,
                       partial);
  else
    hipLaunchKernelGGL(gram_kernel<false>, dim3(grid), dim3(256), lds, st, L, HL, X, R, S, C, rl, p, rr, sym_left, d,
                       partial);
  if (timed) hipEventRecord(g_ev_stop[g_ev_count++], st);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  if (d.nslabs > 1) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((d.gsize + 255) / 256)), dim3(256), 0, st, partial,
                       d.nslabs, d.gsize, d.gsize, G);
    rc = ttk_launch_status();
  }
  return rc;
}

int ttk_gram_scale_f64(const void* ws, int rl, int p, int rr, int sym_left, double* scale, void* stream) {
  if (!ws || !scale || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const GramDims d = gram_dims(1, 1, rl, p, rr, sym_left);
  hipLaunchKernelGGL(gram_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double*)ws, rl, p, rr,
                     sym_left, d.NX, d.NRpad, scale);
  return ttk_launch_status();
}

int ttk_gram_expand_f64(const void* ws, int rl, int p, int rr, int sym_left, const double* scale, double ridge,
                        double* A_out, void* stream) {
  if (!ws || !A_out || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const GramDims d = gram_dims(1, 1, rl, p, rr, sym_left);
  const int D = rl * p * rr;
  hipLaunchKernelGGL(gram_expand_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)D), dim3(256), 0,
                     (hipStream_t)stream, (const double*)ws, rl, p, rr, sym_left, d.NX, d.NRpad, scale, ridge, A_out);
  return ttk_launch_status();
}

int64_t ttk_rhs_ws_bytes(int64_t S, int C, int rl, int p, int rr) {
  if (S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const RhsDims d = rhs_dims(S, C, rl, p, rr);
  if (d.CH < 4) return TTK_EINVAL;
  return (int64_t)d.nslabs * rl * p * rr * (int64_t)sizeof(double);
}

int ttk_rhs_f64(const double* L, const double* X, const double* R, const double* g, double* b, int64_t S, int C, int rl,
                int p, int rr, void* ws, int64_t ws_bytes, void* stream) {
  if (!L || !X || !R || !g || !b || !ws || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const RhsDims d = rhs_dims(S, C, rl, p, rr);
  if (d.CH < 4) return TTK_EINVAL;
  if (ws_bytes < ttk_rhs_ws_bytes(S, C, rl, p, rr)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t D = (int64_t)rl * p * rr;
  const size_t lds = (size_t)d.CH * (rl + p + rr) * sizeof(double);
  hipLaunchKernelGGL(rhs_kernel, dim3((unsigned)d.nslabs), dim3(256), lds, st, L, X, R, g, S, C, rl, p, rr, d,
                     (double*)ws);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, (const double*)ws,
                     d.nslabs, D, D, b);
  return ttk_launch_status();
}

int ttk_jvp_f64(const double* L, const double* X, const double* R, const double* H, const double* v, double* tmp,
                double* hu, int64_t S, int C, int rl, int p, int rr, void* stream) {
  if (!L || !X || !R || !H || !v || !tmp || !hu || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc = env_step_launch(true, L, X, v, tmp, S * C, C, rl, p, rr, st);
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(jvp_finish_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, tmp, R, H, hu, S, C, rr);
  return ttk_launch_status();
}

int ttk_solve_rhs_f64(const double* b, const double* w, const double* scale, double ridge, double* out, int64_t D,
                      void* stream) {
  if (!b || !out || D < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(solve_rhs_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, w, scale,
                     ridge, out, D);
  return ttk_launch_status();
}

int ttk_dot_f64(const double* x, const double* y, double* out, int64_t n, void* stream) {
  if (!x || !y || !out || n < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, out, n);
  return ttk_launch_status();
}

int ttk_axpy_f64(const double* a, double sign, const double* x, double* y, int64_t n, void* stream) {
  if (!a || !x || !y || n < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, sign, x, y,
                     n);
  return ttk_launch_status();
}

}  // extern "C"
