// ttk_dense.hip -- gfx950 kernels for columns with many features (conv-featurised and operator
// columns: tens of patches per site) and for local problems given by explicit Jacobian rows.
//
// A site with P >= 16 features contracts the features FIRST: T[s,(l,r)] = sum_p X[s,p] core[l,p,r]
// is a plain GEMM shared by all outputs of a sample (rows = samples, K = P), and the two
// environments follow as small per-sample contractions: C p / (C + p) times fewer FLOPs than the
// three-factor Khatri-Rao rows.  Everything here is fp64 on v_mfma_f64_16x16x4_f64 (operand lane
// map as in ttk.hip: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; result
// register t of lane l is D[(l >> 4) + 4 t][l & 15]).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ttk.h"

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// HIP-event brackets around one kernel (ttk.hip; armed by ttk_gram_timing)
bool ttk_timing_begin(int which, hipStream_t st);
void ttk_timing_end(hipStream_t st);
typedef float f4 __attribute__((ext_vector_type(4)));

// Arithmetic of the matrix products of a kernel.  double: v_mfma_f64_16x16x4_f64.  float: v_mfma_f32_16x16x4_f32
// for float32 models (image_convolution_lanczos_CIFAR10.py never sets a dtype): operands are rounded to float32
// when they are staged, products and running sums are float32 at twice the fp64 matrix rate, memory stays
// float64.  Same operand lane map; result register t of lane l is row 4 (l >> 4) + t instead of (l >> 4) + 4 t.
// `banks` = LDS banks in elements: a row stride of 16 mod `banks` spreads the four k of an operand read.
template <typename T>
struct Mm;
template <>
struct Mm<double> {
  typedef d4 acc_t;
  static constexpr int banks = 32;
  static __device__ __forceinline__ int row(int kq, int t) { return kq + 4 * t; }
  static __device__ __forceinline__ acc_t fma(double a, double b, acc_t c) { return MFMA_F64(a, b, c); }
};
template <>
struct Mm<float> {
  typedef f4 acc_t;
  static constexpr int banks = 64;
  static __device__ __forceinline__ int row(int kq, int t) { return 4 * kq + t; }
  static __device__ __forceinline__ acc_t fma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

static inline int dense_status() { return hipGetLastError() == hipSuccess ? TTK_OK : TTK_ELAUNCH; }

// ---------------------------------------------------------------------------------------
// G[a,b][s,p] = sum_x X[s,p,x] F[a,x,b]: the feature vector of a conv column (tensor/layers.py:839-872:
// the edge between the input node X_k and the filter node C_k).  HBM-bound: X is read once, in
// contiguous blocks of rows staged through LDS (row stride px + (px even): conflict-free row walks).
// TI = float reads a float32 data set (image_convolution_lanczos_CIFAR10.py) without a float64 copy.
// ---------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void conv_feature_kernel(const TI* __restrict__ X, const double* __restrict__ F,
                                                            double* __restrict__ G, int64_t rows, int px, int nab, int RB) {
  extern __shared__ double smem[];
  const int pxs = px | 1;
  double* Fs = smem;                 // [px][nab]
  double* Xs = Fs + px * nab;        // [RB][pxs], RB <= 256 rows per workgroup
  const int tid = threadIdx.x;
  for (int k = tid; k < px * nab; k += 256) Fs[k] = F[k];   // F (a, x, b) -> Fs[x][a*cb + b] is done by the host layout
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int nrows = (int)min((int64_t)RB, rows - r0);
  const TI* src = X + r0 * px;
  const int total = nrows * px;
  for (int k0 = tid; k0 < RB * px; k0 += 256 * 8) {
    TI v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[min(k0 + 256 * u, total - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 256 * u;
      if (k < total) {
        const int r = k / px;
        Xs[r * pxs + (k - r * px)] = (double)v[u];
      }
    }
  }
  __syncthreads();
  if (tid < nrows) {
    const double* xr = Xs + tid * pxs;
    for (int ab = 0; ab < nab; ++ab) {
      double acc = 0.0;
      for (int x = 0; x < px; ++x) acc += xr[x] * Fs[x * nab + ab];
      G[(int64_t)ab * rows + r0 + tid] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row GEMM with an on-the-fly right operand and two epilogues.
//   T[m, col] = sum_k A[m,k] B(k, col),  A (M,K) row-major, rows = samples.
// BMODE 0: B (K,N) row-major.  BMODE 1: B(k,(l,r)) = core[l,k,r], core (rl,K,rr): the feature
// contraction of a site.  BMODE 2: B(k,col) = Bt[col,k], Bt (N,K) row-major.
// EPI 0 (TPL = 4, LB = 1): T is stored (M,N), 64 columns per pass.
// EPI 1 (BMODE 1 only): z[m,l] = sum_r T[m,(l,r)] Rv[m,r] is stored instead (the right-environment step
// of a site with many features, or the first half of J v).  A pass then covers LB consecutive l, each
// with its rr columns padded to TPL 16-wide tiles, so every MFMA tile lies inside one l and the row
// dot is a 16-lane shuffle sum; T never exists.
// A workgroup owns 64 rows (one 16-row tile per wave): the A tile ([k][row], stride 65; staged once when
// K fits one chunk) and the B block of a pass ([k][col], stride 16 mod 32 doubles: the four k of an
// operand read fall on different bank halves) go through LDS in nkc equal K chunks of at most 64.
// ---------------------------------------------------------------------------------------
#define GR_NB 24   // prefetch registers per thread for a B block: 64 k x 96 columns / 256 threads
// LDS row strides in elements.  Bs: 16 mod the bank count.  As ([k][row], written along k, read along row):
// odd, so that both walks spread over the banks.
template <typename TC>
__host__ __device__ constexpr int gemm_rows_bstride(int bw) {
  return bw + ((16 - bw) & (Mm<TC>::banks - 1));
}
template <typename TC>
__host__ __device__ constexpr int gemm_rows_astride() {
  return sizeof(TC) == 8 ? 65 : 81;
}
template <int BMODE, int EPI, int TPL, int LB, typename TC = double>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                           double* __restrict__ out, int64_t M, int K, int N, int rl,
                                                           int rr, const double* __restrict__ Rv, int accumulate, int kch) {
  extern __shared__ double smem[];
  typedef Mm<TC> mm;
  constexpr int NT = TPL * LB;
  constexpr int bw = NT * 16;          // columns per pass
  constexpr int bs = gemm_rows_bstride<TC>(bw);   // row stride of Bs, 16 mod the bank count
  constexpr int AS = gemm_rows_astride<TC>();     // row stride of As
  const int Kp = (K + 3) & ~3;
  TC* As = (TC*)smem;                  // [Kp][AS]: the whole A tile, staged once
  TC* Bs = As + Kp * AS;               // [kch][bs]: one K chunk of the B block of a pass
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int mrows = (int)min((int64_t)64, M - m0);
  const int nkc = (K + kch - 1) / kch;
  const int nblocks = EPI == 1 ? (rl + LB - 1) / LB : (N + 63) / 64;
  // The B block of step (blk, kc) travels in registers while the previous step is multiplied: loads with
  // clamped sources in one batch, zero selected when the registers go to LDS.
  TC pb[GR_NB];
  auto b_src = [&](int idx, int blk, int kc0, int kcn, bool& ok) -> int64_t {
    const int k = idx / bw, c = idx - k * bw;
    const int kk = kc0 + min(k, kcn - 1);
    if (EPI == 1) {
      const int lsub = c / (TPL * 16), r = c - lsub * (TPL * 16), l = blk * LB + lsub;
      ok = k < kcn && r < rr && l < rl;
      return ((int64_t)min(l, rl - 1) * K + kk) * rr + min(r, rr - 1);
    }
    const int col = blk * 64 + c;
    const int cc = min(col, N - 1);
    ok = k < kcn && col < N;
    if (BMODE == 0) return (int64_t)kk * N + cc;
    if (BMODE == 1) {
      const int l = cc / rr;
      return ((int64_t)l * K + kk) * rr + (cc - l * rr);
    }
    return (int64_t)cc * K + kk;
  };
  auto fetch_b = [&](int step) {
    const int blk = step / nkc, kc = step - blk * nkc;
    const int kc0 = kc * kch, kcn = min(kch, K - kc0);
    const int total = ((kcn + 3) & ~3) * bw;
#pragma unroll
    for (int u = 0; u < GR_NB; ++u) {
      bool ok;
      pb[u] = (TC)B[b_src(min(tid + 256 * u, total - 1), blk, kc0, kcn, ok)];
    }
  };
  auto store_b = [&](int step) {
    const int blk = step / nkc, kc = step - blk * nkc;
    const int kc0 = kc * kch, kcn = min(kch, K - kc0);
    const int total = ((kcn + 3) & ~3) * bw;
#pragma unroll
    for (int u = 0; u < GR_NB; ++u) {
      const int idx = tid + 256 * u;
      if (idx < total) {
        bool ok;
        (void)b_src(idx, blk, kc0, kcn, ok);
        const int k = idx / bw, c = idx - k * bw;
        Bs[k * bs + c] = ok ? pb[u] : (TC)0;
      }
    }
  };
  const int nsteps = nblocks * nkc;
  if (nsteps > 0) fetch_b(0);
  // A[m0 + r][k], coalesced along k, stored transposed; rows K..Kp are zero
  for (int k0 = tid; k0 < 64 * K; k0 += 256 * 4) {
    double v[4];
    int rr_[4], kk_[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(k0 + 256 * u, 64 * K - 1);
      rr_[u] = idx / K;
      kk_[u] = idx - rr_[u] * K;
      v[u] = A[(m0 + min(rr_[u], mrows - 1)) * K + kk_[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + 256 * u < 64 * K) As[kk_[u] * AS + rr_[u]] = rr_[u] < mrows ? (TC)v[u] : (TC)0;
  }
  for (int idx = tid; idx < (Kp - K) * 64; idx += 256) As[(K + idx / 64) * AS + (idx & 63)] = (TC)0;
  typename mm::acc_t acc[NT];
  for (int step = 0; step < nsteps; ++step) {
    const int blk = step / nkc, kc = step - blk * nkc;
    const int kc0 = kc * kch, kcn = min(kch, K - kc0);
    if (kc == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = typename mm::acc_t{0, 0, 0, 0};
    }
    __syncthreads();   // everybody is done with the previous Bs
    store_b(step);
    __syncthreads();   // (the first time this also publishes As)
    if (step + 1 < nsteps) fetch_b(step + 1);
    const int ksteps = (kcn + 3) >> 2;
    for (int ks = 0; ks < ksteps; ++ks) {
      const TC a = As[(kc0 + ks * 4 + kq) * AS + wave * 16 + i];
      const TC* brow = Bs + (ks * 4 + kq) * bs + i;
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mm::fma(a, brow[j * 16], acc[j]);
    }
    if (kc + 1 < nkc) continue;
    if (EPI == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = wave * 16 + mm::row(kq, t);
          const int col = blk * 64 + j * 16 + i;
          if (r < mrows && col < N) {
            double* o = out + (m0 + r) * N + col;
            *o = accumulate ? *o + (double)acc[j][t] : (double)acc[j][t];
          }
        }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = wave * 16 + mm::row(kq, t);
        const double* rrow = Rv + (m0 + min(r, mrows - 1)) * rr;
        double rvv[TPL];
#pragma unroll
        for (int j = 0; j < TPL; ++j) rvv[j] = (j * 16 + i < rr) ? rrow[min(j * 16 + i, rr - 1)] : 0.0;
#pragma unroll
        for (int lsub = 0; lsub < LB; ++lsub) {
          double z = 0.0;
#pragma unroll
          for (int j = 0; j < TPL; ++j) z += (double)acc[lsub * TPL + j][t] * rvv[j];
          z += __shfl_xor(z, 8, 16);
          z += __shfl_xor(z, 4, 16);
          z += __shfl_xor(z, 2, 16);
          z += __shfl_xor(z, 1, 16);
          const int l = blk * LB + lsub;
          if (i == 0 && r < mrows && l < rl) {
            double* o = out + (m0 + r) * rl + l;
            *o = accumulate ? *o + z : z;
          }
        }
      }
    }
  }
}

// K chunk of the B block (a multiple of 4, at most 64) and dynamic LDS of a gemm_rows launch with `bw`
// columns per pass (the whole A tile of 64 rows x K stays in LDS)
static int gemm_rows_chunk(int K) {
  const int nkc = (K + 63) / 64;
  return (((K + nkc - 1) / nkc) + 3) & ~3;
}
template <typename TC>
static size_t gemm_rows_lds(int K, int kch, int bw) {
  return ((size_t)((K + 3) & ~3) * gemm_rows_astride<TC>() + (size_t)kch * gemm_rows_bstride<TC>(bw)) * sizeof(TC);
}
// longest K: the 64 x K A tile + one 64 x 96 B chunk stay below 158 KiB
template <typename TC>
static constexpr int gemm_rows_max_k() {
  return sizeof(TC) == 8 ? 192 : 256;
}

// ---------------------------------------------------------------------------------------
// Right-environment step of a site with many features, core block resident:
//   z[s,l] (+)= sum_r (sum_k X[s,k] core[l,k,r]) Rv[s,r]
// The row-GEMM kernel above restages the B block of every pass in every 64-row workgroup, and that staging
// (element gathers with their address arithmetic) plus the row dot, not the matrix pipe, is what bounds it.
// Here a workgroup owns LB consecutive l (NT = TPL LB column tiles) for a range of samples and its core block
// goes to LDS once.  Its eight waves then work on their own: a wave takes every eighth 16-row tile of the
// range, copies it (one contiguous block of memory) into a private LDS image [k][row] -- the next tile
// travelling in registers meanwhile -- multiplies, and does the row dot with Rv; no workgroup barrier after
// the first, so the dot / staging instructions of one wave run beside the MFMAs of the others.
// Workgroup -> (sample range, column block): dispatch is round-robin over the 8 XCDs, so blockIdx & 7 picks
// an eighth of the samples and all column blocks of that eighth run on one XCD, whose L2 then holds its part
// of X (3.3 MB at 50 000 x 65) for the ncb passes over it.
// ---------------------------------------------------------------------------------------
#define ER_AS 17   // row stride of a wave's X tile image [k][16 rows]

// sum over the 16 lanes of a DPP row, left in every lane (rotations by 8, 4, 2, 1 inside the row)
__device__ __forceinline__ double row16_sum(double z) {
#define ROW_ROR_ADD(n)                                                                              \
  {                                                                                                 \
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(z), 0x120 + (n), 0xf, 0xf, false);        \
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(z), 0x120 + (n), 0xf, 0xf, false);        \
    z += __hiloint2double(hi, lo);                                                                  \
  }
  ROW_ROR_ADD(8)
  ROW_ROR_ADD(4)
  ROW_ROR_ADD(2)
  ROW_ROR_ADD(1)
#undef ROW_ROR_ADD
  return z;
}
__device__ __forceinline__ float row16_sum(float z) {
#define ROW_ROR_ADD(n) z += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(z), 0x120 + (n), 0xf, 0xf, false));
  ROW_ROR_ADD(8)
  ROW_ROR_ADD(4)
  ROW_ROR_ADD(2)
  ROW_ROR_ADD(1)
#undef ROW_ROR_ADD
  return z;
}

template <int TPL, int LB, int ER_NA, typename TC>   // ER_NA: X tile elements per lane, 16 rows x K <= 64 ER_NA
__global__ __launch_bounds__(512, sizeof(TC) == 4 && ER_NA <= 17 && TPL <= 4 ? 4 : 2) void env_right_resident_kernel(
    const double* __restrict__ X, const double* __restrict__ core, double* __restrict__ out, int64_t S, int K, int rl, int rr,
    const double* __restrict__ Rv, int accumulate, int ncb, int64_t rows_xcd, int64_t rows_wg) {
  extern __shared__ double smem[];
  typedef Mm<TC> mm;
  constexpr int NT = TPL * LB;
  constexpr int bw = NT * 16;
  constexpr int bs = gemm_rows_bstride<TC>(bw);
  const int Kp = (K + 3) & ~3;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  TC* Bs = (TC*)smem;                          // [Kp][bs]: the core block, staged once
  TC* Aw = Bs + Kp * bs + wave * Kp * ER_AS;   // [Kp][ER_AS]: this wave's X tile, [k][row]; column 16 is never read
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int cb = q % ncb, split = q / ncb;
  const int64_t x_end = min(S, (xcd + 1) * rows_xcd);
  const int64_t m_begin = xcd * rows_xcd + split * rows_wg;
  const int64_t m_end = min(x_end, m_begin + rows_wg);
  if (m_begin >= m_end) return;       // the whole workgroup: nothing to wait for yet
  const int l0 = cb * LB;
  // element lane + 64 u of a tile is row e / K, column e % K; elements past the tile go to the unused column
  int adst[ER_NA];
  {
    int r = lane / K, c = lane - r * K;
    const int dr = 64 / K, dc = 64 - dr * K;
#pragma unroll
    for (int u = 0; u < ER_NA; ++u) {
      adst[u] = (lane + 64 * u < 16 * K) ? c * ER_AS + r : 16;
      c += dc;
      r += dr;
      if (c >= K) {
        c -= K;
        ++r;
      }
    }
  }
  // a tile whose 64 ER_NA element reads stay inside X (all but the last few rows) is read without clamping
  const int64_t plain_end = S - (64 * ER_NA + K - 1) / K;
  TC pa[ER_NA];
  auto fetch = [&](int64_t m0) {
    const double* x0 = X + m0 * K + lane;
    if (m0 <= plain_end && m0 + 16 <= m_end) {
#pragma unroll
      for (int u = 0; u < ER_NA; ++u) pa[u] = (TC)x0[64 * u];
    } else {
      const int nvalid = (int)min((int64_t)16, m_end - m0);
      const int lx = nvalid * K - 1 - lane;     // last valid offset relative to this lane's first element
#pragma unroll
      for (int u = 0; u < ER_NA; ++u) pa[u] = (TC)x0[max(min(64 * u, lx), -lane)];
    }
  };
  auto stage = [&](int64_t m0) {
    if (m0 + 16 <= m_end) {
#pragma unroll
      for (int u = 0; u < ER_NA; ++u) Aw[adst[u]] = pa[u];
    } else {
      const int lx = (int)(m_end - m0) * K - 1 - lane;
#pragma unroll
      for (int u = 0; u < ER_NA; ++u) Aw[adst[u]] = 64 * u <= lx ? pa[u] : (TC)0;
    }
  };
  int64_t m0 = m_begin + wave * 16;
  if (m0 < m_end) fetch(m0);
  // the core block: loads in batches of eight, sources clamped, zero selected at the store
  for (int base = 0; base < Kp * bw; base += 512 * 8) {
    double v[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = min(base + tid + 512 * u, Kp * bw - 1);
      const int k = idx / bw, c = idx - k * bw;
      const int lsub = c / (TPL * 16), r = c - lsub * (TPL * 16), l = l0 + lsub;
      ok[u] = k < K && r < rr && l < rl;
      v[u] = core[((int64_t)min(l, rl - 1) * K + min(k, K - 1)) * rr + min(r, rr - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + tid + 512 * u;
      if (idx < Kp * bw) {
        const int k = idx / bw, c = idx - k * bw;
        Bs[k * bs + c] = ok[u] ? (TC)v[u] : (TC)0;
      }
    }
  }
  for (int e = lane; e < (Kp - K) * 16; e += 64) Aw[(K + e / 16) * ER_AS + (e & 15)] = (TC)0;
  // columns of Rv a lane multiplies with (past rr: any column, its products are zero)
  int cj[TPL];
#pragma unroll
  for (int j = 0; j < TPL; ++j) cj[j] = min(j * 16 + i, rr - 1);
  __syncthreads();                    // the core block is complete; from here on the waves run on their own
  for (; m0 < m_end; m0 += 128) {
    const int mrows = (int)min((int64_t)16, m_end - m0);
    asm volatile("" ::: "memory");    // the products of the previous tile have consumed their operand reads
    stage(m0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS writes land in order; nobody else reads this image
    if (m0 + 128 < m_end) fetch(m0 + 128);
    // the right-environment rows of this tile, wanted after the products
    TC rvv[4][TPL];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double* rrow = Rv + (m0 + min(mm::row(kq, t), mrows - 1)) * rr;
#pragma unroll
      for (int j = 0; j < TPL; ++j) rvv[t][j] = (TC)rrow[cj[j]];
    }
    typename mm::acc_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = typename mm::acc_t{0, 0, 0, 0};
    const TC* a = Aw + kq * ER_AS + i;
    const TC* b = Bs + kq * bs + i;
    // two k steps per turn: the operands of one are read while the other is multiplied
    TC a0 = a[0], b0[NT], a1, b1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b0[j] = b[j * 16];
    for (int ks = 0; ks < Kp; ks += 8) {
      const int k1 = min(ks + 4, Kp - 4), k2 = min(ks + 8, Kp - 4);
      a1 = a[k1 * ER_AS];
#pragma unroll
      for (int j = 0; j < NT; ++j) b1[j] = b[k1 * bs + j * 16];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = mm::fma(a0, b0[j], acc[j]);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a[k2 * ER_AS];
#pragma unroll
      for (int j = 0; j < NT; ++j) b0[j] = b[k2 * bs + j * 16];
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 4 < Kp) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mm::fma(a1, b1[j], acc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // row dot with Rv in the arithmetic of the products; lane lsub of a row keeps the sum of l0 + lsub
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      TC zsel = (TC)0;
#pragma unroll
      for (int lsub = 0; lsub < LB; ++lsub) {
        TC z = (TC)0;
#pragma unroll
        for (int j = 0; j < TPL; ++j) z += acc[lsub * TPL + j][t] * rvv[t][j];
        z = row16_sum(z);
        zsel = (i == lsub) ? z : zsel;
      }
      const int r = mm::row(kq, t), l = l0 + i;
      if (i < LB && r < mrows && l < rl) {
        double* o = out + (m0 + r) * rl + l;
        *o = accumulate ? *o + (double)zsel : (double)zsel;
      }
    }
  }
}

struct EnvResDims {
  int ncb, nsplit;
  int64_t rows_xcd, rows_wg;
  size_t lds;
};
// extents of a resident launch for NT column tiles per workgroup; lds = 0 when the site does not fit
template <typename TC>
static EnvResDims env_resident_dims(int64_t S, int rl, int K, int lb, int nt) {
  EnvResDims d;
  const int Kp = (K + 3) & ~3;
  d.lds = ((size_t)Kp * gemm_rows_bstride<TC>(nt * 16) + (size_t)8 * Kp * ER_AS) * sizeof(TC);
  if (K > 96 || K < 4 || d.lds > 156 * 1024) d.lds = 0;
  d.ncb = (rl + lb - 1) / lb;
  d.rows_xcd = ((S + 7) / 8 + 63) / 64 * 64;
  // Sample ranges per XCD: the launch runs in rounds of `slots` workgroups (one per compute unit, two when the
  // LDS image allows), a workgroup costs its tiles plus about two tiles' worth of staging the core block.
  const int64_t slots = 256 * (d.lds && d.lds <= 76 * 1024 ? 2 : 1);
  const int64_t tiles = d.rows_xcd / 64;
  int64_t best = -1, best_cost = 0;
  for (int64_t ns = 1; ns <= tiles; ++ns) {
    const int64_t per = (tiles + ns - 1) / ns, nsplit = (tiles + per - 1) / per;
    const int64_t rounds = (8 * d.ncb * nsplit + slots - 1) / slots;
    const int64_t cost = rounds * (per + 2);
    if (best < 0 || cost < best_cost) {
      best = per;
      best_cost = cost;
    }
  }
  d.rows_wg = best * 64;
  d.nsplit = (int)((tiles + best - 1) / best);
  return d;
}

// out[s,c,j] (+)= sum_l L[s,c,l] T[s,l,j]: the output index of the left environment applied to the
// feature-contracted core (second stage of the left step of a site with many features; also
// M[s,c,u] = L . (A . R) of a feature node's Jacobian).  HBM-bound on T.  A workgroup takes whole
// samples (as many as fit 256 threads at one thread per (s, j), or one sample in several passes over j):
// their L rows go through LDS, T is read coalesced along j.  The output rows may sit inside a wider tensor
// (sample stride ostride_s, output-index stride ostride_c, j contiguous): the rows J[s,c,a,(x,b)] of a conv
// filter node between convolution bonds are written in place, one call per left channel a.
__global__ __launch_bounds__(256) void lcontract_kernel(const double* __restrict__ L, const double* __restrict__ T,
                                                         double* __restrict__ out, int64_t S, int C, int rl, int n,
                                                         int spb, int accumulate, int64_t ostride_s, int64_t ostride_c) {
  extern __shared__ double Ls[];   // [spb][C * rl]
  const int64_t s0 = (int64_t)blockIdx.x * spb;
  const int ns = (int)min((int64_t)spb, S - s0);
  const int CL = C * rl;
  for (int k = threadIdx.x; k < ns * CL; k += 256) Ls[k] = L[s0 * CL + k];
  __syncthreads();
  for (int idx = threadIdx.x; idx < ns * n; idx += 256) {
    const int sl = idx / n, j = idx - sl * n;
    const double* Lrow = Ls + sl * CL;
    const double* Ts = T + (s0 + sl) * (int64_t)rl * n + j;
    for (int c0 = 0; c0 < C; c0 += 8) {
      double acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = 0.0;
      for (int l = 0; l < rl; ++l) {
        const double t = Ts[(int64_t)l * n];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += Lrow[min(c0 + c, C - 1) * rl + l] * t;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c0 + c < C) {
          double* o = out + (s0 + sl) * ostride_s + (c0 + c) * ostride_c + j;
          *o = accumulate ? *o + acc[c] : acc[c];
        }
    }
  }
}

// out[b][s,c,j] = sum_a sum_l L_a[s,c,l] T_ab[s,l,j]: the left-environment step through a column between convolution
// bonds once its features are contracted (T_ab = G_ab . core for all terms as ONE row GEMM over the stacked features):
// the sum over the left channels a stays in registers, one launch for all b (blockIdx.y) instead of one per term.
// L (ca,S,C,rl), T (ca,nb,S,rl,n), out (nb,S,C,n).  Same work split as lcontract_kernel.
__global__ __launch_bounds__(256) void lcontract_sum_kernel(const double* __restrict__ L, const double* __restrict__ T,
                                                             double* __restrict__ out, int64_t S, int C, int rl, int n,
                                                             int spb, int ca, int nb) {
  extern __shared__ double Ls[];   // [ca][spb][C * rl]
  const int64_t s0 = (int64_t)blockIdx.x * spb;
  const int b = blockIdx.y;
  const int ns = (int)min((int64_t)spb, S - s0);
  const int CL = C * rl;
  for (int a = 0; a < ca; ++a)
    for (int k = threadIdx.x; k < ns * CL; k += 256) Ls[a * spb * CL + k] = L[((int64_t)a * S + s0) * CL + k];
  __syncthreads();
  for (int idx = threadIdx.x; idx < ns * n; idx += 256) {
    const int sl = idx / n, j = idx - sl * n;
    for (int c0 = 0; c0 < C; c0 += 8) {
      double acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = 0.0;
      for (int a = 0; a < ca; ++a) {
        const double* Lrow = Ls + (a * spb + sl) * CL;
        const double* Ts = T + ((((int64_t)a * nb + b) * S + s0 + sl) * rl) * n + j;
        for (int l = 0; l < rl; ++l) {
          const double t = Ts[(int64_t)l * n];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += Lrow[min(c0 + c, C - 1) * rl + l] * t;
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c0 + c < C) out[(((int64_t)b * S + s0 + sl) * C + c0 + c) * n + j] = acc[c];
    }
  }
}

// J[s,c,x,b] (+)= sum_p Mv_b[s,c,p] X[s,p,x] for b < nb, written with sample stride jstride_s, row stride jstride_c,
// column stride xstride and the nb results of a (c, x) side by side (the Jacobian rows of a conv filter node
// C_k[a,x,b] sit at J[s,c,a,x,b]; with nb channels in one call the image rows are read once, not nb times).
// Mv_b = Mv + b mstride_b.  One workgroup per sample: X[s] and the Mv_b[s] through LDS, one thread per (c, x, b).
// HBM-bound on X and Mv.
template <typename TI>
__global__ __launch_bounds__(256) void patch_contract_kernel(const double* __restrict__ Mv, const TI* __restrict__ X,
                                                              double* __restrict__ J, int C, int P, int px,
                                                              int64_t jstride_s, int64_t jstride_c, int64_t xstride,
                                                              int accumulate, int nb, int64_t mstride_b) {
  extern __shared__ double smem[];
  double* Xs = smem;            // [P][px]
  double* Ms = Xs + P * px;     // [nb][C][P]
  const int64_t s = blockIdx.x;
  const TI* xs = X + s * (int64_t)P * px;
  // eight unconditional (clamped) loads per thread in flight; the image row first, it is the long one
  const int nx = P * px, nm = C * P;
  for (int k0 = threadIdx.x; k0 < nx; k0 += 256 * 8) {
    TI t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = xs[min(k0 + 256 * u, nx - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 256 * u < nx) Xs[k0 + 256 * u] = (double)t[u];
  }
  for (int b = 0; b < nb; ++b) {
    const double* ms = Mv + b * mstride_b + s * (int64_t)nm;
    for (int k0 = threadIdx.x; k0 < nm; k0 += 256 * 4) {
      double t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = ms[min(k0 + 256 * u, nm - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + 256 * u < nm) Ms[b * nm + k0 + 256 * u] = t[u];
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C * px * nb; o += 256) {
    const int b = o % nb, cx = o / nb;
    const int c = cx / px, x = cx - c * px;
    const double* mrow = Ms + b * nm + c * P;
    const double* xcol = Xs + x;
    double a0 = 0.0, a1 = 0.0;   // two chains, LDS reads of four patches issued together
    int pp = 0;
    for (; pp + 4 <= P; pp += 4) {
      const double m0 = mrow[pp], m1 = mrow[pp + 1], m2 = mrow[pp + 2], m3 = mrow[pp + 3];
      const double x0 = xcol[pp * px], x1 = xcol[(pp + 1) * px], x2 = xcol[(pp + 2) * px], x3 = xcol[(pp + 3) * px];
      a0 += m0 * x0;
      a1 += m1 * x1;
      a0 += m2 * x2;
      a1 += m3 * x3;
    }
    for (; pp < P; ++pp) a0 += mrow[pp] * xcol[pp * px];
    const double acc = a0 + a1;
    double* dst = J + s * jstride_s + c * jstride_c + x * xstride + b;
    *dst = accumulate ? *dst + acc : acc;
  }
}

// Cm[D1 x D2] (+)= A^T B, A (n, D1), B (n, D2) row-major: the dense Gram J^T (H J) of a local problem
// given by explicit Jacobian rows (several Khatri-Rao terms: tensor/network.py:207,212 evaluated on
// rows).  64 x 64 output tiles, 4 waves as 2 x 2 of 32 x 32 wave tiles; both operands are k-major in
// memory, so a 16-row slice of each is copied as it is into LDS ([k][col], stride 65), the next slice
// travelling in registers meanwhile.
__global__ __launch_bounds__(256, 4) void gemm_tn_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         double* __restrict__ Cm, int64_t n, int D1, int D2,
                                                         int accumulate) {
  __shared__ double As[16][65];
  __shared__ double Bs[16][65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int sk = tid >> 4, sc = (tid & 15) * 4;      // staged element (k = sk, columns sc .. sc + 3)
  d4 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
  double pa[4], pb[4];
  auto fetch = [&](int64_t k0) {
    const int64_t kk = min(k0 + sk, n - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[e] = A[kk * D1 + min(r0 + sc + e, D1 - 1)];
      pb[e] = B[kk * D2 + min(c0 + sc + e, D2 - 1)];
    }
  };
  fetch(0);
  for (int64_t k0 = 0; k0 < n; k0 += 16) {
    if (k0) __syncthreads();
    const bool live = k0 + sk < n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      As[sk][sc + e] = (live && r0 + sc + e < D1) ? pa[e] : 0.0;
      Bs[sk][sc + e] = (live && c0 + sc + e < D2) ? pb[e] : 0.0;
    }
    __syncthreads();
    if (k0 + 16 < n) fetch(k0 + 16);
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        af[u] = As[q + kq][wr * 32 + u * 16 + li];
        bf[u] = Bs[q + kq][wc * 32 + u * 16 + li];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = MFMA_F64(af[u], bf[v], acc[u][v]);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = r0 + wr * 32 + u * 16 + kq + 4 * t, c = c0 + wc * 32 + v * 16 + li;
        if (r < D1 && c < D2) {
          double* o = Cm + (int64_t)r * D2 + c;
          *o = accumulate ? *o + acc[u][v][t] : acc[u][v][t];
        }
      }
}

// Dense matrix -> Cholesky storage of ttk_chol_solve_f64: lower triangle of A / scale + ridge I, identity
// on the padded diagonal, zero elsewhere in the padded rows; scale[0] = mean |diag A| (1 if 0) is
// computed by dense_scale_kernel first (solve_system's scaling, tensor/network.py:296-302).
__global__ __launch_bounds__(1024) void dense_scale_kernel(const double* __restrict__ A, int D, double* __restrict__ scale) {
  __shared__ double wsum[16];
  double acc = 0.0;
  for (int k = threadIdx.x; k < D; k += 1024) acc += fabs(A[(int64_t)k * D + k]);
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

__global__ __launch_bounds__(256) void dense_to_chol_kernel(const double* __restrict__ A, int D, const double* __restrict__ scale,
                                                             double ridge, double* __restrict__ aug, int ld) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col > row || col >= ld) return;
  double v;
  if (row >= D) {
    v = (row == col) ? 1.0 : 0.0;
  } else {
    v = A[(int64_t)row * D + col];
    if (scale) v = v / scale[0];
    if (row == col) v += ridge;
  }
  aug[(int64_t)row * ld + col] = v;
}

// rows of X a workgroup of conv_feature_kernel stages (its LDS image stays below 60 KiB)

// ---------------------------------------------------------------------------------------
// Right-hand side / J^T u of a site with many features (16 <= p <= 256, rr <= 128), features first:
//   b[l, pp, r] = sum_s X[s, pp] * (GL[s, l] R[s, r]),      GL[s, l] = sum_c g[s, c] L[s, c, l]
// as a GEMM over the samples with rows pp and columns (l, r): the feature rows of a chunk of samples are
// staged once per workgroup and serve all its output columns, whereas the three-factor kernel of the few-feature
// sites would stream every slab of samples once per 96 output columns (33 times at 48 x 65 x 48).
// A wave owns NCW consecutive 16-column tiles (l, r-tile) and all PTC row tiles: PTC x NCW accumulator tiles, so
// an operand read from LDS feeds NCW (features) or PTC (right factor) MFMAs; the eight waves of a workgroup
// share the chunk images of X, R and GL (32 samples, double-buffered: one barrier per chunk, the next chunk
// travelling in registers meanwhile) and form GL[s,l] R[s,r] on the fly.  Grid = (groups of 8 NCW column tiles,
// slabs of samples); every slab writes its own partial result, a second kernel adds the slabs in fixed order.
// ---------------------------------------------------------------------------------------
#define RF_CH 32   // samples per chunk
template <int PTC, int NCW, typename TC>   // p <= 16 PTC
__global__ __launch_bounds__(512, sizeof(TC) == 4 && PTC <= 8 ? 4 : 2) void rhs_feature_kernel(
    const double* __restrict__ GL, const double* __restrict__ X, const double* __restrict__ R, double* __restrict__ partial,
    int64_t S, int rl, int p, int rr, int64_t slab_len, int gw) {
  typedef Mm<TC> mm;
  constexpr int CH = RF_CH;
  constexpr int XQ = PTC;                 // X elements per thread and chunk: CH * 16 PTC / 512
  constexpr int RQ = 8;                   // R elements per thread and chunk: CH * 128 / 512
  extern __shared__ double sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int nrt = (rr + 15) >> 4, pt = (p + 15) >> 4;
  // A chunk image is the three blocks as they lie in memory: Xs [CH][p] | Rs [CH][rr] | Gs [CH][gw].  Tile columns
  // past p (past rr) read on into the next row or block: finite values that only reach rows pp >= p (columns
  // r >= rr) of the accumulators, which are never stored.
  const int bufsz = CH * (p + rr + gw);
  TC* img = (TC*)sm;
  const int64_t s_begin = (int64_t)blockIdx.y * slab_len;
  const int64_t s_end = min(S, s_begin + slab_len);
  const int64_t D = (int64_t)rl * p * rr;
  double* dst = partial + (int64_t)blockIdx.y * D;
  const int ntiles = rl * nrt;
  const int jt_wg = blockIdx.x * (8 * NCW);         // first column tile of the workgroup
  const int l_lo = jt_wg / nrt;                     // Gs holds GL[:, l_lo .. l_lo + gw)
  // this wave's column tiles: LDS offsets of their GL and R entries (tiles past the end repeat the last one)
  int goff[NCW], roff[NCW];
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int jt = min(jt_wg + wave * NCW + c, ntiles - 1);
    const int l = jt / nrt;
    goff[c] = l - l_lo;
    roff[c] = (jt - l * nrt) * 16 + li;
  }
  const int rq = (CH * rr + 511) / 512;             // live R registers
  const int gq = (CH * gw + 511) / 512;             // live GL registers (at most 2: gw <= 32)
  TC px[XQ], pr[RQ], pg[2];
  auto fetch = [&](int64_t c0) {
    const int nvalid = (int)min((int64_t)CH, s_end - c0);
    const double* x0 = X + c0 * p + tid;
    const int lx = nvalid * p - 1 - tid;
#pragma unroll
    for (int q = 0; q < XQ; ++q) px[q] = (TC)x0[max(min(512 * q, lx), -tid)];
    const double* r0 = R + c0 * rr + tid;
    const int lr = nvalid * rr - 1 - tid;
#pragma unroll
    for (int q = 0; q < RQ; ++q)
      if (q < rq) pr[q] = (TC)r0[max(min(512 * q, lr), -tid)];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < gq) {
        const int e = min(tid + 512 * q, CH * gw - 1);
        const int k = e / gw, j = e - k * gw;
        pg[q] = (TC)GL[(c0 + min(k, nvalid - 1)) * rl + min(l_lo + j, rl - 1)];
      }
  };
  auto stage = [&](int64_t c0, int buf) {
    const int nvalid = (int)min((int64_t)CH, s_end - c0);
    TC* b = img + buf * bufsz;
    const int lx = nvalid * p - 1 - tid;
#pragma unroll
    for (int q = 0; q < XQ; ++q)
      if (tid + 512 * q < CH * p) b[tid + 512 * q] = 512 * q <= lx ? px[q] : (TC)0;     // samples past the slab: zero features
#pragma unroll
    for (int q = 0; q < RQ; ++q)
      if (q < rq && tid + 512 * q < CH * rr) b[CH * p + tid + 512 * q] = pr[q];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < gq && tid + 512 * q < CH * gw) b[CH * (p + rr) + tid + 512 * q] = pg[q];
  };
  typename mm::acc_t acc[PTC][NCW];
#pragma unroll
  for (int u = 0; u < PTC; ++u)
#pragma unroll
    for (int c = 0; c < NCW; ++c) acc[u][c] = typename mm::acc_t{0, 0, 0, 0};
  if (s_begin < s_end) {
    fetch(s_begin);
    stage(s_begin, 0);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t c0 = s_begin; c0 < s_end; c0 += CH, buf ^= 1) {
    const bool more = c0 + CH < s_end;
    if (more) fetch(c0 + CH);
    const TC* xs = img + buf * bufsz + kq * p + li;
    const TC* rs = img + buf * bufsz + CH * p + kq * rr;
    const TC* gs = img + buf * bufsz + CH * (p + rr) + kq * gw;
    // Row tiles past p are read like the others (the image has room behind it) and skipped at the products.
    TC a0[PTC], g0[NCW], r0[NCW], a1[PTC], g1[NCW], r1[NCW];
    auto ld = [&](int ks, TC* a, TC* g, TC* r) {
#pragma unroll
      for (int c = 0; c < NCW; ++c) {
        g[c] = gs[ks * 4 * gw + goff[c]];
        r[c] = rs[ks * 4 * rr + roff[c]];
      }
#pragma unroll
      for (int u = 0; u < PTC; ++u) a[u] = xs[ks * 4 * p + 16 * u];
    };
    auto mma = [&](const TC* a, const TC* g, const TC* r) {
      TC bf[NCW];
#pragma unroll
      for (int c = 0; c < NCW; ++c) bf[c] = g[c] * r[c];
#pragma unroll
      for (int u = 0; u < PTC; ++u)
        if (u < pt) {
#pragma unroll
          for (int c = 0; c < NCW; ++c) acc[u][c] = mm::fma(a[u], bf[c], acc[u][c]);
        }
    };
    if constexpr (sizeof(TC) == 8 && PTC <= 5) {
      // two waves per SIMD and registers to spare: read ahead explicitly
      ld(0, a0, g0, r0);
#pragma unroll
      for (int ks = 0; ks < CH / 4; ks += 2) {
        ld(ks + 1, a1, g1, r1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, g0, r0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < CH / 4) ld(ks + 2, a0, g0, r0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, g1, r1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // (float32: four waves per SIMD cover the operand reads of each other)
#pragma unroll 2
      for (int ks = 0; ks < CH / 4; ++ks) {
        ld(ks, a0, g0, r0);
        mma(a0, g0, r0);
      }
    }
    if (more) stage(c0 + CH, buf ^ 1);
    __syncthreads();   // the next chunk is complete, and everybody is done with this one
  }
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int jt = jt_wg + wave * NCW + c;
    if (jt < ntiles) {
      const int l = jt / nrt, r = (jt - l * nrt) * 16 + li;
      if (r < rr) {
#pragma unroll
        for (int u = 0; u < PTC; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int pp = 16 * u + mm::row(kq, t);
            if (u < pt && pp < p) dst[((int64_t)l * p + pp) * rr + r] = (double)acc[u][c][t];
          }
      }
    }
  }
}

// out[i] (+)= sum over the slabs of partial[slab][i] for many slabs and few outputs: 16 outputs x 16 slab groups per
// workgroup, every thread's loads independent and in flight together, the 16 group sums added in fixed order
__global__ __launch_bounds__(256) void dense_reduce_wide_kernel(const double* __restrict__ partial, int nslabs, int64_t n,
                                                                double* __restrict__ out, int accumulate) {
  __shared__ double part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 16 + o;
  const int64_t src = idx < n ? idx : n - 1;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = g;
  for (; k + 48 < nslabs; k += 64) {
    a0 += partial[(int64_t)k * n + src];
    a1 += partial[(int64_t)(k + 16) * n + src];
    a2 += partial[(int64_t)(k + 32) * n + src];
    a3 += partial[(int64_t)(k + 48) * n + src];
  }
  for (; k < nslabs; k += 16) a0 += partial[(int64_t)k * n + src];
  part[g][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && idx < n) {
    double acc = accumulate ? out[idx] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += part[q][o];
    out[idx] = acc;
  }
}

struct RhsFeatDims {
  int ptc, ncw, ngroups, nslabs, gw;
  int64_t slab_len;
  size_t lds;
};
template <typename TC>
static RhsFeatDims rhs_feature_dims(int64_t S, int rl, int p, int rr) {
  RhsFeatDims d;
  d.ptc = p <= 64 ? 4 : (p <= 80 ? 5 : (p <= 128 ? 8 : 16));
  d.ncw = d.ptc <= 5 ? 3 : (d.ptc == 8 ? 2 : 1);
  const int nrt = (rr + 15) / 16;
  const int per = 8 * d.ncw;                              // column tiles per workgroup
  d.ngroups = (rl * nrt + per - 1) / per;
  d.gw = (per - 1) / nrt + 2;                             // l values a run of `per` tiles can touch
  if (d.gw > rl) d.gw = rl;
  if (d.gw < 1) d.gw = 1;
  d.lds = ((size_t)2 * RF_CH * (p + rr + d.gw) + 16 * d.ptc) * sizeof(TC);   // + room for reads of row tiles past p
  // one round of workgroups (two per compute unit when the float32 images leave room), at least four chunks a slab
  const int wgs = 256 * (sizeof(TC) == 4 && d.lds <= 78 * 1024 && d.ptc * d.ncw <= 15 ? 2 : 1);
  int64_t ns = wgs / d.ngroups;
  const int64_t maxs = (S + 4 * RF_CH - 1) / (4 * RF_CH);
  if (ns > maxs) ns = maxs;
  if (ns > 128) ns = 128;
  if (ns < 1) ns = 1;
  d.slab_len = ((S + ns - 1) / ns + RF_CH - 1) / RF_CH * RF_CH;
  d.nslabs = (int)((S + d.slab_len - 1) / d.slab_len);
  return d;
}

// extents the features-first right-hand side takes: the two fp64 chunk images must fit the LDS
static bool rhs_feature_fits(int64_t S, int rl, int p, int rr) {
  return S >= 1 && rl >= 1 && p >= 16 && p <= 256 && rr >= 1 && rr <= 128 && p + rr <= 280;
}

static int conv_feature_rows(int px, int nab) {
  const int64_t room = 60 * 1024 / 8 - (int64_t)px * nab;
  int64_t rb = room / (px | 1);
  if (rb > 256) rb = 256;
  return (int)rb;
}

// Dynamic LDS above 64 KiB has to be allowed once per kernel AND per device (the attribute belongs to the
// device's copy of the code object).  flags: one per device, owned by the call site.  Two threads that race
// here both set the same attribute; nothing else is shared.
#define DENSE_MAX_DEVICES 64
static bool dense_allow_lds(const void* fn, bool* flags) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DENSE_MAX_DEVICES) return false;
  if (flags[dev]) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess) return false;
  flags[dev] = true;
  return true;
}

template <typename TC>
static bool gemm_rows_allow_lds() {
  static bool flags[7][DENSE_MAX_DEVICES];
  const void* fns[7] = {(const void*)gemm_rows_kernel<0, 0, 4, 1, TC>, (const void*)gemm_rows_kernel<1, 0, 4, 1, TC>,
                        (const void*)gemm_rows_kernel<2, 0, 4, 1, TC>, (const void*)gemm_rows_kernel<1, 1, 1, 6, TC>,
                        (const void*)gemm_rows_kernel<1, 1, 2, 3, TC>, (const void*)gemm_rows_kernel<1, 1, 3, 2, TC>,
                        (const void*)gemm_rows_kernel<1, 1, 4, 1, TC>};
  bool ok = true;
  for (int k = 0; k < 7; ++k) ok = dense_allow_lds(fns[k], flags[k]) && ok;
  return ok;
}

template <typename TC>
static int gemm_rows_launch(const double* A, const double* B, double* T, int64_t M, int K, int N, int bmode, int rl, int rr,
                            int accumulate, void* stream) {
  if (!A || !B || !T || M < 1 || K < 1 || N < 1 || bmode < 0 || bmode > 2) return TTK_EINVAL;
  if (bmode == 1 && (rl < 1 || rr < 1 || (int64_t)rl * rr != N)) return TTK_EINVAL;
  const dim3 grid((unsigned)((M + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  const int kch = gemm_rows_chunk(K);
  if (K > gemm_rows_max_k<TC>()) return TTK_EINVAL;
  const size_t lds = gemm_rows_lds<TC>(K, kch, 64);
  if (!gemm_rows_allow_lds<TC>()) return TTK_ELAUNCH;
  const double* none = nullptr;
  if (bmode == 0)
    hipLaunchKernelGGL((gemm_rows_kernel<0, 0, 4, 1, TC>), grid, dim3(256), lds, st, A, B, T, M, K, N, rl, rr, none, accumulate,
                       kch);
  else if (bmode == 1)
    hipLaunchKernelGGL((gemm_rows_kernel<1, 0, 4, 1, TC>), grid, dim3(256), lds, st, A, B, T, M, K, N, rl, rr, none, accumulate,
                       kch);
  else
    hipLaunchKernelGGL((gemm_rows_kernel<2, 0, 4, 1, TC>), grid, dim3(256), lds, st, A, B, T, M, K, N, rl, rr, none, accumulate,
                       kch);
  return dense_status();
}

template <typename TC>
static int env_right_dense_launch(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                                  int rr, int accumulate, void* stream) {
  if (!R || !X || !core || !out || S < 1 || rl < 1 || p < 1 || rr < 1 || rr > 128 || p > gemm_rows_max_k<TC>()) return TTK_EINVAL;
  if (!gemm_rows_allow_lds<TC>()) return TTK_ELAUNCH;
  const dim3 grid((unsigned)((S + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  const int kch = gemm_rows_chunk(p);
  const int N = rl * rr;
  {
    // core block resident (the usual case: p <= 96): one launch over (XCD, column block, sample range)
    const int tpl = (rr + 15) / 16, lb = tpl == 1 ? 6 : (tpl == 2 ? 3 : (tpl == 3 ? 2 : 1));
    const EnvResDims d = env_resident_dims<TC>(S, rl, p, lb, tpl * lb);
    if (!d.lds && rr > 64) return TTK_EINVAL;   // bonds above 64 only on the resident route (see ttk_env_right_dense_ok)
    if (d.lds) {
      const dim3 rgrid((unsigned)(8 * d.ncb * d.nsplit));
      const bool timed = ttk_timing_begin(4, st);
#define ER_LAUNCH(TPL_, LB_, NA_)                                                                                        \
  {                                                                                                                      \
    static bool allowed[DENSE_MAX_DEVICES];                                                                              \
    if (!dense_allow_lds((const void*)env_right_resident_kernel<TPL_, LB_, NA_, TC>, allowed)) return TTK_ELAUNCH;       \
    hipLaunchKernelGGL((env_right_resident_kernel<TPL_, LB_, NA_, TC>), rgrid, dim3(512), d.lds, st, X, core, out, S, p, \
                       rl, rr, R, accumulate, d.ncb, d.rows_xcd, d.rows_wg);                                             \
  }
#define ER_LAUNCH_NA(TPL_, LB_)       \
  if (p <= 68)                        \
    ER_LAUNCH(TPL_, LB_, 17)          \
  else                                \
    ER_LAUNCH(TPL_, LB_, 24)
      switch (tpl) {
        case 1:
          ER_LAUNCH_NA(1, 6)
          break;
        case 2:
          ER_LAUNCH_NA(2, 3)
          break;
        case 3:
          ER_LAUNCH_NA(3, 2)
          break;
        case 4:
          ER_LAUNCH_NA(4, 1)
          break;
        case 5:
          ER_LAUNCH_NA(5, 1)
          break;
        case 6:
          ER_LAUNCH_NA(6, 1)
          break;
        case 7:
          ER_LAUNCH_NA(7, 1)
          break;
        default:
          ER_LAUNCH_NA(8, 1)
          break;
      }
#undef ER_LAUNCH_NA
#undef ER_LAUNCH
      if (timed) ttk_timing_end(st);
      return dense_status();
    }
  }
  // (tiles per l) x (l per pass): six 16-wide tiles per pass where the bond allows it
  switch ((rr + 15) / 16) {
    case 1:
      hipLaunchKernelGGL((gemm_rows_kernel<1, 1, 1, 6, TC>), grid, dim3(256), gemm_rows_lds<TC>(p, kch, 96), st, X, core, out, S, p,
                         N, rl, rr, R, accumulate, kch);
      break;
    case 2:
      hipLaunchKernelGGL((gemm_rows_kernel<1, 1, 2, 3, TC>), grid, dim3(256), gemm_rows_lds<TC>(p, kch, 96), st, X, core, out, S, p,
                         N, rl, rr, R, accumulate, kch);
      break;
    case 3:
      hipLaunchKernelGGL((gemm_rows_kernel<1, 1, 3, 2, TC>), grid, dim3(256), gemm_rows_lds<TC>(p, kch, 96), st, X, core, out, S, p,
                         N, rl, rr, R, accumulate, kch);
      break;
    default:
      hipLaunchKernelGGL((gemm_rows_kernel<1, 1, 4, 1, TC>), grid, dim3(256), gemm_rows_lds<TC>(p, kch, 64), st, X, core, out, S, p,
                         N, rl, rr, R, accumulate, kch);
      break;
  }
  return dense_status();
}

template <typename TC>
static int rhs_feature_launch(const double* GL, const double* X, const double* R, double* b, int64_t S, int rl, int p, int rr,
                              int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  if (!GL || !X || !R || !b || !ws || !rhs_feature_fits(S, rl, p, rr)) return TTK_EINVAL;
  if (ws_bytes < ttk_rhs_feature_ws_bytes(S, rl, p, rr)) return TTK_EWORKSPACE;
  const RhsFeatDims d = rhs_feature_dims<TC>(S, rl, p, rr);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)d.ngroups, (unsigned)d.nslabs);
  double* partial = (double*)ws;
  const bool timed = ttk_timing_begin(3, st);
#define RF_LAUNCH(PTC_, NCW_)                                                                                          \
  {                                                                                                                    \
    static bool allowed[DENSE_MAX_DEVICES];                                                                            \
    if (!dense_allow_lds((const void*)rhs_feature_kernel<PTC_, NCW_, TC>, allowed)) return TTK_ELAUNCH;                \
    hipLaunchKernelGGL((rhs_feature_kernel<PTC_, NCW_, TC>), grid, dim3(512), d.lds, st, GL, X, R, partial, S, rl, p, \
                       rr, d.slab_len, d.gw);                                                                          \
  }
  if (d.ptc == 4)
    RF_LAUNCH(4, 3)
  else if (d.ptc == 5)
    RF_LAUNCH(5, 3)
  else if (d.ptc == 8)
    RF_LAUNCH(8, 2)
  else
    RF_LAUNCH(16, 1)
#undef RF_LAUNCH
  if (timed) ttk_timing_end(st);
  int rc = dense_status();
  if (rc != TTK_OK) return rc;
  const int64_t D = (int64_t)rl * p * rr;
  hipLaunchKernelGGL(dense_reduce_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)partial,
                     d.nslabs, D, b, accumulate);
  return dense_status();
}

// ---------------------------------------------------------------------------------------
// A column between two convolution bonds (tensor/layers.py:839-872: the left neighbour's filter leaves ca
// channels, the right one's cb): its Jacobian is the sum of ca cb Khatri-Rao terms (L_a, G_ab, R_b), all of them
// small.  Contracting the two SMALL indices around the features first keeps every term of the sum inside one
// kernel: no (S, rl rr) intermediate, one launch per operator half whatever ca cb is.
//
// bond_jv_kernel:  u[s,c] = sum_a sum_l L_a[s,c,l] z_a[s,l],
//                  z_a[s,l] = sum_b sum_p G_ab[s,p] Y_b[s,l,p],   Y_b[s,l,p] = sum_r v[l,p,r] R_b[s,r].
//   A wave owns 16 samples.  For one l, Y_b[., l, 16 features] is one MFMA tile (rows = features, columns = the
//   samples, K = r) whose result lane (sample, kq) holds four CONSECUTIVE features 16 pt + 4 kq + t (row i of the
//   tile carries feature 4 (i & 3) + (i >> 2)): every G_ab[s,p] is loaded by exactly one lane, 32 contiguous bytes
//   at a time, and multiplied in that lane; z_a[s,l] stays in the lane as a partial sum over its features and
//   meets the other three kq groups once, at the end.  v sits in LDS as [l][r][p] (r zero-padded to a multiple
//   of 4; features past p read p - 1 with a zero G factor).  z goes to memory (ca S rl numbers) and
//   bond_u_kernel closes the sum with the left environments, read as one contiguous stream.
// bond_gl_kernel:  GL_a[s,l] = sum_c g[s,c] L_a[s,c,l] for all a in one launch.
// bond_rhs_kernel: b[l,p,r] = sum_s sum_a T_a[s,p,r] GL_a[s,l],    T_a[s,p,r] = sum_b G_ab[s,p] R_b[s,r].
//   K of the MFMA is the left channel a (ca <= 4: one instruction per sample, feature tile and r): the A operand
//   lane (feature, a) forms T_a from ITS feature's G_ab[s,p] (cb loads, 16 consecutive features per row: whole
//   128-byte segments) and the wave-uniform R_b[s,r] (scalar loads); the B operand lane (a, l) is GL_a[s,l] as
//   stored.  The 8 waves split the feature tiles and the r between them; a workgroup walks one slab of samples
//   and leaves its partial sums for dense_reduce_wide_kernel.
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

template <int CA, int RLT>   // left channels; left-bond indices a lane keeps partial sums for (rl <= RLT)
__global__ __launch_bounds__(256) void bond_jv_kernel(const double* __restrict__ G,
                                                      const double* __restrict__ R, const double* __restrict__ v,
                                                      double* __restrict__ zout, int64_t S, int rl, int p, int rr, int cb) {
  extern __shared__ double bond_vs[];   // [rl][4 ksn][p]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ksn = (rr + 3) >> 2, r4 = 4 * ksn;
  const int nv = rl * r4 * p;
  for (int e = threadIdx.x; e < nv; e += 256) {
    const int pp = e % p, r = (e / p) % r4, l = e / (p * r4);
    bond_vs[e] = r < rr ? v[((int64_t)l * p + pp) * rr + r] : 0.0;
  }
  __syncthreads();
  const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (s0 >= S) return;
  const int64_t s = s0 + li;
  const bool live = s < S;
  const int64_t sc = live ? s : S - 1;
  const int64_t astride = (int64_t)cb * S * p;
  const int fo = 4 * (li & 3) + (li >> 2);   // the feature of tile row li
  const int npt = (p + 15) >> 4;
  double z[CA][RLT];
#pragma unroll
  for (int a = 0; a < CA; ++a)
#pragma unroll
    for (int l = 0; l < RLT; ++l) z[a][l] = 0.0;
  for (int b = 0; b < cb; ++b) {
    double rb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int r = 4 * ks + kq;
      rb[ks] = (live && r < rr) ? R[((int64_t)b * S + sc) * rr + r] : 0.0;
    }
    const double* g0 = G + ((int64_t)b * S + sc) * p;
    for (int pt = 0; pt < npt; ++pt) {
      const int f0 = 16 * pt + 4 * kq;
      double gv[CA][4];
      if (f0 + 4 <= p) {
        // 32 contiguous bytes of the lane's row per channel: two 16-byte loads (the rows are 8-byte aligned)
#pragma unroll
        for (int a = 0; a < CA; ++a) {
          const d2u* gp2 = (const d2u*)(g0 + a * astride + f0);
          const d2u lo = gp2[0], hi = gp2[1];
          gv[a][0] = lo[0];
          gv[a][1] = lo[1];
          gv[a][2] = hi[0];
          gv[a][3] = hi[1];
        }
      } else {
#pragma unroll
        for (int a = 0; a < CA; ++a)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const double gval = g0[a * astride + (f0 + t < p ? f0 + t : p - 1)];
            gv[a][t] = f0 + t < p ? gval : 0.0;
          }
      }
      const int fa = 16 * pt + fo < p ? 16 * pt + fo : p - 1;
      const double* vcol = bond_vs + kq * p + fa;
#pragma unroll
      for (int l = 0; l < RLT; ++l) {
        if (l < rl) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (ks < ksn) acc = MFMA_F64(vcol[(l * r4 + 4 * ks) * p], rb[ks], acc);
#pragma unroll
          for (int a = 0; a < CA; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) z[a][l] = fma(gv[a][t], acc[t], z[a][l]);
        }
      }
    }
  }
  // the four feature groups of a sample meet; lane kq of the sample stores the l = kq, kq + 4, ...
#pragma unroll
  for (int a = 0; a < CA; ++a)
#pragma unroll
    for (int l = 0; l < RLT; ++l) {
      double t = z[a][l];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (live && l < rl && (l & 3) == kq) zout[((int64_t)a * S + s) * rl + l] = t;
    }
}

// u[s,c] (+)= sum_a sum_l L_a[s,c,l] z_a[s,l]: the left environments close the sum of bond_jv_kernel.  One thread per
// (s, c): a wave reads 64 consecutive rows of L (contiguous), z[s] is the same for the C threads of a sample.
__global__ __launch_bounds__(256) void bond_u_kernel(const double* __restrict__ Lenv, const double* __restrict__ z,
                                                     double* __restrict__ u, int64_t S, int C, int rl, int ca, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * C) return;
  const int64_t s = i / C;
  double a0 = 0.0, a1 = 0.0;
  for (int a = 0; a < ca; ++a) {
    const double* lp = Lenv + ((int64_t)a * S * C + i) * rl;
    const double* zp = z + ((int64_t)a * S + s) * rl;
    int l = 0;
    for (; l + 2 <= rl; l += 2) {
      a0 = fma(lp[l], zp[l], a0);
      a1 = fma(lp[l + 1], zp[l + 1], a1);
    }
    if (l < rl) a0 = fma(lp[l], zp[l], a0);
  }
  u[i] = (accumulate ? u[i] : 0.0) + (a0 + a1);
}

// GL_a[s,l] = sum_c hu[s,c] L_a[s,c,l],  hu[s,d] = sum_c H[s,c,d] u[s,c],  u[s,c] = sum_a sum_l L_a[s,c,l] z_a[s,l]:
// prediction, Hessian product and g . L between the two halves of the operator in one kernel; the left environments
// come from memory once (the second pass over a workgroup's few samples hits L1).  A workgroup takes ns samples:
// one thread per (s, c), then one per (s, a, l).
__global__ __launch_bounds__(256) void bond_lhl_kernel(const double* __restrict__ Lenv, const double* __restrict__ H,
                                                       const double* __restrict__ z, double* __restrict__ GL, int64_t S, int C,
                                                       int rl, int ca, int ns) {
  __shared__ double ub[256], hb[256];
  const int64_t s0 = (int64_t)blockIdx.x * ns;
  const int cnt = (int)min((int64_t)ns, S - s0);
  const int t = threadIdx.x;
  if (t < cnt * C) {
    const int sl = t / C, c = t - sl * C;
    const int64_t s = s0 + sl;
    double a0 = 0.0, a1 = 0.0;
    for (int a = 0; a < ca; ++a) {
      const double* lp = Lenv + (((int64_t)a * S + s) * C + c) * rl;
      const double* zp = z + ((int64_t)a * S + s) * rl;
      int l = 0;
      for (; l + 2 <= rl; l += 2) {
        a0 = fma(lp[l], zp[l], a0);
        a1 = fma(lp[l + 1], zp[l + 1], a1);
      }
      if (l < rl) a0 = fma(lp[l], zp[l], a0);
    }
    ub[t] = a0 + a1;
  }
  __syncthreads();
  if (t < cnt * C) {
    const int sl = t / C, d = t - sl * C;
    const double* hp = H + (s0 + sl) * (int64_t)C * C + d;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc = fma(hp[(int64_t)c * C], ub[sl * C + c], acc);
    hb[t] = acc;
  }
  __syncthreads();
  const int na = ca * rl;
  if (t < cnt * na) {
    const int sl = t / na, o = t - sl * na, a = o / rl, l = o - a * rl;
    const int64_t s = s0 + sl;
    const double* lp = Lenv + ((int64_t)a * S + s) * C * rl + l;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc = fma(hb[sl * C + c], lp[(int64_t)c * rl], acc);
    GL[((int64_t)a * S + s) * rl + l] = acc;
  }
}

__global__ __launch_bounds__(256) void bond_gl_kernel(const double* __restrict__ g, const double* __restrict__ Lenv,
                                                      double* __restrict__ GL, int64_t S, int C, int rl, int ca) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)ca * S * rl) return;
  const int l = (int)(i % rl);
  const int64_t as = i / rl, s = as % S;
  double acc = 0.0;
  for (int c = 0; c < C; ++c) acc = fma(g[s * C + c], Lenv[(as * C + c) * rl + l], acc);
  GL[i] = acc;
}

template <int NR>   // right-bond indices per wave; ptw feature tiles x 8 / ptw groups of NR right-bond indices = the 8 waves
__global__ __launch_bounds__(512, 2) void bond_rhs_kernel(const double* __restrict__ GL, const double* __restrict__ G,
                                                       const double* __restrict__ R, double* __restrict__ partial, int64_t S,
                                                       int rl, int p, int rr, int ca, int cb, int64_t slab, int ptw) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;   // kq = the left channel a
  const int pt = wave % ptw, r0 = (wave / ptw) * NR;
  const int r0c = r0 + NR <= rr ? r0 : (rr - NR > 0 ? rr - NR : 0);
  const int pp = pt * 16 + li;
  const bool gok = kq < ca && pp < p, lok = kq < ca && li < rl;
  const int64_t s_lo = (int64_t)blockIdx.x * slab;
  const int64_t s_hi = s_lo + slab < S ? s_lo + slab : S;
  const int64_t bstride = (int64_t)S * p;
  const double* gp = G + (int64_t)(kq < ca ? kq : 0) * cb * bstride + (pp < p ? pp : p - 1);
  const double* glp = GL + (int64_t)(kq < ca ? kq : 0) * S * rl + (li < rl ? li : rl - 1);
  d4 acc[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  if (pt * 16 < p && r0 < rr) {
    // channels past cb read channel 0 with a zero G factor, and the wave's NR right-bond indices start at r0c <= r0
    // so that all of them exist (tiles below r0 belong to the neighbouring wave and are not stored): the loop over
    // a sample has no branches
    int64_t rof[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) rof[b] = (int64_t)(b < cb ? b : 0) * S * rr + r0c;
    double gn[4], gln;
#pragma unroll
    for (int b = 0; b < 4; ++b) gn[b] = (b < cb && gok) ? gp[b * bstride + s_lo * p] : 0.0;
    gln = lok ? glp[s_lo * rl] : 0.0;
    for (int64_t s = s_lo; s < s_hi; ++s) {
      // R_b[s, r] is the same for every lane of the wave: scalar loads, a scalar operand of the multiply-add
      const double* rs = R + s * rr;
      double rv[4][NR];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < NR; ++j) rv[b][j] = rs[rof[b] + j];
      double g[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) g[b] = gn[b];
      const double glv = gln;
      const int64_t sn = s + 1 < s_hi ? s + 1 : s;
#pragma unroll
      for (int b = 0; b < 4; ++b) gn[b] = (b < cb && gok) ? gp[b * bstride + sn * p] : 0.0;
      gln = lok ? glp[sn * rl] : 0.0;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        double t = g[0] * rv[0][j];
#pragma unroll
        for (int b = 1; b < 4; ++b) t = fma(g[b], rv[b][j], t);
        acc[j] = MFMA_F64(t, glv, acc[j]);
      }
    }
  }
  const int64_t D = (int64_t)rl * p * rr;
  double* out = partial + (int64_t)blockIdx.x * D;
  if (pt * 16 < p && li < rl) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (r0c + j < r0 || r0c + j >= rr) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int po = pt * 16 + kq + 4 * t;
        if (po < p) out[((int64_t)li * p + po) * rr + r0c + j] = acc[j][t];
      }
    }
  }
}

static bool bond_fits(int64_t S, int C, int rl, int p, int rr, int ca, int cb) {
  return S >= 1 && C >= 1 && C <= 256 && rl >= 1 && rl <= 16 && rr >= 1 && rr <= 16 && p >= 1 && p <= 128 && ca >= 1 && ca <= 4 &&
         cb >= 1 && cb <= 4 && (int64_t)p * 4 * ((rr + 3) / 4) * rl * 8 <= 65536;
}

static int64_t bond_rhs_slabs(int64_t S) {
  const int64_t n = (S + 31) / 32;
  return n < 1024 ? n : 1024;
}

// ---------------------------------------------------------------------------------------
// The operator of a conv filter node between convolution bonds without its Jacobian rows.  The rows factor as
//   J[s,c,(a,k)] = sum_l L_a[s,c,l] Z[s,l,k],   k = (x, b),   Z[s,l,(x,b)] = sum_patch (A . R_b)[s,l,patch] X[s,patch,x]
// (tensor/layers.py:839-872 contracted with both environments), and rl n + ca C rl numbers per sample (Z and the
// left environments) are several times fewer than the C ca n of the rows: every product of an iterative solve
// streams the factors instead.  filter_hjv_kernel, one wave per sample, the sample's factors in the wave's own LDS:
//   y[a,l]  = sum_k Z[l,k] v[a,k]            u[c]  = sum_{a,l} L_a[c,l] y[a,l]         (MODE 1 stores u and stops)
//   hu[d]   = sum_c H[s,c,d] u[c]                                                      (MODE 2 starts here: hu = g[s])
//   gl[a,l] = sum_c hu[c] L_a[c,l]           q[a,k] += sum_l gl[a,l] Z[l,k]
// The two products with Z are MFMA tiles with the channel a as the column (one instruction does the work of 64
// lane-serial multiply-adds: the kernel is bound by instruction issue otherwise, not by memory); q stays in the
// accumulators over the samples of a wave, the waves of a workgroup add theirs in LDS and the workgroup sums go to
// dense_reduce_wide_kernel.  The phases of a sample are wave-local: LDS operations of one wave
// complete in order, so only the compiler has to be kept from moving them.
static __device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// `doubles` consecutive numbers from memory into a wave's LDS image by LDS-DMA (a wave instruction moves 64 units
// to a wave-uniform LDS base, lane i to base + i * unit): 16-byte units when `wide` (doubles even, source and image
// 16-byte aligned), else 4-byte units.  Lanes past the end are masked off.
static __device__ __forceinline__ void wave_dma_copy(const double* __restrict__ g, int doubles, double* lds, int lane,
                                                     bool wide) {
  const char* src = (const char*)g;
  if (wide) {
    const int units = doubles >> 1;
    for (int it = 0; it * 64 < units; ++it) {
      const int u = it * 64 + lane;
      if (u < units)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)u * 16),
                                         (void __attribute__((address_space(3)))*)((char*)lds + it * 1024), 16, 0, 0);
    }
  } else {
    const int units = doubles * 2;
    for (int it = 0; it * 64 < units; ++it) {
      const int u = it * 64 + lane;
      if (u < units)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)u * 4),
                                         (void __attribute__((address_space(3)))*)((char*)lds + it * 256), 4, 0, 0);
    }
  }
}

template <int MODE, int NKT>   // NKT: 16-column tiles of q a wave keeps (n <= 16 NKT)
__global__ __launch_bounds__(256) void filter_hjv_kernel(const double* __restrict__ Lenv, const double* __restrict__ Z,
                                                         const double* __restrict__ H, const double* __restrict__ x,
                                                         double* __restrict__ out, int64_t S, int C, int rl, int n, int ca,
                                                         int nwaves, int wide_ok) {
  extern __shared__ __attribute__((aligned(16))) double fsm[];
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int na = ca * rl, ncl = C * rl, n4 = (n + 3) & ~3, zn = rl * n;
  const bool hlds = C <= 16;
  // per-wave images (16-byte aligned): Z [rl][n], L_a [C][rl] per a, H [C][C] when hlds
  const int zreg = (zn + 1) & ~1, lreg = (ncl + 1) & ~1, hreg = hlds ? (C * C + 1) & ~1 : 0;
  const int per_wave = (zreg + ca * lreg + hreg + 2 * na + ca * C + 2 * C + 1) & ~1;
  const int vreg = (ca * n4 + 1) & ~1;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* vs = fsm;                                   // [ca][n4], zero beyond n   (MODE 0, 1)
  double* Zs = fsm + vreg + wave * per_wave;
  double* Ls = Zs + zreg;
  double* Hs = Ls + ca * lreg;
  double* ys = Hs + hreg;                             // [ca rl]
  double* gs = ys + na;                               // [ca rl]
  double* ps = gs + na;                               // [ca][C]
  double* us = ps + ca * C;                           // [C]
  double* hs = us + C;                                // [C]
  if (MODE != 2) {
    for (int e = threadIdx.x; e < ca * n4; e += 256) {
      const int a = e / n4, k = e - a * n4;
      vs[e] = k < n ? x[a * n + k] : 0.0;
    }
    __syncthreads();
  }
  d4 q[NKT];   // q[kt][t] of lane (a = li, kq): column 16 kt + kq + 4 t of row a
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) q[kt] = d4{0.0, 0.0, 0.0, 0.0};
  const int gw = blockIdx.x * nwaves + wave;
  const int64_t stride = (int64_t)gridDim.x * nwaves;
  const int ksn = n4 >> 2, ksl = (rl + 3) >> 2, nkt = (n + 15) >> 4;
  const int lrow = li < rl ? li : rl - 1;
  if (wave < nwaves) {
    // A sample's factors go straight to the wave's LDS images (LDS-DMA: no staging registers, all loads of the
    // sample in flight together): one memory round trip per sample, hidden by the other waves of the CU.
    const bool wide = wide_ok != 0;   // every block a whole number of 16-byte units at a 16-byte aligned address
    for (int64_t s = gw; s < S; s += stride) {
      wave_dma_copy(Z + s * (int64_t)zn, zn, Zs, lane, wide);
      for (int a = 0; a < ca; ++a) wave_dma_copy(Lenv + ((int64_t)a * S + s) * ncl, ncl, Ls + a * lreg, lane, wide);
      if (MODE == 0 && hlds) wave_dma_copy(H + s * (int64_t)C * C, C * C, Hs, lane, wide);
      double hrow = 0.0;
      if (MODE == 2) hrow = x[s * C + min(lane, C - 1)];
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the images have landed
      wave_lds_order();
      if (MODE != 2) {
        // y[a,l] = sum_k Z[l,k] v[a,k]: one MFMA tile (rows l, columns a), K = k
        d4 yacc = {0.0, 0.0, 0.0, 0.0};
        const double* zr = Zs + lrow * n;
        const double* vr = vs + (li < ca ? li : 0) * n4;
#pragma unroll 4
        for (int ks = 0; ks < ksn; ++ks) {
          const int k = 4 * ks + kq;
          const double av = zr[k < n ? k : n - 1];
          const double bv = li < ca ? vr[k] : 0.0;
          yacc = MFMA_F64(av, bv, yacc);
        }
        if (li < ca) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (kq + 4 * t < rl) ys[li * rl + kq + 4 * t] = yacc[t];
        }
        wave_lds_order();
        // u[c] = sum_a sum_l L_a[c,l] y[a,l]: one lane per (a, c), then the ca partial sums of a c
        for (int o = lane; o < ca * C; o += 64) {
          const int a = o / C;
          const double* lr = Ls + a * lreg + (o - a * C) * rl;
          const double* yr = ys + a * rl;
          double acc = 0.0;
#pragma unroll 4
          for (int l = 0; l < rl; ++l) acc = fma(lr[l], yr[l], acc);
          ps[o] = acc;
        }
        wave_lds_order();
        if (lane < C) {
          double acc = 0.0;
          for (int a = 0; a < ca; ++a) acc += ps[a * C + lane];
          if (MODE == 1)
            out[s * C + lane] = acc;
          else
            us[lane] = acc;
        }
        wave_lds_order();
        if (MODE == 1) continue;
        if (lane < C) {
          double acc = 0.0;
          if (hlds) {
#pragma unroll 4
            for (int c = 0; c < C; ++c) acc = fma(Hs[c * C + lane], us[c], acc);
          } else {
            const double* hp = H + s * (int64_t)C * C + lane;
            for (int c = 0; c < C; ++c) acc = fma(hp[(int64_t)c * C], us[c], acc);
          }
          hrow = acc;
        }
      }
      if (lane < C) hs[lane] = hrow;
      wave_lds_order();
      if (lane < na) {
        const int a = lane / rl, l = lane - a * rl;
        double acc = 0.0;
#pragma unroll 4
        for (int c = 0; c < C; ++c) acc = fma(hs[c], Ls[a * lreg + c * rl + l], acc);
        gs[lane] = acc;
      }
      wave_lds_order();
      // q[a,k] += sum_l gl[a,l] Z[l,k]: tiles with rows k, columns a, K = l
      double gb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int l = 4 * ks + kq;
        gb[ks] = (ks < ksl && li < ca && l < rl) ? gs[li * rl + l] : 0.0;
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nkt) {
          const int k = 16 * kt + li;
          const double* zc = Zs + (k < n ? k : n - 1);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if (ks < ksl) {
              const int l = 4 * ks + kq;
              q[kt] = MFMA_F64(zc[(l < rl ? l : rl - 1) * n], gb[ks], q[kt]);
            }
          }
        }
      }
      wave_lds_order();
    }
  }
  if (MODE == 1) return;
  // the waves' sums, added in wave order; the wave regions are free (all waves are past their samples)
  __syncthreads();
  double* red = fsm + vreg;   // [nwaves][ca n]
  if (wave < nwaves && li < ca) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 16 * kt + kq + 4 * t;
        if (k < n) red[wave * ca * n + li * n + k] = q[kt][t];
      }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ca * n; e += 256) {
    double acc = 0.0;
    for (int w = 0; w < nwaves; ++w) acc += red[w * ca * n + e];
    out[(int64_t)blockIdx.x * ca * n + e] = acc;
  }
}

struct FilterDims {
  int nwaves, grid;
  size_t lds;
};

static bool filter_dims(int64_t S, int C, int rl, int n, int ca, FilterDims* d) {
  if (S < 1 || C < 1 || C > 64 || rl < 1 || rl > 16 || n < 1 || n > 256 || ca < 1 || ca > 4) return false;
  const size_t zreg = ((size_t)rl * n + 1) & ~(size_t)1, lreg = ((size_t)C * rl + 1) & ~(size_t)1;
  const size_t hreg = C <= 16 ? ((size_t)C * C + 1) & ~(size_t)1 : 0;
  const size_t per_wave =
      ((zreg + ca * lreg + hreg + 2 * (size_t)ca * rl + (size_t)ca * C + 2 * (size_t)C + 1) & ~(size_t)1) * sizeof(double);
  const size_t shared = (((size_t)ca * ((n + 3) & ~3) + 1) & ~(size_t)1) * sizeof(double);
  if (per_wave < (size_t)ca * n * sizeof(double)) return false;   // the wave sums reuse the wave regions
  int nw = 4;
  while (nw > 0 && shared + nw * per_wave > 64 * 1024) --nw;
  if (nw < 1) return false;
  d->nwaves = nw;
  d->lds = shared + nw * per_wave;
  const int64_t want = (S + nw - 1) / nw;
  const int per_cu = (int)((160 * 1024) / d->lds);                   // workgroups of a CU by LDS (at most 4 by waves)
  const int64_t cap = 256 * (int64_t)(per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu);
  d->grid = (int)(want < cap ? want : cap);
  return true;
}

// ---------------------------------------------------------------------------------------
// Front-ends that turn raw inputs into the per-site feature tensors the columns read.
//
// feature_map_kernel: the local feature maps of TNMLRegressor (models/tnml.py:11-23) for all N sites in one
// launch: out[i][s][:] = [x^0 .. x^degree] (kind 0) or [cos(pi x / 2), sin(pi x / 2)] (kind 1) of x = X[s][i].
// out is (N, S, p): site i's (S, p) tensor is one contiguous slice.  Powers 0..3 are 1, x, x*x, x*x*x (what
// torch's x ** d evaluates for those exponents), higher ones pow(x, d).
//
// patchify_kernel: non-overlapping / strided k x k patches of an image batch (S, Cin, H, W) as rows
// (S, patches + 1, Cin k k + 1) with a zero last row, a zero last column and a one in the corner:
// torch.nn.functional.unfold(...).transpose(-2, -1) followed by the two torch.cat and the corner assignment of
// image_convolution_lanczos_MNIST.py:29-32 (pixel index = c k k + ki k + kj, patch index = oh OW + ow).
// Both are byte shuffles bound by HBM: consecutive lanes write consecutive output elements.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feature_map_kernel(const double* __restrict__ X, int64_t S, int N, int kind, int p,
                                                           double* __restrict__ out) {
  const int64_t total = (int64_t)N * S * p;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx % p);
    const int64_t is = idx / p;
    const int64_t s = is % S;
    const int i = (int)(is / S);
    const double x = X[s * N + i];
    double v;
    if (kind == 0) {
      v = d == 0 ? 1.0 : d == 1 ? x : d == 2 ? x * x : d == 3 ? x * x * x : pow(x, (double)d);
    } else {
      const double a = 1.5707963267948966 * x;
      v = d == 0 ? cos(a) : sin(a);
    }
    out[idx] = v;
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* __restrict__ img, int64_t S, int Cin, int H, int W, int k,
                                                        int stride, int OH, int OW, TO* __restrict__ out) {
  const int px = Cin * k * k + 1, P = OH * OW + 1;
  const int64_t total = S * P * px;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int x = (int)(idx % px);
    const int64_t sp = idx / px;
    const int pt = (int)(sp % P);
    const int64_t s = sp / P;
    TO v = (TO)0;
    if (pt == P - 1 || x == px - 1) {
      v = (pt == P - 1 && x == px - 1) ? (TO)1 : (TO)0;
    } else {
      const int c = x / (k * k), kk = x - c * k * k, ki = kk / k, kj = kk - ki * k;
      const int oh = pt / OW, ow = pt - oh * OW;
      v = (TO)img[((s * Cin + c) * H + (oh * stride + ki)) * W + (ow * stride + kj)];
    }
    out[idx] = v;
  }
}

template <typename TI>
static int patch_contract_launch(const double* M, const TI* X, double* J, int64_t S, int C, int P, int px, int nb,
                                 int64_t mstride_b, int64_t jstride_s, int64_t jstride_c, int64_t xstride, int accumulate,
                                 void* stream) {
  if (!M || !X || !J || S < 1 || C < 1 || P < 1 || px < 1 || nb < 1) return TTK_EINVAL;
  const size_t lds = ((size_t)P * px + (size_t)nb * C * P) * sizeof(double);
  if (lds > 64 * 1024) return TTK_EINVAL;
  hipLaunchKernelGGL(patch_contract_kernel<TI>, dim3((unsigned)S), dim3(256), lds, (hipStream_t)stream, M, X, J, C, P, px,
                     jstride_s, jstride_c, xstride, accumulate, nb, mstride_b);
  return dense_status();
}

extern "C" {

int ttk_feature_map_f64(const double* X, int64_t S, int N, int kind, int p, double* out, void* stream) {
  if (!X || !out || S < 1 || N < 1 || kind < 0 || kind > 1 || p < 1 || (kind == 1 && p != 2)) return TTK_EINVAL;
  const int64_t total = (int64_t)N * S * p;
  int64_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(feature_map_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, S, N, kind, p, out);
  return dense_status();
}

// in / out element types: 0 float32 -> float32, 1 float64 -> float64, 2 float32 -> float64
int ttk_patchify(const void* img, int64_t S, int Cin, int H, int W, int k, int stride, int types, void* out, void* stream) {
  if (!img || !out || S < 1 || Cin < 1 || k < 1 || stride < 1 || H < k || W < k || types < 0 || types > 2) return TTK_EINVAL;
  const int OH = (H - k) / stride + 1, OW = (W - k) / stride + 1;
  const int64_t total = S * (int64_t)(OH * OW + 1) * (int64_t)(Cin * k * k + 1);
  int64_t grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipStream_t st = (hipStream_t)stream;
  if (types == 0)
    hipLaunchKernelGGL((patchify_kernel<float, float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)img, S, Cin, H, W, k,
                       stride, OH, OW, (float*)out);
  else if (types == 1)
    hipLaunchKernelGGL((patchify_kernel<double, double>), dim3((unsigned)grid), dim3(256), 0, st, (const double*)img, S, Cin, H, W,
                       k, stride, OH, OW, (double*)out);
  else
    hipLaunchKernelGGL((patchify_kernel<float, double>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)img, S, Cin, H, W, k,
                       stride, OH, OW, (double*)out);
  return dense_status();
}

int ttk_filter_ok(int64_t S, int C, int rl, int n, int ca) {
  FilterDims d;
  return filter_dims(S, C, rl, n, ca, &d) ? 1 : 0;
}

int64_t ttk_filter_ws_bytes(int64_t S, int C, int rl, int n, int ca) {
  FilterDims d;
  if (!filter_dims(S, C, rl, n, ca, &d)) return 0;
  return (int64_t)d.grid * ca * n * (int64_t)sizeof(double);
}

static int filter_launch(int mode, const double* L, const double* Z, const double* H, const double* x, double* out, int64_t S,
                         int C, int rl, int n, int ca, void* ws, int64_t ws_bytes, void* stream) {
  FilterDims d;
  if (!L || !Z || !x || !out || !filter_dims(S, C, rl, n, ca, &d)) return TTK_EINVAL;
  if (mode == 0 && !H) return TTK_EINVAL;
  if (mode != 1 && (!ws || ws_bytes < ttk_filter_ws_bytes(S, C, rl, n, ca))) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  // 16-byte DMA units need even block lengths and 16-byte aligned bases (the C ABI promises 8)
  const bool hl = mode == 0 && C <= 16;
  const int wide = ((rl * n) % 2 == 0 && (C * rl) % 2 == 0 && (!hl || (C * C) % 2 == 0) &&
                    (((uintptr_t)L | (uintptr_t)Z | (hl ? (uintptr_t)H : 0)) & 15) == 0)
                       ? 1
                       : 0;
#define FILTER_GO(MODE_, NKT_, OUT_)                                                                                  \
  hipLaunchKernelGGL((filter_hjv_kernel<MODE_, NKT_>), dim3(d.grid), dim3(256), d.lds, st, L, Z, H, x, OUT_, S, C, rl, n, ca, \
                     d.nwaves, wide)
#define FILTER_NKT(MODE_)                                                                                              \
  {                                                                                                                    \
    const int nkt = (n + 15) / 16;                                                                                     \
    if (nkt <= 2)                                                                                                      \
      FILTER_GO(MODE_, 2, partial);                                                                                    \
    else if (nkt <= 5)                                                                                                 \
      FILTER_GO(MODE_, 5, partial);                                                                                    \
    else if (nkt <= 8)                                                                                                 \
      FILTER_GO(MODE_, 8, partial);                                                                                    \
    else if (nkt <= 13)                                                                                                \
      FILTER_GO(MODE_, 13, partial);                                                                                   \
    else                                                                                                               \
      FILTER_GO(MODE_, 16, partial);                                                                                   \
  }
  if (mode == 0)
    FILTER_NKT(0)
  else if (mode == 1)
    FILTER_GO(1, 1, out);
  else
    FILTER_NKT(2)
#undef FILTER_NKT
#undef FILTER_GO
  int rc = dense_status();
  if (rc != TTK_OK || mode == 1) return rc;
  const int64_t D = (int64_t)ca * n;
  hipLaunchKernelGGL(dense_reduce_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)partial, d.grid, D,
                     out, 0);
  return dense_status();
}

int ttk_filter_matvec_f64(const double* L, const double* Z, const double* H, const double* v, double* q, int64_t S, int C,
                          int rl, int n, int ca, void* ws, int64_t ws_bytes, void* stream) {
  return filter_launch(0, L, Z, H, v, q, S, C, rl, n, ca, ws, ws_bytes, stream);
}

int ttk_filter_jv_f64(const double* L, const double* Z, const double* v, double* u, int64_t S, int C, int rl, int n, int ca,
                      void* stream) {
  return filter_launch(1, L, Z, nullptr, v, u, S, C, rl, n, ca, nullptr, 0, stream);
}

int ttk_filter_rhs_f64(const double* L, const double* Z, const double* g, double* b, int64_t S, int C, int rl, int n, int ca,
                       void* ws, int64_t ws_bytes, void* stream) {
  return filter_launch(2, L, Z, nullptr, g, b, S, C, rl, n, ca, ws, ws_bytes, stream);
}

int ttk_bond_ok(int64_t S, int C, int rl, int p, int rr, int ca, int cb) { return bond_fits(S, C, rl, p, rr, ca, cb) ? 1 : 0; }

int64_t ttk_bond_jv_ws_bytes(int64_t S, int rl, int ca) {
  if (S < 1 || rl < 1 || ca < 1) return 0;
  return (int64_t)ca * S * rl * (int64_t)sizeof(double);
}

static int bond_z_launch(const double* G, const double* R, const double* v, double* z, int64_t S, int rl, int p, int rr, int ca,
                         int cb, hipStream_t st) {
  const int ks = (rr + 3) / 4;
  const dim3 grid((unsigned)((S + 63) / 64));
  const size_t lds = (size_t)p * 4 * ks * rl * sizeof(double);
  const bool timed = ttk_timing_begin(6, st);
#define BOND_JV(CA_, RLT_)                                                                                             \
  hipLaunchKernelGGL((bond_jv_kernel<CA_, RLT_>), grid, dim3(256), lds, st, G, R, v, z, S, rl, p, rr, cb)
#define BOND_JV_RL(CA_)                                                                                                \
  {                                                                                                                    \
    if (rl <= 4)                                                                                                       \
      BOND_JV(CA_, 4);                                                                                                 \
    else if (rl <= 8)                                                                                                  \
      BOND_JV(CA_, 8);                                                                                                 \
    else if (rl <= 12)                                                                                                 \
      BOND_JV(CA_, 12);                                                                                                \
    else                                                                                                               \
      BOND_JV(CA_, 16);                                                                                                \
  }
  switch (ca) {
    case 1: BOND_JV_RL(1) break;
    case 2: BOND_JV_RL(2) break;
    case 3: BOND_JV_RL(3) break;
    default: BOND_JV_RL(4) break;
  }
#undef BOND_JV_RL
#undef BOND_JV
  if (timed) ttk_timing_end(st);
  return dense_status();
}

int ttk_bond_z_f64(const double* G, const double* R, const double* v, double* z, int64_t S, int rl, int p, int rr, int ca, int cb,
                   void* stream) {
  if (!G || !R || !v || !z || !bond_fits(S, 1, rl, p, rr, ca, cb)) return TTK_EINVAL;
  return bond_z_launch(G, R, v, z, S, rl, p, rr, ca, cb, (hipStream_t)stream);
}

int ttk_bond_jv_f64(const double* L, const double* G, const double* R, const double* v, double* u, int64_t S, int C, int rl,
                    int p, int rr, int ca, int cb, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  if (!L || !G || !R || !v || !u || !ws || !bond_fits(S, C, rl, p, rr, ca, cb)) return TTK_EINVAL;
  if (ws_bytes < ttk_bond_jv_ws_bytes(S, rl, ca)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* z = (double*)ws;
  int rc = bond_z_launch(G, R, v, z, S, rl, p, rr, ca, cb, st);
  if (rc != TTK_OK) return rc;
  const int64_t nu = S * C;
  hipLaunchKernelGGL(bond_u_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, L, (const double*)z, u, S, C, rl, ca,
                     accumulate);
  return dense_status();
}

int ttk_bond_lhl_f64(const double* L, const double* H, const double* z, double* GL, int64_t S, int C, int rl, int ca,
                     void* stream) {
  if (!L || !H || !z || !GL || S < 1 || C < 1 || rl < 1 || ca < 1 || C > 256 || ca * rl > 256) return TTK_EINVAL;
  const int per = C > ca * rl ? C : ca * rl;
  const int ns = 256 / per;
  hipLaunchKernelGGL(bond_lhl_kernel, dim3((unsigned)((S + ns - 1) / ns)), dim3(256), 0, (hipStream_t)stream, L, H, z, GL, S, C,
                     rl, ca, ns);
  return dense_status();
}

int ttk_bond_gl_f64(const double* g, const double* L, double* GL, int64_t S, int C, int rl, int ca, void* stream) {
  if (!g || !L || !GL || S < 1 || C < 1 || rl < 1 || ca < 1) return TTK_EINVAL;
  const int64_t n = (int64_t)ca * S * rl;
  hipLaunchKernelGGL(bond_gl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, L, GL, S, C, rl, ca);
  return dense_status();
}

int64_t ttk_bond_rhs_ws_bytes(int64_t S, int rl, int p, int rr) {
  if (S < 1 || rl < 1 || p < 1 || rr < 1) return 0;
  return bond_rhs_slabs(S) * (int64_t)rl * p * rr * (int64_t)sizeof(double);
}

int ttk_bond_rhs_f64(const double* GL, const double* G, const double* R, double* b, int64_t S, int rl, int p, int rr, int ca,
                     int cb, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  if (!GL || !G || !R || !b || !ws || !bond_fits(S, 1, rl, p, rr, ca, cb)) return TTK_EINVAL;
  if (ws_bytes < ttk_bond_rhs_ws_bytes(S, rl, p, rr)) return TTK_EWORKSPACE;
  const int64_t nslabs = bond_rhs_slabs(S);
  const int64_t slab = (S + nslabs - 1) / nslabs;
  const int64_t used = (S + slab - 1) / slab;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  const bool timed = ttk_timing_begin(5, st);
  const int ptw = p <= 16 ? 1 : p <= 32 ? 2 : p <= 64 ? 4 : 8;      // feature tiles (waves along p); 8 / ptw groups of r
  const int nr = (rr + 8 / ptw - 1) / (8 / ptw);                    // right-bond indices per wave
#define BOND_RHS(NR_)                                                                                                  \
  hipLaunchKernelGGL(bond_rhs_kernel<NR_>, dim3((unsigned)used), dim3(512), 0, st, GL, G, R, partial, S, rl, p, rr, ca, cb, \
                     slab, ptw)
  switch (nr) {
    case 1: BOND_RHS(1); break;
    case 2: BOND_RHS(2); break;
    case 3: BOND_RHS(3); break;
    case 4: BOND_RHS(4); break;
    case 5: BOND_RHS(5); break;
    case 6: BOND_RHS(6); break;
    case 7: BOND_RHS(7); break;
    case 8: BOND_RHS(8); break;
    default: BOND_RHS(16); break;
  }
#undef BOND_RHS
  if (timed) ttk_timing_end(st);
  int rc = dense_status();
  if (rc != TTK_OK) return rc;
  const int64_t D = (int64_t)rl * p * rr;
  hipLaunchKernelGGL(dense_reduce_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)partial,
                     (int)used, D, b, accumulate);
  return dense_status();
}

int ttk_conv_feature_f64(const double* X, const double* F, double* G, int64_t rows, int px, int nab, void* stream) {
  if (!X || !F || !G || rows < 1 || px < 1 || nab < 1) return TTK_EINVAL;
  const int RB = conv_feature_rows(px, nab);
  if (RB < 1) return TTK_EINVAL;
  const size_t lds = ((size_t)px * nab + (size_t)RB * (px | 1)) * sizeof(double);
  hipLaunchKernelGGL(conv_feature_kernel<double>, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), lds, (hipStream_t)stream,
                     X, F, G, rows, px, nab, RB);
  return dense_status();
}

int ttk_conv_feature_f32(const float* X, const double* F, double* G, int64_t rows, int px, int nab, void* stream) {
  if (!X || !F || !G || rows < 1 || px < 1 || nab < 1) return TTK_EINVAL;
  const int RB = conv_feature_rows(px, nab);
  if (RB < 1) return TTK_EINVAL;
  const size_t lds = ((size_t)px * nab + (size_t)RB * (px | 1)) * sizeof(double);
  hipLaunchKernelGGL(conv_feature_kernel<float>, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), lds, (hipStream_t)stream,
                     X, F, G, rows, px, nab, RB);
  return dense_status();
}

int ttk_gemm_rows_f64(const double* A, const double* B, double* T, int64_t M, int K, int N, int bmode, int rl, int rr,
                      int accumulate, void* stream) {
  return gemm_rows_launch<double>(A, B, T, M, K, N, bmode, rl, rr, accumulate, stream);
}

int ttk_gemm_rows_mm32(const double* A, const double* B, double* T, int64_t M, int K, int N, int bmode, int rl, int rr,
                       int accumulate, void* stream) {
  return gemm_rows_launch<float>(A, B, T, M, K, N, bmode, rl, rr, accumulate, stream);
}

int ttk_env_right_dense_ok(int64_t S, int rl, int p, int rr, int mm32) {
  if (S < 1 || rl < 1 || p < 1 || rr < 1 || rr > 128 || p > (mm32 ? gemm_rows_max_k<float>() : gemm_rows_max_k<double>())) return 0;
  if (rr <= 64) return 1;
  const int tpl = (rr + 15) / 16;
  return (mm32 ? env_resident_dims<float>(S, rl, p, 1, tpl).lds : env_resident_dims<double>(S, rl, p, 1, tpl).lds) ? 1 : 0;
}

int ttk_env_right_dense_f64(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                            int rr, int accumulate, void* stream) {
  return env_right_dense_launch<double>(R, X, core, out, S, rl, p, rr, accumulate, stream);
}

int ttk_env_right_dense_mm32(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                             int rr, int accumulate, void* stream) {
  return env_right_dense_launch<float>(R, X, core, out, S, rl, p, rr, accumulate, stream);
}

int ttk_lcontract_strided_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n,
                              int64_t ostride_s, int64_t ostride_c, int accumulate, void* stream) {
  if (!L || !T || !out || S < 1 || C < 1 || rl < 1 || n < 1 || ostride_c < n || ostride_s < (int64_t)C * n) return TTK_EINVAL;
  int spb = n >= 256 ? 1 : 256 / n;
  const int fit = (int)(48 * 1024 / ((size_t)C * rl * sizeof(double)));
  if (fit < 1) return TTK_EINVAL;
  if (spb > fit) spb = fit;
  const size_t lds = (size_t)spb * C * rl * sizeof(double);
  hipLaunchKernelGGL(lcontract_kernel, dim3((unsigned)((S + spb - 1) / spb)), dim3(256), lds, (hipStream_t)stream, L, T, out,
                     S, C, rl, n, spb, accumulate, ostride_s, ostride_c);
  return dense_status();
}

int ttk_lcontract_sum_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n, int ca, int nb,
                          void* stream) {
  if (!L || !T || !out || S < 1 || C < 1 || rl < 1 || n < 1 || ca < 1 || nb < 1 || nb > 65535) return TTK_EINVAL;
  int spb = n >= 256 ? 1 : 256 / n;
  const int fit = (int)(48 * 1024 / ((size_t)ca * C * rl * sizeof(double)));
  if (fit < 1) return TTK_EINVAL;
  if (spb > fit) spb = fit;
  const size_t lds = (size_t)ca * spb * C * rl * sizeof(double);
  hipLaunchKernelGGL(lcontract_sum_kernel, dim3((unsigned)((S + spb - 1) / spb), (unsigned)nb), dim3(256), lds,
                     (hipStream_t)stream, L, T, out, S, C, rl, n, spb, ca, nb);
  return dense_status();
}

int ttk_lcontract_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n, int accumulate,
                      void* stream) {
  return ttk_lcontract_strided_f64(L, T, out, S, C, rl, n, (int64_t)C * n, n, accumulate, stream);
}

int ttk_patch_contract_f64(const double* M, const double* X, double* J, int64_t S, int C, int P, int px, int64_t jstride_s,
                           int64_t jstride_c, int64_t xstride, int accumulate, void* stream) {
  return patch_contract_launch<double>(M, X, J, S, C, P, px, 1, 0, jstride_s, jstride_c, xstride, accumulate, stream);
}

int ttk_patch_contract_f32(const double* M, const float* X, double* J, int64_t S, int C, int P, int px, int64_t jstride_s,
                           int64_t jstride_c, int64_t xstride, int accumulate, void* stream) {
  return patch_contract_launch<float>(M, X, J, S, C, P, px, 1, 0, jstride_s, jstride_c, xstride, accumulate, stream);
}

int ttk_patch_contract_multi_ok(int C, int P, int px, int nb) {
  return C >= 1 && P >= 1 && px >= 1 && nb >= 1 && ((size_t)P * px + (size_t)nb * C * P) * sizeof(double) <= 64 * 1024 ? 1 : 0;
}

int ttk_patch_contract_multi_f64(const double* M, const double* X, double* J, int64_t S, int C, int P, int px, int nb,
                                 void* stream) {
  return patch_contract_launch<double>(M, X, J, S, C, P, px, nb, S * (int64_t)C * P, (int64_t)C * px * nb, (int64_t)px * nb,
                                       nb, 0, stream);
}

int ttk_patch_contract_multi_f32(const double* M, const float* X, double* J, int64_t S, int C, int P, int px, int nb,
                                 void* stream) {
  return patch_contract_launch<float>(M, X, J, S, C, P, px, nb, S * (int64_t)C * P, (int64_t)C * px * nb, (int64_t)px * nb, nb,
                                      0, stream);
}

int ttk_gemm_tn_f64(const double* A, const double* B, double* Cm, int64_t n, int D1, int D2, int accumulate, void* stream) {
  if (!A || !B || !Cm || n < 1 || D1 < 1 || D2 < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((D2 + 63) / 64), (unsigned)((D1 + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, A, B, Cm, n, D1, D2, accumulate);
  return dense_status();
}

int64_t ttk_rhs_feature_ws_bytes(int64_t S, int rl, int p, int rr) {
  if (!rhs_feature_fits(S, rl, p, rr)) return TTK_EINVAL;
  // the float32 launch may use more (never fewer than half as many) slabs: size for the larger of the two
  const int ns = max(rhs_feature_dims<double>(S, rl, p, rr).nslabs, rhs_feature_dims<float>(S, rl, p, rr).nslabs);
  return (int64_t)ns * rl * p * rr * (int64_t)sizeof(double);
}

int ttk_rhs_feature_f64(const double* GL, const double* X, const double* R, double* b, int64_t S, int rl, int p, int rr,
                        int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  return rhs_feature_launch<double>(GL, X, R, b, S, rl, p, rr, accumulate, ws, ws_bytes, stream);
}

int ttk_rhs_feature_mm32(const double* GL, const double* X, const double* R, double* b, int64_t S, int rl, int p, int rr,
                         int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  return rhs_feature_launch<float>(GL, X, R, b, S, rl, p, rr, accumulate, ws, ws_bytes, stream);
}

int ttk_dense_scale_f64(const double* A, int D, double* scale, void* stream) {
  if (!A || !scale || D < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(dense_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A, D, scale);
  return dense_status();
}

int ttk_chol_prepare_dense_f64(const double* A, int D, const double* scale, double ridge, const double* b, const double* w,
                               void* chol_ws, void* stream) {
  if (!A || !b || !chol_ws || D < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (D + 63) / 64, ld = nb * 64;
  double* aug = (double*)chol_ws;
  hipLaunchKernelGGL(dense_to_chol_kernel, dim3((unsigned)((ld + 255) / 256), (unsigned)ld), dim3(256), 0, st, A, D, scale, ridge,
                     aug, ld);
  if (hipMemsetAsync(aug + (int64_t)ld * ld, 0, (size_t)64 * ld * sizeof(double), st) != hipSuccess) return TTK_ELAUNCH;
  return ttk_solve_rhs_f64(b, w, scale, ridge, aug + (int64_t)ld * ld, (int64_t)D, stream);
}

}  // extern "C"
