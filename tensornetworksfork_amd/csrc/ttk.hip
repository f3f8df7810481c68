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
#include <cstdlib>
#include <mutex>
#include <stdint.h>
#include <type_traits>

#include "ttk.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

static inline int ttk_launch_status() { return hipGetLastError() == hipSuccess ? TTK_OK : TTK_ELAUNCH; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Dynamic LDS above 64 KiB has to be allowed once per kernel and per device (the attribute belongs to the
// device's copy of the code object).  flags: one per device, owned by the call site; threads that race here
// set the same attribute twice, nothing else is shared.
#define TTK_MAX_DEVICES 64
static bool allow_big_lds(const void* fn, bool* flags, int bytes) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TTK_MAX_DEVICES) return false;
  if (flags[dev]) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  flags[dev] = true;
  return true;
}

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
__global__ __launch_bounds__(256, 2) void env_step_kernel(const double* __restrict__ E, const double* __restrict__ X,
                                                        const double* __restrict__ core, double* __restrict__ out,
                                                        int64_t M, int C, int e, int p, int n, int KC, int accumulate) {
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
  // rows past the end are clamped, not predicated (a load under a per-element branch is waited
  // for one by one); they are zeroed after the load and never stored
  for (int idx = lane; idx < 16 * e; idx += 64) {
    const int ii = idx & 15, l = idx >> 4;
    const int64_t row = row0 + ii;
    const double v = E[min(row, M - 1) * e + l];
    Ew[l * 16 + ii] = row < M ? v : 0.0;
  }
  for (int idx = lane; idx < 16 * p; idx += 64) {
    const int ii = idx & 15, pp = idx >> 4;
    const int64_t row = row0 + ii;
    const double v = X[(min(row, M - 1) / C) * p + pp];
    Xw[pp * 16 + ii] = row < M ? v : 0.0;
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
        v = core[(int64_t)min(kk, K - 1) * n + nb0 + min(nn, nloc - 1)];
        if (kk >= K || nn >= nloc) v = 0.0;
      } else {
        nn = idx / KC;
        k = idx - nn * KC;
        const int kk = kc0 + k;
        v = core[(int64_t)(nb0 + min(nn, nloc - 1)) * K + min(kk, K - 1)];
        if (kk >= K || nn >= nloc) v = 0.0;
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
        if (row < M && col < nloc) {
          double* o = out + row * n + nb0 + col;
          *o = accumulate ? *o + acc[j][t] : acc[j][t];
        }
      }
    }
  }
}

// Environment step with the whole core resident in LDS (K * npad doubles), for the common case of
// a small core and very many rows.  A workgroup stages the core once, then its four waves walk
// over 16-row tiles (grid stride): the E and X rows of a tile are fetched into registers (the
// next tile's while the current one is multiplied), transposed into a per-wave LDS tile, and the
// left operand E[row,ei] * X[row,pi] is formed per lane from a (ei, pi) offset table.
// mode 0: out[row, :] (+)= result;  mode 1: out[row] (+)= <result row, Rdot[row / C, :]> [* Hmul[row]]  (J v).
//
// For the right-hand step the core rows in LDS are column-permuted: element (k, j) sits at column
// j ^ swz(k), which spreads a column of the core over all banks (that step stores the core
// transposed, lanes along k).  The left-hand step keeps the core exactly as it is in memory.
#define ENVF_SLOTS 12   // prefetch registers per lane: ceil(e / 4) + ceil(p / 4) of them are used
template <bool LEFT, int NT>
__global__ __launch_bounds__(256, 2) void env_fast_kernel(const double* __restrict__ E, const double* __restrict__ X,
                                                           const double* __restrict__ core, double* __restrict__ out,
                                                           int Mi, int C, int e, int p, int n, int accumulate,
                                                           int mode, const double* __restrict__ Rdot,
                                                           const double* __restrict__ Hmul) {
  extern __shared__ double smem[];
  constexpr int npad = NT * 16;
  constexpr int SW = (NT % 2 == 0) ? 1 : 0;
  const int K = e * p;
  const int Kp = (K + 15) & ~15;                       // the k loop runs in groups of 16; rows K..Kp are zero
  double* Bs = smem;                                   // [Kp][npad]
  int* ktab = (int*)(Bs + Kp * npad);                  // [Kp] packed offsets into the wave tiles
  double* tiles = (double*)(ktab + Kp);                // per wave: Ew[e][17], Xw[p][17]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  double* Ew = tiles + wave * (e + p) * 17;
  double* Xw = Ew + e * 17;
  // only the right-hand step permutes (its transposing store needs it); 16 lanes are served per
  // LDS cycle, so the MFMA operand reads of a plain row-major core are conflict free as they are
  auto swz = [](int k) { return LEFT ? 0 : (((k >> 1) & 15) | (SW ? ((k & 1) << 4) : 0)); };
  const bool dma = LEFT && n == npad && ((K * npad) & 127) == 0;
  if (dma) {
    // the core is the LDS image as it stands: copied by global_load_lds (1 KiB per wave instruction,
    // everything in flight at once, no registers), rows K..Kp zeroed beside it
    const int units = (K * npad) >> 1;
    for (int it = wave; it * 64 < units; it += 4)
      __builtin_amdgcn_global_load_lds(
          (const void __attribute__((address_space(1)))*)((const char*)core + (size_t)(it * 64 + lane) * 16),
          (void __attribute__((address_space(3)))*)((char*)Bs + it * 1024), 16, 0, 0);
    for (int idx = K * npad + tid; idx < Kp * npad; idx += 256) Bs[idx] = 0.0;
  } else if (LEFT) {   // core (e, p, n) = [K][n]: lanes along a row, eight rows in flight per thread
    constexpr int RPP = 256 / npad;
    const int jj = tid % npad, kk = tid / npad;
    const bool on = kk < RPP;
    for (int k0 = kk; k0 < Kp; k0 += RPP * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = core[min(k0 + u * RPP, K - 1) * n + min(jj, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * RPP;
        if (on && k < Kp) Bs[k * npad + (jj ^ swz(k))] = (jj < n && k < K) ? v[u] : 0.0;
      }
    }
  } else {      // core (n, p, e) = [n][K]: lanes along k, eight columns in flight per thread
    for (int k = tid; k < Kp; k += 256) {
      const int kc = min(k, K - 1), sz = swz(k);
#pragma unroll
      for (int j0 = 0; j0 < npad; j0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = core[min(j0 + u, n - 1) * K + kc];
#pragma unroll
        for (int u = 0; u < 8; ++u) Bs[k * npad + ((j0 + u) ^ sz)] = (j0 + u < n && k < K) ? v[u] : 0.0;
      }
    }
  }
  for (int k = tid; k < Kp; k += 256) {
    int ei, pi;
    if (LEFT) {
      ei = k / p;
      pi = k - ei * p;
    } else {
      pi = k / e;
      ei = k - pi * e;
    }
    ktab[k] = k < K ? (ei * 17) | ((pi * 17) << 16) : 0;   // padding: any finite operand, its core row is zero
  }
  // Prefetch slots.  E slots (q < QE): the 16 x e block of a tile is contiguous, lane + 64 q walks
  // it and (row, column) of each slot follows by carrying, with no division per slot.  X slots:
  // lane = (row in tile, feature mod 4); the sample of a row is row / C.
  const int QE = (e + 3) >> 2, QX = (p + 3) >> 2;
  const int ntiles = (Mi + 15) >> 4;
  const int stride = (int)gridDim.x * 4;
  int tile = (int)blockIdx.x * 4 + wave;
  int e_dst[ENVF_SLOTS];
  {
    int r = lane / e, l = lane - r * e;
    const int dr = 64 / e, dl = 64 - dr * e;
#pragma unroll
    for (int q = 0; q < ENVF_SLOTS; ++q) {
      e_dst[q] = (lane + 64 * q < 16 * e) ? l * 17 + r : -1;
      l += dl;
      r += dr;
      if (l >= e) {
        l -= e;
        ++r;
      }
    }
  }
  double pe[ENVF_SLOTS];
  auto fetch = [&](int t) {
    const int row0 = t << 4;
    const int xr = min(row0 + i, Mi - 1);
    const double* xsrc = X + (C == 1 ? xr : xr / C) * p;
    const double* esrc = E + row0 * e + lane;
    const int elim = (Mi - row0) * e - lane - 1;   // last valid offset from esrc
#pragma unroll
    for (int q = 0; q < ENVF_SLOTS; ++q) {
      if (q < QE)
        pe[q] = esrc[min(64 * q, elim)];
      else if (q < QE + QX)
        pe[q] = xsrc[min(kq + 4 * (q - QE), p - 1)];
    }
  };
  if (tile < ntiles) fetch(tile);
  if (dma) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of the core has landed
  __syncthreads();
  const int ib = i ^ (kq >> 1);                          // low swizzle bits that depend on the lane only
  const int par = (SW && !LEFT) ? (kq & 1) : 0;
  // Results leave one tile late (DEFER): a wave's stores and its prefetch loads retire in issue order,
  // so stores issued right before the next tile's loads would be waited for together with them;
  // issued a whole tile earlier they have long completed.  Costs NT*8 registers, so only for NT <= 2.
  constexpr bool DEFER = NT <= 2;
  d4 hold[NT];
  int hold_row0 = -1;
  auto flush = [&](const d4 (&v)[NT], int row0) {
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = row0 + kq + 4 * t;
          const int col = j * 16 + i;
          if (row < Mi && col < n) {
            double* o = out + (int64_t)row * n + col;
            *o = accumulate ? *o + v[j][t] : v[j][t];
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = row0 + kq + 4 * t;
        if (i == 0 && row < Mi) out[row] = accumulate ? out[row] + v[0][t] : v[0][t];
      }
    }
  };
  for (; tile < ntiles; tile += stride) {
    const int row0 = tile << 4;
#pragma unroll
    for (int q = 0; q < ENVF_SLOTS; ++q) {
      if (q < QE) {
        if (e_dst[q] >= 0) Ew[e_dst[q]] = pe[q];
      } else if (q < QE + QX) {
        const int pp = kq + 4 * (q - QE);
        if (pp < p) Xw[pp * 17 + i] = pe[q];
      }
    }
    if (DEFER && hold_row0 >= 0) flush(hold, hold_row0);
    if (tile + stride < ntiles) fetch(tile + stride);
    // mode 1: the factors of the row dot products, fetched now and used after the k loop
    double rv[DEFER ? 4 : 1][NT], hv[DEFER ? 4 : 1];
    if (DEFER && mode != 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rc = min(row0 + kq + 4 * t, Mi - 1);
        const double* rp = Rdot + (int64_t)(C == 1 ? rc : rc / C) * n;
#pragma unroll
        for (int j = 0; j < NT; ++j) rv[t][j] = rp[min(j * 16 + i, n - 1)];
        hv[t] = Hmul ? Hmul[rc] : 1.0;
      }
    }
    // Sixteen k per group (four MFMA steps per column tile).  The loop is software pipelined by
    // hand: the offsets of group g+2 and the operands of group g+1 are in flight while group g is
    // multiplied.  NT is a template parameter: a run-time test around an MFMA would make the
    // compiler wait for every product.
    d4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
    const int ng = Kp >> 4;
    int o0[4], o1[4];
    double ea[4], xa[4], ba[4][NT], eb[4], xb[4], bb[4][NT];
    auto ld_off = [&](int g, int (&o)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = ktab[min(g * 16 + u * 4 + kq, Kp - 1)];
    };
    // odd: parity of the group, known at compile time in the unrolled loop below; with it the
    // swizzle of row k = 16 g + 4 u + kq is a constant XOR on top of ib
    auto ld_dat = [&](int g, int odd, const int (&o)[4], double (&ev)[4], double (&xv)[4], double (&bv)[4][NT]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ev[u] = Ew[(o[u] & 0xffff) + i];
        xv[u] = Xw[(o[u] >> 16) + i];
        const double* brow = Bs + min(g * 16 + u * 4 + kq, Kp - 1) * npad + (LEFT ? i : (ib ^ (2 * u + 8 * odd)));
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[u][j] = brow[(j ^ par) << 4];
      }
    };
    auto mul = [&](const double (&ev)[4], const double (&xv)[4], const double (&bv)[4][NT]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double av = ev[u] * xv[u];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = MFMA_F64(av, bv[u][j], acc[j]);
      }
    };
    ld_off(0, o0);
    ld_off(1, o1);
    ld_dat(0, 0, o0, ea, xa, ba);
    for (int g = 0; g < ng; g += 2) {
      ld_dat(g + 1, 1, o1, eb, xb, bb);
      ld_off(g + 2, o0);
      mul(ea, xa, ba);
      if (g + 1 < ng) {
        ld_dat(g + 2, 0, o0, ea, xa, ba);
        ld_off(g + 3, o1);
        mul(eb, xb, bb);
      }
    }
    if (mode != 0) {   // row dot products; the result of row kq + 4 t lands in acc[0][t] of the lanes i == 0
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        double u = 0.0, h;
        if (DEFER) {
#pragma unroll
          for (int j = 0; j < NT; ++j) u += (j * 16 + i < n) ? acc[j][t] * rv[t][j] : 0.0;
          h = hv[t];
        } else {   // wide tiles have no registers to spare for the prefetch
          const int rc = min(row0 + kq + 4 * t, Mi - 1);
          const double* rp = Rdot + (int64_t)(C == 1 ? rc : rc / C) * n;
#pragma unroll
          for (int j = 0; j < NT; ++j) u += (j * 16 + i < n) ? acc[j][t] * rp[min(j * 16 + i, n - 1)] : 0.0;
          h = Hmul ? Hmul[rc] : 1.0;
        }
        u += __shfl_xor(u, 8, 16);
        u += __shfl_xor(u, 4, 16);
        u += __shfl_xor(u, 2, 16);
        u += __shfl_xor(u, 1, 16);
        acc[0][t] = u * h;   // h: C == 1, the Hessian is one number per row (else 1)
      }
    }
    if (DEFER) {
#pragma unroll
      for (int j = 0; j < NT; ++j) hold[j] = acc[j];
      hold_row0 = row0;
    } else {
      flush(acc, row0);
    }
  }
  if (DEFER && hold_row0 >= 0) flush(hold, hold_row0);
}

// The same step for bond dimensions e = 4 * E4 (16 or 32) and at most 8 features, without any
// per-tile staging: lane (i, kq) of a wave owns row i of its 16-row tile and keeps the entries
// E[row, 4 t + kq] (t < E4) and X[row, :] in registers, fetched straight from memory (the next
// tile's while the current one is multiplied).  With the contraction ordered as (feature, t) the
// MFMA operand of a step is er[t] * xr[feature]: one multiply, no LDS read, no offset table; only
// the core rows come from LDS.  Core staging, epilogues and modes as in env_fast_kernel.
template <bool LEFT, int NT, int E4>
__global__ __launch_bounds__(256, 2) void env_reg_kernel(const double* __restrict__ E, const double* __restrict__ X,
                                                          const double* __restrict__ core, double* __restrict__ out,
                                                          int Mi, int C, int p, int n, int accumulate, int mode,
                                                          const double* __restrict__ Rdot,
                                                          const double* __restrict__ Hmul) {
  extern __shared__ double smem[];
  constexpr int npad = NT * 16;
  constexpr int SW = (NT % 2 == 0) ? 1 : 0;
  constexpr int e = 4 * E4;
  const int K = e * p;                                 // a multiple of 4: no padded rows
  double* Bs = smem;                                   // [K][npad]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  auto swz = [](int k) { return LEFT ? 0 : (((k >> 1) & 15) | (SW ? ((k & 1) << 4) : 0)); };
  const bool dma = LEFT && n == npad && ((K * npad) & 127) == 0;
  if (dma) {
    const int units = (K * npad) >> 1;
    for (int it = wave; it * 64 < units; it += 4)
      __builtin_amdgcn_global_load_lds(
          (const void __attribute__((address_space(1)))*)((const char*)core + (size_t)(it * 64 + lane) * 16),
          (void __attribute__((address_space(3)))*)((char*)Bs + it * 1024), 16, 0, 0);
  } else if (LEFT) {   // core (e, p, n) = [K][n]
    constexpr int RPP = 256 / npad;
    const int jj = tid % npad, kk = tid / npad;
    const bool on = kk < RPP;
    for (int k0 = kk; k0 < K; k0 += RPP * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = core[min(k0 + u * RPP, K - 1) * n + min(jj, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * RPP;
        if (on && k < K) Bs[k * npad + jj] = jj < n ? v[u] : 0.0;
      }
    }
  } else {             // core (n, p, e) = [n][K], stored transposed and column-permuted
    for (int k = tid; k < K; k += 256) {
      const int sz = swz(k);
#pragma unroll
      for (int j0 = 0; j0 < npad; j0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = core[min(j0 + u, n - 1) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) Bs[k * npad + ((j0 + u) ^ sz)] = (j0 + u < n) ? v[u] : 0.0;
      }
    }
  }
  const int ntiles = (Mi + 15) >> 4;
  const int stride = (int)gridDim.x * 4;
  int tile = (int)blockIdx.x * 4 + wave;
  double er[E4], xr[8], ern[E4], xrn[8];
  // unconditional loads with clamped indices (features past p repeat the last one and are never used)
  auto fetch = [&](int t, double (&ev)[E4], double (&xv)[8]) {
    const int row = min((t << 4) + i, Mi - 1);
    const double* es = E + row * e + kq;
    const double* xs = X + (C == 1 ? row : row / C) * p;
#pragma unroll
    for (int u = 0; u < E4; ++u) ev[u] = es[4 * u];
#pragma unroll
    for (int q = 0; q < 8; ++q) xv[q] = xs[min(q, p - 1)];
  };
  if (tile < ntiles) fetch(tile, ern, xrn);
  if (dma) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of the core has landed
  __syncthreads();
  // lane part of the core row address of a step: LEFT row (4 t + kq) p + feature, else feature e + 4 t + kq
  const int lane_row = LEFT ? kq * p : kq;
  constexpr bool DEFER = NT <= 2;
  d4 hold[NT];
  int hold_row0 = -1;
  auto flush = [&](const d4 (&v)[NT], int row0) {
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = row0 + kq + 4 * t;
          const int col = j * 16 + i;
          if (row < Mi && col < n) {
            double* o = out + (int64_t)row * n + col;
            *o = accumulate ? *o + v[j][t] : v[j][t];
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = row0 + kq + 4 * t;
        if (i == 0 && row < Mi) out[row] = accumulate ? out[row] + v[0][t] : v[0][t];
      }
    }
  };
  for (; tile < ntiles; tile += stride) {
    const int row0 = tile << 4;
#pragma unroll
    for (int u = 0; u < E4; ++u) er[u] = ern[u];
#pragma unroll
    for (int q = 0; q < 8; ++q) xr[q] = xrn[q];
    if (DEFER && hold_row0 >= 0) flush(hold, hold_row0);
    if (tile + stride < ntiles) fetch(tile + stride, ern, xrn);
    double rv[DEFER ? 4 : 1][NT], hv[DEFER ? 4 : 1];
    if (DEFER && mode != 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rc = min(row0 + kq + 4 * t, Mi - 1);
        const double* rp = Rdot + (int64_t)(C == 1 ? rc : rc / C) * n;
#pragma unroll
        for (int j = 0; j < NT; ++j) rv[t][j] = rp[min(j * 16 + i, n - 1)];
        hv[t] = Hmul ? Hmul[rc] : 1.0;
      }
    }
    d4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < p) {   // uniform; E4 steps of straight-line code per feature
#pragma unroll
        for (int u = 0; u < E4; ++u) {
          const int krow = (LEFT ? 4 * u * p + q : q * e + 4 * u) + lane_row;
          const double a = er[u] * xr[q];
          const double* brow = Bs + krow * npad + i;
          if (LEFT) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = MFMA_F64(a, brow[j * 16], acc[j]);
          } else {
            const int sz = swz(krow);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = MFMA_F64(a, Bs[krow * npad + ((j * 16 + i) ^ sz)], acc[j]);
          }
        }
      }
    }
    if (mode != 0) {   // row dot products; the result of row kq + 4 t lands in acc[0][t] of the lanes i == 0
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        double u = 0.0, h;
        if (DEFER) {
#pragma unroll
          for (int j = 0; j < NT; ++j) u += (j * 16 + i < n) ? acc[j][t] * rv[t][j] : 0.0;
          h = hv[t];
        } else {
          const int rc = min(row0 + kq + 4 * t, Mi - 1);
          const double* rp = Rdot + (int64_t)(C == 1 ? rc : rc / C) * n;
#pragma unroll
          for (int j = 0; j < NT; ++j) u += (j * 16 + i < n) ? acc[j][t] * rp[min(j * 16 + i, n - 1)] : 0.0;
          h = Hmul ? Hmul[rc] : 1.0;
        }
        u += __shfl_xor(u, 8, 16);
        u += __shfl_xor(u, 4, 16);
        u += __shfl_xor(u, 2, 16);
        u += __shfl_xor(u, 1, 16);
        acc[0][t] = u * h;
      }
    }
    if (DEFER) {
#pragma unroll
      for (int j = 0; j < NT; ++j) hold[j] = acc[j];
      hold_row0 = row0;
    } else {
      flush(acc, row0);
    }
  }
  if (DEFER && hold_row0 >= 0) flush(hold, hold_row0);
}

static bool env_fast_ok(int e, int p, int n, size_t* lds) {
  const int K = e * p, npad = (n + 15) & ~15;
  const int Kp = (K + 15) & ~15;
  const size_t bytes = ((size_t)Kp * npad) * 8 + (size_t)Kp * 4 + (size_t)4 * (e + p) * 17 * 8;
  *lds = bytes;
  return npad <= 64 && (e + 3) / 4 + (p + 3) / 4 <= ENVF_SLOTS && bytes <= 80 * 1024 && e * 17 < 65536 && p * 17 < 32768;
}

static int g_env_reg = 1;   // 0: the register-operand variant is not used (experiments)

// Launches the resident-core kernel when the shape allows it; returns 1 when it does not (nothing launched).
static int env_fast_launch(bool left, const double* E, const double* X, const double* core, double* out, int64_t M,
                           int C, int e, int p, int n, int accumulate, int mode, const double* Rdot,
                           const double* Hmul, hipStream_t st) {
  size_t flds;
  if (!(env_fast_ok(e, p, n, &flds) && M * (int64_t)(e > n ? e : n) < (1ll << 31) && p <= 64)) return 1;
  const int64_t ntiles = (M + 15) >> 4;
  const unsigned grid = (unsigned)((ntiles + 3) / 4 < 512 ? (ntiles + 3) / 4 : 512);
  const int nt = (n + 15) >> 4;
  if ((e == 16 || e == 32) && p <= 8 && g_env_reg && !(!left && nt == 4 && e == 32)) {   // (that one variant spills)
    const void* regv[16] = {
        (const void*)env_reg_kernel<true, 1, 4>,  (const void*)env_reg_kernel<true, 2, 4>,
        (const void*)env_reg_kernel<true, 3, 4>,  (const void*)env_reg_kernel<true, 4, 4>,
        (const void*)env_reg_kernel<true, 1, 8>,  (const void*)env_reg_kernel<true, 2, 8>,
        (const void*)env_reg_kernel<true, 3, 8>,  (const void*)env_reg_kernel<true, 4, 8>,
        (const void*)env_reg_kernel<false, 1, 4>, (const void*)env_reg_kernel<false, 2, 4>,
        (const void*)env_reg_kernel<false, 3, 4>, (const void*)env_reg_kernel<false, 4, 4>,
        (const void*)env_reg_kernel<false, 1, 8>, (const void*)env_reg_kernel<false, 2, 8>,
        (const void*)env_reg_kernel<false, 3, 8>, (const void*)env_reg_kernel<false, 4, 8>};
    static bool reg_lds_set[16][TTK_MAX_DEVICES];
    for (int v = 0; v < 16; ++v)
      if (!allow_big_lds(regv[v], reg_lds_set[v], 80 * 1024)) return TTK_ELAUNCH;
    const size_t rlds = (size_t)e * p * (nt * 16) * sizeof(double);   // <= 32 * 8 * 64 * 8 = 128 KiB ... checked below
    if (rlds <= 80 * 1024) {
      int Mi = (int)M;
      void* args[] = {(void*)&E, (void*)&X, (void*)&core, (void*)&out, (void*)&Mi, (void*)&C, (void*)&p, (void*)&n,
                      (void*)&accumulate, (void*)&mode, (void*)&Rdot, (void*)&Hmul};
      const void* fn = regv[(left ? 0 : 8) + (e == 32 ? 4 : 0) + nt - 1];
      if (hipLaunchKernel(fn, dim3(grid), dim3(256), args, rlds, st) != hipSuccess) return TTK_ELAUNCH;
      return ttk_launch_status();
    }
  }
  static bool big_lds_set[8][TTK_MAX_DEVICES];
  const void* variants[8] = {(const void*)env_fast_kernel<true, 1>,  (const void*)env_fast_kernel<true, 2>,
                             (const void*)env_fast_kernel<true, 3>,  (const void*)env_fast_kernel<true, 4>,
                             (const void*)env_fast_kernel<false, 1>, (const void*)env_fast_kernel<false, 2>,
                             (const void*)env_fast_kernel<false, 3>, (const void*)env_fast_kernel<false, 4>};
  for (int v = 0; v < 8; ++v)
    if (!allow_big_lds(variants[v], big_lds_set[v], 80 * 1024)) return TTK_ELAUNCH;
  int Mi = (int)M;
  void* args[] = {(void*)&E, (void*)&X, (void*)&core, (void*)&out, (void*)&Mi, (void*)&C, (void*)&e, (void*)&p,
                  (void*)&n, (void*)&accumulate, (void*)&mode, (void*)&Rdot, (void*)&Hmul};
  if (hipLaunchKernel(variants[(left ? 0 : 4) + nt - 1], dim3(grid), dim3(256), args, flds, st) != hipSuccess)
    return TTK_ELAUNCH;
  return ttk_launch_status();
}

static int env_step_launch(bool left, const double* E, const double* X, const double* core, double* out, int64_t M,
                           int C, int e, int p, int n, int accumulate, hipStream_t st) {
  const int frc = env_fast_launch(left, E, X, core, out, M, C, e, p, n, accumulate, 0, nullptr, nullptr, st);
  if (frc != 1) return frc;
  const int K = e * p;
  int KC = round_up(K, 4);
  int kc_max = (48 * 1024) / (64 * 8);  // 96 rows of 64 doubles = 48 KiB
  // wide bonds with many features: the per-wave operand rows take 512 (e + p) bytes, the core chunk what is left
  const int kc_fit = (int)(((int64_t)150 * 1024 / 8 - 64 * (int64_t)(e + p)) / 64) & ~3;
  if (kc_fit < kc_max) kc_max = kc_fit;
  if (kc_max < 4) return TTK_EINVAL;
  if (KC > kc_max) KC = kc_max;
  const size_t lds = ((size_t)KC * 64 + 4 * 16 * (size_t)(e + p)) * sizeof(double);
  if (lds > 150 * 1024) return TTK_EINVAL;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((n + 63) / 64));
  if (left)
    hipLaunchKernelGGL(env_step_kernel<true>, grid, dim3(256), lds, st, E, X, core, out, M, C, e, p, n, KC, accumulate);
  else
    hipLaunchKernelGGL(env_step_kernel<false>, grid, dim3(256), lds, st, E, X, core, out, M, C, e, p, n, KC, accumulate);
  return ttk_launch_status();
}

// ---------------------------------------------------------------------------------------
// y[s,c] = sum_r L[s,c,r] R[s,r]; 8 lanes per output row, shuffle reduction.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void predict_kernel(const double* __restrict__ L, const double* __restrict__ R,
                                                       double* __restrict__ y, int64_t M, int C, int r, int accumulate) {
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
  if (row < M && sub == 0) y[row] = accumulate ? y[row] + acc : acc;
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
      g[s * C + c] = w * (pc - (kind == 3 ? (c == label ? 1.0 : 0.0) : t[c]));
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

// J[s,c,(l,p,r)] (+)= L[s,c,l] X[s,p] R[s,r]: explicit Jacobian rows, only for local problems that are
// not a single Khatri-Rao term (conv-featurised columns with a conv bond).
__global__ __launch_bounds__(256) void kr_rows_kernel(const double* __restrict__ L, const double* __restrict__ X,
                                                       const double* __restrict__ R, double* __restrict__ J, int64_t S,
                                                       int C, int rl, int p, int rr, int accumulate) {
  const int64_t D = (int64_t)rl * p * rr;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * C * D) return;
  const int64_t sc = idx / D;
  const int k = (int)(idx - sc * D);
  const int r = k % rr, lp = k / rr, pp = lp % p, l = lp / p;
  const int64_t s_ = sc / C;
  const double v = L[sc * rl + l] * X[s_ * p + pp] * R[s_ * rr + r];
  J[idx] = accumulate ? J[idx] + v : v;
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
// (a,c) tile of two consecutive b (2 x 3x3 MFMA tiles, 18 independent accumulators: the
// P and RR operand products are shared by both b, which halves the fp64 multiplies, LDS
// reads and address arithmetic per MFMA); the 4 waves of a workgroup take 4 consecutive
// tiles and share the LDS-staged factor rows of a sample chunk.  blockIdx.x % nslabs selects the sample slab, so workgroups that share an XCD
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

#define GRAM_MAX_CLASSED 64   // pair classes are honoured up to this many (p,q) pairs (p <= 10)
struct GramDims {
  int MP, MPpad, NX, NR, NRpad, ntm, ntn, nbp, ntiles, ngroups, nslabs, CH, bufd;
  int64_t slab_len, gsize;
  // Feature-pair classes: pairs (p,q) whose products X_p X_q coincide for every sample (e.g.
  // powers of one variable: X_p X_q = x^(p+q)) share one slice of the Gram.  Only the nrep
  // representatives reps[] are computed; repof[b] names the representative of pair b.
  int nrep, classed;
  int reps[GRAM_MAX_CLASSED], repof[GRAM_MAX_CLASSED];
};

static int g_gram_slabs = 0;   // > 0: slab count forced (experiments)
static GramDims gram_dims(int64_t S, int C, int rl, int p, int rr, int sym, int nrep = 0) {
  GramDims d;
  d.classed = 0;
  d.MP = sym ? rl * (rl + 1) / 2 : rl * rl;
  d.MPpad = round_up(d.MP, 48);
  d.NX = p * (p + 1) / 2;
  d.NR = rr * (rr + 1) / 2;
  d.NRpad = round_up(d.NR, 48);
  d.ntm = d.MPpad / 48;
  d.ntn = d.NRpad / 48;
  d.nrep = nrep > 0 ? nrep : d.NX;
  d.nbp = (d.nrep + 1) / 2;            // a wave tile carries two (p,q) pairs
  d.ntiles = d.nbp * d.ntm * d.ntn;
  d.ngroups = (d.ntiles + 3) / 4;
  d.gsize = (int64_t)d.MPpad * d.NX * d.NRpad;
  // sample slabs: enough workgroups to fill 256 CUs x 2 several times over, a multiple of 8 for
  // large S (one slab per XCD under round-robin dispatch), chosen to keep the last wave of
  // workgroups reasonably full
  int64_t cap = S / 512;
  if (cap < 1) cap = 1;
  int best = 1;
  if (S >= 8 * 2048) {
    double best_cost = 1e30;
    for (int ns = 8; ns <= 32 && ns <= cap; ns += 8) {
      // workgroups are dispatched as slots free up, so the tail costs about half a round whatever the
      // count; more, smaller slabs shorten it (measured: 16 slabs 4 % faster than 8 at 182 groups)
      const double rounds = (double)d.ngroups * ns / 512.0;
      const double cost = (rounds + 0.5) / rounds + 0.004 * ns;   // tail waste + price of one more partial copy
      if (cost < best_cost) {
        best_cost = cost;
        best = ns;
      }
    }
  } else {
    int want = (1024 + d.ngroups - 1) / d.ngroups;
    if (want > cap) want = (int)cap;
    if (want > 64) want = 64;
    best = want < 1 ? 1 : want;
  }
  d.nslabs = g_gram_slabs > 0 ? g_gram_slabs : best;
  d.slab_len = round_up64((S + d.nslabs - 1) / d.nslabs, 4);
  // chunk of samples in LDS: W doubles per sample, two buffers of at most 32 KiB each
  const int width = 2 * C * rl + d.NX + rr + 1;
  int ch = (32 * 1024) / (width * 8);
  ch = ch / 4 * 4;
  if (ch > 64) ch = 64;
  d.bufd = ((ch * width * 8 + 1023) / 1024) * 128;
  d.CH = ch;
  return d;
}

// Packed per-sample factor table
//   F[s] = [ L[s,:,:] | HL[s,:,:] | XX[s,:] | R[s,:] | 0 ]      (W = 2*C*rl + p(p+1)/2 + rr + 1 doubles)
// with HL[s,d,l] = sum_c H[s,c,d] L[s,c,l] and XX[s,(p<=q)] = X[s,p] X[s,q].  One contiguous row
// range of F is the LDS image of a sample chunk, so the Gram kernel stages a chunk with one
// batch of unconditional loads; the trailing zero is where lanes of padded tile rows/columns
// point, so that padding needs no mask multiply (fp64 VALU multiplies and fp64 MFMA share one
// datapath on gfx950: every v_mul_f64 in the inner loop costs MFMA issue time).
__global__ __launch_bounds__(256) void gram_pack_kernel(const double* __restrict__ L, const double* __restrict__ H,
                                                         const double* __restrict__ X, const double* __restrict__ R,
                                                         int64_t S, int C, int rl, int p, int rr,
                                                         double* __restrict__ F) {
  const int CW = C * rl, NX = p * (p + 1) / 2, W = 2 * CW + NX + rr + 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = ((S + 3) & ~(int64_t)3) * W;
  if (idx >= total) return;
  // (a 32-bit division where the table has fewer than 2^31 entries: the 64-bit one is several times the work)
  const int64_t s_ = total < ((int64_t)1 << 31) ? (int64_t)((unsigned)idx / (unsigned)W) : idx / W;
  const int j = (int)(idx - s_ * W);
  double v = 0.0;
  if (s_ >= S) {
    // rows up to the next multiple of four exist and are zero: the Gram kernel multiplies samples four at a time
  } else if (j < CW) {
    v = L[s_ * CW + j];
  } else if (j < 2 * CW) {
    const int d = (j - CW) / rl, l = (j - CW) - d * rl;
    for (int c = 0; c < C; ++c) v += H[(s_ * C + c) * C + d] * L[(s_ * C + c) * rl + l];
  } else if (j < 2 * CW + NX) {
    int pp, qq;
    tri_decode(j - 2 * CW, p, pp, qq);
    v = X[s_ * p + pp] * X[s_ * p + qq];
  } else if (j < 2 * CW + NX + rr) {
    v = R[s_ * rr + (j - 2 * CW - NX)];
  }
  F[idx] = v;
}

// One chunk = consecutive rows of F, copied as they are into one of two LDS buffers by
// global_load_lds (each lane moves 16 bytes, a wave instruction 1 KiB to a wave-uniform LDS base):
// no staging registers, no LDS store phase, and the copy of the next chunk runs under the
// products of the current one.  Lanes past the end of the chunk re-read its last 16 bytes into
// the slack behind the image (buffers are whole KiB).
__device__ __forceinline__ void gram_chunk_copy(const double* __restrict__ g, int doubles, double* lds, int wave,
                                                int lane) {
  const int units = doubles >> 1;   // 16-byte units; doubles is even (rows are copied four at a time)
  const char* src = (const char*)g;
  for (int it = wave; it * 64 < units; it += 4) {
    const int u = min(it * 64 + lane, units - 1);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)u * 16),
                                     (void __attribute__((address_space(3)))*)((char*)lds + it * 1024), 16, 0, 0);
  }
}

// The products of one staged chunk for one wave tile: per group of 4 samples the operands
// P[a] = H_l L_m (x XX_b) and Q[c] = R_r R_t are formed from the chunk image and multiplied
// on the matrix cores.  TWO = the tile carries two (p,q) pairs that share P and Q.
// fp64 multiplies and fp64 MFMA share one datapath; the arrangement below - the operands of one
// group of 4 samples formed in one block, then its 18 MFMAs - measured fastest: multiplies of the
// next group placed between the MFMAs (one at a time, or as one block after the first six) cost
// 3-5 %, two groups per block (NS = 2) 7 %.
template <bool C1, bool TWO, int NS>
__device__ __forceinline__ void gram_step_block(const double* __restrict__ row, int W, int C, int rl, int oX0, int oX1,
                                                const int (&oH)[3], const int (&oL)[3], const int (&oR1)[3],
                                                const int (&oR2)[3], d4 (&acc)[2][3][3]) {
  double aop0[NS][3], aop1[NS][3], bop[NS][3];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const double* r = row + n * 4 * W;
    const double xx0 = r[oX0], xx1 = TWO ? r[oX1] : 0.0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      double pa;
      if (C1) {
        pa = r[oH[u]] * r[oL[u]];
      } else {
        pa = 0.0;
        if (oL[u] != W - 1)
          for (int d = 0; d < C; ++d) pa += r[oH[u] + d * rl] * r[oL[u] + d * rl];
      }
      aop0[n][u] = pa * xx0;
      if (TWO) aop1[n][u] = pa * xx1;
      bop[n][u] = r[oR1[u]] * r[oR2[u]];
    }
  }
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        acc[0][u][v] = MFMA_F64(aop0[n][u], bop[n][v], acc[0][u][v]);
        if (TWO) acc[1][u][v] = MFMA_F64(aop1[n][u], bop[n][v], acc[1][u][v]);
      }
}

template <bool C1, bool TWO>
__device__ __forceinline__ void gram_steps(const double* __restrict__ smem, int steps, int kq, int W, int C, int rl,
                                           int oX0, int oX1, const int (&oH)[3], const int (&oL)[3],
                                           const int (&oR1)[3], const int (&oR2)[3], d4 (&acc)[2][3][3]) {
  const double* row = smem + kq * W;
  for (int ks = 0; ks < steps; ++ks, row += 4 * W)
    gram_step_block<C1, TWO, 1>(row, W, C, rl, oX0, oX1, oH, oL, oR1, oR2, acc);
}

template <bool C1>
__global__ __launch_bounds__(256, 2) void gram_kernel(const double* __restrict__ F, int64_t S, int C, int rl, int p,
                                                       int rr, int sym, GramDims gd, double* __restrict__ partial,
                                                       const unsigned* __restrict__ cancel, unsigned generation) {
  extern __shared__ double smem[];
  // a caller that has no use for this Gram any more writes its generation into the cancel word: workgroups that
  // start after that leave at once (the load is issued here and looked at after the index set-up below)
  const unsigned cancelled = cancel ? __hip_atomic_load(cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const int CW = C * rl;
  const int CH = gd.CH;
  const int W = 2 * CW + gd.NX + rr + 1;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int slab = blockIdx.x % gd.nslabs;
  const int group = blockIdx.x / gd.nslabs;
  const int tile = group * 4 + wave;
  const bool active = tile < gd.ntiles;
  int bp = 0, tm = 0, tn = 0;
  if (active) {
    bp = tile / (gd.ntm * gd.ntn);
    const int rem = tile - bp * (gd.ntm * gd.ntn);
    tm = rem / gd.ntn;
    tn = rem - tm * gd.ntn;
  }
  const bool has1 = 2 * bp + 1 < gd.nrep;
  const int b0 = gd.classed ? gd.reps[2 * bp] : 2 * bp;
  const int b1 = has1 ? (gd.classed ? gd.reps[2 * bp + 1] : 2 * bp + 1) : 0;
  // element offsets inside a sample row of the chunk image; W - 1 is the zero column
  const int oX0 = 2 * CW + b0;
  const int oX1 = has1 ? 2 * CW + b1 : W - 1;
  int oH[3], oL[3], oR1[3], oR2[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int a = tm * 48 + u * 16 + i;
    int l_ = 0, m_ = 0;
    if (a < gd.MP) {
      if (sym) {
        tri_decode(a, rl, l_, m_);
      } else {
        l_ = a / rl;
        m_ = a - l_ * rl;
      }
      oH[u] = CW + l_;
      oL[u] = m_;
    } else {
      oH[u] = oL[u] = W - 1;
    }
    const int c = tn * 48 + u * 16 + i;
    if (c < gd.NR) {
      int r_, t_;
      tri_decode(c, rr, r_, t_);
      oR1[u] = 2 * CW + gd.NX + r_;
      oR2[u] = 2 * CW + gd.NX + t_;
    } else {
      oR1[u] = oR2[u] = W - 1;
    }
  }
  d4 acc[2][3][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) acc[h][u][v] = d4{0.0, 0.0, 0.0, 0.0};

  const int64_t s_begin = (int64_t)slab * gd.slab_len;
  int64_t s_end = s_begin + gd.slab_len;
  if (s_end > S) s_end = S;
  const int bufd = gd.bufd;   // doubles per LDS buffer
  int cur = 0;
  if (cancel && cancelled == generation) return;   // uniform over the workgroup
  if (s_begin < s_end)
    gram_chunk_copy(F + s_begin * W, (((int)min((int64_t)CH, s_end - s_begin) + 3) & ~3) * W, smem, wave, lane);
  for (int64_t c0 = s_begin; c0 < s_end; c0 += CH) {
    const int64_t nvalid = min((int64_t)CH, s_end - c0);
    // this wave's part of the chunk has landed; after the barrier everybody's has, and every
    // wave is done with the other buffer
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    if (c0 + CH < s_end)
      gram_chunk_copy(F + (c0 + CH) * W, (((int)min((int64_t)CH, s_end - c0 - CH) + 3) & ~3) * W,
                      smem + (cur ^ 1) * bufd, wave, lane);
    if (active) {
      const double* img = smem + cur * bufd;
      const int steps = (int)((nvalid + 3) >> 2);
      if (has1)
        gram_steps<C1, true>(img, steps, kq, W, C, rl, oX0, oX1, oH, oL, oR1, oR2, acc);
      else   // the last, unpaired (p,q) of an odd count: half the products
        gram_steps<C1, false>(img, steps, kq, W, C, rl, oX0, oX1, oH, oL, oR1, oR2, acc);
    }
    cur ^= 1;
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
          dst[((int64_t)a * gd.NX + b0) * gd.NRpad + c] = acc[0][u][v][t];
          if (has1) dst[((int64_t)a * gd.NX + b1) * gd.NRpad + c] = acc[1][u][v][t];
        }
  }
}

// out[i] = sum_k partial[k*stride + i], fixed order
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const double* __restrict__ partial, int nslabs,
                                                            int64_t stride, int64_t n, double* __restrict__ out,
                                                            int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  double acc = accumulate ? out[idx] : 0.0;
  for (int k = 0; k < nslabs; ++k) acc += partial[(int64_t)k * stride + idx];
  out[idx] = acc;
}

// The same sum for many slabs and few outputs (the right-hand side: hundreds of slabs of a few
// thousand numbers): 16 outputs x 16 slab groups per workgroup, every thread's loads
// independent and in flight together, the 16 group sums added in fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const double* __restrict__ partial, int nslabs,
                                                                 int64_t stride, int64_t n, double* __restrict__ out,
                                                                 int accumulate) {
  __shared__ double part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 16 + o;
  const int64_t src = idx < n ? idx : n - 1;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = g;
  for (; k + 48 < nslabs; k += 64) {
    a0 += partial[(int64_t)k * stride + src];
    a1 += partial[(int64_t)(k + 16) * stride + src];
    a2 += partial[(int64_t)(k + 32) * stride + src];
    a3 += partial[(int64_t)(k + 48) * stride + src];
  }
  for (; k < nslabs; k += 16) a0 += partial[(int64_t)k * stride + src];
  part[g][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && idx < n) {
    double acc = accumulate ? out[idx] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += part[q][o];
    out[idx] = acc;
  }
}

// Slab reduction of a classed Gram: one thread per element of a representative slice adds the
// slabs once and stores the sum into the slice of every member of the class (with one slab and
// partial == out this only replicates, in place).
__global__ __launch_bounds__(256) void gram_reduce_classed_kernel(const double* __restrict__ partial, int nslabs,
                                                                   int64_t stride, GramDims gd, double* __restrict__ out) {
  // two adjacent columns per thread (rows are a multiple of 48 columns long and 16-byte aligned): 16-byte
  // loads, four slabs in flight per pass of the loop; the slabs are added in their order for either column
  const int half = gd.NRpad >> 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)gd.MPpad * gd.nrep * half) return;
  const int c = 2 * (int)(idx % half);
  const int64_t ak = idx / half;
  const int k = (int)(ak % gd.nrep);
  const int64_t a = ak / gd.nrep;
  const int b0 = gd.reps[k];
  const int64_t src = (a * gd.NX + b0) * gd.NRpad + c;
  double acc0 = 0.0, acc1 = 0.0;
  int s_ = 0;
  for (; s_ + 4 <= nslabs; s_ += 4) {
    double2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *(const double2*)(partial + (int64_t)(s_ + q) * stride + src);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc0 += v[q].x;
      acc1 += v[q].y;
    }
  }
  for (; s_ < nslabs; ++s_) {
    const double2 v = *(const double2*)(partial + (int64_t)s_ * stride + src);
    acc0 += v.x;
    acc1 += v.y;
  }
  for (int b = b0; b < gd.NX; ++b)   // members come after their representative
    if (gd.repof[b] == b0) *(double2*)(out + (a * gd.NX + b) * gd.NRpad + c) = double2{acc0, acc1};
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
// A workgroup owns a slab of samples and streams it through LDS in chunks (the next chunk's
// loads are in flight while the current one is multiplied); each wave owns a block of 2 x 3
// output tiles, so that per four samples 2 + 3*2 operand reads and 3 multiplies feed 6 MFMAs.
// With one output (C == 1) the gradient scales the feature factor while it is staged,
// otherwise GL is formed by a small kernel first.  Every slab writes its own partial result;
// a second kernel adds the slabs in fixed order.
// LDS row of a sample: [ A (rl) | X (p) | R (rr) | .. | 0 ], row stride W = 16 mod 32 doubles so that
// the four sample rows of one MFMA operand fall on different banks.
// ---------------------------------------------------------------------------------------
#define RHS_QA 8   // prefetch registers per thread for the A, X and R parts of a chunk
#define RHS_QX 2
#define RHS_QR 8
struct RhsDims {
  int mt, ntl, nv, bn, nblocks, passes, nslabs, CH, W;
  int64_t slab_len;
};

static int g_rhs_max_slabs = 512;

static RhsDims rhs_dims(int64_t S, int C, int rl, int p, int rr) {
  RhsDims d;
  d.mt = (rl + 15) / 16;
  d.ntl = (p * rr + 15) / 16;
  // a wave owns 2 x 3 output tiles; with a single row tile (left bond <= 16) 1 x 3, 1 x 6, 1 x 8 or 1 x 12: the
  // narrowest that keeps the four waves to one pass, so that no empty tile is multiplied
  d.nv = 3;
  if (d.mt == 1) {
    // passes x (tiles a wave multiplies per pass + what staging a pass costs, in the same unit)
    const int cand[4] = {3, 6, 8, 12};
    int best = 1 << 30;
    for (int k = 0; k < 4; ++k) {
      const int bn = (d.ntl + cand[k] - 1) / cand[k];
      const int cost = ((bn + 3) / 4) * (cand[k] + 4);
      if (cost < best) {
        best = cost;
        d.nv = cand[k];
      }
    }
  }
  d.bn = (d.ntl + d.nv - 1) / d.nv;
  d.nblocks = ((d.mt + 1) / 2) * d.bn;
  d.passes = (d.nblocks + 3) / 4;
  const int wd = rl + p + rr + 1;
  d.W = wd <= 16 ? 16 : ((wd - 16 + 31) / 32) * 32 + 16;
  int ch = (40 * 1024) / (d.W * 8);
  if (ch > 64) ch = 64;
  const int big = rl > rr ? rl : rr;
  if (ch * big > 256 * RHS_QA) ch = 256 * RHS_QA / big;
  if (ch * p > 256 * RHS_QX) ch = 256 * RHS_QX / p;
  d.CH = ch / 4 * 4;
  int64_t ns = (S + 63) / 64;
  if (ns > g_rhs_max_slabs) ns = g_rhs_max_slabs;
  if (ns < 1) ns = 1;
  d.nslabs = (int)ns;
  d.slab_len = round_up64((S + d.nslabs - 1) / d.nslabs, 4);
  return d;
}

// GL[s,l] = sum_c g[s,c] L[s,c,l]
__global__ __launch_bounds__(256) void rhs_gl_kernel(const double* __restrict__ g, const double* __restrict__ L,
                                                      double* __restrict__ GL, int64_t S, int C, int rl) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * rl) return;
  const int64_t s = idx / rl;
  const int l = (int)(idx - s * rl);
  double v = 0.0;
  for (int c = 0; c < C; ++c) v += g[s * C + c] * L[(s * C + c) * rl + l];
  GL[idx] = v;
}

// LDS destinations of the prefetch slots of one part of a chunk: the part is `width` columns of
// CH consecutive samples, contiguous in memory; slot q of a lane holds element tid + 256 q.
template <int NQ>
__device__ __forceinline__ void rhs_slot_dst(int tid, int width, int W, int col0, int total, int (&dst)[NQ], int (&srow)[NQ]) {
  int r = tid / width, c = tid - r * width;
  const int dr = 256 / width, dc = 256 - dr * width;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    dst[q] = (tid + 256 * q < total) ? r * W + col0 + c : -1;
    srow[q] = r;
    c += dc;
    r += dr;
    if (c >= width) {
      c -= width;
      ++r;
    }
  }
}

template <int MU, int NV>   // output tiles of a wave: MU row tiles x NV column tiles
__global__ __launch_bounds__(256, 2) void rhs_kernel(const double* __restrict__ A, const double* __restrict__ X,
                                                      const double* __restrict__ R, const double* __restrict__ gx,
                                                      int64_t S, int rl, int p, int rr, RhsDims rd,
                                                      double* __restrict__ partial) {
  extern __shared__ double smem[];
  const int CH = rd.CH, W = rd.W;
  const int N = p * rr;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int slab = blockIdx.x;
  const int64_t s_begin = (int64_t)slab * rd.slab_len;
  int64_t s_end = s_begin + rd.slab_len;
  if (s_end > S) s_end = S;
  double* dst = partial + (int64_t)slab * rl * N;
  for (int k = tid; k < CH; k += 256) smem[k * W + W - 1] = 0.0;   // the zero column padded lanes read
  int dA[RHS_QA], dX[RHS_QX], dR[RHS_QR], rA[RHS_QA], rX[RHS_QX], rR[RHS_QR];
  rhs_slot_dst<RHS_QA>(tid, rl, W, 0, CH * rl, dA, rA);
  rhs_slot_dst<RHS_QX>(tid, p, W, rl, CH * p, dX, rX);
  rhs_slot_dst<RHS_QR>(tid, rr, W, rl + p, CH * rr, dR, rR);
  double pa[RHS_QA], px[RHS_QX], pg[RHS_QX], pr[RHS_QR];
  // One batch of loads with clamped indices, every slot unconditionally (unused slots re-read the
  // last element) and nothing that consumes them: a load inside a branch, or a use next to its
  // load, is waited for on the spot, one load at a time.  Zeroing of samples past the slab end
  // and the gradient factor are applied when the registers go to LDS.
  auto fetch = [&](int64_t c0) {
    const int nvalid = (int)min((int64_t)CH, s_end - c0);
    const double* a0 = A + c0 * rl + tid;
    const double* x0 = X + c0 * p + tid;
    const double* r0 = R + c0 * rr + tid;
    const double* gsrc = gx ? gx + c0 : X + c0 * p;   // without a gradient factor: any readable address, value unused
    const int la = nvalid * rl - 1 - tid, lx = nvalid * p - 1 - tid, lr = nvalid * rr - 1 - tid;
#pragma unroll
    for (int q = 0; q < RHS_QA; ++q) pa[q] = a0[max(min(256 * q, la), -tid)];
#pragma unroll
    for (int q = 0; q < RHS_QX; ++q) {
      px[q] = x0[max(min(256 * q, lx), -tid)];
      pg[q] = gsrc[min(rX[q], nvalid - 1)];
    }
#pragma unroll
    for (int q = 0; q < RHS_QR; ++q) pr[q] = r0[max(min(256 * q, lr), -tid)];
  };
  const int bm = (rd.mt + 1) >> 1;
  for (int pass = 0; pass < rd.passes; ++pass) {
    const int blk = pass * 4 + wave;
    const bool active = blk < rd.nblocks;
    const int tm = active ? blk / rd.bn : 0, tn = active ? blk - tm * rd.bn : 0;
    (void)bm;
    int oA[MU], oX[NV], oR[NV];
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      const int l = (tm * MU + u) * 16 + i;
      oA[u] = l < rl ? l : W - 1;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int n = (tn * NV + v) * 16 + i;
      if (n < N) {
        const int pn = n / rr;
        oX[v] = rl + pn;
        oR[v] = rl + p + (n - pn * rr);
      } else {
        oX[v] = oR[v] = W - 1;
      }
    }
    d4 acc[MU][NV];
#pragma unroll
    for (int u = 0; u < MU; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
    if (s_begin < s_end) fetch(s_begin);
    for (int64_t c0 = s_begin; c0 < s_end; c0 += CH) {
      const int nvalid = (int)min((int64_t)CH, s_end - c0);
      __syncthreads();   // every wave is done with the previous chunk
      {
        const int la = nvalid * rl - 1 - tid, lx = nvalid * p - 1 - tid, lr = nvalid * rr - 1 - tid;
#pragma unroll
        for (int q = 0; q < RHS_QA; ++q)
          if (dA[q] >= 0) smem[dA[q]] = 256 * q <= la ? pa[q] : 0.0;
#pragma unroll
        for (int q = 0; q < RHS_QX; ++q)
          if (dX[q] >= 0) smem[dX[q]] = 256 * q <= lx ? (gx ? px[q] * pg[q] : px[q]) : 0.0;
#pragma unroll
        for (int q = 0; q < RHS_QR; ++q)
          if (dR[q] >= 0) smem[dR[q]] = 256 * q <= lr ? pr[q] : 0.0;
      }
      __syncthreads();
      if (c0 + CH < s_end) fetch(c0 + CH);
      if (active) {
        const int steps = (nvalid + 3) >> 2;
#pragma unroll 2
        for (int ks = 0; ks < steps; ++ks) {
          const double* row = smem + (ks * 4 + kq) * W;
          double av[MU];
#pragma unroll
          for (int u = 0; u < MU; ++u) av[u] = row[oA[u]];
          // up to six column tiles at a time: one batch of LDS reads, then their products and MFMAs (all of a wide
          // tile row at once leaves the compiler no registers to batch the reads in)
          constexpr int HV = NV <= 6 ? NV : (NV % 6 == 0 ? 6 : 4);
#pragma unroll
          for (int h = 0; h < NV; h += HV) {
            double xv[HV], rv[HV];
#pragma unroll
            for (int v = 0; v < HV; ++v) {
              xv[v] = row[oX[h + v]];
              rv[v] = row[oR[h + v]];
            }
#pragma unroll
            for (int v = 0; v < HV; ++v) {
              const double bv = xv[v] * rv[v];
#pragma unroll
              for (int u = 0; u < MU; ++u) acc[u][h + v] = MFMA_F64(av[u], bv, acc[u][h + v]);
            }
          }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int l = (tm * MU + u) * 16 + kq + 4 * t;
            const int n = (tn * NV + v) * 16 + i;
            if (l < rl && n < N) dst[(int64_t)l * N + n] = acc[u][v][t];
          }
    }
  }
}

// Sum over the 256 threads of a workgroup, the same value on every thread; fixed order.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();   // red may still be read from the previous sum
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------
// Fused matrix-free product for one output and square bonds e = n in {16, 32}, p <= 8 (every
// middle site of a tensor train with one output):
//   q = J^T (h o (J v)),   J[s,(l,pp,r)] = L[s,l] X[s,pp] R[s,r]
// in ONE pass over the samples.  One workgroup of eight waves per compute unit owns a slab of samples
// and keeps the vector v (as the (e p) x n matrix of the step  L (x) X -> R) in LDS.  The waves split
// into two groups that work on consecutive chunks of 64 samples at the same time, so that each SIMD
// always holds one wave of either kind and the matrix pipe stays fed while the other wave loads,
// shuffles or stores:
//   waves 0-3 (chunk c)    u_s = h_s <(L_s (x) X_s) v, R_s> for 16 samples each: the operand
//                          L[s,l] X[s,pp] is formed in registers from rows fetched straight from
//                          memory one chunk ahead (lane = sample row), the result is dotted with R_s;
//                          the wave then leaves its rows [L | X | R | u] in the LDS image of chunk c;
//   waves 4-7 (chunk c-1)  acc[l,(pp,r)] += sum_s (u_s L[s,l]) X[s,pp] R[s,r] from the image of chunk
//                          c-1, a wave owns 2 (1 for e = 16) x TW output tiles across all chunks.
// One barrier per chunk hands an image over.  Each workgroup writes its own partial q (summed in fixed
// order by a second kernel) and its part of <v, q> = sum_s u_s (J v)_s, which conjugate gradients need.
// MODE 1 is one conjugate-gradient iteration's first half: the vector is p = r + beta p_old, formed
// while it is staged (beta from the partial sums of <r, r> the previous iteration left; workgroup 0
// stores p and the new rho), and the stopping test of scipy.sparse.linalg.cg runs first.
// cgst = [rho slot 0, rho slot 1, atol, stopped, iterations done].
// ---------------------------------------------------------------------------------------
#define MVF_CH 64
__device__ __forceinline__ double block_sum_512(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

template <int E4, int TW, int MODE>
__global__ __launch_bounds__(512, 2) void matvec_fused_kernel(
    const double* __restrict__ Lg, const double* __restrict__ Xg, const double* __restrict__ Rg,
    const double* __restrict__ Hg, const double* __restrict__ vsrc, const double* __restrict__ pold,
    double* __restrict__ pnew, double* __restrict__ cgst, const double* __restrict__ rho_part, int nrp, int it,
    int64_t S, int p, int64_t slab_len, int W, double* __restrict__ partial, double* __restrict__ pq_part) {
  extern __shared__ double smem[];
  __shared__ double red[8];
  constexpr int e = 4 * E4, n = e, MT = e / 16, NT = e / 16;
  const int K = e * p, N = p * n, D = K * n;
  double* Bs = smem;                   // [K][n]
  double* img0 = Bs + D;               // two images [MVF_CH][W]: L | X | R | u | 0
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  double beta = 0.0;
  if (MODE == 1) {
    if (cgst[3] != 0.0) return;
    double rho_new;
    if (it == 0) {
      rho_new = cgst[1];
    } else {
      double a = 0.0;
      for (int k = tid; k < nrp; k += 512) a += rho_part[k];
      rho_new = block_sum_512(a, red);
    }
    if (sqrt(rho_new) < cgst[2] || rho_new == 0.0) {
      if (blockIdx.x == 0 && tid == 0) cgst[3] = 1.0;
      return;
    }
    if (it > 0) beta = rho_new / cgst[(it + 1) & 1];
    if (blockIdx.x == 0 && tid == 0) cgst[it & 1] = rho_new;
  }
  const int64_t s_begin = (int64_t)blockIdx.x * slab_len;
  int64_t s_end = s_begin + slab_len;
  if (s_end > S) s_end = S;
#if defined(MVF_DIAG) && (MVF_DIAG & 8)
  s_end = s_begin;
#endif
  const int nchunks = s_end > s_begin ? (int)((s_end - s_begin + MVF_CH - 1) / MVF_CH) : 0;
  // ---- producer registers: rows of the next chunk, fetched before the vector is staged
  double er[E4], xr[8], rv[4][NT], hv[4];
  auto fetch = [&](int c) {   // tile `wave` of chunk c; rows past the slab end are clamped here and zeroed at use
    const int64_t row0 = s_begin + (int64_t)c * MVF_CH + wave * 16;
    const int64_t last = s_end - 1;
    const int64_t ri = min(row0 + i, last);
    const double* es = Lg + ri * e + kq;
    const double* xs = Xg + ri * p;
#pragma unroll
    for (int u = 0; u < E4; ++u) er[u] = es[4 * u];
#pragma unroll
    for (int q = 0; q < 8; ++q) xr[q] = xs[min(q, p - 1)];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t rt_ = min(row0 + kq + 4 * t, last);
#pragma unroll
      for (int j = 0; j < NT; ++j) rv[t][j] = Rg[rt_ * n + j * 16 + i];
      hv[t] = Hg ? Hg[rt_] : 1.0;
    }
  };
  if (wave < 4 && nchunks > 0) fetch(0);
  for (int k0 = tid; k0 < D; k0 += 512 * 12) {   // twelve per thread in flight: the headline core in one batch
    double a[12], b[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int k = min(k0 + 512 * u, D - 1);
      a[u] = vsrc[k];
      b[u] = (MODE == 1 && it > 0) ? pold[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int k = k0 + 512 * u;
      if (k < D) {
        const double val = MODE == 1 ? a[u] + beta * b[u] : a[u];
        Bs[k] = val;
        if (MODE == 1 && blockIdx.x == 0) pnew[k] = val;
      }
    }
  }
  for (int k = tid; k < 2 * MVF_CH; k += 512) img0[k * W + W - 1] = 0.0;
  // ---- consumer set-up: rows l = u 16 + i, columns (pp, r) = ((wave - 4) TW + v) 16 + i
  int oX[TW], oR[TW];
#pragma unroll
  for (int v = 0; v < TW; ++v) {
    const int col = (((wave - 4) & 3) * TW + v) * 16 + i;
    if (col < N) {
      const int pn = col / n;
      oX[v] = e + pn;
      oR[v] = e + p + (col - pn * n);
    } else {
      oX[v] = oR[v] = W - 1;
    }
  }
  d4 acc[MT][TW];
#pragma unroll
  for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int v = 0; v < TW; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
  double pq_acc = 0.0;
  const int ucol = 2 * e + p;
  __syncthreads();   // the vector is staged
  for (int c = 0; c <= nchunks; ++c) {
    if (wave < 4) {
      if (c < nchunks) {
        const int nvalid = (int)min((int64_t)MVF_CH, s_end - s_begin - (int64_t)c * MVF_CH);
        double* img = img0 + (c & 1) * MVF_CH * W;
        if (wave * 16 < nvalid) {   // wave-uniform
          const bool rowok = wave * 16 + i < nvalid;
          double ec[E4], xc[8], rc[4][NT], hc[4];
#pragma unroll
          for (int u = 0; u < E4; ++u) ec[u] = rowok ? er[u] : 0.0;
#pragma unroll
          for (int q = 0; q < 8; ++q) xc[q] = xr[q];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool ok = wave * 16 + kq + 4 * t < nvalid;
#pragma unroll
            for (int j = 0; j < NT; ++j) rc[t][j] = ok ? rv[t][j] : 0.0;
            hc[t] = hv[t];
          }
          if (c + 1 < nchunks) fetch(c + 1);
          // two accumulator sets (even / odd l-groups): four independent chains keep the matrix pipe issuing
          d4 a1[NT], a2[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) a1[j] = a2[j] = d4{0.0, 0.0, 0.0, 0.0};
#if !(defined(MVF_DIAG) && (MVF_DIAG & 1))
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (q < p) {   // uniform
#pragma unroll
              for (int u = 0; u < E4; u += 2) {
                const double a = ec[u] * xc[q], b = ec[u + 1] * xc[q];
                const double* brow = Bs + ((4 * u + kq) * p + q) * n + i;
                const double* crow = brow + 4 * p * n;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                  a1[j] = MFMA_F64(a, brow[j * 16], a1[j]);
                  a2[j] = MFMA_F64(b, crow[j * 16], a2[j]);
                }
              }
            }
          }
#endif
#pragma unroll
          for (int j = 0; j < NT; ++j) a1[j] += a2[j];
          // leave the rows in the image: L and X by row owner (lane i), R and u by result layout (row kq + 4 t)
          double* row = img + (wave * 16 + i) * W;
#pragma unroll
          for (int u = 0; u < E4; ++u) row[4 * u + kq] = ec[u];
#pragma unroll
          for (int q = 0; q < 8; q += 4)
            if (q + kq < p) row[e + q + kq] = rowok ? (kq == 0 ? xc[q] : kq == 1 ? xc[q + 1] : kq == 2 ? xc[q + 2] : xc[q + 3]) : 0.0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            double* rrow = img + (wave * 16 + kq + 4 * t) * W;
            double jv = 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              rrow[e + p + j * 16 + i] = rc[t][j];
              jv += a1[j][t] * rc[t][j];
            }
            jv += __shfl_xor(jv, 8, 16);
            jv += __shfl_xor(jv, 4, 16);
            jv += __shfl_xor(jv, 2, 16);
            jv += __shfl_xor(jv, 1, 16);
            const double u_ = jv * hc[t];
            if (i == 0) {
              rrow[ucol] = u_;
              pq_acc += jv * u_;
            }
          }
        }
      }
    } else if (c >= 1) {
      const int cc = c - 1;
      const int nvalid = (int)min((int64_t)MVF_CH, s_end - s_begin - (int64_t)cc * MVF_CH);
      const double* img = img0 + (cc & 1) * MVF_CH * W;
#if defined(MVF_DIAG) && (MVF_DIAG & 2)
      const int steps = 0;
#else
      const int steps = (nvalid + 3) >> 2;
#endif
#pragma unroll 2
      for (int ks = 0; ks < steps; ++ks) {
        const double* row = img + (ks * 4 + kq) * W;
        const double us = row[ucol];
        double av[MT], bv[TW];
#pragma unroll
        for (int u = 0; u < MT; ++u) av[u] = row[u * 16 + i] * us;
#pragma unroll
        for (int v = 0; v < TW; ++v) bv[v] = row[oX[v]] * row[oR[v]];
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
          for (int v = 0; v < TW; ++v) acc[u][v] = MFMA_F64(av[u], bv[v], acc[u][v]);
      }
    }
    __syncthreads();
  }
  if (wave >= 4) {
    double* dst = partial + (int64_t)blockIdx.x * D;
#pragma unroll
    for (int u = 0; u < MT; ++u)
#pragma unroll
      for (int v = 0; v < TW; ++v)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int l = u * 16 + kq + 4 * t;
          const int col = ((wave - 4) * TW + v) * 16 + i;
          if (col < N) dst[(int64_t)l * N + col] = acc[u][v][t];
        }
  }
  const double pq = block_sum_512(pq_acc, red);
  if (tid == 0) pq_part[blockIdx.x] = pq;
}

// Second half of a fused conjugate-gradient iteration: q = sum of the workgroup partials (16 outputs x
// 16 slab groups per workgroup, fixed order), alpha = rho / <p, q> from the partial dots, x += alpha p,
// r -= alpha q, and this workgroup's part of the new <r, r>.
__global__ __launch_bounds__(256) void cg_reduce_update_kernel(const double* __restrict__ partial, int nslabs,
                                                                int64_t D, const double* __restrict__ pq_part,
                                                                double* __restrict__ cgst, int it,
                                                                double* __restrict__ x, double* __restrict__ r,
                                                                const double* __restrict__ pvec,
                                                                double* __restrict__ rho_part) {
  __shared__ double part[16][17];
  __shared__ double red[4];
  if (cgst[3] != 0.0) return;
  double a = 0.0;
  for (int k = threadIdx.x; k < nslabs; k += 256) a += pq_part[k];
  const double alpha = cgst[it & 1] / block_sum_256(a, red);
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 16 + o;
  const int64_t src = idx < D ? idx : D - 1;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = g;
  for (; k + 48 < nslabs; k += 64) {
    a0 += partial[(int64_t)k * D + src];
    a1 += partial[(int64_t)(k + 16) * D + src];
    a2 += partial[(int64_t)(k + 32) * D + src];
    a3 += partial[(int64_t)(k + 48) * D + src];
  }
  for (; k < nslabs; k += 16) a0 += partial[(int64_t)k * D + src];
  part[g][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0) {
    double rr = 0.0;
    if (idx < D) {
      double q = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) q += part[j][o];
      x[idx] += alpha * pvec[idx];
      const double rn = r[idx] - alpha * q;
      r[idx] = rn;
      rr = rn * rn;
    }
    rr += __shfl_xor(rr, 8, 16);
    rr += __shfl_xor(rr, 4, 16);
    rr += __shfl_xor(rr, 2, 16);
    rr += __shfl_xor(rr, 1, 16);
    if (o == 0) rho_part[blockIdx.x] = rr;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cgst[4] += 1.0;
}

struct MvfDims {
  int ok, E4, TW, W, nwg, nrp;
  int64_t slab_len, D;
  size_t lds;
};

static MvfDims mvf_dims(int64_t S, int C, int rl, int p, int rr) {
  MvfDims d;
  d.ok = 0;
  if (C != 1 || rl != rr || (rl != 16 && rl != 32) || p < 1 || p > 8 || S < 1) return d;
  const int e = rl;
  d.E4 = e / 4;
  d.TW = (p * (e / 16) + 3) / 4;
  d.W = (2 * e + p + 2) | 1;   // odd: sample rows of the image fall on different banks both row-wise and column-wise
  d.D = (int64_t)e * p * e;
  int64_t nwg = (S + MVF_CH - 1) / MVF_CH;
  if (nwg > 256) nwg = 256;   // one workgroup (eight waves, most of the LDS) per compute unit
  d.nwg = (int)nwg;
  d.slab_len = round_up64((S + nwg - 1) / nwg, 4);
  d.nrp = (int)((d.D + 15) / 16);
  d.lds = ((size_t)d.D + (size_t)2 * MVF_CH * d.W) * sizeof(double);
  d.ok = d.lds <= 150 * 1024;
  return d;
}

typedef void (*mvf_fn_t)(const double*, const double*, const double*, const double*, const double*, const double*, double*,
                         double*, const double*, int, int, int64_t, int, int64_t, int, double*, double*);
template <int MODE>
static mvf_fn_t mvf_pick(int E4, int TW) {
  if (E4 == 4) {
    switch (TW) {
      case 1: return matvec_fused_kernel<4, 1, MODE>;
      default: return matvec_fused_kernel<4, 2, MODE>;
    }
  }
  switch (TW) {
    case 1: return matvec_fused_kernel<8, 1, MODE>;
    case 2: return matvec_fused_kernel<8, 2, MODE>;
    case 3: return matvec_fused_kernel<8, 3, MODE>;
    default: return matvec_fused_kernel<8, 4, MODE>;
  }
}

static bool mvf_timing_begin(hipStream_t st);
static void mvf_timing_end(hipStream_t st);

static int mvf_launch(int mode, const MvfDims& d, const double* L, const double* X, const double* R, const double* H,
                      const double* vsrc, const double* pold, double* pnew, double* cgst, const double* rho_part, int it,
                      int64_t S, int p, double* partial, double* pq_part, hipStream_t st) {
  mvf_fn_t fn = mode ? mvf_pick<1>(d.E4, d.TW) : mvf_pick<0>(d.E4, d.TW);
  static bool allowed[2][2][8][TTK_MAX_DEVICES];
  if (d.TW < 0 || d.TW > 7 || !allow_big_lds((const void*)fn, allowed[mode ? 1 : 0][d.E4 == 8 ? 1 : 0][d.TW], 150 * 1024))
    return TTK_ELAUNCH;
  const bool timed = mvf_timing_begin(st);
  hipLaunchKernelGGL(fn, dim3((unsigned)d.nwg), dim3(512), d.lds, st, L, X, R, H, vsrc, pold, pnew, cgst, rho_part, d.nrp, it,
                     S, p, d.slab_len, d.W, partial, pq_part);
  if (timed) mvf_timing_end(st);
  return ttk_launch_status();
}

// hu[s,d] = sum_c H[s,c,d] u[s,c]
__global__ __launch_bounds__(256) void jvp_hess_kernel(const double* __restrict__ u, const double* __restrict__ H,
                                                        double* __restrict__ hu, int64_t S, int C) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * C) return;
  const int64_t s = idx / C;
  const int d = (int)(idx - s * C);
  double v = 0.0;
  for (int c = 0; c < C; ++c) v += H[(s * C + c) * C + d] * u[s * C + c];
  hu[idx] = v;
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

// Everything the single-launch solve needs before its factorisation, in one launch instead of two kernels and
// three fills: the right-hand-side block row of the storage (row 0 = -(b/scale + ridge w), zero behind it and in
// the 63 rows below), the control words at zero with the spin limit in word 4, the pivot report at zero.
__global__ __launch_bounds__(256) void solve_prep_kernel(const double* __restrict__ b, const double* __restrict__ w,
                                                          const double* __restrict__ scale, double ridge,
                                                          double* __restrict__ rows, int64_t nrows_elems, int64_t D,
                                                          unsigned* __restrict__ ctl, int nctl, unsigned spin,
                                                          int* __restrict__ info) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < nrows_elems) {
    double v = 0.0;
    if (idx < D) {
      v = scale ? b[idx] / scale[0] : b[idx];
      if (w) v += ridge * w[idx];
      v = -v;
    }
    rows[idx] = v;
  }
  if (idx < nctl) ctl[idx] = idx == 4 ? spin : 0u;
  if (idx == 0) *info = 0;
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

// Sum over the 1024 threads of a workgroup, the same value on every thread; fixed order.
__device__ __forceinline__ double block_sum_1024(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();   // red may still be read from the previous sum
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}

// The vector half of one conjugate-gradient iteration in one workgroup (the vectors have the size
// of one core): the stopping test on the current residual, then
//   alpha = rho / <p,q>;  x += alpha p;  r -= alpha q;  rho' = <r,r>;  p = r + (rho'/rho) p.
// state = [rho, atol, stopped, updates done].  Once the test ||r|| < atol holds the kernel only
// sets `stopped` and leaves x, r, p as they are, so iterations queued behind it change nothing.
// REG: the vectors fit eight elements per thread and stay in registers between the passes.
template <bool REG>
__global__ __launch_bounds__(1024) void cg_update_kernel(double* __restrict__ x, double* __restrict__ r,
                                                          double* __restrict__ p, const double* __restrict__ q,
                                                          double* __restrict__ state, int64_t D) {
  __shared__ double red[16];
  const double rho = state[0], atol = state[1];
  if (sqrt(rho) < atol || rho == 0.0) {   // a zero residual (e.g. b == 0 from a zero start) is a solution: no 0/0 step
    if (threadIdx.x == 0) state[2] = 1.0;
    return;
  }
  double rho_new;
  if (REG) {
    double xv[8], rv[8], pv[8], qv[8];
    const int n = (int)D;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = min((int)threadIdx.x + 1024 * u, n - 1);
      xv[u] = x[k];
      rv[u] = r[k];
      pv[u] = p[k];
      qv[u] = q[k];
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += ((int)threadIdx.x + 1024 * u < n) ? pv[u] * qv[u] : 0.0;
    const double alpha = rho / block_sum_1024(acc, red);
    acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      xv[u] += alpha * pv[u];
      rv[u] -= alpha * qv[u];
      acc += ((int)threadIdx.x + 1024 * u < n) ? rv[u] * rv[u] : 0.0;
    }
    rho_new = block_sum_1024(acc, red);
    const double beta = rho_new / rho;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = (int)threadIdx.x + 1024 * u;
      if (k < n) {
        x[k] = xv[u];
        r[k] = rv[u];
        p[k] = rv[u] + beta * pv[u];
      }
    }
  } else {
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < D; k += 1024) acc += p[k] * q[k];
    const double alpha = rho / block_sum_1024(acc, red);
    acc = 0.0;
    for (int64_t k = threadIdx.x; k < D; k += 1024) {
      x[k] += alpha * p[k];
      const double rn = r[k] - alpha * q[k];
      r[k] = rn;
      acc += rn * rn;
    }
    rho_new = block_sum_1024(acc, red);
    const double beta = rho_new / rho;
    for (int64_t k = threadIdx.x; k < D; k += 1024) p[k] = r[k] + beta * p[k];
  }
  if (threadIdx.x == 0) {
    state[0] = rho_new;
    state[3] += 1.0;
  }
}

// The same update for vectors beyond one workgroup's registers, spread over CGW_WG workgroups in four launches; `ws`
// holds 2 CGW_WG partial sums.  Every launch tests the stopping rule on state[0], which only the last one rewrites.
//   <0> partial <p, q>;  <1> alpha from their sum, x += alpha p, r -= alpha q, partial <r, r>;
//   <2> beta from their sum, p = r + beta p;  cg_wide_scalars_kernel: state[0] = <r, r>, one more update counted.
#define CGW_WG 128
template <int PHASE>
__global__ __launch_bounds__(256) void cg_wide_kernel(double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                       const double* __restrict__ q, double* __restrict__ state,
                                                       double* __restrict__ ws, int64_t D) {
  __shared__ double red[4];
  const double rho = state[0], atol = state[1];
  if (sqrt(rho) < atol || rho == 0.0) {
    if (PHASE == 0 && blockIdx.x == 0 && threadIdx.x == 0) state[2] = 1.0;
    return;
  }
  const int64_t per = ((D + CGW_WG - 1) / CGW_WG + 1) & ~(int64_t)1;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(D, k0 + per);
  if (PHASE == 0) {
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) acc += p[k] * q[k];
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
  } else if (PHASE == 1) {
    const double alpha = rho / block_sum_256(threadIdx.x < CGW_WG ? ws[threadIdx.x] : 0.0, red);
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      x[k] += alpha * p[k];
      const double rn = r[k] - alpha * q[k];
      r[k] = rn;
      acc += rn * rn;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ws[CGW_WG + blockIdx.x] = acc;
  } else {
    const double beta = block_sum_256(threadIdx.x < CGW_WG ? ws[CGW_WG + threadIdx.x] : 0.0, red) / rho;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) p[k] = r[k] + beta * p[k];
  }
}
__global__ __launch_bounds__(256) void cg_wide_scalars_kernel(double* __restrict__ state, const double* __restrict__ ws) {
  __shared__ double red[4];
  const double rho = state[0], atol = state[1];
  if (sqrt(rho) < atol || rho == 0.0) return;
  const double rho_new = block_sum_256(threadIdx.x < CGW_WG ? ws[CGW_WG + threadIdx.x] : 0.0, red);
  if (threadIdx.x == 0) {
    state[0] = rho_new;
    state[3] += 1.0;
  }
}

// The vector half of one Lanczos iteration j >= 1 in one workgroup:
//   w -= beta[j-1] v_prev;  alpha[j-1] = <w, v_cur>;  w -= alpha[j-1] v_cur;  beta[j] = ||w||;  v_next = w / beta[j].
template <bool REG>
__global__ __launch_bounds__(1024) void lanczos_update_kernel(double* __restrict__ w, const double* __restrict__ v_prev,
                                                               const double* __restrict__ v_cur,
                                                               double* __restrict__ v_next, double* __restrict__ alphas,
                                                               double* __restrict__ betas, int j, int64_t D, double tol,
                                                               double* __restrict__ stop) {
  __shared__ double red[16];
  if (*stop != 0.0) return;      // an earlier iteration found beta < tol: the basis ends there
  const double bprev = betas[j - 1];
  double a, bj;
  if (REG) {
    double wv[8], cv[8];
    const int n = (int)D;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = min((int)threadIdx.x + 1024 * u, n - 1);
      wv[u] = w[k];
      cv[u] = v_cur[k];
      if (v_prev) wv[u] -= bprev * v_prev[k];
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += ((int)threadIdx.x + 1024 * u < n) ? wv[u] * cv[u] : 0.0;
    a = block_sum_1024(acc, red);
    acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      wv[u] -= a * cv[u];
      acc += ((int)threadIdx.x + 1024 * u < n) ? wv[u] * wv[u] : 0.0;
    }
    bj = sqrt(block_sum_1024(acc, red));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = (int)threadIdx.x + 1024 * u;
      if (k < n) {
        w[k] = wv[u];
        v_next[k] = wv[u] / bj;
      }
    }
  } else {
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < D; k += 1024) {
      double wk = w[k];
      if (v_prev) {
        wk -= bprev * v_prev[k];
        w[k] = wk;
      }
      acc += wk * v_cur[k];
    }
    a = block_sum_1024(acc, red);
    acc = 0.0;
    for (int64_t k = threadIdx.x; k < D; k += 1024) {
      const double wk = w[k] - a * v_cur[k];
      w[k] = wk;
      acc += wk * wk;
    }
    bj = sqrt(block_sum_1024(acc, red));
    for (int64_t k = threadIdx.x; k < D; k += 1024) v_next[k] = w[k] / bj;
  }
  if (threadIdx.x == 0) {
    alphas[j - 1] = a;
    betas[j] = bj;
    if (bj < tol) *stop = (double)j;
  }
}

// The same step for vectors beyond one workgroup's registers (cores of conv tensor trains: 48 x 65 x 48), spread
// over LZW_WG workgroups in four launches; the partial sums wait in the first 2 LZW_WG entries of v_next, which is
// written last:
//   <0>  w -= beta[j-1] v_prev, partial <w, v_cur>            -> v_next[wg]
//   <1>  alpha = sum of those (every workgroup, fixed order); w -= alpha v_cur, partial <w, w> -> v_next[LZW_WG + wg]
//   lanczos_wide_scalars_kernel (one workgroup): alphas[j-1], betas[j] = sqrt(sum); the only writer of `stop`
//   <2>  v_next = w / betas[j]
#define LZW_WG 128
template <int PHASE>
__global__ __launch_bounds__(256) void lanczos_wide_kernel(double* __restrict__ w, const double* __restrict__ v_prev,
                                                            const double* __restrict__ v_cur, double* __restrict__ v_next,
                                                            const double* __restrict__ betas, int j, int64_t D,
                                                            const double* __restrict__ stop) {
  __shared__ double red[4];
  if (*stop != 0.0) return;
  const int64_t per = ((D + LZW_WG - 1) / LZW_WG + 1) & ~(int64_t)1;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(D, k0 + per);
  if (PHASE == 0) {
    const double bprev = betas[j - 1];
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      double wk = w[k];
      if (v_prev) {
        wk -= bprev * v_prev[k];
        w[k] = wk;
      }
      acc += wk * v_cur[k];
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) v_next[blockIdx.x] = acc;
  } else if (PHASE == 1) {
    double a = threadIdx.x < LZW_WG ? v_next[threadIdx.x] : 0.0;
    a = block_sum_256(a, red);
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      const double wk = w[k] - a * v_cur[k];
      w[k] = wk;
      acc += wk * wk;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) v_next[LZW_WG + blockIdx.x] = acc;
  } else {
    const double bj = betas[j];
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) v_next[k] = w[k] / bj;
  }
}
__global__ __launch_bounds__(256) void lanczos_wide_scalars_kernel(const double* __restrict__ v_next,
                                                                    double* __restrict__ alphas, double* __restrict__ betas,
                                                                    int j, double tol, double* __restrict__ stop) {
  __shared__ double red[4];
  if (*stop != 0.0) return;
  const double a = block_sum_256(threadIdx.x < LZW_WG ? v_next[threadIdx.x] : 0.0, red);
  const double b = block_sum_256(threadIdx.x < LZW_WG ? v_next[LZW_WG + threadIdx.x] : 0.0, red);
  if (threadIdx.x == 0) {
    alphas[j - 1] = a;
    betas[j] = sqrt(b);
    if (sqrt(b) < tol) *stop = (double)j;
  }
}

// The end of a Lanczos-Galerkin step (tensor/network.py:815-824): with m = the iteration at which beta fell
// below the tolerance (`stop`, 0: none did, then m = m_run), solve T_m y = beta_1 e_1 for the symmetric
// tridiagonal T_m (diagonal alphas[0..m), off-diagonal betas[1..m)) by Gaussian elimination with partial
// pivoting (the algorithm of LAPACK's dgtsv, which is what a dense LU of a tridiagonal matrix amounts to).
// One thread does the O(m) recurrence on arrays of 5 m doubles: in LDS where they fit (`in_lds`), else in the
// workspace.  y and m (as y[-1], i.e. ws[0]) are left in the workspace for lanczos_combine_kernel.
__global__ __launch_bounds__(64) void lanczos_tridiag_kernel(const double* __restrict__ alphas, const double* __restrict__ betas,
                                                              const double* __restrict__ stop, int m_run, int mcap,
                                                              double* __restrict__ ws, int in_lds) {
  extern __shared__ double tri_lds[];
  const int m = *stop != 0.0 ? (int)*stop : m_run;
  double* y = ws + 1;
  double* a = in_lds ? tri_lds : ws + 1 + mcap;
  double *d = a, *dl = a + mcap, *du = a + 2 * mcap, *du2 = a + 3 * mcap, *b = a + 4 * mcap;
  for (int i = threadIdx.x; i < m; i += 64) {
    d[i] = alphas[i];
    b[i] = i == 0 ? betas[0] : 0.0;
    if (i + 1 < m) dl[i] = du[i] = betas[i + 1];
    du2[i] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[0] = (double)m;
    for (int i = 0; i + 1 < m; ++i) {
      if (fabs(d[i]) >= fabs(dl[i])) {
        const double fact = dl[i] / d[i];
        d[i + 1] -= fact * du[i];
        b[i + 1] -= fact * b[i];
      } else {
        const double fact = d[i] / dl[i];
        d[i] = dl[i];
        const double t = d[i + 1];
        d[i + 1] = du[i] - fact * t;
        if (i + 2 < m) {
          du2[i] = du[i + 1];
          du[i + 1] = -fact * du[i + 1];
        }
        du[i] = t;
        const double tb = b[i];
        b[i] = b[i + 1];
        b[i + 1] = tb - fact * b[i + 1];
      }
    }
    b[m - 1] /= d[m - 1];
    if (m > 1) b[m - 2] = (b[m - 2] - du[m - 2] * b[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) b[i] = (b[i] - du[i] * b[i + 1] - du2[i] * b[i + 2]) / d[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += 64) y[i] = b[i];
}

// x = x0 + sum_{j < m} y[j] V[j, :] over the rows of the Krylov basis V (row stride ldv), m and y from the
// workspace lanczos_tridiag_kernel left: one streaming pass over the m basis vectors that were written.
__global__ __launch_bounds__(256) void lanczos_combine_kernel(const double* __restrict__ V, int64_t ldv,
                                                               const double* __restrict__ ws, const double* __restrict__ x0,
                                                               double* __restrict__ x, int64_t D) {
  const int m = (int)ws[0];
  const double* y = ws + 1;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < D; k += (int64_t)gridDim.x * 256) {
    double acc0 = 0.0, acc1 = 0.0;
    int j = 0;
    for (; j + 1 < m; j += 2) {
      acc0 += y[j] * V[(int64_t)j * ldv + k];
      acc1 += y[j + 1] * V[(int64_t)(j + 1) * ldv + k];
    }
    if (j < m) acc0 += y[j] * V[(int64_t)j * ldv + k];
    x[k] = x0[k] + (acc0 + acc1);
  }
}

// out[row] = <J[row, :], v>: explicit Jacobian rows of a narrow node (a conv filter) times a vector.  LPR lanes
// share a row (consecutive entries of a row go to consecutive lanes: a wave reads 64 / LPR whole rows, which lie
// back to back in memory), the vector sits in registers, one pass over J at memory speed.
template <int LPR>
__global__ __launch_bounds__(256) void rows_dot_kernel(const double* __restrict__ J, const double* __restrict__ v,
                                                        double* __restrict__ out, int64_t rows, int D) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  constexpr int NV = 4;   // entries of v per lane held in registers (D <= NV * LPR), the rest read again
  double vr[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) vr[u] = sub + u * LPR < D ? v[sub + u * LPR] : 0.0;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t row = min(r0 + rsel, rows - 1);
    const double* jr = J + row * D;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < NV; ++u) acc += sub + u * LPR < D ? jr[sub + u * LPR] * vr[u] : 0.0;
    for (int d = sub + NV * LPR; d < D; d += LPR) acc += jr[d] * v[d];
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0 && r0 + rsel < rows) out[r0 + rsel] = acc;
  }
}

// Per-sample blocks of explicit rows J[s] (C x D, rows back to back in memory) against the per-sample Hessian, one
// pass over J:  u_s = J_s x_s,  hu_s[d] = sum_c H[s,c,d] u_s[c],  then
//   MODE 0 (x_s = v for every sample):  q = sum_s J_s^T hu_s       the whole operator v -> J^T H J v of a narrow
//                                       node (a conv filter node); one partial q per workgroup, summed in fixed
//                                       order by reduce_slabs_wide_kernel;
//   MODE 1 (x_s = z[s, :]):             out[s, :] = J_s^T hu_s     with J = L (left environments, C x rl per sample)
//                                       this is GL = L^T H (L z): prediction, Hessian product and g.L of a
//                                       many-feature site in one kernel instead of three (L read once, not twice).
// A workgroup stages NS whole samples (NS C D doubles, one contiguous range) in LDS; LPR lanes share a row in the
// first product, the Hessian product is one thread per (sample, output), the last product one thread per column.
template <int MODE, int LPR, int NW = 2>   // NW: 64-column groups of a row (D <= 64 NW)
__global__ __launch_bounds__(256) void rows_hjv_kernel(const double* __restrict__ J, const double* __restrict__ H,
                                                        const double* __restrict__ x, double* __restrict__ out, int64_t S,
                                                        int C, int D, int NS) {
  extern __shared__ double sm[];
  double* Jc = sm;                         // [NS C][D]
  double* ub = Jc + (size_t)NS * C * D;    // [NS C]
  double* hb = ub + NS * C;                // [NS C]
  double* Hc = hb + NS * C;                // [NS][C][C]
  double* xs = Hc + (size_t)NS * C * C;    // MODE 0: v [D]; MODE 1: z of the chunk [NS][D]
  const int tid = threadIdx.x, sub = tid % LPR, grp = tid / LPR;
  constexpr int NGRP = 256 / LPR;
  const int64_t nchunks = (S + NS - 1) / NS;
  constexpr int NU = NW == 2 ? 4 : 8;      // steps of LPR entries per row in the first product (D <= NU LPR)
  double q[NW];                            // MODE 0: columns tid % 64 + 64 w, rows tid / 64 mod 4
#pragma unroll
  for (int w = 0; w < NW; ++w) q[w] = 0.0;
  // A chunk travels memory -> registers -> LDS; the registers are filled for chunk c+1 before chunk c is worked
  // on, so the loads (unconditional, clamped) are in flight during the three phases.
  double tj[16], th[8], tx[16];
  auto fetch = [&](int64_t ch) {
    const int64_t s0 = ch * NS;
    const int ns = (int)min((int64_t)NS, S - s0);
    const int n = ns * C * D, nh = ns * C * C, nx = ns * D;
    const double* sj = J + s0 * C * D;
    const double* sh = H + s0 * C * C;
#pragma unroll
    for (int u = 0; u < 16; ++u) tj[u] = sj[min(tid + 256 * u, n - 1)];
    if (MODE != 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) th[u] = sh[min(tid + 256 * u, nh - 1)];
    } else {   // MODE 2: x = g (S, C), the weights of the rows; no Hessian, no first product
#pragma unroll
      for (int u = 0; u < 8; ++u) th[u] = x[s0 * C + min(tid + 256 * u, ns * C - 1)];
    }
    if (MODE == 1) {
      const double* sx = x + s0 * D;
#pragma unroll
      for (int u = 0; u < 16; ++u) tx[u] = sx[min(tid + 256 * u, nx - 1)];
    }
  };
  if (MODE == 0)
    for (int d = tid; d < D; d += 256) xs[d] = x[d];
  if ((int64_t)blockIdx.x < nchunks) fetch(blockIdx.x);
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t s0 = ch * NS;
    const int ns = (int)min((int64_t)NS, S - s0);
    const int rows = ns * C, n = rows * D;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (tid + 256 * u < n) Jc[tid + 256 * u] = tj[u];
    if (MODE != 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (tid + 256 * u < rows * C) Hc[tid + 256 * u] = th[u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (tid + 256 * u < rows) hb[tid + 256 * u] = th[u];
    }
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (tid + 256 * u < ns * D) xs[tid + 256 * u] = tx[u];
    }
    __syncthreads();
    if (ch + gridDim.x < nchunks) fetch(ch + gridDim.x);
    // D <= NU LPR: NU predicated steps per row, two rows per pass, all LDS reads of a pass issued together
    if (MODE != 2)
    for (int row = grp; row < rows; row += 2 * NGRP) {
      const int rowb = min(row + NGRP, rows - 1);
      const double* ja = Jc + row * D;
      const double* jb = Jc + rowb * D;
      const double* xa = MODE == 0 ? xs : xs + (row / C) * D;
      const double* xb = MODE == 0 ? xs : xs + (rowb / C) * D;
      double pa[NU], pb[NU], va[NU], vb[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int d = min(sub + u * LPR, D - 1);
        pa[u] = ja[d];
        pb[u] = jb[d];
        va[u] = xa[d];
        vb[u] = xb[d];
      }
      double acca = 0.0, accb = 0.0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const bool ok = sub + u * LPR < D;
        acca += ok ? pa[u] * va[u] : 0.0;
        accb += ok ? pb[u] * vb[u] : 0.0;
      }
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) {
        acca += __shfl_xor(acca, off, 64);
        accb += __shfl_xor(accb, off, 64);
      }
      if (sub == 0) {
        ub[row] = acca;
        if (row + NGRP < rows) ub[row + NGRP] = accb;
      }
    }
    if (MODE != 2) {
      __syncthreads();
      for (int t = tid; t < rows; t += 256) {
        const int sl = t / C, d = t - sl * C;
        const double* Hs = Hc + sl * C * C;
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += Hs[c * C + d] * ub[sl * C + c];
        hb[t] = acc;
      }
      __syncthreads();
    }
    if (MODE != 1) {
      const int d = tid & 63, g4 = tid >> 6;
      int dc[NW];
      bool okc[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        dc[w] = min(d + 64 * w, D - 1);
        okc[w] = d + 64 * w < D;
      }
      int row = g4;
      for (; row + 4 < rows; row += 8) {   // two rows per pass
        const double h0 = hb[row], h1 = hb[row + 4];
        double j0[NW], j1[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          j0[w] = Jc[row * D + dc[w]];
          j1[w] = Jc[(row + 4) * D + dc[w]];
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          q[w] += okc[w] ? j0[w] * h0 : 0.0;
          q[w] += okc[w] ? j1[w] * h1 : 0.0;
        }
      }
      for (; row < rows; row += 4) {
        const double h = hb[row];
#pragma unroll
        for (int w = 0; w < NW; ++w) q[w] += okc[w] ? Jc[row * D + dc[w]] * h : 0.0;
      }
    } else {
      for (int t = tid; t < ns * D; t += 256) {
        const int sl = t / D, d = t - sl * D;
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += Jc[(sl * C + c) * D + d] * hb[sl * C + c];
        out[s0 * D + t] = acc;
      }
    }
    __syncthreads();
  }
  if (MODE != 1) {   // the four row groups of each column, fixed order
    double* red = sm;   // [4][64 NW]
#pragma unroll
    for (int w = 0; w < NW; ++w) red[(tid >> 6) * (64 * NW) + 64 * w + (tid & 63)] = q[w];
    __syncthreads();
    for (int d = tid; d < D; d += 256)
      out[(int64_t)blockIdx.x * D + d] = (red[d] + red[64 * NW + d]) + (red[128 * NW + d] + red[192 * NW + d]);
  }
}

// The vector half of one MINRES iteration (Paige & Saunders, as scipy.sparse.linalg.minres runs it
// without shift or preconditioner) in one workgroup; every scalar of the recurrence lives in `st`
// on the device, so the host never reads anything inside the loop.  On entry y = A v.
// st: 0 istop (0 = running; SciPy's codes, 6 = iteration limit, -1 = beta2 = 0), 1 itn, 2 oldb, 3 beta,
//     4 dbar, 5 epsln, 6 phibar, 7 rhs1, 8 rhs2, 9 tnorm2, 10 gmax, 11 gmin, 12 cs, 13 sn, 14 beta1,
//     15 rtol, 16 maxiter, 17 machine epsilon.
// Vectors: y (A v in, r2 of the new iteration out), r1, r2, v (next Lanczos vector out), w, w2, x.
// Once istop != 0 the kernel returns at once: calls queued behind the last iteration change nothing.
__global__ __launch_bounds__(1024) void minres_update_kernel(double* __restrict__ y, double* __restrict__ r1,
                                                              double* __restrict__ r2, double* __restrict__ v,
                                                              double* __restrict__ w, double* __restrict__ w2,
                                                              double* __restrict__ x, double* __restrict__ st,
                                                              int64_t D) {
  __shared__ double red[16];
  if (st[0] != 0.0) return;
  const double itn = st[1] + 1.0;
  double oldb = st[2], beta = st[3], dbar = st[4], epsln = st[5], phibar = st[6], rhs1 = st[7], rhs2 = st[8];
  double tnorm2 = st[9], gmax = st[10], gmin = st[11], cs = st[12], sn = st[13];
  const double beta1 = st[14], rtol = st[15], maxiter = st[16], eps = st[17];
  // y <- A v - (beta / oldb) r1;  alfa = <v, y>
  const double c1 = itn >= 2.0 ? beta / oldb : 0.0;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < D; k += 1024) {
    const double yk = itn >= 2.0 ? y[k] - c1 * r1[k] : y[k];
    y[k] = yk;
    acc += v[k] * yk;
  }
  const double alfa = block_sum_1024(acc, red);
  // y <- y - (alfa / beta) r2;  r1 <- r2;  r2 <- y;  beta' = ||y||
  const double c2 = alfa / beta;
  acc = 0.0;
  for (int64_t k = threadIdx.x; k < D; k += 1024) {
    const double r2k = r2[k];
    const double yk = y[k] - c2 * r2k;
    r1[k] = r2k;
    r2[k] = yk;
    acc += yk * yk;
  }
  oldb = beta;
  beta = sqrt(block_sum_1024(acc, red));
  tnorm2 += alfa * alfa + oldb * oldb + beta * beta;
  double istop = 0.0;
  if (itn == 1.0 && beta / beta1 <= 10.0 * eps) istop = -1.0;
  const double oldeps = epsln;
  const double delta = cs * dbar + sn * alfa;
  const double gbar = sn * dbar - cs * alfa;
  epsln = sn * beta;
  dbar = -cs * beta;
  const double root = hypot(gbar, dbar);
  const double gamma = fmax(hypot(gbar, beta), eps);
  cs = gbar / gamma;
  sn = beta / gamma;
  const double phi = cs * phibar;
  phibar = sn * phibar;
  const double denom = 1.0 / gamma;
  // w1 <- w2; w2 <- w; w <- (v - oldeps w1 - delta w2) / gamma;  x += phi w;  v <- y / beta' (the next Lanczos vector)
  const double rb = 1.0 / beta;
  acc = 0.0;
  for (int64_t k = threadIdx.x; k < D; k += 1024) {
    const double w1k = w2[k], w2k = w[k];
    const double wk = (v[k] - oldeps * w1k - delta * w2k) * denom;
    w2[k] = w2k;
    w[k] = wk;
    const double xk = x[k] + phi * wk;
    x[k] = xk;
    acc += xk * xk;
    v[k] = rb * r2[k];
  }
  const double ynorm = sqrt(block_sum_1024(acc, red));
  gmax = fmax(gmax, gamma);
  gmin = fmin(gmin, gamma);
  const double z = rhs1 / gamma;
  rhs1 = rhs2 - delta * z;
  rhs2 = -epsln * z;
  const double Anorm = sqrt(tnorm2);
  const double epsx = Anorm * ynorm * eps;
  const double rnorm = phibar;
  const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
  const double test2 = Anorm == 0.0 ? INFINITY : root / Anorm;
  const double Acond = gmax / gmin;
  if (istop == 0.0) {
    const double t1 = 1.0 + test1, t2 = 1.0 + test2;
    if (t2 <= 1.0) istop = 2.0;
    if (t1 <= 1.0) istop = 1.0;
    if (itn >= maxiter) istop = 6.0;
    if (Acond >= 0.1 / eps) istop = 4.0;
    if (epsx >= beta1) istop = 3.0;
    if (test2 <= rtol) istop = 2.0;
    if (test1 <= rtol) istop = 1.0;
  }
  if (threadIdx.x == 0) {
    st[0] = istop; st[1] = itn; st[2] = oldb; st[3] = beta; st[4] = dbar; st[5] = epsln; st[6] = phibar;
    st[7] = rhs1; st[8] = rhs2; st[9] = tnorm2; st[10] = gmax; st[11] = gmin; st[12] = cs; st[13] = sn;
  }
}

// The same iteration for vectors beyond one workgroup's reach (cores of conv tensor trains), spread over MRW_WG
// workgroups: three vector launches with their partial sums in `ws` (3 MRW_WG doubles) and one scalar launch, the only
// one that writes the state, so that every workgroup of an iteration reads the same scalars.
#define MRW_WG 128
struct MinresStep {   // the scalars of one iteration that the vector updates need
  double oldeps, delta, gamma, phi, beta_new;
};
__device__ __forceinline__ MinresStep minres_step(const double* __restrict__ st, double alfa, double beta_new) {
  const double dbar = st[4], epsln = st[5], phibar = st[6], cs = st[12], sn = st[13], eps = st[17];
  MinresStep m;
  m.oldeps = epsln;
  m.delta = cs * dbar + sn * alfa;
  const double gbar = sn * dbar - cs * alfa;
  m.gamma = fmax(hypot(gbar, beta_new), eps);
  m.phi = (gbar / m.gamma) * phibar;
  m.beta_new = beta_new;
  return m;
}
template <int PHASE>
__global__ __launch_bounds__(256) void minres_wide_kernel(double* __restrict__ y, double* __restrict__ r1,
                                                           double* __restrict__ r2, double* __restrict__ v,
                                                           double* __restrict__ w, double* __restrict__ w2,
                                                           double* __restrict__ x, const double* __restrict__ st,
                                                           double* __restrict__ ws, int64_t D) {
  __shared__ double red[4];
  if (st[0] != 0.0) return;
  const double itn = st[1] + 1.0, oldb = st[2], beta = st[3];
  const int64_t per = ((D + MRW_WG - 1) / MRW_WG + 1) & ~(int64_t)1;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(D, k0 + per);
  if (PHASE == 0) {          // y <- A v - (beta / oldb) r1;  partial <v, y>
    const double c1 = itn >= 2.0 ? beta / oldb : 0.0;
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      const double yk = itn >= 2.0 ? y[k] - c1 * r1[k] : y[k];
      y[k] = yk;
      acc += v[k] * yk;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
  } else if (PHASE == 1) {   // y <- y - (alfa / beta) r2;  r1 <- r2;  r2 <- y;  partial <y, y>
    const double alfa = block_sum_256(threadIdx.x < MRW_WG ? ws[threadIdx.x] : 0.0, red);
    const double c2 = alfa / beta;
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      const double r2k = r2[k];
      const double yk = y[k] - c2 * r2k;
      r1[k] = r2k;
      r2[k] = yk;
      acc += yk * yk;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ws[MRW_WG + blockIdx.x] = acc;
  } else {                   // w, x and the next Lanczos vector;  partial <x, x>
    const double alfa = block_sum_256(threadIdx.x < MRW_WG ? ws[threadIdx.x] : 0.0, red);
    const double bnew = sqrt(block_sum_256(threadIdx.x < MRW_WG ? ws[MRW_WG + threadIdx.x] : 0.0, red));
    const MinresStep m = minres_step(st, alfa, bnew);
    const double denom = 1.0 / m.gamma, rb = 1.0 / bnew;
    double acc = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
      const double w1k = w2[k], w2k = w[k];
      const double wk = (v[k] - m.oldeps * w1k - m.delta * w2k) * denom;
      w2[k] = w2k;
      w[k] = wk;
      const double xk = x[k] + m.phi * wk;
      x[k] = xk;
      acc += xk * xk;
      v[k] = rb * r2[k];
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) ws[2 * MRW_WG + blockIdx.x] = acc;
  }
}
// The scalar half of the iteration: the recurrences and stopping tests of minres_update_kernel on the three sums.
__global__ __launch_bounds__(256) void minres_wide_scalars_kernel(double* __restrict__ st, const double* __restrict__ ws) {
  __shared__ double red[4];
  if (st[0] != 0.0) return;
  const double alfa = block_sum_256(threadIdx.x < MRW_WG ? ws[threadIdx.x] : 0.0, red);
  const double bsq = block_sum_256(threadIdx.x < MRW_WG ? ws[MRW_WG + threadIdx.x] : 0.0, red);
  const double xsq = block_sum_256(threadIdx.x < MRW_WG ? ws[2 * MRW_WG + threadIdx.x] : 0.0, red);
  if (threadIdx.x != 0) return;
  const double itn = st[1] + 1.0;
  double oldb = st[2], beta = st[3], dbar = st[4], epsln = st[5], phibar = st[6], rhs1 = st[7], rhs2 = st[8];
  double tnorm2 = st[9], gmax = st[10], gmin = st[11], cs = st[12], sn = st[13];
  const double beta1 = st[14], rtol = st[15], maxiter = st[16], eps = st[17];
  oldb = beta;
  beta = sqrt(bsq);
  tnorm2 += alfa * alfa + oldb * oldb + beta * beta;
  double istop = 0.0;
  if (itn == 1.0 && beta / beta1 <= 10.0 * eps) istop = -1.0;
  const double delta = cs * dbar + sn * alfa;
  const double gbar = sn * dbar - cs * alfa;
  epsln = sn * beta;
  dbar = -cs * beta;
  const double root = hypot(gbar, dbar);
  const double gamma = fmax(hypot(gbar, beta), eps);
  cs = gbar / gamma;
  sn = beta / gamma;
  phibar = sn * phibar;
  const double ynorm = sqrt(xsq);
  gmax = fmax(gmax, gamma);
  gmin = fmin(gmin, gamma);
  const double z = rhs1 / gamma;
  rhs1 = rhs2 - delta * z;
  rhs2 = -epsln * z;
  const double Anorm = sqrt(tnorm2);
  const double epsx = Anorm * ynorm * eps;
  const double rnorm = phibar;
  const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
  const double test2 = Anorm == 0.0 ? INFINITY : root / Anorm;
  const double Acond = gmax / gmin;
  if (istop == 0.0) {
    const double t1 = 1.0 + test1, t2 = 1.0 + test2;
    if (t2 <= 1.0) istop = 2.0;
    if (t1 <= 1.0) istop = 1.0;
    if (itn >= maxiter) istop = 6.0;
    if (Acond >= 0.1 / eps) istop = 4.0;
    if (epsx >= beta1) istop = 3.0;
    if (test2 <= rtol) istop = 2.0;
    if (test1 <= rtol) istop = 1.0;
  }
  st[0] = istop; st[1] = itn; st[2] = oldb; st[3] = beta; st[4] = dbar; st[5] = epsln; st[6] = phibar;
  st[7] = rhs1; st[8] = rhs2; st[9] = tnorm2; st[10] = gmax; st[11] = gmin; st[12] = cs; st[13] = sn;
}

// ---------------------------------------------------------------------------------------
// Blocked right-looking Cholesky with the right-hand side carried as an extra block row.
//
// Storage: `aug` is ((nb+1)*64) x ld row-major, ld = nb*64.  Rows < nb*64 hold the SPD
// matrix (lower triangle is read; order n padded to nb*64 with an identity block), block
// row nb holds the right-hand side in its first row and zeros below.  Because
// [A b; b^T .] = [L 0; y^T .][L^T y; 0 .] the forward substitution y = L^-1 b falls out of
// the panel solves of that extra row.  Per 64-column panel j:
//   chol_panel_kernel : L_jj = chol(A_jj) and L_ij = A_ij L_jj^-T for every block row i > j, in
//                       one launch (the diagonal block is factorised by every workgroup; L_jj is
//                       stored beside the matrix, A_jj itself stays as it was);
//   chol_syrk_kernel  : A_ik -= L_i,panel * L_k,panel^T on the lower trailing blocks (MFMA).
// Panels are applied to the trailing matrix two at a time (K = 128) so that the trailing
// matrix is swept half as often; the block column in between gets a narrow K = 64 update
// inside the panel kernel.  The backward substitution x = L^-T y walks the block rows upward,
// four per launch, with the inverses of the diagonal blocks from chol_inv_kernel.
// ---------------------------------------------------------------------------------------
#define CHB 64

// 1/sqrt(d) from the hardware estimate plus two Newton steps (full double precision), and
// sqrt(d) = d * rsqrt(d) with one correction; much shorter dependency chain than sqrt + divide.
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

// Uniform broadcast of one lane's double through two scalar lane reads (no LDS round trip).
__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// Panel j in one launch, blocked by 16 columns.  The 64 x 64 diagonal block T (every workgroup
// factorises its own copy: no launch, no global round trip between factor and solve) and the
// workgroup's own block row B (block row i = j + blockIdx.x; workgroup 0 only stores the
// factor) sit in LDS.  Work items per 16-column sub-panel kb (P = T[:, kb], X = B[:, kb]):
//   F(kb)   wave 0 : right-looking factorisation of the sub-panel, one matrix row per lane with
//                    its 16 entries in registers.  The column loop carries only the reciprocal
//                    pivot (A[r][cc] -= u_r u_cc / d, pivot column broadcast by scalar lane
//                    reads); the 1/sqrt(d) scaling of the 16 columns follows the loop.
//   S(kb)   wave 1 : the same column sweep on the rows of B (X L^T = B), coefficients read from
//                    T as LDS broadcasts.
//   TU(kb)         : T[:, right] -= P P^T  (MFMA, K = 16);  BU(kb): B[:, right] -= X P^T.
// Only the tile column that the next sub-panel needs (TUcrit, BUcrit) sits between two
// factorisation phases; S(kb-1) and the remaining tiles run beside F(kb):
//   phase A_k : wave 0 F(k) | wave 1 S(k-1) | waves 2,3 TUrest(k-1), BUrest(k-2)
//   phase B_k : all waves   TUcrit(k), BUcrit(k-1)
// With `fuse` set the pending update of block column j by the previous panel (K = 64, MFMA) is
// applied while the blocks are loaded, so that panels can be applied to the rest of the
// trailing matrix two at a time.
typedef double (*chol_lds_t)[CHB + 1];

#ifdef TTK_DEBUG_HOOKS
__device__ long long g_dbg[64];
#define DBG(k) if (blockIdx.x == 1 && tid == 0 && (j == 40 || j == 41)) g_dbg[(j - 40) * 16 + (k)] = wall_clock64();
#else
#define DBG(k)
#endif

// dst tile (tr, tc) -= rows tr of src-panel (columns pc..pc+15 of A) times rows tc of T's panel
__device__ __forceinline__ void chol_tile_update(chol_lds_t dst, chol_lds_t lhs, chol_lds_t T, int tr, int tc, int pc,
                                                 int li, int kq) {
  d4 acc;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = dst[tr * 16 + kq + 4 * t][tc * 16 + li];
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 4) acc = MFMA_F64(-lhs[tr * 16 + li][pc + k0 + kq], T[tc * 16 + li][pc + k0 + kq], acc);
#pragma unroll
  for (int t = 0; t < 4; ++t) dst[tr * 16 + kq + 4 * t][tc * 16 + li] = acc[t];
}

__device__ __forceinline__ double rcp_f64(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(y, fma(-d, y, 1.0), y);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
}

__global__ __launch_bounds__(256, 2) void chol_panel_kernel(double* __restrict__ aug, int ld, int j, int fuse,
                                                            int* __restrict__ info, double* __restrict__ Lfac) {
  __shared__ double T[CHB][CHB + 1];
  __shared__ double B[CHB][CHB + 1];
  __shared__ double S0[CHB][CHB + 1];
  __shared__ double S1[CHB][CHB + 1];
  __shared__ double rsd[CHB];
  __shared__ double piv[16];
  __shared__ double colb[2][CHB];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int i = j + blockIdx.x;
  const bool own = blockIdx.x != 0;
  DBG(0)
  double* diag = aug + ((int64_t)j * CHB) * ld + (int64_t)j * CHB;
  double* blk = aug + ((int64_t)i * CHB) * ld + (int64_t)j * CHB;   // == diag for workgroup 0
  if (fuse) {
    // S0[kk][r] = L_j,j-1[r][kk],  S1[kk][r] = L_i,j-1[r][kk]; all loads unconditional and in flight together
    const double* Pd = diag - CHB;
    const double* Pi = blk - CHB;
    double pd[16], pi[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      pd[e] = Pd[(int64_t)(idx >> 6) * ld + (idx & 63)];
      pi[e] = Pi[(int64_t)(idx >> 6) * ld + (idx & 63)];
    }
    d4 ud[4], ui[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = wave * 16 + kq + 4 * u, col = t * 16 + li;
        ud[t][u] = diag[(int64_t)row * ld + col];
        ui[t][u] = blk[(int64_t)row * ld + col];
      }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      S0[idx & 63][idx >> 6] = pd[e];
      S1[idx & 63][idx >> 6] = pi[e];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t * 16 + li > wave * 16 + kq + 4 * u) ud[t][u] = 0.0;   // only the lower triangle of A_jj is kept
    __syncthreads();
#pragma unroll 4
    for (int k0 = 0; k0 < CHB; k0 += 4) {
      const double ad = -S0[k0 + kq][wave * 16 + li];
      const double ai = -S1[k0 + kq][wave * 16 + li];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = S0[k0 + kq][t * 16 + li];
        ud[t] = MFMA_F64(ad, b, ud[t]);
        ui[t] = MFMA_F64(ai, b, ui[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        T[wave * 16 + kq + 4 * u][t * 16 + li] = ud[t][u];
        B[wave * 16 + kq + 4 * u][t * 16 + li] = ui[t][u];
      }
  } else {
    double vd[16], vb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      vd[e] = diag[(int64_t)(idx >> 6) * ld + (idx & 63)];
      vb[e] = blk[(int64_t)(idx >> 6) * ld + (idx & 63)];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      const int rr_ = idx >> 6, cc = idx & 63;
      T[rr_][cc] = (cc <= rr_) ? vd[e] : 0.0;
      B[rr_][cc] = vb[e];
    }
  }
  __syncthreads();
  DBG(1)
  bool bad = false;
  int badcol = 0;
#pragma unroll 1
  for (int ph = 0; ph <= 4; ++ph) {
    const int c0 = ph * 16;
    // ---------------- phase A
    if (wave == 0) {
      if (ph < 4) {   // F(ph)
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = T[lane][c0 + c];
        // The pivots are parked in LDS for the vectorised 1/sqrt after the loop.  A non-positive or
        // NaN pivot is not repaired here: what follows it is garbage, `info` reports it and the
        // caller discards the result.
        double d = readlane_f64(a[0], c0);
        double rc = rcp_f64(d);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (lane == 0) piv[c] = d;
          // The column goes to LDS for the bulk of the updates (uniform-address reads: one
          // instruction per value instead of two scalar lane reads); a wave's LDS accesses execute
          // in order, the empty asm only keeps the compiler from moving the reads above the write.
          colb[c & 1][lane] = a[c];
          asm volatile("" ::: "memory");
          const double t = -a[c] * rc;
          if (c + 1 < 16) {   // next pivot first, by lane reads: its reciprocal chain runs under the other updates
            a[c + 1] = fma(t, readlane_f64(a[c], c0 + c + 1), a[c + 1]);
            d = readlane_f64(a[c + 1], c0 + c + 1);
            rc = rcp_f64(d);
          }
#pragma unroll
          for (int cc = c + 2; cc < 16; ++cc) a[cc] = fma(t, colb[c & 1][c0 + cc], a[cc]);
        }
        const double dv = piv[lane & 15];
        double rsv, sqv;
        rsqrt_sqrt(dv, rsv, sqv);
        const unsigned long long badmask = __ballot(!(dv > 0.0)) & 0xffffull;
        if (badmask && !bad) {
          bad = true;
          badcol = c0 + __builtin_ctzll(badmask);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] *= readlane_f64(rsv, c);
        if (lane < 16) rsd[c0 + lane] = rsv;
        if (lane >= c0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) T[lane][c0 + c] = a[c];
        }
      }
    } else if (wave == 1) {
      if (own && ph >= 1) {   // S(ph-1)
        const int s0 = c0 - 16;
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = B[lane][s0 + c];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          a[c] *= rsd[s0 + c];
#pragma unroll
          for (int cc = c + 1; cc < 16; ++cc) a[cc] = fma(-a[c], T[s0 + cc][s0 + c], a[cc]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) B[lane][s0 + c] = a[c];
      }
    } else {
      int n = wave - 2;   // tiles dealt alternately to waves 2 and 3
      if (ph >= 1) {      // TUrest(ph-1): lower tiles with column >= ph+1
        for (int tj = ph + 1; tj < 4; ++tj)
          for (int ti = tj; ti < 4; ++ti, ++n)
            if (!(n & 1)) chol_tile_update(T, T, T, ti, tj, c0 - 16, li, kq);
      }
      if (own && ph >= 2) {   // BUrest(ph-2): tile columns >= ph
        for (int tc = ph; tc < 4; ++tc)
          for (int tr = 0; tr < 4; ++tr, ++n)
            if (!(n & 1)) chol_tile_update(B, B, T, tr, tc, c0 - 32, li, kq);
      }
    }
    __syncthreads();
    DBG(2 + 2 * ph)
    if (ph == 4) break;
    // ---------------- phase B: TUcrit(ph) = tiles (ti, ph+1); BUcrit(ph-1) = tiles (tr, ph)
    {
      int n = wave;
      if (ph < 3)
        for (int ti = ph + 1; ti < 4; ++ti, ++n)
          if (!(n & 3)) chol_tile_update(T, T, T, ti, ph + 1, c0, li, kq);
      if (own && ph >= 1)
        for (int tr = 0; tr < 4; ++tr, ++n)
          if (!(n & 3)) chol_tile_update(B, B, T, tr, ph, c0 - 16, li, kq);
    }
    __syncthreads();
    DBG(3 + 2 * ph)
  }
  if (!own) {
    if (wave == 0 && bad && lane == 0 && *info == 0) *info = j * CHB + badcol + 1;
    // The factor of the diagonal block goes to its own buffer, never over A_jj: the other workgroups
    // of this launch read A_jj whenever they happen to start.
    double* Ld = Lfac + (int64_t)j * CHB * CHB;
    for (int idx = tid; idx < CHB * CHB; idx += 256) {
      const int rr_ = idx >> 6, cc = idx & 63;
      Ld[idx] = (cc <= rr_) ? T[rr_][cc] : 0.0;
    }
  } else {
    for (int idx = tid; idx < CHB * CHB; idx += 256) {
      const int rr_ = idx >> 6, cc = idx & 63;
      blk[(int64_t)rr_ * ld + cc] = B[rr_][cc];
    }
  }
  DBG(11)
}

// invL[j] = inverse of the lower-triangular diagonal block L_jj, all blocks in one launch
// (needed only by the backward substitution).  L_jj is where the panel kernel left it: in
// invL[j] itself, which this kernel overwrites after it has the block in LDS.  One column per 4 lanes, no barrier in the
// column loop: lane q writes and re-reads exactly the entries Z[i][col] with i % 4 == q.
__global__ __launch_bounds__(256) void chol_inv_kernel(double* __restrict__ invL_all) {
  __shared__ double T[CHB][CHB + 1];
  __shared__ double Z[CHB][CHB + 1];
  const int tid = threadIdx.x;
  const int j = blockIdx.x;
  const double* blk = invL_all + (int64_t)j * CHB * CHB;
  for (int idx = tid; idx < CHB * CHB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    T[r][c] = (c <= r) ? blk[idx] : 0.0;
    Z[r][c] = 0.0;
  }
  __syncthreads();
  const int col = tid >> 2, q = tid & 3;
  for (int i = col; i < CHB; ++i) {
    double s = 0.0;
    for (int jj = col + ((q - col) & 3); jj < i; jj += 4) s += T[i][jj] * Z[jj][col];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const double z = (i == col) ? 1.0 / T[i][i] : -s / T[i][i];
    if ((i & 3) == q) Z[i][col] = z;
  }
  __syncthreads();
  double* out = invL_all + (int64_t)j * CHB * CHB;
  for (int idx = tid; idx < CHB * CHB; idx += 256) out[idx] = Z[idx >> 6][idx & 63];
}

// Trailing update  C[r][c] -= sum_{kk < K} P[r][kk] P[c][kk]  with P the K panel columns that
// start at column pc0, on the lower TS x TS tiles of the region that starts at row/column o0
// (a multiple of 64).  4 waves as 2x2, each a (TS/2)^2 wave tile of 16x16 MFMA tiles; the
// accumulators start at zero and absorb the C tile underneath the first k-chunks, the row
// operand is negated, so the result is stored without a read-modify-write tail.
// The panel rows of the tile are staged k-major through LDS in chunks of 16 columns, the next
// chunk travelling in registers while the current one is multiplied.
// Tile order: 4x4 super-tiles; blocks b, b+8, b+16, ... (the blocks that share an XCD under
// round-robin dispatch) walk the 16 tiles of one super-tile, so its 2 x 4 x TS panel rows
// are fetched into that XCD's L2 once per super-tile.  Placement affects speed only.
// Measured at order 6144, K = 128: 64x64 tiles (4 workgroups per CU) beat 128x128 tiles
// (2 per CU) at every trailing size - 109 vs 130 us for 6016 rows, 11 vs 26 us for 1024 -
// because work arrives in finer units (a 128x128 round of 512 slots is 43 us long) and
// more co-resident waves cover the staging barriers.
#define SYK 16
#define SYS 4
#define SYRK_SMALL_TILES (1 << 30)   // measured: the 64x64 tiles win at every trailing size up to 6144
// NT = MFMA tiles per wave and dimension: 4 -> 128x128 workgroup tiles, 2 -> 64x64 (finer load
// balance and a shorter single-tile latency for small trailing matrices).
template <int NT>
__global__ __launch_bounds__(256, (NT == 4 ? 2 : 4)) void chol_syrk_kernel(double* __restrict__ aug, int ld, int pc0,
                                                                             int K, int o0, int nrows, int ncols,
                                                                             int nst_rows) {
  constexpr int TS = 32 * NT;       // workgroup tile
  constexpr int WT = 16 * NT;       // wave tile
  constexpr int EPT = NT * SYK / 8; // staged doubles per thread, row operand and column operand each
  constexpr int TPR = SYK / EPT;    // threads per staged row
  __shared__ double As[SYK][TS + 1];
  __shared__ double Bs[SYK][TS + 1];
  const int b = blockIdx.x;
  int ti, tk;
  if (nst_rows <= 0) {   // one tile column: blockIdx.x is the tile row
    ti = b;
    tk = 0;
  } else {
    const int st = (b / (8 * SYS * SYS)) * 8 + (b & 7);      // super-tile (lower-triangular enumeration)
    const int within = (b >> 3) & (SYS * SYS - 1);
    int si = 0, rem = st;
    while (rem > si) {
      rem -= si + 1;
      ++si;
    }
    if (si >= nst_rows) return;
    ti = si * SYS + (within & (SYS - 1));
    tk = rem * SYS + (within / SYS);
  }
  const int r0 = o0 + ti * TS, c0 = o0 + tk * TS;
  if (ti < tk || r0 >= nrows || c0 >= ncols) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int srow = tid / TPR, sk = (tid % TPR) * EPT;
  // rows past the end are clamped, not predicated (a load under a per-element branch is
  // waited for one by one); what they feed is never stored.
  const double* Ap = aug + (int64_t)min(r0 + srow, nrows - 1) * ld + pc0 + sk;
  const double* Bp = aug + (int64_t)min(c0 + srow, ncols - 1) * ld + pc0 + sk;
  double pa[EPT], pb[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    pa[e] = Ap[e];
    pb[e] = Bp[e];
  }
  // The accumulators start at zero; the C tile is fetched in NT parts (one 16-row group of
  // every wave per k-chunk) underneath the products of the first chunks and added in, so
  // that its latency is not paid in front of the loop.
  d4 acc[NT][NT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int v = 0; v < NT; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
  const int ccol0 = c0 + wc * WT + li;
  int q = 0;
  for (int kc = 0; kc < K; kc += SYK, ++q) {
    double ct[NT][4];
    if (q < NT) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double* crow = aug + (int64_t)min(r0 + wr * WT + q * 16 + kq + 4 * t, nrows - 1) * ld;
#pragma unroll
        for (int v = 0; v < NT; ++v) ct[v][t] = crow[min(ccol0 + v * 16, ncols - 1)];
      }
    }
    if (kc) __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      As[sk + e][srow] = -pa[e];   // negated here, not at the load: nothing may wait on the prefetch early
      Bs[sk + e][srow] = pb[e];
    }
    __syncthreads();
    if (kc + SYK < K) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        pa[e] = Ap[kc + SYK + e];
        pb[e] = Bp[kc + SYK + e];
      }
    }
#pragma unroll
    for (int k0 = 0; k0 < SYK; k0 += 4) {
      double af[NT], bf[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        af[u] = As[k0 + kq][wr * WT + u * 16 + li];
        bf[u] = Bs[k0 + kq][wc * WT + u * 16 + li];
      }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int v = 0; v < NT; ++v) acc[u][v] = MFMA_F64(af[u], bf[v], acc[u][v]);
    }
    if (q < NT) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
        if (u == q) {
#pragma unroll
          for (int v = 0; v < NT; ++v)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[u][v][t] += ct[v][t];
        }
    }
  }
  for (; q < NT; ++q) {   // short K: parts the loop did not reach
#pragma unroll
    for (int u = 0; u < NT; ++u)
      if (u == q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double* crow = aug + (int64_t)min(r0 + wr * WT + q * 16 + kq + 4 * t, nrows - 1) * ld;
#pragma unroll
          for (int v = 0; v < NT; ++v) acc[u][v][t] += crow[min(ccol0 + v * 16, ncols - 1)];
        }
      }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = r0 + wr * WT + u * 16 + kq + 4 * t;
      if (row < nrows) {
#pragma unroll
        for (int v = 0; v < NT; ++v) {
          const int col = c0 + wc * WT + v * 16 + li;
          if (col < ncols) aug[(int64_t)row * ld + col] = acc[u][v][t];
        }
      }
    }
}

// Backward substitution x = L^-T y, BKG block rows per launch (jlo .. jhi, walked downward).
// Every workgroup first solves the small triangular system of the group itself
//   x_j = invL_jj^T (y_j - sum_{j < m <= jhi} L_mj^T x_m)
// (recomputed everywhere: no second launch, workgroup 0 stores x), then workgroup k < jlo
// removes the group from its own block: y_k -= sum_m L_mk^T x_m.  y is the first row of the
// extra block row of `aug`.  256 threads = 64 columns x 4 row groups of 16 rows; everything a
// thread multiplies is fetched into registers up front (unconditional, coalesced loads), so the
// serial part runs on registers and LDS reductions only.
#define BKG 4
__global__ __launch_bounds__(256) void chol_back_kernel(double* __restrict__ aug, int ld, int nb, int jlo, int jhi,
                                                        const double* __restrict__ invL_all, double* __restrict__ x,
                                                        int n) {
  __shared__ double part[BKG][4][CHB];
  __shared__ double xs[BKG][CHB];
  __shared__ double ys[BKG][CHB];
  double* y = aug + ((int64_t)nb * CHB) * ld;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int ng = jhi - jlo + 1;
  const int k = blockIdx.x;
  const bool tail = k < jlo;
  double iv[BKG][16], lg[BKG][BKG - 1][16], gl[BKG][16];
#pragma unroll
  for (int m = 0; m < BKG; ++m) {
    const int mm = m < ng ? m : ng - 1;   // clamped, never used when m >= ng
    const double* invL = invL_all + (int64_t)(jlo + mm) * CHB * CHB;
#pragma unroll
    for (int e = 0; e < 16; ++e) iv[m][e] = invL[(g * 16 + e) * CHB + c];
#pragma unroll
    for (int t = 0; t < BKG - 1; ++t) {   // rows of block jlo+m against columns of block jlo+t (t < m)
      const int tt = t < mm ? t : 0;
      const double* Lmt = aug + ((int64_t)(jlo + mm) * CHB + g * 16) * ld + (int64_t)(jlo + tt) * CHB + c;
#pragma unroll
      for (int e = 0; e < 16; ++e) lg[m][t][e] = Lmt[(int64_t)e * ld];
    }
  }
#pragma unroll
  for (int m = 0; m < BKG; ++m) {
    const int mm = m < ng ? m : ng - 1;
    const double* Lmk = aug + ((int64_t)(jlo + mm) * CHB + g * 16) * ld + (int64_t)(tail ? k : 0) * CHB + c;
#pragma unroll
    for (int e = 0; e < 16; ++e) gl[m][e] = Lmk[(int64_t)e * ld];
  }
  if (g < ng) ys[g][c] = y[(jlo + g) * CHB + c];
  __syncthreads();
#pragma unroll
  for (int m = BKG - 1; m >= 0; --m) {
    if (m < ng) {   // uniform
      double acc = 0.0;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += (g * 16 + e >= c ? iv[m][e] : 0.0) * ys[m][g * 16 + e];
      part[0][g][c] = acc;
      __syncthreads();
      if (g == 0) {
        const double v = part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
        xs[m][c] = v;
        if (k == 0 && (jlo + m) * CHB + c < n) x[(jlo + m) * CHB + c] = v;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < BKG - 1; ++t)
        if (t < m) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < 16; ++e) s += lg[m][t][e] * xs[m][g * 16 + e];
          part[t][g][c] = s;
        }
      __syncthreads();
      if (g < m) ys[g][c] -= part[g][0][c] + part[g][1][c] + part[g][2][c] + part[g][3][c];
      __syncthreads();
    }
  }
  if (tail) {
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < BKG; ++m)
      if (m < ng) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s += gl[m][e] * xs[m][g * 16 + e];
      }
    part[0][g][c] = s;
    __syncthreads();
    if (g == 0) y[k * CHB + c] -= part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
  }
}

// ---------------------------------------------------------------------------------------
// The whole factorisation as ONE launch: left-looking tile tasks pulled from device-side
// queues in an order in which every task depends only on earlier ones.
//
// Tiles are 64 x 64.  A worker task owns tiles (i, j) and (i+1, j) below the sub-diagonal
// (i = j+2, j+4, ...): it forms
//     C = A_ij - sum_{k < j} L_ik L_jk^T            (MFMA, accumulators in registers)
// as soon as the tiles L_ik, L_jk exist, then L_ij = C W_j^T with W_j = L_jj^-1 (one more MFMA
// product straight from the accumulators) and publishes L_ij.  The diagonal task of column j
// owns tiles (j, j) and (j+1, j): it accumulates both, factorises the diagonal tile with the
// 16-column in-register sub-panels of chol_panel_kernel, inverts the factor blockwise (16 x 16
// triangular inverses by substitution, the other blocks by MFMA products), finishes its second
// tile and publishes L_{j+1,j} (what the next diagonal task waits for), then W_j.  The second tile's
// last term, whose left factor a worker produces meanwhile, is applied beside the factorisation
// when it is there in time (dag_late_quarter).  Nothing waits for a kernel boundary: the
// chain of diagonal tasks runs beside the accumulation of all tiles to its right, which is where
// the flops are.  The right-hand side rides along as block row nb, as in the multi-launch path.
//
// Roles.  The factorisation of a diagonal tile is one long chain of dependent fp64 operations;
// beside waves that keep the matrix pipe busy it runs three times slower.  So the first
// workgroup to arrive on each of the first `nchain` compute units takes diagonal tasks only and
// every other workgroup that lands on such a unit leaves at once; all remaining workgroups are
// workers.  Both queues are served in column order and each task waits only for tasks that come
// earlier in the merged order (diagonal of column j, then the tiles of column j), so whatever
// has been claimed can always finish: no deadlock, whatever the grid and the placement.
//
// Hand-off between workgroups (per-XCD L2s are not coherent, L1 is never refreshed): every
// byte another workgroup will read is stored write-through (sc1) and every load of such bytes
// is an sc1 load, so no cache ever holds an old copy; after its stores every wave waits for
// vmcnt(0), the workgroup meets at a barrier, and one lane raises the row's progress word by an
// sc1 store.  rowready[i] = number of finished tiles of block row i counted from the left
// (k + 1 after tile (i, k); i + 1 once W_i is out).  Waiting lanes poll that word with relaxed
// sc1 loads; every spin is bounded and gives up through the abort word, so every wave ends.
// ---------------------------------------------------------------------------------------
#define DAG_SLOT 32                    // 4-byte words per control slot (one 128-byte line each)
#define DAG_CU_WORDS 4096              // per compute unit (8 dies x 256 ids): arrivals, role
#ifdef TTK_DAG_TRACE
// timestamps (100 MHz) of the diagonal chain, 16 per block column, behind the table of compute units
#define DAG_TRACE(col, k) \
  if (tid == 0) ((long long*)(ctl + (size_t)(nb + 2) * DAG_SLOT + DAG_CU_WORDS))[(col) * 16 + (k)] = wall_clock64();
#define DAG_TRACE_VAL(col, k, v) \
  ((long long*)(ctl + (size_t)(nb + 2) * DAG_SLOT + DAG_CU_WORDS))[(col) * 16 + (k)] = (long long)(v);
#else
#define DAG_TRACE(col, k)
#define DAG_TRACE_VAL(col, k, v)
#endif
#define DAG_SPIN_LIMIT (1u << 21)     // polls before a waiting lane gives up (seconds)
typedef unsigned int dag_u4 __attribute__((ext_vector_type(4)));

__shared__ double dagR[CHB][CHB + 1];   // in turn: diagonal tile, W
__shared__ double dagS[CHB][CHB + 1];   // in turn: staging of the accumulation, result tile on its way out
__shared__ double dagDv[4][16][17];      // inverses of the four 16 x 16 diagonal blocks of the factor
__shared__ double dag_rsd[CHB];
__shared__ double dag_piv[16];
__shared__ double dag_colb[2][CHB];
__shared__ unsigned dag_word[4];
__shared__ unsigned dag_role;

__device__ __forceinline__ unsigned dag_poll(unsigned* w) {
  return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dag_post(unsigned* w, unsigned v) {
  __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// two doubles by one 16-byte sc1 load / write-through store at byte offset `off` of the buffer
__device__ __forceinline__ void dag_load2(__amdgpu_buffer_rsrc_t r, unsigned off, double& a, double& b) {
  const dag_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
  a = __hiloint2double((int)v.y, (int)v.x);
  b = __hiloint2double((int)v.w, (int)v.z);
}
__device__ __forceinline__ void dag_store2(__amdgpu_buffer_rsrc_t r, unsigned off, double a, double b) {
  dag_u4 v;
  v.x = (unsigned)__double2loint(a);
  v.y = (unsigned)__double2hiint(a);
  v.z = (unsigned)__double2loint(b);
  v.w = (unsigned)__double2hiint(b);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

// Waits until block rows `ra`, `rb` and `rc` have at least `need` finished tiles.  One lane polls, the workgroup
// meets at a barrier.  Returns the progress seen (>= need), or 0 when the launch was aborted (or this wait ran
// out of patience and aborted it).  need >= 1.
__device__ __forceinline__ unsigned dag_wait(unsigned* rowready, unsigned* ctl, int ra, int rb, int rc, unsigned need,
                                             int tid) {
  const unsigned spin_limit = ctl[4];   // written by the host before the launch, constant during it
  if (tid < 64) {   // the first wave: lanes 0, 1, 2 look at one row each, one round trip per look
    unsigned a = 0, spins = 0;
    unsigned* mine = rowready + (size_t)(tid == 0 ? ra : (tid == 1 ? rb : rc)) * DAG_SLOT;
    for (;;) {
      const unsigned v = tid < 3 ? dag_poll(mine) : 0u;
      a = min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 1)),
              (unsigned)__builtin_amdgcn_readlane((int)v, 2));
      if (a >= need) break;
      if ((++spins & 63u) == 0 && (dag_poll(ctl + 1) != 0 || spins > spin_limit)) {   // uniform over the wave
        if (tid == 0) dag_post(ctl + 1, 1u);
        a = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (tid == 0) dag_word[2] = a;
  }
  __syncthreads();
  const unsigned seen = dag_word[2];
  __syncthreads();
  return seen;
}

// Accumulation over the block columns [kb0, kb1) for NQ tiles that share the column operand (block row j):
//   acc[q][u][t] -= sum_k L_j[c = 16u + 4t + kq][k] * L_qrow[q][r = 16 wave + li][k]      (C^T, see dag_finish_tile)
// Operands travel global -> registers -> LDS (k-major chunks of 16) -> MFMA fragments.  The registers hold one
// whole block column (4 chunks); while a block is multiplied the next one is already on its way, if it exists
// (else the task waits at the frontier and then fetches the four chunks together).  Returns false on abort.
template <int NQ>
__device__ __forceinline__ bool dag_accumulate(d4 (&acc)[NQ][4], __amdgpu_buffer_rsrc_t rs, int ld, int j,
                                               const int (&qrow)[NQ], int kb0, int kb1, unsigned* rowready,
                                               unsigned* ctl, int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  double (*Ps)[CHB + 1] = (double (*)[CHB + 1]) & dagS[0][0];                       // [16][65], negated
  double (*Qs)[NQ * CHB + 1] = (double (*)[NQ * CHB + 1])(&dagS[0][0] + 16 * (CHB + 1));
  if (kb0 >= kb1) return true;
  const unsigned pbase = (unsigned)(((int64_t)(j * CHB + srow) * ld + sk) * 8);
  unsigned qbase[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) qbase[q] = (unsigned)(((int64_t)(qrow[q] * CHB + srow) * ld + sk) * 8);
  double pa[4][4], qa[4][NQ][4];
  auto fetch = [&](int kb) {   // the whole block column kb: 4 chunks of 16
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned o = (unsigned)(kb * 4 + c) * 128u;
      dag_load2(rs, pbase + o, pa[c][0], pa[c][1]);
      dag_load2(rs, pbase + o + 16, pa[c][2], pa[c][3]);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        dag_load2(rs, qbase[q] + o, qa[c][q][0], qa[c][q][1]);
        dag_load2(rs, qbase[q] + o + 16, qa[c][q][2], qa[c][q][3]);
      }
    }
  };
  unsigned avail = 0;
  bool have = false;
  bool first = true;
  for (int kb = kb0; kb < kb1; ++kb) {
    if (!have) {
      if ((unsigned)kb >= avail) {
        avail = dag_wait(rowready, ctl, j, qrow[0], qrow[NQ - 1], (unsigned)kb + 1u, tid);
        if (!avail) return false;
      }
      fetch(kb);
    }
    const bool pre = kb + 1 < kb1 && (unsigned)(kb + 1) < avail;
    // Two copies of the block body, with and without the prefetch: inside each the loads are unconditional,
    // so the compiler can count how many younger loads may still be in flight when a chunk's registers are
    // needed, instead of waiting for all of them.
    auto body = [&](auto prefetch) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!first) __syncthreads();   // the fragments of the chunk before have been read
        first = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Ps[sk + e][srow] = -pa[c][e];
#pragma unroll
          for (int q = 0; q < NQ; ++q) Qs[sk + e][q * CHB + srow] = qa[c][q][e];
        }
        __syncthreads();
        if (decltype(prefetch)::value) {   // this chunk's registers are free: the same chunk of the next block column
          const unsigned o = (unsigned)((kb + 1) * 4 + c) * 128u;
          dag_load2(rs, pbase + o, pa[c][0], pa[c][1]);
          dag_load2(rs, pbase + o + 16, pa[c][2], pa[c][3]);
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            dag_load2(rs, qbase[q] + o, qa[c][q][0], qa[c][q][1]);
            dag_load2(rs, qbase[q] + o + 16, qa[c][q][2], qa[c][q][3]);
          }
        }
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
          double af[4], bf[NQ];
#pragma unroll
          for (int u = 0; u < 4; ++u) af[u] = Ps[k0 + kq][16 * u + li];
#pragma unroll
          for (int q = 0; q < NQ; ++q) bf[q] = Qs[k0 + kq][q * CHB + 16 * wave + li];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[q][u] = MFMA_F64(af[u], bf[q], acc[q][u]);
        }
      }
    };
    if (pre)
      body(std::true_type{});
    else
      body(std::false_type{});
    have = pre;
  }
  __syncthreads();   // the staging area is free again
  return true;
}

// One block column (64 values of k) of one tile, for the two places where the diagonal chain waits for a
// single late term: dag_block_fetch brings both operands into registers (all loads in flight together),
// dag_block_apply<NC> multiplies them in, NC chunks of 16 per LDS fill (column operand k-major in the first
// 16 NC rows of dagS, row operand k-major in Qbuf, which is dagR while that is free or the upper half of dagS).
struct DagBlock {
  double p[4][4], q[4][4];
};
__device__ __forceinline__ void dag_block_fetch(DagBlock& b, __amdgpu_buffer_rsrc_t rs, int ld, int j, int qrow, int kb,
                                                int tid) {
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  const unsigned pbase = (unsigned)(((int64_t)(j * CHB + srow) * ld + kb * CHB + sk) * 8);
  const unsigned qbase = (unsigned)(((int64_t)(qrow * CHB + srow) * ld + kb * CHB + sk) * 8);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dag_load2(rs, pbase + c * 128, b.p[c][0], b.p[c][1]);
    dag_load2(rs, pbase + c * 128 + 16, b.p[c][2], b.p[c][3]);
    dag_load2(rs, qbase + c * 128, b.q[c][0], b.q[c][1]);
    dag_load2(rs, qbase + c * 128 + 16, b.q[c][2], b.q[c][3]);
  }
}
template <int NC>
__device__ __forceinline__ void dag_block_apply(const DagBlock& b, d4 (&acc)[4], double (*Qbuf)[CHB + 1], int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int srow = tid >> 2, sk = (tid & 3) * 4;
#pragma unroll
  for (int c0 = 0; c0 < 4; c0 += NC) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dagS[16 * c + sk + e][srow] = -b.p[c0 + c][e];
        Qbuf[16 * c + sk + e][srow] = b.q[c0 + c][e];
      }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 16 * NC; k0 += 4) {
      double af[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) af[u] = dagS[k0 + kq][16 * u + li];
      const double bf = Qbuf[k0 + kq][16 * wave + li];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = MFMA_F64(af[u], bf, acc[u]);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ double dag_w(int a, int b, int r, int c);
// L = C W^T for the tile whose C^T sits in `c` (accumulator layout of dag_accumulate).  W is read from dagR,
// row-major (PACKED: as a worker loaded it) or in the layout dag_invert leaves it in; the result goes through
// dagS to block (irow, j) of the matrix by coalesced write-through stores.  dagS must be free on entry; it may
// be reused after the next barrier.
template <bool PACKED>
__device__ __forceinline__ void dag_finish_tile(const d4 (&c)[4], __amdgpu_buffer_rsrc_t rs, int ld, int irow, int j,
                                                int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  d4 o[4];
#pragma unroll
  for (int up = 0; up < 4; ++up) {
    o[up] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u <= up; ++u)   // W is lower triangular: blocks right of the diagonal are zero
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double w = PACKED ? dagR[16 * up + li][16 * u + 4 * t + kq] : dag_w(up, u, li, 4 * t + kq);
        o[up] = MFMA_F64(w, c[u][t], o[up]);
      }
  }
#pragma unroll
  for (int up = 0; up < 4; ++up)
#pragma unroll
    for (int t = 0; t < 4; ++t) dagS[16 * wave + li][16 * up + 4 * t + kq] = o[up][t];
  __syncthreads();
  const unsigned base = (unsigned)(((int64_t)irow * CHB * ld + (int64_t)j * CHB) * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int id = tid + 256 * e;
    const int r = id >> 5, pc = (id & 31) * 2;
    dag_store2(rs, base + (unsigned)(((int64_t)r * ld + pc) * 8), dagS[r][pc], dagS[r][pc + 1]);
  }
}

// One wave: right-looking factorisation of the 16-column sub-panel at c0 of the tile in dagR, one matrix row
// per lane (the F step of chol_panel_kernel).  Returns 0, or 1 + the column of the first non-positive pivot.
__device__ __forceinline__ int dag_f16(int c0, int lane) {
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = dagR[lane][c0 + c];
  double d = readlane_f64(a[0], c0);
  double rc = rcp_f64(d);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (lane == 0) dag_piv[c] = d;
    dag_colb[c & 1][lane] = a[c];
    asm volatile("" ::: "memory");
    const double t = -a[c] * rc;
    if (c + 1 < 16) {
      a[c + 1] = fma(t, readlane_f64(a[c], c0 + c + 1), a[c + 1]);
      d = readlane_f64(a[c + 1], c0 + c + 1);
      rc = rcp_f64(d);
    }
#pragma unroll
    for (int cc = c + 2; cc < 16; ++cc) a[cc] = fma(t, dag_colb[c & 1][c0 + cc], a[cc]);
  }
  const double dv = dag_piv[lane & 15];
  double rsv, sqv;
  rsqrt_sqrt(dv, rsv, sqv);
  const unsigned long long badmask = __ballot(!(dv > 0.0)) & 0xffffull;
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] *= readlane_f64(rsv, c);
  if (lane < 16) dag_rsd[c0 + lane] = rsv;
  if (lane >= c0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) dagR[lane][c0 + c] = a[c];
  }
  return badmask ? c0 + __builtin_ctzll(badmask) + 1 : 0;
}

// One wave: inverse of the 16 x 16 lower-triangular block at s0 by forward substitution, one column per lane;
// row r of the block is fetched while row r-1 is summed (the empty asm keeps the compiler from issuing all
// 120 loads at once).  The inverse goes to dagDv as a full block (zeros above the diagonal).
__device__ __forceinline__ void dag_dinv16(int s0, int lane) {
  const int cc = lane & 15;
  double z[16], cur[16], nxt[16];
  z[0] = cc == 0 ? dag_rsd[s0] : 0.0;
  cur[0] = dagR[s0 + 1][s0];
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    if (r + 1 < 16) {
#pragma unroll
      for (int k = 0; k <= r; ++k) nxt[k] = dagR[s0 + r + 1][s0 + k];
    }
    double sa = 0.0, sb = 0.0;   // two chains: half the dependent latency
#pragma unroll
    for (int k = 0; k < r; ++k) {
      if (k & 1)
        sb = fma(cur[k], z[k], sb);
      else
        sa = fma(cur[k], z[k], sa);
    }
    const double rr_ = dag_rsd[s0 + r];
    z[r] = (r == cc) ? rr_ : (r > cc ? -(sa + sb) * rr_ : 0.0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k <= r; ++k) cur[k] = nxt[k];
  }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dagDv[s0 >> 4][r][cc] = z[r];
  }
}

// The second tile's last term in quarters.  C2 = A_{j+1,j} - sum_k L_{j+1,k} L_{jk}^T misses its term k = j-1
// until a worker has published L_{j+1,j-1}, which happens while the diagonal tile is being factorised.  C2 waits
// in dagS (plain row-major) meanwhile, and the term is applied to that copy in four quarters of 16 values of k,
// always in the same order and with the same assignment of k to lanes, whoever applies them: waves 2 and 3
// beside the factorisation and the inversion when the flag is already up (NG = 2: 32 rows per wave), all four
// waves after the inversion for whatever is left (NG = 1: 16 rows per wave).  The bits of the result therefore
// do not depend on when the worker was done.  Operands come straight from memory in MFMA operand layout:
// lane (li, kq) takes k = 8 grp + 2 kq, + 1 of a group of eight by one 16-byte sc1 load.
__shared__ unsigned dag_late;   // quarters released: 0, or 4 once the worker's tile has been seen
template <int NG>
__device__ __forceinline__ void dag_late_quarter(__amdgpu_buffer_rsrc_t rs, int ld, int j, int qtr, int row0,
                                                 int lane) {
  const int li = lane & 15, kq = lane >> 4;
  const unsigned col = (unsigned)(((j - 1) * CHB + qtr * 16 + 2 * kq) * 8);
  double pa[4][2][2], qa[NG][2][2];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      dag_load2(rs, (unsigned)((int64_t)(j * CHB + 16 * u + li) * ld * 8) + col + grp * 64, pa[u][grp][0],
                pa[u][grp][1]);
#pragma unroll
    for (int g = 0; g < NG; ++g)
      dag_load2(rs, (unsigned)((int64_t)((j + 1) * CHB + row0 + 16 * g + li) * ld * 8) + col + grp * 64,
                qa[g][grp][0], qa[g][grp][1]);
  }
  d4 acc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[g][u][t] = dagS[row0 + 16 * g + li][16 * u + 4 * t + kq];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[g][u] = MFMA_F64(-pa[u][grp][e], qa[g][grp][e], acc[g][u]);
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) dagS[row0 + 16 * g + li][16 * u + 4 * t + kq] = acc[g][u][t];
}
// One lane: releases the four quarters once the worker's tile is out (a single look, no wait).
__device__ __forceinline__ void dag_late_release(unsigned* rowready, int j) {
  if (dag_late == 0u && dag_poll(rowready + (size_t)(j + 1) * DAG_SLOT) >= (unsigned)j) dag_late = 4u;
}

// All four waves: W = L^-1 of the factorised tile in dagR, by block distance from the diagonal:
// W_ab = -Dinv_a * sum_{m=b}^{a-1} L_am W_mb, W_bb = Dinv_b.  Block W_ab, a > b (and the sum before it), is
// kept untransposed in the UPPER block (b, a) of dagR: W[16a + r][16b + c] = dagR[16b + r][16a + c]; the
// diagonal blocks stay in dagDv.  dag_w reads W from that layout.
__device__ __forceinline__ double dag_w(int a, int b, int r, int c) {   // W[16a + r][16b + c], a >= b
  return a == b ? dagDv[a][r][c] : dagR[16 * b + r][16 * a + c];
}
// Q_ab = sum_{m=b}^{a-1} L_am W_mb into the home of block (a, b) (one wave).
template <int A, int B>
__device__ __forceinline__ void dag_inv_sum(int li, int kq) {
  d4 q = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = B; m < A; ++m)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double x = dagR[16 * A + li][16 * m + 4 * s + kq];                                       // L_am[li][k]
      const double y = m == B ? dagDv[B][4 * s + kq][li] : dagR[16 * B + 4 * s + kq][16 * m + li];   // W_mb[k][li]
      q = MFMA_F64(x, y, q);
    }
#pragma unroll
  for (int t = 0; t < 4; ++t) dagR[16 * B + kq + 4 * t][16 * A + li] = q[t];
}
// W_ab = -Dinv_a Q_ab in place (one wave; its LDS accesses execute in order, so the block can be written and
// read back without a barrier when the same wave formed Q just before).
template <int A, int B>
__device__ __forceinline__ void dag_inv_fin(int li, int kq) {
  asm volatile("" ::: "memory");
  d4 w = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) w = MFMA_F64(-dagDv[A][li][4 * s + kq], dagR[16 * B + 4 * s + kq][16 * A + li], w);
#pragma unroll
  for (int t = 0; t < 4; ++t) dagR[16 * B + kq + 4 * t][16 * A + li] = w[t];
}
// The inverse of the last diagonal block is not there yet when this starts (the factorisation has just
// finished its last sub-panel): wave 1 forms it in the first phase, beside the blocks that do not need it.
// Waves 2 and 3 have little to do in the first phase: with lq >= 0 they first apply quarter lq of the second
// tile's last term (dag_late_quarter).
__device__ __forceinline__ void dag_invert(int tid, int lq, __amdgpu_buffer_rsrc_t rs, int ld, int j) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  if (wave == 0) {
    dag_inv_sum<1, 0>(li, kq);
    dag_inv_fin<1, 0>(li, kq);
  } else if (wave == 1) {
    dag_dinv16(48, lane);
  } else if (wave == 2) {
    if (lq >= 0) dag_late_quarter<2>(rs, ld, j, lq, 0, lane);
    dag_inv_sum<2, 1>(li, kq);
    dag_inv_fin<2, 1>(li, kq);
  } else {
    if (lq >= 0) dag_late_quarter<2>(rs, ld, j, lq, 32, lane);
    dag_inv_sum<3, 2>(li, kq);
  }
  __syncthreads();
  if (wave == 0) {
    dag_inv_fin<3, 2>(li, kq);
  } else if (wave == 1) {
    dag_inv_sum<2, 0>(li, kq);
    dag_inv_fin<2, 0>(li, kq);
  } else if (wave == 2) {
    dag_inv_sum<3, 1>(li, kq);   // L_31 Dinv_1 + L_32 W_21
    dag_inv_fin<3, 1>(li, kq);
  }
  __syncthreads();
  if (wave == 0) {
    dag_inv_sum<3, 0>(li, kq);   // L_30 Dinv_0 + L_31 W_10 + L_32 W_20
    dag_inv_fin<3, 0>(li, kq);
  }
  __syncthreads();
}

// The factorisation of the tile in dagR by 16-column sub-panels (see chol_panel_kernel): wave 0 factorises a
// sub-panel while wave 1 inverts the diagonal 16 x 16 block of the one before (the last block is left to
// dag_invert) and waves 2, 3 update the tiles
// further right; the tile column the next sub-panel needs is updated by all waves in between.  Returns 0, or
// 1 + the column of the first non-positive pivot (meaningful in wave 0).
__device__ __forceinline__ int dag_factor(int j, int nb, unsigned* ctl, __amdgpu_buffer_rsrc_t rs, int ld,
                                          unsigned* rowready, int& late_done, int tid) {
  (void)nb; (void)ctl;   // named by the trace macro only
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kq = lane >> 4;
  chol_lds_t R = dagR;
  int badcol = 0;
  int avail = 0, done = 0;   // quarters of the second tile's last term: released / applied (the same in every wave)
#pragma unroll 1
  for (int ph = 0; ph < 4; ++ph) {
    const int c0 = ph * 16;
    if (wave == 0) {
      const int bc = dag_f16(c0, lane);
      if (bc && !badcol) badcol = bc;
    } else if (wave == 1) {
      // the only writer of dag_late, and only here: the other waves read it between the two barriers below
      if (j > 0 && lane == 0) {
        const unsigned before = dag_late;
        dag_late_release(rowready, j);
        if (ph == 0) {   // nothing else to do in this phase: two more looks
          dag_late_release(rowready, j);
          dag_late_release(rowready, j);
        }
        if (before == 0u && dag_late != 0u) { DAG_TRACE_VAL(j, 10, wall_clock64()) }
      }
      if (ph >= 1) dag_dinv16(c0 - 16, lane);
    } else {
      int n = wave - 2;
      if (ph >= 1) {
        for (int tj = ph + 1; tj < 4; ++tj)
          for (int ti = tj; ti < 4; ++ti, ++n)
            if (!(n & 1)) chol_tile_update(R, R, R, ti, tj, c0 - 16, li, kq);
      }
      if (avail > done) dag_late_quarter<2>(rs, ld, j, done, 32 * (wave - 2), lane);
    }
    if (avail > done) ++done;
    __syncthreads();
    avail = (int)dag_late;
    if (ph == 0) { DAG_TRACE(j, 12) }
    if (ph == 1) { DAG_TRACE(j, 14) }
    if (ph == 2) { DAG_TRACE(j, 15) }
    if (ph == 3) break;
    {
      int n = wave;
      for (int ti = ph + 1; ti < 4; ++ti, ++n)
        if (!(n & 3)) chol_tile_update(R, R, R, ti, ph + 1, c0, li, kq);
    }
    __syncthreads();
    if (ph == 0) { DAG_TRACE(j, 13) }
  }
  // one more quarter beside the first phase of the inversion (see dag_invert)
  late_done = done | (avail > done ? 0x100 : 0);
  return badcol;
}

// Where the tiles of A come from.  With G set the factorisation reads A straight from the compressed Gram,
//   A[(l,pp,r)][(m,q,t)] = G[a(l,m)][b(pp,q)][c(r,t)] / scale + ridge (row == col),   identity beyond order D,
// exactly what gram_expand_lower_kernel would have written into the storage first (same operations in the same
// order, so the same bits): the 8 D^2 / 2 bytes of the expanded lower triangle are then never written or read.
// The right-hand side (block row nb) always comes from the storage.
struct DagGram {
  const double* G;
  const double* scale;
  double ridge;
  int rl, p, rr, sym, NX, NRpad, D;
  float inv_rr, inv_p;   // 1 / rr, 1 / p: quotients of indices below 2^21 by one multiply (see dag_div)
};
// x / d for 0 <= x < 2^21 with inv = 1.0f / d: (x + 0.5) * inv is at least 0.5 / d away from every integer, the
// rounding error of the product stays below x * 2^-22 < 0.5 / d.
__device__ __forceinline__ int dag_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// The 16 entries (row, col0 + 16 u + 4 t + kq) of A that one lane's accumulators of a tile start from.  The row's
// indices are split once, the 16 loads are unconditional (indices clamped to the matrix) and in flight together;
// scale, ridge and the identity padding are applied afterwards.  `sc` = scale[0], read once per kernel.
template <bool LAZY>
__device__ __forceinline__ void dag_a_tile(d4 (&out)[4], const double* __restrict__ aug, int ld, const DagGram& gs,
                                           double sc, int row, int col0, int kq) {
  if (!LAZY || row >= ld) {   // (uniform over the tile: only the right-hand-side block row lies beyond ld)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) out[u][t] = aug[(int64_t)row * ld + col0 + 16 * u + 4 * t + kq];
    return;
  }
  const int rowc = min(row, gs.D - 1);
  const int lp = dag_div(rowc, gs.inv_rr), r = rowc - lp * gs.rr, l = dag_div(lp, gs.inv_p), pp = lp - l * gs.p;
  double g[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int colc = min(col0 + 16 * u + 4 * t + kq, gs.D - 1);
      const int mq = dag_div(colc, gs.inv_rr), tt = colc - mq * gs.rr, m = dag_div(mq, gs.inv_p), q = mq - m * gs.p;
      const int a = gs.sym ? tri_index(min(l, m), max(l, m), gs.rl) : l * gs.rl + m;
      const int b = tri_index(min(pp, q), max(pp, q), gs.p);
      const int c = tri_index(min(r, tt), max(r, tt), gs.rr);
      g[u][t] = gs.G[((int64_t)a * gs.NX + b) * gs.NRpad + c];
    }
  const bool scaled = gs.scale != nullptr;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = col0 + 16 * u + 4 * t + kq;
      double v = g[u][t];
      if (scaled) v = v / sc;
      if (row == col) v += gs.ridge;
      if (row >= gs.D || col >= gs.D) v = row == col ? 1.0 : 0.0;   // identity beyond order D
      out[u][t] = v;
    }
}

template <bool LAZY>
__global__ __launch_bounds__(256, 2) void chol_dag_kernel(double* __restrict__ aug, int ld, int nb, unsigned aug_bytes,
                                                          double* __restrict__ Winv, int* __restrict__ info,
                                                          unsigned* __restrict__ ctl, int nchain, DagGram gs) {
  chol_lds_t R = dagR;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)aug, 0, (int)aug_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)Winv, 0, nb * CHB * CHB * (int)sizeof(double), 0x00020000);
  unsigned* rowready = ctl + DAG_SLOT;
  const double sc = (LAZY && gs.scale) ? gs.scale[0] : 1.0;
  // worker tasks: column j has rows j+2 .. nb in pairs, (nb - j) / 2 tasks
  int ntasks = 0;
  for (int jj = 0; jj < nb; ++jj) ntasks += (nb - jj) / 2;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));    // HW_ID[15:8]: engine, array, unit
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));             // XCC_ID[3:0]
    unsigned* cu = ctl + (size_t)(nb + 2) * DAG_SLOT + 2 * (((xcc & 7u) << 8) | (hw & 255u));
    unsigned role = 0;
    if (atomicAdd(cu, 1u) == 0) {
      role = atomicAdd(ctl + 3, 1u) < (unsigned)nchain ? 2u : 1u;
      dag_post(cu + 1, role);
    } else {
      for (unsigned spins = 0; (role = dag_poll(cu + 1)) == 0 && spins < DAG_SPIN_LIMIT; ++spins) __builtin_amdgcn_s_sleep(2);
      if (role == 2) role = 3;
    }
    dag_role = role;
  }
  __syncthreads();
  const unsigned role = dag_role;
  if (role != 1 && role != 2) return;
  for (;;) {
    // The thread index is made opaque once per task: everything derived from it (LDS and global
    // offsets of the unrolled code below) is then computed where it is used instead of being kept
    // in registers across the whole task loop.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, kq = lane >> 4;
    if (tid == 0) {
      dag_word[0] = atomicAdd(role == 2 ? ctl + 2 : ctl, 1u);
      dag_word[1] = dag_poll(ctl + 1);
    }
    __syncthreads();
    const int task = (int)dag_word[0];
    const bool stop = dag_word[1] != 0;
    __syncthreads();
    const bool diag = role == 2;
    if (stop) {   // a wait somewhere ran out of patience: the result is not a factorisation
      if (tid == 0) atomicCAS(info, 0, -1);
      return;
    }
    if (task >= (diag ? nb : ntasks)) return;
    const int r_own = 16 * wave + li;   // row of the tile this lane's accumulators belong to
    if (!diag) {
      // ------------------------------------------------------------ tiles (i, j), (i+1, j); i = j+2, j+4, ...
      int j = 0, rem = task;
      while (rem >= (nb - j) / 2) {
        rem -= (nb - j) / 2;
        ++j;
      }
      const int i = j + 2 + 2 * rem;
      const bool two = i + 1 <= nb;
      const int i2 = two ? i + 1 : i;   // a lone last row is simply done twice
      d4 acc[2][4];
      dag_a_tile<LAZY>(acc[0], aug, ld, gs, sc, i * CHB + r_own, j * CHB, kq);
      dag_a_tile<LAZY>(acc[1], aug, ld, gs, sc, i2 * CHB + r_own, j * CHB, kq);
      const int qrow[2] = {i, i2};
      if (!dag_accumulate<2>(acc, rs, ld, j, qrow, 0, j, rowready, ctl, tid) ||
          !dag_wait(rowready, ctl, j, j, j, (unsigned)j + 1u, tid)) {   // ... and W_j
        if (tid == 0) atomicCAS(info, 0, -1);
        return;
      }
      {
        double wv[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int id = tid + 256 * e;
          dag_load2(rw, (unsigned)((j * CHB * CHB + id * 2) * 8), wv[2 * e], wv[2 * e + 1]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int id = tid + 256 * e;
          const int r = id >> 5, pc = (id & 31) * 2;
          R[r][pc] = wv[2 * e];
          R[r][pc + 1] = wv[2 * e + 1];
        }
      }
      __syncthreads();
      // each tile is published as soon as it is out: the upper one may be what the diagonal chain needs next
      dag_finish_tile<true>(acc[0], rs, ld, i, j, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) dag_post(rowready + (size_t)i * DAG_SLOT, (unsigned)(j + 1));
      if (two) {
        dag_finish_tile<true>(acc[1], rs, ld, i2, j, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) dag_post(rowready + (size_t)i2 * DAG_SLOT, (unsigned)(j + 1));
      }
      continue;
    }
    // ------------------------------------------------------------------ diagonal task of column j
    const int j = task;
    DAG_TRACE(j, 0)
    d4 acc[2][4];
    dag_a_tile<LAZY>(acc[0], aug, ld, gs, sc, j * CHB + r_own, j * CHB, kq);
    dag_a_tile<LAZY>(acc[1], aug, ld, gs, sc, (j + 1) * CHB + r_own, j * CHB, kq);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * u + 4 * t + kq > r_own) acc[0][u][t] = 0.0;   // only the lower triangle of A_jj is used
    {
      // Both tiles up to block column j-2; then the diagonal tile's last term, L_{j,j-1}, which is what the
      // chain waits for.  The second tile's last term, L_{j+1,j-1}, comes from a worker and is taken in behind
      // the factorisation.
      const int qrow[2] = {j, j + 1};
      if (!dag_accumulate<2>(acc, rs, ld, j, qrow, 0, j - 1, rowready, ctl, tid)) {
        if (tid == 0) atomicCAS(info, 0, -1);
        return;
      }
      DAG_TRACE(j, 8)
      if (j > 0) {
        if (!dag_wait(rowready, ctl, j, j, j, (unsigned)j, tid)) {
          if (tid == 0) atomicCAS(info, 0, -1);
          return;
        }
        DAG_TRACE(j, 9)
        DagBlock blk;
        dag_block_fetch(blk, rs, ld, j, j, j - 1, tid);
        dag_block_apply<4>(blk, acc[0], dagR, tid);
      }
    }
    DAG_TRACE(j, 1)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 16 * u + 4 * t + kq;
        R[r_own][c] = c <= r_own ? acc[0][u][t] : 0.0;
      }
    // The second tile's accumulators wait in dagS meanwhile: the unrolled chains of the factorisation want
    // every register they can get (parked in front of the barrier, so that the first sub-panel does not
    // share the LDS with these stores).
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) dagS[r_own][16 * u + 4 * t + kq] = acc[1][u][t];
    if (tid == 0) dag_late = 0u;
    __syncthreads();
    DAG_TRACE(j, 2)
    int late = 0;
    const int badcol = dag_factor(j, nb, ctl, rs, ld, rowready, late, tid);
    if (wave == 0 && badcol && lane == 0) atomicCAS(info, 0, j * CHB + badcol);
    DAG_TRACE(j, 3)
    int late_done = late & 0xff;
    dag_invert(tid, (late & 0x100) ? late_done : -1, rs, ld, j);
    if (late & 0x100) ++late_done;
    DAG_TRACE(j, 4)
    if (tid == 0) { DAG_TRACE_VAL(j, 11, late_done) }
    if (j > 0 && late_done < 4) {   // what is left of the second tile's last term (all of it if the worker was late)
      if (dag_late == 0u && !dag_wait(rowready, ctl, j + 1, j + 1, j + 1, (unsigned)j, tid)) {
        if (tid == 0) atomicCAS(info, 0, -1);
        return;
      }
#pragma unroll 1
      for (int q = late_done; q < 4; ++q) dag_late_quarter<1>(rs, ld, j, q, 16 * wave, lane);
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[1][u][t] = dagS[r_own][16 * u + 4 * t + kq];
    __syncthreads();   // dagS is about to be used for staging again
    DAG_TRACE(j, 5)
    // L_{j+1,j} = C W^T first: it is what the next diagonal task waits for.  W_j follows; the workers that wait
    // for it feed the diagonal chain only a whole factorisation later.
    dag_finish_tile<false>(acc[1], rs, ld, j + 1, j, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) dag_post(rowready + (size_t)(j + 1) * DAG_SLOT, (unsigned)(j + 1));
    DAG_TRACE(j, 6)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int id = tid + 256 * e;
      const int r = id >> 5, pc = (id & 31) * 2;
      const int a = r >> 4, b = pc >> 4;
      const double w0 = a >= b ? dag_w(a, b, r & 15, pc & 15) : 0.0;
      const double w1 = a >= b ? dag_w(a, b, r & 15, (pc & 15) + 1) : 0.0;
      dag_store2(rw, (unsigned)((j * CHB * CHB + id * 2) * 8), w0, w1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) dag_post(rowready + (size_t)j * DAG_SLOT, (unsigned)(j + 1));
    DAG_TRACE(j, 7)
  }
}

// Backward substitution x = L^-T y as ONE launch behind the single-launch factorisation: the group solves and
// block updates of chol_back_kernel (same operations in the same order, hence the same bits), with the kernel
// boundaries between groups replaced by progress words.  Workgroup `blockIdx.x` owns block row k = nb-1-blockIdx.x,
// i.e. workgroups are dispatched in the order in which they can finish: a workgroup only ever waits for rows of
// higher index.  It walks the groups of BKG rows from the top down to its own; for a group above its own it
// removes the group from its y_k (kept in LDS), and when every group above has been removed it publishes y_k
// (write-through stores, vmcnt(0), barrier, progress word = BACK_READY in the row's slot of the factorisation's
// control block, whose counts stay far below that value).  The matrices of a group (which depend on the
// factorisation only) are fetched before the wait for its rows.  Every spin is bounded and ends through the
// abort word like the factorisation's.
#define BACK_READY 0x40000000u
__device__ __forceinline__ double back_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void back_store(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void chol_back_chain_kernel(double* __restrict__ aug, int ld, int nb,
                                                              const double* __restrict__ invL_all, double* __restrict__ x,
                                                              int n, unsigned* __restrict__ ctl, int* __restrict__ info) {
  __shared__ double part[BKG][4][CHB];
  __shared__ double xs[BKG][CHB];
  __shared__ double ys[BKG][CHB];
  __shared__ double yk[CHB];
  __shared__ unsigned seen;
  unsigned* rowready = ctl + DAG_SLOT;
  const unsigned spin_limit = ctl[4];
  double* y = aug + ((int64_t)nb * CHB) * ld;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int k = nb - 1 - (int)blockIdx.x;
  const int tk = (nb - 1 - k) / BKG;
  if (threadIdx.x < CHB) yk[c] = y[k * CHB + c];
  if (tk == 0 && threadIdx.x == 0) dag_post(rowready + (size_t)k * DAG_SLOT, BACK_READY);   // nothing above it
  for (int t = 0; t <= tk; ++t) {
    const int jhi = nb - 1 - BKG * t;
    const int jlo = jhi - BKG + 1 > 0 ? jhi - BKG + 1 : 0;
    const int ng = jhi - jlo + 1;
    const bool tail = t < tk;
    double iv[BKG][16], lg[BKG][BKG - 1][16], gl[BKG][16];
#pragma unroll
    for (int m = 0; m < BKG; ++m) {
      const int mm = m < ng ? m : ng - 1;   // clamped, never used when m >= ng
      const double* invL = invL_all + (int64_t)(jlo + mm) * CHB * CHB;
#pragma unroll
      for (int e = 0; e < 16; ++e) iv[m][e] = invL[(g * 16 + e) * CHB + c];
#pragma unroll
      for (int u = 0; u < BKG - 1; ++u) {   // rows of block jlo+m against columns of block jlo+u (u < m)
        const int uu = u < mm ? u : 0;
        const double* Lmt = aug + ((int64_t)(jlo + mm) * CHB + g * 16) * ld + (int64_t)(jlo + uu) * CHB + c;
#pragma unroll
        for (int e = 0; e < 16; ++e) lg[m][u][e] = Lmt[(int64_t)e * ld];
      }
    }
#pragma unroll
    for (int m = 0; m < BKG; ++m) {
      const int mm = m < ng ? m : ng - 1;
      const double* Lmk = aug + ((int64_t)(jlo + mm) * CHB + g * 16) * ld + (int64_t)(tail ? k : 0) * CHB + c;
#pragma unroll
      for (int e = 0; e < 16; ++e) gl[m][e] = Lmk[(int64_t)e * ld];
    }
    if (g == 0) {   // the rows of the group are final: lane m of the first wave looks at row jlo + m, one round trip per look
      unsigned ok = 1, spins = 0;
      for (;;) {
        const unsigned a = c < ng ? dag_poll(rowready + (size_t)(jlo + c) * DAG_SLOT) : BACK_READY;
        if (__all(a >= BACK_READY)) break;
        if ((++spins & 63u) == 0 && (dag_poll(ctl + 1) != 0 || spins > spin_limit)) {   // uniform over the wave
          if (c == 0) dag_post(ctl + 1, 1u);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (c == 0) seen = ok;
    }
    __syncthreads();
    if (!seen) {
      if (threadIdx.x == 0) atomicCAS(info, 0, -1);
      return;
    }
    if (g < ng) ys[g][c] = (jlo + g == k) ? yk[c] : back_load(y + (jlo + g) * CHB + c);
    __syncthreads();
#pragma unroll
    for (int m = BKG - 1; m >= 0; --m) {
      if (m < ng) {   // uniform
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += (g * 16 + e >= c ? iv[m][e] : 0.0) * ys[m][g * 16 + e];
        part[0][g][c] = acc;
        __syncthreads();
        if (g == 0) {
          const double v = part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
          xs[m][c] = v;
          if (!tail && k == jlo && (jlo + m) * CHB + c < n) x[(jlo + m) * CHB + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BKG - 1; ++u)
          if (u < m) {
            double sacc = 0.0;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc += lg[m][u][e] * xs[m][g * 16 + e];
            part[u][g][c] = sacc;
          }
        __syncthreads();
        if (g < m) ys[g][c] -= part[g][0][c] + part[g][1][c] + part[g][2][c] + part[g][3][c];
        __syncthreads();
      }
    }
    if (tail) {
      double sacc = 0.0;
#pragma unroll
      for (int m = 0; m < BKG; ++m)
        if (m < ng) {
#pragma unroll
          for (int e = 0; e < 16; ++e) sacc += gl[m][e] * xs[m][g * 16 + e];
        }
      part[0][g][c] = sacc;
      __syncthreads();
      if (g == 0) yk[c] -= part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
      __syncthreads();
      if (t == tk - 1) {   // every group above has been removed: y_k is final
        if (g == 0) back_store(y + k * CHB + c, yk[c]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) dag_post(rowready + (size_t)k * DAG_SLOT, BACK_READY);
      }
    }
  }
}

// Gram expansion straight into the Cholesky storage: lower triangle of A/scale + ridge*I,
// identity on the padded diagonal, zero elsewhere in the padded rows.
__global__ __launch_bounds__(256) void gram_expand_lower_kernel(const double* __restrict__ G, int rl, int p, int rr,
                                                                 int sym, int NX, int NRpad,
                                                                 const double* __restrict__ scale, double ridge,
                                                                 double* __restrict__ A, int ld) {
  const int D = rl * p * rr;
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col > row || col >= ld) return;
  double v;
  if (row >= D) {
    v = (row == col) ? 1.0 : 0.0;
  } else {
    const int r = row % rr, lp = row / rr, pp = lp % p, l = lp / p;
    const int t = col % rr, mq = col / rr, q = mq % p, m = mq / p;
    const int a = sym ? tri_index(min(l, m), max(l, m), rl) : l * rl + m;
    const int b = tri_index(min(pp, q), max(pp, q), p);
    const int c = tri_index(min(r, t), max(r, t), rr);
    v = G[((int64_t)a * NX + b) * NRpad + c];
    if (scale) v = v / scale[0];
    if (row == col) v += ridge;
  }
  A[(int64_t)row * ld + col] = v;
}

// ---------------------------------------------------------------------------------------
// Householder QR of a tall-skinny core matrix (m x n, m*n small) in one workgroup, LAPACK
// conventions (dgeqr2 + dorg2r): beta = -sign(alpha) * ||x||, v = x / (alpha - beta) with
// v[0] = 1, tau = (beta - alpha) / beta, R's diagonal = beta; Q = H_1 ... H_k applied to the
// first k columns of the identity, k = min(m, n).  Same signs as torch.linalg.qr on the CPU,
// which is what fixes the gauge of the orthonormalised cores.
// LDS: A (m x n, column-major, stride m+1) and tau; one column of reflector work per step.
// ---------------------------------------------------------------------------------------
// Register variant for m <= 256, n <= 32 (every core of the headline model): column c lives in
// wave c % 4, slot c / 4, four rows per lane (row = lane + 64 e), so dot products and updates are
// register arithmetic plus one wave reduction; a reflector crosses waves through a double-buffered
// LDS vector (one barrier per column) and is kept in LDS for the Q formation, which needs no
// barrier at all (columns are independent).  Same arithmetic and sign convention as qr_kernel.
// Sums of eight per-lane values over the 64 lanes of a wave, every total on every lane.  A butterfly in which
// the lanes first split the eight sums among themselves (offsets 32, 16, 8: a lane passes on the half it gives
// up, 4 + 2 + 1 exchanges), then finish their own one (offsets 1, 2, 4 within a row of 16 lanes), then read all eight back from the lanes
// that hold them (scalar lane reads): 10 exchanges + 8 lane reads instead of 8 x 6 exchanges; fixed order.
__device__ __forceinline__ double qr_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// v of another lane of the same row of 16 by a data-parallel-primitive move (no trip through the LDS crossbar):
// 0xB1 / 0x4E = lanes xor 1 / xor 2 (quad permutations), 0x141 / 0x140 = mirror within 8 / 16 lanes, 0x128 = rotation by 8.
template <int CTRL>
__device__ __forceinline__ double qr_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, the same value (and bits) on every lane
__device__ __forceinline__ double qr_wave_sum(double v) {
  v += qr_dpp<0xB1>(v);
  v += qr_dpp<0x4E>(v);
  v += qr_dpp<0x141>(v);
  v += qr_dpp<0x140>(v);
  return (qr_readlane(v, 0) + qr_readlane(v, 16)) + (qr_readlane(v, 32) + qr_readlane(v, 48));
}
__device__ __forceinline__ void qr_wave_sum8(double (&acc)[8], int lane) {
  double a4[4], a2[2];
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double give = h5 ? acc[q] : acc[q + 4], keep = h5 ? acc[q + 4] : acc[q];
    a4[q] = keep + __shfl_xor(give, 32, 64);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double give = h4 ? a4[q] : a4[q + 2], keep = h4 ? a4[q + 2] : a4[q];
    a2[q] = keep + __shfl_xor(give, 16, 64);
  }
  double one = (h3 ? a2[1] : a2[0]) + qr_dpp<0x128>(h3 ? a2[0] : a2[1]);   // row rotation by 8 = lane xor 8
  one += qr_dpp<0xB1>(one);
  one += qr_dpp<0x4E>(one);
  one += qr_dpp<0x141>(one);
  // sum q = 4 b5 + 2 b4 + b3 sits in the lanes with those bits: read it from the first of them
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = qr_readlane(one, ((q >> 2) & 1) * 32 + ((q >> 1) & 1) * 16 + (q & 1) * 8);
}

// Where the QR kernels find element (r, c) of the matrix they factorise and put those of Q (m x k) and R (k x n):
// offset + r * row stride + c * column stride, strides of either sign.  A gauge move to the left is the QR of the
// doubly flipped transposed core (tensor/network.py:686-690): with these maps it reads the core as it lies in memory
// and writes the new core and the bond factor as they are used, no flip or transposition kernels around it.
struct QrMap {
  int64_t a_off, a_sr, a_sc, q_off, q_sr, q_sc, r_off, r_sr, r_sc;
};

#define QRS_RPL 4
#define QRS_CPW 8
__global__ __launch_bounds__(256) void qr_small_kernel(const double* __restrict__ Ain, int m, int n,
                                                        double* __restrict__ Q, double* __restrict__ Rout, QrMap mp) {
  __shared__ double V[32][256];      // reflector j: V[j][row], zero above j, one at j
  __shared__ double tau[32];
  const int k = min(m, n);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double a[QRS_CPW][QRS_RPL];
#pragma unroll
  for (int q = 0; q < QRS_CPW; ++q)
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) {
      const int c = wave + 4 * q, r = lane + 64 * e;
      const double v = Ain[mp.a_off + (int64_t)min(r, m - 1) * mp.a_sr + (int64_t)min(c, n - 1) * mp.a_sc];
      a[q][e] = (r < m && c < n) ? v : 0.0;
    }
  for (int j = 0; j < k; ++j) {
    if (wave == (j & 3)) {
      const int qj = j >> 2;
      double col[QRS_RPL];
#pragma unroll
      for (int q = 0; q < QRS_CPW; ++q)
        if (q == qj) {
#pragma unroll
          for (int e = 0; e < QRS_RPL; ++e) col[e] = a[q][e];
        }
      double part = 0.0, mine = 0.0;
#pragma unroll
      for (int e = 0; e < QRS_RPL; ++e) {
        const int r = lane + 64 * e;
        if (r > j) part += col[e] * col[e];
        if (r == j) mine = col[e];
      }
      part = qr_wave_sum(part);
      const double alpha = qr_readlane(mine, j & 63);   // the lane that holds row j
      double t = 0.0, beta = alpha, scale = 0.0;
      if (part != 0.0) {
        const double nrm = sqrt(alpha * alpha + part);
        beta = alpha >= 0.0 ? -nrm : nrm;
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
#pragma unroll
      for (int e = 0; e < QRS_RPL; ++e) {
        const int r = lane + 64 * e;
        const double v = r > j ? col[e] * scale : (r == j ? 1.0 : 0.0);
        V[j][r] = v;
        col[e] = r > j ? v : (r == j ? beta : col[e]);
      }
#pragma unroll
      for (int q = 0; q < QRS_CPW; ++q)
        if (q == qj) {
#pragma unroll
          for (int e = 0; e < QRS_RPL; ++e) a[q][e] = col[e];
        }
      if (lane == 0) tau[j] = t;
    }
    __syncthreads();
    double v[QRS_RPL];
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) v[e] = V[j][lane + 64 * e];
    const double t = tau[j];
    double acc[QRS_CPW];
#pragma unroll
    for (int q = 0; q < QRS_CPW; ++q) {
      double d = 0.0;
#pragma unroll
      for (int e = 0; e < QRS_RPL; ++e) d += v[e] * a[q][e];
      acc[q] = d;
    }
    qr_wave_sum8(acc, lane);
#pragma unroll
    for (int q = 0; q < QRS_CPW; ++q) {
      const int c = wave + 4 * q;
      if (c > j && c < n) {
        const double w = t * acc[q];
#pragma unroll
        for (int e = 0; e < QRS_RPL; ++e) a[q][e] -= v[e] * w;
      }
    }
  }
  // R = upper triangle (the diagonal holds beta)
#pragma unroll
  for (int q = 0; q < QRS_CPW; ++q)
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) {
      const int c = wave + 4 * q, r = lane + 64 * e;
      if (r < k && c < n) Rout[mp.r_off + (int64_t)r * mp.r_sr + (int64_t)c * mp.r_sc] = (r <= c) ? a[q][e] : 0.0;
    }
  __syncthreads();   // every reflector is in V
  // Q = H_1 ... H_k I[:, :k], own columns, last reflector first; H_j leaves unit columns left of j alone
#pragma unroll
  for (int q = 0; q < QRS_CPW; ++q)
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) a[q][e] = (lane + 64 * e == wave + 4 * q) ? 1.0 : 0.0;
  for (int j = k - 1; j >= 0; --j) {
    double v[QRS_RPL];
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) v[e] = V[j][lane + 64 * e];
    const double t = tau[j];
    double acc[QRS_CPW];
#pragma unroll
    for (int q = 0; q < QRS_CPW; ++q) {
      double d = 0.0;
#pragma unroll
      for (int e = 0; e < QRS_RPL; ++e) d += v[e] * a[q][e];
      acc[q] = d;
    }
    qr_wave_sum8(acc, lane);
#pragma unroll
    for (int q = 0; q < QRS_CPW; ++q) {
      const int c = wave + 4 * q;
      if (c >= j && c < k) {
        const double w = t * acc[q];
#pragma unroll
        for (int e = 0; e < QRS_RPL; ++e) a[q][e] -= v[e] * w;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QRS_CPW; ++q)
#pragma unroll
    for (int e = 0; e < QRS_RPL; ++e) {
      const int c = wave + 4 * q, r = lane + 64 * e;
      if (r < m && c < k) Q[mp.q_off + (int64_t)r * mp.q_sr + (int64_t)c * mp.q_sc] = a[q][e];
    }
}

__global__ __launch_bounds__(256) void qr_kernel(const double* __restrict__ Ain, int m, int n, double* __restrict__ Q,
                                                  double* __restrict__ Rout, QrMap mp) {
  extern __shared__ double smem[];
  const int k = min(m, n);
  const int ldm = m + 1;
  double* A = smem;                 // A[c * ldm + r]
  double* tau = A + (size_t)n * ldm;
  double* Qw = tau + k + 8 + n;     // Qw[c * ldm + r], m x k
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool rows_fast = mp.a_sr == 1 || mp.a_sr == -1;   // consecutive threads along whichever index is dense in memory
  for (int idx = tid; idx < m * n; idx += 256) {
    int r, c;
    if (rows_fast) {
      c = idx / m;
      r = idx - c * m;
    } else {
      r = idx / n;
      c = idx - r * n;
    }
    A[c * ldm + r] = Ain[mp.a_off + (int64_t)r * mp.a_sr + (int64_t)c * mp.a_sc];
  }
  __syncthreads();
  // Columns are dealt to the four waves (c mod 4).  The owner of column j forms its reflector
  // (wave-level reduction), one barrier publishes it, then every wave applies it to its own
  // columns: dot product and update stay inside the wave, so a step costs one barrier.
  for (int j = 0; j < k; ++j) {
    double* vj = A + (size_t)j * ldm;
    if (wave == (j & 3)) {
      double part = 0.0;
      for (int r = j + 1 + lane; r < m; r += 64) part += vj[r] * vj[r];
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      const double alpha = vj[j];
      double t = 0.0, beta = alpha, scale = 0.0;
      if (part != 0.0) {
        const double nrm = sqrt(alpha * alpha + part);
        beta = alpha >= 0.0 ? -nrm : nrm;
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      for (int r = j + 1 + lane; r < m; r += 64) vj[r] *= scale;
      if (lane == 0) {
        tau[j] = t;
        vj[j] = beta;
      }
    }
    __syncthreads();
    const double t = tau[j];
    // own columns right of j, eight at a time: their dot products are reduced together so that the
    // shuffle latencies overlap instead of adding up column after column
    for (int c0 = j + 1 + ((wave - (j + 1)) & 3); c0 < n; c0 += 32) {
      double acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = min(c0 + 4 * q, n - 1);
        const double* ac = A + (size_t)c * ldm;
        double a = 0.0;
        for (int r = j + 1 + lane; r < m; r += 64) a += vj[r] * ac[r];
        acc[q] = a;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 4 * q;
        if (c < n) {
          double* ac = A + (size_t)c * ldm;
          const double w = t * (acc[q] + ac[j]);
          for (int r = j + 1 + lane; r < m; r += 64) ac[r] -= vj[r] * w;
          if (lane == 0) ac[j] -= w;
        }
      }
    }
  }
  __syncthreads();
  // R (k x n, upper triangular)
  for (int idx = tid; idx < k * n; idx += 256) {
    const int r = idx / n, c = idx - r * n;
    Rout[mp.r_off + (int64_t)r * mp.r_sr + (int64_t)c * mp.r_sc] = (r <= c) ? A[c * ldm + r] : 0.0;
  }
  // Q = H_1 ... H_k I[:, :k] (dorg2r).  Column c of the identity is touched by H_j only for
  // j <= c, and the columns are independent: each wave runs its own columns through their
  // reflectors from the last one backwards, no barrier.
  for (int c0 = wave; c0 < k; c0 += 32) {   // eight own columns together (interleaved reductions)
    int jtop = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + 4 * q;
      if (c < k) {
        double* qc = Qw + (size_t)c * ldm;
        for (int r = lane; r < m; r += 64) qc[r] = (r == c) ? 1.0 : 0.0;
        jtop = c;
      }
    }
    for (int j = jtop; j >= 0; --j) {
      const double* vj = A + (size_t)j * ldm;
      const double t = tau[j];
      double acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = min(c0 + 4 * q, k - 1);
        const double* qc = Qw + (size_t)c * ldm;
        double a = 0.0;
        for (int r = j + 1 + lane; r < m; r += 64) a += vj[r] * qc[r];
        acc[q] = a;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 4 * q;
        if (c < k && c >= j) {   // H_j leaves the unit columns left of j alone
          double* qc = Qw + (size_t)c * ldm;
          const double w = t * (acc[q] + qc[j]);
          for (int r = j + 1 + lane; r < m; r += 64) qc[r] -= vj[r] * w;
          if (lane == 0) qc[j] -= w;
        }
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < m * k; idx += 256) {
    int r, c;
    if (rows_fast) {
      c = idx / m;
      r = idx - c * m;
    } else {
      r = idx / k;
      c = idx - r * k;
    }
    Q[mp.q_off + (int64_t)r * mp.q_sr + (int64_t)c * mp.q_sc] = Qw[c * ldm + r];
  }
}

// File-scope scratch of ttk_rhs_static_f64, the entry point for callers that cannot allocate.  Every
// GPU the library runs on holds its own copy until the process ends.  In host code the array's name
// is only a handle for hipGetSymbolAddress; the address kernels may use is looked up once per device,
// with that device current, and kept.
#define TTK_STATIC_SCRATCH_DOUBLES (4 * 1024 * 1024)   // 32 MiB
static __device__ __align__(256) double scratch[TTK_STATIC_SCRATCH_DOUBLES];
static double* g_scratch_ptr[TTK_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static double* static_scratch_ptr() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TTK_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess) return nullptr;
    g_scratch_ptr[dev] = (double*)ptr;
  }
  return g_scratch_ptr[dev];
}

// Event pairs around the Gram kernel (diagnostic; armed by ttk_gram_timing).
static const int kTimingPool = 2048;
static hipEvent_t g_ev_start[kTimingPool], g_ev_stop[kTimingPool];
static bool g_ev_created = false, g_ev_armed = false;
static int g_ev_count = 0;
static int g_ev_which = 1;   // which kernel is bracketed: see ttk_gram_timing

// Shared with ttk_dense.hip: `which` names the kernel about to be launched on `st`.
bool ttk_timing_begin(int which, hipStream_t st) {
  const bool timed = g_ev_armed && g_ev_which == which && g_ev_count < kTimingPool;
  if (timed) (void)hipEventRecord(g_ev_start[g_ev_count], st);
  return timed;
}
void ttk_timing_end(hipStream_t st) { (void)hipEventRecord(g_ev_stop[g_ev_count++], st); }

static bool mvf_timing_begin(hipStream_t st) { return ttk_timing_begin(2, st); }
static void mvf_timing_end(hipStream_t st) { ttk_timing_end(st); }

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
  g_ev_which = enable >= 1 && enable <= 6 ? enable : 1;
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
                          int p, int rr, int accumulate, void* stream) {
  if (!E || !X || !core || !out || S < 0 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  return env_step_launch(true, E, X, core, out, S * C, C, rl, p, rr, accumulate, (hipStream_t)stream);
}

int ttk_env_step_right_f64(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                           int rr, int accumulate, void* stream) {
  if (!R || !X || !core || !out || S < 0 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  return env_step_launch(false, R, X, core, out, S, 1, rr, p, rl, accumulate, (hipStream_t)stream);
}

int ttk_predict_f64(const double* L, const double* R, double* y, int64_t S, int C, int r, int accumulate, void* stream) {
  if (!L || !R || !y || S < 0 || C < 1 || r < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  const int64_t M = S * C;
  hipLaunchKernelGGL(predict_kernel, dim3((unsigned)((M * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, R, y,
                     M, C, r, accumulate);
  return ttk_launch_status();
}

int ttk_loss_f64(int kind, double w, const double* y_pred, const double* y, double* loss, double* g, double* H,
                 int64_t S, int C, void* stream) {
  if (kind < 0 || kind > 3 || !y_pred || !y || !loss || !g || !H || S < 0 || C < 1) return TTK_EINVAL;
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

int ttk_kr_rows_f64(const double* L, const double* X, const double* R, double* J, int64_t S, int C, int rl, int p, int rr,
                    int accumulate, void* stream) {
  if (!L || !X || !R || !J || S < 0 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  if (S == 0) return TTK_OK;
  const int64_t n = S * C * (int64_t)rl * p * rr;
  hipLaunchKernelGGL(kr_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, X, R, J, S,
                     C, rl, p, rr, accumulate);
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
  // the slab count depends on how many pair classes a call names: size for the worst case
  int nslabs = d.nslabs;
  if (d.NX <= GRAM_MAX_CLASSED)
    for (int nrep = 1; nrep < d.NX; ++nrep) {
      const int ns = gram_dims(S, C, rl, p, rr, sym_left, nrep).nslabs;
      if (ns > nslabs) nslabs = ns;
    }
  const int64_t W = 2 * (int64_t)C * rl + d.NX + rr + 1;
  return ((int64_t)(1 + (nslabs > 1 ? nslabs : 0)) * d.gsize + (S + 3) * W) * (int64_t)sizeof(double);
}

int ttk_gram_f64(const double* L, const double* H, const double* X, const double* R, int64_t S, int C, int rl, int p,
                 int rr, int sym_left, const int* x_pair_rep, void* ws, int64_t ws_bytes, void* stream) {
  return ttk_gram_cancellable_f64(L, H, X, R, S, C, rl, p, rr, sym_left, x_pair_rep, ws, ws_bytes, nullptr, 0u, stream);
}

int ttk_gram_cancel(void* cancel_word, unsigned generation, void* stream) {
  if (!cancel_word) return TTK_EINVAL;
  if (hipMemsetD32Async((hipDeviceptr_t)cancel_word, (int)generation, 1, (hipStream_t)stream) != hipSuccess) return TTK_ELAUNCH;
  return TTK_OK;
}

int ttk_gram_cancellable_f64(const double* L, const double* H, const double* X, const double* R, int64_t S, int C, int rl,
                             int p, int rr, int sym_left, const int* x_pair_rep, void* ws, int64_t ws_bytes,
                             const void* cancel_word, unsigned generation, void* stream) {
  if (!L || !H || !X || !R || !ws || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const int NX = p * (p + 1) / 2;
  int nrep = 0;
  if (x_pair_rep && NX <= GRAM_MAX_CLASSED) {
    for (int b = 0; b < NX; ++b) {
      const int r = x_pair_rep[b];
      if (r < 0 || r > b || x_pair_rep[r] != r) return TTK_EINVAL;
      if (r == b) ++nrep;
    }
    if (nrep == NX) nrep = 0;
  }
  GramDims d = gram_dims(S, C, rl, p, rr, sym_left, nrep);
  if (d.CH < 4) return TTK_EINVAL;
  if (nrep) {
    d.classed = 1;
    int k = 0;
    for (int b = 0; b < NX; ++b) {
      d.repof[b] = x_pair_rep[b];
      if (x_pair_rep[b] == b) d.reps[k++] = b;
    }
  }
  if (ws_bytes < ttk_gram_ws_bytes(S, C, rl, p, rr, sym_left)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* G = (double*)ws;
  double* partial = d.nslabs > 1 ? G + d.gsize : G;
  double* F = G + (int64_t)(1 + (d.nslabs > 1 ? d.nslabs : 0)) * d.gsize;
  const int64_t W = 2 * (int64_t)C * rl + d.NX + rr + 1;
  hipLaunchKernelGGL(gram_pack_kernel, dim3((unsigned)((((S + 3) & ~(int64_t)3) * W + 255) / 256)), dim3(256), 0, st, L, H, X, R, S, C, rl, p,
                     rr, F);
  const size_t lds = (size_t)2 * d.bufd * sizeof(double);
  const unsigned grid = (unsigned)d.ngroups * (unsigned)d.nslabs;
  const bool timed = g_ev_armed && g_ev_which == 1 && g_ev_count < kTimingPool;
  if (timed) (void)hipEventRecord(g_ev_start[g_ev_count], st);
  if (C == 1)
    hipLaunchKernelGGL(gram_kernel<true>, dim3(grid), dim3(256), lds, st, (const double*)F, S, C, rl, p, rr, sym_left,
                       d, partial, (const unsigned*)cancel_word, generation);
  else
    hipLaunchKernelGGL(gram_kernel<false>, dim3(grid), dim3(256), lds, st, (const double*)F, S, C, rl, p, rr, sym_left,
                       d, partial, (const unsigned*)cancel_word, generation);
  if (timed) (void)hipEventRecord(g_ev_stop[g_ev_count++], st);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  if (d.classed) {
    const int64_t nred = (int64_t)d.MPpad * d.nrep * (d.NRpad / 2);
    hipLaunchKernelGGL(gram_reduce_classed_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, st,
                       (const double*)partial, d.nslabs, d.gsize, d, G);
    rc = ttk_launch_status();
  } else if (d.nslabs > 1) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((d.gsize + 255) / 256)), dim3(256), 0, st, partial,
                       d.nslabs, d.gsize, d.gsize, G, 0);
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

// workspace: [ aug ((nb+1)*64 x nb*64) | inverses of the diagonal blocks (nb x 64 x 64) | control block of the
// single-launch factorisation: task counters + abort word in slot 0, one progress word per block row, the
// table of compute units ]
static inline int64_t chol_ctl_offset(int64_t nb) {
  return round_up64(((nb + 1) * CHB * nb * CHB + nb * CHB * CHB) * (int64_t)sizeof(double), 256);
}
static inline int64_t chol_ctl_bytes(int64_t nb) {
  int64_t bytes = ((nb + 2) * DAG_SLOT + DAG_CU_WORDS) * (int64_t)sizeof(unsigned);
#ifdef TTK_DAG_TRACE
  bytes += nb * 16 * (int64_t)sizeof(long long);
#endif
  return bytes;
}

int64_t ttk_chol_ws_bytes(int n) {
  if (n < 1) return TTK_EINVAL;
  const int64_t nb = (n + CHB - 1) / CHB;
  return chol_ctl_offset(nb) + chol_ctl_bytes(nb) + 256;
}

int ttk_chol_prepare_f64(const void* gram_ws, int rl, int p, int rr, int sym_left, const double* scale, double ridge,
                         const double* b, const double* w, void* chol_ws, void* stream) {
  if (!gram_ws || !b || !chol_ws || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const GramDims d = gram_dims(1, 1, rl, p, rr, sym_left);
  const int n = rl * p * rr;
  const int nb = (n + CHB - 1) / CHB;
  const int ld = nb * CHB;
  double* aug = (double*)chol_ws;
  hipLaunchKernelGGL(gram_expand_lower_kernel, dim3((unsigned)((ld + 255) / 256), (unsigned)ld), dim3(256), 0, st,
                     (const double*)gram_ws, rl, p, rr, sym_left, d.NX, d.NRpad, scale, ridge, aug, ld);
  if (hipMemsetAsync(aug + (int64_t)ld * ld, 0, (size_t)CHB * ld * sizeof(double), st) != hipSuccess) return TTK_ELAUNCH;
  hipLaunchKernelGGL(solve_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, w, scale, ridge,
                     aug + (int64_t)ld * ld, (int64_t)n);
  return ttk_launch_status();
}

static int g_group4_min = 2048;   // trailing order above which panels are applied four at a time
static int g_syrk_tile = 0;   // 0: chosen from the trailing size
static void chol_syrk_launch(hipStream_t st, double* aug, int ld, int pc0, int K, int o0, int nrows, int ncols) {
  if (o0 >= ncols || o0 >= nrows) return;
  // 64x64 tiles while the 128x128 ones would leave compute units idle or half a round of slots empty
  const int t128 = (nrows - o0 + 127) / 128;
  const int64_t n128 = (int64_t)t128 * (t128 + 1) / 2;
  const int TS = g_syrk_tile ? g_syrk_tile : (n128 < SYRK_SMALL_TILES ? 64 : 128);
  const int trows = (nrows - o0 + TS - 1) / TS;
  if (ncols - o0 <= TS) {
    if (TS == 64)
      hipLaunchKernelGGL(chol_syrk_kernel<2>, dim3((unsigned)trows), dim3(256), 0, st, aug, ld, pc0, K, o0, nrows, ncols, 0);
    else
      hipLaunchKernelGGL(chol_syrk_kernel<4>, dim3((unsigned)trows), dim3(256), 0, st, aug, ld, pc0, K, o0, nrows, ncols, 0);
    return;
  }
  const int nst_rows = (trows + SYS - 1) / SYS;
  const int nst = nst_rows * (nst_rows + 1) / 2;
  const unsigned grid = (unsigned)((nst + 7) / 8) * 8 * SYS * SYS;
  if (TS == 64)
    hipLaunchKernelGGL(chol_syrk_kernel<2>, dim3(grid), dim3(256), 0, st, aug, ld, pc0, K, o0, nrows, ncols, nst_rows);
  else
    hipLaunchKernelGGL(chol_syrk_kernel<4>, dim3(grid), dim3(256), 0, st, aug, ld, pc0, K, o0, nrows, ncols, nst_rows);
}

#ifdef TTK_DEBUG_HOOKS
int ttk_debug_syrk(double* aug, int ld, int pc0, int K, int o0, int nrows, int ncols, void* stream) {
  chol_syrk_launch((hipStream_t)stream, aug, ld, pc0, K, o0, nrows, ncols);
  return 0;
}
int ttk_debug_syrk_tile(int ts) { g_syrk_tile = ts; return 0; }
int ttk_debug_group(int g) { g_group4_min = g; return 0; }
int ttk_debug_rhs_slabs(int n) { g_rhs_max_slabs = n; return 0; }
int ttk_debug_gram_slabs(int n) { g_gram_slabs = n; return 0; }
int ttk_debug_env_reg(int on) { g_env_reg = on; return 0; }
int ttk_debug_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(long long) * 64); }
#endif

// 0: one launch per panel / trailing update; 1: the single-launch factorisation; -1: single launch from
// CHOL_DAG_MIN_NB block columns on.  The single launch is bound by its chain of diagonal tasks (about 28 us
// per block column); the multi-launch form pays panel + trailing update per column, which is less up to an
// order of about 2000 (measured, multi vs single: 0.67 / 0.68 ms at 1536, 0.91 / 0.90 at 2048, 2.11 / 1.80 at
// 4096, 3.97 / 2.89 at 6144).
#define CHOL_DAG_MIN_NB 32
static int g_chol_mode = -1;
static int g_dag_chain = 0;   // compute units kept for the diagonal chain; 0: chosen from the order
static int g_dag_spin_limit = DAG_SPIN_LIMIT;
static int g_back_chain = 1;  // the backward substitution behind the single-launch factorisation as one launch
int ttk_chol_mode(int mode) {
  if (mode == 200 || mode == 201) {   // testing: 200 one launch per group of block rows, 201 one launch in all
    g_back_chain = mode - 200;
    return g_chol_mode;
  }
  if (mode >= 1000) {   // testing: polls before a waiting lane gives up (1000 restores the default)
    g_dag_spin_limit = mode > 1000 ? mode - 1000 : (int)DAG_SPIN_LIMIT;
    return g_chol_mode;
  }
  if (mode >= 100) {   // tuning: 100 + units
    g_dag_chain = mode - 100;
    return g_chol_mode;
  }
  const int old = g_chol_mode;
  if (mode >= -1 && mode <= 1) g_chol_mode = mode;
  return old;
}

// workgroups of the single-launch factorisation that fit the device at once (more would only queue)
static int chol_dag_slots(bool lazy) {
  static std::mutex mu;
  static int slots[2][TTK_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TTK_MAX_DEVICES) return 256;
  std::lock_guard<std::mutex> lock(mu);
  int& sl = slots[lazy ? 1 : 0][dev];
  if (sl == 0) {   // occupancy of the instantiation that is launched
    int per_cu = 0, cus = 0;
    const hipError_t e = lazy ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_dag_kernel<true>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_dag_kernel<false>, 256, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    sl = per_cu * cus;
  }
  return sl;
}

static int chol_back_substitute(hipStream_t st, double* aug, int ld, int nb, const double* invL, double* x, int n) {
  for (int jhi = nb - 1; jhi >= 0; jhi -= BKG) {
    const int jlo = jhi - BKG + 1 > 0 ? jhi - BKG + 1 : 0;
    hipLaunchKernelGGL(chol_back_kernel, dim3((unsigned)(jlo > 0 ? jlo : 1)), dim3(256), 0, st, aug, ld, nb, jlo, jhi, invL,
                       x, n);
  }
  return ttk_launch_status();
}

// gs != nullptr: the matrix is still in the compressed Gram workspace gs->G (see DagGram); the single-launch
// factorisation reads it from there, the multi-launch one has it expanded into the storage first.
// whether a solve of nb block rows runs the single-launch factorisation
static bool chol_uses_dag(int nb) {
  const int64_t aug_bytes = (int64_t)(nb + 1) * CHB * nb * CHB * (int64_t)sizeof(double);
  return (g_chol_mode == 1 || (g_chol_mode < 0 && nb >= CHOL_DAG_MIN_NB)) && aug_bytes < (int64_t(1) << 32);
}

// prepped: solve_prep_kernel has already cleared the pivot report and the control words (single-launch form only)
static int chol_solve_impl(void* chol_ws, int n, double* x, int* info, hipStream_t st, const DagGram* gs,
                           bool prepped = false) {
  const int nb = (n + CHB - 1) / CHB;
  const int ld = nb * CHB;
  const int nbr = nb + 1;
  const int nrows = nbr * CHB, ncols = ld;
  double* aug = (double*)chol_ws;
  double* invL = aug + (int64_t)nbr * CHB * ld;
  const int64_t aug_bytes = (int64_t)nbr * CHB * ld * (int64_t)sizeof(double);
  const bool dag = chol_uses_dag(nb);
  if (!(prepped && dag) && hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return TTK_ELAUNCH;
  if (dag) {
    // every word the kernel polls or counts starts at zero
    unsigned* ctl = (unsigned*)((char*)chol_ws + chol_ctl_offset(nb));
    if (!prepped) {
      if (hipMemsetAsync(ctl, 0, (size_t)chol_ctl_bytes(nb), st) != hipSuccess) return TTK_ELAUNCH;
      if (hipMemsetD32Async((hipDeviceptr_t)(ctl + 4), g_dag_spin_limit, 1, st) != hipSuccess) return TTK_ELAUNCH;
    }
    // every slot of the device gets a workgroup, so that each compute unit sees its first arrival at once
    const int slots = chol_dag_slots(gs != nullptr);
    int nchain = g_dag_chain > 0 ? g_dag_chain : max(2, min(24, nb / 4));
    nchain = max(1, min(nchain, slots / 16));   // most of the device must stay with the workers
    if (gs)
      hipLaunchKernelGGL(chol_dag_kernel<true>, dim3((unsigned)slots), dim3(256), 0, st, aug, ld, nb, (unsigned)aug_bytes,
                         invL, info, ctl, nchain, *gs);
    else
      hipLaunchKernelGGL(chol_dag_kernel<false>, dim3((unsigned)slots), dim3(256), 0, st, aug, ld, nb, (unsigned)aug_bytes,
                         invL, info, ctl, nchain, DagGram{});
    int rc = ttk_launch_status();
    if (rc != TTK_OK) return rc;
    // invL already holds the inverses.  One launch for the whole substitution while every workgroup of it fits the
    // device at once (they wait for each other); the one-launch-per-group form beyond (ttk_chol_mode(200) forces it)
    if (g_back_chain && nb <= slots) {
      hipLaunchKernelGGL(chol_back_chain_kernel, dim3((unsigned)nb), dim3(256), 0, st, aug, ld, nb, (const double*)invL, x, n, ctl,
                         info);
      return ttk_launch_status();
    }
    return chol_back_substitute(st, aug, ld, nb, invL, x, n);
  }
  if (gs)
    hipLaunchKernelGGL(gram_expand_lower_kernel, dim3((unsigned)((ld + 255) / 256), (unsigned)ld), dim3(256), 0, st, gs->G,
                       gs->rl, gs->p, gs->rr, gs->sym, gs->NX, gs->NRpad, gs->scale, gs->ridge, aug, ld);
  // One stream.  (A look-ahead that factorises the next pair of panels on a second,
  // high-priority stream beside the trailing update was measured and gained nothing: the
  // panel workgroups queue behind resident trailing-update workgroups for CU space.)
  // Panels are applied to the trailing matrix in pairs (K = 128).  While the trailing matrix is large,
  // two pairs are combined: the first pair goes onto the next pair's two block columns only (one narrow
  // launch), then all four panels (K = 256) onto everything right of them, which sweeps the large
  // trailing matrix half as often; for small trailing matrices the extra launch costs more than it saves.
  int base = -1;   // first panel of an open group of four
  for (int j = 0; j < nb; ++j) {
    // odd panels first absorb the previous panel's update of their own block column
    hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)(nbr - j)), dim3(256), 0, st, aug, ld, j, j & 1, info, invL);
    if (j + 1 >= nb) break;
    if (base < 0) {
      if (j & 1) {
        if ((nb - (j + 1)) * CHB > g_group4_min && j + 2 < nb) {
          chol_syrk_launch(st, aug, ld, (j - 1) * CHB, 2 * CHB, (j + 1) * CHB, nrows, min(ncols, (j + 3) * CHB));
          base = j - 1;
        } else {
          chol_syrk_launch(st, aug, ld, (j - 1) * CHB, 2 * CHB, (j + 1) * CHB, nrows, ncols);
        }
      }
    } else if (j - base == 3) {
      chol_syrk_launch(st, aug, ld, base * CHB, 4 * CHB, (j + 1) * CHB, nrows, ncols);
      base = -1;
    }
  }
  hipLaunchKernelGGL(chol_inv_kernel, dim3((unsigned)nb), dim3(256), 0, st, invL);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  return chol_back_substitute(st, aug, ld, nb, invL, x, n);
}

int ttk_chol_solve_f64(void* chol_ws, int n, double* x, int* info, void* stream) {
  if (!chol_ws || !x || !info || n < 1) return TTK_EINVAL;
  return chol_solve_impl(chol_ws, n, x, info, (hipStream_t)stream, nullptr);
}

int ttk_chol_solve_gram_f64(const void* gram_ws, int rl, int p, int rr, int sym_left, const double* scale, double ridge,
                            const double* b, const double* w, void* chol_ws, double* x, int* info, void* stream) {
  if (!gram_ws || !b || !chol_ws || !x || !info || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const GramDims d = gram_dims(1, 1, rl, p, rr, sym_left);
  const int n = rl * p * rr;
  const int nb = (n + CHB - 1) / CHB;
  const int ld = nb * CHB;
  double* aug = (double*)chol_ws;
  // the right-hand side row (and the zero rows below it) is all that goes into the storage up front
  const bool prepped = chol_uses_dag(nb);
  if (prepped) {
    unsigned* ctl = (unsigned*)((char*)chol_ws + chol_ctl_offset(nb));
    const int64_t elems = (int64_t)CHB * ld;
    hipLaunchKernelGGL(solve_prep_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, b, w, scale, ridge,
                       aug + (int64_t)ld * ld, elems, (int64_t)n, ctl, (int)(chol_ctl_bytes(nb) / (int64_t)sizeof(unsigned)),
                       g_dag_spin_limit, info);
  } else {
    if (hipMemsetAsync(aug + (int64_t)ld * ld, 0, (size_t)CHB * ld * sizeof(double), st) != hipSuccess) return TTK_ELAUNCH;
    hipLaunchKernelGGL(solve_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, w, scale, ridge,
                       aug + (int64_t)ld * ld, (int64_t)n);
  }
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  DagGram gs;
  gs.G = (const double*)gram_ws;
  gs.scale = scale;
  gs.ridge = ridge;
  gs.rl = rl; gs.p = p; gs.rr = rr; gs.sym = sym_left; gs.NX = d.NX; gs.NRpad = d.NRpad; gs.D = n;
  gs.inv_rr = 1.0f / (float)rr;
  gs.inv_p = 1.0f / (float)p;
  return chol_solve_impl(chol_ws, n, x, info, st, &gs, prepped);
}

int64_t ttk_rhs_ws_bytes(int64_t S, int C, int rl, int p, int rr) {
  if (S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const RhsDims d = rhs_dims(S, C, rl, p, rr);
  if (d.CH < 4) return TTK_EINVAL;
  // slab partials, then GL when there are several outputs
  return ((int64_t)d.nslabs * rl * p * rr + (C > 1 ? S * rl : 0)) * (int64_t)sizeof(double);
}

// The right-hand side with its scratch (slab partials, then GL when there are several outputs) at `ws`.
static int rhs_run(const double* L, const double* X, const double* R, const double* g, double* b, int64_t S, int C,
                   int rl, int p, int rr, int accumulate, const RhsDims& d, double* ws, hipStream_t st) {
  const int64_t D = (int64_t)rl * p * rr;
  const double* A = L;
  const double* gx = g;
  if (C > 1) {
    double* GL = ws + (int64_t)d.nslabs * D;
    hipLaunchKernelGGL(rhs_gl_kernel, dim3((unsigned)((S * rl + 255) / 256)), dim3(256), 0, st, g, L, GL, S, C, rl);
    A = GL;
    gx = nullptr;
  }
  const size_t lds = (size_t)d.CH * d.W * sizeof(double);
  if (d.mt == 1 && d.nv == 3)
    hipLaunchKernelGGL((rhs_kernel<1, 3>), dim3((unsigned)d.nslabs), dim3(256), lds, st, A, X, R, gx, S, rl, p, rr, d, ws);
  else if (d.mt == 1 && d.nv == 8)
    hipLaunchKernelGGL((rhs_kernel<1, 8>), dim3((unsigned)d.nslabs), dim3(256), lds, st, A, X, R, gx, S, rl, p, rr, d, ws);
  else if (d.mt == 1 && d.nv == 12)
    hipLaunchKernelGGL((rhs_kernel<1, 12>), dim3((unsigned)d.nslabs), dim3(256), lds, st, A, X, R, gx, S, rl, p, rr, d, ws);
  else if (d.mt == 1)
    hipLaunchKernelGGL((rhs_kernel<1, 6>), dim3((unsigned)d.nslabs), dim3(256), lds, st, A, X, R, gx, S, rl, p, rr, d, ws);
  else
    hipLaunchKernelGGL((rhs_kernel<2, 3>), dim3((unsigned)d.nslabs), dim3(256), lds, st, A, X, R, gx, S, rl, p, rr, d, ws);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  if (d.nslabs > 32)
    hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)ws,
                       d.nslabs, D, D, b, accumulate);
  else
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, (const double*)ws,
                       d.nslabs, D, D, b, accumulate);
  return ttk_launch_status();
}

int ttk_rhs_f64(const double* L, const double* X, const double* R, const double* g, double* b, int64_t S, int C, int rl,
                int p, int rr, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
  if (!L || !X || !R || !g || !b || !ws || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const RhsDims d = rhs_dims(S, C, rl, p, rr);
  if (d.CH < 4) return TTK_EINVAL;
  if (ws_bytes < ttk_rhs_ws_bytes(S, C, rl, p, rr)) return TTK_EWORKSPACE;
  if (S * (int64_t)(rl > rr ? rl : rr) >= (1ll << 40)) return TTK_EINVAL;
  return rhs_run(L, X, R, g, b, S, C, rl, p, rr, accumulate, d, (double*)ws, (hipStream_t)stream);
}

int64_t ttk_rhs_static_max_bytes(void) {
  // Also the set-up call: the array's address on the current device is looked up here (when a device is
  // there), so that a later ttk_rhs_static_f64 under stream capture finds it and touches no symbol table.
  (void)static_scratch_ptr();
  return (int64_t)TTK_STATIC_SCRATCH_DOUBLES * (int64_t)sizeof(double);
}

int ttk_rhs_static_f64(const double* L, const double* X, const double* R, const double* g, double* b, int64_t S, int C,
                       int rl, int p, int rr, int accumulate, void* stream) {
  if (!L || !X || !R || !g || !b || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  const RhsDims d = rhs_dims(S, C, rl, p, rr);
  if (d.CH < 4) return TTK_EINVAL;
  if (S * (int64_t)(rl > rr ? rl : rr) >= (1ll << 40)) return TTK_EINVAL;
  if (ttk_rhs_ws_bytes(S, C, rl, p, rr) > (int64_t)TTK_STATIC_SCRATCH_DOUBLES * (int64_t)sizeof(double)) return TTK_EWORKSPACE;
  double* ws = static_scratch_ptr();
  if (!ws) return TTK_ELAUNCH;
  // every slab partial and every GL entry is written before it is read: nothing has to start at zero
  return rhs_run(L, X, R, g, b, S, C, rl, p, rr, accumulate, d, ws, (hipStream_t)stream);
}

int64_t ttk_matvec_fused_ws_bytes(int64_t S, int C, int rl, int p, int rr) {
  const MvfDims d = mvf_dims(S, C, rl, p, rr);
  if (!d.ok) return TTK_EINVAL;
  return ((int64_t)d.nwg * d.D + d.nwg + d.nrp) * (int64_t)sizeof(double);
}

int ttk_matvec_fused_f64(const double* L, const double* X, const double* R, const double* H, const double* v, double* out,
                         int64_t S, int rl, int p, int rr, void* ws, int64_t ws_bytes, void* stream) {
  if (!L || !X || !R || !v || !out || !ws) return TTK_EINVAL;
  const MvfDims d = mvf_dims(S, 1, rl, p, rr);
  if (!d.ok) return TTK_EINVAL;
  if (ws_bytes < ttk_matvec_fused_ws_bytes(S, 1, rl, p, rr)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  double* pq_part = partial + (int64_t)d.nwg * d.D;
  int rc = mvf_launch(0, d, L, X, R, H, v, nullptr, nullptr, nullptr, nullptr, 0, S, p, partial, pq_part, st);
  if (rc != TTK_OK) return rc;
  if (d.nwg > 32)
    hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((d.D + 15) / 16)), dim3(256), 0, st, (const double*)partial,
                       d.nwg, d.D, d.D, out, 0);
  else
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((d.D + 255) / 256)), dim3(256), 0, st, (const double*)partial,
                       d.nwg, d.D, d.D, out, 0);
  return ttk_launch_status();
}

int ttk_cg_fused_iter_f64(const double* L, const double* X, const double* R, const double* H, double* x, double* r,
                          double* pbuf, double* state, int it, int64_t S, int rl, int p, int rr, void* ws,
                          int64_t ws_bytes, void* stream) {
  if (!L || !X || !R || !x || !r || !pbuf || !state || !ws || it < 0) return TTK_EINVAL;
  const MvfDims d = mvf_dims(S, 1, rl, p, rr);
  if (!d.ok) return TTK_EINVAL;
  if (ws_bytes < ttk_matvec_fused_ws_bytes(S, 1, rl, p, rr)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  double* pq_part = partial + (int64_t)d.nwg * d.D;
  double* rho_part = pq_part + d.nwg;
  double* pnew = pbuf + (int64_t)(it & 1) * d.D;
  const double* pold = pbuf + (int64_t)((it + 1) & 1) * d.D;
  int rc = mvf_launch(1, d, L, X, R, H, r, pold, pnew, state, rho_part, it, S, p, partial, pq_part, st);
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(cg_reduce_update_kernel, dim3((unsigned)d.nrp), dim3(256), 0, st, (const double*)partial, d.nwg, d.D,
                     (const double*)pq_part, state, it, x, r, (const double*)pnew, rho_part);
  return ttk_launch_status();
}

int ttk_gl_f64(const double* g, const double* L, double* GL, int64_t S, int C, int rl, void* stream) {
  if (!g || !L || !GL || S < 1 || C < 1 || rl < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(rhs_gl_kernel, dim3((unsigned)((S * rl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, L, GL, S,
                     C, rl);
  return ttk_launch_status();
}

int ttk_jvp_f64(const double* L, const double* X, const double* R, const double* H, const double* v, double* tmp,
                double* hu, int64_t S, int C, int rl, int p, int rr, void* stream) {
  if (!L || !X || !R || !H || !v || !tmp || !hu || S < 1 || C < 1 || rl < 1 || p < 1 || rr < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // resident-core kernel with the dot against R (and, for one output, the Hessian) in its epilogue:
  // the (S*C, rr) intermediate never reaches memory
  int rc = env_fast_launch(true, L, X, v, C == 1 ? hu : tmp, S * C, C, rl, p, rr, 0, 1, R, C == 1 ? H : nullptr, st);
  if (rc == TTK_OK && C > 1) {
    hipLaunchKernelGGL(jvp_hess_kernel, dim3((unsigned)((S * C + 255) / 256)), dim3(256), 0, st, tmp, H, hu, S, C);
    return ttk_launch_status();
  }
  if (rc != 1) return rc;
  rc = env_step_launch(true, L, X, v, tmp, S * C, C, rl, p, rr, 0, st);
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(jvp_finish_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, tmp, R, H, hu, S, C, rr);
  return ttk_launch_status();
}

int64_t ttk_qr_lds_bytes(int m, int n) {
  if (m < 1 || n < 1) return TTK_EINVAL;
  const int64_t k = m < n ? m : n;
  return ((int64_t)n * (m + 1) + k + 8 + n + k * (int64_t)(m + 1)) * (int64_t)sizeof(double);
}

static int qr_launch(const double* A, int m, int n, double* Q, double* R, const QrMap& mp, hipStream_t st) {
  if (m <= 64 * QRS_RPL && n <= 4 * QRS_CPW) {
    hipLaunchKernelGGL(qr_small_kernel, dim3(1), dim3(256), 0, st, A, m, n, Q, R, mp);
    return ttk_launch_status();
  }
  const int64_t lds = ttk_qr_lds_bytes(m, n);
  if (lds > 150 * 1024) return TTK_EINVAL;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TTK_ELAUNCH;
  hipLaunchKernelGGL(qr_kernel, dim3(1), dim3(256), (size_t)lds, st, A, m, n, Q, R, mp);
  return ttk_launch_status();
}

int ttk_qr_f64(const double* A, int m, int n, double* Q, double* R, void* stream) {
  if (!A || !Q || !R || m < 1 || n < 1) return TTK_EINVAL;
  const int k = m < n ? m : n;
  const QrMap mp = {0, n, 1, 0, k, 1, 0, n, 1};
  return qr_launch(A, m, n, Q, R, mp, (hipStream_t)stream);
}

int ttk_qr_right_f64(const double* T, int n, int m, double* Qt, double* Rm, void* stream) {
  if (!T || !Qt || !Rm || m < 1 || n < 1) return TTK_EINVAL;
  const int64_t k = m < n ? m : n, M = m, N = n;
  // element (r, c) of the flipped transposed core is T[n-1-c][m-1-r]; Q[r][c] goes to Qt[k-1-c][m-1-r], R[r][c] to Rm[n-1-c][k-1-r]
  const QrMap mp = {(N - 1) * M + (M - 1), -1, -M, (k - 1) * M + (M - 1), -1, -M, (N - 1) * k + (k - 1), -1, -k};
  return qr_launch(T, m, n, Qt, Rm, mp, (hipStream_t)stream);
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

int ttk_cg_update_f64(double* x, double* r, double* p, const double* q, double* state, int64_t D, void* stream) {
  if (!x || !r || !p || !q || !state || D < 1) return TTK_EINVAL;
  if (D <= 8192)
    hipLaunchKernelGGL(cg_update_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, r, p, q, state, D);
  else
    hipLaunchKernelGGL(cg_update_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, r, p, q, state, D);
  return ttk_launch_status();
}

int64_t ttk_minres_update_ws_bytes(int64_t D) {
  if (D < 1) return TTK_EINVAL;
  return D > 8192 ? 3 * MRW_WG * (int64_t)sizeof(double) : 0;
}

int ttk_minres_update_ws_f64(double* y, double* r1, double* r2, double* v, double* w, double* w2, double* x, double* state,
                             int64_t D, void* ws, int64_t ws_bytes, void* stream) {
  if (!y || !r1 || !r2 || !v || !w || !w2 || !x || !state || D < 1) return TTK_EINVAL;
  if (D <= 8192 || !ws) return ttk_minres_update_f64(y, r1, r2, v, w, w2, x, state, D, stream);
  if (ws_bytes < ttk_minres_update_ws_bytes(D)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* p = (double*)ws;
  hipLaunchKernelGGL(minres_wide_kernel<0>, dim3(MRW_WG), dim3(256), 0, st, y, r1, r2, v, w, w2, x, (const double*)state, p, D);
  hipLaunchKernelGGL(minres_wide_kernel<1>, dim3(MRW_WG), dim3(256), 0, st, y, r1, r2, v, w, w2, x, (const double*)state, p, D);
  hipLaunchKernelGGL(minres_wide_kernel<2>, dim3(MRW_WG), dim3(256), 0, st, y, r1, r2, v, w, w2, x, (const double*)state, p, D);
  hipLaunchKernelGGL(minres_wide_scalars_kernel, dim3(1), dim3(256), 0, st, state, (const double*)p);
  return ttk_launch_status();
}

int64_t ttk_cg_update_ws_bytes(int64_t D) {
  if (D < 1) return TTK_EINVAL;
  return D > 8192 ? 2 * CGW_WG * (int64_t)sizeof(double) : 0;
}

int ttk_cg_update_ws_f64(double* x, double* r, double* p, const double* q, double* state, int64_t D, void* ws,
                         int64_t ws_bytes, void* stream) {
  if (!x || !r || !p || !q || !state || D < 1) return TTK_EINVAL;
  if (D <= 8192 || !ws) return ttk_cg_update_f64(x, r, p, q, state, D, stream);
  if (ws_bytes < ttk_cg_update_ws_bytes(D)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* w = (double*)ws;
  hipLaunchKernelGGL(cg_wide_kernel<0>, dim3(CGW_WG), dim3(256), 0, st, x, r, p, q, state, w, D);
  hipLaunchKernelGGL(cg_wide_kernel<1>, dim3(CGW_WG), dim3(256), 0, st, x, r, p, q, state, w, D);
  hipLaunchKernelGGL(cg_wide_kernel<2>, dim3(CGW_WG), dim3(256), 0, st, x, r, p, q, state, w, D);
  hipLaunchKernelGGL(cg_wide_scalars_kernel, dim3(1), dim3(256), 0, st, state, (const double*)w);
  return ttk_launch_status();
}

int ttk_minres_update_f64(double* y, double* r1, double* r2, double* v, double* w, double* w2, double* x, double* state,
                          int64_t D, void* stream) {
  if (!y || !r1 || !r2 || !v || !w || !w2 || !x || !state || D < 1) return TTK_EINVAL;
  hipLaunchKernelGGL(minres_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, y, r1, r2, v, w, w2, x, state, D);
  return ttk_launch_status();
}

int ttk_lanczos_update_f64(double* w, const double* v_prev, const double* v_cur, double* v_next, double* alphas,
                           double* betas, int j, int64_t D, double tol, double* stop, void* stream) {
  if (!w || !v_cur || !v_next || !alphas || !betas || !stop || j < 1 || D < 1) return TTK_EINVAL;
  if (D <= 8192)
    hipLaunchKernelGGL(lanczos_update_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, v_prev, v_cur, v_next,
                       alphas, betas, j, D, tol, stop);
  else if (D < 2 * LZW_WG * 32) {
    hipLaunchKernelGGL(lanczos_update_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, v_prev, v_cur, v_next,
                       alphas, betas, j, D, tol, stop);
  } else {
    hipStream_t st = (hipStream_t)stream;
    const double* cstop = stop;
    hipLaunchKernelGGL(lanczos_wide_kernel<0>, dim3(LZW_WG), dim3(256), 0, st, w, v_prev, v_cur, v_next, betas, j, D, cstop);
    hipLaunchKernelGGL(lanczos_wide_kernel<1>, dim3(LZW_WG), dim3(256), 0, st, w, v_prev, v_cur, v_next, betas, j, D, cstop);
    hipLaunchKernelGGL(lanczos_wide_scalars_kernel, dim3(1), dim3(256), 0, st, (const double*)v_next, alphas, betas, j, tol,
                       stop);
    hipLaunchKernelGGL(lanczos_wide_kernel<2>, dim3(LZW_WG), dim3(256), 0, st, w, v_prev, v_cur, v_next, betas, j, D, cstop);
  }
  return ttk_launch_status();
}

int ttk_lanczos_fused_block_f64(const double* L, const double* X, const double* R, const double* H, double* V, int64_t ldv,
                                double* alphas, double* betas, double* stop, int j_first, int count, double tol, double* w,
                                int64_t S, int rl, int p, int rr, void* ws, int64_t ws_bytes, void* stream) {
  if (!V || !alphas || !betas || !stop || !w || j_first < 1 || count < 0) return TTK_EINVAL;
  const int64_t D = (int64_t)rl * p * rr;
  if (ldv < D) return TTK_EINVAL;
  for (int j = j_first; j < j_first + count; ++j) {
    int rc = ttk_matvec_fused_f64(L, X, R, H, V + (int64_t)(j - 1) * ldv, w, S, rl, p, rr, ws, ws_bytes, stream);
    if (rc != TTK_OK) return rc;
    rc = ttk_lanczos_update_f64(w, j > 1 ? V + (int64_t)(j - 2) * ldv : nullptr, V + (int64_t)(j - 1) * ldv, V + (int64_t)j * ldv,
                                alphas, betas, j, D, tol, stop, stream);
    if (rc != TTK_OK) return rc;
  }
  return TTK_OK;
}

int64_t ttk_lanczos_finish_ws_bytes(int m_cap) {
  if (m_cap < 1) return TTK_EINVAL;
  return (int64_t)(1 + 6 * (int64_t)m_cap) * (int64_t)sizeof(double);
}

int ttk_lanczos_finish_f64(const double* V, int64_t ldv, const double* alphas, const double* betas, const double* stop,
                           int m_run, int m_cap, const double* x0, double* x, int64_t D, void* ws, int64_t ws_bytes,
                           void* stream) {
  if (!V || !alphas || !betas || !stop || !x0 || !x || !ws || D < 1 || ldv < D || m_run < 1 || m_cap < m_run)
    return TTK_EINVAL;
  if (ws_bytes < ttk_lanczos_finish_ws_bytes(m_cap)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)5 * m_cap * sizeof(double);
  static bool allowed[TTK_MAX_DEVICES];
  int in_lds = lds <= 64 * 1024;
  if (!in_lds && lds <= 150 * 1024) in_lds = allow_big_lds((const void*)lanczos_tridiag_kernel, allowed, 150 * 1024) ? 1 : 0;
  hipLaunchKernelGGL(lanczos_tridiag_kernel, dim3(1), dim3(64), in_lds ? lds : 0, st, alphas, betas, stop, m_run, m_cap,
                     (double*)ws, in_lds);
  int grid = (int)((D + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(lanczos_combine_kernel, dim3(grid), dim3(256), 0, st, V, ldv, (const double*)ws, x0, x, D);
  return ttk_launch_status();
}

// samples per LDS chunk and grid of rows_hjv_kernel
static bool rows_hjv_dims(int64_t S, int C, int D, int mode, int* NS, int* grid, size_t* lds) {
  if (S < 1 || C < 1 || C > 32 || D < 1 || D > (mode != 1 ? 512 : 128) || (int64_t)C * D > 4096) return false;
  int ns = 4096 / (C * D);
  if (ns > 2048 / (C * C)) ns = 2048 / (C * C);   // the Hessians of the chunk sit in LDS too
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  const int64_t nchunks = (S + ns - 1) / ns;
  *NS = ns;
  *grid = (int)min((int64_t)1024, nchunks);
  size_t doubles = (size_t)ns * C * D + 2 * (size_t)ns * C + (size_t)ns * C * C + (mode == 0 ? (size_t)D : (size_t)ns * D);
  if (doubles < 2048) doubles = 2048;   // the final reduction of mode 0 reuses the buffer as [4][64 NW]
  *lds = doubles * sizeof(double);
  return true;
}
static void rows_hjv_launch(int mode, const double* J, const double* H, const double* x, double* out, int64_t S, int C,
                            int D, int NS, int grid, size_t lds, hipStream_t st) {
  void (*fn)(const double*, const double*, const double*, double*, int64_t, int, int, int);
  if (D <= 32)
    fn = mode == 2 ? rows_hjv_kernel<2, 8> : (mode ? rows_hjv_kernel<1, 8> : rows_hjv_kernel<0, 8>);
  else if (D <= 64)
    fn = mode == 2 ? rows_hjv_kernel<2, 16> : (mode ? rows_hjv_kernel<1, 16> : rows_hjv_kernel<0, 16>);
  else if (D <= 128)
    fn = mode == 2 ? rows_hjv_kernel<2, 32> : (mode ? rows_hjv_kernel<1, 32> : rows_hjv_kernel<0, 32>);
  else   // explicit rows of a filter node with a convolution bond (4 x 17 x 4 = 272 entries)
    fn = mode == 2 ? rows_hjv_kernel<2, 64, 8> : rows_hjv_kernel<0, 64, 8>;
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, st, J, H, x, out, S, C, D, NS);
}

int64_t ttk_rows_matvec_ws_bytes(int64_t S, int C, int D) {
  int NS, grid;
  size_t lds;
  if (!rows_hjv_dims(S, C, D, 0, &NS, &grid, &lds)) return TTK_EINVAL;
  return (int64_t)grid * D * (int64_t)sizeof(double);
}

int ttk_rows_matvec_f64(const double* J, const double* H, const double* v, double* q, int64_t S, int C, int D, void* ws,
                        int64_t ws_bytes, void* stream) {
  if (!J || !H || !v || !q || !ws) return TTK_EINVAL;
  int NS, grid;
  size_t lds;
  if (!rows_hjv_dims(S, C, D, 0, &NS, &grid, &lds)) return TTK_EINVAL;
  if (ws_bytes < (int64_t)grid * D * (int64_t)sizeof(double)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  rows_hjv_launch(0, J, H, v, (double*)ws, S, C, D, NS, grid, lds, st);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)ws, grid,
                     (int64_t)D, (int64_t)D, q, 0);
  return ttk_launch_status();
}

int ttk_rows_rhs_f64(const double* J, const double* g, double* b, int64_t S, int C, int D, void* ws, int64_t ws_bytes,
                     void* stream) {
  if (!J || !g || !b || !ws) return TTK_EINVAL;
  int NS, grid;
  size_t lds;
  if (!rows_hjv_dims(S, C, D, 0, &NS, &grid, &lds)) return TTK_EINVAL;
  if (ws_bytes < (int64_t)grid * D * (int64_t)sizeof(double)) return TTK_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  rows_hjv_launch(2, J, J, g, (double*)ws, S, C, D, NS, grid, lds, st);
  int rc = ttk_launch_status();
  if (rc != TTK_OK) return rc;
  hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, st, (const double*)ws, grid,
                     (int64_t)D, (int64_t)D, b, 0);
  return ttk_launch_status();
}

int ttk_lhl_f64(const double* L, const double* H, const double* z, double* GL, int64_t S, int C, int rl, void* stream) {
  if (!L || !H || !z || !GL) return TTK_EINVAL;
  int NS, grid;
  size_t lds;
  if (!rows_hjv_dims(S, C, rl, 1, &NS, &grid, &lds)) return TTK_EINVAL;
  rows_hjv_launch(1, L, H, z, GL, S, C, rl, NS, grid, lds, (hipStream_t)stream);
  return ttk_launch_status();
}

int ttk_rows_dot_f64(const double* J, const double* v, double* out, int64_t rows, int D, void* stream) {
  if (!J || !v || !out || rows < 1 || D < 1) return TTK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 32) {
    const int64_t waves = (rows + 7) / 8;
    hipLaunchKernelGGL(rows_dot_kernel<8>, dim3((unsigned)min((int64_t)4096, (waves + 3) / 4)), dim3(256), 0, st, J, v, out,
                       rows, D);
  } else if (D <= 64) {
    const int64_t waves = (rows + 3) / 4;
    hipLaunchKernelGGL(rows_dot_kernel<16>, dim3((unsigned)min((int64_t)4096, (waves + 3) / 4)), dim3(256), 0, st, J, v, out,
                       rows, D);
  } else {
    hipLaunchKernelGGL(rows_dot_kernel<64>, dim3((unsigned)min((int64_t)4096, (rows + 3) / 4)), dim3(256), 0, st, J, v, out,
                       rows, D);
  }
  return ttk_launch_status();
}

}  // extern "C"
