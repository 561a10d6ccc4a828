// gemm.hip -- MFMA GEMM for gfx950 (MI355X).
//
// C[M,N] = op(A)[M,K] * op(B)[K,N] + beta*C, row-major, arbitrary M/N/K/strides.
// replaces: nn.Linear(2H,V,bias=False) (reference timit/models/model_ctc.py:137,166) and the time-parallel
// input projections X*W_ih^T of nn.LSTM/GRU/RNN (model_ctc.py:33), plus their dgrad/wgrad GEMMs.
//
// Design (wave64, 4 SIMDs/CU):
//   * 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave a 64x64 sub-tile =
//     2x2 MFMA tiles of v_mfma_f32_32x32x2_f32 (exact f32 == fmaf chain) -> 64 accumulator VGPRs.
//   * BK=16 K-slab staged through LDS in k-major order ([k][m] / [k][n]) so that a wave's MFMA operand
//     fetch (lane l reads element (m = l&31, k = l>>5)) is a conflict-free ds_read_b32 of 32 consecutive
//     floats per half-wave; row pad +4 floats keeps the transposing stores <= 2-way conflicted.
//   * global -> register -> LDS staging with the next slab's global loads issued before the MFMAs of the
//     current slab (register prefetch), float4 (16 B/lane) global loads when the operand is 16-B aligned.
//   * deterministic split-K (grid.z) through a caller-provided workspace + a second reduce pass, used when
//     the M*N tile grid alone cannot fill the 256 CUs (weight-gradient GEMMs, K = T*B = 25 600).
//   * XCD-aware tile order: consecutive workgroup ids are dealt round-robin to the 8 XCDs, so the remap
//     gives each XCD a contiguous band of M-tiles that share the same B panel in its private L2.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PAD = 4;

// ---- global -> registers (8 floats per thread per operand slab) --------------------------------
// KC: operand element (r, k) at src[(r0+r)*ld + k0+k]  (k contiguous)
__device__ __forceinline__ void g2r_kc(const float *__restrict__ src, int ld, int r0, int R, int k0, int Kend,
                                       bool vec, int tid, float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j;
    const int r = idx >> 2, kq = (idx & 3) * 4;
    const int gr = r0 + r, gk = k0 + kq;
    const float *p = src + (size_t)gr * ld + gk;
    if (vec && gr < R && gk + 3 < Kend) {
      const float4 v = *reinterpret_cast<const float4 *>(p);
      reg[4 * j + 0] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) reg[4 * j + c] = (gr < R && gk + c < Kend) ? p[c] : 0.0f;
    }
  }
}
__device__ __forceinline__ void r2s_kc(float (*s)[BM + PAD], int tid, const float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j;
    const int r = idx >> 2, kq = (idx & 3) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) s[kq + c][r] = reg[4 * j + c];
  }
}
// MC: operand element (r, k) at src[(k0+k)*ld + r0+r]  (r contiguous)
__device__ __forceinline__ void g2r_mc(const float *__restrict__ src, int ld, int r0, int R, int k0, int Kend,
                                       bool vec, int tid, float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j;
    const int k = idx >> 5, rq = (idx & 31) * 4;
    const int gk = k0 + k, gr = r0 + rq;
    const float *p = src + (size_t)gk * ld + gr;
    if (vec && gk < Kend && gr + 3 < R) {
      const float4 v = *reinterpret_cast<const float4 *>(p);
      reg[4 * j + 0] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) reg[4 * j + c] = (gk < Kend && gr + c < R) ? p[c] : 0.0f;
    }
  }
}
__device__ __forceinline__ void r2s_mc(float (*s)[BM + PAD], int tid, const float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j;
    const int k = idx >> 5, rq = (idx & 31) * 4;
    *reinterpret_cast<float4 *>(&s[k][rq]) = make_float4(reg[4 * j], reg[4 * j + 1], reg[4 * j + 2], reg[4 * j + 3]);
  }
}

// TA: A stored [K][M] (m contiguous).  TB: B stored [N][K] (k contiguous).
// one 128x128 output tile (`bid`) of K-split `split`
template <bool TA, bool TB>
__device__ __forceinline__ void f32_tile(float (*sA)[BM + PAD], float (*sB)[BN + PAD], int bid, int split, int M, int N, int K,
                                         const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, float *__restrict__ C,
                                         int ldc, float beta, int kchunk, float *__restrict__ ws, int tiles_n, bool vecA, bool vecB) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;   // consecutive ids walk N first: share the A panel
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * kchunk;
  const int kend = min(K, kbeg + kchunk);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  float ra[8], rb[8];
  auto loadA = [&](int k0) {
    if (TA) g2r_mc(A, lda, m0, M, k0, kend, vecA, tid, ra);
    else g2r_kc(A, lda, m0, M, k0, kend, vecA, tid, ra);
  };
  auto loadB = [&](int k0) {
    if (TB) g2r_kc(B, ldb, n0, N, k0, kend, vecB, tid, rb);
    else g2r_mc(B, ldb, n0, N, k0, kend, vecB, tid, rb);
  };
  if (kbeg < kend) {
    loadA(kbeg);
    loadB(kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();   // previous slab fully consumed
    if (TA) r2s_mc(sA, tid, ra); else r2s_kc(sA, tid, ra);
    if (TB) r2s_kc(sB, tid, rb); else r2s_mc(sB, tid, rb);
    __syncthreads();
    if (k0 + BK < kend) {   // prefetch next slab into registers while the MFMAs run
      loadA(k0 + BK);
      loadB(k0 + BK);
    }
    const int kl = lane >> 5, ml = lane & 31;
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
      const float a0 = sA[k + kl][wm * 64 + ml], a1 = sA[k + kl][wm * 64 + 32 + ml];
      const float b0 = sB[k + kl][wn * 64 + ml], b1 = sB[k + kl][wn * 64 + 32 + ml];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float *out = ws ? ws + (size_t)split * M * N : C;
  const int ldo = ws ? N : ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][e];
          float *p = out + (size_t)row * ldo + col;
          if (!ws && beta != 0.0f) v += beta * *p;
          *p = v;
        }
      }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                       const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                       int ldc, float beta, int kchunk, float *__restrict__ ws,
                                                       int tiles_m, int tiles_n, bool vecA, bool vecB) {
  __shared__ __attribute__((aligned(16))) float sA[BK][BM + PAD];
  __shared__ __attribute__((aligned(16))) float sB[BK][BN + PAD];
  // XCD-aware remap (bijective for any tile count): blocks b, b+8, b+16.. share an XCD/L2.
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  f32_tile<TA, TB>(sA, sB, bid, blockIdx.y, M, N, K, A, lda, B, ldb, C, ldc, beta, kchunk, ws, tiles_n, vecA, vecB);
}

// XCD-filtered variant for the weight-gradient side stream (see gemm_planes_nt_queue_kernel)
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_queue_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                             const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                                             float beta, int kchunk, float *__restrict__ ws, int tiles_m, int tiles_n,
                                                             bool vecA, bool vecB, int splits, unsigned xcd_allow,
                                                             unsigned *__restrict__ queue) {
  __shared__ __attribute__((aligned(16))) float sA[BK][BM + PAD];
  __shared__ __attribute__((aligned(16))) float sB[BK][BN + PAD];
  __shared__ int s_item;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (!((xcd_allow >> (x & 15)) & 1u)) return;
  const int nt = tiles_m * tiles_n, total = nt * splits;
  for (;;) {
    if (threadIdx.x == 0) s_item = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int item = s_item;
    if (item >= total) return;
    f32_tile<TA, TB>(sA, sB, item % nt, item / nt, M, N, K, A, lda, B, ldb, C, ldc, beta, kchunk, ws, tiles_n, vecA, vecB);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// precision = 1: split-operand bf16 MFMA ("bf16x3").  Every f32 operand is split on the fly into
// hi = bf16(x), lo = bf16(x - hi) while it is staged into LDS, and the product is accumulated in f32 as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 (3 MFMAs = 96 cycles per 32x32x16 instead of
// 8 x 64 = 512 cycles of f32 MFMA).  16 mantissa bits per operand -> relative error ~2^-16 per product, i.e.
// f32-class results (max-abs 1e-5 on unit-scale data) at ~5x the f32-MFMA rate.
// LDS image: [row][k] with k contiguous (8 bf16 = one ds_read_b128 MFMA fragment), row stride 40 bf16 = 80 B so the
// 16 lanes of a read group land on 16 distinct 16-B bank slots.
// ------------------------------------------------------------------------------------------------
constexpr int XBK = 32, XLD = 40;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16(float x, unsigned short &hi, unsigned short &lo) {
  hi = f2bf(x);
  lo = f2bf(x - __uint_as_float((unsigned)hi << 16));
}

// KC: element (r,k) at src[(r0+r)*ld + k0+k];  16 floats per thread: 4 x float4 along k
__device__ __forceinline__ void x_g2r_kc(const float *__restrict__ src, int ld, int r0, int R, int k0, int Kend, bool vec,
                                         int tid, float (&reg)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j;
    const int r = idx >> 3, kq = (idx & 7) * 4;
    const int gr = r0 + r, gk = k0 + kq;
    const float *p = src + (size_t)gr * ld + gk;
    if (vec && gr < R && gk + 3 < Kend) {
      const float4 v = *reinterpret_cast<const float4 *>(p);
      reg[4 * j] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) reg[4 * j + c] = (gr < R && gk + c < Kend) ? p[c] : 0.0f;
    }
  }
}
__device__ __forceinline__ void x_r2s_kc(unsigned short (*sh)[XLD], unsigned short (*sl)[XLD], int tid, const float (&reg)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j;
    const int r = idx >> 3, kq = (idx & 7) * 4;
    unsigned short h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) split_bf16(reg[4 * j + c], h[c], l[c]);
    *reinterpret_cast<uint2 *>(&sh[r][kq]) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2 *>(&sl[r][kq]) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}
// MC: element (r,k) at src[(k0+k)*ld + r0+r];  4 x float4 along r, transposed into the [row][k] image
__device__ __forceinline__ void x_g2r_mc(const float *__restrict__ src, int ld, int r0, int R, int k0, int Kend, bool vec,
                                         int tid, float (&reg)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j;
    const int k = idx >> 5, rq = (idx & 31) * 4;
    const int gk = k0 + k, gr = r0 + rq;
    const float *p = src + (size_t)gk * ld + gr;
    if (vec && gk < Kend && gr + 3 < R) {
      const float4 v = *reinterpret_cast<const float4 *>(p);
      reg[4 * j] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) reg[4 * j + c] = (gk < Kend && gr + c < R) ? p[c] : 0.0f;
    }
  }
}
__device__ __forceinline__ void x_r2s_mc(unsigned short (*sh)[XLD], unsigned short (*sl)[XLD], int tid, const float (&reg)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j;
    const int k = idx >> 5, rq = (idx & 31) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned short h, l;
      split_bf16(reg[4 * j + c], h, l);
      sh[rq + c][k] = h;
      sl[rq + c][k] = l;
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                          const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                          int ldc, float beta, int kchunk, float *__restrict__ ws,
                                                          int tiles_m, int tiles_n, bool vecA, bool vecB) {
  __shared__ __attribute__((aligned(16))) unsigned short sAh[BM][XLD], sAl[BM][XLD], sBh[BN][XLD], sBl[BN][XLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * kchunk;
  const int kend = min(K, kbeg + kchunk);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  float ra[16], rb[16];
  auto loadA = [&](int k0) {
    if (TA) x_g2r_mc(A, lda, m0, M, k0, kend, vecA, tid, ra);
    else x_g2r_kc(A, lda, m0, M, k0, kend, vecA, tid, ra);
  };
  auto loadB = [&](int k0) {
    if (TB) x_g2r_kc(B, ldb, n0, N, k0, kend, vecB, tid, rb);
    else x_g2r_mc(B, ldb, n0, N, k0, kend, vecB, tid, rb);
  };
  if (kbeg < kend) { loadA(kbeg); loadB(kbeg); }
  for (int k0 = kbeg; k0 < kend; k0 += XBK) {
    __syncthreads();
    if (TA) x_r2s_mc(sAh, sAl, tid, ra); else x_r2s_kc(sAh, sAl, tid, ra);
    if (TB) x_r2s_kc(sBh, sBl, tid, rb); else x_r2s_mc(sBh, sBl, tid, rb);
    __syncthreads();
    if (k0 + XBK < kend) { loadA(k0 + XBK); loadB(k0 + XBK); }
    const int ml = lane & 31, kg = (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < XBK; ks += 16) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8_t *>(&sAh[wm * 64 + i * 32 + ml][ks + kg]);
        al[i] = *reinterpret_cast<const bf16x8_t *>(&sAl[wm * 64 + i * 32 + ml][ks + kg]);
        bh[i] = *reinterpret_cast<const bf16x8_t *>(&sBh[wn * 64 + i * 32 + ml][ks + kg]);
        bl[i] = *reinterpret_cast<const bf16x8_t *>(&sBl[wn * 64 + i * 32 + ml][ks + kg]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  float *out = ws ? ws + (size_t)blockIdx.y * M * N : C;
  const int ldo = ws ? N : ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][e];
          float *p = out + (size_t)row * ldo + col;
          if (!ws && beta != 0.0f) v += beta * *p;
          *p = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// precision = 1, main path: bf16x3 on PRE-SPLIT operand planes.
//   1. one HBM-bound pass per operand splits the f32 matrix into hi/lo bf16 planes laid out [rows][Kp] with the
//      contraction index contiguous and zero-padded to a multiple of 64 (the pass transposes when the operand is
//      stored contraction-major), so that
//   2. a single "NT" kernel  C[m,n] = sum_k A[m,k] * B[n,k]  serves all four transpose cases with no conversion and
//      no transposition in its inner loop: 128x128x64 tiles, 16-B global loads -> XOR-swizzled LDS rows of 128 B
//      (conflict-free ds_write_b128 / ds_read_b128: slot = 8*(row&1) + (chunk ^ ((row>>1)&7))), register prefetch of
//      the next K-slab, 3 x v_mfma_f32_32x32x16_bf16 per fragment pair (lo*hi, hi*lo, hi*hi), f32 accumulate.
// ------------------------------------------------------------------------------------------------
constexpr int PBK = 64;

__global__ void split_rows_kernel(const float *__restrict__ src, int ld, int R, int C, int Cp, unsigned short *__restrict__ hi,
                                  unsigned short *__restrict__ lo) {
  // out[r][c] (row stride Cp) = split(src[r*ld + c]), zero for c >= C; 4 elements per thread
  const size_t total = (size_t)R * (Cp / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (Cp / 4)), c = (int)(i - (size_t)r * (Cp / 4)) * 4;
    unsigned short h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = (c + k < C) ? src[(size_t)r * ld + c + k] : 0.0f;
      split_bf16(v, h[k], l[k]);
    }
    *reinterpret_cast<uint2 *>(hi + (size_t)r * Cp + c) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2 *>(lo + (size_t)r * Cp + c) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

// out[c][r] (row stride Rp) = split(src[r*ld + c]); 64x64 tiles through LDS; zero for r >= R (k padding)
// shift: output row k takes source row k - shift when that lies in [0, R), zero otherwise (a matrix delayed / advanced by
// |shift| contraction steps: h_prev of the recurrent weight gradient)
__device__ __forceinline__ void split_transpose_tile(float (*t)[65], int bx, int by, const float *__restrict__ src, int ld, int R, int C, int Rp,
                                                     unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, int shift) {
  const int r0 = by * 64, c0 = bx * 64;
  const int tid = threadIdx.x;
  // fast path (every large operand): 16-B global loads along the columns, 16-B plane stores along the rows (8 bf16 of one
  // output row per lane: a wave writes eight 128-B runs), LDS image padded to 65 floats so that both the row-wise scalar
  // writes and the column-wise reads are conflict-free.  8 global instructions per thread instead of 48.
  const bool vec = c0 + 64 <= C && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0;
  if (vec) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j, i = idx >> 4, c4 = idx & 15;
      const int r = r0 + i - shift;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 0 && r < R && r0 + i < R) v = *reinterpret_cast<const float4 *>(src + (size_t)r * ld + c0 + 4 * c4);
      t[i][4 * c4 + 0] = v.x; t[i][4 * c4 + 1] = v.y; t[i][4 * c4 + 2] = v.z; t[i][4 * c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = tid + 256 * j, c = g >> 3, ro = g & 7;
      unsigned hw[4], lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h0, l0, h1, l1;
        split_bf16(t[ro * 8 + 2 * e][c], h0, l0);
        split_bf16(t[ro * 8 + 2 * e + 1][c], h1, l1);
        hw[e] = (unsigned)h0 | ((unsigned)h1 << 16);
        lw[e] = (unsigned)l0 | ((unsigned)l1 << 16);
      }
      const size_t o = (size_t)(c0 + c) * Rp + r0 + ro * 8;
      *reinterpret_cast<uint4 *>(hi + o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4 *>(lo + o) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    return;
  }
  const int tx = tid & 63, ty = tid >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i - shift, c = c0 + tx;
    t[i][tx] = (r >= 0 && r < R && r0 + i < R && c < C) ? src[(size_t)r * ld + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rp) {
      unsigned short h, l;
      split_bf16(t[tx][i], h, l);
      hi[(size_t)c * Rp + r] = h;
      lo[(size_t)c * Rp + r] = l;
    }
  }
}

__global__ __launch_bounds__(256) void split_transpose_kernel(const float *__restrict__ src, int ld, int R, int C, int Rp,
                                                              unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, int shift) {
  __shared__ float t[64][65];
  split_transpose_tile(t, blockIdx.x, blockIdx.y, src, ld, R, C, Rp, hi, lo, shift);
}
// XCD-filtered variant (see gemm_planes_nt_queue_kernel): tiles come from an atomic queue, workgroups off `xcd_allow` exit
__global__ __launch_bounds__(256) void split_transpose_queue_kernel(const float *__restrict__ src, int ld, int R, int C, int Rp,
                                                                    unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, int tiles_x,
                                                                    int tiles_y, unsigned xcd_allow, unsigned *__restrict__ queue, int shift) {
  __shared__ float t[64][65];
  __shared__ int s_item;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (!((xcd_allow >> (x & 15)) & 1u)) return;
  for (;;) {
    if (threadIdx.x == 0) s_item = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int item = s_item;
    if (item >= tiles_x * tiles_y) return;
    split_transpose_tile(t, item % tiles_x, item / tiles_x, src, ld, R, C, Rp, hi, lo, shift);
    __syncthreads();
  }
}

__device__ __forceinline__ int pswz(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }   // byte offset in a plane tile

// One (64*TI) x (64*TJ) output tile (`bid`) of K-split `split`.  2 x 2 waves, each owning (32*TI) x (32*TJ) = TI x TJ
// MFMA tiles: per 16-k step a wave reads 2*(TI + TJ) fragments from LDS for 3*TI*TJ MFMAs, so the LDS traffic per MFMA
// falls from 0.67 fragments (TI = TJ = 2) to 0.5 (2 x 4 / 4 x 2) -- the 128x128 kernel is LDS-bound, not MFMA-bound.
// sm = dynamic LDS: [Ah | Al] (64*TI rows x 128 B each) then [Bh | Bl] (64*TJ rows x 128 B each).
// Measured and dropped: 32-k stages (32 KB of LDS, 156 VGPRs: three workgroups per CU instead of two) -- bit-identical, 1-4 %
// slower (dx 25 600 x 640 x 2 560: 436 vs 418 us with its split passes), so occupancy is not what holds the 128x128 tile at
// ~36 % of the MFMA rate; PMC (profiles/r01_pmc_gemm_probe_v10.txt): the A planes cross the fabric 2.3 x per XCD.
template <int TI, int TJ>
__device__ __forceinline__ void plane_tile(unsigned char *sm, int bid, int split, int M, int N, int Kp,
                                           const unsigned short *__restrict__ Ah, const unsigned short *__restrict__ Al,
                                           const unsigned short *__restrict__ Bh, const unsigned short *__restrict__ Bl,
                                           float *__restrict__ C, int ldc, float beta, int kchunk, float *__restrict__ ws, int tiles_n) {
  constexpr int TBM = 64 * TI, TBN = 64 * TJ;
  constexpr int CA = TBM * 8 / 256, CB = TBN * 8 / 256;              // 16-B staging chunks per thread and plane
  unsigned char *sAh = sm, *sAl = sm + TBM * 128, *sBh = sm + 2 * TBM * 128, *sBl = sBh + TBN * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;
  const int kbeg = split * kchunk;
  const int kend = min(Kp, kbeg + kchunk);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // element offsets of this thread's chunks in the A-side and B-side planes (rows clamped: out-of-range rows are never stored)
  unsigned offA[CA], offB[CB];
#pragma unroll
  for (int j = 0; j < CA; ++j) offA[j] = (unsigned)min(m0 + ((tid + 256 * j) >> 3), M - 1) * (unsigned)Kp + ((tid + 256 * j) & 7) * 8;
#pragma unroll
  for (int j = 0; j < CB; ++j) offB[j] = (unsigned)min(n0 + ((tid + 256 * j) >> 3), N - 1) * (unsigned)Kp + ((tid + 256 * j) & 7) * 8;
  u32x4 pAh[CA], pAl[CA], pBh[CB], pBl[CB];      // ext-vector values (arrays of HIP's uint4 class end up in scratch)
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      pAh[j] = *reinterpret_cast<const u32x4 *>(Ah + offA[j] + k0);
      pAl[j] = *reinterpret_cast<const u32x4 *>(Al + offA[j] + k0);
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      pBh[j] = *reinterpret_cast<const u32x4 *>(Bh + offB[j] + k0);
      pBl[j] = *reinterpret_cast<const u32x4 *>(Bl + offB[j] + k0);
    }
  };
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += PBK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      const int so = pswz((tid + 256 * j) >> 3, (tid + 256 * j) & 7);
      *reinterpret_cast<u32x4 *>(sAh + so) = pAh[j];
      *reinterpret_cast<u32x4 *>(sAl + so) = pAl[j];
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int so = pswz((tid + 256 * j) >> 3, (tid + 256 * j) & 7);
      *reinterpret_cast<u32x4 *>(sBh + so) = pBh[j];
      *reinterpret_cast<u32x4 *>(sBl + so) = pBl[j];
    }
    __syncthreads();
    if (k0 + PBK < kend) gload(k0 + PBK);
    const int ml = lane & 31, g = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + g;
      bf16x8_t ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int ra = wm * 32 * TI + i * 32 + ml;
        ah[i] = *reinterpret_cast<const bf16x8_t *>(sAh + pswz(ra, kc));
        al[i] = *reinterpret_cast<const bf16x8_t *>(sAl + pswz(ra, kc));
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int rb = wn * 32 * TJ + j * 32 + ml;
        bh[j] = *reinterpret_cast<const bf16x8_t *>(sBh + pswz(rb, kc));
        bl[j] = *reinterpret_cast<const bf16x8_t *>(sBl + pswz(rb, kc));
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  float *out = ws ? ws + (size_t)split * M * N : C;
  const int ldo = ws ? N : ldc;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * 32 * TJ + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * TI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][e];
          float *p = out + (size_t)row * ldo + col;
          if (!ws && beta != 0.0f) v += beta * *p;
          *p = v;
        }
      }
    }
}

template <int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_planes_nt_kernel(int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                             const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                             const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc, float beta,
                                                             int kchunk, float *__restrict__ ws, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];       // Ah | Al | Bh | Bl tiles
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  plane_tile<TI, TJ>(psm, bid, blockIdx.y, M, N, Kp, Ah, Al, Bh, Bl, C, ldc, beta, kchunk, ws, tiles_n);
}

// Same GEMM for a side stream that must stay off the XCDs a persistent recurrence is running on: workgroups that wake up
// on an XCD outside `xcd_allow` (bit x = XCD x, read from HW_REG_XCC_ID) exit at once, the others pull (tile, split)
// items from an atomic queue until it is empty.  The launch is sized to fill the allowed XCDs, not the tile count.
template <int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_planes_nt_queue_kernel(int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                                   const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                                   const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc,
                                                                   float beta, int kchunk, float *__restrict__ ws, int tiles_m, int tiles_n,
                                                                   int splits, unsigned xcd_allow, unsigned *__restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  __shared__ int s_item;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (!((xcd_allow >> (x & 15)) & 1u)) return;
  const int nt = tiles_m * tiles_n, total = nt * splits;
  for (;;) {
    if (threadIdx.x == 0) s_item = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int item = s_item;
    if (item >= total) return;
    plane_tile<TI, TJ>(psm, item % nt, item / nt, M, N, Kp, Ah, Al, Bh, Bl, C, ldc, beta, kchunk, ws, tiles_n);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 / 256 x 128 plane tiles for the activation-sized products (M = T*B rows).
// The 128 x 128 tile above moves 64 KB of planes through a CU per 64-k slab for 3 x 128 x 128 x 64 MACs: at the MFMA rate that
// is ~42 B/clk per CU out of an L2 that delivers ~56, so the kernel is L2-bandwidth bound at ~36 % of the MFMA rate (and the A
// planes cross the fabric 2.3 x per XCD).  A 256 x 256 tile halves the bytes per MAC (21 B/clk), a 256 x 128 tile takes 31.
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 (or 128 x 32) outputs = 4 x WNT MFMA tiles of 32 x 32: 128 / 64
//     accumulator VGPRs; per 16-k step 8 + 4 (or 8 + 2) ds_read_b128 feed 24 (12) v_mfma_f32_32x32x16_bf16;
//   * 32-k stages, two LDS buffers (2 x 64 KB / 2 x 48 KB): the next stage is copied global -> LDS by the DMA path
//     (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) while the current one is multiplied; ONE barrier per stage;
//   * LDS image: rows of 64 B (4 chunks of 16 B = 8 bf16), chunk' = chunk ^ ((row >> 2) & 3): the 16 lanes of every ds_read_b128
//     group hit 16 distinct bank slots.  The DMA writes LDS lane-linearly, so the swizzle is applied to the SOURCE address;
//   * same products, same k order inside a 16-k step and across steps as the 128 x 128 tile: bit-identical results.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool v_is_magic(float v) { return v == 1.2345678e33f; }   // (debug switch of tools/gemm_bench.py: keeps the accumulators live)
__device__ __forceinline__ int qswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int WNT>      // 32-column MFMA tiles per wave: 2 -> 256 x 256 workgroup tile, 1 -> 256 x 128
__global__ __launch_bounds__(512) void gemm_planes_nt256_kernel(int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                                const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                                const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc, float beta,
                                                                int tiles_m, int tiles_n, int dbg) {
  constexpr int TBM = 256, TBN = 128 * WNT;
  constexpr int A_BYTES = TBM * 64, B_BYTES = TBN * 64;              // one plane of one stage
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;                   // Ah | Al | Bh | Bl
  constexpr int IA = TBM * 4 / 512, IB = TBN * 4 / 512;              // DMA instructions per thread and plane
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {   // XCD-aware remap: blocks b, b + 8, ... share an XCD / L2; give each XCD a contiguous band of tiles (N fastest: shared A panel)
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x16 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // DMA source offsets (elements) of this lane: LDS position p = (inst * 8 + wave) * 64 + lane -> row p >> 2, swizzled chunk p & 3
  unsigned offA[IA], offB[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offA[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)Kp + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offB[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)Kp + chunk * 8;
  }
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  auto issue = [&](int k0, int buf) {
    unsigned char *sb = qsm + buf * STAGE;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int dst = (i * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ah + offA[i] + k0), (lptr_t)(sb + dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Al + offA[i] + k0), (lptr_t)(sb + A_BYTES + dst), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int dst = (i * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Bl + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + B_BYTES + dst), 16, 0, 0);
    }
  };
  const int nst = Kp / 32;
  const int ml = lane & 31, g = lane >> 5;
  // fragment loads of one 16-k half: 8 + 2*WNT ds_read_b128
  auto load_frags = [&](const unsigned char *sb, int ks, bf16x8_t (&ah)[4], bf16x8_t (&al)[4], bf16x8_t (&bh)[WNT], bf16x8_t (&bl)[WNT]) {
    const int cq = ks * 2 + g;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int o = qswz(wn * 32 * WNT + j * 32 + ml, cq);
      bh[j] = *reinterpret_cast<const bf16x8_t *>(sb + 2 * A_BYTES + o);
      bl[j] = *reinterpret_cast<const bf16x8_t *>(sb + 2 * A_BYTES + B_BYTES + o);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = qswz(wm * 128 + i * 32 + ml, cq);
      ah[i] = *reinterpret_cast<const bf16x8_t *>(sb + o);
      al[i] = *reinterpret_cast<const bf16x8_t *>(sb + A_BYTES + o);
    }
  };
  auto multiply = [&](const bf16x8_t (&ah)[4], const bf16x8_t (&al)[4], const bf16x8_t (&bh)[WNT], const bf16x8_t (&bl)[WNT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };
  issue(0, 0);
  for (int s = 0; s < nst; ++s) {
    __syncthreads();                       // stage s has landed (the DMA is waited for here), everybody is done with stage s - 1
    const unsigned char *sb = qsm + (s & 1) * STAGE;
    // Both 16-k halves of the stage are requested from LDS up front (two register sets): the second half's reads return behind
    // the first half's 24 MFMAs instead of behind an s_waitcnt in front of every second MFMA group (the compiler's own
    // schedule waited nine times per stage with 2-4 reads in flight)
    bf16x8_t ah0[4], al0[4], bh0[WNT], bl0[WNT], ah1[4], al1[4], bh1[WNT], bl1[WNT];
    load_frags(sb, 0, ah0, al0, bh0, bl0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nst && !(dbg & 2)) issue((s + 1) * 32, (s + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(sb, 1, ah1, al1, bh1, bl1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(ah0, al0, bh0, bl0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(ah1, al1, bh1, bl1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N && (!(dbg & 1) || v_is_magic(acc[i][j][e]))) {
          float v = acc[i][j][e];
          float *p = C + (size_t)row * ldc + col;
          if (beta != 0.0f) v += beta * *p;
          if (dbg & 4) __builtin_nontemporal_store(v, p); else *p = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Ping-pong schedule of the same tile (MI355X_MICROARCH.md "Two waves per SIMD"): the workgroup's two waves on a SIMD share that
// SIMD's matrix pipe, so when all eight waves run the same phase (barrier -> LDS reads -> 48 MFMAs) the pipe idles through every
// read / DMA-issue phase.  Here the wave halves (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; one of each per SIMD) run
// half a 16-k step apart: while one half issues its 24 MFMAs the other fetches its next fragments from LDS and issues its share
// of the next stage's DMA; an s_barrier separates the phases (4 per 32-k stage).  Per stage s:
//     phase 4s     A: read fragments (s, k-half 0)                              B: multiply (s-1, 1)
//     phase 4s+1   A: multiply (s, 0) + DMA share of stage s+1                  B: read (s, 0)
//     phase 4s+2   A: read (s, 1), own DMA pieces landed                        B: multiply (s, 0) + DMA share of s+1
//     phase 4s+3   A: multiply (s, 1)                                           B: read (s, 1), own DMA pieces landed
// Buffer (s+1)&1 is rewritten from phase 4s+1 on: its last readers were phases 4s-2 (A) and 4s-1 (B); it is first read in 4s+4.
// Same products in the same order per accumulator: bit-identical to the other plane tiles.
// ------------------------------------------------------------------------------------------------
// (sched_barrier: MFMAs are register-only instructions, a "memory" clobber alone lets the scheduler carry them across the barrier
// into the other phase -- which is exactly what the phases exist to prevent)
__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_barrier_vm() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef CTCN_GEMM_STATS
// development (tools/mb_gemm_pp.hip): per barrier site of the main loop, the cycles a wave spends before it reaches the barrier (its phase's own
// work) and inside it (waiting for the other waves), summed over the stages of one tile: waves 0 and 4 of blocks 0 and 1000
__device__ long long *g_gemm_stats;
#define PPB_SITE(i, vm)                                                                                                   \
  do {                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (vm) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    const long long ta_ = clock64();                                                                                      \
    asm volatile("s_barrier" ::: "memory");                                                                               \
    const long long tb_ = clock64();                                                                                      \
    gs_work[i] += ta_ - gs_prev; gs_wait[i] += tb_ - ta_; gs_prev = tb_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  } while (0)
#define PPB(i) PPB_SITE(i, false)
#define PPB_VM(i) PPB_SITE(i, true)
#else
#define PPB(i) pp_barrier()
#define PPB_VM(i) pp_barrier_vm()
#endif
// one 256 x (128 * WNT) output tile (tile number `bid`, row-major over tiles_m x tiles_n) in the ping-pong schedule
// NP = 3: the bf16x3 products (al*bh, ah*bl, ah*bh); NP = 1 (option "gemm_bf16_single"): ah*bh only -- plain bf16 operands, f32 accumulate: the lo
// planes are neither copied nor read (the LDS image keeps its layout, the lo halves of a stage stay unused)
template <int WNT, int NP = 3>
__device__ __forceinline__ void planes256pp_tile(unsigned char *qsm, int bid, int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                 const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                 const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc, float beta, int tiles_n, int dbg) {
  constexpr int TBM = 256, TBN = 128 * WNT;
  constexpr int A_BYTES = TBM * 64, B_BYTES = TBN * 64;
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  constexpr int IA = TBM * 4 / 512, IB = TBN * 4 / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;
#ifdef CTCN_GEMM_STATS
  const long long gs_entry = clock64();
#endif

  f32x16 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  unsigned offA[IA], offB[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offA[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)Kp + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offB[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)Kp + chunk * 8;
  }
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  auto issue = [&](int k0, int buf) {
    unsigned char *sb = qsm + buf * STAGE;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int dst = (i * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ah + offA[i] + k0), (lptr_t)(sb + dst), 16, 0, 0);
      if constexpr (NP == 3) __builtin_amdgcn_global_load_lds((gptr_t)(Al + offA[i] + k0), (lptr_t)(sb + A_BYTES + dst), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int dst = (i * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + dst), 16, 0, 0);
      if constexpr (NP == 3) __builtin_amdgcn_global_load_lds((gptr_t)(Bl + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + B_BYTES + dst), 16, 0, 0);
    }
  };
  const int nst = Kp / 32;
  const int ml = lane & 31, g = lane >> 5;
  bf16x8_t ah[4], al[4], bh[WNT], bl[WNT];
  // LDS byte offsets of this lane's fragments inside a stage, per 16-k half (loop-invariant)
  int oa[2][4], ob[2][WNT];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) oa[ks][i] = qswz(wm * 128 + i * 32 + ml, ks * 2 + g);
#pragma unroll
    for (int j = 0; j < WNT; ++j) ob[ks][j] = 2 * A_BYTES + qswz(wn * 32 * WNT + j * 32 + ml, ks * 2 + g);
  }
  auto load_frags = [&](int stage, int ks) {
    const unsigned char *sb = qsm + (stage & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8_t *>(sb + ob[ks][j]);
      if constexpr (NP == 3) bl[j] = *reinterpret_cast<const bf16x8_t *>(sb + B_BYTES + ob[ks][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8_t *>(sb + oa[ks][i]);
      if constexpr (NP == 3) al[i] = *reinterpret_cast<const bf16x8_t *>(sb + A_BYTES + oa[ks][i]);
    }
  };
  // one DMA piece (1 KB) of this wave's share of a stage: n = 0 .. NPIECE - 1
  auto issue_piece = [&](int n, int k0, int buf) {
    unsigned char *sb = qsm + buf * STAGE;
    if constexpr (NP == 3) {
      if (n < 2 * IA) {
        const int i = n >> 1, dst = (i * 8 + wave) * 1024;
        if (n & 1) __builtin_amdgcn_global_load_lds((gptr_t)(Al + offA[i] + k0), (lptr_t)(sb + A_BYTES + dst), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(Ah + offA[i] + k0), (lptr_t)(sb + dst), 16, 0, 0);
      } else {
        const int m = n - 2 * IA, i = m >> 1, dst = (i * 8 + wave) * 1024;
        if (m & 1) __builtin_amdgcn_global_load_lds((gptr_t)(Bl + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + B_BYTES + dst), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[i] + k0), (lptr_t)(sb + 2 * A_BYTES + dst), 16, 0, 0);
      }
    } else {
      if (n < IA) __builtin_amdgcn_global_load_lds((gptr_t)(Ah + offA[n] + k0), (lptr_t)(sb + (n * 8 + wave) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[n - IA] + k0), (lptr_t)(sb + 2 * A_BYTES + ((n - IA) * 8 + wave) * 1024), 16, 0, 0);
    }
  };
  constexpr int NPIECE = NP == 3 ? 2 * IA + 2 * IB : IA + IB, NMM = 4 * WNT;
  // 24 (12) MFMAs -- NP = 1: 8 (4) --; with `dma` the wave's DMA pieces of the next stage are issued between them (an LDS-DMA instruction costs the
  // issuing wave ~60 cycles among bare MFMAs and 100-185 inside a read phase: MI355X_MICROARCH.md), one piece per (i, j) tile
  auto multiply = [&](auto dma, int k0, int buf) {
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          // term order per accumulator: al*bh, ah*bl, ah*bh (small terms first) -- as in the other plane tiles
          if (NP == 3 && t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          else if (NP == 3 && t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          const int n = t * NMM + i * WNT + j;                 // running MFMA number; pieces after MFMAs 1, 3, 5, ... (NP = 1: after every MFMA)
          const int pc = NP == 3 ? ((n & 1) ? (n >> 1) : NPIECE) : n;
          if (pc < NPIECE) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(dma)::value) issue_piece(pc, k0, buf);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };
  // ONE instruction stream for both halves -- read (s,0) | multiply | read (s,1) | multiply, a barrier after each step -- which
  // half B enters one barrier late: at every moment one half multiplies while the other reads
  issue(0, 0);
  pp_barrier_vm();                              // stage 0 landed
  if (wm == 1) pp_barrier();
  constexpr std::integral_constant<bool, true> with_dma{};
  constexpr std::integral_constant<bool, false> no_dma{};
#ifdef CTCN_GEMM_STATS
  long long gs_work[4] = {0, 0, 0, 0}, gs_wait[4] = {0, 0, 0, 0};
  const long long gs_t0 = clock64();
  long long gs_prev = gs_t0;
#endif
  for (int s = 0; s + 1 < nst; ++s) {
    load_frags(s, 0);
    PPB(0);
    multiply(with_dma, (s + 1) * 32, (s + 1) & 1);          // + DMA share of stage s + 1 (buffer last read >= 2 steps ago by either half)
    PPB(1);
    load_frags(s, 1);
    if (dbg & 8) PPB(2); else PPB_VM(2);                    // own DMA pieces of stage s + 1 landed  (dbg 8: timing experiment, wrong results)
    multiply(no_dma, 0, 0);
    PPB(3);
  }
#ifdef CTCN_GEMM_STATS
  if (lane == 0 && (wave & 3) == 0 && (blockIdx.x == 0 || blockIdx.x == 1000) && g_gemm_stats) {
    long long *o = g_gemm_stats + ((blockIdx.x ? 2 : 0) + (wave >> 2)) * 16;
    for (int i = 0; i < 4; ++i) { o[i] = gs_work[i]; o[4 + i] = gs_wait[i]; }
    o[8] = clock64() - gs_t0; o[9] = nst - 1;
  }
#endif
  load_frags(nst - 1, 0);
  pp_barrier();
  multiply(no_dma, 0, 0);
  pp_barrier();
  load_frags(nst - 1, 1);
  pp_barrier();
  multiply(no_dma, 0, 0);
  pp_barrier();
  if (wm == 0) pp_barrier();                    // (half B's last multiply)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N && (!(dbg & 1) || v_is_magic(acc[i][j][e]))) {
          float v = acc[i][j][e];
          float *p = C + (size_t)row * ldc + col;
          if (beta != 0.0f) v += beta * *p;
          if (dbg & 4) __builtin_nontemporal_store(v, p); else *p = v;
        }
      }
    }
#ifdef CTCN_GEMM_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && (wave & 3) == 0 && (blockIdx.x == 0 || blockIdx.x == 1000) && g_gemm_stats)
    g_gemm_stats[((blockIdx.x ? 2 : 0) + (wave >> 2)) * 16 + 10] = clock64() - gs_entry;
#endif
}

template <int WNT, int NP = 3>
__global__ __launch_bounds__(512) void gemm_planes_nt256pp_kernel(int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                                  const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                                  const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc, float beta,
                                                                  int tiles_m, int tiles_n, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  planes256pp_tile<WNT, NP>(qsm, bid, M, N, Kp, Ah, Al, Bh, Bl, C, ldc, beta, tiles_n, dbg);
}

// XCD-filtered form for a side stream next to a persistent recurrence (see gemm_planes_nt_queue_kernel): workgroups off `xcd_allow`
// leave, the others take tiles from an atomic queue.  The same tile code: bit-identical results.
template <int WNT, int NP = 3>
__global__ __launch_bounds__(512) void gemm_planes_nt256pp_queue_kernel(int M, int N, int Kp, const unsigned short *__restrict__ Ah,
                                                                        const unsigned short *__restrict__ Al, const unsigned short *__restrict__ Bh,
                                                                        const unsigned short *__restrict__ Bl, float *__restrict__ C, int ldc, float beta,
                                                                        int tiles_m, int tiles_n, unsigned xcd_allow, unsigned *__restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  __shared__ int s_item;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (!((xcd_allow >> (x & 15)) & 1u)) return;
  const int nt = tiles_m * tiles_n;
  for (;;) {
    if (threadIdx.x == 0) s_item = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int item = s_item;
    if (item >= nt) return;
    planes256pp_tile<WNT, NP>(qsm, item, M, N, Kp, Ah, Al, Bh, Bl, C, ldc, beta, tiles_n, 0);
    __syncthreads();
  }
}

// The same tile with the A operand taken straight from the float32 matrix (row-major, k contiguous) and split into hi / lo bf16
// while it is staged: no plane pass over A (for da = 25 600 x 2 560 that pass moves 524 MB, 91 us -- a third of the product's
// own time).  Per stage a thread loads its 16 floats (64 B of one row) a stage ahead into registers, and after the stage's
// MFMAs converts them (the same split_bf16 as the plane pass: identical planes, bit-identical results) and writes four 16-B
// chunks into the other LDS buffer; the conversion is VALU work next to the other wave's MFMAs.  B (the weights) still comes
// pre-split through the DMA path.
template <int WNT>
__global__ __launch_bounds__(512) void gemm_planes_nt256_af32_kernel(int M, int N, int K, int Kp, const float *__restrict__ A, int lda,
                                                                     const unsigned short *__restrict__ Bh, const unsigned short *__restrict__ Bl,
                                                                     float *__restrict__ C, int ldc, float beta, int tiles_m, int tiles_n) {
  constexpr int TBM = 256, TBN = 128 * WNT;
  constexpr int A_BYTES = TBM * 64, B_BYTES = TBN * 64;
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  constexpr int IB = TBN * 4 / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x16 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  unsigned offB[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offB[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)Kp + chunk * 8;
  }
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  auto issue_b = [&](int k0, int buf) {
    unsigned char *sb = qsm + buf * STAGE + 2 * A_BYTES;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int dst = (i * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[i] + k0), (lptr_t)(sb + dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Bl + offB[i] + k0), (lptr_t)(sb + B_BYTES + dst), 16, 0, 0);
    }
  };
  // A staging of this thread: row tid >> 1 of the tile, 16 consecutive k (two 16-B bf16 chunks per plane)
  const int arow = tid >> 1, akh = (tid & 1) * 16;
  const float *aptr = A + (size_t)min(m0 + arow, M - 1) * lda + akh;
  f32x4 av[4];
  // (bare loads from clamped addresses: a select on the loaded value would make the wave wait for the load where it is issued;
  // K % 4 == 0, so a 16-B piece is inside the matrix or outside it as a whole, and the ones outside are zeroed when they are split)
  auto load_a = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4 *>(aptr + min(k0 + 4 * q, max(K - akh - 4, 0)));
  };
  auto store_a = [&](int buf, int k0) {
    unsigned char *sb = qsm + buf * STAGE;
    if (k0 + 32 > K) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k0 + akh + 4 * q >= K) av[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned hw[4], lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h0, l0, h1, l1;
        split_bf16(av[2 * h + (e >> 1)][(e & 1) * 2], h0, l0);
        split_bf16(av[2 * h + (e >> 1)][(e & 1) * 2 + 1], h1, l1);
        hw[e] = (unsigned)h0 | ((unsigned)h1 << 16);
        lw[e] = (unsigned)l0 | ((unsigned)l1 << 16);
      }
      const int o = qswz(arow, (tid & 1) * 2 + h);
      *reinterpret_cast<u32x4 *>(sb + o) = (u32x4){hw[0], hw[1], hw[2], hw[3]};
      *reinterpret_cast<u32x4 *>(sb + A_BYTES + o) = (u32x4){lw[0], lw[1], lw[2], lw[3]};
    }
  };
  const int nst = Kp / 32;
  const int ml = lane & 31, g = lane >> 5;
  load_a(0);
  issue_b(0, 0);
  store_a(0, 0);
  for (int s = 0; s < nst; ++s) {
    __syncthreads();                       // stage s is in LDS (A written by the waves, B landed), everybody is done with stage s - 1
    if (s + 1 < nst) { load_a((s + 1) * 32); issue_b((s + 1) * 32, (s + 1) & 1); }
    const unsigned char *sb = qsm + (s & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cq = ks * 2 + g;
      bf16x8_t ah[4], al[4], bh[WNT], bl[WNT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = qswz(wm * 128 + i * 32 + ml, cq);
        ah[i] = *reinterpret_cast<const bf16x8_t *>(sb + o);
        al[i] = *reinterpret_cast<const bf16x8_t *>(sb + A_BYTES + o);
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int o = qswz(wn * 32 * WNT + j * 32 + ml, cq);
        bh[j] = *reinterpret_cast<const bf16x8_t *>(sb + 2 * A_BYTES + o);
        bl[j] = *reinterpret_cast<const bf16x8_t *>(sb + 2 * A_BYTES + B_BYTES + o);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < nst) store_a((s + 1) & 1, (s + 1) * 32);  // (the other buffer: last read in stage s - 1, released by the barrier above)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][e];
          float *p = C + (size_t)row * ldc + col;
          if (beta != 0.0f) v += beta * *p;
          *p = v;
        }
      }
    }
}

// The float32-A tile in the ping-pong schedule of gemm_planes_nt256pp_kernel (one instruction stream, half B one barrier behind):
//     read (s, k-half 0) | multiply + the wave's DMA pieces of B(s+1) | read (s, 1), convert + store A(s+1), own DMA landed |
//     multiply + global loads of A(s+2)
// A thread stages its own 16 floats of a row of A: loaded during the second multiply of stage s-1, split into hi / lo (the same
// split_bf16 as everywhere: identical planes) and written into the other LDS buffer during the second read phase of stage s, three
// phases later.
// NP = 1 (option "gemm_bf16_single"): ah*bh only -- A rounded to bf16 as it is staged, the lo plane of B neither copied nor read.
template <int WNT, int NP = 3>
__global__ __launch_bounds__(512) void gemm_planes_nt256pp_af32_kernel(int M, int N, int K, int Kp, const float *__restrict__ A, int lda,
                                                                       const unsigned short *__restrict__ Bh, const unsigned short *__restrict__ Bl,
                                                                       float *__restrict__ C, int ldc, float beta, int tiles_m, int tiles_n) {
  constexpr int TBM = 256, TBN = 128 * WNT;
  constexpr int A_BYTES = TBM * 64, B_BYTES = TBN * 64;
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  constexpr int IB = TBN * 4 / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x16 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  unsigned offB[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int p = (i * 8 + wave) * 64 + lane, row = p >> 2, chunk = (p & 3) ^ ((row >> 2) & 3);
    offB[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)Kp + chunk * 8;
  }
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  constexpr int NBP = NP == 3 ? 2 * IB : IB;                   // DMA pieces of B per wave and stage
  auto issue_b_piece = [&](int n, int k0, int buf) {            // n = 0 .. NBP-1
    unsigned char *sb = qsm + buf * STAGE + 2 * A_BYTES;
    if constexpr (NP == 3) {
      const int i = n >> 1, dst = (i * 8 + wave) * 1024;
      if (n & 1) __builtin_amdgcn_global_load_lds((gptr_t)(Bl + offB[i] + k0), (lptr_t)(sb + B_BYTES + dst), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[i] + k0), (lptr_t)(sb + dst), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gptr_t)(Bh + offB[n] + k0), (lptr_t)(sb + (n * 8 + wave) * 1024), 16, 0, 0);
    }
  };
  // A staging of this thread: the 16-B bf16 chunk `ach` (8 consecutive k = two float4) of rows arow0 and 128 + arow0 of the tile -- four
  // lanes per row, so that the eight lanes of a ds_write_b128 group cover two whole rows = 128 contiguous bytes (two lanes per row, 32 B
  // apart, hit every bank twice: 22 % of the kernel's LDS cycles were write conflicts).  The loads are bare, from clamped addresses: a
  // select on the loaded value made the wave wait for every load where it is issued, i.e. inside the multiply phase.  K % 4 == 0: a
  // piece is inside the matrix or outside it as a whole, and the ones outside are zeroed when they are split.
  const int arow0 = tid >> 2, ach = tid & 3;
  // aw[2 * r + h]: row r (arow0 / 128 + arow0) of the tile, floats 4 h .. 4 h + 3 of the thread's eight.  The loads are issued by hand (uniform
  // base = first row of the tile + k0 in SGPRs, this lane's row / column as a 32-bit offset: no address arithmetic per stage) and waited for
  // by hand (WAIT_A): with loads of two stages in flight hipcc puts s_waitcnt vmcnt(0) in front of the first use of the older ones, i.e.
  // waits for the loads it has just issued
  u32x4 aw[4];
  const float *atile = A + (size_t)m0 * lda;
  const unsigned arel[2] = {(unsigned)min(arow0, M - 1 - m0) * (unsigned)lda, (unsigned)min(128 + arow0, M - 1 - m0) * (unsigned)lda};
  auto load_a_piece = [&](int q, int k0) {                     // q = 2 * row + half
    // ONE asm statement per piece (an if / else around two of them lets the compiler merge the two "results" with register copies behind the
    // branch -- of registers whose load is still in flight).  The stage that holds the end of K is clamped per lane (the split zeroes what
    // lies beyond K); every other stage adds k0 on the scalar side
    const int c = 8 * ach + 4 * (q & 1);
    const bool whole = k0 + 32 <= K;
    const float *sbase = atile + (whole ? k0 : 0);
    const unsigned voff = (arel[q >> 1] + (unsigned)(whole ? c : min(k0 + c, K - 4))) * 4u;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(aw[q]) : "v"(voff), "s"(sbase) : "memory");
  };
#ifdef CTCN_GEMM_NOVM      // (timing experiment of tools/mb_gemm_pp.hip: wrong results)
#define WAIT_A(N, r) asm volatile("" : "+v"(aw[2 * (r)]), "+v"(aw[2 * (r) + 1])::"memory")
#else
#define WAIT_A(N, r) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(aw[2 * (r)]), "+v"(aw[2 * (r) + 1])::"memory")
#endif
  // two neighbours at a time: hi = bf16(x) of both in one v_cvt_pk_bf16_f32, lo = bf16(x - hi) likewise (the split_bf16 values)
  auto split_pair = [](float a, float b, unsigned &hw, unsigned &lw) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
    const f32x2_t r = {a - __uint_as_float(hw << 16), b - __uint_as_float(hw & 0xffff0000u)};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  };
  // half h of row r (four consecutive floats) of the stage that starts at k0, in place: {x0, x1, x2, x3} -> {hi01, hi23, lo01, lo23}
  auto convert_half = [&](int r, int h, int k0) {
    u32x4 w = aw[2 * r + h];
    if (k0 + 32 > K && k0 + 8 * ach + 4 * h >= K) w = (u32x4){0u, 0u, 0u, 0u};          // (K % 4 == 0: a 16-B piece is inside or outside as a whole)
    unsigned h0, l0, h1, l1;
    if constexpr (NP == 3) {
      split_pair(__uint_as_float(w[0]), __uint_as_float(w[1]), h0, l0);
      split_pair(__uint_as_float(w[2]), __uint_as_float(w[3]), h1, l1);
    } else {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      h0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){__uint_as_float(w[0]), __uint_as_float(w[1])}, bf16x2_t));
      h1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){__uint_as_float(w[2]), __uint_as_float(w[3])}, bf16x2_t));
      l0 = l1 = 0u;
    }
    aw[2 * r + h] = (u32x4){h0, h1, l0, l1};
  };
  // the converted row r -> its 16-B chunk of either plane, as two 8-B halves each (straight from the register pairs convert_half left; four
  // lanes per row: a row's 64 B are contiguous)
  auto store_a_row = [&](int r, int buf) {
    unsigned char *sb = qsm + buf * STAGE + qswz(arow0 + 128 * r, ach);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<u32x2 *>(sb + 8 * h) = (u32x2){aw[2 * r + h][0], aw[2 * r + h][1]};
      if constexpr (NP == 3) *reinterpret_cast<u32x2 *>(sb + A_BYTES + 8 * h) = (u32x2){aw[2 * r + h][2], aw[2 * r + h][3]};
    }
  };
  const int nst = Kp / 32;
  const int ml = lane & 31, g = lane >> 5;
  bf16x8_t ah[4], al[4], bh[WNT], bl[WNT];
  int oa[2][4], ob[2][WNT];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) oa[ks][i] = qswz(wm * 128 + i * 32 + ml, ks * 2 + g);
#pragma unroll
    for (int j = 0; j < WNT; ++j) ob[ks][j] = 2 * A_BYTES + qswz(wn * 32 * WNT + j * 32 + ml, ks * 2 + g);
  }
  auto load_frags = [&](int stage, int ks) {
    const unsigned char *sb = qsm + (stage & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8_t *>(sb + ob[ks][j]);
      if constexpr (NP == 3) bl[j] = *reinterpret_cast<const bf16x8_t *>(sb + B_BYTES + ob[ks][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8_t *>(sb + oa[ks][i]);
      if constexpr (NP == 3) al[i] = *reinterpret_cast<const bf16x8_t *>(sb + A_BYTES + oa[ks][i]);
    }
  };
  constexpr int NMM = 4 * WNT;
  // One multiply phase: 3 * NMM MFMAs with the wave's other work of the stage between them, one item behind an MFMA:
  //   WHICH 1 (first multiply of stage s):  DMA pieces of B(s+1) | split of A(s+1) row 1 (loaded three phases ago)
  //   WHICH 2 (second multiply):            split of A(s+2) row 0 (loaded three phases ago)
  // HAS1: stage s + 1 exists, HAS2: stage s + 2 exists.  The float32 -> hi / lo split (12 VALU instructions per four floats) sits in the
  // shadow of the wave's OWN MFMAs, ~5 cycles per instruction: in the read phases, next to the other half's MFMAs, the same instructions cost
  // 15-20 cycles each (as one block of both rows that phase was 1 100-1 400 cycles long against the 430-500 of the MFMAs it is meant to hide
  // behind; one row per read phase: 800 -- tools/mb_gemm_pp.hip).  The A loads leave from the READ phases (row 0 of A(s+2) in the first one of
  // stage s, row 1 in the second, each right behind the ds_write that freed its registers).  In flight, oldest first (loads and DMA pieces
  // return in order), when the split starts: first multiply: A(s+1) row 1 (2 loads), A(s+2) row 0 (2, HAS2), this phase's 2 IB DMA pieces;
  // second multiply: A(s+2) row 1 (2) -- row 0 and the DMA pieces were waited for in front of the barrier before it
  auto multiply = [&](auto which, auto has1, auto has2, int s) {
    constexpr int WHICH = decltype(which)::value;
    constexpr bool HAS1 = decltype(has1)::value, HAS2 = decltype(has2)::value;
    constexpr int CV0 = NP * NMM - 3;                          // the two splits go behind the third- and second-to-last MFMA (the loads have had the longest time)
    static_assert(NBP <= CV0, "DMA slots and split slots of a multiply phase overlap");
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          if (NP == 3 && t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
          else if (NP == 3 && t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          const int n = t * NMM + i * WNT + j;
          if (n < NBP || n == CV0 || n == CV0 + 1) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WHICH == 1) {
              if (HAS1 && n < NBP) issue_b_piece(n, (s + 1) * 32, (s + 1) & 1);
              if (HAS1 && n == CV0) {          // A(s+1) row 1 has landed: what may stay in flight is A(s+2) row 0 (2 loads, HAS2) + this phase's NBP pieces
                constexpr int OUT = (HAS2 ? 2 : 0) + NBP;
                if constexpr (OUT == 1) WAIT_A(1, 1); else if constexpr (OUT == 2) WAIT_A(2, 1); else if constexpr (OUT == 3) WAIT_A(3, 1);
                else if constexpr (OUT == 4) WAIT_A(4, 1); else { static_assert(OUT <= 4 || OUT == 6, "vmcnt literal"); WAIT_A(6, 1); }
              }
              if (HAS1 && n >= CV0) convert_half(1, n - CV0, (s + 1) * 32);
            } else {
              if (HAS2 && n == CV0) WAIT_A(2, 0);
              if (HAS2 && n >= CV0) convert_half(0, n - CV0, (s + 2) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };
  constexpr std::integral_constant<int, 1> first{};
  constexpr std::integral_constant<int, 2> second{};
  constexpr std::integral_constant<bool, true> yes{};
  constexpr std::integral_constant<bool, false> no{};
  // prologue: stage 0 into buffer 0 (A through the registers, B by DMA), all of A(1) into the registers
#pragma unroll
  for (int q = 0; q < 4; ++q) aw[q] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) load_a_piece(q, 0);
#pragma unroll
  for (int n = 0; n < NBP; ++n) issue_b_piece(n, 0, 0);
  WAIT_A(0, 0); WAIT_A(0, 1);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    convert_half(r, 0, 0); convert_half(r, 1, 0);
    store_a_row(r, 0);
  }
  if (nst > 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the ds_writes have read aw)
#pragma unroll
    for (int q = 0; q < 4; ++q) load_a_piece(q, 32);
    WAIT_A(2, 0);
    convert_half(0, 0, 32); convert_half(0, 1, 32);           // row 0 of A(1): the first read phase stores it; row 1 is split in the first multiply
  }
  pp_barrier_vm();
  if (wm == 1) pp_barrier();
#ifdef CTCN_GEMM_STATS
  long long gs_work[4] = {0, 0, 0, 0}, gs_wait[4] = {0, 0, 0, 0};
  const long long gs_t0 = clock64();
  long long gs_prev = gs_t0;
#endif
  auto stage = [&](int s, auto has1, auto has2) {
    constexpr bool HAS1 = decltype(has1)::value, HAS2 = decltype(has2)::value;
    load_frags(s, 0);
    if constexpr (HAS1) store_a_row(0, (s + 1) & 1);          // A(s+1) row 0 (split in the previous multiply phase) -> the other buffer (last read two phases ago)
    if constexpr (HAS2) {                                     // its registers take A(s+2) row 0 (the ds_write above has read them at issue; the load returns hundreds of cycles later)
      __builtin_amdgcn_sched_barrier(0);
      load_a_piece(0, (s + 2) * 32); load_a_piece(1, (s + 2) * 32);
    }
    PPB(0);
    multiply(first, has1, has2, s);
    PPB(1);
    load_frags(s, 1);
    if constexpr (HAS1) store_a_row(1, (s + 1) & 1);          // A(s+1) row 1 (split in the multiply phase just before)
    if constexpr (HAS2) {
      __builtin_amdgcn_sched_barrier(0);
      load_a_piece(2, (s + 2) * 32); load_a_piece(3, (s + 2) * 32);
    }
    // this wave's DMA pieces of B(s+1) have landed (and A(s+2) row 0, older; the two loads of row 1, just issued, stay in flight)
#ifdef CTCN_GEMM_STATS
    PPB_VM(2);
#else
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HAS2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    multiply(second, has1, has2, s);
    PPB(3);
  };
  int s = 0;
  for (; s + 2 < nst; ++s) stage(s, yes, yes);
#ifdef CTCN_GEMM_STATS
  if (lane == 0 && (wave & 3) == 0 && (blockIdx.x == 0 || blockIdx.x == 1000) && g_gemm_stats) {
    long long *o = g_gemm_stats + ((blockIdx.x ? 2 : 0) + (wave >> 2)) * 16;
    for (int i = 0; i < 4; ++i) { o[i] = gs_work[i]; o[4 + i] = gs_wait[i]; }
    o[8] = clock64() - gs_t0; o[9] = nst - 2; o[10] = 0;
  }
#endif
  if (s + 1 < nst) { stage(s, yes, no); ++s; }
  stage(s, no, no);
  if (wm == 0) pp_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][e];
          float *p = C + (size_t)row * ldc + col;
          if (beta != 0.0f) v += beta * *p;
          *p = v;
        }
      }
    }
}


__global__ void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ C, int M, int N, int ldc, int splits, float beta);

// ------------------------------------------------------------------------------------------------
// TN tile: C = A^T B for two CONTRACTION-MAJOR float32 operands (A: K x M, B: K x N, k the slow index) -- the weight gradients
// dW = da^T x, whose operands are the activations exactly as the recurrences leave them (K = T*B rows).  The plane path above needs
// them transposed AND split first (split_transpose_kernel: a read + write pass over both operands per GEMM); this tile takes the
// float32 rows as they are:
//   * a thread loads 16-B pieces of 4 neighbouring m of one k (8 lanes = one 128-B line of a row), splits them into hi / lo with
//     the same split_bf16 as everywhere and writes them with ds_write_b64 into a K-MAJOR LDS image: [32-m subtile][32 k][32 m]
//     bf16, rows of 64 B -- a wave's write covers 512 contiguous bytes, conflict-free;
//   * the MFMA operand (lane = m, 8 consecutive k) is the transpose of that image: two ds_read_b64_tr_b16 per fragment (each
//     16-lane group reads a 4 k x 16 m block and receives it transposed), so no shuffle and no second LDS pass;
//   * the ping-pong schedule of gemm_planes_nt256pp_af32_kernel with BOTH operands through the registers: global loads of stage
//     s + 2 between the MFMAs of the second multiply phase of stage s, their hi / lo split between the MFMAs of the first multiply
//     phase of stage s + 1 (VALU work in the shadow of the wave's own MFMAs), LDS stores in the read phase after it;
//   * split-K with an atomic (tile, split) queue on the XCDs of `xcd_mask`: partial tiles land in the workspace and
//     splitk_reduce_kernel adds them up -- the order of the splits is fixed, so the result does not depend on which workgroup
//     computed what.
// Products and k order inside a 16-k step are those of the other bf16x3 tiles (al*bh, ah*bl, ah*bh).
// ------------------------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

// NP = 1 (option "gemm_bf16_single"): ah*bh only -- both operands rounded to bf16 as they are staged, no lo planes in LDS.
template <int WNT, int NP = 3>
__global__ __launch_bounds__(512) void gemm_tn_f32_pp_kernel(int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                             int ldb, float *__restrict__ C, int ldc, float beta, int kchunk, int splits,
                                                             float *__restrict__ part, int tiles_m, int tiles_n, unsigned xcd_mask,
                                                             unsigned *__restrict__ queue) {
  constexpr int TBM = 256, TBN = 128 * WNT;
  constexpr int A_BYTES = TBM * 64, B_BYTES = TBN * 64;
  constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  constexpr int NB = 2 * WNT;            // 16-B pieces of B per thread and stage (A: 4)
  constexpr int NMM = 4 * WNT;
  extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
  __shared__ int s_item;
  // workgroups off the XCDs of `xcd_mask` leave; the others pull (split, tile) items from ONE queue, tiles of a split next to each other
  // (an owner XCD per split -- its workgroups walking one k window together through their L2 -- measured 8-20 % slower: the L2s
  // already catch 73 % of the requests and the fixed ownership costs balance)
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (!((xcd_mask >> (xcc & 15u)) & 1u)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = tiles_m * tiles_n;
  const int krow = lane >> 3, c4 = 4 * (lane & 7);
  const int bsub = WNT == 2 ? wave : (wave & 3), bk0 = WNT == 2 ? 0 : 16 * (wave >> 2);
  // LDS byte offsets: this thread's pieces, and its transposing fragment reads (see tools/tr_probe: lane l of ds_read_b64_tr_b16 gets
  // column (l & 15) + 16 * ((l >> 4) & 1), rows 8 * (l >> 5) + 0..3 when lane i of a 16-lane group addresses row i / 4, columns 4 * (i % 4))
  const int st_a = wave * 2048 + krow * 64 + (lane & 7) * 8;
  const int st_b = 2 * A_BYTES + bsub * 2048 + (bk0 + krow) * 64 + (lane & 7) * 8;
  const int troff = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  typedef __attribute__((address_space(3))) s16x4_t *lds4_t;

  for (;;) {
    if (tid == 0) s_item = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_item);   // (uniform: everything derived from it -- tile origin, k window, buffer resources -- stays scalar)
    if (item >= nt * splits) return;
    const int z = item / nt, tile = item - z * nt;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);
    const int nst = (kend - kbeg + 31) / 32;

    f32x16 acc[4][WNT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int acol = m0 + 32 * wave + c4, bcol = n0 + 32 * bsub + c4;
    const bool a_ok = acol < M, b_ok = bcol < N;                 // (M, N multiples of 4: a piece is inside or outside as a whole)
    // pw[q]: piece q of the stage in flight (0..3: A, 4..: B) -- first the four floats as loaded, then (convert_piece) the packed
    // words {hi01, hi23, lo01, lo23} in the same registers.  A piece beyond M / N / the k window is loaded from 16 zero bytes next to
    // the queue word instead: the select is on the ADDRESS, so nothing waits for the data inside the multiply phase the loads are
    // issued in, and the split needs no special case
    u32x4 pw[4 + NB];
    // Round 3: the eight rows of a piece through a buffer resource made on the SCALAR side per piece -- base = the piece's first row, extent =
    // its rows inside the k window -- and ONE loop-invariant lane offset (row krow, this lane's columns; 0xfffffff0 for a lane whose columns lie
    // beyond M / N): rows past the window and lanes past the edge read zeros from the range check, and the multiply phase that issues the
    // loads loses the ~18 VALU instructions per load of 64-bit address arithmetic and selects it carried (1 320 -> ~1 000 cycles,
    // tools/mb_gemm_pp.hip)
    const unsigned voff_a = a_ok ? (unsigned)((krow * lda + acol) * 4) : 0xfffffff0u;
    const unsigned voff_b = b_ok ? (unsigned)((krow * ldb + bcol) * 4) : 0xfffffff0u;
    auto load_piece = [&](int q, int k0) {                       // k0 = first k of the stage
      const int kr = q < 4 ? k0 + 8 * q : k0 + bk0 + 8 * (q - 4);                      // first of the piece's eight rows (uniform)
      const int ld = q < 4 ? lda : ldb;
      const float *base = (q < 4 ? A : B) + (size_t)min(kr, K - 1) * ld;
      int rows = min(8, kend - kr);
      asm volatile("" : "+s"(rows));                             // (kept on the scalar side: as max(0, min(8, .)) it becomes a v_med3 and the resource a waterfall loop)
      const int nrec = max(0, rows) * ld * 4;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, nrec, 0x00020000);
      pw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, q < 4 ? voff_a : voff_b, 0, 0);
    };
    // two neighbours at a time: hi = bf16(x) of both in one v_cvt_pk_bf16_f32, lo = bf16(x - hi) likewise (the split_bf16 values)
    auto split_pair = [](float a, float b, unsigned &hw, unsigned &lw) {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      hw = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
      const f32x2_t r = {a - __uint_as_float(hw << 16), b - __uint_as_float(hw & 0xffff0000u)};
      lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    };
    auto convert_piece = [&](int q) {
      unsigned h01, l01, h23, l23;
      if constexpr (NP == 3) {
        split_pair(__uint_as_float(pw[q][0]), __uint_as_float(pw[q][1]), h01, l01);
        split_pair(__uint_as_float(pw[q][2]), __uint_as_float(pw[q][3]), h23, l23);
      } else {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){__uint_as_float(pw[q][0]), __uint_as_float(pw[q][1])}, bf16x2_t));
        h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){__uint_as_float(pw[q][2]), __uint_as_float(pw[q][3])}, bf16x2_t));
        l01 = l23 = 0u;
      }
      pw[q] = (u32x4){h01, h23, l01, l23};
    };
    auto store_ab = [&](int buf) {                               // converted pieces -> the K-major image
      unsigned char *sb = qsm + buf * STAGE;
#pragma unroll
      for (int q = 0; q < 4 + NB; ++q) {
        unsigned char *p = sb + (q < 4 ? st_a + q * 512 : st_b + (q - 4) * 512);
        *reinterpret_cast<uint2 *>(p) = make_uint2(pw[q][0], pw[q][1]);
        if constexpr (NP == 3) *reinterpret_cast<uint2 *>(p + (q < 4 ? A_BYTES : B_BYTES)) = make_uint2(pw[q][2], pw[q][3]);
      }
    };
    bf16x8_t ah[4], al[4], bh[WNT], bl[WNT];
    auto tr_frag = [&](const unsigned char *p) {
      const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p));
      const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p + 256));
      return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto load_frags = [&](int stage, int ks) {
      const unsigned char *sb = qsm + (stage & 1) * STAGE + ks * 1024 + troff;
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const unsigned char *p = sb + 2 * A_BYTES + (wn * WNT + j) * 2048;
        bh[j] = tr_frag(p);
        if constexpr (NP == 3) bl[j] = tr_frag(p + B_BYTES);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char *p = sb + (wm * 4 + i) * 2048;
        ah[i] = tr_frag(p);
        if constexpr (NP == 3) al[i] = tr_frag(p + A_BYTES);
      }
    };
    // MODE 0: bare MFMAs; 1: + this thread's 4 + NB global loads of the stage that begins at k0, one after every other MFMA;
    // 2: + the split of the pieces loaded last into hi / lo words, one piece after every other MFMA
    auto multiply = [&](auto mode, int k0) {
      constexpr int MODE = decltype(mode)::value;
      constexpr int NITEM = MODE == 1 ? 4 : 4 + NB;                  // (MODE 1: the A pieces only -- B's are loaded in the read phase after)
#pragma unroll
      for (int t = 0; t < NP; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < WNT; ++j) {
            if (NP == 3 && t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
            else if (NP == 3 && t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            const int n = t * NMM + i * WNT + j;
            if constexpr (NP == 3) {
              if (MODE != 0 && (n & 1) && (n >> 1) < NITEM) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 1) load_piece(n >> 1, k0);
                if constexpr (MODE == 2) convert_piece(n >> 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            } else if (MODE != 0 && 2 * n < NITEM) {                   // a third of the MFMAs: two items behind every one of them
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 2 * n; u < 2 * n + 2 && u < NITEM; ++u) {
                if constexpr (MODE == 1) load_piece(u, k0);
                if constexpr (MODE == 2) convert_piece(u);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    };
    constexpr std::integral_constant<int, 0> bare{};
    constexpr std::integral_constant<int, 1> with_loads{};
    constexpr std::integral_constant<int, 2> with_split{};
    // prologue: stage 0 into buffer 0, stage 1 into the registers
#pragma unroll
    for (int q = 0; q < 4 + NB; ++q) load_piece(q, kbeg);
#pragma unroll
    for (int q = 0; q < 4 + NB; ++q) convert_piece(q);
    store_ab(0);
    if (nst > 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) load_piece(q, kbeg + 32);
    }
    pp_barrier();
    if (wm == 1) pp_barrier();
    // a stage = four phases: read (s, k-half 0) + loads of B(s+1) | multiply + split of the pieces of stage s + 1 (A loaded two phases ago) |
    //                        read (s, 1) + LDS stores of stage s + 1 (buffer last read two phases ago) | multiply + loads of A(s+2)
    // (round 3: all 4 + NB loads in the second multiply phase made it the longest of the four -- 1 000 cycles against a first read phase of
    // 400 whose half then waits 600 for the other half's multiply; B's NB loads moved into that read phase.  The compiler's counted vmcnt
    // waits in the split keep the order: A's pieces, the older loads, are split first)
#ifdef CTCN_GEMM_STATS
    long long gs_work[4] = {0, 0, 0, 0}, gs_wait[4] = {0, 0, 0, 0};
    const long long gs_t0 = clock64();
    long long gs_prev = gs_t0;
#endif
    auto stage = [&](int s, auto do_store, auto c1) {
      load_frags(s, 0);
      if constexpr (decltype(do_store)::value == 1) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4; q < 4 + NB; ++q) load_piece(q, kbeg + (s + 1) * 32);       // (their registers were stored to LDS two phases ago)
      }
      PPB(0);
      if constexpr (decltype(do_store)::value == 1) multiply(with_split, 0);
      else multiply(bare, 0);
      PPB(1);
      load_frags(s, 1);
      if constexpr (decltype(do_store)::value == 1) store_ab((s + 1) & 1);
      PPB(2);
      multiply(c1, kbeg + (s + 2) * 32);
      PPB(3);
    };
    int s = 0;
    for (; s + 2 < nst; ++s) stage(s, with_loads, with_loads);
#ifdef CTCN_GEMM_STATS
    if (lane == 0 && (wave & 3) == 0 && item == 0 && g_gemm_stats) {
      long long *o = g_gemm_stats + (wave >> 2) * 16;
      for (int i = 0; i < 4; ++i) { o[i] = gs_work[i]; o[4 + i] = gs_wait[i]; }
      o[8] = clock64() - gs_t0; o[9] = nst - 2; o[10] = 0;
    }
#endif
    if (s + 1 < nst) { stage(s, with_loads, bare); ++s; }
    stage(s, bare, bare);
    if (wm == 0) pp_barrier();

    float *o = splits > 1 ? part + (size_t)z * M * N : C;
    const int ldo = splits > 1 ? N : ldc;
    const float bt = splits > 1 ? 0.0f : beta;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int col = n0 + wn * 32 * WNT + j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (row < M && col < N) {
            float v = acc[i][j][e];
            float *p = o + (size_t)row * ldo + col;
            if (bt != 0.0f) v += bt * *p;
            *p = v;
          }
        }
      }
    __syncthreads();
  }
}

// split count and launch of the TN tile; `part` must hold splits * M * N floats when splits > 1
static int tn_splits(int M, int N, int K, int wnt, size_t part_bytes, int cus) {
  const int tiles = ceil_div(M, 256) * ceil_div(N, 128 * wnt);
  // `cus`: the CUs the launch may run on -- 256, or (round 5, option "tn_splits_xcd") the CUs of the XCDs of xcd_allow: one round of items there
  // instead of two rounds of half-length items (each item pays a prologue, a 128-KB partial store and its share of the reduce pass)
  int s = std::min(std::max(cus / tiles, 1), std::max(K / 512, 1));      // <= 256 items: one round on a whole device
  s = std::min(s, 32);                                                    // (a handful of tiles: the reduce pass over the partials would take over)
  if (const int f = ctcn_get_option("tn_splits_force")) s = std::min(f, std::max(K / 64, 1));       // (development)
  s = std::min(s, (int)(part_bytes / ((size_t)M * N * sizeof(float))));
  return s < 2 ? 1 : s;
}
// option "gemm_bf16_single" (default 0): 1 = the 256-row tiles (plane, float32-A and TN forms: every product of T*B rows) multiply the bf16 roundings of
// their operands ONCE (ah*bh, f32 accumulate) instead of the three bf16x3 products -- the "bf16 tolerance" BASELINE.json's north_star states
// (loss / activations within 1e-3), against the default's f32-equivalent 1e-5.  The recurrent matmul and the small tiles keep bf16x3.
static bool bf16_single() { return ctcn_get_option("gemm_bf16_single") != 0; }

template <int WNT>
static int launch_tn(hipStream_t st, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, float beta, int splits,
                     float *part, unsigned xcd_allow, unsigned *queue) {
  const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 128 * WNT);
  int kchunk = ceil_div(ceil_div(K, splits), 32) * 32;
  splits = ceil_div(K, kchunk);
  const size_t lds = (size_t)2 * (2 * 256 * 64 + 2 * 128 * WNT * 64);
  auto kern = bf16_single() ? gemm_tn_f32_pp_kernel<WNT, 1> : gemm_tn_f32_pp_kernel<WNT, 3>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CTCN_HIP(hipMemsetAsync(queue, 0, 128, st));              // the queue word, and at +64 the 16 zero bytes pieces outside the operands are loaded from
  const int nx = std::min(ctcn_device_xcds(), 16);
  const unsigned xcd_mask = xcd_allow ? xcd_allow : (nx > 1 ? (1u << nx) - 1u : 0xffffu);
  hipLaunchKernelGGL(kern, dim3(ctcn_device_cus()), dim3(512), lds, st, M, N, K, A, lda, B, ldb, C, ldc, beta, kchunk, splits, part, tiles_m, tiles_n, xcd_mask,
                     queue);
  CTCN_LAUNCH_CHECK();
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min((size_t)2048, ceil_div_z(total, 256))), dim3(256), 0, st, (const float *)part, C, M, N, ldc,
                       splits, beta);
    CTCN_LAUNCH_CHECK();
  }
  return CTCN_OK;
}

template <int WNT>
static int launch_planes256_af32(hipStream_t st, int M, int N, int K, int Kp, const float *A, int lda, const unsigned short *bh, const unsigned short *bl,
                                 float *C, int ldc, float beta) {
  const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 128 * WNT);
  const size_t lds = (size_t)2 * (2 * 256 * 64 + 2 * 128 * WNT * 64);
  auto kern = ctcn_get_option("gemm_pingpong") ? (bf16_single() ? gemm_planes_nt256pp_af32_kernel<WNT, 1> : gemm_planes_nt256pp_af32_kernel<WNT, 3>)
                                               : gemm_planes_nt256_af32_kernel<WNT>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, M, N, K, Kp, A, lda, bh, bl, C, ldc, beta, tiles_m, tiles_n);
  return CTCN_OK;
}

template <int WNT>
static int launch_planes256(hipStream_t st, int M, int N, int Kp, const unsigned short *ah, const unsigned short *al, const unsigned short *bh,
                            const unsigned short *bl, float *C, int ldc, float beta) {
  const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 128 * WNT);
  const size_t lds = (size_t)2 * (2 * 256 * 64 + 2 * 128 * WNT * 64);
  const int dbg = ctcn_get_option("gemm_dbg");
  auto kern = ctcn_get_option("gemm_pingpong") ? (bf16_single() ? gemm_planes_nt256pp_kernel<WNT, 1> : gemm_planes_nt256pp_kernel<WNT, 3>)
                                               : gemm_planes_nt256_kernel<WNT>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, tiles_m, tiles_n, dbg);
  return CTCN_OK;
}

template <int WNT>
static int launch_planes256_queue(hipStream_t st, int M, int N, int Kp, const unsigned short *ah, const unsigned short *al, const unsigned short *bh,
                                  const unsigned short *bl, float *C, int ldc, float beta, unsigned xcd_allow, unsigned *queue) {
  const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 128 * WNT);
  const size_t lds = (size_t)2 * (2 * 256 * 64 + 2 * 128 * WNT * 64);
  auto kern = bf16_single() ? gemm_planes_nt256pp_queue_kernel<WNT, 1> : gemm_planes_nt256pp_queue_kernel<WNT, 3>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(ctcn_device_cus()), dim3(512), lds, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, tiles_m, tiles_n, xcd_allow, queue);
  return CTCN_OK;
}

// launch one of the three tile shapes (dynamic LDS = 2 planes x (BM + BN) rows x 128 B)
template <int TI, int TJ>
static int launch_planes(bool queued, int nt, int psplits, hipStream_t st, int M, int N, int Kp, const unsigned short *ah, const unsigned short *al,
                         const unsigned short *bh, const unsigned short *bl, float *C, int ldc, float beta, int pchunk, float *part, int tiles_m,
                         int tiles_n, unsigned xcd_allow, unsigned *queue) {
  const size_t lds = (size_t)2 * (64 * TI + 64 * TJ) * 128;
  if (queued) {
    auto kern = gemm_planes_nt_queue_kernel<TI, TJ>;
    if (lds > 64 * 1024) CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(2 * ctcn_device_cus()), dim3(256), lds, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, pchunk, part, tiles_m, tiles_n,
                       psplits, xcd_allow, queue);
  } else {
    auto kern = gemm_planes_nt_kernel<TI, TJ>;
    if (lds > 64 * 1024) CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nt, psplits), dim3(256), lds, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, pchunk, part, tiles_m, tiles_n);
  }
  return CTCN_OK;
}

__global__ void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ C, int M, int N, int ldc,
                                     int splits, float beta) {
  const size_t total = (size_t)M * N;
  // the partials are summed in split order (the result does not depend on the launch), eight loads in flight at a time.
  // Round 5: 16-B loads / stores and a 32-bit division per FOUR outputs where the shape allows (every product of the training step:
  // N, ldc multiples of 4, 16-B aligned C and partials) -- the same additions in the same order, bit-identical to the scalar form below.
  if ((N & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)C | (uintptr_t)ws) & 15) == 0 && total < ((size_t)1 << 33)) {
    const unsigned n4 = (unsigned)(N >> 2), total4 = (unsigned)(total >> 2);
    const float4 *w4 = reinterpret_cast<const float4 *>(ws);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
      float4 s = {0.0f, 0.0f, 0.0f, 0.0f};
      int z = 0;
      for (; z + 8 <= splits; z += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w4[(size_t)(z + u) * total4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; z < splits; ++z) { const float4 v = w4[(size_t)z * total4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      const unsigned m = i / n4, n = (i - m * n4) << 2;
      float4 *p = reinterpret_cast<float4 *>(C + (size_t)m * ldc + n);
      if (beta != 0.0f) { const float4 o = *p; s.x += beta * o.x; s.y += beta * o.y; s.z += beta * o.z; s.w += beta * o.w; }
      *p = s;
    }
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(z + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; ++z) s += ws[(size_t)z * total + i];
    const size_t m = i / N, n = i - m * N;
    float *p = C + m * ldc + n;
    if (beta != 0.0f) s += beta * *p;
    *p = s;
  }
}

__global__ void transpose01_kernel(const float *__restrict__ in, float *__restrict__ out, int A, int B, int C) {
  // out[b][a][c] = in[a][b][c]
  const size_t total = (size_t)A * B * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const size_t ab = i / C;
    const int a = ab % A;     // out index = (b*A + a)*C + c
    const int b = ab / A;
    out[i] = in[((size_t)a * B + b) * C + c];
  }
}

}  // namespace

// the TN tile (gemm_tn_f32_pp_kernel) takes C = A^T B with 16-B aligned rows of whole float4 pieces, large K, and a workspace for its
// queue word (+ the split-K partials)
static bool tn_eligible(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, int precision, const void *ws,
                        size_t ws_bytes) {
  return precision == 1 && transA && !transB && ws && ws_bytes >= 1024 && ctcn_get_option("gemm_tn") != 0 && M >= 128 && N >= 32 && K >= 1024 &&
         M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
         lda < (1 << 24) && ldb < (1 << 24);                     // (eight rows of a piece behind one 32-bit buffer offset)
}
// xcd_allow: 0 = whole device; otherwise (precision 1 plane path only) the XCDs the GEMM workgroups may run on
// b_shift: ctcn_gemm_shift_b's shift of B along k, applied while B is split (plane path only)
static int gemm_core(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
                     int ldc, float beta, int precision, void *ws, size_t ws_bytes, void *stream, unsigned xcd_allow, GemmPlanes *planes, int b_shift) {
  GemmPlanes scratch_planes;                  // a caller without a sequence: nothing to reuse, nothing remembered
  GemmPlanes &pl = planes ? *planes : scratch_planes;
  const bool want_same_a = pl.same_a, want_same_b = pl.same_b;
  pl.same_a = pl.same_b = false;              // one-shot
  CTCN_REQUIRE(precision == 0 || precision == 1, "ctcn_gemm: precision %d (0 = f32 MFMA, 1 = bf16x3 split MFMA)", precision);
  CTCN_REQUIRE(M > 0 && N > 0 && K >= 0, "ctcn_gemm: bad dims M=%d N=%d K=%d", M, N, K);
  CTCN_REQUIRE(A && B && C, "ctcn_gemm: null pointer");
  CTCN_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "ctcn_gemm: leading dim too small");
  hipStream_t st = (hipStream_t)stream;
  const int tiles_m = ceil_div(M, BM), tiles_n = ceil_div(N, BN);
  const int nt = tiles_m * tiles_n;
  int splits = 1;
  if (nt < 256 && K >= 512 && ws && ws_bytes >= (size_t)2 * M * N * sizeof(float)) {
    splits = ceil_div(512, nt);
    splits = min(splits, K / 256);
    splits = min(splits, (int)(ws_bytes / ((size_t)M * N * sizeof(float))));
    if (splits < 2) splits = 1;
  }
  int kchunk = K;
  if (splits > 1) {
    kchunk = ceil_div(ceil_div(K, splits), XBK) * XBK;
    splits = ceil_div(K, kchunk);
  }
  if (tn_eligible(transA, transB, M, N, K, A, lda, B, ldb, precision, ws, ws_bytes) && b_shift == 0) {
    // both operands contraction-major (weight gradients): the TN tile reads the float32 rows as they are -- no plane pass
    pl.a_valid = false; pl.b_valid = false;
    const int wnt = ceil_div(N, 256) * 256 == ceil_div(N, 128) * 128 ? 2 : 1;     // (N = 640 as 3 x 256 or 5 x 128: the same time)
    unsigned *queue = (unsigned *)((char *)ws + ((ws_bytes - 256) & ~(size_t)255));
    int tn_cus = 256;
    if (xcd_allow && ctcn_get_option("tn_splits_xcd") != 0) {
      const int nx = std::max(1, std::min(ctcn_device_xcds(), 16));
      tn_cus = std::max(ctcn_device_cus() / nx * __builtin_popcount(xcd_allow & ((1u << nx) - 1u)), 32);
    }
    const int tsplits = tn_splits(M, N, K, wnt, ws_bytes - 512, tn_cus);
    return wnt == 2 ? launch_tn<2>(st, M, N, K, A, lda, B, ldb, C, ldc, beta, tsplits, (float *)ws, xcd_allow, queue)
                    : launch_tn<1>(st, M, N, K, A, lda, B, ldb, C, ldc, beta, tsplits, (float *)ws, xcd_allow, queue);
  }
  if (precision == 1 && K >= 64 && ws) {
    // pre-split planes in the workspace: [Ah | Al | Bh | Bl | split-K partials]
    const int Kp = ceil_div(K, PBK) * PBK;
    const size_t a_el = (size_t)M * Kp, b_el = (size_t)N * Kp;
    const size_t plane_bytes = align_up(2 * (a_el + b_el) * sizeof(unsigned short), 256);
    if (ws_bytes >= plane_bytes + 1024) {
      unsigned short *ah = (unsigned short *)ws, *al = ah + a_el, *bh = al + a_el, *bl = bh + b_el;
      float *part = (float *)((char *)ws + plane_bytes);
      // the last 256 bytes of the workspace hold the work queue of the XCD-filtered variant
      unsigned *queue = (unsigned *)((char *)ws + ((ws_bytes - 256) & ~(size_t)255));
      const size_t part_bytes = ws_bytes - plane_bytes - 512;
      if (xcd_allow) CTCN_HIP(hipMemsetAsync(queue, 0, 16, st));
      int nq = 0;      // queue words 1, 2: the operand splits
      auto split = [&](const float *src, int ld, bool contraction_major, int rows, unsigned short *hi, unsigned short *lo, int shift) {
        if (!contraction_major) {   // src[row*ld + k]
          const size_t total = (size_t)rows * (Kp / 4);
          hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)std::min((size_t)8192, ceil_div_z(total, 256))), dim3(256), 0, st, src, ld, rows, K, Kp, hi, lo);
        } else {                    // src[k*ld + row] -> transpose
          if (xcd_allow)
            hipLaunchKernelGGL(split_transpose_queue_kernel, dim3(ctcn_opt_side_split_wgs() * ctcn_device_cus()), dim3(256), 0, st, src, ld, K, rows, Kp, hi, lo, ceil_div(rows, 64),
                               ceil_div(Kp, 64), xcd_allow, queue + (++nq), shift);
          else
            hipLaunchKernelGGL(split_transpose_kernel, dim3(ceil_div(rows, 64), ceil_div(Kp, 64)), dim3(256), 0, st, src, ld, K, rows, Kp, hi, lo, shift);
        }
      };
      const bool same_a = want_same_a && pl.a_valid && pl.A == A && pl.lda == lda && pl.M == M && pl.Ka == K && pl.transA == transA && pl.ws_a == ws &&
                          pl.st_a == (void *)st;
      const bool same_b = want_same_b && pl.b_valid && pl.B == B && pl.ldb == ldb && pl.N == N && pl.Kb == K && pl.transB == transB && pl.shift == b_shift &&
                          pl.ws_b == ws && pl.st_b == (void *)st && pl.bh == (const void *)bh;
      pl.B = B; pl.ldb = ldb; pl.N = N; pl.Kb = K; pl.transB = transB; pl.shift = b_shift; pl.ws_b = ws;
      pl.st_b = (void *)st; pl.bh = (const void *)bh; pl.b_valid = true;     // (every branch below leaves B's planes at bh / bl)
      // activation-sized products (M = T*B): the 256-row tiles, when they give the device at least ~0.75 workgroups per CU; a
      // row-major A (k contiguous) is then split while it is staged, without a plane pass of its own
      const int wnt256 = (N % 256 == 0 || (N > 512 && ceil_div(N, 256) * 256 - N <= N / 8)) ? 2 : 1;
      const bool use256 = !xcd_allow && ctcn_get_option("gemm_tile256") != 0 && M >= 1024 && N >= 96 &&
                          (long)ceil_div(M, 256) * ceil_div(N, 128 * wnt256) * 4 >= (long)ctcn_device_cus() * 3;
      // (every N-tile converts its A rows again.  Stand-alone the inline split now wins at every shape of the bench configurations --
      // 25 600 x 1 280 x 640: 134 vs 150 us, 25 600 x 640 x 2 560: 278 vs 354 us, 25 600 x 2 560 x 640 (10 N-tiles): 253 vs 262 us,
      // 76 800 x 3 072 x 1 024 (12 N-tiles): 1 333 vs 1 351 us -- but with a limit of 16 N-tiles instead of 5 no step got faster (cfg2
      // 13.74 / 13.75 ms in an A/B inside one session, cfg4 58.2 vs 57.6-58.3), so the limit stays where it was tested longest)
      const bool a_inline = use256 && !same_a && !transA && ctcn_get_option("gemm_a_inline") != 0 && K >= 32 && K % 4 == 0 && lda % 4 == 0 &&
                            ((uintptr_t)A & 15) == 0 && ceil_div(N, 128 * wnt256) <= 5;
      if (a_inline) {
        pl.a_valid = false;                         // no A planes in the workspace after this call
        if (!same_b) split(B, ldb, transB == 0, N, bh, bl, b_shift);
        CTCN_LAUNCH_CHECK();
        const int lrc = wnt256 == 2 ? launch_planes256_af32<2>(st, M, N, K, Kp, A, lda, bh, bl, C, ldc, beta)
                                    : launch_planes256_af32<1>(st, M, N, K, Kp, A, lda, bh, bl, C, ldc, beta);
        if (lrc) return lrc;
        CTCN_LAUNCH_CHECK();
        return CTCN_OK;
      }
      if (!same_a) split(A, lda, transA != 0, M, ah, al, 0);
      pl.A = A; pl.lda = lda; pl.M = M; pl.Ka = K; pl.transA = transA; pl.ws_a = ws; pl.st_a = (void *)st;
      pl.a_valid = true;
      if (!same_b) split(B, ldb, transB == 0, N, bh, bl, b_shift);
      CTCN_LAUNCH_CHECK();
      // tile shape: 128x128 (two workgroups per CU).  Option gemm_big_tiles: 256x128 / 128x256 tiles (one per CU, 96 KB of LDS)
      // when they fill the device at least once without more padding -- 25 % fewer LDS fragment reads per MFMA, yet measured
      // 7 % SLOWER at cfg2 / 4 % at cfg4 (the second resident workgroup hides the staging barriers better), so off by default
      int shape = 0, ptm = tiles_m, ptn = tiles_n;
      {
        const long t22 = (long)tiles_m * tiles_n * 128 * 128;
        const int m42 = ceil_div(M, 256), n42 = ceil_div(N, 128), m24 = ceil_div(M, 128), n24 = ceil_div(N, 256);
        const long t42 = (long)m42 * n42 * 256 * 128, t24 = (long)m24 * n24 * 128 * 256;
        const int min_tiles = ctcn_device_cus();
        const bool ok42 = m42 * n42 >= min_tiles && t42 <= t22 + t22 / 32, ok24 = m24 * n24 >= min_tiles && t24 <= t22 + t22 / 32;
        if (ctcn_opt_gemm_big_tiles() && (ok42 || ok24)) {
          if (ok24 && (!ok42 || t24 <= t42)) { shape = 2; ptm = m24; ptn = n24; }
          else { shape = 1; ptm = m42; ptn = n42; }
        }
      }
      if (use256) {
        const int lrc256 = wnt256 == 2 ? launch_planes256<2>(st, M, N, Kp, ah, al, bh, bl, C, ldc, beta)
                                       : launch_planes256<1>(st, M, N, Kp, ah, al, bh, bl, C, ldc, beta);
        if (lrc256) return lrc256;
        CTCN_LAUNCH_CHECK();
        return CTCN_OK;
      }
      // side stream: the 256-row tiles from a queue when they fill the allowed XCDs' CUs at least 3/4 in whole rounds of one tile per CU
      // (the chunk GEMMs of the pipelined input projection are sized for that: rnn.hip)
      if (xcd_allow && ctcn_get_option("gemm_tile256") != 0 && ctcn_get_option("gemm_pingpong") != 0 && M >= 1024 && N >= 96) {
        const int cus = ctcn_device_cus() / std::max(1, std::min(ctcn_device_xcds(), 16)) * __builtin_popcount(xcd_allow);
        const int t256 = ceil_div(M, 256) * ceil_div(N, 128 * wnt256), rounds = ceil_div(t256, std::max(cus, 1));
        if (cus > 0 && t256 * 4 >= rounds * cus * 3) {
          const int lrc = wnt256 == 2 ? launch_planes256_queue<2>(st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, xcd_allow, queue)
                                      : launch_planes256_queue<1>(st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, xcd_allow, queue);
          if (lrc) return lrc;
          CTCN_LAUNCH_CHECK();
          return CTCN_OK;
        }
      }
      const int pnt = ptm * ptn;
      int psplits = 1;
      if (pnt < 256 && Kp >= 1024 && part_bytes >= (size_t)2 * M * N * sizeof(float)) {
        psplits = std::min(std::min(ceil_div(512, pnt), Kp / 512), (int)(part_bytes / ((size_t)M * N * sizeof(float))));
        if (psplits < 2) psplits = 1;
      }
      int pchunk = Kp;
      if (psplits > 1) {
        pchunk = ceil_div(ceil_div(Kp, psplits), PBK) * PBK;
        psplits = ceil_div(Kp, pchunk);
      }
      float *pp = psplits > 1 ? part : (float *)nullptr;
      int lrc;
      if (shape == 1) lrc = launch_planes<4, 2>(xcd_allow != 0, pnt, psplits, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, pchunk, pp, ptm, ptn, xcd_allow, queue);
      else if (shape == 2) lrc = launch_planes<2, 4>(xcd_allow != 0, pnt, psplits, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, pchunk, pp, ptm, ptn, xcd_allow, queue);
      else lrc = launch_planes<2, 2>(xcd_allow != 0, pnt, psplits, st, M, N, Kp, ah, al, bh, bl, C, ldc, beta, pchunk, pp, ptm, ptn, xcd_allow, queue);
      if (lrc) return lrc;
      CTCN_LAUNCH_CHECK();
      if (psplits > 1) {
        const size_t total = (size_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min((size_t)2048, ceil_div_z(total, 256))), dim3(256), 0, st, (const float *)part, C, M,
                           N, ldc, psplits, beta);
        CTCN_LAUNCH_CHECK();
      }
      return CTCN_OK;
    }
  }
  pl.a_valid = false; pl.b_valid = false;                        // not the plane path: nothing to reuse
  const bool vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
  const bool vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
  float *wsp = splits > 1 ? (float *)ws : nullptr;
  dim3 grid(nt, splits), block(256);
#define LAUNCH(KERN, TA, TB)                                                                             \
  hipLaunchKernelGGL((KERN<TA, TB>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc, beta, kchunk, wsp, \
                     tiles_m, tiles_n, vecA, vecB)
  if (precision == 0 && xcd_allow && ws && ws_bytes >= (size_t)splits * M * N * sizeof(float) * (splits > 1 ? 1 : 0) + 512) {
    // side stream next to a persistent recurrence: tiles from an atomic queue, workgroups off the allowed XCDs exit
    unsigned *queue = (unsigned *)((char *)ws + ((ws_bytes - 256) & ~(size_t)255));
    CTCN_HIP(hipMemsetAsync(queue, 0, 4, st));
    const dim3 qgrid(2 * ctcn_device_cus());
#define QLAUNCH(TA, TB)                                                                                              \
  hipLaunchKernelGGL((gemm_f32_queue_kernel<TA, TB>), qgrid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc, beta, kchunk, wsp, \
                     tiles_m, tiles_n, vecA, vecB, splits, xcd_allow, queue)
    if (transA) { if (transB) QLAUNCH(true, true); else QLAUNCH(true, false); }
    else        { if (transB) QLAUNCH(false, true); else QLAUNCH(false, false); }
#undef QLAUNCH
  } else if (precision == 0) {
    if (transA) { if (transB) LAUNCH(gemm_f32_kernel, true, true); else LAUNCH(gemm_f32_kernel, true, false); }
    else        { if (transB) LAUNCH(gemm_f32_kernel, false, true); else LAUNCH(gemm_f32_kernel, false, false); }
  } else {
    if (transA) { if (transB) LAUNCH(gemm_bf16x3_kernel, true, true); else LAUNCH(gemm_bf16x3_kernel, true, false); }
    else        { if (transB) LAUNCH(gemm_bf16x3_kernel, false, true); else LAUNCH(gemm_bf16x3_kernel, false, false); }
  }
#undef LAUNCH
  CTCN_LAUNCH_CHECK();
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int blocks = (int)min((size_t)2048, ceil_div_z(total, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float *)ws, C, M, N, ldc, splits, beta);
    CTCN_LAUNCH_CHECK();
  }
  return CTCN_OK;
}

// C = A^T * shift_k(B) for contraction-major A (K x M) and B (K x N): B_eff[k] = B[k - shift] when 0 <= k - shift < K, else 0 --
// the recurrent weight gradient dW_hh = da^T h_prev with h_prev = y delayed (forward direction) or advanced (reverse) by one
// timestep.  On the bf16x3 plane path the shift is applied while B is split, so that A = da^T keeps the SAME K window as in
// dW_ih = da^T x and its planes can be reused (GemmPlanes::same_a); elsewhere the window is narrowed instead.
int ctcn_gemm_on_xcds(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
                      int ldc, float beta, int precision, void *ws, size_t ws_bytes, void *stream, unsigned xcd_allow, GemmPlanes *planes) {
  return gemm_core(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, beta, precision, ws, ws_bytes, stream, xcd_allow, planes, 0);
}

int ctcn_gemm_shift_b(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, float beta, int precision,
                      void *ws, size_t ws_bytes, void *stream, unsigned xcd_allow, int shift, GemmPlanes *planes) {
  const int as = shift < 0 ? -shift : shift;
  CTCN_REQUIRE(as < K, "ctcn_gemm_shift_b: |shift| %d >= K %d", as, K);
  const int Kp = ceil_div(K, PBK) * PBK;
  const size_t plane_bytes = align_up(2 * ((size_t)M * Kp + (size_t)N * Kp) * sizeof(unsigned short), 256);
  const bool plane = precision == 1 && K >= 64 && ws && ws_bytes >= plane_bytes + 1024 &&
                     !tn_eligible(1, 0, M, N, K - as, A, lda, B, ldb, precision, ws, ws_bytes);   // (the TN tile narrows the window: no planes to share)
  if (!plane) {
    if (planes) planes->same_a = false;
    const float *A2 = shift > 0 ? A + (size_t)as * lda : A, *B2 = shift < 0 ? B + (size_t)as * ldb : B;
    return gemm_core(1, 0, M, N, K - as, A2, lda, B2, ldb, C, ldc, beta, precision, ws, ws_bytes, stream, xcd_allow, planes, 0);
  }
  return gemm_core(1, 0, M, N, K, A, lda, B, ldb, C, ldc, beta, precision, ws, ws_bytes, stream, xcd_allow, planes, shift);
}

extern "C" int ctcn_gemm(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B,
                         int ldb, float *C, int ldc, float beta, int precision, void *ws, size_t ws_bytes,
                         void *stream) {
  return gemm_core(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, beta, precision, ws, ws_bytes, stream, 0u, nullptr, 0);
}

// the same transposition of two equally shaped tensors in ONE launch (W_hh of the two directions before a backward recurrence)
__global__ void transpose01_pair_kernel(const float *__restrict__ in0, const float *__restrict__ in1, float *__restrict__ out0, float *__restrict__ out1,
                                        int A, int B, int C) {
  const float *in = blockIdx.y ? in1 : in0;
  float *out = blockIdx.y ? out1 : out0;
  const size_t total = (size_t)A * B * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const size_t ab = i / C;
    const int a = ab % A;
    const int b = ab / A;
    out[i] = in[((size_t)a * B + b) * C + c];
  }
}
// C == 1: a plain matrix transpose, out[b][a] = in[a][b], through a 32 x 33 LDS tile (both sides coalesced: 8.6 -> ~3 us for the
// two 1 280 x 320 W_hh of a cfg2 layer)
__global__ __launch_bounds__(256) void transpose2d_pair_kernel(const float *__restrict__ in0, const float *__restrict__ in1, float *__restrict__ out0,
                                                               float *__restrict__ out1, int A, int B) {
  __shared__ float tile[32][33];
  const float *in = blockIdx.z ? in1 : in0;
  float *out = blockIdx.z ? out1 : out0;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int b0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = a0 + ty + 8 * k, b = b0 + tx;
    if (a < A && b < B) tile[ty + 8 * k][tx] = in[(size_t)a * B + b];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = b0 + ty + 8 * k, a = a0 + tx;
    if (a < A && b < B) out[(size_t)b * A + a] = tile[tx][ty + 8 * k];
  }
}
int ctcn_transpose01_pair(const float *in0, const float *in1, float *out0, float *out1, int A, int B, int C, void *stream) {
  CTCN_REQUIRE(in0 && in1 && out0 && out1 && A > 0 && B > 0 && C > 0, "ctcn_transpose01_pair: bad args");
  if (C == 1) {
    hipLaunchKernelGGL(transpose2d_pair_kernel, dim3(ceil_div(B, 32), ceil_div(A, 32), 2), dim3(256), 0, (hipStream_t)stream, in0, in1, out0, out1, A, B);
    CTCN_LAUNCH_CHECK();
    return CTCN_OK;
  }
  const size_t total = (size_t)A * B * C;
  const int blocks = (int)min((size_t)4096, ceil_div_z(total, 256));
  hipLaunchKernelGGL(transpose01_pair_kernel, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, in0, in1, out0, out1, A, B, C);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_transpose01(const float *in, float *out, int A, int B, int C, void *stream) {
  CTCN_REQUIRE(in && out && A > 0 && B > 0 && C > 0, "ctcn_transpose01: bad args");
  const size_t total = (size_t)A * B * C;
  const int blocks = (int)min((size_t)4096, ceil_div_z(total, 256));
  hipLaunchKernelGGL(transpose01_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, A, B, C);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
