// rnn.hip -- bias-free (bi)directional LSTM / GRU / tanh-RNN layer, forward and backward, for gfx950.
//
// replaces: rnn_type(input_size, hidden_size, bidirectional, bias=False)(x) in BatchRNN.forward
// (reference timit/models/model_ctc.py:24-25,33; rnn_type chosen at timit/steps/train_ctc.py:20) and its
// autograd backward (train_ctc.py:63).  Arithmetic: SURVEY.md Appendix A.1/A.2/A.2b.
//
// Structure (per layer):
//   1. time-parallel input projection  Gx = X * W_ih^T  for all T*B rows: one MFMA GEMM per direction
//      (gemm.hip) written straight into the gate reserve (T,B,dirs,G*H).
//   2. the serial recurrence: ONE persistent launch per layer (rnn_fwd_persist / rnn_bwd_persist, below) when the grid can
//      be co-resident, else one launch per timestep (rnn_fwd_step / rnn_bwd_step), both directions in the same launch.
//      Per-timestep kernels: a workgroup owns a slice of hidden units for all batch rows:
//        forward : 4 hidden units x 4 gates = one 16-column MFMA N-tile (grid H/4 x dirs = 160 WGs at H=320)
//        backward: 16 hidden units, K = G*H (grid H/16 x dirs, 16 waves per WG to split the 4x longer K)
//      The recurrent matmul  h_prev[B,K] * W_slice[16,K]^T  runs on v_mfma_f32_16x16x4_f32 (exact f32).
//      K is split over the waves of the workgroup AND over the 4 k-lanes of the MFMA: lane (r,q) of wave w
//      owns a contiguous k-range, so every operand is fetched with 16-B global loads straight into VGPRs
//      (all loads of a step in flight at once: the step is latency-bound, not FLOP-bound); partial
//      accumulators are combined through LDS, then the gate non-linearities / cell update run in the
//      same kernel (fused epilogue) and write h_t, the saved activations and c_t.
//   3. backward-through-time mirrors it with W_hh^T; dW_ih, dW_hh and dX are deferred to three large
//      MFMA GEMMs over the saved d(pre-activation) slab (K = T*B) after the time loop.
//
// Layouts: x (T,B,I); y (T,B,dirs*H); gates (T,B,dirs,G*H); aux (T,B,dirs,H) [LSTM: c_t, GRU: W_hn*h_{t-1}].
// No packing / masking: all T padded frames are processed, reverse direction starts at t=T-1 (as torch).
#include <algorithm>
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include <type_traits>

namespace {

// Gate activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp each): |error| < 3e-7 absolute, a
// quarter of the instruction count of expf / tanhf + IEEE division -- they sit on the serial path of every timestep.
__device__ __forceinline__ float act_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float act_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(x));
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// bf16x3 recurrent matmul (precision = 1): every f32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and the
// product accumulates lo*hi + hi*lo + hi*hi in f32 on v_mfma_f32_16x16x32_bf16 (16 cycles per 32 k, against 8 x 32
// cycles for the same k on the f32 MFMA): the matmul leaves the serial path of the timestep almost entirely.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
union Bf16Pack { u32x4 u; bf16x8_t v; };
__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, bf16x8_t &hi, bf16x8_t &lo) {
  unsigned h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? a[e] : b[e - 4];
    h[e] = f2bf(x);
    l[e] = f2bf(x - __uint_as_float(h[e] << 16));
  }
  Bf16Pack ph, pl;
  ph.u = (u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  pl.u = (u32x4){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
  hi = ph.v; lo = pl.v;
}
// W operand of one lane: 8 consecutive k of row `brow` (null -> zeros), k beyond K are zeros
__device__ __forceinline__ void load_w8(const float *brow, int k0, int K, bf16x8_t &hi, bf16x8_t &lo) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float *src = brow ? brow : reinterpret_cast<const float *>(&zero);
  f32x4 a = zero, b = zero;
  if (brow) {
    const f32x4 va = *reinterpret_cast<const f32x4 *>(src + min(k0, K - 4));
    const f32x4 vb = *reinterpret_cast<const f32x4 *>(src + min(k0 + 4, K - 4));
    a = k0 < K ? va : zero;
    b = k0 + 4 < K ? vb : zero;
  }
  split8(a, b, hi, lo);
}

struct RnnArgs {
  int cell, T, B, H, D, G, step;
  const float *w0, *w1;  // fwd: W_hh (G*H,H) per direction; bwd: W_hh^T (H,G*H)
  float *y, *gates, *aux;
  const float *dy;
  float *state;  // bwd: carried dc (LSTM) / dh*z (GRU), (B,D,H)
};

// Recurrent matmul helper.  acc[mt] (16x16 tile: rows = batch mt*16.., cols = this WG's 16 columns)
// += A[rows, K] * Bm[16 cols, K]^T.  K is split over the `nwaves` waves of the workgroup (wave w owns the
// contiguous range [w*16*KQ4, (w+1)*16*KQ4) of every super-chunk of nwaves*16*KQ4 floats) and, inside a wave,
// over the 4 k-lanes q of the MFMA with a 16-byte interleave: lane (r,q) loads the float4 at
// k = base + s*16 + q*4, so one load instruction covers a contiguous 64-B segment of each of the 16 rows
// (16 cache lines per wave-instruction instead of 64) and all loads of the step are in flight at once.
// A element (row,k) lives at a1[row*ld1 + k] for k < ksplit, else a2[row*ld2 + k - ksplit].
template <int MT, int KQ4>
__device__ __forceinline__ void rec_mm(const float *__restrict__ a1, int ld1, int ksplit, const float *__restrict__ a2,
                                       int ld2, int rows, const float *__restrict__ brow, bool bvalid, int K,
                                       int nwaves, int wave, int q, int r, f32x4 (&acc)[MT]) {
  const int sc_stride = nwaves * 16 * KQ4;
  const int nsc = (K + sc_stride - 1) / sc_stride;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // Every load below is UNCONDITIONAL on a clamped (always valid) address and masked afterwards with a register
  // select: a predicated load makes hipcc branch around it and wait for it alone, which serialises the operand
  // fetches of a step into dependent L2 round trips.  Operands are native ext-vector values (not HIP's float4
  // class) so that the fully unrolled arrays stay in VGPRs instead of scratch.
  const float *brow_s = bvalid ? brow : a1;
  const int rmax = rows - 1;
  for (int sc = 0; sc < nsc; ++sc) {
    const int kb = sc * sc_stride + wave * 16 * KQ4 + q * 4;
    f32x4 av[MT][KQ4], bv[KQ4];
#pragma unroll
    for (int s = 0; s < KQ4; ++s) {
      const int k = kb + 16 * s;
      const int kc = min(k, K - 4);
      bv[s] = *reinterpret_cast<const f32x4 *>(brow_s + (bvalid ? kc : 0));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = min(mt * 16 + r, rmax);
        const float *p = (kc < ksplit) ? a1 + (size_t)row * ld1 + kc : a2 + (size_t)row * ld2 + (kc - ksplit);
        av[mt][s] = *reinterpret_cast<const f32x4 *>(p);
      }
    }
#pragma unroll
    for (int s = 0; s < KQ4; ++s) {
      const bool kin = (kb + 16 * s) < K;
      bv[s] = (bvalid && kin) ? bv[s] : zero;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt][s] = (kin && mt * 16 + r < rows) ? av[mt][s] : zero;
    }
#pragma unroll
    for (int s = 0; s < KQ4; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s][c], bv[s][c], acc[mt], 0, 0, 0);
  }
}

// Combine the NW per-wave partial tiles through LDS (fixed summation order -> deterministic), at most
// PW waves per pass so the staging buffer stays small.  Tile element (row, col) -> outs[row][col].
template <int MT, int NW, int PW>
__device__ __forceinline__ void reduce_tiles(const f32x4 (&acc)[MT], float *red /*[PW][MT*256]*/,
                                             float (*outs)[17], int tid, int nthreads) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int pass = 0; pass < NW / PW; ++pass) {
    if (wave / PW == pass) {
      const int wl = wave - pass * PW;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wl * MT + mt) * 256 + lane * 4 + e] = acc[mt][e];
    }
    lds_barrier();
    for (int i = tid; i < MT * 256; i += nthreads) {
      const int mt = i >> 8, le = i & 255;
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < PW; ++w) s += red[(w * MT + mt) * 256 + le];
      const int ln = le >> 2, reg = le & 3;
      // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
      float *o = &outs[mt * 16 + (ln >> 4) * 4 + reg][ln & 15];
      *o = pass == 0 ? s : *o + s;
    }
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// forward step.  grid = (H/HS, dirs, ceil(B / (16*MT))), 256 threads.
// The epilogue's global operands (input-projection pre-activations, c_{t-1} / h_{t-1}) are fetched BEFORE the
// recurrent matmul so that their HBM latency overlaps the operand loads of the matmul.
// ------------------------------------------------------------------------------------------------
template <int MT, int KQ4>
__global__ __launch_bounds__(256) void rnn_fwd_step(RnnArgs p) {
  constexpr int NW = 4;
  __shared__ float red[NW * MT * 256];
  __shared__ float outs[MT * 16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int d = blockIdx.y, b0 = blockIdx.z * (16 * MT);
  const int H = p.H, G = p.G, D = p.D, B = p.B;
  const int Bc = min(16 * MT, B - b0);
  const int t = d == 0 ? p.step : p.T - 1 - p.step;
  const int tp = p.step == 0 ? -1 : (d == 0 ? t - 1 : t + 1);
  const bool tanh_cell = p.cell == CTCN_CELL_TANH;
  const int HS = tanh_cell ? 16 : 4;
  const int j0 = blockIdx.x * HS;
  const int gate = tanh_cell ? 0 : (r >> 2), jj = tanh_cell ? r : (r & 3);
  const bool bvalid = gate < G && (j0 + jj) < H;
  const float *W = d == 0 ? p.w0 : p.w1;
  const float *brow = W + (size_t)(gate * H + j0 + jj) * H;

  // epilogue work item of this thread (LSTM/GRU: MT*16 rows x 4 units <= 256 items -> one per thread)
  const int bl = tid / HS, jl = tid - bl * HS;
  const int j = j0 + jl, b = b0 + bl;
  const bool item = !tanh_cell && bl < Bc && j < H;
  const size_t row_t = (size_t)t * B + b;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float prev = 0.0f;   // c_{t-1} (LSTM) / h_{t-1} (GRU)
  if (item) {
    const float *gt = p.gates + (row_t * D + d) * (size_t)(G * H);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < G) pre[k] = gt[k * H + j];
    if (tp >= 0)
      prev = p.cell == CTCN_CELL_LSTM ? p.aux[(((size_t)tp * B + b) * D + d) * H + j]
                                      : p.y[((size_t)tp * B + b) * D * H + d * H + j];
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tp >= 0) {
    const float *abase = p.y + ((size_t)tp * B + b0) * D * H + d * H;
    rec_mm<MT, KQ4>(abase, D * H, H, abase, D * H, Bc, brow, bvalid, H, NW, wave, q, r, acc);
  }
  reduce_tiles<MT, NW, 4>(acc, red, outs, tid, 256);

  if (item) {
    float *gt = p.gates + (row_t * D + d) * (size_t)(G * H);
    float *yt = p.y + row_t * D * H + d * H;
    if (p.cell == CTCN_CELL_LSTM) {
      const float i_ = act_sigmoid(outs[bl][0 * 4 + jl] + pre[0]);
      const float f_ = act_sigmoid(outs[bl][1 * 4 + jl] + pre[1]);
      const float g_ = act_tanh(outs[bl][2 * 4 + jl] + pre[2]);
      const float o_ = act_sigmoid(outs[bl][3 * 4 + jl] + pre[3]);
      const float c = f_ * prev + i_ * g_;
      gt[0 * H + j] = i_; gt[1 * H + j] = f_; gt[2 * H + j] = g_; gt[3 * H + j] = o_;
      p.aux[(row_t * D + d) * H + j] = c;
      yt[j] = o_ * act_tanh(c);
    } else {
      const float hn = outs[bl][2 * 4 + jl];
      const float r_ = act_sigmoid(outs[bl][0 * 4 + jl] + pre[0]);
      const float z_ = act_sigmoid(outs[bl][1 * 4 + jl] + pre[1]);
      const float n_ = act_tanh(pre[2] + r_ * hn);
      gt[0 * H + j] = r_; gt[1 * H + j] = z_; gt[2 * H + j] = n_;
      p.aux[(row_t * D + d) * H + j] = hn;
      yt[j] = (1.0f - z_) * n_ + z_ * prev;
    }
  }
  if (tanh_cell) {
    for (int it = tid; it < Bc * 16; it += 256) {
      const int bl2 = it >> 4, jl2 = it & 15;
      if (j0 + jl2 >= H) continue;
      const size_t rt = (size_t)t * B + b0 + bl2;
      p.y[rt * D * H + d * H + j0 + jl2] = act_tanh(outs[bl2][jl2] + p.gates[(rt * D + d) * (size_t)H + j0 + jl2]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward step (processing order reversed: the forward direction walks t = T-1..0).
// grid = (ceil(H/16), dirs, ceil(B/16)), 1024 threads: one 16(batch) x 16(hidden) tile of
// dh_rec = d(pre-act)_{next} * W_hh per workgroup, K = G*H split over 16 waves.
// ------------------------------------------------------------------------------------------------
template <int KQ4>
__global__ __launch_bounds__(1024) void rnn_bwd_step(RnnArgs p) {
  constexpr int NW = 16, PW = 8, MT = 1;
  __shared__ float red[PW * MT * 256];
  __shared__ float outs[MT * 16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int d = blockIdx.y, b0 = blockIdx.z * 16;
  const int H = p.H, G = p.G, D = p.D, B = p.B, T = p.T;
  const int Bc = min(16, B - b0);
  const int GH = G * H;
  // forward direction: t = T-1-step, processed-just-before = t+1, previous-in-forward-order = t-1
  const int t = d == 0 ? T - 1 - p.step : p.step;
  const int tn = p.step == 0 ? -1 : (d == 0 ? t + 1 : t - 1);
  const int tp = d == 0 ? (t > 0 ? t - 1 : -1) : (t < T - 1 ? t + 1 : -1);
  const int j0 = blockIdx.x * 16;
  const bool bvalid = (j0 + r) < H;
  const float *WT = d == 0 ? p.w0 : p.w1;
  const float *brow = WT + (size_t)(j0 + r) * GH;

  // epilogue item of this thread and its global operands, fetched ahead of the matmul
  const int bl = tid >> 4, jl = tid & 15, j = j0 + jl, b = b0 + bl;
  const bool item = tid < 256 && bl < Bc && j < H;
  const size_t row_t = (size_t)t * B + b;
  float sv[4] = {0.f, 0.f, 0.f, 0.f}, dyv = 0.f, e0 = 0.f, e1 = 0.f, stv = 0.f;
  if (item) {
    const float *gt = p.gates + (row_t * D + d) * (size_t)GH;
    dyv = p.dy[row_t * D * H + d * H + j];
    if (p.cell == CTCN_CELL_TANH) {
      e0 = p.y[row_t * D * H + d * H + j];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < G) sv[k] = gt[k * H + j];
      e0 = p.aux[(row_t * D + d) * H + j];                                   // c_t (LSTM) / W_hn h_{t-1} (GRU)
      if (tp >= 0)
        e1 = p.cell == CTCN_CELL_LSTM ? p.aux[(((size_t)tp * B + b) * D + d) * H + j]   // c_{t-1}
                                      : p.y[((size_t)tp * B + b) * D * H + d * H + j];  // h_{t-1}
      stv = p.state[((size_t)b * D + d) * H + j];
    }
  }

  f32x4 acc[MT];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tn >= 0) {
    const size_t rown = (size_t)tn * B + b0;
    const float *a1 = p.gates + (rown * D + d) * (size_t)GH;
    if (p.cell == CTCN_CELL_GRU) {
      const float *a2 = p.aux + (rown * D + d) * (size_t)H;   // d(W_hn h) = dan * r
      rec_mm<MT, KQ4>(a1, D * GH, 2 * H, a2, D * H, Bc, brow, bvalid, GH, NW, wave, q, r, acc);
    } else {
      rec_mm<MT, KQ4>(a1, D * GH, GH, a1, D * GH, Bc, brow, bvalid, GH, NW, wave, q, r, acc);
    }
  }
  reduce_tiles<MT, NW, PW>(acc, red, outs, tid, 1024);

  if (item) {
    float *gt = p.gates + (row_t * D + d) * (size_t)GH;
    float dh = dyv + outs[bl][jl];
    if (p.cell == CTCN_CELL_LSTM) {
      const float i_ = sv[0], f_ = sv[1], g_ = sv[2], o_ = sv[3];
      const float tc = act_tanh(e0);
      const float do_ = dh * tc;
      const float dc = dh * o_ * (1.0f - tc * tc) + stv;
      gt[0 * H + j] = dc * g_ * i_ * (1.0f - i_);
      gt[1 * H + j] = dc * e1 * f_ * (1.0f - f_);
      gt[2 * H + j] = dc * i_ * (1.0f - g_ * g_);
      gt[3 * H + j] = do_ * o_ * (1.0f - o_);
      p.state[((size_t)b * D + d) * H + j] = dc * f_;
    } else if (p.cell == CTCN_CELL_GRU) {
      dh += stv;
      const float r_ = sv[0], z_ = sv[1], n_ = sv[2], hn = e0, hp = e1;
      const float dn = dh * (1.0f - z_);
      const float dz = dh * (hp - n_);
      const float dan = dn * (1.0f - n_ * n_);
      gt[0 * H + j] = dan * hn * r_ * (1.0f - r_);
      gt[1 * H + j] = dz * z_ * (1.0f - z_);
      gt[2 * H + j] = dan;
      p.aux[(row_t * D + d) * H + j] = dan * r_;
      p.state[((size_t)b * D + d) * H + j] = dh * z_;
    } else {
      gt[j] = dh * (1.0f - e0 * e0);
    }
  }
}

// ================================================================================================
// Persistent forward recurrence: ONE launch per layer (DESIGN.md section 5 has the measured history).
//
// A per-timestep launch costs 7.0 us = 2.8 us dependent-launch floor + 2.2 us cold operand fetch (the launch boundary
// invalidates every XCD's L2) + 2.0 us of work.  Here a workgroup owns `hsu` hidden units (all gates) of ONE 16-row
// batch tile for the whole sequence: its W_hh slice lives in VGPRs (f32, or hi/lo bf16 planes at precision 1), c / h
// of its (row, unit) items in one register per thread, and per step it exchanges only h_t with the other workgroups
// of its (direction, batch tile) GROUP:
//   * placement: in XCD-local mode all workgroups of a group run on one XCD (persist_role), so the hand-off goes through
//     that XCD's L2: plain write-back stores, L1-bypassing (sc1) loads; 0.3 us one way instead of 0.6 us across the fabric
//     with write-through stores (tools/mb_xcd.hip).  Device-scope mode (same code, sc1 stores) is the fallback;
//   * publish: the items' h_t go through a 1-KB LDS image; ONE communication wave stores them as 16-B granules in the
//     order the consumers' MFMA lanes read them, drains its stores, then one lane stores the workgroup's flag = step+1;
//   * consume: the communication wave polls the group's flags (one per lane, 2 polls in flight), a barrier releases the
//     other waves, and every wave loads its K-slice of the tile with fully coalesced 1-KB loads straight into the MFMA
//     A-operand registers (no LDS staging);
//   * everything else stays off that chain: the reserve stores (gates, c, y) and the next step's pre-activation loads are
//     issued by the item waves after the hand-off; barriers wait for LDS only.
// Two parity buffers suffice (a workgroup can publish step s+1 only after every producer of its tile published step s,
// i.e. after all reads of step s-1).  Every spin is bounded: on a timeout the sticky status word is set, the layer
// output is poisoned with NaN and all workgroups leave.
// Hand-off variants measured and rejected (cfg2 forward, us per step; THIS = 3.1 f32 / 2.2 bf16x3): per-timestep launches
// 7.0 | 8-B {value,tag} granules swept into registers 5.9 | device-scope flag + 16-B write-through payload + LDS fill 4.9 |
// per-wave flags 5.7-5.9 | 8-B granule tiles without flags 5.5 | 16-B {v0,v1,v2,tag} granules swept block-wide 6.9 |
// 2-3 polls in flight / paced polls over the fabric 5.0-5.7 | buffer_inv sc0 + plain loads: stale L1 lines (incorrect) |
// XCD-local, no flags at all: {bf16 hi, lo} words with the step number in the LSBs of lo, every wave re-loading its stale
// granules: 2.3-2.5, i.e. no gain over flag + data (2.2) -- the re-load traffic of 160 spinning waves delays the stores |
// the same with the bwd_scatter protocol (tag in the LSB of every dword, one chunk polled, the other five fetched together
// and validated): 2.20 vs 2.03 -- here the polling waves are the item waves, whose polls return behind their own reserve
// stores / pre-activation loads (in-order vm queue), and six chunks cost six tag reductions per step |
// slices of 10 units (32 workgroups per group, one per CU, 4-byte publish pieces): 2.37 -- the flag wait (~2 100 cycles) is not
// the doubled-up CUs, it is the poll round trip itself; slices of 4: 2.55; 1 / 3 / 4 polls in flight: 2.08 / 2.09 / 2.15 |
// 5 waves per workgroup (16 units = 4 item waves + a communication wave of its own, 20 workgroups per group, one per CU, no
// stragglers): flag wait 2100 -> 1500 cycles but tile loads and publish longer, 2.05 us either way |
// mixed slices (round 2, option "rnn_mixed_slices"): 24 x 12 + 8 x 4 units = one workgroup per CU of the XCD, no CU hosting two:
// 2.05 vs 1.92 us -- again the one-per-CU geometry loses; two workgroups interleaving on a CU hide each other's latencies.
// grid: device scope (slices, dirs, batch tiles); XCD-local nx * (wpx + spare) x 1 x 1.  256 threads, all working
// workgroups co-resident (occupancy-checked on the host).
// ================================================================================================
// Bound of every in-launch spin, in poll round trips (~0.25-0.6 us each): ~10-20 s.  Long enough for a foreign resident kernel -- an
// RCCL all-reduce parked on some CUs until its slowest peer arrives -- to finish and let the rest of the grid become resident
// (DESIGN.md section 6); short enough that a genuinely lost role still ends the launch (status word set, output poisoned).
constexpr int SPIN_LIMIT = 1 << 25;

struct PersistArgs {
  RnnArgs a;
  float *hx;            // fwd: [2 parity][D][btiles][16][H] h payload;  bwd: [2][D][btiles][16][G*H] d(pre-act) payload
  unsigned *flags;      // [2 parity][D][btiles][nslices]
  int *status;
  int spin_limit;
  int local;            // 1: every (direction, batch-tile) group sits on ONE XCD and hands off through that XCD's L2
                        //    (write-back stores + L1-bypassing loads: ~0.3 us one way); 0: device scope (write-through, ~0.6 us)
  int nx;               // XCDs of the device (local mode: group g lives on XCD g % nx; see persist_role)
  int nsl, nbt;         // slices per group, batch tiles
  int wpx;              // local mode: working workgroups per XCD
  unsigned *tickets;    // local mode: nx zeroed counters (role tickets per XCD)
  int hsu;              // forward: hidden units per workgroup (<= 4*NT)
  int poll_delay;       // rnn_fwd_tagged: 64-cycle sleeps between the barrier and an exchange wave's first poll of a step
  int chunk_T, nchunk;  // rnn_fwd_tagged, pipelined input projection: frames per time chunk (0 = all pre-activations are there at launch)
  unsigned *chunk_ready; //   ... and the counter the side stream raises after each chunk PAIR (p covers chunks p and nchunk-1-p)
  int slow, slow_x;      // options "rnn_slow_items" / "rnn_slow_exchange" (parity harness): the SLOW instantiations of the persistent kernels -- item / exchange waves sleep this many x 64 cycles at the start of their phase of every step
  int xperm;             // option "xcd_interleave" (eight XCDs): which physical XCD is logical XCD g, i.e. hosts group g (persist_role)
  int nbig, hsu_small;  // forward, mixed slices (nbig > 0): slices 0 .. nbig-1 own `hsu` units each, the others `hsu_small`
  int poll_depth;       // XCD-local mode: flag polls kept in flight (1..4)
  int rsv_nt;           // rnn_bwd_scatter2: reserve loads / stores carry the non-temporal hint (streaming data must not evict the exchange tiles from L2)
  int tagmode;          // rnn_bwd_scatter: 1 = no flags, every float of a partial block carries the step tag in its LSB and the
                        // gathering wave polls the block itself; 0 = stores drained, then a flag per block
  float *ydrop;         // rnn_fwd_tagged: when set, the inverted dropout of y (Philox4x32-10, the dropout kernel's counters) is stored here as well
  int drop_bwd;         // rnn_bwd_scatter: dy is the gradient of the DROPPED output: the keep mask (same counters) is applied as dy is consumed
  float drop_p, drop_scale;
  unsigned long long drop_seed, drop_off;
#ifdef CTCN_PERSIST_STATS
  long long *stats;   // development instrumentation (tools/mb_step.hip only)
#endif
};


__device__ __forceinline__ f32x4 ld_sc1_f4(const __amdgpu_buffer_rsrc_t &rs, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);          // aux 16 = sc1: bypass L1
  return (f32x4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ void st_sc1_f4(const __amdgpu_buffer_rsrc_t &rs, unsigned byte_off, f32x4 v) {
  const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, byte_off, 0, 16);                        // write-through
}

__device__ __forceinline__ void st_f4(const __amdgpu_buffer_rsrc_t &rs, unsigned byte_off, f32x4 v, int local) {
  const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  if (local) __builtin_amdgcn_raw_buffer_store_b128(u, rs, byte_off, 0, 0);              // write-back into this XCD's L2
  else __builtin_amdgcn_raw_buffer_store_b128(u, rs, byte_off, 0, 16);                   // write-through
}
__device__ __forceinline__ void st_f2(const __amdgpu_buffer_rsrc_t &rs, unsigned byte_off, float a, float b, int local) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 u = {__float_as_uint(a), __float_as_uint(b)};
  if (local) __builtin_amdgcn_raw_buffer_store_b64(u, rs, byte_off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b64(u, rs, byte_off, 0, 16);
}
__device__ __forceinline__ void st_u1(const __amdgpu_buffer_rsrc_t &rs, unsigned byte_off, unsigned v, int local) {
  if (local) __builtin_amdgcn_raw_buffer_store_b32(v, rs, byte_off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(v, rs, byte_off, 0, 16);
}

// Wave 0 polls `n` flags (lane-strided relaxed L1-bypassing loads) until all equal `want`; false on timeout / global
// abort.  Keeping 2-3 polls in flight or pacing them was measured slower (extra traffic delays the flag itself).
__device__ __forceinline__ bool poll_flags(const unsigned *flags, int n, unsigned want, int lane, int spin_limit, int *status) {
  for (int spins = 0;; ++spins) {
    bool ok = true;
    for (int i = lane; i < n; i += 64)
      ok = ok && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
    if (__all(ok)) return true;
    if (spins > spin_limit) return false;
    if ((spins & 255) == 255 && status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// XCD-local variant for n <= 64 flags (one per lane): DEPTH polls stay in flight, re-issued as each returns, so a flag
// that lands between two polls is seen after ~RT/DEPTH instead of a full L2 round trip (the flag line lives in this
// XCD's L2, whose bandwidth the extra polls do not dent; over the fabric the same trick was measured slower).
template <int DEPTH>
__device__ __forceinline__ bool poll_flags_pipelined(const unsigned *flags, int n, unsigned want, int lane, int spin_limit, int *status) {
  const unsigned *p = flags + min(lane, n - 1);
  const bool idle = lane >= n;
  unsigned v[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) {
    v[i] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i + 1 < DEPTH) __builtin_amdgcn_s_sleep(2);
  }
  for (int spins = 0;; ++spins) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      if (__all(idle || v[i] == want)) return true;
      v[i] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (spins > spin_limit) return false;
    if ((spins & 63) == 63 && status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
  }
}
__device__ __forceinline__ bool poll_group(const unsigned *flags, int n, unsigned want, int lane, const PersistArgs &pa) {
  if (pa.local && n <= 64) {
    if ((pa.poll_depth & 255) == 2) return poll_flags_pipelined<2>(flags, n, want, lane, pa.spin_limit, pa.status);
    if ((pa.poll_depth & 255) == 3) return poll_flags_pipelined<3>(flags, n, want, lane, pa.spin_limit, pa.status);
    if ((pa.poll_depth & 255) >= 4) return poll_flags_pipelined<4>(flags, n, want, lane, pa.spin_limit, pa.status);
  }
  return poll_flags(flags, n, want, lane, pa.spin_limit, pa.status);
}

// Item traffic of the persistent kernels (reserve stores, next-step operand loads) goes through buffer instructions: ONE
// loop-invariant resource per tensor, a per-thread byte offset that never changes, and the timestep as the scalar
// offset operand -- no 64-bit VGPR address arithmetic and no selects on loaded values in the loop (both made hipcc wait
// on vmcnt in the middle of the "fire and forget" tail, which put the HBM latency of the reserve traffic on the critical
// path of every timestep).  The host only takes the persistent path when every tensor is smaller than 4 GB.
// What is left of that cost is the in-order vmcnt itself: the next step's operand-tile loads of an item wave queue behind
// its HBM-latency reserve loads (0.3 us per backward step with, 0 without them; warming L2 two steps ahead from the same
// wave only moves the stall).  Dedicated memory waves (4 matmul-free waves moving the reserve traffic through LDS) were
// built and measured too: item tail 1400 -> 380 cycles but 3.0 us per step instead of 2.8 -- the cost follows the HBM
// misses of the CU, not the wave that issues them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_rsrc(const float *base, size_t floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, base ? (int)(unsigned)(floats * 4) : 0, 0x00020000);
}
__device__ __forceinline__ float ld_slab(const __amdgpu_buffer_rsrc_t &rs, unsigned off, unsigned slab_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, slab_off, 0));
}
__device__ __forceinline__ void st_slab(const __amdgpu_buffer_rsrc_t &rs, unsigned off, unsigned slab_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off, slab_off, 0);
}

// The same store, invisible to hipcc's s_waitcnt bookkeeping.  gfx9 counts loads and stores in ONE counter (vmcnt), loads
// return in order among themselves but stores do not, so as soon as a wave has a load AND a store pending hipcc waits with
// vmcnt(0) for any loaded register -- which, for an item wave, turns "loaded two steps ahead" into "loaded one step ahead".
// With the stores hidden it sees loads only and emits the counted wait (vmcnt(7): the seven younger loads may stay in
// flight).  That wait is still sufficient: completions >= pending - 7, of which at most all stores, and the loads among
// them are the oldest ones.  Dword stores only (no store-data hazard), the "memory" clobber keeps the order.
__device__ __forceinline__ void st_slab_untracked(const __amdgpu_buffer_rsrc_t &rs, unsigned off, unsigned slab_off, float v) {
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen" ::"v"(v), "v"(off), "s"(rs), "s"(slab_off) : "memory");
}

// ... and the matching load: issued by hand INTO `dst` and waited for by hand (`s_waitcnt vmcnt(7)` + the registers as
// "+v" operands right before their first use), because with loads of two steps in flight across the back-edge of the
// step loop hipcc falls back to vmcnt(0) for the older set even when it sees nothing but loads.
__device__ __forceinline__ void ld_slab_untracked(float &dst, const __amdgpu_buffer_rsrc_t &rs, unsigned off, unsigned slab_off) {
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(dst) : "v"(off), "s"(rs), "s"(slab_off) : "memory");
}

// Role of a workgroup in a persistent launch.  Device-scope mode: grid (slices, dirs, batch tiles).  XCD-local mode:
// a 1-D grid of nx * (wpx + spare) workgroups.  The hardware deals workgroups round-robin over the XCDs (starting
// wherever the previous dispatch stopped), so every XCD receives >= wpx of them; each workgroup reads the XCD it
// actually runs on (HW_REG_XCC_ID) and draws a ticket there: ticket k on XCD x is slice k % nsl of group
// (k / nsl) * nx + x, tickets >= wpx (and groups past the end) exit at once.  All slices of a group therefore share
// one L2 no matter how the dispatcher ordered them; a role that never shows up ends in the bounded-spin abort.
struct PersistRole { int slice, d, bt; bool active; };
__device__ __forceinline__ PersistRole persist_role(const PersistArgs &pa, int D, int *s_ticket) {
  PersistRole r;
  if (!pa.local) { r.slice = blockIdx.x; r.d = blockIdx.y; r.bt = blockIdx.z; r.active = true; return r; }
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  const int xcd = (int)(x & 15);
  if (threadIdx.x == 0) *s_ticket = xcd < pa.nx ? (int)atomicAdd(pa.tickets + xcd, 1u) : 0x7fffffff;
  __syncthreads();
  const int idx = __builtin_amdgcn_readfirstlane(*s_ticket);   // uniform by construction: keep the role (and every address derived from it) in SGPRs
  // logical XCD of the physical one, 4 bits each (option "xcd_interleave" = 1 .. 5, eight XCDs): the inverse of the orders {0 2 4 6 1 3 5 7},
  // {0 1 4 5 2 3 6 7}, {0 3 4 7 1 2 5 6}, {0 2 5 7 1 3 4 6}, {0 4 1 5 2 6 3 7} (ops._XCD_ORDERS)
  constexpr unsigned k_logical[6] = {0x76543210u, 0x73625140u, 0x76325410u, 0x37621540u, 0x37265140u, 0x75316420u};
  const int lxcd = pa.xperm ? (int)((k_logical[pa.xperm] >> (4 * xcd)) & 15u) : xcd;
  const int lg = idx / pa.nsl, group = lg * pa.nx + lxcd;
  r.slice = idx - lg * pa.nsl;
  r.active = idx < pa.wpx && group < D * pa.nbt;
  r.d = group % D; r.bt = group / D;
  return r;
}

// Cross-wave reduction of the persistent kernels: every wave parks its partial C tiles in LDS, ONE barrier, then each item
// thread adds up the partials of exactly the elements it needs, in wave order (the same fixed order as reduce_tiles, so
// the sums are bit-identical).  A 16x16 tile occupies 4 row-quads x 96 floats (64 used) and tiles are 400 floats apart:
// the quads / tiles an item wave touches then fall into disjoint bank groups and every read is conflict-free.
constexpr int RT_Q = 96, RT_T = 400;
__device__ __forceinline__ void park_tile(float *red, int slot, int lane, const f32x4 &acc) {
  *reinterpret_cast<f32x4 *>(red + slot * RT_T + (lane >> 4) * RT_Q + (lane & 15) * 4) = acc;   // C map: col = lane & 15, rows 4*(lane>>4) + 0..3
}
__device__ __forceinline__ int parked_at(int row, int col) { return (row >> 2) * RT_Q + col * 4 + (row & 3); }

// CELL >= 0: the cell type as a compile-time constant (branch-free gate math; instantiated for the LSTM at precision 1);
// CELL = -1: read from the arguments
template <int NT, int KQ4, int PREC, int CELL>
__global__ __launch_bounds__(256) void rnn_fwd_persist(PersistArgs pa) {
  constexpr int NW = 4;
  // parked partials, in 16-B slots: tile stride 72, unit stride 18 (16 rows + 2): 72 = 8 and 18 = 2 (mod 16) spread the 16 lanes
  // of a ds_read_b128 group (2 rows x 4 units x 2 tiles of an item wave) over all 16 slots of the 256-B bank window
  constexpr int PK_T = 72, PK_Q = 18;
  const RnnArgs &p = pa.a;
  __shared__ __attribute__((aligned(16))) float red[NW * NT * RT_T];
  __shared__ __attribute__((aligned(16))) float hpub[16][16];   // h_t of this workgroup's units: f32 [row][unit], or (precision 1) two bf16 planes [hi|lo][row][unit]
  __shared__ int s_abort;
  __shared__ int s_pub;                     // step whose h block the communication wave has handed over (release of the reserve traffic)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int cell = CELL >= 0 ? CELL : pa.a.cell;
  const int G = CELL == CTCN_CELL_LSTM ? 4 : (CELL == CTCN_CELL_GRU ? 3 : (CELL == CTCN_CELL_TANH ? 1 : p.G));
  const int H = p.H, D = p.D, B = p.B, T = p.T;
  __shared__ int s_ticket;
  const PersistRole role = persist_role(pa, D, &s_ticket);
  if (!role.active) return;
  const int d = role.d, bt = role.bt, nbt = pa.nbt, nsl = pa.nsl, slice = role.slice, local = pa.local;
  const int b0 = bt * 16;
  const int Bc = min(16, B - b0);
  const bool tanh_cell = cell == CTCN_CELL_TANH;
  // hidden units owned by this workgroup (<= 4*NT; 16 for tanh).  Mixed slices: one workgroup per CU of the XCD with a few light
  // ones (e.g. H = 320 on 32 CUs: 24 x 12 + 8 x 4 units) instead of 40 equal slices of which 16 share 8 CUs and set the period
  const bool small_slice = pa.nbig > 0 && slice >= pa.nbig;
  const int HSU = small_slice ? pa.hsu_small : pa.hsu;
  const int j0 = small_slice ? pa.nbig * pa.hsu + (slice - pa.nbig) * pa.hsu_small : slice * pa.hsu;
  const float *W = d == 0 ? p.w0 : p.w1;
  const int kb = wave * 16 * KQ4 + q * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (tid == 0) { s_abort = 0; s_pub = 0; }

  // W_hh slice of this lane, resident for the whole sequence: tile nt covers units j0+4nt..+3 (x 4 gates)
  constexpr int KB = (KQ4 + 1) / 2;                       // precision 1: 32-k blocks per wave (4 * KB * 32 >= H)
  f32x4 bv[NT][KQ4];
  bf16x8_t whi[NT][KB], wlo[NT][KB];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    // W_hh is the FIRST MFMA operand here (C = W h^T): its fragment row r is output column "unit r >> 2, gate r & 3" of
    // tile nt, so that a lane's four C registers (rows 4q .. 4q+3 of C) are the FOUR GATES of unit q for batch row lane & 15
    const int gate = tanh_cell ? 0 : (r & 3), jj = tanh_cell ? r : (nt * 4 + (r >> 2));
    const bool bvalid = gate < G && jj < HSU && (j0 + jj) < H;
    const float *brow = bvalid ? W + (size_t)(gate * H + j0 + jj) * H : W;
    if constexpr (PREC == 0) {
#pragma unroll
      for (int s = 0; s < KQ4; ++s) {
        const int k = kb + 16 * s;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(brow + min(k, H - 4));
        bv[nt][s] = (bvalid && k < H) ? v : zero;
      }
    } else {
#pragma unroll
      for (int i = 0; i < KB; ++i) load_w8(bvalid ? brow : nullptr, 32 * (wave * KB + i) + 8 * q, H, whi[nt][i], wlo[nt][i]);
    }
  }
  // published h tile: 16-float column chunks, each stored in MFMA-A lane order [k-quad q][row][4 floats], so that a
  // consuming lane (row r = lane & 15, quad q = lane >> 4) reads ITS float4 of chunk c at (c * 64 + lane) * 16 bytes:
  // one fully coalesced 1-KB load per chunk, straight into the MFMA operand registers (no LDS staging, no fill barrier)
  // precision 1: 32-column blocks of two bf16 planes, [block][hi | lo][k-octet q][row][8 bf16]: the lane's hi / lo
  // MFMA-A fragments of block b sit at b * 2048 + {0, 1024} + lane * 16 bytes (the buffer is zeroed per call, so
  // rows / columns nobody publishes read as 0)
  const int nch = PREC == 0 ? (H + 15) >> 4 : (H + 31) >> 5;
  const size_t tile_f = (size_t)nch * (PREC == 0 ? 256 : 512);                                  // floats per h tile
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.hx, 0, (int)((size_t)2 * D * nbt * tile_f * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(pa.flags, 0, (int)((size_t)2 * D * nbt * nsl * 4), 0x00020000);
  // per-parity element / byte offsets of this group's flag row and hand-off tile (kept out of the time loop)
  const unsigned flg_el[2] = {(unsigned)(((0 * D + d) * nbt + bt) * nsl), (unsigned)(((1 * D + d) * nbt + bt) * nsl)};
  const unsigned tile_b[2] = {(unsigned)((((size_t)0 * D + d) * nbt + bt) * tile_f * 4), (unsigned)((((size_t)1 * D + d) * nbt + bt) * tile_f * 4)};

  // epilogue item of this thread: (row bl, unit jl) fixed for all steps -> c / h of the unit stay in a register
  const int bl = tid / HSU, jl = tid - bl * HSU;
  const int j = j0 + jl, b = b0 + bl;
  const bool item = bl < 16 && bl < Bc && j < H;
  const int ont = tanh_cell ? 0 : (jl >> 2), ojl = tanh_cell ? jl : (jl & 3);                 // tile / column group of the unit
  float state = 0.0f;   // c_{t-1} (LSTM) / h_{t-1} (GRU)
  // Wave roles.  The units' (row, unit) items occupy threads 0 .. 16*HSU-1; when that leaves the last wave without
  // items it becomes the COMMUNICATION wave: it alone polls the flags, fetches the h tile into LDS and publishes this
  // workgroup's block, so its in-order vmcnt never queues behind the item waves' operand prefetches and reserve
  // stores (which then have a whole step to complete).  Otherwise wave 0 does both jobs (correct, slower).
  const int cw = 16 * HSU <= 192 ? 3 : 0;
  // x-projection pre-activations of this item, always one step ahead: refilled right after the gate math consumed them
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  // running pointers of this item into the gate slab, the c / hn reserve and y: timestep t of the current step,
  // advanced by +-one timestep per step (no 64-bit index arithmetic in the loop)
  const long slab_g = (long)B * D * G * H, slab_h = (long)B * D * H;       // floats per timestep of gates / of aux, y
  const int bcl = min(b, B - 1), jcl = min(j, H - 1);
  const unsigned vg0 = (unsigned)(((bcl * D + d) * (G * H) + jcl) * 4);
  const unsigned vg1 = vg0 + (unsigned)(min(1, G - 1) * H * 4), vg2 = vg0 + (unsigned)(min(2, G - 1) * H * 4), vg3 = vg0 + (unsigned)(min(3, G - 1) * H * 4);
  const unsigned vh = (unsigned)(((bcl * D + d) * H + jcl) * 4);            // aux (T,B,D,H) and y (T,B,D*H) share it
  const __amdgpu_buffer_rsrc_t rg = whole_rsrc(p.gates, (size_t)T * slab_g), ra = whole_rsrc(p.aux, (size_t)T * slab_h), ry = whole_rsrc(p.y, (size_t)T * slab_h);
  const unsigned sg_b = (unsigned)(slab_g * 4), sh_b = (unsigned)(slab_h * 4);            // bytes per timestep slab
  if (item) {
    const unsigned o = (unsigned)(d == 0 ? 0 : T - 1) * sg_b;
    pre[0] = ld_slab(rg, vg0, o); pre[1] = ld_slab(rg, vg1, o); pre[2] = ld_slab(rg, vg2, o); pre[3] = ld_slab(rg, vg3, o);
  }
  __syncthreads();
#ifdef CTCN_PERSIST_STATS
  long long st_poll = 0, st_fill = 0, st_mm = 0, st_red = 0, st_epi = 0, st_t0 = clock64();
  long long st_e1 = 0, st_e2 = 0, st_e3 = 0, st_e4 = 0;   // gate math (to hpub barrier) | publish stores issued | drained | flag + reserve issue
  long long st_i0 = 0, st_i1 = 0, st_i2 = 0;             // item wave: partial sums | gate math + split + hpub writes
#endif

  for (int s = 0; s < T; ++s) {
#ifdef CTCN_PERSIST_STATS
    const long long c_a = clock64();
    long long c_p = c_a, c_f = c_a;
#endif
    const int t = d == 0 ? s : T - 1 - s;
    const size_t row_t = (size_t)t * B + b;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero;
    if (s > 0) {
      const int par = (s - 1) & 1;
      if (wave == cw) {
        const unsigned *fl = pa.flags + flg_el[par];
        const bool ok = poll_group(fl, nsl, (unsigned)s, lane, pa);
        if (!ok && lane == 0) {
          s_abort = 1;
          if (pa.status) atomicCAS(pa.status, 0, 101);
        }
      }
      lds_barrier();
      if (s_abort) break;
#ifdef CTCN_PERSIST_STATS
      c_p = clock64();
#endif
      const unsigned tbase = tile_b[par];
      if constexpr (PREC == 0) {
        f32x4 av[KQ4];
#pragma unroll
        for (int si = 0; si < KQ4; ++si) {
          const int k = kb + 16 * si, c = min(wave * KQ4 + si, nch - 1);
          const f32x4 v = ld_sc1_f4(rs, tbase + (unsigned)((c * 64 + lane) * 16));
          av[si] = (k < H && r < Bc) ? v : zero;
        }
#ifdef CTCN_PERSIST_STATS
        c_f = clock64();
#endif
#pragma unroll
        for (int si = 0; si < KQ4; ++si)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt][si][c], av[si][c], acc[nt], 0, 0, 0);
      } else {
        u32x4 ah[KB], al[KB];                     // plain vectors: a union here made hipcc wait after every second load
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const unsigned off = tbase + (unsigned)(min(wave * KB + i, nch - 1) * 2048 + lane * 16);   // blocks past the end: W is 0 there
          ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
          al[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 16);
        }
        __builtin_amdgcn_sched_barrier(0);        // all operand loads in flight before the first MFMA (hipcc otherwise reuses two register sets and serialises them)
#ifdef CTCN_PERSIST_STATS
        c_f = clock64();
#endif
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const bf16x8_t ahv = __builtin_bit_cast(bf16x8_t, ah[i]), alv = __builtin_bit_cast(bf16x8_t, al[i]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[nt][i], alv, acc[nt], 0, 0, 0);   // small terms first
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[nt][i], ahv, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[nt][i], ahv, acc[nt], 0, 0, 0);
          }
        }
      }
    }
#ifdef CTCN_PERSIST_STATS
    const long long c_b = clock64();
#endif
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)      // park [wave][tile][unit q][row]: one 16-B slot = the four gates (tanh cell: four units) of a row
      *reinterpret_cast<f32x4 *>(red + (((wave * NT + nt) * PK_T + q * PK_Q + r) << 2)) = acc[nt];
    lds_barrier();
#ifdef CTCN_PERSIST_STATS
    const long long c_c = clock64();
    st_poll += c_p - c_a; st_fill += c_f - c_p; st_mm += c_b - c_f; st_red += c_c - c_b;
#endif

    float sv0 = 0.f, sv1 = 0.f, sv2 = 0.f, sv3 = 0.f, sv4 = 0.f, hval = 0.f;     // values this item stores after the hand-off
#ifdef CTCN_PERSIST_STATS
    long long c_i0 = c_c;
#endif
    if (item) {
      // recurrent pre-activations of this item: gate g sits in column g*4 + ojl of tile ont (tanh cell: column jl)
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      if (tanh_cell) {                                                           // unit jl = register jl & 3 of slot (jl >> 2, row)
        const float *rp = red + (((jl >> 2) * PK_Q + bl) << 2) + (jl & 3);
#pragma unroll
        for (int w = 0; w < NW; ++w) o[0] += rp[(w * NT * PK_T) << 2];
      } else {                                                                   // one 16-B read per wave: the item's four gates
        const float *rp = red + ((ont * PK_T + ojl * PK_Q + bl) << 2);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(rp + ((w * NT * PK_T) << 2));
          o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
        }
      }
#ifdef CTCN_PERSIST_STATS
      if (o[0] + o[1] + o[2] + o[3] == 12345.678f) st_i2 += 1;                  // consume the sums before the stamp
      c_i0 = clock64();
#endif
      if (cell == CTCN_CELL_LSTM) {
        const float i_ = act_sigmoid(o[0] + pre[0]);
        const float f_ = act_sigmoid(o[1] + pre[1]);
        const float g_ = act_tanh(o[2] + pre[2]);
        const float o_ = act_sigmoid(o[3] + pre[3]);
        const float c = f_ * state + i_ * g_;
        hval = o_ * act_tanh(c);
        state = c;
        sv0 = i_; sv1 = f_; sv2 = g_; sv3 = o_; sv4 = c;
      } else if (cell == CTCN_CELL_GRU) {
        const float hn = o[2];
        const float r_ = act_sigmoid(o[0] + pre[0]);
        const float z_ = act_sigmoid(o[1] + pre[1]);
        const float n_ = act_tanh(pre[2] + r_ * hn);
        hval = (1.0f - z_) * n_ + z_ * state;
        state = hval;
        sv0 = r_; sv1 = z_; sv2 = n_; sv4 = hn;
      } else {
        hval = act_tanh(o[0] + pre[0]);
      }
      if constexpr (PREC == 0) {
        hpub[bl][jl] = hval;
      } else {
        unsigned short *hp = reinterpret_cast<unsigned short *>(&hpub[0][0]);
        const unsigned hi = f2bf(hval);
        hp[bl * 16 + jl] = (unsigned short)hi;
        hp[256 + bl * 16 + jl] = f2bf(hval - __uint_as_float(hi << 16));
      }
    }
#ifdef CTCN_PERSIST_STATS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long c_d0 = clock64();
    st_i0 += c_i0 - c_c; st_i1 += c_d0 - c_i0;
#endif
    lds_barrier();
#ifdef CTCN_PERSIST_STATS
    const long long c_d = clock64();
    long long c_e = c_d, c_g = c_d;
#endif
    // publish this workgroup's 16 x HSU block of h_t: 16-B stores by the communication wave, drained, then the flag
    if (s + 1 < T && wave == cw) {
      const int par = s & 1;
      const unsigned tbase = tile_b[par];
      // lanes run row-fastest so that consecutive lanes write consecutive 16-B (8-B) granules of the tile
      const int row = lane & 15, rest = lane >> 4;
      if constexpr (PREC == 1) {
        const unsigned short *hp = reinterpret_cast<const unsigned short *>(&hpub[0][0]);
        if ((HSU & 7) == 0) {                                      // one 16-B granule = 8 units of one plane
          const int plane = rest & 1, oct = rest >> 1, col = j0 + oct * 8;
          if (oct * 8 < HSU && row < Bc && col < H) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(hp + plane * 256 + row * 16 + oct * 8);
            const unsigned off = tbase + (unsigned)((col >> 5) * 2048 + plane * 1024 + (((col >> 3) & 3) * 16 + row) * 16);
            if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
          }
        } else {                                                   // pieces of 4 units (8 B)
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          for (int rr = rest; rr < 2 * (HSU / 4); rr += 4) {
            const int plane = rr & 1, pc = rr >> 1, col = j0 + pc * 4;
            if (row < Bc && col < H) {
              const u32x2 v = *reinterpret_cast<const u32x2 *>(hp + plane * 256 + row * 16 + pc * 4);
              const unsigned off = tbase + (unsigned)((col >> 5) * 2048 + plane * 1024 + (((col >> 3) & 3) * 16 + row) * 16 + ((col >> 2) & 1) * 8);
              if (local) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0);
              else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 16);
            }
          }
        }
      } else {                                                     // f32: one 16-B granule = 4 units (HSU % 4 == 0)
        const int c4 = rest, col = j0 + c4 * 4;
        if (c4 * 4 < HSU && row < Bc && col < H) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(&hpub[row][c4 * 4]);
          st_f4(rs, tbase + (unsigned)(((col >> 4) * 256 + (((col >> 2) & 3) * 16 + row) * 4) * 4), v, local);
        }
      }
#ifdef CTCN_PERSIST_STATS
      c_e = clock64();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the storing wave drains its stores (to L2 / to memory)
#ifdef CTCN_PERSIST_STATS
      c_g = clock64();
#endif
      if (lane == 0) st_u1(rf, (flg_el[par] + (unsigned)slice) * 4, (unsigned)(s + 1), local);
    }
    // The item waves' reserve stores (6 scattered dword stores per item) must not enter this CU's memory pipeline ahead of
    // the 16-B hand-off stores: queued behind them the "drain" above took 1 100-1 400 cycles instead of 300.  They are held
    // back (an LDS word, polled) until the communication wave has raised its flag.
    if (wave == cw) {
      if (lane == 0) __hip_atomic_store(&s_pub, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (item) {
      while (__hip_atomic_load(&s_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != s + 1) __builtin_amdgcn_s_sleep(1);
    }
    // off the critical path (item waves): the reserve (gates, c / hn) and y leave after the hand-off, and the next
    // step's pre-activations are requested a whole step before the gate math needs them
    {
      const unsigned og = (unsigned)t * sg_b, oh = (unsigned)t * sh_b;
      const unsigned on = (unsigned)(s + 1 < T ? (d == 0 ? t + 1 : t - 1) : t) * sg_b;      // past the end: re-read t (unused)
      if (item) {
        if (cell == CTCN_CELL_LSTM) {
          st_slab(rg, vg0, og, sv0); st_slab(rg, vg1, og, sv1); st_slab(rg, vg2, og, sv2); st_slab(rg, vg3, og, sv3);
          st_slab(ra, vh, oh, sv4);
        } else if (cell == CTCN_CELL_GRU) {
          st_slab(rg, vg0, og, sv0); st_slab(rg, vg1, og, sv1); st_slab(rg, vg2, og, sv2);
          st_slab(ra, vh, oh, sv4);
        }
        st_slab(ry, vh, oh, hval);
        pre[0] = ld_slab(rg, vg0, on); pre[1] = ld_slab(rg, vg1, on); pre[2] = ld_slab(rg, vg2, on); pre[3] = ld_slab(rg, vg3, on);
      }
    }
#ifdef CTCN_PERSIST_STATS
    const long long c_h = clock64();
    st_epi += c_h - c_c; st_e1 += c_d - c_c; st_e2 += c_e - c_d; st_e3 += c_g - c_e; st_e4 += c_h - c_g;
    if (tid == 0) st_e1 += 0;
    (void)c_d0;
#endif
  }
#ifdef CTCN_PERSIST_STATS
  if (pa.stats && slice == 7 && d == 0 && bt == 0 && tid == cw * 64) {
    pa.stats[0] = st_poll; pa.stats[1] = st_fill; pa.stats[2] = st_mm; pa.stats[3] = st_red; pa.stats[4] = st_epi; pa.stats[5] = clock64() - st_t0;
    pa.stats[6] = st_e1; pa.stats[7] = st_e2; pa.stats[8] = st_e3; pa.stats[9] = st_e4;
  }
  if (pa.stats && slice == 7 && d == 0 && bt == 0 && tid == 0) { pa.stats[10] = st_i0; pa.stats[11] = st_i1; pa.stats[12] = st_i2; }
  if (pa.stats && d == 0 && bt == 0 && tid == cw * 64 && slice < 64) {      // per-slice poll / rest split + the CU it ran on
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    pa.stats[16 + slice * 3] = st_poll; pa.stats[17 + slice * 3] = st_fill + st_mm + st_red + st_epi; pa.stats[18 + slice * 3] = hw;
  }
#endif
  // a hand-off timed out: nothing this launch produced can be trusted.  Every (row, unit) item poisons its WHOLE output
  // column with NaN (all T frames -- a single poisoned frame can sit in the padding of every shorter utterance and be masked
  // out of the CTC loss), so the loss of this step is NaN for every model configuration, besides the sticky status word.
  if (s_abort && item)
    for (int tt = 0; tt < T; ++tt) st_slab(ry, vh, (unsigned)tt * sh_b, __uint_as_float(0x7fc00000u));
}

// Launch a persistent kernel only if every working workgroup is co-resident (occupancy query x CU count, with one
// block of slack per CU when the API admits more than one: ROCm 7.2 can over-report by one, MI355X_MICROARCH.md).
// Device-scope mode: the whole 3-D grid against all CUs.  XCD-local mode (a.local): a.nx * wpx workgroups, wpx of
// them working on each XCD (the rest exit at once), against the CUs of ONE XCD.
template <class Kern>
bool launch_resident(Kern kern, dim3 grid, int threads, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  int per_cu = 0;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu <= 0) return false;
  const int cus = a.local ? ctcn_device_cus() / a.nx : ctcn_device_cus();
  // device scope, one workgroup per CU: keep 8 CUs of slack; XCD-local: a group may take every CU of its XCD (nothing else
  // of this stream runs beside a persistent launch, and a role that cannot become resident ends in the bounded-spin abort)
  const long cap = (long)cus * (per_cu > 1 ? per_cu - 1 : 1) - (per_cu > 1 || a.local ? 0 : 8);
  const long need = a.local ? wpx : (long)grid.x * grid.y * grid.z;
  if (need > cap) return false;
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, a);
  return true;
}

template <int NT, int PREC, int CELL>
bool launch_fwd_persist_nt(int kq4, dim3 grid, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  switch (kq4) {
    case 1: return launch_resident(rnn_fwd_persist<NT, 1, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 2: return launch_resident(rnn_fwd_persist<NT, 2, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 3: return launch_resident(rnn_fwd_persist<NT, 3, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 4: return launch_resident(rnn_fwd_persist<NT, 4, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 5: return launch_resident(rnn_fwd_persist<NT, 5, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 6: return launch_resident(rnn_fwd_persist<NT, 6, PREC, CELL>, grid, 256, lds, st, a, wpx);
    case 8: return launch_resident(rnn_fwd_persist<NT, 8, PREC, CELL>, grid, 256, lds, st, a, wpx);
    default: return false;
  }
}
template <int PREC, int CELL>
bool launch_fwd_persist_p(int nt, int kq4, dim3 grid, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  switch (nt) {
    case 1: return launch_fwd_persist_nt<1, PREC, CELL>(kq4, grid, lds, st, a, wpx);
    case 2: return launch_fwd_persist_nt<2, PREC, CELL>(kq4, grid, lds, st, a, wpx);
    case 3: return launch_fwd_persist_nt<3, PREC, CELL>(kq4, grid, lds, st, a, wpx);
    case 4: return launch_fwd_persist_nt<4, PREC, CELL>(kq4, grid, lds, st, a, wpx);
    default: return false;
  }
}
bool launch_fwd_persist(int prec, int nt, int kq4, dim3 grid, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  if (prec && a.a.cell == CTCN_CELL_LSTM) return launch_fwd_persist_p<1, CTCN_CELL_LSTM>(nt, kq4, grid, lds, st, a, wpx);
  if (prec && a.a.cell == CTCN_CELL_GRU) return launch_fwd_persist_p<1, CTCN_CELL_GRU>(nt, kq4, grid, lds, st, a, wpx);
  return prec ? launch_fwd_persist_p<1, -1>(nt, kq4, grid, lds, st, a, wpx) : launch_fwd_persist_p<0, -1>(nt, kq4, grid, lds, st, a, wpx);
}

// ================================================================================================
// Persistent forward recurrence, TAGGED GATHER (round 2; precision 1, LSTM / GRU, H % 32 == 0, XCD-local placement).
//
// rnn_fwd_persist hands h_t over with data + flag: the publishing wave drains its stores, raises a flag, the consumers poll the
// flags, pass a barrier and only then fetch the tile -- three L2 trips in series per timestep (store acknowledge, flag
// store -> poll, operand load).  Here the structure of rnn_bwd_scatter is applied to the forward product:
//   * a workgroup owns 16 hidden units (4 gates each) of one 16-row batch tile; 1024 threads.  Waves 0..3 hold the 256 (row, unit)
//     items -- gate math, ALL reserve traffic, and the publish: each item stores ONE dword {bf16 hi | bf16 lo << 16} of its h_t
//     straight into the group's tile, with the step tag in the LSB of lo (bit 16).  No drain, no flag;
//   * waves 4..15 are exchange waves: wave 4 + g owns the 32-k blocks g, g + 12 of the contraction.  It re-reads its 2-KB block
//     (two coalesced 1-KB loads in MFMA-B lane order) until all 8 dwords of every lane carry the tag of the step -- the polled
//     registers ARE the operand: two v_perm per dword pair split them into the hi / lo fragments, 12 v_mfma_f32_16x16x32_bf16
//     per block multiply them by the wave's resident W_hh fragments (4 tiles = 16 units x 4 gates, C = W h^T), the four partial
//     tiles are parked in LDS;
//   * ONE barrier per step (parked partials -> items).  The items' publish needs none: the exchange waves learn of it through L2.
// A torn or stale block (any granularity down to a dword) is simply read again; the parity buffers start zeroed and the tag
// alternates between the two uses of a buffer, as in rnn_bwd_scatter.  The tag costs the LSB of lo: h is carried to 2^-16
// relative instead of 2^-17 (the saved y stays the exact float32 value).  Layout of the tile: [32-k block][half][octet q][row][4
// dwords]: unit u of a block sits at octet u >> 3, half (u >> 2) & 1, dword u & 3, so each of a lane's two 16-B loads is one
// fully coalesced 1-KB wave load.
// Measured (tools/mb_step.hip, cfg2 layer, us per step; the flag kernel = 2.00 in the same run): THIS, one poll in flight, first poll
// 8 x 64 cycles after the barrier: 1.60-1.65 (a poll round trip is ~600 cycles, 1.4 polls per step; no delay 1.66-1.74, >= 12 sleeps
// 1.78-1.89) | two polls in flight 1.76 | sentinel spin (one dword per producing store instruction, then one full fetch) 1.86 | a flag
// wave (item waves raise undrained words behind their publish, one idle wave polls them and releases the exchange waves through LDS,
// tags stay the proof of arrival): 2.14-2.23 -- the words show up ~2 000 cycles after the publish although every block is then valid
// on its first fetch | first poll timed from the workgroup's OWN publish (LDS counter) + 2..12 sleeps: every block valid on the first
// fetch, but polls that coincide with the publishes of the whole XCD take 1 000-1 350 cycles instead of ~600: 1.71-1.93.
// The reserve traffic of the item waves is paced by accident: inside the wave-role branch hipcc keeps the (uniform) timestep offsets in
// VGPRs and wraps each of the ten buffer instructions in a waterfall loop (~890 cycles of issue per step).  With scalar offsets
// (readfirstlane) the ten instructions of 80 item waves per XCD reach L2 together with the publishes: 1.83; scalar offsets + an
// explicit pause of 16-32 x 64 cycles before them: poll round trip 300-430 cycles instead of 600, but 1.64-1.69 overall (the step is
// set by the slowest of the ten exchange waves, a ~1 000-cycle tail); the same + pre-activations fetched two steps ahead into
// alternating register sets (as rnn_bwd_scatter does): 1.76-1.93.  Left as it is.
// ================================================================================================
// RSV (round 3, option "fwd_rsv_lds"): the item waves' reserve traffic -- 6-7 scattered dword stores and 3-4 dword loads per item and step, ~28
// + 16 wave instructions per workgroup issued through waterfall loops -- goes through LDS instead: the items park their saved activations,
// c / hn, h and the dropped h as float32 [array][row][unit]; after the next barrier, in the pause before its first poll, exchange wave a
// stores array a of the PREVIOUS step with ONE 16-B store per lane (64 B contiguous per row); item wave g brings gate g's pre-activations of
// step s + 2 into a 3-deep LDS ring with ONE global_load_lds_dwordx4 (counted vmcnt wait before the barrier, as rnn_bwd_scatter2).
// (round 6) The parked tiles are DOUBLE BUFFERED by step parity.  With one barrier per step the exchange waves enter step s + 1 while the item waves
// still read the tiles of step s, and nothing but time kept a wave that found its blocks of h_s already published (by workgroups ahead of this one)
// from parking step s + 1 over them: ~1 000 cycles of poll pause, round trip and MFMAs against ~300 of reading -- unless the item waves stall.  At a
// direction's FIRST step they do (their code is not in the instruction cache yet): the cfg4 divergence of round 6 -- in ~1 % of the training steps
// some item waves of one or a few bottom-layer workgroups added partial products of step 1 to their pre-activations of step 0 (the saved W_hn h of the
// GRU, exactly 0 at that step, was not; profiles/r06_divergence_root_cause.txt), depending on where the code object lay in memory (per process) and on
// the build.  With two buffers the tiles of step s are next written in step s + 2, behind barrier s + 1, which an item wave passes only after its reads.
// SLOW: the parity harness's instantiation (option "rnn_slow_items"): the item waves sleep pa.slow x 64 cycles before they read the tiles, every step --
// results must not change (tests/test_gpu_kernels.py:test_rnn_fwd_tagged_with_slow_item_waves).
template <int NBW, int CELL, bool RSV = false, bool SLOW = false>
__global__ __launch_bounds__(1024) void rnn_fwd_tagged(PersistArgs pa) {
  constexpr int NGW = 12, NMT = 4;                           // exchange waves, MFMA tiles (4 units x 4 gates each) per workgroup
  constexpr int PQ = 17;                                     // parked slots per (tile, unit) row: 16 batch rows + 1 (bank spread)
  constexpr int RED_N = NGW * NMT * 4 * PQ * 4;              // floats of one set of parked tiles
  const RnnArgs &p = pa.a;
  __shared__ __attribute__((aligned(16))) float red2[2 * RED_N];
  __shared__ uint4 dropw[4][64];                   // fused dropout: the Philox groups of an item wave's next four steps
  __shared__ __attribute__((aligned(16))) float outq[RSV ? 2 : 1][RSV ? 7 : 1][RSV ? 256 : 4];   // RSV: values to store, by step parity
  __shared__ __attribute__((aligned(16))) float preq[RSV ? 3 : 1][RSV ? 4 : 1][RSV ? 256 : 4];   // RSV: pre-activations of three steps
  __shared__ int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  constexpr int G = CELL == CTCN_CELL_LSTM ? 4 : 3;
  const int H = p.H, D = p.D, B = p.B, T = p.T;
  const PersistRole role = persist_role(pa, D, &s_ticket);
  if (!role.active) return;
  const int d = role.d, bt = role.bt, nbt = pa.nbt, slice = role.slice;
  const int b0 = bt * 16, j0 = slice * 16;
  const int Bc = min(16, B - b0);
  const float *W = d == 0 ? p.w0 : p.w1;
  const int nblk = H >> 5;
  const int gw = wave - 4;

  // exchange waves: W_hh fragments (first MFMA operand) of this lane for its blocks: tile mt, row r = (unit mt*4 + (r >> 2), gate r & 3)
  bf16x8_t whi[NBW][NMT], wlo[NBW][NMT];
#pragma unroll
  for (int bw = 0; bw < NBW; ++bw) {
    // block of slot (wave, bw), rotated by the slice number: the workgroups of a group walk the tile in different orders
    const int slot = gw + NGW * bw, blk = slot < nblk ? (slot + slice) % nblk : nblk;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const int gate = r & 3, jj = mt * 4 + (r >> 2);
      const bool wv = wave >= 4 && blk < nblk && gate < G && (j0 + jj) < H;
      load_w8(wv ? W + (size_t)(gate * H + j0 + jj) * H : nullptr, 32 * blk + 8 * q, H, whi[bw][mt], wlo[bw][mt]);
    }
  }
  const size_t tile_b1 = (size_t)nblk * 2048;                                              // bytes per h tile
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.hx, 0, (int)((size_t)2 * D * nbt * tile_b1), 0x00020000);
  const unsigned tile_b[2] = {(unsigned)((((size_t)0 * D + d) * nbt + bt) * tile_b1), (unsigned)((((size_t)1 * D + d) * nbt + bt) * tile_b1)};

  // item of this thread (waves 0..3): row bl, unit jl of the slice; c / h of the unit stay in a register
  const int bl = (tid >> 4) & 15, jl = tid & 15, j = j0 + jl, b = b0 + bl;
  const bool item = tid < 256 && bl < Bc && j < H;
  float state = 0.0f;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  const long slab_g = (long)B * D * G * H, slab_h = (long)B * D * H;
  const int bcl = min(b, B - 1), jcl = min(j, H - 1);
  const unsigned vg0 = (unsigned)(((bcl * D + d) * (G * H) + jcl) * 4);
  const unsigned vg1 = vg0 + (unsigned)(H * 4), vg2 = vg0 + (unsigned)(2 * H * 4), vg3 = vg0 + (unsigned)(min(3, G - 1) * H * 4);
  const unsigned vh = (unsigned)(((bcl * D + d) * H + jcl) * 4);
  const __amdgpu_buffer_rsrc_t rg = whole_rsrc(p.gates, (size_t)T * slab_g), ra = whole_rsrc(p.aux, (size_t)T * slab_h), ry = whole_rsrc(p.y, (size_t)T * slab_h);
  const __amdgpu_buffer_rsrc_t ryd = whole_rsrc(pa.ydrop ? pa.ydrop : p.y, (size_t)T * slab_h);
  const unsigned sg_b = (unsigned)(slab_g * 4), sh_b = (unsigned)(slab_h * 4);
  // the item's dword in the published tile: block j >> 5, unit u = j & 31 -> [half (u>>2)&1][octet u>>3][row][dword u&3]
  const unsigned pub_off = (unsigned)((j >> 5) * 2048 + (((((j >> 2) & 1) * 4 + ((j & 31) >> 3)) * 16 + bl) * 4 + (j & 3)) * 4);
  // RSV: lane = (row lane >> 2, unit quad lane & 3) of the 16 x 16 item tile; clamped (valid) addresses for rows / quads outside it
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int xrow = min(b0 + (lane >> 2), B - 1), xj = min(j0 + 4 * (lane & 3), H - 4);
  const size_t xg = ((size_t)xrow * D + d) * (size_t)(G * H) + xj, xh = ((size_t)xrow * D + d) * H + xj;
  const bool xvalid = (lane >> 2) < Bc && j0 + 4 * (lane & 3) < H;
  auto pre_dma = [&](int step, int set) {          // item wave g: gate g's pre-activations of `step` (clamped) -> preq[set][g]
    const int sc = min(step, T - 1), ts = d == 0 ? sc : T - 1 - sc;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.gates + (size_t)ts * slab_g + xg + (size_t)wave * H), (lptr_t)&preq[set][RSV ? wave : 0][0], 16, 0, 0);
  };
  // (round 6) the store goes through the tensors' loop-invariant buffer resources (SGPRs), a per-lane byte offset that never changes and the timestep as
  // the scalar offset: with 64-bit pointers hipcc kept the per-array base addresses in a table in memory and in scratch, and every exchange wave that
  // stores an array fetched them -- two dependent memory round trips -- in the pause in front of its first poll
  // (array a on exchange wave a.  Measured and not kept: array a on wave 11 - a, so that the waves that own TWO blocks of the contraction at H = 512 -- waves
  // 0 .. 3 -- keep their pause free: cfg4 forward 2.22 -> 2.36 us per step; their later first poll is the better one)
  const int sa = gw;
  const unsigned sv_off = (unsigned)((sa < 4 ? xg + (size_t)(sa < 0 ? 0 : sa) * H : xh) * 4);
  auto service_store = [&](int step) {             // exchange wave a: array a of `step`, one 16-B store per lane
    const int ts = d == 0 ? step : T - 1 - step, a = sa;
    if (a < 7 && xvalid && (a < G || a == 4 || a == 5 || (a == 6 && pa.ydrop))) {
      const u32x4 v = *reinterpret_cast<const u32x4 *>(&outq[RSV ? step & 1 : 0][RSV ? a : 0][RSV ? lane * 4 : 0]);
      const unsigned og_ = __builtin_amdgcn_readfirstlane((unsigned)ts * sg_b), oh_ = __builtin_amdgcn_readfirstlane((unsigned)ts * sh_b);   // uniform: SGPR operands, no waterfall
      if (a < 4) __builtin_amdgcn_raw_buffer_store_b128(v, rg, sv_off, og_, 0);
      else if (a == 4) __builtin_amdgcn_raw_buffer_store_b128(v, ra, sv_off, oh_, 0);
      else if (a == 5) __builtin_amdgcn_raw_buffer_store_b128(v, ry, sv_off, oh_, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, ryd, sv_off, oh_, 0);
    }
  };
  if constexpr (RSV) {
    if (wave < G) { pre_dma(0, 0); pre_dma(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (item) {
    const unsigned o = (unsigned)(d == 0 ? 0 : T - 1) * sg_b;
    pre[0] = ld_slab(rg, vg0, o); pre[1] = ld_slab(rg, vg1, o); pre[2] = ld_slab(rg, vg2, o);
    if (G == 4) pre[3] = ld_slab(rg, vg3, o);
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  int have_chunks = 0;                                         // chunk pairs known to be complete (pair 0 is computed before the launch)
  int pset = 0;                                                // RSV: ring set of the current step (s % 3)
  // (round 4, measured and not kept: issue priorities -- s_setprio 2 for the exchange waves, and / or 3 for the item waves from the barrier
  // to their publish and 0 for their reserve traffic, here and in rnn_bwd_scatter: 13.36 ms per cfg2 step without, 13.38-13.43 with; the
  // phases of the two wave kinds barely overlap, so there is nothing to arbitrate)
  __syncthreads();
#ifdef CTCN_PERSIST_STATS
  long long zx[4] = {0, 0, 0, 0}, zi[6] = {0, 0, 0, 0, 0, 0}, zt0 = clock64(), z_prev = zt0, zq = 0;
#endif

  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
#ifndef CTCN_RED_SINGLE
    float *const red = red2 + (s & 1) * RED_N;                // the parked tiles of this step
#else
    float *const red = red2;                                  // (tools: the single buffer of rounds 2-5, to show that the SLOW test catches it)
#endif
#ifdef CTCN_PERSIST_STATS
    const long long z_a = clock64();
    long long z_p = z_a, z_m = z_a;
#endif
    if (wave >= 4) {
      if constexpr (SLOW) { for (int i = 0; i < pa.slow_x; ++i) __builtin_amdgcn_s_sleep(1); }
      f32x4 acc[NMT];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) acc[mt] = zero;
      // in the pause before the first poll: the reserve values of step s - 2 (its item phase ended before the barrier this wave has just
      // passed; the item phase of step s - 1 runs NOW and fills the other parity)
      if constexpr (RSV) { if (s > 1) service_store(s - 2); }
      if (s > 0) {
        const int par = (s - 1) & 1;
        const unsigned tb = (((((unsigned)(s - 1)) >> 1) & 1u) ^ 1u) << 16;
        // nothing can arrive before the items have done their sums and gate math and the stores have crossed L2: polls issued
        // earlier only load the L2 channels the publishes are about to need
        for (int i = 0; i < pa.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int bw = 0; bw < NBW; ++bw) {
          const int slot = gw + NGW * bw, blk = slot < nblk ? (slot + slice) % nblk : nblk;
          if (blk < nblk) {
            const unsigned boff = tile_b[par] + (unsigned)(blk * 2048 + lane * 16);
            u32x4 v0, v1;
            auto valid = [&](const u32x4 &a0, const u32x4 &a1) {
              const unsigned all1 = a0.x & a0.y & a0.z & a0.w & a1.x & a1.y & a1.z & a1.w;
              const unsigned any1 = a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w;
              const bool okl = ((all1 & 0x10000u) == tb) && ((any1 & 0x10000u) == tb);
              return __builtin_amdgcn_ballot_w64(okl) == ~0ull;
            };
            for (int spins = 0;; ++spins) {
#ifdef CTCN_PERSIST_STATS
              const long long z_q0 = clock64();
#endif
              v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 16);
              v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, boff + 1024, 0, 16);
#ifdef CTCN_PERSIST_STATS
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
              zx[3] += clock64() - z_q0; zq += 1;
#endif
              if (valid(v0, v1)) break;
              if (spins > pa.spin_limit || ((spins & 63) == 63 && pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (pa.status) atomicCAS(pa.status, 0, 111);             // give up: the launch finishes with a poisoned output
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
#ifdef CTCN_PERSIST_STATS
            z_p = clock64();
#endif
            // split the {hi | lo << 16} dwords into the two MFMA fragments (8 consecutive k per lane), tag bits cleared
            Bf16Pack fh, fl;
            fh.u = (u32x4){__builtin_amdgcn_perm(v0.y, v0.x, 0x05040100u), __builtin_amdgcn_perm(v0.w, v0.z, 0x05040100u),
                           __builtin_amdgcn_perm(v1.y, v1.x, 0x05040100u), __builtin_amdgcn_perm(v1.w, v1.z, 0x05040100u)};
            fl.u = (u32x4){__builtin_amdgcn_perm(v0.y, v0.x, 0x07060302u) & 0xFFFEFFFEu, __builtin_amdgcn_perm(v0.w, v0.z, 0x07060302u) & 0xFFFEFFFEu,
                           __builtin_amdgcn_perm(v1.y, v1.x, 0x07060302u) & 0xFFFEFFFEu, __builtin_amdgcn_perm(v1.w, v1.z, 0x07060302u) & 0xFFFEFFFEu};
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[bw][mt], fl.v, acc[mt], 0, 0, 0);   // small terms first
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[bw][mt], fh.v, acc[mt], 0, 0, 0);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[bw][mt], fh.v, acc[mt], 0, 0, 0);
            }
          }
        }
      }
      // park: slot (wave, tile, unit q, row r) = the four gates of unit mt*4 + q for batch row r
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) *reinterpret_cast<f32x4 *>(red + ((((gw * NMT + mt) * 4 + q) * PQ + r) << 2)) = acc[mt];
#ifdef CTCN_PERSIST_STATS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      z_m = clock64();
#endif
    }
    if constexpr (RSV) { if (wave < G) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }   // this wave's pre-activation DMA of step s has landed (the youngest, s + 1, may be in flight)
    lds_barrier();
#ifdef CTCN_PERSIST_STATS
    const long long z_b = clock64();
    if (wave >= 4) { zx[0] += z_p - z_a; zx[1] += z_m - z_p; zx[2] += z_b - z_m; }
    long long z_i1 = z_b, z_i2 = z_b, z_i3 = z_b;
#endif
    if (wave < 4) {
      if constexpr (SLOW) { for (int i = 0; i < pa.slow; ++i) __builtin_amdgcn_s_sleep(1); }
      if constexpr (RSV) {
        pre[0] = preq[pset][0][tid]; pre[1] = preq[pset][1][tid]; pre[2] = preq[pset][2][tid];
        if constexpr (G == 4) pre[3] = preq[pset][RSV ? 3 : 0][tid];
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      const float *rp = red + ((((jl >> 2) * 4 + (jl & 3)) * PQ + bl) << 2);
#pragma unroll
      for (int w = 0; w < NGW; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(rp + ((w * NMT * 4 * PQ) << 2));
        o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
      }
#ifdef CTCN_PERSIST_STATS
      if (o[0] + o[1] + o[2] + o[3] == 12345.678f) zi[5] += 1;
      z_i1 = clock64();
#endif
      float sv0 = 0.f, sv1 = 0.f, sv2 = 0.f, sv3 = 0.f, sv4 = 0.f, hval = 0.f;
      if (item) {
        if constexpr (CELL == CTCN_CELL_LSTM) {
          const float i_ = act_sigmoid(o[0] + pre[0]);
          const float f_ = act_sigmoid(o[1] + pre[1]);
          const float g_ = act_tanh(o[2] + pre[2]);
          const float o_ = act_sigmoid(o[3] + pre[3]);
          const float c = f_ * state + i_ * g_;
          hval = o_ * act_tanh(c);
          state = c;
          sv0 = i_; sv1 = f_; sv2 = g_; sv3 = o_; sv4 = c;
        } else {
          const float hn = o[2];
          const float r_ = act_sigmoid(o[0] + pre[0]);
          const float z_ = act_sigmoid(o[1] + pre[1]);
          const float n_ = act_tanh(pre[2] + r_ * hn);
          hval = (1.0f - z_) * n_ + z_ * state;
          state = hval;
          sv0 = r_; sv1 = z_; sv2 = n_; sv4 = hn;
        }
      }
#ifdef CTCN_PERSIST_STATS
      if (hval == 12345.678f) zi[5] += 1;
      z_i2 = clock64();
#endif
      if (s + 1 < T) {             // publish (every lane of the item waves: rows / units nobody owns publish a tagged 0)
        const unsigned hi = f2bf(hval);
        const unsigned lo = f2bf(hval - __uint_as_float(hi << 16));
        const unsigned tbit = ((((unsigned)s) >> 1) & 1u) ^ 1u;
        const unsigned word = hi | (((lo & ~1u) | tbit) << 16);
        // (16-B publishes -- the four dwords of a lane quad gathered with DPP broadcasts, one store from 16 lanes per wave, 64 write requests
        // per workgroup instead of 256 -- measured in round 3: 1.66-1.67 vs 1.60-1.64 us per step: not kept)
        __builtin_amdgcn_raw_buffer_store_b32(word, rs, tile_b[s & 1] + pub_off, 0, 0);   // write-back into this XCD's L2
      }
#ifdef CTCN_PERSIST_STATS
      z_i3 = clock64();
#endif
      // reserve traffic, behind the publish in this wave's queue: saved activations, c / hn, y out; next step's pre-activations in
      const unsigned og = (unsigned)t * sg_b, oh = (unsigned)t * sh_b;
      const unsigned on = (unsigned)(s + 1 < T ? (d == 0 ? t + 1 : t - 1) : t) * sg_b;
      if (pa.chunk_T > 0 && s + (RSV ? 2 : 1) < T) {
        // pipelined input projection: the pre-activations of the time chunk the next step falls into (RSV: the step after it, fetched now)
        // are written by a GEMM on the side stream (other XCDs) while this kernel runs; chunk pair p is complete once the counter shows p
        const int tn = d == 0 ? t + (RSV ? 2 : 1) : t - (RSV ? 2 : 1), c = tn / pa.chunk_T, need = min(c, pa.nchunk - 1 - c);
        if (need > have_chunks) {
          for (int spins = 0;; ++spins) {
            have_chunks = (int)__hip_atomic_load(pa.chunk_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (have_chunks >= need) break;
            if (spins > pa.spin_limit || ((spins & 63) == 63 && pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              if (pa.status) atomicCAS(pa.status, 0, 112);
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
        }
      }
      if constexpr (RSV) {
        if (item) {
          float *oq = &outq[s & 1][0][tid];
          oq[0] = sv0; oq[256] = sv1; oq[512] = sv2;
          if constexpr (CELL == CTCN_CELL_LSTM) oq[768] = sv3;
          oq[1024] = sv4; oq[1280] = hval;
        }
        if (wave < G) pre_dma(s + 2, pset == 0 ? 2 : pset - 1);        // ring set (s + 2) % 3
      } else if (item) {
        st_slab(rg, vg0, og, sv0); st_slab(rg, vg1, og, sv1); st_slab(rg, vg2, og, sv2);
        if constexpr (CELL == CTCN_CELL_LSTM) st_slab(rg, vg3, og, sv3);
        st_slab(ra, vh, oh, sv4);
        st_slab(ry, vh, oh, hval);
        pre[0] = ld_slab(rg, vg0, on); pre[1] = ld_slab(rg, vg1, on); pre[2] = ld_slab(rg, vg2, on);
        if constexpr (CELL == CTCN_CELL_LSTM) pre[3] = ld_slab(rg, vg3, on);
      }
      if (pa.ydrop && tid < 256) {
        // the layer's dropout, fused (ctcn_rnn_fwd_dropout): the keep bit of element i of y is word (i & 3) of Philox group offset + (i >> 2)
        // -- exactly what dropout_kernel computes.  An item wave needs 16 groups per step (4 rows x 4 unit quads), so every fourth step
        // its 64 lanes compute the 64 groups of the next four steps (one Philox4x32-10 per lane, ~900 wave cycles) and park them in
        // LDS; per step a lane then reads its word (a per-step Philox in every lane cost the recurrence as much as the pass it saved).
        // The dropped value goes into an HBM that is idle during the recurrence instead of a pass over y after it.
        if ((s & 3) == 0) {
          const int sk = min(s + (lane >> 4), T - 1), tk = d == 0 ? sk : T - 1 - sk;
          const int br = min(b0 + 4 * (tid >> 6) + ((lane >> 2) & 3), B - 1), j4 = min(j0 + 4 * (lane & 3), H - 4);
          const unsigned idxk = (unsigned)((((size_t)tk * B + br) * D + d) * H + j4);
          uint32_t rr[4];
          philox4(pa.drop_seed, pa.drop_off + (idxk >> 2), rr);
          dropw[tid >> 6][lane] = make_uint4(rr[0], rr[1], rr[2], rr[3]);
        }
        const uint32_t w = reinterpret_cast<const uint32_t *>(&dropw[tid >> 6][(s & 3) * 16 + (lane >> 4) * 4 + ((lane & 15) >> 2)])[lane & 3];
        if constexpr (RSV) { if (item) outq[s & 1][RSV ? 6 : 0][tid] = ((w >> 8) * (1.0f / 16777216.0f) >= pa.drop_p) ? hval * pa.drop_scale : 0.0f; }
        else if (item) st_slab(ryd, vh, oh, ((w >> 8) * (1.0f / 16777216.0f) >= pa.drop_p) ? hval * pa.drop_scale : 0.0f);
      }
      if constexpr (RSV) pset = pset == 2 ? 0 : pset + 1;
#ifdef CTCN_PERSIST_STATS
      { const long long z_e = clock64(); zi[0] += z_b - z_prev; zi[1] += z_i1 - z_b; zi[2] += z_i2 - z_i1; zi[3] += z_i3 - z_i2; zi[4] += z_e - z_i3; z_prev = z_e; }
#endif
    }
  }
  if constexpr (RSV) {
    lds_barrier();
    if (wave >= 4) { if (T > 1) service_store(T - 2); service_store(T - 1); }
  }
#ifdef CTCN_PERSIST_STATS
  if (pa.stats && slice == 3 && d == 0 && bt == 0 && tid == 4 * 64) { pa.stats[0] = zx[0]; pa.stats[1] = zx[1]; pa.stats[2] = zx[2]; pa.stats[3] = clock64() - zt0; pa.stats[4] = zx[3]; pa.stats[5] = zq; }
  if (pa.stats && slice == 3 && d == 0 && bt == 0 && tid == 0) for (int i = 0; i < 5; ++i) pa.stats[8 + i] = zi[i];
#endif
  const bool bad = pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (bad && item)             // a hand-off timed out: poison the whole output column (see rnn_fwd_persist)
    for (int tt = 0; tt < T; ++tt) {
      st_slab(ry, vh, (unsigned)tt * sh_b, __uint_as_float(0x7fc00000u));
      if (pa.ydrop) st_slab(ryd, vh, (unsigned)tt * sh_b, __uint_as_float(0x7fc00000u));
    }
}

template <int CELL, bool SLOW>
bool launch_fwd_tagged_cs(int nbw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  if (a.poll_depth & 256) {       // option "fwd_rsv_lds": reserve traffic through LDS (16-B aligned reserves)
    switch (nbw) {
      case 1: return launch_resident(rnn_fwd_tagged<1, CELL, true, SLOW>, grid, 1024, 0, st, a, wpx);
      case 2: return launch_resident(rnn_fwd_tagged<2, CELL, true, SLOW>, grid, 1024, 0, st, a, wpx);
      default: return false;
    }
  }
  switch (nbw) {
    case 1: return launch_resident(rnn_fwd_tagged<1, CELL, false, SLOW>, grid, 1024, 0, st, a, wpx);
    case 2: return launch_resident(rnn_fwd_tagged<2, CELL, false, SLOW>, grid, 1024, 0, st, a, wpx);
    default: return false;
  }
}
template <int CELL>
bool launch_fwd_tagged_c(int nbw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  return (a.slow > 0 || a.slow_x > 0) ? launch_fwd_tagged_cs<CELL, true>(nbw, grid, st, a, wpx) : launch_fwd_tagged_cs<CELL, false>(nbw, grid, st, a, wpx);
}
bool launch_fwd_tagged(int nbw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  if (a.a.cell == CTCN_CELL_LSTM) return launch_fwd_tagged_c<CTCN_CELL_LSTM>(nbw, grid, st, a, wpx);
  if (a.a.cell == CTCN_CELL_GRU) return launch_fwd_tagged_c<CTCN_CELL_GRU>(nbw, grid, st, a, wpx);
  return false;
}

// ================================================================================================
// Persistent backward recurrence: ONE launch per layer, same hand-off scheme as rnn_fwd_persist.
// A workgroup (1024 threads = 16 waves) owns one 16(batch) x 16(hidden) tile of dh_rec = d(pre-act)_{next} * W_hh:
// its 16 rows of W_hh^T (K = G*H floats each) stay in VGPRs (K split over 16 waves x 4 k-lanes), dc / dh*z of its 256
// (row, unit) items stay in one register per thread, and per step it pulls only the 16 x K tile of the other
// workgroups' d(pre-activation) (80 KB at H=320), loaded straight into the MFMA operand registers.  Waves 0..3 hold the
// items and publish (LDS-staged, 16-B granules); wave 15 polls.  grid as rnn_fwd_persist with 16-unit slices.
// ================================================================================================
template <int KQ4, int PREC>
__global__ __launch_bounds__(1024) void rnn_bwd_persist(PersistArgs pa) {
  constexpr int NW = 16;
  const RnnArgs &p = pa.a;
  __shared__ __attribute__((aligned(16))) float red[NW * RT_T];     // parked partial tiles
  __shared__ __attribute__((aligned(16))) float stage[1024];        // publish stage: 16 x 16 x G block in the granule order of the tile
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int H = p.H, G = p.G, D = p.D, B = p.B, T = p.T;
  __shared__ int s_ticket;
  const PersistRole role = persist_role(pa, D, &s_ticket);
  if (!role.active) return;
  const int d = role.d, bt = role.bt, nbt = pa.nbt, nsl = pa.nsl, slice = role.slice, local = pa.local;
  const int b0 = bt * 16, j0 = slice * 16;
  const int Bc = min(16, B - b0);
  const int K = G * H;
  const float *WT = d == 0 ? p.w0 : p.w1;
  const bool bvalid = (j0 + r) < H;
  const float *brow = bvalid ? WT + (size_t)(j0 + r) * K : WT;
  const int kb = wave * 16 * KQ4 + q * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (tid == 0) s_abort = 0;

  constexpr int KB = (KQ4 + 1) / 2;                  // precision 1: 32-k blocks per wave (16 * KB * 32 >= K)
  f32x4 bv[KQ4];
  bf16x8_t whi[KB], wlo[KB];
  if constexpr (PREC == 0) {
#pragma unroll
    for (int s = 0; s < KQ4; ++s) {
      const int k = kb + 16 * s;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(brow + min(k, K - 4));
      bv[s] = (bvalid && k < K) ? v : zero;
    }
  } else {
#pragma unroll
    for (int i = 0; i < KB; ++i) load_w8(bvalid ? brow : nullptr, 32 * (wave * KB + i) + 8 * q, K, whi[i], wlo[i]);
  }
  // published tile: column chunks in MFMA-A lane order, f32 (precision 0) or hi / lo bf16 planes (see rnn_fwd_persist)
  const int nch = PREC == 0 ? (K + 15) >> 4 : (K + 31) >> 5;
  const size_t tile_f = (size_t)nch * (PREC == 0 ? 256 : 512);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.hx, 0, (int)((size_t)2 * D * nbt * tile_f * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(pa.flags, 0, (int)((size_t)2 * D * nbt * nsl * 4), 0x00020000);
  // per-parity element / byte offsets of this group's flag row and hand-off tile (kept out of the time loop)
  const unsigned flg_el[2] = {(unsigned)(((0 * D + d) * nbt + bt) * nsl), (unsigned)(((1 * D + d) * nbt + bt) * nsl)};
  const unsigned tile_b[2] = {(unsigned)((((size_t)0 * D + d) * nbt + bt) * tile_f * 4), (unsigned)((((size_t)1 * D + d) * nbt + bt) * tile_f * 4)};

  const int bl = tid >> 4, jl = tid & 15, j = j0 + jl, b = b0 + bl;
  const bool item = tid < 256 && bl < Bc && j < H;
  float state = 0.0f;   // carried dc (LSTM) / dh*z (GRU)
  // Wave roles: waves 0..3 hold the 256 (row, unit) items and publish; wave 15 (no items) polls the flags, so its
  // in-order vmcnt never queues behind the item waves' reserve stores and next-step operand loads (requested right
  // after the hand-off, a whole step before the gate math needs them).
  float sv[4] = {0.f, 0.f, 0.f, 0.f}, dyv = 0.f, e0 = 0.f, e1 = 0.f;
  // per-thread byte offsets inside one timestep slab (see slab_rsrc); the backward pass walks time in the direction
  // tdir = -1 for the forward layer direction, +1 for the reverse one
  const int tdir = d == 0 ? -1 : 1;
  const long slab_g = (long)B * D * K, slab_h = (long)B * D * H;
  const int bcl = min(b, B - 1), jcl = min(j, H - 1);
  const unsigned vg0 = (unsigned)(((bcl * D + d) * K + jcl) * 4);
  const unsigned vg1 = vg0 + (unsigned)(min(1, G - 1) * H * 4), vg2 = vg0 + (unsigned)(min(2, G - 1) * H * 4), vg3 = vg0 + (unsigned)(min(3, G - 1) * H * 4);
  const unsigned vh = (unsigned)(((bcl * D + d) * H + jcl) * 4);
  const bool is_lstm = p.cell == CTCN_CELL_LSTM, is_tanh = p.cell == CTCN_CELL_TANH;
  // e0: c_t (LSTM) / W_hn h_{t-1} (GRU) / y_t (tanh);  e1: c of the next step in walking order (LSTM) / its y (GRU), 0 when
  // that step does not exist (applied where e1 is consumed, never as a select on the value in flight)
  const __amdgpu_buffer_rsrc_t rg = whole_rsrc(p.gates, (size_t)T * slab_g), rdy = whole_rsrc(p.dy, (size_t)T * slab_h);
  const __amdgpu_buffer_rsrc_t r0 = whole_rsrc(is_tanh ? p.y : p.aux, (size_t)T * slab_h), r1 = whole_rsrc(is_lstm ? p.aux : p.y, (size_t)T * slab_h);
  const __amdgpu_buffer_rsrc_t ra = whole_rsrc(p.aux, (size_t)T * slab_h);
  const unsigned sg_b = (unsigned)(slab_g * 4), sh_b = (unsigned)(slab_h * 4);            // bytes per timestep slab
  bool e1_valid = T > 1 && !is_tanh;
  if (item) {
    const int t0 = d == 0 ? T - 1 : 0, t1 = T > 1 ? t0 + tdir : t0;
    const unsigned og = (unsigned)t0 * sg_b, oh = (unsigned)t0 * sh_b;
    sv[0] = ld_slab(rg, vg0, og); sv[1] = ld_slab(rg, vg1, og); sv[2] = ld_slab(rg, vg2, og); sv[3] = ld_slab(rg, vg3, og);
    dyv = ld_slab(rdy, vh, oh); e0 = ld_slab(r0, vh, oh); e1 = ld_slab(r1, vh, (unsigned)t1 * sh_b);
  }
  __syncthreads();

#ifdef CTCN_PERSIST_STATS
  long long bt_poll = 0, bt_mm = 0, bt_red = 0, bt_math = 0, bt_copy = 0, bt_tail = 0, bt_t0 = clock64();
#endif
  for (int s = 0; s < T; ++s) {
#ifdef CTCN_PERSIST_STATS
    const long long q_a = clock64();
    long long q_p = q_a;
#endif
    const int t = d == 0 ? T - 1 - s : s;
    const size_t row_t = (size_t)t * B + b;
    f32x4 acc[1];
    acc[0] = zero;
    if (s > 0) {
      const int par = (s - 1) & 1;
      if (wave == NW - 1) {
        const unsigned *fl = pa.flags + flg_el[par];
        if (!poll_group(fl, nsl, (unsigned)s, lane, pa) && lane == 0) {
          s_abort = 1;
          if (pa.status) atomicCAS(pa.status, 0, 201);
        }
      }
      lds_barrier();
      if (s_abort) break;
#ifdef CTCN_PERSIST_STATS
      q_p = clock64();
#endif
      const unsigned tbase = tile_b[par];
      f32x4 acc1 = zero;      // two independent accumulator chains hide the dependent MFMA latency
      if constexpr (PREC == 0) {
        f32x4 av[KQ4];
#pragma unroll
        for (int si = 0; si < KQ4; ++si) {
          const int k = kb + 16 * si, c = min(wave * KQ4 + si, nch - 1);
          const f32x4 v = ld_sc1_f4(rs, tbase + (unsigned)((c * 64 + lane) * 16));
          av[si] = (k < K && r < Bc) ? v : zero;
        }
#pragma unroll
        for (int si = 0; si < KQ4; ++si)
#pragma unroll
          for (int c = 0; c < 4; c += 2) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[si][c], bv[si][c], acc[0], 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[si][c + 1], bv[si][c + 1], acc1, 0, 0, 0);
          }
      } else {
        u32x4 ah[KB], al[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const unsigned off = tbase + (unsigned)(min(wave * KB + i, nch - 1) * 2048 + lane * 16);   // blocks past the end: W is 0 there
          ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
          al[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 16);
        }
        __builtin_amdgcn_sched_barrier(0);        // all operand loads in flight before the first MFMA
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const bf16x8_t ahv = __builtin_bit_cast(bf16x8_t, ah[i]), alv = __builtin_bit_cast(bf16x8_t, al[i]);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alv, whi[i], acc1, 0, 0, 0);                    // small terms
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahv, wlo[i], acc1, 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahv, whi[i], acc[0], 0, 0, 0);
        }
      }
      acc[0] += acc1;
    }
#ifdef CTCN_PERSIST_STATS
    const long long q_m = clock64();
#endif
    park_tile(red, wave, lane, acc[0]);
    lds_barrier();
#ifdef CTCN_PERSIST_STATS
    const long long q_r = clock64();
    long long q_e = q_r, q_c = q_r;
#endif

    const int par = s & 1;
    const unsigned px = tile_b[par];                                                                // published tile (bytes)
    float out[4] = {0.f, 0.f, 0.f, 0.f};      // values the next step multiplies by W_hh (published), per gate block
    float dan = 0.f;
    float rec = 0.0f;                                      // (d(pre-act)_{next} W_hh)[row bl][unit jl], summed in wave order
    if (tid < 256) {
      const float *rp = red + parked_at(bl, jl);
#pragma unroll
      for (int w = 0; w < NW; ++w) rec += rp[w * RT_T];
    }
    if (item) {
      float dh = dyv + rec;
      const float e1u = e1_valid ? e1 : 0.0f;                 // c / h of a step before the sequence start is 0
      if (p.cell == CTCN_CELL_LSTM) {
        const float i_ = sv[0], f_ = sv[1], g_ = sv[2], o_ = sv[3];
        const float tc = act_tanh(e0);
        const float do_ = dh * tc;
        const float dc = dh * o_ * (1.0f - tc * tc) + state;
        out[0] = dc * g_ * i_ * (1.0f - i_);
        out[1] = dc * e1u * f_ * (1.0f - f_);
        out[2] = dc * i_ * (1.0f - g_ * g_);
        out[3] = do_ * o_ * (1.0f - o_);
        state = dc * f_;
      } else if (p.cell == CTCN_CELL_GRU) {
        dh += state;
        const float r_ = sv[0], z_ = sv[1], n_ = sv[2], hn = e0, hp = e1u;
        const float dn = dh * (1.0f - z_);
        const float dz = dh * (hp - n_);
        dan = dn * (1.0f - n_ * n_);
        out[0] = dan * hn * r_ * (1.0f - r_);
        out[1] = dz * z_ * (1.0f - z_);
        out[2] = dan * r_;
        state = dh * z_;
      } else {
        out[0] = dh * (1.0f - e0 * e0);
      }
      // stage this workgroup's 16 x 16 x G block in LDS in the granule order of the tile
      if (s + 1 < T) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < G) {
            if constexpr (PREC == 0) {
              stage[((k * 4 + (jl >> 2)) * 16 + bl) * 4 + (jl & 3)] = out[k];
            } else {
              unsigned short *sp = reinterpret_cast<unsigned short *>(stage);
              const unsigned hi = f2bf(out[k]);
              sp[((k * 2 + (jl >> 3)) * 16 + bl) * 8 + (jl & 7)] = (unsigned short)hi;
              sp[(((4 + k) * 2 + (jl >> 3)) * 16 + bl) * 8 + (jl & 7)] = f2bf(out[k] - __uint_as_float(hi << 16));
            }
          }
      }
    }
    if (s + 1 < T) {
      lds_barrier();
#ifdef CTCN_PERSIST_STATS
      q_e = clock64();
#endif
      // copy-out by waves 0..3: every lane one 16-B granule, 16 consecutive rows = 256 contiguous bytes of the tile
      if (tid < 256) {
        const int row = tid & 15;
        if constexpr (PREC == 0) {
          const int k = tid >> 6, qq = (tid >> 4) & 3, col = k * H + j0 + 4 * qq;
          if (k < G && row < Bc && j0 + 4 * qq < H) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&stage[((k * 4 + qq) * 16 + row) * 4]);
            st_f4(rs, px + (unsigned)(((col >> 4) * 256 + (((col >> 2) & 3) * 16 + row) * 4) * 4), v, local);
          }
        } else {
          const int plane = tid >> 7, k = (tid >> 5) & 3, oct = (tid >> 4) & 1, col = k * H + j0 + 8 * oct;
          if (k < G && row < Bc && j0 + 8 * oct < H) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned short *>(stage) + (((plane * 4 + k) * 2 + oct) * 16 + row) * 8);
            const unsigned off = px + (unsigned)((col >> 5) * 2048 + plane * 1024 + (((col >> 3) & 3) * 16 + row) * 16);
            if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
          }
        }
      }
    }
    if (s + 1 < T) {
      if (wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores (to L2 / to memory)
      lds_barrier();
#ifdef CTCN_PERSIST_STATS
      q_c = clock64();
#endif
      if (tid == 0) st_u1(rf, (flg_el[par] + (unsigned)slice) * 4, (unsigned)(s + 1), local);
    }
    // off the critical path (item waves): d(pre-activation) for the deferred dW / dX GEMMs leaves after the hand-off,
    // and the next step's saved forward values are requested a whole step ahead
    {
      const unsigned og = (unsigned)t * sg_b, oh = (unsigned)t * sh_b;
      const int tn = s + 1 < T ? t + tdir : t, tp = s + 2 < T ? tn + tdir : tn;             // past the end: re-read (unused)
      const unsigned ogn = (unsigned)tn * sg_b, ohn = (unsigned)tn * sh_b, ohp = (unsigned)tp * sh_b;
#ifdef CTCN_PERSIST_STATS
      const bool dbg_no_st = (pa.poll_depth & 512) != 0, dbg_no_ld = (pa.poll_depth & 256) != 0;
#else
      constexpr bool dbg_no_st = false, dbg_no_ld = false;
#endif
      if (item && !dbg_no_st) {
        if (is_lstm) {
          st_slab(rg, vg0, og, out[0]); st_slab(rg, vg1, og, out[1]); st_slab(rg, vg2, og, out[2]); st_slab(rg, vg3, og, out[3]);
        } else if (p.cell == CTCN_CELL_GRU) {
          st_slab(rg, vg0, og, out[0]); st_slab(rg, vg1, og, out[1]); st_slab(rg, vg2, og, dan);
          st_slab(ra, vh, oh, out[2]);
        } else {
          st_slab(rg, vg0, og, out[0]);
        }
      }
      if (item && !dbg_no_ld) {
        sv[0] = ld_slab(rg, vg0, ogn); sv[1] = ld_slab(rg, vg1, ogn); sv[2] = ld_slab(rg, vg2, ogn); sv[3] = ld_slab(rg, vg3, ogn);
        dyv = ld_slab(rdy, vh, ohn); e0 = ld_slab(r0, vh, ohn); e1 = ld_slab(r1, vh, ohp);
      }
      e1_valid = s + 2 < T && !is_tanh;
    }
#ifdef CTCN_PERSIST_STATS
    { const long long q_z = clock64(); bt_poll += q_p - q_a; bt_mm += q_m - q_p; bt_red += q_r - q_m; bt_math += q_e - q_r; bt_copy += q_c - q_e; bt_tail += q_z - q_c; }
#endif
  }
#ifdef CTCN_PERSIST_STATS
  if (pa.stats && slice == 3 && d == 0 && bt == 0 && (tid == 0 || tid == 960)) {
    long long *o = pa.stats + (tid == 0 ? 0 : 8);
    o[0] = bt_poll; o[1] = bt_mm; o[2] = bt_red; o[3] = bt_math; o[4] = bt_copy; o[5] = bt_tail; o[6] = clock64() - bt_t0;
  }
#endif
  if (s_abort && item)       // poison d(pre-activation) of every frame: dx and both weight gradients of the layer become NaN
    for (int tt = 0; tt < T; ++tt) st_slab(rg, vg0, (unsigned)tt * sg_b, __uint_as_float(0x7fc00000u));
}

// ================================================================================================
// Persistent backward recurrence, SCATTER formulation (the default).
//
// rnn_bwd_persist (above) makes every workgroup pull the whole 16 x G*H tile of d(pre-activation) each step: 80 KB per CU
// and step at H = 320, i.e. 1250 cycles of its 64 B/clk L2 port before the matmul can finish -- the reason the backward
// step was 0.65 us slower than the forward one.  Here the product is split the other way: dh_{t-1} = da_t * W_hh, and a
// workgroup multiplies ITS OWN 16 x (G*16) block of da_t (it has just computed it) by its G*16 rows of W_hh, for all H
// outputs: nsl partial 16 x 16 tiles, one per owner of 16 hidden units, which it scatters as 1-KB blocks (the MFMA C
// layout, one 16-B store per lane).  The owner gathers the nsl partials addressed to it (wave 4 + g loads sources g,
// g + 12, ...), and the sum over sources runs through the same parked-tile / item-sum code as the sum over waves did before.
// Wave roles: waves 0..3 hold the 256 (row, unit) items -- gate math, the staged A operand and ALL reserve traffic (loaded
// two steps ahead into alternating register sets, stored without ever being waited for); waves 4..15 gather, multiply and
// scatter, with nothing else in their vm queue.  A step has two barriers (parked partials, staged A operand).
// Hand-off (TAGGED, the default): no flags.  Every float of a partial tile carries a step tag in its mantissa LSB and the
// gathering wave re-reads its 1-KB block until all 256 floats show it: no store drain on the producer, no flag round trip
// on the consumer (1.99 us per step against 2.14).  (A pause before the first poll of a step, which pays in rnn_fwd_tagged, does not
// here: 0 / 3 / 6 / 9 / 12 sleeps of 64 cycles: 1.85 / 1.83 / 1.85 / 1.88 / 1.98 us.)  !TAGGED: every (owner, source) tile has its own flag, raised by the
// wave that stored it (after draining its stores) and polled by the wave that gathers it.
// Per step a CU now reads 1 KB x nsl and writes 1 KB x nsl (20 KB each at H = 320), d(pre-activation) never travels
// between workgroups (it only goes to the reserve for the deferred GEMMs), and W_hh costs 16 VGPRs per tile.
// k order inside the 64-wide block: precision 1: two 32-k MFMA blocks, lane octet q -> gate 2*blk + (q >> 1), units
// 8*(q & 1) .. +7; precision 0: k = gate * 16 + unit, consumed 4 at a time by v_mfma_f32_16x16x4_f32.
// grid as rnn_bwd_persist; NTW = ceil(nsl / 12) tiles (and source blocks) per exchange wave.
// ================================================================================================
// CELL: the cell type as a compile-time constant (the gate math is then branch-free: with the run-time `p.cell` the item path
// carried ~20 scalar branches and as many phi copies per step)
// SLOW: the parity harness's instantiation (options "rnn_slow_items" / "rnn_slow_exchange", as rnn_fwd_tagged): item waves sleep before they read the parked
// tiles, exchange waves before they read the staged operand, every step -- results must not change (test_rnn_bwd_with_slow_waves)
template <int NTW, int PREC, bool TAGGED, int CELL, bool SLOW = false>
__global__ __launch_bounds__(1024) void rnn_bwd_scatter(PersistArgs pa) {
  constexpr int NW = 16;
  const RnnArgs &p = pa.a;
  __shared__ __attribute__((aligned(16))) float red[NW * RT_T];        // parked partial tiles (one per wave)
  __shared__ __attribute__((aligned(16))) float stage[1024];           // this workgroup's da block as MFMA A operand
  __shared__ uint4 dropw[4][64];                                       // fused dropout gradient: the Philox groups of an item wave's next four steps
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  constexpr int G = CELL == CTCN_CELL_LSTM ? 4 : (CELL == CTCN_CELL_GRU ? 3 : 1);
  const int H = p.H, D = p.D, B = p.B, T = p.T;
  __shared__ int s_ticket;
  const PersistRole role = persist_role(pa, D, &s_ticket);
  if (!role.active) return;
  const int d = role.d, bt = role.bt, nbt = pa.nbt, nsl = pa.nsl, slice = role.slice, local = pa.local;
  const int b0 = bt * 16, j0 = slice * 16;
  const int Bc = min(16, B - b0);
  const int K = G * H;
  const float *WT = d == 0 ? p.w0 : p.w1;                               // W_hh^T: row = hidden unit (output n), K columns
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (tid == 0) s_abort = 0;
  for (int i = tid; i < 1024; i += 1024) stage[i] = 0.0f;              // rows / units nobody owns stay 0

  // Wave roles.  Waves 0..3 hold the (row, unit) items: gate math, the A-operand stage and ALL reserve traffic (loads two
  // steps ahead, stores never waited for) -- they take no part in the exchange, so no "stores landed" wait ever queues
  // behind an HBM access.  Waves 4..15 gather the partial tiles addressed to this workgroup, multiply and scatter: wave
  // 4 + g owns output tiles g, g + 12, ... (tile t = the 16 hidden units of slice t) and nothing else is in its vm queue.
  constexpr int NGW = 12, NTG = NTW;
  const int gw = wave - 4;
  // W operand of this lane for its NTW output tiles
  bf16x8_t whi[NTW][2], wlo[NTW][2];
  float wf[NTW][16];
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
    const int own = wave >= 4 ? gw + NGW * tw : 0;
    const int n = 16 * own + r;                                        // output unit of this lane's B column
    const bool nvalid = wave >= 4 && own < nsl && n < H;
    const float *wrow = WT + (size_t)min(n, H - 1) * K;
    if constexpr (PREC == 1) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int gate = 2 * blk + (q >> 1), u0 = j0 + 8 * (q & 1);
        const bool kv = nvalid && gate < G && u0 < H;                  // H % 8 == 0: an octet is valid as a whole
        load_w8(kv ? wrow + (size_t)gate * H + u0 : nullptr, 0, 8, whi[tw][blk], wlo[tw][blk]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int kk = 4 * m + q, gate = kk >> 4, unit = j0 + (kk & 15);
        const float v = wrow[(size_t)min(gate, G - 1) * H + min(unit, H - 1)];
        wf[tw][m] = (nvalid && gate < G && unit < H) ? v : 0.0f;
      }
    }
  }
  // partial blocks: [par][d][bt][owner slice][source slice][256 floats in MFMA C order]
  const size_t tile_f = (size_t)nsl * nsl * 256;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.hx, 0, (int)((size_t)2 * D * nbt * tile_f * 4), 0x00020000);
  // one flag per (owner, source) block: the wave that stored the block raises it, the wave that gathers it polls it -- no
  // workgroup-wide flag, hence no barrier between "all producers done" and the loads, none between the stores and the flag
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(pa.flags, 0, (int)((size_t)2 * D * nbt * nsl * nsl * 4), 0x00020000);
  const unsigned flg_el[2] = {(unsigned)(((0 * D + d) * nbt + bt) * nsl * nsl), (unsigned)(((1 * D + d) * nbt + bt) * nsl * nsl)};
  const unsigned tile_b[2] = {(unsigned)((((size_t)0 * D + d) * nbt + bt) * tile_f * 4), (unsigned)((((size_t)1 * D + d) * nbt + bt) * tile_f * 4)};

  const int bl = tid >> 4, jl = tid & 15, j = j0 + jl, b = b0 + bl;
  const bool item = tid < 256 && bl < Bc && j < H;
  float state = 0.0f;   // carried dc (LSTM) / dh*z (GRU)
  // saved activations / dy / c / c_prev of the item, two sets: set (u & 1) holds step u and is refilled for step u + 2 as
  // soon as step u has consumed it, so a reserve load has two whole steps to return (HBM under the side stream's GEMMs)
  float sv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dyv[2] = {0.f, 0.f}, e0[2] = {0.f, 0.f}, e1[2] = {0.f, 0.f};
  const int tdir = d == 0 ? -1 : 1;
  const long slab_g = (long)B * D * K, slab_h = (long)B * D * H;
  const int bcl = min(b, B - 1), jcl = min(j, H - 1);
  const unsigned vg0 = (unsigned)(((bcl * D + d) * K + jcl) * 4);
  const unsigned vg1 = vg0 + (unsigned)(min(1, G - 1) * H * 4), vg2 = vg0 + (unsigned)(min(2, G - 1) * H * 4), vg3 = vg0 + (unsigned)(min(3, G - 1) * H * 4);
  const unsigned vh = (unsigned)(((bcl * D + d) * H + jcl) * 4);
  constexpr bool is_lstm = CELL == CTCN_CELL_LSTM, is_tanh = CELL == CTCN_CELL_TANH;
  const __amdgpu_buffer_rsrc_t rg = whole_rsrc(p.gates, (size_t)T * slab_g), rdy = whole_rsrc(p.dy, (size_t)T * slab_h);
  const __amdgpu_buffer_rsrc_t r0 = whole_rsrc(is_tanh ? p.y : p.aux, (size_t)T * slab_h), r1 = whole_rsrc(is_lstm ? p.aux : p.y, (size_t)T * slab_h);
  const __amdgpu_buffer_rsrc_t ra = whole_rsrc(p.aux, (size_t)T * slab_h);
  const unsigned sg_b = (unsigned)(slab_g * 4), sh_b = (unsigned)(slab_h * 4);
  // reserve values of step u (timestep tu; e1 belongs to the timestep of step u + 1) into set `ps`; steps past the end re-read the last one (unused)
  auto load_set = [&](int u, auto PSC) {
    constexpr int ps = decltype(PSC)::value;
    const int uc = min(u, T - 1), tu = d == 0 ? T - 1 - uc : uc, tu1 = uc + 1 < T ? tu + tdir : tu;
    const unsigned og = (unsigned)tu * sg_b, oh = (unsigned)tu * sh_b;
    ld_slab_untracked(sv[ps][0], rg, vg0, og); ld_slab_untracked(sv[ps][1], rg, vg1, og); ld_slab_untracked(sv[ps][2], rg, vg2, og);
    ld_slab_untracked(sv[ps][3], rg, vg3, og);
    ld_slab_untracked(dyv[ps], rdy, vh, oh); ld_slab_untracked(e0[ps], r0, vh, oh); ld_slab_untracked(e1[ps], r1, vh, (unsigned)tu1 * sh_b);
  };
  // fused dropout gradient (ctcn_rnn_bwd_dropout): as in rnn_fwd_tagged, an item wave computes the 64 Philox groups of its next four steps
  // at once (lane = step * 16 + row * 4 + unit quad) and parks them in LDS; the element index is that of dy[t][b][d*H + j]
  auto drop_block = [&](int s0) {
    const int sk = min(s0 + (lane >> 4), T - 1), tk = d == 0 ? T - 1 - sk : sk;
    const int br = min(b0 + 4 * wave + ((lane >> 2) & 3), B - 1), j4 = min(j0 + 4 * (lane & 3), H - 4);
    const unsigned idxk = (unsigned)((((size_t)tk * B + br) * D + d) * H + j4);
    uint32_t rr[4];
    philox4(pa.drop_seed, pa.drop_off + (idxk >> 2), rr);
    dropw[wave][lane] = make_uint4(rr[0], rr[1], rr[2], rr[3]);
  };
  if (pa.drop_bwd && tid < 256) drop_block(0);
  if (item) {
    load_set(0, std::integral_constant<int, 0>{});
    load_set(1, std::integral_constant<int, 1>{});
  }
  __syncthreads();

#ifdef CTCN_PERSIST_STATS
  long long zt[6] = {0, 0, 0, 0, 0, 0}, zi[4] = {0, 0, 0, 0}, zt0 = clock64();
#endif
  auto step = [&](const int s, auto PSC) {
    constexpr int ps = decltype(PSC)::value;
#ifdef CTCN_PERSIST_STATS
    const long long z_a = clock64();
    long long z_p = z_a, z_g = z_a, z_m = z_a, z_s = z_a, z_d = z_a;
#endif
    const int t = d == 0 ? T - 1 - s : s;
    uint32_t dropword = 0;
    if (pa.drop_bwd && tid < 256)
      dropword = reinterpret_cast<const uint32_t *>(&dropw[wave][(s & 3) * 16 + (lane >> 4) * 4 + ((lane & 15) >> 2)])[lane & 3];
    float rec = 0.0f;                                      // (da_{next} W_hh)[row bl][unit jl]
    if (s > 0) {
      const int par = (s - 1) & 1;
      if (wave >= 4) {
        // gather: the partial tiles addressed to this workgroup, sources gw, gw + 12, ... (summed in that fixed order)
        f32x4 sum = zero;
        if constexpr (!TAGGED) {
          // each block is loaded as soon as ITS flag shows this step
#pragma unroll
          for (int tg = 0; tg < NTG; ++tg) {
            const int src = gw + NGW * tg;
            if (src < nsl) {
              const unsigned *fl = pa.flags + flg_el[par] + slice * nsl + src;
              for (int spins = 0; __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)s; ++spins) {
                if (spins > pa.spin_limit || ((spins & 63) == 63 && pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                  if (pa.status) atomicCAS(pa.status, 0, 201);        // give up: the launch finishes with a poisoned output
                  break;
                }
                __builtin_amdgcn_s_sleep(1);
              }
              sum += ld_sc1_f4(rs, tile_b[par] + (unsigned)(((slice * nsl + src) * 64 + lane) * 16));
            }
          }
        } else {
          // no flags: the blocks themselves are polled, one after the other (polling a wave's two blocks together measured
          // 4 % slower: the second is nearly always there by the time the first one is).  A producer wrote its block at step
          // s - 1 with tag ((s-1)>>1 & 1) ^ 1 in the LSB of every float (the buffer starts zeroed and the tag alternates between
          // the uses of a parity buffer); a block counts once all 256 floats carry the tag, so a half-landed block (any
          // granularity down to a dword) is simply read again
          const unsigned tb = ((((unsigned)(s - 1)) >> 1) & 1u) ^ 1u;
#pragma unroll
          for (int tg = 0; tg < NTG; ++tg) {
            const int src = gw + NGW * tg;
            if (src < nsl) {
              const unsigned boff = tile_b[par] + (unsigned)(((slice * nsl + src) * 64 + lane) * 16);
              for (int spins = 0;; ++spins) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 16);
                const bool okl = ((v.x & v.y & v.z & v.w & 1u) == tb) && (((v.x | v.y | v.z | v.w) & 1u) == tb);
                if (__builtin_amdgcn_ballot_w64(okl) == ~0ull) {
                  sum += (f32x4){__uint_as_float(v.x & ~1u), __uint_as_float(v.y & ~1u), __uint_as_float(v.z & ~1u), __uint_as_float(v.w & ~1u)};
                  break;
                }
                if (spins > pa.spin_limit || ((spins & 63) == 63 && pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                  if (pa.status) atomicCAS(pa.status, 0, 202);          // give up: the launch finishes with a poisoned output
                  break;
                }
                __builtin_amdgcn_s_sleep(1);
              }
            }
          }
        }
        park_tile(red, gw, lane, sum);
      }
      lds_barrier();
#ifdef CTCN_PERSIST_STATS
      z_p = z_g = clock64();
#endif
      if (tid < 256) {
        if constexpr (SLOW) { for (int i = 0; i < pa.slow; ++i) __builtin_amdgcn_s_sleep(1); }
        const float *rp = red + parked_at(bl, jl);
#pragma unroll
        for (int w = 0; w < NGW; ++w) rec += rp[w * RT_T];
      }
    }
#ifdef CTCN_PERSIST_STATS
    if (rec == 12345.678f) zi[3] += 1;         // consume the sum before the stamp
    const long long z_i0 = clock64();
#endif

    float out[4] = {0.f, 0.f, 0.f, 0.f};      // d(pre-activation) the next step multiplies by W_hh, per gate block
    float dan = 0.f;
    // the seven reserve values of this step have landed once at most the seven loads of the NEXT step's set (issued one step
    // later, in order behind them) are still in flight; stores, which vmcnt counts too, can only make this wait longer
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(sv[ps][0]), "+v"(sv[ps][1]), "+v"(sv[ps][2]), "+v"(sv[ps][3]), "+v"(dyv[ps]), "+v"(e0[ps]), "+v"(e1[ps])::"memory");
    if (item) {
      float dyd = dyv[ps];
      if (pa.drop_bwd)            // (a separate multiply, as the dropout pass would round it, then the add)
        dyd = ((dropword >> 8) * (1.0f / 16777216.0f) >= pa.drop_p) ? __fmul_rn(dyd, pa.drop_scale) : 0.0f;
      float dh = dyd + rec;
      const float e1u = (s + 1 < T && !is_tanh) ? e1[ps] : 0.0f;   // c / h of a step before the sequence start is 0
      if constexpr (CELL == CTCN_CELL_LSTM) {
        const float i_ = sv[ps][0], f_ = sv[ps][1], g_ = sv[ps][2], o_ = sv[ps][3];
        const float tc = act_tanh(e0[ps]);
        const float do_ = dh * tc;
        const float dc = dh * o_ * (1.0f - tc * tc) + state;
        out[0] = dc * g_ * i_ * (1.0f - i_);
        out[1] = dc * e1u * f_ * (1.0f - f_);
        out[2] = dc * i_ * (1.0f - g_ * g_);
        out[3] = do_ * o_ * (1.0f - o_);
        state = dc * f_;
      } else if constexpr (CELL == CTCN_CELL_GRU) {
        dh += state;
        const float r_ = sv[ps][0], z_ = sv[ps][1], n_ = sv[ps][2], hn = e0[ps], hp = e1u;
        const float dn = dh * (1.0f - z_);
        const float dz = dh * (hp - n_);
        dan = dn * (1.0f - n_ * n_);
        out[0] = dan * hn * r_ * (1.0f - r_);
        out[1] = dz * z_ * (1.0f - z_);
        out[2] = dan * r_;
        state = dh * z_;
      } else {
        out[0] = dh * (1.0f - e0[ps] * e0[ps]);
      }
      if (s + 1 < T) {                                        // A operand of this workgroup's product, in MFMA order
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < G) {
            if constexpr (PREC == 1) {
              unsigned short *sp = reinterpret_cast<unsigned short *>(stage);
              const unsigned hi = f2bf(out[k]);
              const int o = ((((k >> 1) * 4) + (k & 1) * 2 + (jl >> 3)) * 16 + bl) * 8 + (jl & 7);     // [blk][q][row][8]
              sp[o] = (unsigned short)hi;
              sp[1024 + o] = f2bf(out[k] - __uint_as_float(hi << 16));
            } else {
              const int kk = k * 16 + jl;                                                              // [m][kq][row]
              stage[((kk >> 2) * 4 + (kk & 3)) * 16 + bl] = out[k];
            }
          }
      }
    }
#ifdef CTCN_PERSIST_STATS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long z_i1 = clock64();
#endif
    if (s + 1 < T) {
      lds_barrier();
#ifdef CTCN_PERSIST_STATS
      z_m = clock64();
      zi[0] += z_i0 - z_g; zi[1] += z_i1 - z_i0; zi[2] += z_m - z_i1;
#endif
      const int par = s & 1;
      if (wave >= 4) {
      if constexpr (SLOW) { for (int i = 0; i < pa.slow_x; ++i) __builtin_amdgcn_s_sleep(1); }
      f32x4 acc[NTW];
      if constexpr (PREC == 1) {
        const unsigned short *sp = reinterpret_cast<const unsigned short *>(stage);
        bf16x8_t ah[2], al[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          ah[blk] = *reinterpret_cast<const bf16x8_t *>(sp + ((blk * 4 + q) * 16 + r) * 8);
          al[blk] = *reinterpret_cast<const bf16x8_t *>(sp + 1024 + ((blk * 4 + q) * 16 + r) * 8);
        }
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw) {
          acc[tw] = zero;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            acc[tw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[blk], whi[tw][blk], acc[tw], 0, 0, 0);   // small terms first
            acc[tw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[blk], wlo[tw][blk], acc[tw], 0, 0, 0);
            acc[tw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[blk], whi[tw][blk], acc[tw], 0, 0, 0);
          }
          if constexpr (TAGGED) {
            // round 3: a tile's block leaves as soon as its chain ends -- the CU's store path (nsl x 1 KB per step, ~50 cycles of issue per
            // block and wave, section 5c of DESIGN.md) then works while the matrix pipes finish the wave's other tile(s)
            const int owner = gw + NGW * tw;
            if (owner < nsl) {
              const unsigned tb = ((((unsigned)s) >> 1) & 1u) ^ 1u;
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[tw][e] = __uint_as_float((__float_as_uint(acc[tw][e]) & ~1u) | tb);
              st_f4(rs, tile_b[par] + (unsigned)(((owner * nsl + slice) * 64 + lane) * 16), acc[tw], local);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        float af[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) af[m] = stage[(m * 4 + q) * 16 + r];
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw) {
          acc[tw] = zero;
#pragma unroll
          for (int m = 0; m < 16; ++m) acc[tw] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], wf[tw][m], acc[tw], 0, 0, 0);
        }
      }
      // scatter: tile tw of this wave belongs to owner slice gw + 12*tw; block (owner, source = this slice)
      if constexpr (!TAGGED || PREC != 1) {
#pragma unroll
      for (int tw = 0; tw < NTW; ++tw) {
        const int owner = gw + NGW * tw;
        if (owner < nsl) {
          if constexpr (TAGGED) {
            const unsigned tb = ((((unsigned)s) >> 1) & 1u) ^ 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[tw][e] = __uint_as_float((__float_as_uint(acc[tw][e]) & ~1u) | tb);
          }
          st_f4(rs, tile_b[par] + (unsigned)(((owner * nsl + slice) * 64 + lane) * 16), acc[tw], local);
        }
      }
      }
#ifdef CTCN_PERSIST_STATS
      z_s = clock64();
#endif
      if constexpr (!TAGGED) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's block stores have landed (L2 / memory) ...
      if (lane == 0) {
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw) {
          const int owner = gw + NGW * tw;
          if (owner < nsl) st_u1(rf, (flg_el[par] + (unsigned)(owner * nsl + slice)) * 4, (unsigned)(s + 1), local);   // ... raise their flags
        }
      }
      }
      }
#ifdef CTCN_PERSIST_STATS
      z_d = clock64();
#endif
    }
    // off the critical path (item waves): d(pre-activation) for the deferred dW / dX GEMMs, next step's saved values
    {
      const unsigned og = (unsigned)t * sg_b, oh = (unsigned)t * sh_b;
      if (item) {
        if constexpr (is_lstm) {
          st_slab_untracked(rg, vg0, og, out[0]); st_slab_untracked(rg, vg1, og, out[1]); st_slab_untracked(rg, vg2, og, out[2]); st_slab_untracked(rg, vg3, og, out[3]);
        } else if constexpr (CELL == CTCN_CELL_GRU) {
          st_slab_untracked(rg, vg0, og, out[0]); st_slab_untracked(rg, vg1, og, out[1]); st_slab_untracked(rg, vg2, og, dan);
          st_slab_untracked(ra, vh, oh, out[2]);
        } else {
          st_slab_untracked(rg, vg0, og, out[0]);
        }
        load_set(s + 2, PSC);                                 // this set is free again: refill it for step s + 2
      }
      if (pa.drop_bwd && tid < 256 && (s & 3) == 3) drop_block(s + 1);
    }
#ifdef CTCN_PERSIST_STATS
    { const long long z_z = clock64(); zt[0] += z_p - z_a; zt[1] += z_g - z_p; zt[2] += z_m - z_g; zt[3] += z_s - z_m; zt[4] += z_d - z_s; zt[5] += z_z - z_d; }
#endif
  };
  for (int s = 0; s < T; s += 2) {                            // unrolled by two: the reserve set of a step is a compile-time index
    step(s, std::integral_constant<int, 0>{});
    if (s + 1 < T) step(s + 1, std::integral_constant<int, 1>{});
  }
#ifdef CTCN_PERSIST_STATS
  if (pa.stats && slice == 3 && d == 0 && bt == 0 && (tid == 0 || tid == 960)) {
    long long *o = pa.stats + (tid == 0 ? 0 : 8);
    for (int i = 0; i < 6; ++i) o[i] = zt[i];
    o[6] = clock64() - zt0;
    if (tid == 0) for (int i = 0; i < 3; ++i) pa.stats[16 + i] = zi[i];
  }
#endif
  const bool bad = pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (bad && item)           // poison d(pre-activation) of every frame: dx and both weight gradients of the layer become NaN
    for (int tt = 0; tt < T; ++tt) st_slab(rg, vg0, (unsigned)tt * sg_b, __uint_as_float(0x7fc00000u));
}

// ================================================================================================
// Persistent backward recurrence, SCATTER formulation with ITEM-WAVE GATHER (round 3; precision 1, tagged hand-off).
//
// rnn_bwd_scatter above pays two workgroup barriers per step: gathering waves sum the partial tiles addressed to the workgroup and park
// them (barrier), the item waves add the twelve parked partials, run the gate math and stage the A operand (barrier), the exchange
// waves multiply and scatter.  Here the item waves gather THEMSELVES: lane (row, unit) of item wave w needs, of every source block,
// exactly one float, and the 64 floats of the wave are one contiguous 256-B quarter of the block (MFMA C layout: rows 4w .. 4w+3) --
// nsl coalesced dword loads per lane, all in flight together, each dword carrying its own tag, summed in a fixed order in registers.
// No park, no partial sums through LDS, ONE barrier per step (staged A operand).
// That only works if nothing else sits in the item waves' in-order load queue, so ALL reserve traffic moves to the exchange waves,
// which have nothing on the chain any more between their scatter and the next barrier:
//   * loads: an exchange wave brings an array (one of the four saved gates, dy, c / W_hn h, c_prev / h_prev) of all 256 items of a step
//     into LDS with ONE global_load_lds_dwordx4 (lane = row x unit quad), three steps ahead into a ring of three sets; before the barrier
//     of each step it waits with vmcnt(k), k = its loads per step: loads return in order, so "at most k operations outstanding" means
//     every load older than the youngest k has landed -- two whole steps per load, as before, and 64-bit addresses (no 4-GB limit);
//   * stores: the item waves leave d(pre-activation) as float32 in LDS next to the bf16 A operand, each exchange wave stores one gate
//     with 16-B stores (64 B contiguous per row instead of 4-B pieces).
// The A operand, the float32 copy and the reserve sets are double / triple buffered, so the one barrier orders everything.
// 256 + 64 * NEW threads: four item waves + NEW exchange waves, each owning NTE = ceil(nsl / NEW) output tiles (W_hh fragments: 16 VGPRs per
// tile) -- the product is run with NEW = 8 (768 threads, two exchange waves per SIMD, 168 VGPRs per wave: the fragments and the polled
// registers of an item lane cannot share the 128 of a 1024-thread workgroup, and ONE exchange wave per SIMD issues an MFMA only every 16
// cycles where the pipe takes one every ~8 from two).  A tile's 6 MFMAs run as one dependent chain and its block is stored as soon as the
// chain ends, so the CU's store path works during the other tiles' MFMAs.
// W_hh fragments, tags, tile layout in the hand-off buffer and the role placement are those of rnn_bwd_scatter; an item lane group g sums
// the blocks 4 i + g in ascending order, the four groups are combined lower group first.  NTE <= 5: up to 40 slices (H <= 640).
// ================================================================================================
// SLOW: as rnn_bwd_scatter (item waves sleep at the start of their gather, exchange waves behind the barrier)
template <int NEW, int NTE, int CELL, bool SLOW = false>
__global__ __launch_bounds__(256 + 64 * NEW) void rnn_bwd_scatter2(PersistArgs pa) {
  const RnnArgs &p = pa.a;
  constexpr int NTHR = 256 + 64 * NEW, NPL = (NEW * NTE + 3) / 4;       // NPL: 16-B poll loads per item lane (4 blocks each)
  __shared__ __attribute__((aligned(16))) float stage[2][1024];        // da block as MFMA A operand (bf16 hi | lo), double buffered
  __shared__ __attribute__((aligned(16))) float outf[2][4][256];       // da block as float32 [gate][row][unit]: the reserve stores
  __shared__ __attribute__((aligned(16))) float resv[3][7][256];       // reserve values of three steps: [array][row][unit]
  __shared__ uint4 dropw[4][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // scalar: every role decision below is a scalar branch
  const int r = lane & 15, q = lane >> 4;
  constexpr int G = CELL == CTCN_CELL_LSTM ? 4 : (CELL == CTCN_CELL_GRU ? 3 : 1);
  constexpr bool is_lstm = CELL == CTCN_CELL_LSTM, is_gru = CELL == CTCN_CELL_GRU, is_tanh = CELL == CTCN_CELL_TANH;
  const int H = p.H, D = p.D, B = p.B, T = p.T;
  __shared__ int s_ticket;
  const PersistRole role = persist_role(pa, D, &s_ticket);
  if (!role.active) return;
  const int d = role.d, bt = role.bt, nbt = pa.nbt, nsl = pa.nsl, slice = role.slice, local = pa.local;
  const int b0 = bt * 16, j0 = slice * 16;
  const int Bc = min(16, B - b0);
  const int K = G * H;
  const float *WT = d == 0 ? p.w0 : p.w1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 2048; i += NTHR) (&stage[0][0])[i] = 0.0f;        // rows / units nobody owns stay 0

  const int ew = wave - 4;                         // exchange wave 0 .. NEW-1 owns output tiles ew, ew + NEW, ...
  bf16x8_t whi[NTE][2], wlo[NTE][2];
#pragma unroll
  for (int tw = 0; tw < NTE; ++tw) {
    const int own = wave >= 4 ? ew + NEW * tw : 0;
    const int n = 16 * own + r;
    const bool nvalid = wave >= 4 && own < nsl && n < H;
    const float *wrow = WT + (size_t)min(n, H - 1) * K;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int gate = 2 * blk + (q >> 1), u0 = j0 + 8 * (q & 1);
      const bool kv = nvalid && gate < G && u0 < H;
      load_w8(kv ? wrow + (size_t)gate * H + u0 : nullptr, 0, 8, whi[tw][blk], wlo[tw][blk]);
    }
  }
  const size_t tile_f = (size_t)nsl * nsl * 256;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.hx, 0, (int)((size_t)2 * D * nbt * tile_f * 4), 0x00020000);
  const unsigned tile_b[2] = {(unsigned)((((size_t)0 * D + d) * nbt + bt) * tile_f * 4), (unsigned)((((size_t)1 * D + d) * nbt + bt) * tile_f * 4)};

  const int bl = tid >> 4, jl = tid & 15, j = j0 + jl, b = b0 + bl;
  const bool item = tid < 256 && bl < Bc && j < H;
  float state = 0.0f;
  const int tdir = d == 0 ? -1 : 1;
  const size_t slab_g = (size_t)B * D * K, slab_h = (size_t)B * D * H;

  // ---- reserve traffic of the exchange waves -------------------------------------------------------------------------------
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  // lane = (row lane >> 2, unit quad lane & 3); rows / quads outside the tile read a clamped (valid) address and are never used
  const int xrow = min(b0 + (lane >> 2), B - 1), xj = min(j0 + 4 * (lane & 3), H - 4);
  const size_t xg = ((size_t)xrow * D + d) * K + xj, xh = ((size_t)xrow * D + d) * H + xj;      // float offsets inside a timestep slab
  const bool xvalid = (lane >> 2) < Bc && j0 + 4 * (lane & 3) < H;
  // arrays of the cell: LSTM 0..6; GRU 0, 1, 2, 4, 5, 6; tanh 4, 5.  Exchange wave e loads arrays e and e + NEW.
  auto needed = [&](int a) { return a < 7 && (is_lstm || (is_gru ? a != 3 : (a == 4 || a == 5))); };
  auto dma1 = [&](int a, int x, int set) {         // array a of step x (clamped) -> resv[set][a]
    const int xc = min(x, T - 1), tx = d == 0 ? T - 1 - xc : xc, tx1 = xc + 1 < T ? tx + tdir : tx;
    const float *src;
    if (a < 4) src = p.gates + (size_t)tx * slab_g + xg + (size_t)min(a, G - 1) * H;
    else if (a == 4) src = p.dy + (size_t)tx * slab_h + xh;
    else if (a == 5) src = (is_tanh ? p.y : p.aux) + (size_t)tx * slab_h + xh;
    else src = (is_lstm ? p.aux : p.y) + (size_t)tx1 * slab_h + xh;
    if (pa.rsv_nt) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&resv[set][a][0], 16, 0, 2);      // aux 2 = nt
    else __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&resv[set][a][0], 16, 0, 0);
  };
  auto dma = [&](int x, int set) {
    if (needed(ew)) dma1(ew, x, set);
    if (needed(ew + NEW)) dma1(ew + NEW, x, set);
  };
  const int ndma = wave >= 4 ? (needed(ew) ? 1 : 0) + (needed(ew + NEW) ? 1 : 0) : 0;
  // fused dropout gradient: as rnn_bwd_scatter
  auto drop_block = [&](int s0) {
    const int sk = min(s0 + (lane >> 4), T - 1), tk = d == 0 ? T - 1 - sk : sk;
    const int br = min(b0 + 4 * wave + ((lane >> 2) & 3), B - 1), j4 = min(j0 + 4 * (lane & 3), H - 4);
    const size_t idxk = (((size_t)tk * B + br) * D + d) * H + j4;           // 64-bit: this kernel also serves reserves past 4 GB
    uint32_t rr[4];
    philox4(pa.drop_seed, pa.drop_off + (idxk >> 2), rr);
    dropw[wave][lane] = make_uint4(rr[0], rr[1], rr[2], rr[3]);
  };
  if (pa.drop_bwd && wave < 4) drop_block(0);
  if (wave >= 4) { dma(0, 0); dma(1, 1); dma(2, 2); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // byte offset of this item lane's float inside a partial block (MFMA C layout: float ((row >> 2) * 16 + unit) * 4 + (row & 3))
  const unsigned poll_off = (unsigned)(slice * nsl) * 1024u + (unsigned)(wave & 3) * 256u + (unsigned)((lane & 15) * 16 + (lane >> 4) * 4);
  int set = 0;                                     // s % 3
#ifdef CTCN_PERSIST_STATS
  long long zs[6] = {0, 0, 0, 0, 0, 0}, zq[3] = {0, 0, 0}, z0 = clock64();
#endif
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? T - 1 - s : s;
    const int sb = s & 1;
#ifdef CTCN_PERSIST_STATS
    const long long z_a = clock64();
    long long z_b = z_a, z_c = z_a;
#endif
    if (wave < 4) {
      // ---------------- item waves: gather, gate math, stage ------------------------------------------------------------------
      if constexpr (SLOW) { for (int i = 0; i < pa.slow; ++i) __builtin_amdgcn_s_sleep(1); }
      uint32_t dropword = 0;
      if (pa.drop_bwd) dropword = reinterpret_cast<const uint32_t *>(&dropw[wave][(s & 3) * 16 + (lane >> 4) * 4 + ((lane & 15) >> 2)])[lane & 3];
      const float *rv = &resv[set][0][0] + tid;
      const float sv0 = rv[0], sv1 = rv[256], sv2 = rv[512], sv3 = rv[768], dyv = rv[1024], e0 = rv[1280], e1 = rv[1536];
      float rec = 0.0f;
      if (s > 0) {
        const int par = (s - 1) & 1;
        const unsigned tb = ((((unsigned)(s - 1)) >> 1) & 1u) ^ 1u;
        const unsigned base = tile_b[par] + poll_off;
        for (int i = 0; i < pa.poll_delay; ++i) __builtin_amdgcn_s_sleep(1);
        // 16-B loads: lane group g = lane >> 4 reads, of block 4 i + g, the wave's 256-B quarter (16 lanes x 16 B: the four rows of one
        // unit per lane) -- NTE loads per lane cover all nsl blocks.  (One dword per lane and block -- the item's own float of every
        // block -- was built first: with L1 bypassed every LANE is an L2 request, 4x as many, and a poll round took ~4 000 cycles.)
        u32x4 v[NPL];
        const unsigned lbase = base - (unsigned)((lane & 15) * 16 + (lane >> 4) * 4) + (unsigned)(lane & 15) * 16u;
        for (int spins = 0;; ++spins) {
          // one round, then all tags at once: tb = 1 -> the AND of the LSBs must be 1, tb = 0 -> their OR must be 0.  A round that finds
          // a block missing is simply repeated (a failed round must be cheap, the successful one minimal)
#pragma unroll
          for (int i = 0; i < NPL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lbase + (unsigned)min(4 * i + q, nsl - 1) * 1024u, 0, 16);   // past nsl: the last block again
          unsigned va = v[0].x & v[0].y & v[0].z & v[0].w, vo = v[0].x | v[0].y | v[0].z | v[0].w;
#pragma unroll
          for (int i = 1; i < NPL; ++i) { va &= v[i].x & v[i].y & v[i].z & v[i].w; vo |= v[i].x | v[i].y | v[i].z | v[i].w; }
          const bool okl = ((tb ? va : ~vo) & 1u) != 0;
#ifdef CTCN_PERSIST_STATS
          zs[4] += 1;
#endif
          if (__builtin_amdgcn_ballot_w64(okl) == ~0ull) break;
          if (spins > pa.spin_limit || ((spins & 63) == 63 && pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (pa.status) atomicCAS(pa.status, 0, 203);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        // lane (g, unit): partial sums over the blocks 4 i + g, for the four rows e of its unit (blocks in ascending order)
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const bool in = 4 * i + q < nsl;                  // (x + 0.0f == x: a select, not a branch)
          ps[0] += in ? __uint_as_float(v[i].x & ~1u) : 0.0f; ps[1] += in ? __uint_as_float(v[i].y & ~1u) : 0.0f;
          ps[2] += in ? __uint_as_float(v[i].z & ~1u) : 0.0f; ps[3] += in ? __uint_as_float(v[i].w & ~1u) : 0.0f;
        }
        // transpose-reduce over the four lane groups: the item of lane (g, unit) is row e = g.  v_permlane32_swap exchanges the upper
        // half of its first operand with the lower half of the second: (p_e, p_{e+2}) -> {lower group's, upper group's} contribution to
        // the row pair this half keeps; v_permlane16_swap does the same for odd / even 16-lane rows.  Fixed order: lower group first.
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps[0]), __float_as_uint(ps[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps[1]), __float_as_uint(ps[3]), false, false);
        const float h0 = __uint_as_float(s02[0]) + __uint_as_float(s02[1]), h1 = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
        const auto sf = __builtin_amdgcn_permlane16_swap(__float_as_uint(h0), __float_as_uint(h1), false, false);
        rec = __uint_as_float(sf[0]) + __uint_as_float(sf[1]);
      }
#ifdef CTCN_PERSIST_STATS
      if (rec == 12345.678f) zs[5] += 1;
      z_b = clock64();
#endif
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      float dan = 0.f;
      if (item) {
        float dyd = dyv;
        if (pa.drop_bwd) dyd = ((dropword >> 8) * (1.0f / 16777216.0f) >= pa.drop_p) ? __fmul_rn(dyd, pa.drop_scale) : 0.0f;
        float dh = dyd + rec;
        const float e1u = (s + 1 < T && !is_tanh) ? e1 : 0.0f;
        if constexpr (is_lstm) {
          const float i_ = sv0, f_ = sv1, g_ = sv2, o_ = sv3;
          const float tc = act_tanh(e0);
          const float do_ = dh * tc;
          const float dc = dh * o_ * (1.0f - tc * tc) + state;
          out[0] = dc * g_ * i_ * (1.0f - i_);
          out[1] = dc * e1u * f_ * (1.0f - f_);
          out[2] = dc * i_ * (1.0f - g_ * g_);
          out[3] = do_ * o_ * (1.0f - o_);
          state = dc * f_;
        } else if constexpr (is_gru) {
          dh += state;
          const float r_ = sv0, z_ = sv1, n_ = sv2, hn = e0, hp = e1u;
          const float dn = dh * (1.0f - z_);
          const float dz = dh * (hp - n_);
          dan = dn * (1.0f - n_ * n_);
          out[0] = dan * hn * r_ * (1.0f - r_);
          out[1] = dz * z_ * (1.0f - z_);
          out[2] = dan * r_;
          state = dh * z_;
        } else {
          out[0] = dh * (1.0f - e0 * e0);
        }
        if (s + 1 < T) {
          unsigned short *sp = reinterpret_cast<unsigned short *>(&stage[sb][0]);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < G) {
              const unsigned hi = f2bf(out[k]);
              const int o = ((((k >> 1) * 4) + (k & 1) * 2 + (jl >> 3)) * 16 + bl) * 8 + (jl & 7);     // [blk][q][row][8]
              sp[o] = (unsigned short)hi;
              sp[1024 + o] = f2bf(out[k] - __uint_as_float(hi << 16));
            }
        }
        // float32 copy for the reserve: LSTM 4 gates; GRU r, z, dan | dan * r (-> aux); tanh 1
        outf[sb][0][tid] = out[0];
        if constexpr (!is_tanh) { outf[sb][1][tid] = out[1]; outf[sb][2][tid] = is_gru ? dan : out[2]; outf[sb][3][tid] = is_gru ? out[2] : out[3]; }
      }
    } else {
      // every reserve load but this wave's youngest `ndma` has landed (see header)
      if (ndma == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (ndma == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#ifdef CTCN_PERSIST_STATS
      z_b = clock64();
#endif
    }
#ifdef CTCN_PERSIST_STATS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    z_c = clock64();
#endif
    lds_barrier();
#ifdef CTCN_PERSIST_STATS
    const long long z_d = clock64();
    long long z_e = z_d;
#endif
    if (wave >= 4) {
      // ---------------- exchange waves: multiply, scatter, reserve traffic ----------------------------------------------------
      if constexpr (SLOW) { for (int i = 0; i < pa.slow_x; ++i) __builtin_amdgcn_s_sleep(1); }
      if (s + 1 < T) {
        const int par = s & 1;
        const unsigned short *sp = reinterpret_cast<const unsigned short *>(&stage[sb][0]);
        bf16x8_t ah[2], al[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          ah[blk] = *reinterpret_cast<const bf16x8_t *>(sp + ((blk * 4 + q) * 16 + r) * 8);
          al[blk] = *reinterpret_cast<const bf16x8_t *>(sp + 1024 + ((blk * 4 + q) * 16 + r) * 8);
        }
        const unsigned tbs = ((((unsigned)s) >> 1) & 1u) ^ 1u;
#ifdef CTCN_PERSIST_STATS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long z_x0 = clock64();
        zq[0] += z_x0 - z_d;
        __builtin_amdgcn_sched_barrier(0);
#endif
        // one chain after the other (a dependent chain issues at the same 16 cycles per MFMA as independent ones, tools/mb_mfma16.hip),
        // each tile's block stored as soon as its chain ends: the CU's store path (20 x 1 KB per step, ~200 cycles per block when all
        // exchange waves store together) then works WHILE the matrix pipes do, instead of after them
#pragma unroll
        for (int tw = 0; tw < NTE; ++tw) {
          const int owner = ew + NEW * tw;
          if (owner < nsl) {
            f32x4 acc = zero;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[blk], whi[tw][blk], acc, 0, 0, 0);   // small terms first
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[blk], wlo[tw][blk], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[blk], whi[tw][blk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __uint_as_float((__float_as_uint(acc[e]) & ~1u) | tbs);
            st_f4(rs, tile_b[par] + (unsigned)(((owner * nsl + slice) * 64 + lane) * 16), acc, local);
          }
        }
#ifdef CTCN_PERSIST_STATS
        __builtin_amdgcn_sched_barrier(0);
        { const long long z_x3 = clock64(); zq[2] += z_x3 - z_x0; }
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#ifdef CTCN_PERSIST_STATS
      z_e = clock64();
#endif
      if (ew < (is_tanh ? 1 : 4)) {
        const f32x4 vst = *reinterpret_cast<const f32x4 *>(&outf[sb][ew][lane * 4]);
        float *dst = (is_gru && ew == 3) ? p.aux + (size_t)t * slab_h + xh : p.gates + (size_t)t * slab_g + xg + (size_t)ew * H;
        if (xvalid) {
          if (pa.rsv_nt) __builtin_nontemporal_store(vst, reinterpret_cast<f32x4 *>(dst));
          else *reinterpret_cast<f32x4 *>(dst) = vst;
        }
      }
      // LAST in program order: the vmcnt(ndma) wait before the next barrier then lets exactly these loads stay in flight, whatever the
      // order in which the counter retires loads against stores
      dma(s + 3, set);                             // set s % 3 was consumed before the barrier above
    } else {
      if (pa.drop_bwd && (s & 3) == 3) drop_block(s + 1);
    }
    set = set == 2 ? 0 : set + 1;
#ifdef CTCN_PERSIST_STATS
    { const long long z_f = clock64(); zs[0] += z_b - z_a; zs[1] += z_c - z_b; zs[2] += z_d - z_c; zs[3] += z_e - z_d; zs[5] += z_f - z_e; }
#endif
  }
#ifdef CTCN_PERSIST_STATS
  if (pa.stats && slice == 3 && d == 0 && bt == 0 && (tid == 0 || tid == 256)) {
    long long *o = pa.stats + (tid == 0 ? 0 : 8);
    for (int i = 0; i < 6; ++i) o[i] = zs[i];
    o[6] = clock64() - z0;
    if (tid == 256) { pa.stats[16] = zq[0]; pa.stats[17] = zq[1]; pa.stats[18] = zq[2]; }
  }
#endif
  const bool bad = pa.status && __hip_atomic_load(pa.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (bad && item)           // poison d(pre-activation) of every frame: dx and both weight gradients of the layer become NaN
    for (int tt = 0; tt < T; ++tt) p.gates[(size_t)tt * slab_g + ((size_t)b * D + d) * K + j] = __uint_as_float(0x7fc00000u);
}

template <int NEW, int CELL>
bool launch_bwd_scatter2_c(int nte, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  constexpr int THR = 256 + 64 * NEW;
  if constexpr (CELL != CTCN_CELL_TANH) {
    if (a.slow > 0 || a.slow_x > 0)           // the parity harness's SLOW instantiations (LSTM / GRU)
      switch (nte) {
        case 1: return launch_resident(rnn_bwd_scatter2<NEW, 1, CELL, true>, grid, THR, 0, st, a, wpx);
        case 2: return launch_resident(rnn_bwd_scatter2<NEW, 2, CELL, true>, grid, THR, 0, st, a, wpx);
        case 3: return launch_resident(rnn_bwd_scatter2<NEW, 3, CELL, true>, grid, THR, 0, st, a, wpx);
        case 4: return launch_resident(rnn_bwd_scatter2<NEW, 4, CELL, true>, grid, THR, 0, st, a, wpx);
        case 5: return launch_resident(rnn_bwd_scatter2<NEW, 5, CELL, true>, grid, THR, 0, st, a, wpx);
        default: return false;
      }
  }
  switch (nte) {
    case 1: return launch_resident(rnn_bwd_scatter2<NEW, 1, CELL>, grid, THR, 0, st, a, wpx);
    case 2: return launch_resident(rnn_bwd_scatter2<NEW, 2, CELL>, grid, THR, 0, st, a, wpx);
    case 3: return launch_resident(rnn_bwd_scatter2<NEW, 3, CELL>, grid, THR, 0, st, a, wpx);
    case 4: return launch_resident(rnn_bwd_scatter2<NEW, 4, CELL>, grid, THR, 0, st, a, wpx);
    case 5: return launch_resident(rnn_bwd_scatter2<NEW, 5, CELL>, grid, THR, 0, st, a, wpx);
    default: return false;
  }
}
// nsl slices -> exchange waves (8: two per SIMD, 768 threads, 168 VGPRs per wave) and tiles per exchange wave
bool launch_bwd_scatter2(int nsl, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  const int nte = (nsl + 7) / 8;
  switch (a.a.cell) {
    case CTCN_CELL_LSTM: return launch_bwd_scatter2_c<8, CTCN_CELL_LSTM>(nte, grid, st, a, wpx);
    case CTCN_CELL_GRU: return launch_bwd_scatter2_c<8, CTCN_CELL_GRU>(nte, grid, st, a, wpx);
    case CTCN_CELL_TANH: return launch_bwd_scatter2_c<8, CTCN_CELL_TANH>(nte, grid, st, a, wpx);
    default: return false;
  }
}

template <bool TAGGED, int CELL>
bool launch_bwd_scatter_c(int ntw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  if constexpr (TAGGED && CELL != CTCN_CELL_TANH) {
    if (a.slow > 0 || a.slow_x > 0)           // the parity harness's SLOW instantiations (tagged hand-off, LSTM / GRU)
      switch (ntw) {
        case 1: return launch_resident(rnn_bwd_scatter<1, 1, TAGGED, CELL, true>, grid, 1024, 0, st, a, wpx);
        case 2: return launch_resident(rnn_bwd_scatter<2, 1, TAGGED, CELL, true>, grid, 1024, 0, st, a, wpx);
        default: return false;
      }
  }
  switch (ntw) {      // precision 1 only: the host never picks the scatter formulation for the f32 matmul
    case 1: return launch_resident(rnn_bwd_scatter<1, 1, TAGGED, CELL>, grid, 1024, 0, st, a, wpx);
    case 2: return launch_resident(rnn_bwd_scatter<2, 1, TAGGED, CELL>, grid, 1024, 0, st, a, wpx);
    default: return false;
  }
}
template <bool TAGGED>
bool launch_bwd_scatter_p(int ntw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  switch (a.a.cell) {
    case CTCN_CELL_LSTM: return launch_bwd_scatter_c<TAGGED, CTCN_CELL_LSTM>(ntw, grid, st, a, wpx);
    case CTCN_CELL_GRU: return launch_bwd_scatter_c<TAGGED, CTCN_CELL_GRU>(ntw, grid, st, a, wpx);
    case CTCN_CELL_TANH: return launch_bwd_scatter_c<TAGGED, CTCN_CELL_TANH>(ntw, grid, st, a, wpx);
    default: return false;
  }
}
bool launch_bwd_scatter(int prec, int ntw, dim3 grid, hipStream_t st, const PersistArgs &a, int wpx) {
  if (!prec) return false;
  return a.tagmode ? launch_bwd_scatter_p<true>(ntw, grid, st, a, wpx) : launch_bwd_scatter_p<false>(ntw, grid, st, a, wpx);
}

template <int PREC>
bool launch_bwd_persist_p(int kq4, dim3 grid, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  switch (kq4) {
    case 1: return launch_resident(rnn_bwd_persist<1, PREC>, grid, 1024, lds, st, a, wpx);
    case 2: return launch_resident(rnn_bwd_persist<2, PREC>, grid, 1024, lds, st, a, wpx);
    case 3: return launch_resident(rnn_bwd_persist<3, PREC>, grid, 1024, lds, st, a, wpx);
    case 4: return launch_resident(rnn_bwd_persist<4, PREC>, grid, 1024, lds, st, a, wpx);
    case 5: return launch_resident(rnn_bwd_persist<5, PREC>, grid, 1024, lds, st, a, wpx);
    case 6: return launch_resident(rnn_bwd_persist<6, PREC>, grid, 1024, lds, st, a, wpx);
    case 8: return launch_resident(rnn_bwd_persist<8, PREC>, grid, 1024, lds, st, a, wpx);
    default: return false;
  }
}
bool launch_bwd_persist(int prec, int kq4, dim3 grid, size_t lds, hipStream_t st, const PersistArgs &a, int wpx) {
  return prec ? launch_bwd_persist_p<1>(kq4, grid, lds, st, a, wpx) : launch_bwd_persist_p<0>(kq4, grid, lds, st, a, wpx);
}

// XCDs of the current device, if a 64-workgroup probe kernel (reads HW_REG_XCC_ID) finds its workgroups dealt evenly
// over them; 1 when the XCD-local hand-off cannot be used.  Probed once per device.
__global__ void xcc_probe_kernel(int *out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 15);
}
extern "C" int ctcn_device_xcds(void) {
  static int cache[64];
  static bool have[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
  if (have[dev]) return cache[dev];
  int nx = 1, h[64];
  int *buf = nullptr;
  if (hipMalloc(&buf, sizeof(h)) == hipSuccess) {
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(64), dim3(64), 0, 0, buf);
    if (hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      int mx = 0;
      for (int i = 0; i < 64; ++i) mx = h[i] > mx ? h[i] : mx;
      nx = mx + 1;
      int cnt[16] = {0};
      for (int i = 0; i < 64; ++i) cnt[h[i] & 15]++;
      for (int x = 0; x < nx; ++x)
        if (cnt[x] * nx != 64) nx = 1;                                  // not dealt evenly: no XCD-local mode
      if (nx > 1 && ctcn_device_cus() % nx != 0) nx = 1;
      if (getenv("CTCN_DEBUG")) {
        fprintf(stderr, "ctcn_device_xcds: cus %d nx %d xcc:", ctcn_device_cus(), nx);
        for (int i = 0; i < 64; ++i) fprintf(stderr, " %d", h[i]);
        fprintf(stderr, "\n");
      }
    } else if (getenv("CTCN_DEBUG")) fprintf(stderr, "ctcn_device_xcds: probe copy failed: %s\n", hipGetErrorString(hipGetLastError()));
    (void)hipFree(buf);
  }
  cache[dev] = nx; have[dev] = true;
  return nx;
}

int pick_kq4(int K, int nwaves, int mt, int budget) {
  const int cand[4] = {5, 4, 2, 1};
  int best = 1, best_cost = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int kq = cand[i];
    if (mt * kq > budget) continue;
    const int stride = nwaves * 16 * kq;
    const int cost = ceil_div(K, stride) * kq;
    if (cost < best_cost) { best_cost = cost; best = kq; }
  }
  return best;
}

template <int MT>
void launch_fwd(int kq4, dim3 grid, hipStream_t st, const RnnArgs &a) {
  switch (kq4) {
    case 5: hipLaunchKernelGGL((rnn_fwd_step<MT, 5>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((rnn_fwd_step<MT, 4>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((rnn_fwd_step<MT, 2>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((rnn_fwd_step<MT, 1>), grid, dim3(256), 0, st, a); break;
  }
}
void launch_bwd(int kq4, dim3 grid, hipStream_t st, const RnnArgs &a) {
  switch (kq4) {
    case 5: hipLaunchKernelGGL((rnn_bwd_step<5>), grid, dim3(1024), 0, st, a); break;
    case 4: hipLaunchKernelGGL((rnn_bwd_step<4>), grid, dim3(1024), 0, st, a); break;
    case 2: hipLaunchKernelGGL((rnn_bwd_step<2>), grid, dim3(1024), 0, st, a); break;
    default: hipLaunchKernelGGL((rnn_bwd_step<1>), grid, dim3(1024), 0, st, a); break;
  }
}

int gates_of(int cell) { return cell == CTCN_CELL_LSTM ? 4 : (cell == CTCN_CELL_GRU ? 3 : 1); }

// which recurrent kernel the last forward / backward call of this process launched (diagnostics: ctcn_rnn_last_kernel; bench.py names
// the kernel its roofline object describes from this, not from what the host expects)
// diagnostic only (ctcn_rnn_last_kernel): pointers to string literals, relaxed atomics -- process-wide, the last call of ANY thread wins (the
// backward pass runs on the autograd thread, the reader is the main thread: a thread-local would hide it)
struct LastKernel { std::atomic<const char *> v{""}; void operator=(const char *s) { v.store(s, std::memory_order_relaxed); } operator const char *() const { return v.load(std::memory_order_relaxed); } };
LastKernel g_last_kernel[2];

// One line on stderr (per distinct reason, per process) when a layer that asked for the persistent recurrence runs one launch
// per timestep instead: that path is 4x slower and nothing else would tell the user.  CTCN_QUIET=1 silences it.
void log_fallback(const char *which, int T, int B, int H, int dirs, const char *reason) {
  static unsigned seen = 0;
  unsigned h = 5381;
  for (const char *c = which; *c; ++c) h = h * 33 + (unsigned char)*c;
  for (const char *c = reason; *c; ++c) h = h * 33 + (unsigned char)*c;
  const unsigned bit = 1u << (h & 31);
  if ((seen & bit) || getenv("CTCN_QUIET")) return;
  seen |= bit;
  fprintf(stderr, "libctcn: %s T=%d B=%d H=%d dirs=%d: persistent recurrence not used (%s) -> one launch per timestep (about 4x slower)\n",
          which, T, B, H, dirs, reason);
}

}  // namespace

extern "C" const char *ctcn_rnn_last_kernel(int which) { return g_last_kernel[which ? 1 : 0]; }

extern "C" size_t ctcn_rnn_scratch_bytes(int cell, int B, int H, int dirs) {
  const size_t G = gates_of(cell);
  return align_up((size_t)dirs * G * H * H * sizeof(float), 256) + align_up((size_t)B * dirs * H * sizeof(float), 256);
}

// (The mirror image for the backward pass -- the dx GEMM of a time chunk on a second side stream as soon as both directions have passed
// the chunk: d(pre-activation) written through to memory, per-chunk counters raised by the item waves, a one-wave wait kernel in
// front of each chunk GEMM -- was built and measured in round 2: bit-identical, but 15.2 instead of 14.3 ms per cfg2 step.  The XCDs
// the backward recurrence leaves idle are already full with the weight-gradient GEMMs of the layer above (6.9 ms of side-stream work
// per step against 6.4 ms of recurrences): the chunk GEMMs finish 0.5 ms AFTER the recurrence and the write-through stores cost it
// 0.13 ms per launch.  Not kept.)
// Dropout of the layer output fused into the forward recurrence (ctcn_rnn_fwd_dropout): the request is consumed by the tagged-gather
// launch; any other path leaves it pending and the dropout kernel runs behind the recurrence instead (the same values either way).
// (round 3: the ABI keeps no state between calls any more.  What used to be armed by ctcn_set_fwd_overlap / ctcn_set_prelaunch_event and
// handed over in thread-locals by the _dropout entry points travels in the caller's `ctcn_rnn_call` (include/ctcn.h); a NULL pointer or a
// zeroed struct is the plain call.)
// ---- the projection pipeline's plan (pure arithmetic; ctcn_diag_pipeline_chunks exposes it to the CPU tests) ---------------------------------
// Returns the number of time chunks (even, 8..24) with which the input projection of a bidirectional layer is pipelined with its forward
// recurrence -- the first pair of chunks in front of the launch, the others by XCD-filtered GEMMs on the side stream while the recurrence
// runs, a counter per pair -- or 0 when no chunking pays.  Every condition below comes from a measured loss (round 4, `tools/ab_shapes.sh`:
// cfg2's model over T, B, H; ms per training step with the pipeline | without):
//  * one round: a chunk's GEMM ((T / n) * B rows x 2 * G * H columns; ragged last tile allowed) is one round of 256-row tiles on the idle
//    XCDs' CUs, at least 3/4 of them busy -- cfg2 (T = 800, B = 32, H = 320): 10 chunks of 2 560 rows = 10 x 10 tiles on 128 CUs: 13.21 | 13.64;
//    T = 1 000: 12 chunks of 2 688 rows: 16.69 | 17.28; T = 600 / 700: eight chunks: 10.34 | 10.45, 11.76 | 12.10; B = 24: 12.54 | 13.09;
//  * at least 72 steps per chunk: a pair costs the side stream ~130-150 us whatever its size (two one-round GEMMs, their splits, the counter),
//    which the recurrence must take at least as long to consume -- T = 400 (cfg3; 50 steps per chunk): 8.51 | 7.88; B = 40 (two idle XCDs: only
//    24 chunks of 34 steps pass the one-round test): 18.8 | 16.2;
//  * throughput: a pair's flops at ~32.5 TFLOP/s per idle XCD (cfg2's pair of 16.8 GFLOP takes 129 us on four) within the time the recurrence
//    needs for the chunk's steps (1.2 + H / 800 us each; measured 1.36 .. 1.7 for H = 128 .. 384), 8 % slack -- H = 384: 16.09 | 15.66;
//  * eight CUs of every recurrence XCD stay free, counting the launch's spare workgroups: the side GEMMs' workgroups dealt to a recurrence XCD
//    must start there to leave -- H = 384 (24 + 3 workgroups per XCD) with only its bottom layer's tiny projection pipelined: 16.4 | 15.7.
// (Measured and not a rule: chunks that fill less than 3/4 of the side CUs but hold >= 72 steps -- B = 16 gains 1.6 %, B = 20 at T = 650 loses 1.7 %.)
// Which physical XCD hosts group g of an XCD-local persistent recurrence (option "xcd_interleave", include/ctcn.h).  An order other than 0 is
// applied only where the recurrence leaves XCDs idle (groups < XCDs) -- where it decides which XCDs the side-stream GEMMs get; a recurrence that
// takes every XCD (cfg4) keeps group g on XCD g.  The host derives its xcd_allow masks from the same rule (ops._idle_xcd_mask: no idle XCD, no mask).
static int xcd_order_for(int nx, int groups) {
  // ("xcd_interleave_force": the parity harness applies the order to a launch that takes every XCD as well -- a relabelling of XCDs, bit-identical
  // results: tests/test_gpu_kernels.py:test_rnn_results_do_not_depend_on_the_xcd_order)
  return (nx == 8 && (groups < nx || ctcn_get_option("xcd_interleave_force") != 0)) ? std::min(std::max(ctcn_get_option("xcd_interleave"), 0), 5) : 0;
}

static int plan_projection_pipeline(int T, int B, int I, int H, int dirs, int G, int nxd, int cus, unsigned xcd_allow, int min_input) {
  if (nxd <= 1 || !xcd_allow || dirs != 2 || I < min_input) return 0;
  const int GH = G * H, N2 = 2 * GH, nidle = __builtin_popcount(xcd_allow), cus_side = cus / nxd * nidle;
  const int wpx = ceil_div(dirs * ceil_div(B, 16), nxd) * (H / 16);
  if (wpx + std::max(2, wpx / 8) + 8 > cus / nxd) return 0;
  const int wnt = (N2 % 256 == 0 || (N2 > 512 && ceil_div(N2, 256) * 256 - N2 <= N2 / 8)) ? 2 : 1, tiles_n = ceil_div(N2, 128 * wnt);
  for (int n : {8, 10, 12, 14, 16, 20, 24}) {
    const int ct = ceil_div(T, n);
    const long rows = (long)ct * B, tiles = (rows + 255) / 256 * tiles_n;
    const double pair_us = 2.0 * (2.0 * ct * B) * N2 * (double)I / (32.5e6 * nidle), rec_us = ct * (1.2 + H / 800.0);
    if (ct >= 72 && rows >= 1024 && T >= 4 * n && ct * (n - 1) < T && tiles <= cus_side && tiles * 4 >= (long)cus_side * 3 && pair_us <= 1.08 * rec_us) return n;
  }
  return 0;
}
extern "C" int ctcn_diag_pipeline_chunks(int cell, int T, int B, int I, int H, int dirs, int xcds, int cus, unsigned xcd_allow) {
  if (cell < 0 || cell > 2 || T <= 0 || B <= 0 || I <= 0 || H <= 0 || xcds <= 0 || cus <= 0) return -1;
  return plan_projection_pipeline(T, B, I, H, dirs, gates_of(cell), xcds, cus, xcd_allow, 0);
}

static const ctcn_rnn_call k_plain_call = {};
__global__ void set_counter_kernel(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// the recurrence of ctcn_rnn_fwd_ex; `drop_pending`: in: the call asks for the dropped output, out: false once a recurrent kernel has stored it
static int rnn_fwd_impl(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                        const float *w_hh0, const float *w_ih1, const float *w_hh1, float *y, float *gates,
                        float *aux, int precision, void *ws, size_t ws_bytes, void *stream, const ctcn_rnn_call &call, bool &drop_pending) {
  CTCN_REQUIRE(cell >= 0 && cell <= 2, "ctcn_rnn_fwd: unknown cell %d", cell);
  CTCN_REQUIRE(T > 0 && B > 0 && I > 0 && H > 0 && (dirs == 1 || dirs == 2), "ctcn_rnn_fwd: bad dims");
  if (H % 4 != 0) { ctcn_set_error("ctcn_rnn_fwd: hidden size %d must be a multiple of 4", H); return CTCN_EUNSUPPORTED; }
  CTCN_REQUIRE(x && w_ih0 && w_hh0 && y && gates && (dirs == 1 || (w_ih1 && w_hh1)), "ctcn_rnn_fwd: null pointer");
  CTCN_REQUIRE(cell == CTCN_CELL_TANH || aux, "ctcn_rnn_fwd: aux reserve required for LSTM/GRU");
  CTCN_REQUIRE(((uintptr_t)w_hh0 % 16 == 0) && ((uintptr_t)y % 16 == 0) && (dirs == 1 || (uintptr_t)w_hh1 % 16 == 0),
               "ctcn_rnn_fwd: w_hh / y must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int G = gates_of(cell);
  const int GH = G * H;
  const float *w_ih[2] = {w_ih0, w_ih1};
  // both directions project the same x and the gate reserve holds their pre-activations side by side (row = dirs*GH floats):
  // with the two W_ih stacked in the workspace one N = 2*GH product does the work of two (one pass over x, one launch)
  bool proj_done = ctcn_opt_recurrence_only();
  struct { void *stream, *event, *ws; size_t ws_bytes; unsigned xcd_allow; } const ov = {call.side_stream, call.side_event, call.side_ws, call.side_ws_bytes, call.xcd_allow};
  int *const status_word = call.status ? call.status : ctcn_status_word();
  // pipelined projection: needs the tagged-gather kernel (the only one that checks the chunk counter), both W_ih stacked, idle XCDs
  const int nxd_p = ctcn_opt_handoff() ? ctcn_device_xcds() : 1;
  // time chunks of the pipelined projection (plan_projection_pipeline, above): a fitting even count, or 8 when none fits (then the pipeline
  // is used only with option "fwd_pipe_any_chunking")
  int NCHUNK = ov.xcd_allow && nxd_p > 1 ? plan_projection_pipeline(T, B, I, H, dirs, G, nxd_p, ctcn_device_cus(), ov.xcd_allow, ctcn_get_option("fwd_pipe_min_input")) : 0;
  const bool chunking_fits = NCHUNK > 0;
  if (!chunking_fits) NCHUNK = 8;
  const int chunk_T = ceil_div(T, NCHUNK);
  const bool piped = !proj_done && ov.stream && ov.event && ov.ws && ov.xcd_allow != 0 && dirs == 2 && w_ih1 == w_ih0 + (size_t)GH * I &&
                     ctcn_opt_rnn_persistent() && ctcn_get_option("rnn_fwd_tagged") && precision == 1 && cell != CTCN_CELL_TANH && H % 32 == 0 &&
                     H / 32 <= 24 && nxd_p > 1 && T >= 4 * NCHUNK && chunk_T * (NCHUNK - 1) < T &&
                     // only with such a chunking (round 4): the pipeline runs neck and neck with the recurrence where it fits (cfg2: a pair of
                     // 2 x 2 560 rows takes the side stream ~130 us, the recurrence consumes it in 128: 13.64 -> 13.21 ms per step), and
                     // falls behind where it does not -- cfg3 (400 recurrent steps: 8 chunks of 1 600 rows on the 128-row tiles, the fixed cost
                     // of a pair spread over 50 steps instead of 80) ran 8.51 ms per step with it and 7.87 without.  Option
                     // "fwd_pipe_any_chunking" = 1 lifts the restriction (the tests that hold the pipeline to the inline order use it)
                     (chunking_fits || ctcn_get_option("fwd_pipe_any_chunking") != 0) &&
                     // the LAST chunk holds at least two frames: the RSV prologue fetches the pre-activations of steps 0 and 1 without looking
                     // at the chunk counter, and step 1 of the reverse direction is frame T - 2 -- with a one-frame last chunk (8 chunks:
                     // T = 36 / 43 / 50 / 57) that frame belongs to pair 1, which the side stream writes after the launch
                     T - chunk_T * (NCHUNK - 1) >= 2 &&
                     (size_t)T * B * dirs * GH * sizeof(float) < ((size_t)1 << 32) &&
                     // the side GEMMs' workgroups that land on a recurrence XCD must be able to START there (to exit at once): the
                     // 1024-thread workgroups of the recurrence leave no room on their own CUs, so some CUs of the XCD must stay free --
                     // otherwise the GEMM cannot finish before the recurrence does, which is waiting for it (H = 512: 32 of 32 CUs).
                     // (Four is the deadlock guard, for any chunking; the plan asks for eight.)
                     ceil_div(dirs * ceil_div(B, 16), nxd_p) * (H / 16) + 4 <= ctcn_device_cus() / nxd_p;
  GemmPlanes pl_main, pl_side;                           // what this call's GEMMs left in the main / the side workspace (operand planes reused within the call)
  auto project_chunk = [&](int c, void *wsp, size_t wsb, void *strm, unsigned allow, GemmPlanes &pl) -> int {
    const int t0 = c * chunk_T, t1 = std::min(T, t0 + chunk_T);
    return ctcn_gemm_on_xcds(0, 1, (t1 - t0) * B, 2 * GH, I, x + (size_t)t0 * B * I, I, w_ih0, I, gates + (size_t)t0 * B * 2 * GH, 2 * GH, 0.0f, precision,
                             wsp, wsb, strm, allow, &pl);
  };
  if (getenv("CTCN_LOG_PIPE")) fprintf(stderr, "libctcn: rnn_fwd T=%d B=%d I=%d H=%d: projection pipeline %s (chunks %d x %d steps, fits=%d)\n", T, B, I, H, piped ? "ON" : "off", NCHUNK, chunk_T, (int)chunking_fits);
  // (Round 5, tools/fwd_interference_probe.sh: layers 1-3 of cfg2 -- pipelined -- run 1 310 us against the 1 190 of layer 0, which has no pipeline.
  // That is not the recurrence waiting for a chunk: with TWO pairs projected here instead of one -- 119 us more head start, the last side pair
  // then ready before it is needed by the model 155 us per side pair / 119 us per pair consumed -- the step is 0.16 ms SLOWER (13.20 -> 13.36,
  // three pairs: 13.78): the extra pair costs its 75 us here and takes only ~25 off the recurrence.  The slowdown goes with the TIME the bf16x3
  // chunk GEMMs run next to the recurrence, ~30 us per pair, and is gone with the single-product tiles of option "gemm_bf16_single" (1 186-
  // 1 190 us for all four layers) although they still run there for 70 % of the time: the matrix pipes' power, not the counter.)
  // ORDER of the projection's row blocks (option "rnn_proj_order", default 0 = one product over ascending time; 1 = [T/2, T) then [0, T/2), pipelined:
  // the last chunk before the first).  Round 6 built the second order while it took the cfg4 divergence for stale reads of the rows a GEMM wrote last;
  // both orders deviated alike in the after-suite A/Bs (the cause: the parked tiles of rnn_fwd_tagged, above).  Same products, bit-identical results.
  const bool safe_order = ctcn_get_option("rnn_proj_order") != 0 && T >= 16;
  if (piped) {
    int rc = project_chunk(safe_order ? NCHUNK - 1 : 0, ws, ws_bytes, stream, 0, pl_main);
    pl_main.same_b = true;                                // W_ih: split into planes once per stream (reused when the chunk has the same size)
    if (!rc) rc = project_chunk(safe_order ? 0 : NCHUNK - 1, ws, ws_bytes, stream, 0, pl_main);
    if (rc) return rc;
    proj_done = true;
  }
  const int nblk_rows = safe_order ? 2 : 1;              // row blocks of the un-pipelined projection: [T/2, T) then [0, T/2), or all rows at once
  auto block_rows = [&](int blk, int &t0, int &t1) {
    if (nblk_rows == 1) { t0 = 0; t1 = T; }
    else if (blk == 0) { t0 = T / 2; t1 = T; }
    else { t0 = 0; t1 = T / 2; }
  };
  if (!proj_done && dirs == 2) {
    const size_t wcat_bytes = align_up((size_t)2 * GH * I * sizeof(float), 256);
    if (w_ih1 == w_ih0 + (size_t)GH * I) {      // already one (2*GH, I) matrix (optim.FlatAdam places them so): no stacking copies
      for (int blk = 0; blk < nblk_rows; ++blk) {
        int t0, t1;
        block_rows(blk, t0, t1);
        if (blk > 0) pl_main.same_b = true;
        int rc = ctcn_gemm_on_xcds(0, 1, (t1 - t0) * B, 2 * GH, I, x + (size_t)t0 * B * I, I, w_ih0, I, gates + (size_t)t0 * B * 2 * GH, 2 * GH, 0.0f, precision,
                                   ws, ws_bytes, stream, 0u, &pl_main);
        if (rc) return rc;
      }
      proj_done = true;
    } else if (ws && ws_bytes > wcat_bytes + ((size_t)64 << 20)) {
      float *wcat = (float *)ws;
      CTCN_HIP(hipMemcpyAsync(wcat, w_ih0, (size_t)GH * I * sizeof(float), hipMemcpyDeviceToDevice, st));
      CTCN_HIP(hipMemcpyAsync(wcat + (size_t)GH * I, w_ih1, (size_t)GH * I * sizeof(float), hipMemcpyDeviceToDevice, st));
      GemmPlanes pl_cat;
      for (int blk = 0; blk < nblk_rows; ++blk) {
        int t0, t1;
        block_rows(blk, t0, t1);
        if (blk > 0) pl_cat.same_b = true;
        int rc = ctcn_gemm_on_xcds(0, 1, (t1 - t0) * B, 2 * GH, I, x + (size_t)t0 * B * I, I, wcat, I, gates + (size_t)t0 * B * 2 * GH, 2 * GH, 0.0f, precision,
                                   (char *)ws + wcat_bytes, ws_bytes - wcat_bytes, stream, 0u, &pl_cat);
        if (rc) return rc;
      }
      proj_done = true;
    }
  }
  for (int d = 0; d < dirs && !proj_done; ++d) {
    // (one direction per product: a forward-only layer starts on the rows written first; the reverse direction of a two-product layer gets the blocks)
    const int nb_d = d == 1 ? nblk_rows : 1;
    for (int blk = 0; blk < nb_d; ++blk) {
      int t0 = 0, t1 = T;
      if (nb_d > 1) block_rows(blk, t0, t1);
      pl_main.same_a = false; pl_main.same_b = blk > 0;
      int rc = ctcn_gemm_on_xcds(0, 1, (t1 - t0) * B, GH, I, x + (size_t)t0 * B * I, I, w_ih[d], I, gates + (size_t)t0 * B * dirs * GH + (size_t)d * GH, dirs * GH, 0.0f,
                                 precision, ws, ws_bytes, stream, 0u, &pl_main);
      if (rc) return rc;
    }
  }
  RnnArgs a;
  a.cell = cell; a.T = T; a.B = B; a.H = H; a.D = dirs; a.G = G; a.step = 0;
  a.w0 = w_hh0; a.w1 = w_hh1; a.y = y; a.gates = gates; a.aux = aux; a.dy = nullptr; a.state = nullptr;
  const int HS = cell == CTCN_CELL_TANH ? 16 : 4;
  const int MT = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
  dim3 grid(ceil_div(H, HS), dirs, ceil_div(B, 16 * MT));
  // the persistent kernels address each tensor through one 32-bit-offset buffer resource
  const bool fits32 = (size_t)T * B * dirs * GH * sizeof(float) < ((size_t)1 << 32);
  if (ctcn_opt_rnn_persistent() && T > 1 && H % 4 == 0 && fits32) {
    // persistent recurrence: K = H split over 4 waves x 4 k-lanes x KQ4 float4 (one super-chunk)
    int kq = ceil_div(H, 64);
    if (kq == 7) kq = 8;
    const int nbt = ceil_div(B, 16), groups = dirs * nbt;
    const size_t lds = 0;
    // mode 1 = XCD-local (each group's slices on one XCD, hand-off through its L2), mode 0 = device scope
    // candidates in order of preference: XCD-local hand-off (each group's slices on one XCD, through its L2) with
    // 8 / 12 / 16 / 4 hidden units per workgroup -- fewer units = less matmul per step, but the group must stay
    // co-resident on one XCD -- then the device-scope hand-off.  The first one that is co-resident runs.
    struct Cand { int mode, hsu, nbig, hsu_small; };
    Cand cands[8];
    int nc = 0;
    const int nxd = ctcn_opt_handoff() ? ctcn_device_xcds() : 1;
    if (cell == CTCN_CELL_TANH) {
      if (nxd > 1) cands[nc++] = {1, 16, 0, 0};
      cands[nc++] = {0, 16, 0, 0};
    } else {
      if (nxd > 1) {
        // mixed slices: a 12-unit and b 4-unit workgroups, a + b = CUs of one XCD, 12 a + 4 b = H
        const int cpx = ctcn_device_cus() / nxd, nbig = (H - 4 * cpx) / 8;
        if (ctcn_get_option("rnn_mixed_slices") && H % 8 == 0 && (H - 4 * cpx) % 8 == 0 && nbig > 0 && nbig <= cpx && ceil_div(dirs * ceil_div(B, 16), nxd) == 1)
          cands[nc++] = {1, 12, nbig, 4};
        if (H % 8 == 0) cands[nc++] = {1, 8, 0, 0};
        cands[nc++] = {1, 12, 0, 0}; cands[nc++] = {1, 16, 0, 0}; cands[nc++] = {1, 4, 0, 0};
      }
      cands[nc++] = {0, H % 8 == 0 ? 8 : 4, 0, 0};
    }
    const char *why = kq > 8 ? "hidden size above 512" : "no candidate geometry is co-resident on this device";
    // first choice: the tagged gather (rnn_fwd_tagged): 16-unit slices, 1024-thread workgroups, XCD-local, precision 1, LSTM / GRU
    if (ctcn_get_option("rnn_fwd_tagged") && precision == 1 && cell != CTCN_CELL_TANH && H % 32 == 0 && H / 32 <= 24 && nxd > 1) {
      const int nsl = H / 16, wpx = ceil_div(groups, nxd) * nsl;
      const size_t hx_bytes = align_up((size_t)2 * dirs * nbt * (H / 32) * 2048, 256), fl_bytes = 512;
      if (ws && ws_bytes >= hx_bytes + fl_bytes + 512) {
        PersistArgs pa = {};
        pa.a = a;
        char *tail = (char *)ws + ((ws_bytes - hx_bytes - fl_bytes) & ~(size_t)255);
        pa.hx = (float *)tail;
        pa.flags = (unsigned *)(tail + hx_bytes);
        pa.status = status_word;
        pa.spin_limit = SPIN_LIMIT;
        pa.local = 1; pa.nx = nxd; pa.xperm = xcd_order_for(nxd, groups); pa.nsl = nsl; pa.nbt = nbt; pa.hsu = 16; pa.nbig = 0; pa.hsu_small = 0; pa.wpx = wpx;
        // "fwd_rsv_lds": 0 off, 1 on, 2 (default) where it measured faster: more than 24 slices (cfg4, H = 512: 2.30 -> 2.24 us per step, 53.9 -> 53.2 ms
        // per step; cfg2 1.60 -> 1.70, ref_yaml 1.51 -> 1.59: there the waterfall-paced dword traffic of the item waves is the better neighbour of
        // the polls)
        const int rsv_opt = ctcn_get_option("fwd_rsv_lds");
        const bool rsv = (rsv_opt == 1 || (rsv_opt == 2 && nsl > 24)) && ((uintptr_t)gates | (uintptr_t)aux | (uintptr_t)y | (uintptr_t)(drop_pending ? call.y_drop : nullptr)) % 16 == 0;
        pa.poll_depth = 1 | (rsv ? 256 : 0); pa.tagmode = 1; pa.poll_delay = ctcn_get_option("tag_poll_delay");
        pa.tickets = (unsigned *)(tail + hx_bytes + fl_bytes - 256);
#ifdef CTCN_PERSIST_STATS
        pa.stats = nullptr;
#endif
        pa.chunk_T = piped ? chunk_T : 0; pa.nchunk = NCHUNK; pa.chunk_ready = pa.flags;       // (first word of the flag area; tickets sit 256 B further)
        const bool fuse_drop = drop_pending && ctcn_get_option("rnn_fused_dropout") != 0;
        if (fuse_drop) {
          pa.ydrop = call.y_drop; pa.drop_p = call.drop_p; pa.drop_scale = 1.0f / (1.0f - call.drop_p);
          pa.drop_seed = call.drop_seed; pa.drop_off = call.drop_offset;
        }
        pa.slow = ctcn_get_option("rnn_slow_items"); pa.slow_x = ctcn_get_option("rnn_slow_exchange");
        CTCN_HIP(hipMemsetAsync(pa.hx, 0, hx_bytes + fl_bytes, st));            // zeroed tiles (tag 0) | chunk counter | role tickets
        if (piped) CTCN_HIP(hipEventRecord((hipEvent_t)ov.event, st));
        if (launch_fwd_tagged(ceil_div(H / 32, 12), dim3(nxd * (wpx + std::max(2, wpx / 8)), 1, 1), st, pa, wpx)) {
          CTCN_LAUNCH_CHECK();
          { g_last_kernel[0] = "rnn_fwd_tagged"; if (call.launched) *call.launched = "rnn_fwd_tagged"; }
          if (fuse_drop) drop_pending = false;
          if (piped) {          // the remaining chunk pairs, next to the recurrence on the XCDs it does not use
            hipStream_t sd = (hipStream_t)ov.stream;
            CTCN_HIP(hipStreamWaitEvent(sd, (hipEvent_t)ov.event, 0));
            for (int pr = 1; pr < NCHUNK / 2; ++pr) {
              // (W_ih is split into planes by the first of these calls only; splitting x of all six side chunks in that call as well --
              // one pass instead of six -- delays the first pair and measured 35 us per step slower)
              if (pr > 1) pl_side.same_b = true;
              int rc = project_chunk(pr, ov.ws, ov.ws_bytes, ov.stream, ov.xcd_allow, pl_side);
              pl_side.same_b = true;
              if (!rc) rc = project_chunk(NCHUNK - 1 - pr, ov.ws, ov.ws_bytes, ov.stream, ov.xcd_allow, pl_side);
              if (rc) {         // the recurrence is already queued and will wait for this counter: release it (its output is invalid, the
                                // caller sees rc) instead of letting it spin to the hand-off timeout and poison the status word
                hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, sd, pa.chunk_ready, (unsigned)NCHUNK);
                return rc;
              }
              hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, sd, pa.chunk_ready, (unsigned)pr);
            }
            CTCN_LAUNCH_CHECK();
          }
          return CTCN_OK;
        }
        if (piped)              // not co-resident after all: finish the projection here, then the other kernels take over
          for (int c = 1; c < NCHUNK - 1; ++c) { const int rc = project_chunk(c, ws, ws_bytes, stream, 0, pl_main); if (rc) return rc; }
      }
    }
    for (int ci = 0; ci < nc && kq <= 8; ++ci) {
      const int mode = cands[ci].mode, HSU = cands[ci].hsu;
      const int nx = mode ? nxd : 1;
      const int NT = cell == CTCN_CELL_TANH ? 1 : ceil_div(HSU, 4);
      const int nbig = cands[ci].nbig, hsu_small = cands[ci].hsu_small;
      const int nsl = nbig ? nbig + (H - nbig * HSU) / hsu_small : ceil_div(H, HSU);
      const int wpx = ceil_div(groups, nx) * nsl;
      const dim3 pgrid = mode ? dim3(nx * (wpx + std::max(2, wpx / 8)), 1, 1) : dim3(nsl, dirs, nbt);
      const int prec = precision == 1 && HSU % 4 == 0 && H % 8 == 0 ? 1 : 0;       // bf16x3 recurrent matmul
      const size_t hx_bytes = align_up((size_t)2 * dirs * nbt * (prec ? ceil_div(H, 32) * 512 : ceil_div(H, 16) * 256) * sizeof(float), 256);
      const size_t fl_bytes = align_up((size_t)2 * dirs * nbt * nsl * sizeof(unsigned), 256) + 256;   // + role tickets
      if (!ws || ws_bytes < hx_bytes + fl_bytes + 512) { why = "workspace too small for the hand-off tiles"; break; }
      PersistArgs pa = {};
      pa.a = a;
      char *tail = (char *)ws + ((ws_bytes - hx_bytes - fl_bytes) & ~(size_t)255);
      pa.hx = (float *)tail;
      pa.flags = (unsigned *)(tail + hx_bytes);
      pa.status = status_word;
      pa.spin_limit = SPIN_LIMIT;
      pa.local = mode; pa.nx = nx; pa.xperm = mode ? xcd_order_for(nx, groups) : 0; pa.nsl = nsl; pa.nbt = nbt; pa.hsu = HSU; pa.wpx = wpx; pa.poll_depth = ctcn_opt_poll_depth(); pa.tagmode = 0;
      pa.nbig = nbig; pa.hsu_small = hsu_small;
      pa.tickets = (unsigned *)(tail + hx_bytes + fl_bytes - 256);
#ifdef CTCN_PERSIST_STATS
      pa.stats = nullptr;
#endif
      if (prec) CTCN_HIP(hipMemsetAsync(pa.hx, 0, hx_bytes + fl_bytes, st));      // hand-off tiles | flags | tickets are contiguous
      else CTCN_HIP(hipMemsetAsync(pa.flags, 0, fl_bytes, st));
      if (launch_fwd_persist(prec, NT, kq, pgrid, lds, st, pa, wpx)) {
        CTCN_LAUNCH_CHECK();
        { g_last_kernel[0] = "rnn_fwd_persist"; if (call.launched) *call.launched = "rnn_fwd_persist"; }
        return CTCN_OK;
      }
    }
    log_fallback("ctcn_rnn_fwd", T, B, H, dirs, why);
  } else if (ctcn_opt_rnn_persistent() && T > 1) {
    log_fallback("ctcn_rnn_fwd", T, B, H, dirs, "a reserve tensor of 4 GB or more");
  }
  const int kq4 = pick_kq4(H, 4, MT, 20);
  g_last_kernel[0] = "rnn_fwd_step";
  if (call.launched) *call.launched = "rnn_fwd_step";
  for (int s = 0; s < T; ++s) {
    a.step = s;
    if (MT == 1) launch_fwd<1>(kq4, grid, st, a);
    else if (MT == 2) launch_fwd<2>(kq4, grid, st, a);
    else launch_fwd<4>(kq4, grid, st, a);
  }
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_dropout(const float *x, float *y, size_t n, float p, uint64_t seed, uint64_t offset, void *stream);
extern "C" int ctcn_rnn_fwd_ex(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                               const float *w_ih1, const float *w_hh1, float *y, float *gates, float *aux, int precision, void *ws, size_t ws_bytes,
                               void *stream, const ctcn_rnn_call *call) {
  const ctcn_rnn_call &c = call ? *call : k_plain_call;
  CTCN_REQUIRE(!c.y_drop || (c.drop_p >= 0.0f && c.drop_p < 1.0f), "ctcn_rnn_fwd_ex: dropout p=%f outside [0,1)", (double)c.drop_p);
  bool drop_pending = c.y_drop != nullptr;
  const int rc = rnn_fwd_impl(cell, T, B, I, H, dirs, x, w_ih0, w_hh0, w_ih1, w_hh1, y, gates, aux, precision, ws, ws_bytes, stream, c, drop_pending);
  if (rc) return rc;
  // the dropped output was not stored by the recurrence (another kernel than the tagged gather ran): the dropout pass, the same values
  return drop_pending ? ctcn_dropout(y, c.y_drop, (size_t)T * B * dirs * H, c.drop_p, c.drop_seed, c.drop_offset, stream) : CTCN_OK;
}
extern "C" int ctcn_rnn_fwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                            const float *w_hh0, const float *w_ih1, const float *w_hh1, float *y, float *gates,
                            float *aux, int precision, void *ws, size_t ws_bytes, void *stream) {
  return ctcn_rnn_fwd_ex(cell, T, B, I, H, dirs, x, w_ih0, w_hh0, w_ih1, w_hh1, y, gates, aux, precision, ws, ws_bytes, stream, nullptr);
}
extern "C" int ctcn_rnn_fwd_dropout(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                                    const float *w_ih1, const float *w_hh1, float *y, float *gates, float *aux, float *y_drop, float p, uint64_t seed,
                                    uint64_t offset, int precision, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(y_drop && p >= 0.0f && p < 1.0f, "ctcn_rnn_fwd_dropout: y_drop NULL or p=%f outside [0,1)", (double)p);
  ctcn_rnn_call c = {};
  c.y_drop = y_drop; c.drop_p = p; c.drop_seed = seed; c.drop_offset = offset;
  return ctcn_rnn_fwd_ex(cell, T, B, I, H, dirs, x, w_ih0, w_hh0, w_ih1, w_hh1, y, gates, aux, precision, ws, ws_bytes, stream, &c);
}

// The time-parallel GEMMs of the backward pass over the d(pre-activation) slab `gates` (and `aux` for the GRU n-gate):
// dx = da * W_ih (when dx), and -- when `weights` -- dW_ih = da^T x, dW_hh = da^T h_prev.  xcd_allow != 0 keeps the
// bf16x3 GEMM workgroups on those XCDs (side stream next to a persistent recurrence).
static int rnn_bwd_gemms(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_ih1,
                         const float *y, const float *gates, const float *aux, float *dx, float *dw_ih0, float *dw_hh0, float *dw_ih1,
                         float *dw_hh1, float beta_w, int precision, bool weights, unsigned xcd_allow, void *ws, size_t ws_bytes,
                         void *stream) {
  const int G = gates_of(cell), GH = G * H, TB = T * B;
  const float *w_ih[2] = {w_ih0, w_ih1};
  float *dw_ih[2] = {dw_ih0, dw_ih1};
  float *dw_hh[2] = {dw_hh0, dw_hh1};
  // (dx pipelined with the recurrence -- write-through d(pre-activation) stores, a counter per (direction, time chunk), one-lane
  // waiters + chunk GEMMs on a third stream on the idle XCDs -- was built twice in round 2 and removed twice: with the weight
  // gradients on the TN tile the idle XCDs hold ~1.0 ms of weight GEMMs per 1.44 ms recurrence, the 16 chunk products add ~0.7 ms
  // there, the recurrence itself slows to 1.67 ms next to them, and the step goes from 14.0 to 14.9 ms at cfg2 (57.6 -> 57.1 at cfg4).
  // The same machinery for the BOTTOM layer's weight gradients -- the time chunks are the natural split-K partials of dW, so only the
  // last pair would be left for the ~320 us tail after the last recurrence -- loses as well: 16 x 7 small launches on a third stream
  // finish ~1 ms after the recurrence (which slows from 1.48 to 1.57 ms next to them): 13.8 -> 14.3 ms.  The write-through stores and
  // the counters themselves cost 12 us per layer.)
  // (Round 3, after the TN tile left the idle XCDs empty for the last ~0.5 ms of every backward recurrence: a TIMING-ONLY experiment -- three
  // quarters of the dx product on a third stream on the idle XCDs, started by a clock-bounded sleeper kernel 0.7-1.0 ms into the recurrence,
  // the remaining quarter behind the recurrence -- bounds what the real thing (progress counters, L2 write-back per chunk, wait kernels) could
  // give: 13.39 -> 13.20-13.25 ms per cfg2 step, 1.2-1.4 %.  Not built a third time.)
  // dx = [da_fwd | da_rev] [W_ih_fwd ; W_ih_rev]: the reserve already holds both directions side by side (row = dirs*GH floats),
  // so with the two weight matrices stacked in the workspace one K = 2*GH product replaces two K = GH products and the
  // read-modify-write of dx between them (65 MB each way at cfg2)
  GemmPlanes pl;                               // operand planes this call's GEMMs leave in the workspace (da^T is shared by dW_ih and dW_hh)
  bool dx_done = false;
  if (dx && dirs == 2) {
    const size_t wcat_bytes = align_up((size_t)2 * GH * I * sizeof(float), 256);
    if (w_ih1 == w_ih0 + (size_t)GH * I) {      // the two W_ih are already stacked in memory
      int rc = ctcn_gemm_on_xcds(0, 0, TB, I, 2 * GH, gates, 2 * GH, w_ih0, I, dx, I, 0.0f, precision, ws, ws_bytes, stream, xcd_allow);
      if (rc) return rc;
      dx_done = true;
    } else if (ws && ws_bytes > wcat_bytes + ((size_t)64 << 20)) {
      float *wcat = (float *)ws;
      CTCN_HIP(hipMemcpyAsync(wcat, w_ih0, (size_t)GH * I * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
      CTCN_HIP(hipMemcpyAsync(wcat + (size_t)GH * I, w_ih1, (size_t)GH * I * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
      int rc = ctcn_gemm_on_xcds(0, 0, TB, I, 2 * GH, gates, 2 * GH, wcat, I, dx, I, 0.0f, precision, (char *)ws + wcat_bytes, ws_bytes - wcat_bytes,
                                 stream, xcd_allow);
      if (rc) return rc;
      dx_done = true;
    }
  }
  for (int d = 0; d < dirs; ++d) {
    const float *da = gates + (size_t)d * GH;
    const int ldg = dirs * GH;
    int rc;
    if (dx && !dx_done) {
      rc = ctcn_gemm_on_xcds(0, 0, TB, I, GH, da, ldg, w_ih[d], I, dx, I, d == 0 ? 0.0f : 1.0f, precision, ws, ws_bytes, stream, xcd_allow);
      if (rc) return rc;
    }
    if (!weights || !dw_ih[d]) continue;
    rc = ctcn_gemm_on_xcds(1, 0, GH, I, TB, da, ldg, x, I, dw_ih[d], I, beta_w, precision, ws, ws_bytes, stream, xcd_allow, &pl);
    if (rc) return rc;
    // dW_hh = sum_t dgh_t^T h_prev(t);  h_prev(t) = y[t-1] (fwd) / y[t+1] (reverse), zero at the sequence start
    const int Kh = (T - 1) * B;
    if (Kh <= 0) {
      if (beta_w == 0.0f) CTCN_HIP(hipMemsetAsync(dw_hh[d], 0, (size_t)GH * H * sizeof(float), (hipStream_t)stream));
      continue;
    }
    const size_t offA = d == 0 ? (size_t)B : 0, offY = d == 0 ? 0 : (size_t)B;
    const float *yh = y + (size_t)d * H + offY * dirs * H;
    if (cell == CTCN_CELL_GRU) {
      rc = ctcn_gemm_on_xcds(1, 0, 2 * H, H, Kh, da + offA * ldg, ldg, yh, dirs * H, dw_hh[d], H, beta_w, precision, ws, ws_bytes, stream, xcd_allow);
      if (rc) return rc;
      const float *dn = aux + (size_t)d * H + offA * dirs * H;
      rc = ctcn_gemm_on_xcds(1, 0, H, H, Kh, dn, dirs * H, yh, dirs * H, dw_hh[d] + (size_t)2 * H * H, H, beta_w, precision, ws, ws_bytes, stream, xcd_allow);
      if (rc) return rc;
    } else {
      // same A = da^T over the same K = T*B window as dW_ih above (its bf16 planes are reused); h_prev = y shifted by one timestep
      pl.same_a = true;
      rc = ctcn_gemm_shift_b(GH, H, TB, da, ldg, y + (size_t)d * H, dirs * H, dw_hh[d], H, beta_w, precision, ws, ws_bytes, stream, xcd_allow,
                             d == 0 ? B : -B, &pl);
      if (rc) return rc;
    }
  }
  return CTCN_OK;
}

// `call.prelaunch_event` (hipEvent_t): recorded on the stream right before the recurrence is launched, i.e. behind this call's own preparatory
// memsets / transposes: work that the host wants to run NEXT TO the recurrence on another stream waits for this event -- waiting for
// "everything before ctcn_rnn_bwd" instead lets it start early and delay those small kernels (measured: a 5 us memset took 116-122 us when
// the side stream's queue kernels were dispatched first).
extern "C" int ctcn_rnn_bwd_ex(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                               const float *w_hh0, const float *w_ih1, const float *w_hh1, const float *y,
                               float *gates, float *aux, const float *dy, float *dx, float *dw_ih0, float *dw_hh0,
                               float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                               size_t ws_bytes, void *stream, const ctcn_rnn_call *call_p) {
  const ctcn_rnn_call &call = call_p ? *call_p : k_plain_call;
  CTCN_REQUIRE(!call.dy_tmp || (call.drop_p >= 0.0f && call.drop_p < 1.0f), "ctcn_rnn_bwd_ex: dropout p=%f outside [0,1)", (double)call.drop_p);
  int *const status_word = call.status ? call.status : ctcn_status_word();
  void *prelaunch_event = call.prelaunch_event;
  auto record_prelaunch = [&](hipStream_t s_) {
    if (!prelaunch_event) return;
    (void)hipEventRecord((hipEvent_t)prelaunch_event, s_);
    prelaunch_event = nullptr;
  };
  CTCN_REQUIRE(cell >= 0 && cell <= 2, "ctcn_rnn_bwd: unknown cell %d", cell);
  CTCN_REQUIRE(T > 0 && B > 0 && I > 0 && H > 0 && (dirs == 1 || dirs == 2), "ctcn_rnn_bwd: bad dims");
  if (H % 4 != 0) { ctcn_set_error("ctcn_rnn_bwd: hidden size %d must be a multiple of 4", H); return CTCN_EUNSUPPORTED; }
  CTCN_REQUIRE(x && w_ih0 && w_hh0 && y && gates && dy && scratch, "ctcn_rnn_bwd: null pointer");
  CTCN_REQUIRE((dw_ih0 != nullptr) == (dw_hh0 != nullptr), "ctcn_rnn_bwd: dw_ih0 / dw_hh0 must both be given or both be NULL (weights deferred to ctcn_rnn_bwd_weights)");
  CTCN_REQUIRE(dirs == 1 || (w_ih1 && w_hh1 && (!dw_ih0 || (dw_ih1 && dw_hh1))), "ctcn_rnn_bwd: null pointer (reverse direction)");
  CTCN_REQUIRE(cell == CTCN_CELL_TANH || aux, "ctcn_rnn_bwd: aux reserve required for LSTM/GRU");
  CTCN_REQUIRE((uintptr_t)gates % 16 == 0 && (uintptr_t)scratch % 16 == 0 && (cell == CTCN_CELL_TANH || (uintptr_t)aux % 16 == 0),
               "ctcn_rnn_bwd: gates / aux / scratch must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int G = gates_of(cell);
  const int GH = G * H;
  struct { float p; unsigned long long seed, off; float *tmp; } const bd = {call.drop_p, (unsigned long long)call.drop_seed, (unsigned long long)call.drop_offset, call.dy_tmp};
  bool dy_dropped = call.dy_tmp == nullptr;      // true once dy needs no mask any more: no request, or the dropout kernel has produced bd.tmp
  const float *w_hh[2] = {w_hh0, w_hh1};
  float *whhT = (float *)scratch;
  float *state = (float *)((char *)scratch + align_up((size_t)dirs * GH * H * sizeof(float), 256));
  if (dirs == 2) {
    if (int rc = ctcn_transpose01_pair(w_hh[0], w_hh[1], whhT, whhT + (size_t)GH * H, GH, H, 1, stream)) return rc;
  } else if (int rc = ctcn_transpose01(w_hh[0], whhT, GH, H, 1, stream)) {
    return rc;
  }
  RnnArgs a;
  a.cell = cell; a.T = T; a.B = B; a.H = H; a.D = dirs; a.G = G; a.step = 0;
  a.w0 = whhT; a.w1 = whhT + (size_t)GH * H; a.y = const_cast<float *>(y); a.gates = gates; a.aux = aux; a.dy = dy;
  a.state = state;
  dim3 grid(ceil_div(H, 16), dirs, ceil_div(B, 16));
  bool done = false;
  const bool fits32 = (size_t)T * B * dirs * GH * sizeof(float) < ((size_t)1 << 32);   // one 32-bit-offset resource per tensor
  // (a reserve of 4 GB or more: only rnn_bwd_scatter2 -- 64-bit reserve addresses -- may take it; the other persistent kernels address a
  // reserve through one 32-bit buffer resource)
  if (ctcn_opt_rnn_persistent() && T > 1) {
    int kq = ceil_div(GH, 256);
    if (kq == 7) kq = 8;
    const int nbt = grid.z, nsl = grid.x, groups = dirs * nbt;
    const int prec = precision == 1 && H % 8 == 0 ? 1 : 0;           // bf16x3 recurrent matmul
    // scatter formulation (default): partial dh tiles travel, 1 KB per (owner, source) pair; gather formulation: the da tile
    // measured: scatter wins at H = 320 (2.46 vs 2.68 us per step), ties at H = 128, loses at H = 512 (nsl^2 KB of partial tiles
    // per group and step) and at precision 0 (f32 MFMA: 32 cycles x 16 per tile on 4 waves per SIMD)
    // item-wave gather (rnn_bwd_scatter2, option "bwd_item_gather"): tagged hand-off only, up to 40 slices (H <= 640), 16-B aligned reserves
    // (gates / aux are checked above; dy, y and -- when the unfused dropout pass feeds the kernel -- dy_tmp here: 16-B LDS DMA and f32x4 stores)
    // "bwd_item_gather": 0 never, 1 (default) where it measured faster -- more than 20 slices, i.e. H > 320 (tools/mb_bwd2.hip: H = 384 1.58 vs
    // 1.63 us per step, H = 512 2.44 vs 3.38 for the gather formulation, the only other kernel there; H = 320 1.90 vs 1.88, H = 128 1.25 vs 1.19) --, 2 always
    const int ig = ctcn_get_option("bwd_item_gather");
    const bool gather2 = (ig == 2 || (ig >= 1 && (nsl > 20 || !fits32))) && nsl <= 40 && ctcn_opt_handoff_tags() && H % 4 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)y % 16 == 0 &&
                         (!call.dy_tmp || (uintptr_t)call.dy_tmp % 16 == 0);
    const bool scatter = ctcn_opt_bwd_scatter() && prec && nsl <= (gather2 ? 40 : 24);
    const int ntw = nsl <= 12 ? 1 : 2;                        // rnn_bwd_scatter: output tiles per scattering wave (12 of them; nsl <= 24)
    const size_t hx_bytes = scatter ? align_up((size_t)2 * dirs * nbt * nsl * nsl * 1024, 256)
                                    : align_up((size_t)2 * dirs * nbt * (prec ? ceil_div(GH, 32) * 512 : ceil_div(GH, 16) * 256) * sizeof(float), 256);
    const size_t fl_bytes = align_up((size_t)2 * dirs * nbt * nsl * (scatter ? nsl : 1) * sizeof(unsigned), 256) + 256;   // + role tickets
    const size_t lds = 0;
    for (int mode = ctcn_opt_handoff() ? 1 : 0; mode >= 0 && !done && (kq <= 8 || scatter) && (fits32 || (scatter && gather2)) && ws && ws_bytes >= hx_bytes + fl_bytes + 512; --mode) {
      const int nx = mode ? ctcn_device_xcds() : 1;
      if (mode && nx <= 1) continue;
      const int wpx = ceil_div(groups, nx) * nsl;
      PersistArgs pa = {};
      pa.a = a;
      char *tail = (char *)ws + ((ws_bytes - hx_bytes - fl_bytes) & ~(size_t)255);
      pa.hx = (float *)tail;
      pa.flags = (unsigned *)(tail + hx_bytes);
      pa.status = status_word;
      pa.spin_limit = SPIN_LIMIT;
      pa.local = mode; pa.nx = nx; pa.xperm = mode ? xcd_order_for(nx, groups) : 0; pa.nsl = nsl; pa.nbt = nbt; pa.hsu = 16; pa.nbig = 0; pa.hsu_small = 0; pa.wpx = wpx; pa.poll_depth = ctcn_opt_poll_depth(); pa.tagmode = 0;
      pa.tickets = (unsigned *)(tail + hx_bytes + fl_bytes - 256);
#ifdef CTCN_PERSIST_STATS
      pa.stats = nullptr;
#endif
      const dim3 pgrid = mode ? dim3(nx * (wpx + std::max(2, wpx / 8)), 1, 1) : grid;
      // flags + tickets, and the hand-off tiles where they start from zero (tags / partial sums): the two areas are contiguous -> one memset
      if (scatter) {
        pa.tagmode = ctcn_opt_handoff_tags();
        pa.slow = ctcn_get_option("rnn_slow_items"); pa.slow_x = ctcn_get_option("rnn_slow_exchange");
        if (pa.tagmode) CTCN_HIP(hipMemsetAsync(pa.hx, 0, hx_bytes + fl_bytes, st));
        else CTCN_HIP(hipMemsetAsync(pa.flags, 0, fl_bytes, st));
        if (!dy_dropped && ctcn_get_option("rnn_fused_dropout") != 0) {
          pa.drop_bwd = 1; pa.drop_p = bd.p; pa.drop_scale = 1.0f / (1.0f - bd.p); pa.drop_seed = bd.seed; pa.drop_off = bd.off;
        } else if (!dy_dropped) {
          if (int rc = ctcn_dropout(dy, bd.tmp, (size_t)T * B * dirs * H, bd.p, bd.seed, bd.off, stream)) return rc;
          a.dy = bd.tmp; pa.a.dy = bd.tmp; dy_dropped = true;
        }
        record_prelaunch(st);
        if (gather2) {
          pa.rsv_nt = ctcn_get_option("rnn_rsv_nt");
          pa.poll_delay = ctcn_get_option("bwd_poll_delay");
          if (pa.poll_delay < 0) pa.poll_delay = nsl <= 24 ? 16 : 24;        // auto: the exchange phase grows with the tiles per wave (cfg4: 2.69 -> 2.55 us per step)
          done = launch_bwd_scatter2(nsl, pgrid, st, pa, wpx);
          if (done) { g_last_kernel[1] = "rnn_bwd_scatter2"; if (call.launched) *call.launched = "rnn_bwd_scatter2"; }
        } else {
          done = launch_bwd_scatter(prec, ntw, pgrid, st, pa, wpx);
          if (done) { g_last_kernel[1] = "rnn_bwd_scatter"; if (call.launched) *call.launched = "rnn_bwd_scatter"; }
        }
      } else {
        if (!dy_dropped) {
          if (int rc = ctcn_dropout(dy, bd.tmp, (size_t)T * B * dirs * H, bd.p, bd.seed, bd.off, stream)) return rc;
          a.dy = bd.tmp; pa.a.dy = bd.tmp; dy_dropped = true;
        }
        if (prec) CTCN_HIP(hipMemsetAsync(pa.hx, 0, hx_bytes + fl_bytes, st));
        else CTCN_HIP(hipMemsetAsync(pa.flags, 0, fl_bytes, st));
        record_prelaunch(st);
        done = launch_bwd_persist(prec, kq, pgrid, lds, st, pa, wpx);
        if (done) { g_last_kernel[1] = "rnn_bwd_persist"; if (call.launched) *call.launched = "rnn_bwd_persist"; }
      }
    }
  }
  record_prelaunch(st);          // no-op if a persistent launch already consumed the event
  if (!done && ctcn_opt_rnn_persistent() && T > 1)
    log_fallback("ctcn_rnn_bwd", T, B, H, dirs, !fits32 ? "a reserve tensor of 4 GB or more"
                                                     : (ceil_div(GH, 256) > 8 ? "gate width G*H above 2048" : "not co-resident / workspace too small"));
  if (!done) {
    if (!dy_dropped) {
      if (int rc = ctcn_dropout(dy, bd.tmp, (size_t)T * B * dirs * H, bd.p, bd.seed, bd.off, stream)) return rc;
      a.dy = bd.tmp; dy_dropped = true;
    }
    // (the carried dc / dh*z of the per-timestep kernels lives in `state`; the persistent kernels keep it in registers)
    CTCN_HIP(hipMemsetAsync(state, 0, (size_t)B * dirs * H * sizeof(float), st));
    const int kq4 = pick_kq4(GH, 16, 1, 5);
    g_last_kernel[1] = "rnn_bwd_step";
    if (call.launched) *call.launched = "rnn_bwd_step";
    for (int s = 0; s < T; ++s) {
      a.step = s;
      launch_bwd(kq4, grid, st, a);
    }
  }
  CTCN_LAUNCH_CHECK();

  if (ctcn_opt_recurrence_only()) return CTCN_OK;
  // deferred GEMMs over the d(pre-activation) slab now held in `gates` (and `aux` for the GRU n-gate)
  return rnn_bwd_gemms(cell, T, B, I, H, dirs, x, w_ih0, w_ih1, y, gates, aux, dx, dw_ih0, dw_hh0, dw_ih1, dw_hh1, beta_w, precision,
                       dw_ih0 != nullptr, 0u, ws, ws_bytes, stream);
}

extern "C" int ctcn_rnn_bwd_weights(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *y, const float *gates,
                                    const float *aux, float *dw_ih0, float *dw_hh0, float *dw_ih1, float *dw_hh1, float beta_w,
                                    int precision, unsigned xcd_allow, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(cell >= 0 && cell <= 2, "ctcn_rnn_bwd_weights: unknown cell %d", cell);
  CTCN_REQUIRE(T > 0 && B > 0 && I > 0 && H > 0 && (dirs == 1 || dirs == 2), "ctcn_rnn_bwd_weights: bad dims");
  // a direction whose two gradient pointers are both NULL is skipped (the host runs the directions of the bottom layer on two streams)
  CTCN_REQUIRE(x && y && gates && (dw_ih0 != nullptr) == (dw_hh0 != nullptr) && (dirs == 1 || (dw_ih1 != nullptr) == (dw_hh1 != nullptr)) &&
                   (dw_ih0 || (dirs == 2 && dw_ih1)), "ctcn_rnn_bwd_weights: null pointer");
  CTCN_REQUIRE(cell != CTCN_CELL_GRU || aux, "ctcn_rnn_bwd_weights: aux (d of the GRU n-gate) required");
  if (ctcn_opt_recurrence_only()) return CTCN_OK;
  return rnn_bwd_gemms(cell, T, B, I, H, dirs, x, nullptr, nullptr, y, gates, aux, nullptr, dw_ih0, dw_hh0, dw_ih1, dw_hh1, beta_w,
                       precision, true, xcd_allow, ws, ws_bytes, stream);
}

extern "C" int ctcn_rnn_bwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                            const float *w_hh0, const float *w_ih1, const float *w_hh1, const float *y,
                            float *gates, float *aux, const float *dy, float *dx, float *dw_ih0, float *dw_hh0,
                            float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                            size_t ws_bytes, void *stream) {
  return ctcn_rnn_bwd_ex(cell, T, B, I, H, dirs, x, w_ih0, w_hh0, w_ih1, w_hh1, y, gates, aux, dy, dx, dw_ih0, dw_hh0, dw_ih1, dw_hh1, beta_w, precision,
                         scratch, ws, ws_bytes, stream, nullptr);
}

extern "C" int ctcn_rnn_bwd_dropout(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                                    const float *w_ih1, const float *w_hh1, const float *y, float *gates, float *aux, const float *dy, float *dx,
                                    float *dw_ih0, float *dw_hh0, float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                                    size_t ws_bytes, void *stream, float p, uint64_t seed, uint64_t offset, float *dy_tmp) {
  CTCN_REQUIRE(dy_tmp && p >= 0.0f && p < 1.0f, "ctcn_rnn_bwd_dropout: dy_tmp NULL or p=%f outside [0,1)", (double)p);
  ctcn_rnn_call c = {};
  c.dy_tmp = dy_tmp; c.drop_p = p; c.drop_seed = seed; c.drop_offset = offset;
  return ctcn_rnn_bwd_ex(cell, T, B, I, H, dirs, x, w_ih0, w_hh0, w_ih1, w_hh1, y, gates, aux, dy, dx, dw_ih0, dw_hh0, dw_ih1, dw_hh1, beta_w, precision,
                         scratch, ws, ws_bytes, stream, &c);
}
