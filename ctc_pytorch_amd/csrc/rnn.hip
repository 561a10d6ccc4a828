// rnn.hip -- bias-free (bi)directional LSTM / GRU / tanh-RNN layer, forward and backward, for gfx950.
//
// replaces: rnn_type(input_size, hidden_size, bidirectional, bias=False)(x) in BatchRNN.forward
// (reference timit/models/model_ctc.py:24-25,33; rnn_type chosen at timit/steps/train_ctc.py:20) and its
// autograd backward (train_ctc.py:63).  Arithmetic: SURVEY.md Appendix A.1/A.2/A.2b.
//
// Structure (per layer):
//   1. time-parallel input projection  Gx = X * W_ih^T  for all T*B rows: one MFMA GEMM per direction
//      (gemm.hip) written straight into the gate reserve (T,B,dirs,G*H).
//   2. the serial recurrence: one launch per timestep, both directions in the same launch
//      (blockIdx.y = direction).  A workgroup owns a slice of hidden units for all batch rows:
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

#include "common.h"

namespace {

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
    __syncthreads();
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
    __syncthreads();
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
      const float i_ = sigmoidf_(outs[bl][0 * 4 + jl] + pre[0]);
      const float f_ = sigmoidf_(outs[bl][1 * 4 + jl] + pre[1]);
      const float g_ = tanhf(outs[bl][2 * 4 + jl] + pre[2]);
      const float o_ = sigmoidf_(outs[bl][3 * 4 + jl] + pre[3]);
      const float c = f_ * prev + i_ * g_;
      gt[0 * H + j] = i_; gt[1 * H + j] = f_; gt[2 * H + j] = g_; gt[3 * H + j] = o_;
      p.aux[(row_t * D + d) * H + j] = c;
      yt[j] = o_ * tanhf(c);
    } else {
      const float hn = outs[bl][2 * 4 + jl];
      const float r_ = sigmoidf_(outs[bl][0 * 4 + jl] + pre[0]);
      const float z_ = sigmoidf_(outs[bl][1 * 4 + jl] + pre[1]);
      const float n_ = tanhf(pre[2] + r_ * hn);
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
      p.y[rt * D * H + d * H + j0 + jl2] = tanhf(outs[bl2][jl2] + p.gates[(rt * D + d) * (size_t)H + j0 + jl2]);
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
      const float tc = tanhf(e0);
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

}  // namespace

extern "C" size_t ctcn_rnn_scratch_bytes(int cell, int B, int H, int dirs) {
  const size_t G = gates_of(cell);
  return align_up((size_t)dirs * G * H * H * sizeof(float), 256) + align_up((size_t)B * dirs * H * sizeof(float), 256);
}

extern "C" int ctcn_rnn_fwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                            const float *w_hh0, const float *w_ih1, const float *w_hh1, float *y, float *gates,
                            float *aux, int precision, void *ws, size_t ws_bytes, void *stream) {
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
  for (int d = 0; d < dirs; ++d) {
    int rc = ctcn_gemm(0, 1, T * B, GH, I, x, I, w_ih[d], I, gates + (size_t)d * GH, dirs * GH, 0.0f, precision, ws,
                       ws_bytes, stream);
    if (rc) return rc;
  }
  RnnArgs a;
  a.cell = cell; a.T = T; a.B = B; a.H = H; a.D = dirs; a.G = G; a.step = 0;
  a.w0 = w_hh0; a.w1 = w_hh1; a.y = y; a.gates = gates; a.aux = aux; a.dy = nullptr; a.state = nullptr;
  const int HS = cell == CTCN_CELL_TANH ? 16 : 4;
  const int MT = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
  const int kq4 = pick_kq4(H, 4, MT, 20);
  dim3 grid(ceil_div(H, HS), dirs, ceil_div(B, 16 * MT));
  for (int s = 0; s < T; ++s) {
    a.step = s;
    if (MT == 1) launch_fwd<1>(kq4, grid, st, a);
    else if (MT == 2) launch_fwd<2>(kq4, grid, st, a);
    else launch_fwd<4>(kq4, grid, st, a);
  }
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_rnn_bwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                            const float *w_hh0, const float *w_ih1, const float *w_hh1, const float *y,
                            float *gates, float *aux, const float *dy, float *dx, float *dw_ih0, float *dw_hh0,
                            float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                            size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(cell >= 0 && cell <= 2, "ctcn_rnn_bwd: unknown cell %d", cell);
  CTCN_REQUIRE(T > 0 && B > 0 && I > 0 && H > 0 && (dirs == 1 || dirs == 2), "ctcn_rnn_bwd: bad dims");
  if (H % 4 != 0) { ctcn_set_error("ctcn_rnn_bwd: hidden size %d must be a multiple of 4", H); return CTCN_EUNSUPPORTED; }
  CTCN_REQUIRE(x && w_ih0 && w_hh0 && y && gates && dy && dw_ih0 && dw_hh0 && scratch, "ctcn_rnn_bwd: null pointer");
  CTCN_REQUIRE(dirs == 1 || (w_ih1 && w_hh1 && dw_ih1 && dw_hh1), "ctcn_rnn_bwd: null pointer (reverse direction)");
  CTCN_REQUIRE(cell == CTCN_CELL_TANH || aux, "ctcn_rnn_bwd: aux reserve required for LSTM/GRU");
  CTCN_REQUIRE((uintptr_t)gates % 16 == 0 && (uintptr_t)scratch % 16 == 0 && (cell == CTCN_CELL_TANH || (uintptr_t)aux % 16 == 0),
               "ctcn_rnn_bwd: gates / aux / scratch must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int G = gates_of(cell);
  const int GH = G * H;
  const float *w_ih[2] = {w_ih0, w_ih1};
  const float *w_hh[2] = {w_hh0, w_hh1};
  float *dw_ih[2] = {dw_ih0, dw_ih1};
  float *dw_hh[2] = {dw_hh0, dw_hh1};
  float *whhT = (float *)scratch;
  float *state = (float *)((char *)scratch + align_up((size_t)dirs * GH * H * sizeof(float), 256));
  for (int d = 0; d < dirs; ++d) {
    int rc = ctcn_transpose01(w_hh[d], whhT + (size_t)d * GH * H, GH, H, 1, stream);
    if (rc) return rc;
  }
  CTCN_HIP(hipMemsetAsync(state, 0, (size_t)B * dirs * H * sizeof(float), st));

  RnnArgs a;
  a.cell = cell; a.T = T; a.B = B; a.H = H; a.D = dirs; a.G = G; a.step = 0;
  a.w0 = whhT; a.w1 = whhT + (size_t)GH * H; a.y = const_cast<float *>(y); a.gates = gates; a.aux = aux; a.dy = dy;
  a.state = state;
  const int kq4 = pick_kq4(GH, 16, 1, 5);
  dim3 grid(ceil_div(H, 16), dirs, ceil_div(B, 16));
  for (int s = 0; s < T; ++s) {
    a.step = s;
    launch_bwd(kq4, grid, st, a);
  }
  CTCN_LAUNCH_CHECK();

  // deferred GEMMs over the d(pre-activation) slab now held in `gates` (and `aux` for the GRU n-gate)
  const int TB = T * B;
  for (int d = 0; d < dirs; ++d) {
    const float *da = gates + (size_t)d * GH;
    const int ldg = dirs * GH;
    int rc;
    if (dx) {
      rc = ctcn_gemm(0, 0, TB, I, GH, da, ldg, w_ih[d], I, dx, I, d == 0 ? 0.0f : 1.0f, precision, ws, ws_bytes, stream);
      if (rc) return rc;
    }
    rc = ctcn_gemm(1, 0, GH, I, TB, da, ldg, x, I, dw_ih[d], I, beta_w, precision, ws, ws_bytes, stream);
    if (rc) return rc;
    // dW_hh = sum_t dgh_t^T h_prev(t);  h_prev(t) = y[t-1] (fwd) / y[t+1] (reverse), zero at the sequence start
    const int Kh = (T - 1) * B;
    const size_t offA = d == 0 ? (size_t)B : 0, offY = d == 0 ? 0 : (size_t)B;
    const float *yh = y + (size_t)d * H + offY * dirs * H;
    if (cell == CTCN_CELL_GRU) {
      rc = ctcn_gemm(1, 0, 2 * H, H, Kh, da + offA * ldg, ldg, yh, dirs * H, dw_hh[d], H, beta_w, precision, ws, ws_bytes, stream);
      if (rc) return rc;
      const float *dn = aux + (size_t)d * H + offA * dirs * H;
      rc = ctcn_gemm(1, 0, H, H, Kh, dn, dirs * H, yh, dirs * H, dw_hh[d] + (size_t)2 * H * H, H, beta_w, precision, ws, ws_bytes, stream);
      if (rc) return rc;
    } else {
      rc = ctcn_gemm(1, 0, GH, H, Kh, da + offA * ldg, ldg, yh, dirs * H, dw_hh[d], H, beta_w, precision, ws, ws_bytes, stream);
      if (rc) return rc;
    }
  }
  return CTCN_OK;
}
