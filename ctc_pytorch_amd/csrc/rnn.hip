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
// += A[rows, K] * Bm[16 cols, K]^T, with K split over `ngroups` (wave, k-lane) groups.
// A element (row,k) lives at a1[row*ld1 + k] for k < ksplit, else a2[row*ld2 + k - ksplit].
template <int MT, int KQ4>
__device__ __forceinline__ void rec_mm(const float *__restrict__ a1, int ld1, int ksplit, const float *__restrict__ a2,
                                       int ld2, int rows, const float *__restrict__ brow, bool bvalid, int K,
                                       int ngroups, int g, int r, f32x4 (&acc)[MT]) {
  constexpr int kspan = 4 * KQ4;
  const int sc_stride = ngroups * kspan;
  const int nsc = (K + sc_stride - 1) / sc_stride;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sc = 0; sc < nsc; ++sc) {
    const int kb = sc * sc_stride + g * kspan;
    float4 av[MT][KQ4], bv[KQ4];
#pragma unroll
    for (int s = 0; s < KQ4; ++s) {
      const int k = kb + 4 * s;
      const bool kin = k < K;
      bv[s] = (bvalid && kin) ? *reinterpret_cast<const float4 *>(brow + k) : zero;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + r;
        const float *p = (k < ksplit) ? a1 + (size_t)row * ld1 + k : a2 + (size_t)row * ld2 + (k - ksplit);
        av[mt][s] = (kin && row < rows) ? *reinterpret_cast<const float4 *>(p) : zero;
      }
    }
#pragma unroll
    for (int s = 0; s < KQ4; ++s) {
      const float b4[4] = {bv[s].x, bv[s].y, bv[s].z, bv[s].w};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float a4[4] = {av[mt][s].x, av[mt][s].y, av[mt][s].z, av[mt][s].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c], b4[c], acc[mt], 0, 0, 0);
      }
    }
  }
}

// Combine the NW per-wave partial tiles through LDS (fixed summation order -> deterministic), at most
// PW waves per pass so the staging buffer stays <= 32 KiB.  Tile element (row, col) -> outs[row][col].
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
// forward step
// ------------------------------------------------------------------------------------------------
template <int MT, int KQ4>
__global__ __launch_bounds__(256) void rnn_fwd_step(RnnArgs p) {
  constexpr int NW = 4;
  __shared__ float red[NW * MT * 256];
  __shared__ float outs[MT * 16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4, g = wave * 4 + q;
  const int d = blockIdx.y, b0 = blockIdx.z * 64;
  const int H = p.H, G = p.G, D = p.D, B = p.B;
  const int Bc = min(64, B - b0);
  const int t = d == 0 ? p.step : p.T - 1 - p.step;
  const int tp = p.step == 0 ? -1 : (d == 0 ? t - 1 : t + 1);
  const bool tanh_cell = p.cell == CTCN_CELL_TANH;
  const int HS = tanh_cell ? 16 : 4;
  const int j0 = blockIdx.x * HS;
  const int gate = tanh_cell ? 0 : (r >> 2), jj = tanh_cell ? r : (r & 3);
  const bool bvalid = gate < G && (j0 + jj) < H;
  const float *W = d == 0 ? p.w0 : p.w1;
  const float *brow = W + (size_t)(gate * H + j0 + jj) * H;

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tp >= 0) {
    const float *abase = p.y + ((size_t)tp * B + b0) * D * H + d * H;
    rec_mm<MT, KQ4>(abase, D * H, H, abase, D * H, Bc, brow, bvalid, H, 4 * NW, g, r, acc);
  }
  reduce_tiles<MT, NW, 4>(acc, red, outs, tid, 256);

  const int items = Bc * HS;
  for (int it = tid; it < items; it += 256) {
    const int bl = it / HS, jl = it - bl * HS;
    const int j = j0 + jl;
    if (j >= H) continue;
    const int b = b0 + bl;
    const size_t row_t = (size_t)t * B + b;
    float *gt = p.gates + (row_t * D + d) * (size_t)(G * H);
    float *yt = p.y + row_t * D * H + d * H;
    if (p.cell == CTCN_CELL_LSTM) {
      const float ai = outs[bl][0 * 4 + jl] + gt[0 * H + j];
      const float af = outs[bl][1 * 4 + jl] + gt[1 * H + j];
      const float ag = outs[bl][2 * 4 + jl] + gt[2 * H + j];
      const float ao = outs[bl][3 * 4 + jl] + gt[3 * H + j];
      const float i_ = sigmoidf_(ai), f_ = sigmoidf_(af), g_ = tanhf(ag), o_ = sigmoidf_(ao);
      const float cp = tp >= 0 ? p.aux[(((size_t)tp * B + b) * D + d) * H + j] : 0.0f;
      const float c = f_ * cp + i_ * g_;
      gt[0 * H + j] = i_; gt[1 * H + j] = f_; gt[2 * H + j] = g_; gt[3 * H + j] = o_;
      p.aux[(row_t * D + d) * H + j] = c;
      yt[j] = o_ * tanhf(c);
    } else if (p.cell == CTCN_CELL_GRU) {
      const float hn = outs[bl][2 * 4 + jl];
      const float r_ = sigmoidf_(outs[bl][0 * 4 + jl] + gt[0 * H + j]);
      const float z_ = sigmoidf_(outs[bl][1 * 4 + jl] + gt[1 * H + j]);
      const float n_ = tanhf(gt[2 * H + j] + r_ * hn);
      const float hp = tp >= 0 ? p.y[((size_t)tp * B + b) * D * H + d * H + j] : 0.0f;
      gt[0 * H + j] = r_; gt[1 * H + j] = z_; gt[2 * H + j] = n_;
      p.aux[(row_t * D + d) * H + j] = hn;
      yt[j] = (1.0f - z_) * n_ + z_ * hp;
    } else {
      yt[j] = tanhf(outs[bl][jl] + gt[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward step (processing order reversed: forward direction walks t = T-1..0)
// ------------------------------------------------------------------------------------------------
template <int MT, int KQ4>
__global__ __launch_bounds__(1024) void rnn_bwd_step(RnnArgs p) {
  constexpr int NW = 16, PW = 8;
  __shared__ float red[PW * MT * 256];
  __shared__ float outs[MT * 16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4, g = wave * 4 + q;
  const int d = blockIdx.y, b0 = blockIdx.z * 64;
  const int H = p.H, G = p.G, D = p.D, B = p.B, T = p.T;
  const int Bc = min(64, B - b0);
  const int GH = G * H;
  // forward direction: t = T-1-step, next-processed-before = t+1, prev-in-forward-order = t-1
  const int t = d == 0 ? T - 1 - p.step : p.step;
  const int tn = p.step == 0 ? -1 : (d == 0 ? t + 1 : t - 1);
  const int tp = d == 0 ? (t > 0 ? t - 1 : -1) : (t < T - 1 ? t + 1 : -1);
  const int j0 = blockIdx.x * 16;
  const bool bvalid = (j0 + r) < H;
  const float *WT = d == 0 ? p.w0 : p.w1;
  const float *brow = WT + (size_t)(j0 + r) * GH;

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tn >= 0) {
    const size_t rown = (size_t)tn * B + b0;
    const float *a1 = p.gates + (rown * D + d) * (size_t)GH;
    if (p.cell == CTCN_CELL_GRU) {
      const float *a2 = p.aux + (rown * D + d) * (size_t)H;   // d(W_hn h) = dan * r
      rec_mm<MT, KQ4>(a1, D * GH, 2 * H, a2, D * H, Bc, brow, bvalid, GH, 4 * NW, g, r, acc);
    } else {
      rec_mm<MT, KQ4>(a1, D * GH, GH, a1, D * GH, Bc, brow, bvalid, GH, 4 * NW, g, r, acc);
    }
  }
  reduce_tiles<MT, NW, PW>(acc, red, outs, tid, 1024);

  const int bl = tid >> 4, jl = tid & 15, j = j0 + jl;
  if (bl < Bc && j < H) {
    const int b = b0 + bl;
    const size_t row_t = (size_t)t * B + b;
    float *gt = p.gates + (row_t * D + d) * (size_t)GH;
    const float dyv = p.dy[row_t * D * H + d * H + j];
    float dh = dyv + outs[bl][jl];
    if (p.cell == CTCN_CELL_LSTM) {
      float *st = p.state + ((size_t)b * D + d) * H + j;
      const float i_ = gt[0 * H + j], f_ = gt[1 * H + j], g_ = gt[2 * H + j], o_ = gt[3 * H + j];
      const float c = p.aux[(row_t * D + d) * H + j];
      const float cp = tp >= 0 ? p.aux[(((size_t)tp * B + b) * D + d) * H + j] : 0.0f;
      const float tc = tanhf(c);
      const float do_ = dh * tc;
      const float dc = dh * o_ * (1.0f - tc * tc) + *st;
      gt[0 * H + j] = dc * g_ * i_ * (1.0f - i_);
      gt[1 * H + j] = dc * cp * f_ * (1.0f - f_);
      gt[2 * H + j] = dc * i_ * (1.0f - g_ * g_);
      gt[3 * H + j] = do_ * o_ * (1.0f - o_);
      *st = dc * f_;
    } else if (p.cell == CTCN_CELL_GRU) {
      float *st = p.state + ((size_t)b * D + d) * H + j;
      dh += *st;
      const float r_ = gt[0 * H + j], z_ = gt[1 * H + j], n_ = gt[2 * H + j];
      float *hnp = p.aux + (row_t * D + d) * H + j;
      const float hn = *hnp;
      const float hp = tp >= 0 ? p.y[((size_t)tp * B + b) * D * H + d * H + j] : 0.0f;
      const float dn = dh * (1.0f - z_);
      const float dz = dh * (hp - n_);
      const float dan = dn * (1.0f - n_ * n_);
      gt[0 * H + j] = dan * hn * r_ * (1.0f - r_);
      gt[1 * H + j] = dz * z_ * (1.0f - z_);
      gt[2 * H + j] = dan;
      *hnp = dan * r_;
      *st = dh * z_;
    } else {
      const float yv = p.y[row_t * D * H + d * H + j];
      gt[j] = dh * (1.0f - yv * yv);
    }
  }
}

int pick_kq4(int K, int ngroups, int mt, int budget) {
  const int cand[4] = {5, 4, 2, 1};
  int best = 1, best_cost = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int kq = cand[i];
    if (mt * kq > budget) continue;
    const int stride = ngroups * 4 * kq;
    const int cost = ceil_div(K, stride) * kq;
    if (cost < best_cost) { best_cost = cost; best = kq; }
  }
  return best;
}

template <int MT>
int launch_fwd(int kq4, dim3 grid, hipStream_t st, const RnnArgs &a) {
  switch (kq4) {
    case 5: hipLaunchKernelGGL((rnn_fwd_step<MT, 5>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((rnn_fwd_step<MT, 4>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((rnn_fwd_step<MT, 2>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((rnn_fwd_step<MT, 1>), grid, dim3(256), 0, st, a); break;
  }
  return 0;
}
template <int MT>
int launch_bwd(int kq4, dim3 grid, hipStream_t st, const RnnArgs &a) {
  // 1024-thread workgroups: 128 VGPRs per lane, so MT*KQ4 is capped by pick_kq4's budget (MT=4 -> KQ4<=2)
  if constexpr (MT <= 2) {
    if (kq4 == 5) { hipLaunchKernelGGL((rnn_bwd_step<MT, 5>), grid, dim3(1024), 0, st, a); return 0; }
    if (kq4 == 4) { hipLaunchKernelGGL((rnn_bwd_step<MT, 4>), grid, dim3(1024), 0, st, a); return 0; }
  }
  if (kq4 >= 2) hipLaunchKernelGGL((rnn_bwd_step<MT, 2>), grid, dim3(1024), 0, st, a);
  else hipLaunchKernelGGL((rnn_bwd_step<MT, 1>), grid, dim3(1024), 0, st, a);
  return 0;
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
  const int bchunks = ceil_div(B, 64);
  const int mt = B >= 64 ? 4 : ceil_div(B, 16);
  const int MT = mt <= 1 ? 1 : (mt == 2 ? 2 : 4);
  const int kq4 = pick_kq4(H, 16, MT, 20);
  dim3 grid(ceil_div(H, HS), dirs, bchunks);
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
  const int bchunks = ceil_div(B, 64);
  const int mt = B >= 64 ? 4 : ceil_div(B, 16);
  const int MT = mt <= 1 ? 1 : (mt == 2 ? 2 : 4);
  const int kq4 = pick_kq4(GH, 64, MT, MT <= 2 ? 10 : 8);
  dim3 grid(ceil_div(H, 16), dirs, bchunks);
  for (int s = 0; s < T; ++s) {
    a.step = s;
    if (MT == 1) launch_bwd<1>(kq4, grid, st, a);
    else if (MT == 2) launch_bwd<2>(kq4, grid, st, a);
    else launch_bwd<4>(kq4, grid, st, a);
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
