// core.hip -- error channel and device queries of libctcn.so
#include "common.h"

static thread_local char g_err[512] = "";

void ctcn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ctcn_version(void) { return 100; }
extern "C" const char *ctcn_last_error(void) { return g_err; }
extern "C" int ctcn_device_cus(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

// ---- runtime options / sticky device status word ---------------------------------------------------------
// rnn_persistent: 1 = persistent recurrent kernels (W_hh resident in VGPRs, in-launch flag + tile hand-off of h_t);
//                 0 = one launch per timestep.  Same arithmetic, same summation order (tests hold them to 2e-6).
// handoff, poll_depth, rnn_recurrence_only: see include/ctcn.h.
static int g_opt_rnn_persistent = 1;
static int g_opt_handoff = 1;
static int g_opt_poll_depth = 2;
static int g_opt_recurrence_only = 0;
static int g_opt_bwd_scatter = 1;
static int g_opt_handoff_tags = 1;
static int g_opt_side_split_wgs = 8;
static int g_opt_gemm_big_tiles = 0;   // 256x128 / 128x256 plane-GEMM tiles: measured 7 % slower than 128x128 x 2 per CU
static int g_opt_gemm_tile256 = 1;      // 1: 256-row plane-GEMM tiles (global_load_lds staging) for activation-sized products
static int g_opt_gemm_dbg = 0;          // development: bit 0 = 256-tile GEMM skips its stores, bit 1 = skips its DMA loads (results invalid)
static int g_opt_gemm_a_inline = 1;     // 1: 256-row tiles split a row-major float32 A operand while staging it (no plane pass over A)
static int g_opt_conv_mfma = 1;         // 1: Conv2d forward / dgrad / wgrad as MFMA implicit GEMMs; 0: the direct kernels
static int g_opt_gemm_pingpong = 1;     // 1: 256-row plane tiles run the ping-pong schedule (wave halves half a k-step apart)
static int g_opt_rnn_mixed_slices = 0;  // forward recurrence: one workgroup per CU with mixed 12- / 4-unit slices (experiment)
static int g_opt_rnn_fwd_tagged = 1;    // forward persistent recurrence: 1 = tagged gather (rnn_fwd_tagged) where it applies
static int g_opt_rnn_fused_dropout = 1; // 1: ctcn_rnn_fwd_dropout stores the dropped output from inside rnn_fwd_tagged; 0: dropout kernel behind the recurrence
static int g_opt_gemm_tn = 1;           // 1: weight-gradient products (both operands contraction-major) on the TN tile, no plane pass
static int g_opt_tag_poll_delay = 8;    // rnn_fwd_tagged: 64-cycle sleeps before an exchange wave's first poll of a step
static int g_opt_edit_wave = 1;         // 1: edit distance on one wavefront per utterance (anti-diagonals); 0: one lane per utterance
static int g_opt_bwd_item_gather = 1;   // backward scatter recurrence with item-wave gather + exchange-wave reserve traffic (rnn_bwd_scatter2): 0 never, 1 for H > 320, 2 always
static int g_opt_bwd_poll_delay = -1;   // rnn_bwd_scatter2: 64-cycle sleeps before an item wave's first poll of a step; -1 = auto (16 up to 24 slices, 24 beyond)
static int g_opt_fwd_rsv_lds = 2;       // rnn_fwd_tagged: reserve traffic through LDS (16-B stores by the exchange waves, LDS-DMA pre-activation loads): 0 off, 1 on, 2 for H > 384
static int g_opt_fwd_pipe_any_chunking = 0;  // 1: pipeline the input projection with the forward recurrence even when no chunk count fits the side stream's one-round criterion (tests)
static int g_opt_fwd_pipe_min_input = 0;       // smallest layer input width whose projection is pipelined with the forward recurrence
static int g_opt_conv_dbg = 0;          // development: conv_mfma_kernel skips phases (1 window load, 2 MFMA loop, 4 output phase); results invalid
static int g_opt_rnn_rsv_nt = 0;        // rnn_bwd_scatter2: non-temporal hint on the reserve traffic (experiment: keep the exchange tiles in L2 at H = 512)
static int g_opt_gemm_bf16_single = 0;  // 256-row GEMM tiles: one bf16 product (ah*bh) instead of the three bf16x3 products (north_star's bf16 tolerance; gemm.hip)
static int g_opt_xcd_interleave = 1;    // which physical XCD hosts group g of a persistent recurrence that leaves XCDs idle (eight XCDs; 0: XCD g, 1 (default): the even XCDs first, 2-5: other orders; the host's xcd_allow masks follow: ops._idle_xcd_mask)
static int g_opt_bn_rows4 = 1;          // BatchNorm over (T*B, C) rows: column sums with 16-B loads (colreduce_rows4_kernel); 0: the dword kernel
static int g_opt_tn_splits_force = 0;   // development (tools/gemm_tn_bench.py): split-K count of the TN tile, 0 = the rule of gemm.hip:tn_splits
static int g_opt_tn_splits_xcd = 1;     // TN weight-gradient tile: split count sized for the CUs of xcd_allow (one round of items there), not for the whole device
static int g_opt_beam_generic_threads = 0;  // generic beam kernel: 0 = 256 threads per utterance up to W = 64 and 1 024 beyond; 256 / 1024 = forced
static int g_opt_beam_occ2 = 0;         // fast beam search compiled / launched for TWO workgroups per CU (<= 64 VGPRs, <= 80 KB LDS): 0 off, 1 on, 2 on with the LM in global memory
static int g_opt_beam_fast = 1;         // 1: restructured beam search (W <= 60, W*V <= 3328); 0: the generic kernel always
static int *g_status_dev = nullptr;

extern "C" int ctcn_set_option(const char *name, int value) {
  if (name && !strcmp(name, "rnn_persistent")) { g_opt_rnn_persistent = value; return CTCN_OK; }
  if (name && !strcmp(name, "handoff")) { g_opt_handoff = value; return CTCN_OK; }
  if (name && !strcmp(name, "poll_depth")) { g_opt_poll_depth = value; return CTCN_OK; }
  if (name && !strcmp(name, "rnn_recurrence_only")) { g_opt_recurrence_only = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "bwd_scatter")) { g_opt_bwd_scatter = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "handoff_tags")) { g_opt_handoff_tags = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "side_split_wgs")) { g_opt_side_split_wgs = value < 1 ? 1 : (value > 16 ? 16 : value); return CTCN_OK; }
  if (name && !strcmp(name, "gemm_big_tiles")) { g_opt_gemm_big_tiles = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "beam_fast")) { g_opt_beam_fast = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "conv_dbg")) { g_opt_conv_dbg = value & 7; return CTCN_OK; }
  if (name && !strcmp(name, "rnn_rsv_nt")) { g_opt_rnn_rsv_nt = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "tn_splits_xcd")) { g_opt_tn_splits_xcd = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "xcd_interleave")) { g_opt_xcd_interleave = value < 0 ? 0 : (value > 5 ? 5 : value); return CTCN_OK; }
  if (name && !strcmp(name, "bn_rows4")) { g_opt_bn_rows4 = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "tn_splits_force")) { g_opt_tn_splits_force = value < 0 ? 0 : value; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_bf16_single")) { g_opt_gemm_bf16_single = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "beam_generic_threads")) { g_opt_beam_generic_threads = (value == 256 || value == 1024) ? value : 0; return CTCN_OK; }
  if (name && !strcmp(name, "beam_occ2")) { g_opt_beam_occ2 = value < 0 ? 0 : (value > 2 ? 2 : value); return CTCN_OK; }
  if (name && !strcmp(name, "fwd_pipe_min_input")) { g_opt_fwd_pipe_min_input = value < 0 ? 0 : value; return CTCN_OK; }
  if (name && !strcmp(name, "fwd_pipe_any_chunking")) { g_opt_fwd_pipe_any_chunking = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "fwd_rsv_lds")) { g_opt_fwd_rsv_lds = value < 0 ? 0 : (value > 2 ? 2 : value); return CTCN_OK; }
  if (name && !strcmp(name, "bwd_item_gather")) { g_opt_bwd_item_gather = value < 0 ? 0 : (value > 2 ? 2 : value); return CTCN_OK; }
  if (name && !strcmp(name, "bwd_poll_delay")) { g_opt_bwd_poll_delay = value < 0 ? -1 : (value > 256 ? 256 : value); return CTCN_OK; }
  if (name && !strcmp(name, "conv_mfma")) { g_opt_conv_mfma = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "edit_wave")) { g_opt_edit_wave = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_tn")) { g_opt_gemm_tn = value != 0; return CTCN_OK; }
  if (name && !strcmp(name, "rnn_fused_dropout")) { g_opt_rnn_fused_dropout = value != 0; return CTCN_OK; }
  if (name && !strcmp(name, "tag_poll_delay")) { g_opt_tag_poll_delay = value < 0 ? 0 : (value > 64 ? 64 : value); return CTCN_OK; }
  if (name && !strcmp(name, "rnn_fwd_tagged")) { g_opt_rnn_fwd_tagged = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "rnn_mixed_slices")) { g_opt_rnn_mixed_slices = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_pingpong")) { g_opt_gemm_pingpong = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_a_inline")) { g_opt_gemm_a_inline = value ? 1 : 0; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_dbg")) { g_opt_gemm_dbg = value; return CTCN_OK; }
  if (name && !strcmp(name, "gemm_tile256")) { g_opt_gemm_tile256 = value ? 1 : 0; return CTCN_OK; }
  ctcn_set_error("ctcn_set_option: unknown option %s", name ? name : "(null)");
  return CTCN_EINVAL;
}
extern "C" int ctcn_get_option(const char *name) {
  if (name && !strcmp(name, "rnn_persistent")) return g_opt_rnn_persistent;
  if (name && !strcmp(name, "handoff")) return g_opt_handoff;
  if (name && !strcmp(name, "poll_depth")) return g_opt_poll_depth;
  if (name && !strcmp(name, "rnn_recurrence_only")) return g_opt_recurrence_only;
  if (name && !strcmp(name, "bwd_scatter")) return g_opt_bwd_scatter;
  if (name && !strcmp(name, "handoff_tags")) return g_opt_handoff_tags;
  if (name && !strcmp(name, "side_split_wgs")) return g_opt_side_split_wgs;
  if (name && !strcmp(name, "gemm_big_tiles")) return g_opt_gemm_big_tiles;
  if (name && !strcmp(name, "beam_fast")) return g_opt_beam_fast;
  if (name && !strcmp(name, "beam_occ2")) return g_opt_beam_occ2;
  if (name && !strcmp(name, "beam_generic_threads")) return g_opt_beam_generic_threads;
  if (name && !strcmp(name, "tn_splits_xcd")) return g_opt_tn_splits_xcd;
  if (name && !strcmp(name, "xcd_interleave")) return g_opt_xcd_interleave;
  if (name && !strcmp(name, "bn_rows4")) return g_opt_bn_rows4;
  if (name && !strcmp(name, "tn_splits_force")) return g_opt_tn_splits_force;
  if (name && !strcmp(name, "gemm_bf16_single")) return g_opt_gemm_bf16_single;
  if (name && !strcmp(name, "rnn_rsv_nt")) return g_opt_rnn_rsv_nt;
  if (name && !strcmp(name, "conv_dbg")) return g_opt_conv_dbg;
  if (name && !strcmp(name, "fwd_pipe_min_input")) return g_opt_fwd_pipe_min_input;
  if (name && !strcmp(name, "fwd_pipe_any_chunking")) return g_opt_fwd_pipe_any_chunking;
  if (name && !strcmp(name, "fwd_rsv_lds")) return g_opt_fwd_rsv_lds;
  if (name && !strcmp(name, "bwd_item_gather")) return g_opt_bwd_item_gather;
  if (name && !strcmp(name, "bwd_poll_delay")) return g_opt_bwd_poll_delay;
  if (name && !strcmp(name, "conv_mfma")) return g_opt_conv_mfma;
  if (name && !strcmp(name, "edit_wave")) return g_opt_edit_wave;
  if (name && !strcmp(name, "gemm_tn")) return g_opt_gemm_tn;
  if (name && !strcmp(name, "rnn_fused_dropout")) return g_opt_rnn_fused_dropout;
  if (name && !strcmp(name, "tag_poll_delay")) return g_opt_tag_poll_delay;
  if (name && !strcmp(name, "rnn_fwd_tagged")) return g_opt_rnn_fwd_tagged;
  if (name && !strcmp(name, "rnn_mixed_slices")) return g_opt_rnn_mixed_slices;
  if (name && !strcmp(name, "gemm_pingpong")) return g_opt_gemm_pingpong;
  if (name && !strcmp(name, "gemm_a_inline")) return g_opt_gemm_a_inline;
  if (name && !strcmp(name, "gemm_dbg")) return g_opt_gemm_dbg;
  if (name && !strcmp(name, "gemm_tile256")) return g_opt_gemm_tile256;
  return -1;
}
extern "C" int ctcn_set_status_buffer(int *dev_word) { g_status_dev = dev_word; return CTCN_OK; }
int *ctcn_status_word(void) { return g_status_dev; }
int ctcn_opt_rnn_persistent(void) { return g_opt_rnn_persistent; }
int ctcn_opt_handoff(void) { return g_opt_handoff; }
int ctcn_opt_poll_depth(void) { return g_opt_poll_depth; }
int ctcn_opt_recurrence_only(void) { return g_opt_recurrence_only; }
int ctcn_opt_bwd_scatter(void) { return g_opt_bwd_scatter; }
int ctcn_opt_handoff_tags(void) { return g_opt_handoff_tags; }
int ctcn_opt_side_split_wgs(void) { return g_opt_side_split_wgs; }
int ctcn_opt_gemm_big_tiles(void) { return g_opt_gemm_big_tiles; }
