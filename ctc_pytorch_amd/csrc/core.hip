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
static int g_opt_beam_generic_threads = 0;  // generic beam kernel: 0 = 256 threads per utterance up to W = 64 and 1 024 beyond; 256 / 512 / 1024 = forced
static int g_opt_beam_bitonic = 1;      // generic beam kernel: 1 = up to 256 survivors of the pruning bound are ranked by a bitonic sort; 0 = by counting pairs (rounds 5-6)
static int g_opt_beam_cand_global = 0;  // generic beam kernel: 1 = the candidate table stays in global memory (L2) even where LDS would hold it (the LM table takes the room; two 512-thread searches per CU then fit)
static int g_opt_beam_occ2 = 0;         // fast beam search compiled / launched for TWO workgroups per CU (<= 64 VGPRs, <= 80 KB LDS): 0 off, 1 on, 2 on with the LM in global memory
static int g_opt_beam_fast = 1;         // 1: restructured beam search (W <= 60, W*V <= 3328); 0: the generic kernel always
static int g_opt_rnn_proj_order = 0;     // 1: input projection of a recurrent layer as row blocks [T/2, T) then [0, T/2) (round 6 mitigation attempt; bit-identical; off: the cause was elsewhere, include/ctcn.h)
static int g_opt_rnn_slow_items = 0;    // parity harness: > 0 = rnn_fwd_tagged runs its SLOW instantiation, the item waves sleeping this many x 64 cycles before they read the parked tiles (every step): results must not change
static int g_opt_rnn_slow_exchange = 0; // parity harness: as rnn_slow_items, for the exchange waves (before their first LDS / global read of a step)
static int g_opt_xcd_interleave_force = 0;  // development / parity harness: apply "xcd_interleave" to a recurrence that takes EVERY XCD too (rnn.hip: xcd_order_for)
static int *g_status_dev = nullptr;

// One table for the setter, the getter and the enumerator (ctcn_option_name: what a test harness snapshots -- tests/conftest.py): an option
// that exists is listed here once, with the normalisation its setter applies.
struct OptionRow { const char *name; int *var; int (*norm)(int); };
static const OptionRow k_options[] = {
  {"rnn_persistent", &g_opt_rnn_persistent, [](int value) -> int { return value; }},
  {"handoff", &g_opt_handoff, [](int value) -> int { return value; }},
  {"poll_depth", &g_opt_poll_depth, [](int value) -> int { return value; }},
  {"rnn_recurrence_only", &g_opt_recurrence_only, [](int value) -> int { return value ? 1 : 0; }},
  {"bwd_scatter", &g_opt_bwd_scatter, [](int value) -> int { return value ? 1 : 0; }},
  {"handoff_tags", &g_opt_handoff_tags, [](int value) -> int { return value ? 1 : 0; }},
  {"side_split_wgs", &g_opt_side_split_wgs, [](int value) -> int { return value < 1 ? 1 : (value > 16 ? 16 : value); }},
  {"gemm_big_tiles", &g_opt_gemm_big_tiles, [](int value) -> int { return value ? 1 : 0; }},
  {"beam_fast", &g_opt_beam_fast, [](int value) -> int { return value ? 1 : 0; }},
  {"conv_dbg", &g_opt_conv_dbg, [](int value) -> int { return value & 7; }},
  {"rnn_rsv_nt", &g_opt_rnn_rsv_nt, [](int value) -> int { return value ? 1 : 0; }},
  {"tn_splits_xcd", &g_opt_tn_splits_xcd, [](int value) -> int { return value ? 1 : 0; }},
  {"xcd_interleave", &g_opt_xcd_interleave, [](int value) -> int { return value < 0 ? 0 : (value > 5 ? 5 : value); }},
  {"bn_rows4", &g_opt_bn_rows4, [](int value) -> int { return value ? 1 : 0; }},
  {"tn_splits_force", &g_opt_tn_splits_force, [](int value) -> int { return value < 0 ? 0 : value; }},
  {"gemm_bf16_single", &g_opt_gemm_bf16_single, [](int value) -> int { return value ? 1 : 0; }},
  {"beam_generic_threads", &g_opt_beam_generic_threads, [](int value) -> int { return (value == 256 || value == 512 || value == 1024) ? value : 0; }},
  {"beam_cand_global", &g_opt_beam_cand_global, [](int value) -> int { return value ? 1 : 0; }},
  {"beam_bitonic", &g_opt_beam_bitonic, [](int value) -> int { return value ? 1 : 0; }},
  {"beam_occ2", &g_opt_beam_occ2, [](int value) -> int { return value < 0 ? 0 : (value > 2 ? 2 : value); }},
  {"fwd_pipe_min_input", &g_opt_fwd_pipe_min_input, [](int value) -> int { return value < 0 ? 0 : value; }},
  {"fwd_pipe_any_chunking", &g_opt_fwd_pipe_any_chunking, [](int value) -> int { return value ? 1 : 0; }},
  {"fwd_rsv_lds", &g_opt_fwd_rsv_lds, [](int value) -> int { return value < 0 ? 0 : (value > 2 ? 2 : value); }},
  {"bwd_item_gather", &g_opt_bwd_item_gather, [](int value) -> int { return value < 0 ? 0 : (value > 2 ? 2 : value); }},
  {"bwd_poll_delay", &g_opt_bwd_poll_delay, [](int value) -> int { return value < 0 ? -1 : (value > 256 ? 256 : value); }},
  {"conv_mfma", &g_opt_conv_mfma, [](int value) -> int { return value ? 1 : 0; }},
  {"edit_wave", &g_opt_edit_wave, [](int value) -> int { return value ? 1 : 0; }},
  {"gemm_tn", &g_opt_gemm_tn, [](int value) -> int { return value != 0; }},
  {"rnn_fused_dropout", &g_opt_rnn_fused_dropout, [](int value) -> int { return value != 0; }},
  {"tag_poll_delay", &g_opt_tag_poll_delay, [](int value) -> int { return value < 0 ? 0 : (value > 64 ? 64 : value); }},
  {"rnn_fwd_tagged", &g_opt_rnn_fwd_tagged, [](int value) -> int { return value ? 1 : 0; }},
  {"rnn_mixed_slices", &g_opt_rnn_mixed_slices, [](int value) -> int { return value ? 1 : 0; }},
  {"gemm_pingpong", &g_opt_gemm_pingpong, [](int value) -> int { return value ? 1 : 0; }},
  {"gemm_a_inline", &g_opt_gemm_a_inline, [](int value) -> int { return value ? 1 : 0; }},
  {"gemm_dbg", &g_opt_gemm_dbg, [](int value) -> int { return value; }},
  {"gemm_tile256", &g_opt_gemm_tile256, [](int value) -> int { return value ? 1 : 0; }},
  {"rnn_proj_order", &g_opt_rnn_proj_order, [](int value) -> int { return value ? 1 : 0; }},
  {"xcd_interleave_force", &g_opt_xcd_interleave_force, [](int value) -> int { return value ? 1 : 0; }},
  {"rnn_slow_exchange", &g_opt_rnn_slow_exchange, [](int value) -> int { return value < 0 ? 0 : (value > 4096 ? 4096 : value); }},
  {"rnn_slow_items", &g_opt_rnn_slow_items, [](int value) -> int { return value < 0 ? 0 : (value > 4096 ? 4096 : value); }},
};
static const int k_noptions = (int)(sizeof(k_options) / sizeof(k_options[0]));
static const OptionRow *find_option(const char *name) {
  if (name)
    for (int i = 0; i < k_noptions; ++i)
      if (!strcmp(name, k_options[i].name)) return &k_options[i];
  return nullptr;
}
extern "C" int ctcn_set_option(const char *name, int value) {
  if (const OptionRow *o = find_option(name)) { *o->var = o->norm(value); return CTCN_OK; }
  ctcn_set_error("ctcn_set_option: unknown option %s", name ? name : "(null)");
  return CTCN_EINVAL;
}
extern "C" int ctcn_get_option(const char *name) {
  const OptionRow *o = find_option(name);
  return o ? *o->var : -1;
}
extern "C" const char *ctcn_option_name(int index) { return index >= 0 && index < k_noptions ? k_options[index].name : nullptr; }
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
