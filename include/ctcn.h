/* ctcn.h -- C ABI of libctcn.so: the MI355X (gfx950) CTC acoustic-model hot path.
 *
 * The reference (Diamondfan/CTC_pytorch) has no FFI of its own: its seam is the Python class surface
 * (timit/models/model_ctc.py, timit/utils/ctcDecoder.py, nn.CTCLoss at timit/steps/train_ctc.py:144).
 * Every entry point below replaces one torch op that surface dispatches; the reference call site it
 * replaces is cited per function.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a HIP stream passed as void* (hipStream_t); no torch types.
 *   - every function returns int: 0 = OK, <0 = error (CTCN_E*); text via ctcn_last_error() (thread-local).
 *   - the caller owns every buffer (inputs, outputs, reserve, workspace); the library allocates nothing on
 *     the device and keeps no pointer past the call; all work is enqueued on `stream` and the call returns
 *     without synchronising (except where stated).
 *   - all tensors are dense row-major float32 unless stated; "T,B,C" means time-major (t, b, c).
 */
#ifndef CTCN_H
#define CTCN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CTCN_OK 0
#define CTCN_EINVAL (-1)      /* bad dims / null pointer / misaligned */
#define CTCN_EHIP (-2)        /* HIP runtime error (message carries hipGetErrorString) */
#define CTCN_EUNSUPPORTED (-3)
#define CTCN_EWORKSPACE (-4)  /* workspace too small */

#define CTCN_CELL_LSTM 0 /* gate rows i,f,g,o */
#define CTCN_CELL_GRU 1  /* gate rows r,z,n */
#define CTCN_CELL_TANH 2 /* Elman RNN, tanh */

int ctcn_version(void);
const char *ctcn_last_error(void);
/* number of CUs of the current device (host query, used to size grids / workspaces) */
int ctcn_device_cus(void);
/* XCDs of the current device when consecutive workgroup ids are dealt round-robin over them (probed once), else 1 */
int ctcn_device_xcds(void);
/* options: "rnn_persistent" = 1 (default): the recurrence of ctcn_rnn_fwd/bwd runs as ONE persistent launch per layer
 * (W_hh slice resident in VGPRs, h_t handed between workgroups in-launch); 0: one launch per timestep.
 * "handoff" = 1 (default): persistent launches place each (direction, batch-tile) group on one XCD and hand h_t over
 * through that XCD's L2 when the device allows it (falls back to 0 otherwise); 0: device-scope write-through hand-off.
 * "poll_depth" = 2 (default): flag polls kept in flight per polling wave in the XCD-local hand-off (1..4).
 * "bwd_scatter" = 1 (default): the persistent backward recurrence exchanges partial dh tiles (scatter formulation); 0: every
 * workgroup gathers the whole d(pre-activation) tile (rnn_bwd_persist).
 * "handoff_tags" = 1 (default): the scatter formulation hands its partial tiles over WITHOUT flags: every float carries a
 * step tag in its mantissa LSB (cleared again by the consumer: partial sums lose 1 ulp) and the consumer polls the block itself,
 * which removes the producer's store drain and the consumer's flag round trip from the per-step chain; 0: drain + one flag per block.
 * "side_split_wgs" = 8 (default): workgroups per CU launched by the XCD-filtered operand-split kernels of the weight-gradient
 * side stream (1..16; measured at cfg2: 1 -> 20.2, 2 -> 18.0, 4 -> 17.0, 8..16 -> 16.8 ms per step: the side work is nearly critical).
 * "beam_fast" = 1 (default): ctcn_beam_decode uses the restructured search (wave-local top-W extraction, LM in LDS, ln p precomputed
 * by a parallel pre-pass) whenever W <= 60, W*V <= 3328 and V <= 256 (beams wider than 52 prune nothing: slower); 0: always the generic kernel (same results; ~5x slower at W = 20).
 * Wider beams -- the reference's class default is beam_width = 200 (ctcDecoder.py:170) -- always run the generic kernel beam_kernel<NT>, 1 024 threads per
 * utterance beyond W = 64.  Round 5: its selection is a pruning bound (every wave's k-th largest thread maximum by a ballot search on order-preserving
 * keys) + a rank count of the ~W survivors instead of W block-wide arg-max rounds per frame: the cfg5 batch (128 x 800 x 62) at W = 200 takes
 * 11.8 ms (peaky) / 28.1 ms (flat) instead of 657 / 1 556 ms, W = 256 14.2 / 34.8 instead of 1 066 / 2 514 (profiles/r05_wide_beam.txt).
 * "gemm_tile256" = 1 (default): the bf16x3 GEMM multiplies activation-sized products (M >= 1024 rows, enough tiles to fill the
 * device) with 256 x 256 / 256 x 128 workgroup tiles staged by global_load_lds; 0: always the 128 x 128 tile (same results).
 * "gemm_a_inline" = 1 (default): those tiles take a row-major float32 A operand as it is and split it into bf16 planes while staging
 * it (no separate plane pass over A; same planes, same results); 0: A is pre-split like B.
 * "conv_mfma" = 1 (default): ctcn_conv2d_fwd / _bwd run as implicit GEMMs on v_mfma_f32_16x16x4_f32 (im2col tile of 64 positions and
 * the filter matrix staged in LDS; exact float32 in both matmul precisions); 0: the direct (lane-per-position) kernels.
 * "gemm_pingpong" = 1 (default): the 256-row bf16x3 plane tiles run the ping-pong schedule (the two waves of a SIMD half a
 * 16-k step apart: one multiplies while the other reads its fragments; DMA pieces issued between MFMAs); 0: all waves in phase.
 * Bit-identical results either way.
 * "gemm_tn" = 1 (default): bf16x3 products C = A^T B with BOTH operands contraction-major (transA = 1, transB = 0: the weight
 * gradients dW = da^T x; M >= 128, N >= 32, K >= 1024, 16-B aligned rows) run on the TN tile -- float32 rows split into hi / lo while
 * they are staged, ds_read_b64_tr_b16 fragments, split-K queue -- instead of a transposing plane pass + the NT plane tiles; the same
 * bf16x3 operands and products, the k-sums grouped differently (results agree to float32 rounding of the sums).
 * "rnn_fwd_tagged" = 1 (default): forward persistent recurrence as a tagged gather (rnn_fwd_tagged: no flags, no store drain; the
 * h_t dwords carry the step tag and the gathering waves poll the data itself) where it applies -- precision 1, LSTM / GRU, H % 32 == 0,
 * XCD-local placement; 0: the flag + data kernel (rnn_fwd_persist) everywhere.  "tag_poll_delay" = 8: 64-cycle sleeps between the
 * step barrier and a gathering wave's first poll.
 * "rnn_mixed_slices" = 0 (default): 1 = forward persistent recurrence with one workgroup per CU of an XCD and mixed 12- / 4-unit
 * slices (measured slower than 40 equal slices at H = 320; kept as an experiment switch).
 * "edit_wave" = 1 (default): ctcn_edit_distance runs one wavefront per utterance along anti-diagonals (labels up to 512 symbols);
 * 0: one lane per utterance with its DP row in LDS (also the path for longer labels).
 * "gemm_big_tiles" = 0 (default): 1 lets the bf16x3 GEMM use 256x128 / 128x256 workgroup tiles (same results, measured slower).
 * "tn_splits_xcd" = 1 (default, round 5): the TN tile sizes its split-K count for the CUs of the XCDs it may run on (xcd_allow of
 * ctcn_rnn_bwd_weights: one round of items on the idle XCDs next to a recurrence) instead of for the whole device (two rounds of items half
 * as long, each paying a prologue, a 128-KB partial store and its share of the reduce pass): cfg2 13.22 -> 13.15 ms per step; 0: as before.
 * The k-sums are grouped differently (float32 rounding of the sums), deterministically either way.
 * "xcd_interleave" = 1 (default; round 5): which physical XCD hosts group g -- (direction, 16-row batch tile) -- of a persistent recurrence that
 * leaves XCDs idle on a device of eight, and therefore which XCDs its side-stream GEMMs (pipelined projection chunks, weight gradients) get:
 * 0: XCD g; 1: the even XCDs first {0 2 4 6 | 1 3 5 7}; 2: {0 1 4 5 | ..}; 3: {0 3 4 7 | ..}; 4: {0 2 5 7 | ..}; 5: {0 4 1 5 | ..}.  The host
 * derives its xcd_allow masks from the same order (ops._idle_xcd_mask).  Same results whatever the order.  Measured at cfg2 (two runs each):
 * order 0 13.19 / 13.20 ms per step; orders 1, 3, 4 -- ONE recurrence XCD and one GEMM XCD in every pair (2k, 2k + 1) -- 13.04-13.08; orders 2,
 * 5 -- both XCDs of a pair on the same side -- 13.26-13.31; cfg1 / cfg3 / shipped YAML unchanged.  A recurrence that takes EVERY XCD (cfg4:
 * groups = XCDs) keeps group g on XCD g whatever the option says: there is nothing to place.  (The cfg4 trajectory divergence first seen while the
 * order applied to that launch too had nothing to do with placement: round 6 traced it to the parked tiles of rnn_fwd_tagged)
 * "rnn_proj_order" = 0 (default): 1 issues the input projection of a recurrent layer as the row blocks [T/2, T) then [0, T/2) (pipelined form:
 * the last time chunk before the first) instead of one product over ascending time.  Same products, bit-identical results.  Built in round 6 as a
 * mitigation while the cfg4 trajectory divergence was taken for a stale read behind a kernel boundary; the after-suite A/Bs showed both orders
 * deviating alike and the cause turned out to be the single-buffered parked tiles of rnn_fwd_tagged (rnn.hip) -- kept as a
 * switch for the record, off.
 * "rnn_slow_items" = 0 (parity harness): N > 0 runs the SLOW instantiation of rnn_fwd_tagged, whose item waves sleep N x 64 cycles before they read
 * the parked partial tiles, at every step.  Results must not change (test_rnn_fwd_tagged_with_slow_item_waves): the hand-off inside a workgroup may
 * rest on barriers and buffer parity only, never on which wave is faster.  "rnn_slow_exchange" = N: the same for the exchange waves; both options also select
 * the SLOW instantiations of rnn_bwd_scatter / rnn_bwd_scatter2 (item waves sleep before they read the parked tiles / start their gather, exchange waves
 * behind the barrier before they read the staged operand): test_rnn_bwd_with_slow_waves.
 * "bn_rows4" = 1 (default, round 5): BatchNorm over (rows, C) with C % 4 == 0 forms its column sums with 16-B loads, sixteen row phases per
 * workgroup (colreduce_rows4_kernel); 0: the dword kernel.  Same chunks, same element values, float64 partials grouped differently: the float32
 * results agreed bit for bit wherever compared (tools/bn_rows_probe.py).  cfg2 13.33 -> 13.25 ms per step, cfg4 53.2 -> 52.8.
 * "tn_splits_force" = 0 (default; development): n > 0 forces the split-K count of the TN tile (tools/gemm_tn_splits_probe.py: the rule's own
 * choice -- one round of (tile, split) items on the CUs the launch may use -- is where the time is shortest on the cfg2 / cfg4 products).
 * "gemm_bf16_single" = 0 (default): 1 = the opt-in bf16 mode (round 5).  With precision 1 the 256-row GEMM tiles -- every product over the T*B
 * rows of a layer: input projections, dx, weight gradients (plane, float32-A and TN tile) -- multiply the bf16 ROUNDINGS of their operands once
 * (ah*bh, f32 accumulate) instead of the three bf16x3 products (al*bh + ah*bl + ah*bh); the recurrent matmul and the small tiles keep bf16x3.
 * Result = the product of the rounded operands to f32 summation order (tools/gemm_single_probe.py: 2-5e-6 of mean |C|); against the exact product
 * an element moves by up to 1.5e-2 of mean |C|.  BASELINE.json's north_star tolerance ("loss and activations within 1e-3 bf16 tolerance"):
 * full-size cfg2 / cfg3 / cfg4 loss within 1e-5 .. 3e-5 of the REFERENCE's, gradient norms within 6e-4 .. 4e-3, log-probs 2-3e-3 mean (1.8e-2 max)
 * next to the default mode's (test_bf16_single_mode_against_reference_checksums).  Tile time 1.27-1.84x shorter; cfg2 13.21 -> 12.69 ms per
 * step, cfg3 7.71 -> 7.42, cfg4 52.6 -> 45.5.  The default (0) keeps every parity statement of this header (1e-5 against the reference).
 * "beam_occ2" = 0 (default, round 5): 1 / 2 launch the fast beam search compiled for eight waves per SIMD (<= 64 VGPRs, 43 spilled dwords)
 * with <= 68 KB of dynamic LDS (4 096-slot trie; 2: LM in global memory, 8 192 slots) so that two utterances share a CU.  Same results;
 * measured SLOWER (cfg5, three searches in flight: 279 k -> 248 k utt/s peaky, 128 k -> 97 k flat; profiles/r05_beam_occ2_ab.txt) and kept
 * as an experiment switch.
 * "beam_generic_threads" = 0 (default): the generic beam kernel runs 256 threads per utterance, 1 024 beyond W = 64 or 3 500 candidates per frame;
 * 256 / 512 / 1024 force one (measurements).
 * "beam_bitonic" = 1 (default): the generic beam kernel ranks up to 256 survivors of its pruning bound (the reference's W = 200 leaves ~1.1 W) by a
 * bitonic sort on waves 0-3 (DPP / v_permlane swaps); 0 = by counting pairs, as it does for more survivors.  Same labellings and scores.
 * "beam_cand_global" = 0 (default): 1 keeps the generic kernel's candidate table in global memory (L2) even where the LDS would hold it; the LM
 * table takes the room.  With "beam_generic_threads" = 512 two searches of the reference-default width then share a CU (116 VGPRs, 77 KB of
 * LDS each).  Same results bit for bit (test_beam_generic_kernel_occupancy_options).  Measured on the kernel as it stood at the start of the
 * round's last session (cfg5, W = 200, profiles/r06_wide_beam_probe.txt): +13 % / +22 % utterances/s (peaky / flat) with twelve searches in
 * flight, -13 % with three, one batch 14 % slower -- an opt-in for offline decoding with many batches in flight, not the default.
 * "rnn_rsv_nt" = 0 (default): 1 puts the non-temporal hint on rnn_bwd_scatter2's reserve loads / stores (experiment: -15 % L2 write-backs at H = 512, no
 * change of the step).
 * "conv_dbg" = 0 (default): development only (tools/conv_phase_probe.py) -- conv_mfma_kernel skips its window load (1), MFMA loop (2) and /
 * or output phase (4); results are invalid.
 * "fwd_pipe_any_chunking" = 0 (default): the input projection is pipelined with the forward recurrence (ctcn_rnn_call.side_stream) only
 * when a time-chunk count exists whose chunk pair the side stream digests in one round of 256-row tiles on the idle XCDs (cfg2: 10 chunks
 * of 2 560 rows), with at least 72 steps per chunk, at a flop rate the idle XCDs sustain within the recurrence's time for the chunk, and with
 * eight CUs of every recurrence XCD left free; 1: with the default 8 chunks otherwise too (slower where measured: cfg3 8.51 vs 7.87 ms per
 * step; the parity tests of the pipeline use it).  "fwd_pipe_min_input" = 0: smallest layer input width that is pipelined (experiments).
 * Environment CTCN_LOG_PIPE=1 prints the decision of every ctcn_rnn_fwd_ex call to stderr.
 * "fwd_rsv_lds" = 2 (default): rnn_fwd_tagged moves its reserve traffic through LDS (the items park their values, exchange waves store
 * them with 16-B stores in the pause before their first poll, pre-activations arrive by LDS DMA two steps ahead) for H > 384; 1: wherever the
 * reserves are 16-B aligned; 0: never (scattered dword stores / loads from the item waves).  Same values either way.
 * "bwd_item_gather" = 1 (default): the scatter formulation of the backward recurrence runs as rnn_bwd_scatter2 (item waves gather their own
 * 256-B quarters of the partial tiles, one barrier per step, all reserve traffic on the exchange waves through LDS DMA, 64-bit reserve
 * addresses, up to 40 slices = H <= 640) where it measured faster: more than 20 slices (H > 320); 0: never; 2: wherever it applies.
 * "bwd_poll_delay" = -1 (auto): 64-cycle sleeps before an item wave's first poll of a step in rnn_bwd_scatter2.
 * "rnn_recurrence_only" = 0 (default); 1 is a MEASUREMENT aid: ctcn_rnn_fwd / ctcn_rnn_bwd skip their input-projection
 * and deferred gradient GEMMs so that bench.py can time the recurrent kernel alone -- outputs are not valid. */
int ctcn_set_option(const char *name, int value);
int ctcn_get_option(const char *name);
/* The names ctcn_set_option knows, index 0 .. n-1; NULL past the end.  (Round 6: what a harness needs to snapshot the whole table -- the
 * reference has no counterpart; tests/conftest.py asserts that every GPU test leaves the table as it found it.)
 * "xcd_interleave_force" = 0 (default; development): 1 applies "xcd_interleave" to a recurrence that takes EVERY XCD too (cfg4) -- a
 * relabelling of XCDs with bit-identical results, held to that by test_rnn_results_do_not_depend_on_the_xcd_order. */
const char *ctcn_option_name(int index);
/* optional device int that persistent kernels set to a non-zero code if an in-launch hand-off times out (sticky);
 * the caller zeroes it and reads it at its own synchronisation points.  PROCESS-WIDE default (one word for every call that does not
 * bring its own in ctcn_rnn_call.status): a caller with several models / devices passes per-call words instead. */
int ctcn_set_status_buffer(int *dev_word);

/* ---------------------------------------------------------------------------------------------------
 * GEMM (MFMA, f32 in / f32 accumulate: v_mfma_f32_32x32x2_f32, or bf16x3 split operands: v_mfma_f32_32x32x16_bf16)
 * replaces: nn.Linear (model_ctc.py:137,166) and the input-projection part of nn.LSTM/GRU/RNN
 * (model_ctc.py:33).  C[M,N] = op(A)[M,K] * op(B)[K,N] + beta*C, row-major:
 *   transA==0: A[m*lda+k]   transA!=0: A[k*lda+m]     transB==0: B[k*ldb+n]   transB!=0: B[n*ldb+k]
 * ws/ws_bytes: optional split-K workspace (deterministic two-pass reduce); may be NULL/0.
 * precision: 0 = exact f32 MFMA; 1 = each f32 operand split into bf16 hi + lo planes, three bf16 MFMAs per product
 * (hi*hi + hi*lo + lo*hi), f32 accumulate: ~2^-16 relative operand error instead of bf16's 2^-8. */
int ctcn_gemm(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
              float *C, int ldc, float beta, int precision, void *ws, size_t ws_bytes, void *stream);

/* out[b,a,c] = in[a,b,c]  (x.transpose(0,1), model_ctc.py:175) */
int ctcn_transpose01(const float *in, float *out, int A, int B, int C, void *stream);
/* materialise any <=4-D strided view: out contiguous [d0][d1][d2][d3] = in[i0*s0+i1*s1+i2*s2+i3*s3] (element strides);
 * the .contiguous() calls of CTC_Model.forward (model_ctc.py:153,158) */
int ctcn_copy_strided4(const float *in, float *out, int d0, int d1, int d2, int d3, size_t s0, size_t s1, size_t s2,
                       size_t s3, void *stream);
/* nn.ReLU when it is not fused into the BatchNorm apply pass (LayerCNN with batch_norm=False, model_ctc.py:64) */
int ctcn_relu_fwd(const float *x, float *y, size_t n, void *stream);
int ctcn_relu_bwd(const float *y, const float *dy, float *dx, size_t n, void *stream);
/* nn.MaxPool2d(pooling_size) of LayerCNN (model_ctc.py:52-53,64-65): kernel = stride = (kh, kw), no padding, floor; x / dx are
 * `planes` (= B*C) images of Hi x Wi, y / dy / arg of (Hi/kh) x (Wi/kw).  `arg` (one byte per output) keeps the winner's offset
 * ki*kw + kj inside its window for the backward pass (first maximum wins, NaN replaces anything: torch's scan rule). kh*kw <= 256. */
int ctcn_maxpool2d_fwd(const float *x, float *y, unsigned char *arg, size_t planes, int Hi, int Wi, int kh, int kw, void *stream);
int ctcn_maxpool2d_bwd(const float *dy, const unsigned char *arg, float *dx, size_t planes, int Hi, int Wi, int kh, int kw, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Recurrent layer, bias-free, 1 layer, 1 or 2 directions, zero initial state, no packing/masking.
 * replaces: rnn_type(input_size, hidden_size, bidirectional, bias=False)(x)  (model_ctc.py:24-25,33)
 *   x (T,B,I); w_ih[d] (G*H,I); w_hh[d] (G*H,H); y (T,B,dirs*H) = [fwd | rev]
 *   gates  (T,B,dirs,G*H): reserve; fwd leaves the saved activations, bwd overwrites it with d(pre-act)
 *   aux    (T,B,dirs,H)  : reserve; LSTM cell state c / GRU W_hn*h ; unused (may be NULL) for TANH
 * G = 4 (LSTM) / 3 (GRU) / 1 (TANH).  H must be a multiple of 4. */
size_t ctcn_rnn_scratch_bytes(int cell, int B, int H, int dirs);
int ctcn_rnn_fwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                 const float *w_hh0, const float *w_ih1, const float *w_hh1, float *y, float *gates, float *aux,
                 int precision, void *ws, size_t ws_bytes, void *stream);
/* ctcn_rnn_fwd followed by the layer's inverted dropout (BatchRNN: rnn -> nn.Dropout, timit/models/model_ctc.py:33-34): y as above (the
 * backward pass needs it) and y_drop = ctcn_dropout(y, p, seed, offset), bit for bit.  Where the tagged-gather recurrence applies the
 * dropped values are stored by the recurrence itself (option "rnn_fused_dropout" = 1, the default: Philox in the item waves' idle time,
 * no pass over y afterwards); everywhere else the dropout kernel runs behind the recurrence. */
int ctcn_rnn_fwd_dropout(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                         const float *w_ih1, const float *w_hh1, float *y, float *gates, float *aux, float *y_drop, float p, uint64_t seed,
                         uint64_t offset, int precision, void *ws, size_t ws_bytes, void *stream);
/* dy (T,B,dirs*H); dx (T,B,I) or NULL; dw_* same shapes as w_*; beta_w: 0 overwrite / 1 accumulate into dw.
 * scratch: ctcn_rnn_scratch_bytes() bytes (transposed W_hh + carried dh/dc state). */
int ctcn_rnn_bwd(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0,
                 const float *w_hh0, const float *w_ih1, const float *w_hh1, const float *y, float *gates,
                 float *aux, const float *dy, float *dx, float *dw_ih0, float *dw_hh0, float *dw_ih1,
                 float *dw_hh1, float beta_w, int precision, void *scratch, void *ws, size_t ws_bytes,
                 void *stream);
/* ctcn_rnn_bwd for a layer whose output went through ctcn_rnn_fwd_dropout: dy is the gradient of the DROPPED output, and the layer sees
 * ctcn_dropout(dy, p, seed, offset) -- bit for bit.  The scatter recurrence applies the keep mask as it consumes dy (option
 * "rnn_fused_dropout"); every other path first runs the dropout kernel into dy_tmp (T*B*dirs*H floats) and reads that. */
int ctcn_rnn_bwd_dropout(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                         const float *w_ih1, const float *w_hh1, const float *y, float *gates, float *aux, const float *dy, float *dx,
                         float *dw_ih0, float *dw_hh0, float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                         size_t ws_bytes, void *stream, float p, uint64_t seed, uint64_t offset, float *dy_tmp);
/* Per-call extras of the recurrent layer (ctcn_rnn_fwd_ex / ctcn_rnn_bwd_ex).  The library keeps NO state between calls (round 3: the
 * one-shot setters ctcn_set_prelaunch_event / ctcn_set_fwd_overlap and the thread-local hand-over of the _dropout entry points are gone):
 * everything a call needs beyond its tensors is in this struct, owned by the caller, read during the call only.  Zero-initialise it;
 * a NULL pointer = all zero = the plain ctcn_rnn_fwd / ctcn_rnn_bwd.  Two models on two streams (or two host threads) share nothing but
 * the option table (ctcn_set_option: process-wide tuning switches, read-only during calls) and, unless `status` is given, the process
 * default status word. */
typedef struct ctcn_rnn_call {
  /* dropout of the layer output (BatchRNN: rnn -> nn.Dropout, model_ctc.py:33-34).  fwd: y_drop != NULL asks for y_drop =
   * ctcn_dropout(y, drop_p, drop_seed, drop_offset) beside y.  bwd: dy_tmp != NULL says dy is the gradient of the DROPPED output (same
   * p / seed / offset); dy_tmp (T*B*dirs*H floats) is scratch for the paths that run the dropout kernel first. */
  float drop_p;
  uint64_t drop_seed, drop_offset;
  float *y_drop;
  float *dy_tmp;
  /* fwd: pipeline the input projection with the persistent recurrence.  The library projects the first pair of time chunks on the call's
   * stream, records side_event (hipEvent_t) right before the recurrence launch and issues the remaining chunk GEMMs on side_stream behind
   * that event, restricted to the XCDs of xcd_allow (bit x = XCD x: the ones the recurrence leaves idle), each pair followed by a counter
   * update the recurrence checks when it enters a new chunk.  side_ws / side_ws_bytes: workspace of the side-stream GEMMs.  Ignored (whole
   * projection first) where the tagged-gather recurrence does not apply.  The caller joins side_stream before it reuses x or side_ws. */
  void *side_stream, *side_event, *side_ws;
  size_t side_ws_bytes;
  unsigned xcd_allow;
  /* bwd: hipEvent_t recorded on the call's stream immediately before the recurrence is launched, behind the call's own preparatory
   * memsets / transposes; work meant to run next to that recurrence on another stream (ctcn_rnn_bwd_weights of the layer above) waits
   * for it. */
  void *prelaunch_event;
  /* device int32 the persistent kernels of THIS call set on a hand-off timeout (sticky); NULL: the process default of
   * ctcn_set_status_buffer (which may be NULL too: timeouts then only poison the output). */
  int *status;
  /* out (the one field the library writes): where the name of the recurrent kernel THIS call launched is stored (a string literal:
   * "rnn_fwd_tagged", "rnn_fwd_persist", "rnn_fwd_step", "rnn_bwd_scatter2", "rnn_bwd_scatter", "rnn_bwd_persist", "rnn_bwd_step"); NULL: not
   * wanted.  Per call, unlike the process-wide ctcn_rnn_last_kernel (last writer of any thread wins): what the host's batch-chunk decision
   * reads. */
  const char **launched;
} ctcn_rnn_call;
int ctcn_rnn_fwd_ex(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                    const float *w_ih1, const float *w_hh1, float *y, float *gates, float *aux, int precision, void *ws, size_t ws_bytes,
                    void *stream, const ctcn_rnn_call *call);
int ctcn_rnn_bwd_ex(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *w_ih0, const float *w_hh0,
                    const float *w_ih1, const float *w_hh1, const float *y, float *gates, float *aux, const float *dy, float *dx,
                    float *dw_ih0, float *dw_hh0, float *dw_ih1, float *dw_hh1, float beta_w, int precision, void *scratch, void *ws,
                    size_t ws_bytes, void *stream, const ctcn_rnn_call *call);
/* ctcn_rnn_bwd with dw_ih0 == dw_hh0 == NULL runs the recurrence and dx only and leaves d(pre-activation) in gates
 * (and aux for the GRU n-gate); this call then produces the weight gradients from it: dW_ih = da^T x, dW_hh = da^T h_prev.
 * It has no consumer inside the backward pass, so the host side issues it on a second stream next to the NEXT layer's
 * persistent recurrence; xcd_allow != 0 (bit x = XCD x) keeps the bf16x3 GEMM workgroups of precision 1 on those XCDs,
 * i.e. off the ones the recurrence occupies.  ws must not be shared with calls running concurrently on another stream. */
int ctcn_rnn_bwd_weights(int cell, int T, int B, int I, int H, int dirs, const float *x, const float *y,
                         const float *gates, const float *aux, float *dw_ih0, float *dw_hh0, float *dw_ih1,
                         float *dw_hh1, float beta_w, int precision, unsigned xcd_allow, void *ws, size_t ws_bytes,
                         void *stream);

/* ---------------------------------------------------------------------------------------------------
 * BatchNorm, statistics per channel over (outer x inner) elements; x viewed as (outer, C, inner).
 *   inner==1 : BatchNorm1d over (T*B) rows   (model_ctc.py:29-32, :136,165-166)
 *   inner>1  : BatchNorm2d on NCHW            (model_ctc.py:47,63)
 * train fwd: y = gamma*(x-mean)*rstd+beta; saves mean,rstd (C each); updates running stats in place
 * (momentum 0.1 semantics: rm = (1-mom)*rm + mom*mean; rv uses the unbiased variance).
 * relu!=0 fuses nn.ReLU (model_ctc.py:64) into the apply pass and its mask into the backward pass.
 * ws: >= ctcn_bn_ws_bytes(outer, C, inner) (0 for dims that are not positive). */
size_t ctcn_bn_ws_bytes(int outer, int C, int inner);
int ctcn_bn_fwd_train(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float *save_mean, float *save_rstd, int outer, int C, int inner,
                      float eps, float momentum, int relu, void *ws, size_t ws_bytes, void *stream,
                      long long *num_batches_tracked /* device int64 or NULL: += 1 by the statistics kernel (nn.BatchNorm's counter, no launch of its own) */);
/* Synchronised BatchNorm for data-parallel training (statistics over the global batch): every rank calls
 * _sums (per-channel fp64 [C][2]: sum x, sum x^2 / sum dy', sum dy'*xhat), the host all-reduces the 2*C doubles, and
 * _finish normalises with count_total = global number of elements per channel.  In the backward finish local_sums feed
 * dgamma / dbeta (the gradient all-reduce adds the ranks up), global_sums feed dx. */
int ctcn_bn_fwd_sums(const float *x, double *sums, int outer, int C, int inner, void *ws, size_t ws_bytes, void *stream);
int ctcn_bn_fwd_finish(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, float *save_mean, float *save_rstd, const double *sums, double count_total,
                       int outer, int C, int inner, float eps, float momentum, int relu, void *stream,
                       long long *num_batches_tracked /* as in ctcn_bn_fwd_train */);
int ctcn_bn_bwd_sums(const float *x, const float *y, const float *dy, const float *save_mean, const float *save_rstd,
                     double *sums, int outer, int C, int inner, int relu, void *ws, size_t ws_bytes, void *stream);
int ctcn_bn_bwd_finish(const float *x, const float *y, const float *dy, const float *gamma, const float *save_mean,
                       const float *save_rstd, float *dx, float *dgamma, float *dbeta, const double *local_sums,
                       const double *global_sums, double count_total, int outer, int C, int inner, int relu,
                       float beta_acc, void *ws, size_t ws_bytes, void *stream);
int ctcn_bn_fwd_eval(const float *x, float *y, const float *gamma, const float *beta, const float *running_mean,
                     const float *running_var, int outer, int C, int inner, float eps, int relu, void *stream);
/* y is only read when relu!=0 (mask = y>0). dx may alias dy. */
int ctcn_bn_bwd(const float *x, const float *y, const float *dy, const float *gamma, const float *save_mean,
                const float *save_rstd, float *dx, float *dgamma, float *dbeta, int outer, int C, int inner,
                int relu, float beta_acc, void *ws, size_t ws_bytes, void *stream);
/* BatchNorm (training statistics) + ReLU + inverted dropout in ONE apply pass, and its backward (round 5).
 * replaces: LayerCNN.forward's `x = self.batch_norm(x); x = self.activation(x); ...; x = self.dropout(x)` (model_ctc.py:62-67, no pooling between
 * them) and its autograd.  y_drop = ctcn_dropout(relu(bn(x)), p, seed, offset) bit for bit (same expression per element, same Philox words: word
 * i & 3 of group offset + (i >> 2)); the un-dropped activation is not stored.  ctcn_bn_bwd_dropout takes dy_drop = the gradient of the DROPPED
 * output, regenerates the keep mask from the counters and recomputes the ReLU mask from x (it needs beta for that): dx, dgamma, dbeta as
 * ctcn_dropout backward followed by ctcn_bn_bwd would give them, without the two passes over the dropped / un-dropped tensors.  0 < p < 1. */
int ctcn_bn_fwd_train_dropout(const float *x, float *y_drop, const float *gamma, const float *beta, float *running_mean, float *running_var,
                              float *save_mean, float *save_rstd, int outer, int C, int inner, float eps, float momentum, int relu, void *ws,
                              size_t ws_bytes, void *stream, long long *num_batches_tracked, float p, uint64_t seed, uint64_t offset);
int ctcn_bn_bwd_dropout(const float *x, const float *dy_drop, const float *gamma, const float *beta, const float *save_mean, const float *save_rstd,
                        float *dx, float *dgamma, float *dbeta, int outer, int C, int inner, int relu, float beta_acc, void *ws, size_t ws_bytes,
                        void *stream, float p, uint64_t seed, uint64_t offset);

/* ---------------------------------------------------------------------------------------------------
 * Dropout (inverted, Philox4x32-10 counter RNG keyed by (seed, offset + element index)).
 * replaces: nn.Dropout (model_ctc.py:26,34,58,67).  bwd regenerates the mask from (seed, offset). */
int ctcn_dropout(const float *x, float *y, size_t n, float p, uint64_t seed, uint64_t offset, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Conv2d (bias) NCHW, direct; replaces nn.Conv2d in LayerCNN (model_ctc.py:46,61).
 * x (B,Ci,Hi,Wi) w (Co,Ci,kh,kw) y (B,Co,Ho,Wo), Ho=(Hi+2ph-kh)/sh+1.
 * Default: MFMA implicit GEMMs (option "conv_mfma") for Co <= 64, kh*kw <= 25 and LDS images within 158 KB; everything else on the direct
 * kernels, which (round 6) run a filter bank beyond their 60-KB LDS budget as (output-channel range, input-channel range) slices -- the
 * reference's own example front-end, model_ctc.py:232-233: (1,32,(3,41)) = 123 taps and (32,32,(3,21)) = 258 KB of filters -- slowly, not
 * with an error.  CTCN_EUNSUPPORTED only beyond ~900 taps per (output, input) channel pair. */
size_t ctcn_conv2d_ws_bytes(int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw);
int ctcn_conv2d_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Ci, int Hi, int Wi,
                    int Co, int kh, int kw, int sh, int sw, int ph, int pw, void *stream);
int ctcn_conv2d_bwd(const float *x, const float *w, const float *dy, float *dx /*or NULL*/, float *dw,
                    float *dbias, int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph,
                    int pw, float beta_acc, void *ws, size_t ws_bytes, void *stream);
/* (B,C,T,F) <-> (T,B,C*F), feature index c*F+f   (model_ctc.py:153-158) */
int ctcn_bctf_to_tbcf(const float *in, float *out, int B, int C, int T, int F, void *stream);
int ctcn_tbcf_to_bctf(const float *in, float *out, int B, int C, int T, int F, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * log_softmax over the last dim (+ arg-max, lowest index on ties) and its backward.
 * replaces: nn.LogSoftmax (model_ctc.py:140,168,181); torch.max(out,-1) (train_ctc.py:51, ctcDecoder.py:163)
 * argmax (rows) int32 may be NULL. */
int ctcn_log_softmax_fwd(const float *logits, float *lp, int32_t *argmax, int rows, int V, void *stream);
int ctcn_log_softmax_bwd(const float *lp, const float *dlp, float *dlogits, int rows, int V, void *stream);
int ctcn_argmax(const float *lp, int32_t *argmax, int rows, int V, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * CTC loss (blank = 0, zero_infinity = False); replaces nn.CTCLoss(reduction='sum') forward/backward
 * (train_ctc.py:144,47-48,63).  lp (T,B,V) log-probs; targets (B,Lmax) int64 zero-padded;
 * in_len/tgt_len (B) int64.  alpha: (T,B,2*Lmax+1) reserve.  nll (B): per-utterance negative
 * log-likelihood (+inf when no alignment exists).  Lmax <= 2047 (sixteen lattice states per thread), else CTCN_EUNSUPPORTED.
 * bwd: grad_lp[t,b,c] = gscale[0] * (exp(lp) - exp(logsum_{s:ext(s)=c}(alpha+beta) + nll - lp)) for
 * t < in_len[b], 0 beyond; gscale is a 1-element DEVICE float (the upstream gradient of the summed loss).
 * alpha is overwritten with alpha+beta. */
int ctcn_ctc_fwd(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len,
                 float *alpha, float *nll, int T, int B, int V, int Lmax, void *stream);
int ctcn_ctc_bwd(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len,
                 float *alpha, const float *nll, const float *gscale, float *grad_lp, int T, int B, int V,
                 int Lmax, void *stream);
/* The same loss when a gradient will be wanted (training, train_ctc.py:47 followed by :63): alpha AND beta, each into its
 * own (T,B,2*Lmax+1) lattice, in ONE launch -- the two passes are independent chains of T dependent steps, so side by
 * side they cost one chain -- then ctcn_ctc_grad adds them on the fly (the same single f32 add as ctcn_ctc_bwd's
 * in-place pass: bit-identical gradients) and leaves both lattices untouched. */
int ctcn_ctc_fwd_both(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len,
                      float *alpha, float *beta, float *nll, int T, int B, int V, int Lmax, void *stream);
int ctcn_ctc_grad(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len,
                  const float *alpha, const float *beta, const float *nll, const float *gscale, float *grad_lp,
                  int T, int B, int V, int Lmax, void *stream);
/* out[0] = sum_b nll[b]  (deterministic order) */
int ctcn_sum_f32(const float *x, float *out, int n, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Adam with L2-coupled weight decay over one flat buffer; replaces torch.optim.Adam.step
 * (train_ctc.py:145,65).  step is the 1-based step count. */
int ctcn_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Data-parallel exchange step: SUM all-reduce of the flat float32 gradient buffer (or a slice of it) over RCCL / xGMI,
 * one communicator per rank.  The reference is single-device (train_ctc.py:63-65 loss.backward(); optimizer.step()); utterance-
 * sharded data parallelism inserts this one collective between the two calls.  librccl.so is opened at run time.
 *   rank 0 calls ctcn_comm_unique_id (128 bytes) and hands the blob to the other ranks by any host channel; every rank then
 *   calls ctcn_comm_init (collective: blocks until all `world` ranks arrived).  The communicator is the only object the library
 *   owns; ctcn_comm_allreduce_sum_f32 is enqueued on `stream` (in place) and returns without synchronising.  A handle that ctcn_comm_init
 *   did not hand out, or that ctcn_comm_destroy has taken back, is answered with CTCN_EINVAL (the library keeps the list of live handles)
 *   instead of reaching RCCL, which would dereference it. */
int ctcn_comm_unique_id(void *id128);
int ctcn_comm_init(const void *id128, int rank, int world, void **comm);
int ctcn_comm_allreduce_sum_f32(void *comm, float *buf, size_t n, void *stream);
int ctcn_comm_destroy(void *comm);

/* ---------------------------------------------------------------------------------------------------
 * Greedy decode: collapse of arg-max paths (drop blank, drop frame-to-frame repeats, first lens[b]
 * frames); replaces GreedyDecoder.decode / CTC_Model.compute_wer inner loops
 * (ctcDecoder.py:152-166,80-92; model_ctc.py:190-199).  idx int32, element (t,b) at idx[t*stride_t+b*stride_b]
 * (time-major (T,B): stride_t=B, stride_b=1; batch-major (B,T): stride_t=1, stride_b=T);
 * out_ids (B,T) int32; out_len (B) int32. */
int ctcn_greedy_collapse(const int32_t *idx, size_t stride_t, size_t stride_b, const int32_t *lens, int32_t *out_ids,
                         int32_t *out_len, int T, int B, int blank, void *stream);

/* Levenshtein distance per utterance between collapsed predictions a (B,lda) int32 / a_len (B) int32 and labels
 * b (B,ldb) int64 / b_len (B) int64 -> out (B) int32; replaces editdistance.eval in CTC_Model.compute_wer
 * (model_ctc.py:200).  max_b_len >= max(b_len). */
int ctcn_edit_distance(const int32_t *a, const int32_t *a_len, const int64_t *b, const int64_t *b_len, int32_t *out, int B,
                       int lda, int ldb, int max_b_len, void *stream);
/* (loss, sum of dist[0..B), sum of tgt_len[0..B), *status or 0) as four doubles in out4 (device memory): the per-step statistics of the
 * reference's run_epoch (loss.item(), total_wer's numerator and denominator, timit/steps/train_ctc.py:55-60) plus the sticky hand-off
 * status, gathered in one launch so that the host can fetch them with one small copy a step later. */
int ctcn_step_stats(const float *loss, const int32_t *dist, const int64_t *tgt_len, int B, const int32_t *status, double *out4, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * CTC prefix beam search with bigram LM; replaces BeamDecoder.decode -> ctcBeamSearch.decode
 * (ctcDecoder.py:181-192; BeamSearch.py:73-153).  One workgroup per utterance.
 *   x (T,B,V) float32: log-probs (input_is_prob==0, exp taken on device) or probabilities exp(lp)
 *   lens (B) int32; lm ((V+1)*(V+1)) float64 ln-probs, row V = '<s>', column V = '</s>'
 *   out_ids (B,T) int32, out_len (B) int32, out_score (B) float64 (length-normalised prTotal)
 *   status (B) int32: 0 ok, 1 best labelling empty (reference raises IndexError), 2 log(0) (ValueError), 3 node table of the workspace
 *   exhausted, 4 internal hand-over between the waves of the search timed out (a bug, never seen; outputs zeroed for 2..4)
 *   ws: >= ctcn_beam_ws_bytes(T,B,V,W)
 *   1 <= W <= 1 024 (round 6; was 256): the reference takes any beam_width (ctcDecoder.py:170, default 200).  W <= 60 with W*V <= 3 328 runs
 *   the restructured search, everything else the generic kernel, whose beam state sits in dynamic LDS sized for W (146 KB at W = 1 024);
 *   wider beams: CTCN_EUNSUPPORTED. */
size_t ctcn_beam_ws_bytes(int T, int B, int V, int W);
int ctcn_beam_decode(const float *x, int input_is_prob, const int32_t *lens, const double *lm, double alpha,
                     int W, int blank, int32_t *out_ids, int32_t *out_len, double *out_score, int32_t *status,
                     int T, int B, int V, void *ws, size_t ws_bytes, void *stream);
/* The same search returning the `nbest` (1 <= nbest <= W) best labellings of every utterance: the first nbest entries of the final
 * `last.sort()` of BeamSearch.py:150, of which the reference keeps element [0] ("optional n-best output") -- a stable
 * descending sort by the length-normalised score, entry 0 = what ctcn_beam_decode returns.
 *   out_ids (B,nbest,T) int32, out_len (B,nbest) int32, out_score (B,nbest) float64; out_count (B) int32 or NULL: labellings actually
 *   returned for the utterance (the final beam can hold fewer than nbest; the remaining entries have length 0 and score 0) */
int ctcn_beam_decode_nbest(const float *x, int input_is_prob, const int32_t *lens, const double *lm, double alpha,
                           int W, int blank, int nbest, int32_t *out_ids, int32_t *out_len, double *out_score,
                           int32_t *out_count, int32_t *status, int T, int B, int V, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Diagnostics (not on the compute path).
 * ctcn_diag_squat: `wgs_per_xcd` workgroups of `threads` threads (+ `lds_bytes` of LDS each) on every XCD that only hold their CU
 * slots for `usec` microseconds (clock-bounded, <= 5 s) -- what an RCCL kernel waiting for a slow peer looks like to the rest of the
 * chip.  The data-parallel co-residency contract of the persistent recurrences is tested against it.
 * ctcn_rnn_last_kernel: name of the recurrent kernel the most recent ctcn_rnn_fwd (which = 0) / ctcn_rnn_bwd (which = 1) of this
 * PROCESS launched ("rnn_fwd_tagged", "rnn_fwd_persist", "rnn_fwd_step", "rnn_bwd_scatter", "rnn_bwd_scatter2", "rnn_bwd_persist",
 * "rnn_bwd_step"; "" before the first call).  The one piece of state the library keeps between compute calls: two atomic pointers to
 * string literals (relaxed; with several calling threads the last writer wins -- the backward pass of autograd runs on another thread
 * than the reader, which is why it is not thread-local), read by nothing on the compute path. */
int ctcn_diag_squat(int wgs_per_xcd, int threads, int lds_bytes, unsigned usec, void *stream);
/* ctcn_diag_pipeline_chunks: the plan of the projection pipeline (ctcn_rnn_call.side_stream of ctcn_rnn_fwd_ex) for a layer shape on a device
 * of `xcds` XCDs and `cus` CUs with the idle XCDs `xcd_allow`: the number of time chunks (even, 8..24) it would be pipelined with, 0 = not
 * pipelined (the one-round / 72-steps / throughput / free-CU rules of rnn.hip), -1 = bad arguments.  Pure arithmetic, no device needed. */
int ctcn_diag_pipeline_chunks(int cell, int T, int B, int I, int H, int dirs, int xcds, int cus, unsigned xcd_allow);
const char *ctcn_rnn_last_kernel(int which);

/* ---- host end of the decoders (hostjoin.hip; no kernel, no HIP call: works without a GPU) -------------------------------------------------
 * replaces: `' '.join(self.classes[k] for k in labelling)` (BeamSearch.py:152-153; ctcDecoder.py:60-118 for the greedy decoder), once per
 * utterance of a decoded batch.  ids: B rows of `row_stride` int32 label ids (the pinned copy of ctcn_beam_decode's out_ids), lens[b] valid
 * per row; words / word_off[V] / word_len[V]: the vocabulary's UTF-8 bytes back to back, followed by 16 readable bytes, the byte offset and
 * the byte length of every word (word_len[k] < 0: id k has no word); sep: the byte between two words (0: none).  Writes row b's string to
 * out[out_off[b] .. out_off[b + 1]) and returns out_off[B]; CTCN_EINVAL on bad arguments, CTCN_EWORKSPACE when out_cap is too small (result + 17 bytes), -(16 + k) for an id k outside the vocabulary (the KeyError /
 * IndexError of the Python expression). */
long long ctcn_join_tokens(const int32_t *ids, long long row_stride, const int32_t *lens, int B, const char *words, const int32_t *word_off,
                           const int32_t *word_len, int V, int sep, char *out, long long out_cap, long long *out_off);
/* ctcn_levenshtein: unit-cost edit distance of two int32 sequences (the code points of two strings, or word ids), host code.
 * replaces: Decoder._edit_distance (ctcDecoder.py:131-150; cer :127-129 and wer :118-125 call it once per decoded utterance).  -1 on bad
 * arguments. */
long long ctcn_levenshtein(const int32_t *a, long long na, const int32_t *b, long long nb);

#ifdef __cplusplus
}
#endif
#endif
