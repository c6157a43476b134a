/*
 * pfnl_hip.h — C-ABI of libpfnl_hip.so: the MI355X (gfx950) implementation of PFNL's
 * forward/inference hot path.
 *
 * The reference (psychopa4/PFNL) is pure Python on TensorFlow 1.12 and has no FFI/operator
 * interface of its own; the seam this library replaces is the TF runtime underneath
 *   - PFNL.forward                 (reference model/pfnl.py:39-80)   -> pfnl_forward
 *   - utils.NonLocalBlock          (reference utils.py:18-71)        -> pfnl_op_nonlocal
 *   - tf.layers.Conv2D instances   (reference model/pfnl.py:48-53)   -> pfnl_op_conv2d
 *   - tf.image.resize_images(..,2) (reference model/pfnl.py:63)      -> pfnl_op_bicubic
 *   - tf.train.Saver.restore       (reference model/base_model.py:231-243): variables arrive
 *     through pfnl_set_weight under their TF names (nlvsr/conv0/kernel ...), HWIO float32.
 *   - sess.run(SR_test, feed_dict) (reference model/pfnl.py:252,309) -> pfnl_forward with host
 *     pointers (H2D/D2H inside, like the reference's timing includes).
 *
 * Conventions: every entry point is extern "C", takes plain pointers and sizes, returns 0 on
 * success or a negative pfnl_status; the message of the last failure on the calling thread is
 * available from pfnl_last_error().  A handle is bound to one device and is not re-entrant;
 * distinct handles are independent.  "device pointer" = memory accessible from that device
 * (e.g. a torch-ROCm tensor's data_ptr()); "stream" = a hipStream_t passed as void* (NULL = the
 * handle's own stream).  All tensors are float32 (the *_bf16 single-op hooks alone take bfloat16 bit patterns as
 * uint16_t), contiguous, NHWC-style as in the reference:
 *   input  [B, T, H, W, 3]      values nominally in [0,1]
 *   output [B, 1, s*H, s*W, 3]  not clipped (the harness clips, reference model/pfnl.py:255-257)
 */
#ifndef PFNL_HIP_H
#define PFNL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfnl_handle pfnl_handle;

typedef enum pfnl_status {
    PFNL_OK = 0,
    PFNL_ERR_INVALID = -1,     /* bad argument / shape (odd H or W, unknown tensor name ...) */
    PFNL_ERR_STATE = -2,       /* e.g. forward before finalize_weights, missing weights        */
    PFNL_ERR_HIP = -3,         /* a HIP runtime call failed                                    */
    PFNL_ERR_NOMEM = -4,
    PFNL_ERR_NODEVICE = -5,    /* no gfx950-capable device visible                             */
    PFNL_ERR_COMM = -6,        /* an RCCL call failed / RCCL could not be loaded               */
    PFNL_ERR_RANGE = -7        /* a forward left the range of its binary16-operand kernels (see "strict_fp32"): pfnl_sync after a
                                  device-pointer forward; pfnl_forward itself with host pointers under precision=bf16 (no f32 re-run there) */
} pfnl_status;

/* Mirrors the constants hard-coded in the reference (model/pfnl.py:21-23, 40-43). */
typedef struct pfnl_config {
    int32_t num_frames;   /* T: 7 (reference); 3 and 5 also supported                    */
    int32_t scale;        /* 4 (reference) or 2 (build-defined tail)      */
    int32_t mf;           /* trunk width, must be 64                                     */
    int32_t num_block;    /* progressive-fusion blocks, 20 in the reference (>=0)        */
    int32_t device_id;    /* HIP device ordinal                                          */
    int32_t reserved[3];  /* must be 0                                                   */
} pfnl_config;

const char* pfnl_last_error(void);
int pfnl_version(void);                        /* ABI version, currently 3 (round 6: + pfnl_plan, pfnl_get_option, the _sf0 op hooks; - pfnl_op_conv3x3_wsplit) */
int pfnl_device_count(int* count);

/* ---- model lifetime ---------------------------------------------------------------------- */
int pfnl_create(const pfnl_config* cfg, pfnl_handle** out);
int pfnl_destroy(pfnl_handle* h);

/* Replaces tf.train.Saver.restore (reference model/base_model.py:231-243).  `tf_name` is the TF
 * variable name ("nlvsr/conv1_3/kernel", "nlvsr/nlblock_0/g/g/bias" ...; reference
 * model/pfnl.py:47-53, utils.py:23-26,66-67).  `host` is float32 HWIO (kernels) / [Cout] (biases)
 * and stays owned by the caller; the library keeps its own repacked device copies. */
int pfnl_set_weight(pfnl_handle* h, const char* tf_name, const float* host,
                    const int64_t* shape, int rank);
/* Optional: "nlvsr/nlblock_0/theta/theta/{kernel,bias}" and ".../phi/phi/{kernel,bias}" ([1,1,C,C] / [C], C = 12T) switch
 * the non-local block to the embedded-Gaussian form (reference utils.py:31-42 with nltype 0: theta = conv(x), phi = conv(x));
 * all four or none.  PFNL itself calls the block with nltype 1 (model/pfnl.py:58), where these scopes hold no variables.
 * Number of variables still missing (0 = complete; the optional ones never count). */
int pfnl_missing_weights(pfnl_handle* h, int* count);
/* Repack + upload: splits conv2_i into its shared-`base` half and per-frame half, folds
 * Wg*Ww of the non-local block, chunks the implicit-GEMM weights. */
int pfnl_finalize_weights(pfnl_handle* h);

/* Tuning knobs (all parity-tested):  key "conv3x3" =
 *   "auto"          (default) "split16" when a launch has at least 200 tiles of 8x32 pixels, "winograd" below that;
 *   "winograd"      fused Winograd F(2x2,3x3) on f32 MFMA (2.25x fewer multiplies), persistent
 *                   wave-specialised kernel: matrix waves + helper waves (conv_wino_ws.hip);
 *   "winograd_tile" same maths, one 4-wave workgroup per tile (conv_wino.hip);
 *   "direct"        implicit-GEMM f32 MFMA (conv_mfma.hip);
 *   "split16"       direct 3x3 on the f16 matrix pipe with exactly split fp32 operands (3 f16 MFMAs per product block,
 *                   fp32 accumulation, >= 22 mantissa bits per product: conv_split16.hip);
 * key "strict_fp32" = "off" (default) | "on".  The default fp32 path computes on the f16 matrix pipe with exactly split operands
 *   (fp32 tensors, fp32 accumulation, >= 22 mantissa bits per product) and therefore has a DOMAIN the reference's fp32 kernels do
 *   not have: |activation|, |weight| < 65504, and inputs of the non-local block on a [0,1] scale (|x| < ~350).  Leaving it makes
 *   an operand inf and the result non-finite; the tail kernel flags that.  Host-pointer calls then redo the call on the f32-MFMA
 *   kernels before returning (pfnl_range_reruns counts them), so they cover the whole fp32 range; device-pointer calls are
 *   asynchronous: pfnl_sync returns PFNL_ERR_RANGE.  "on" (or env PFNL_STRICT_FP32=1) uses the f32-MFMA kernels throughout; weights
 *   beyond binary16's range select them by themselves at pfnl_finalize_weights.
 * key "small" = "auto" (default: the small-shape trunk kernels of conv_small.hip when a 3x3 launch has fewer than 200 tiles (256 until round 5) of 8x32
 *   pixels - conv2_i as one 128 -> 64 convolution) | "on" | "off".
 * key "small_c10" = "on" (default since round 4: TWO launches per progressive-fusion block at small shapes - the conv1_i launch also
 *   runs its finished tile through this frame's 64 x 64 slice of conv10_i and writes the partial sum; the conv2_i launch adds the T
 *   partials (+ bias, leaky-relu) on the way into LDS: its `base` source.  No 1x1 launch, no inter-workgroup exchange inside a
 *   launch, fixed summation order) | "off" (three launches, conv10_i on conv_small_1x1_kernel).
 * key "split16_sf" = "on" (default) | "off": with conv3x3 and conv1x1 on "split16", conv1_i and conv10_i write the split format
 *   (hi, lo' binary16 pairs: the MFMA operands themselves) and both halves of conv2_i read it by LDS-DMA (conv_sf.hip).
 * key "split16_chain" = "on" (default) | "off": with split16_sf, conv2_i is one launch - per (clip, tile) the shared half stays in
 *   registers as the initial C of the T per-frame tiles (no pb tensor, no addend reads, 20 launches fewer per forward).
 * key "split16_c10" = "on" (default) | "off": with split16_sf, conv1_i and conv10_i are one launch (conv3x3_c1c10_kernel): per (clip,
 *   tile) the T frame tiles of conv1_i leave as split-format lines through LDS, where conv10_i takes them as MFMA operands - inp1 is
 *   written once and never read back by a 1x1 launch (2 launches per progressive-fusion block with split16_chain).
 * key "split16_mid" = "auto" (default since round 5) | "off": the two-launch block deals out CHAINS - a workgroup takes the T frames of
 *   a (clip, 8x32-pixel tile) - so a launch with fewer chains than CUs leaves most of the chip idle (one clip of 128x128: 64 chains).
 *   Below 136 chains (and above the small-shape rule's 200 tiles; both scale with the device's CU count / 256) the block runs as four launches that deal out single tiles:
 *   conv1_i, conv10_i, the shared half of conv2_i, the per-frame half in flat order (same arithmetic as split16_c10=off +
 *   split16_chain=off; tools/precision_ladder.py has the crossover).  Only with conv3x3=auto and both of those options on.
 * key "split16_splitchains" = "auto" (default since round 6) | "off": the two-launch block deals out whole (clip, tile) chains, so a batch whose
 *   chains are not a whole number of rounds of the grid (256 workgroups) pays a whole extra chain for its last, partial round (5 clips of
 *   128x128 = 1.25 rounds took 1.57x the time of 4).  When that round has at most grid / 2 chains, they are cut by FRAMES into parts (one per
 *   workgroup): conv3x3_c1c10_kernel leaves a part's share of conv10_i's sum as raw fp32 and c10_finalize_kernel adds the parts up in fixed
 *   order (+ leaky-relu, split format); the chain kernel recomputes the shared half per part.  Deterministic; batches that are whole rounds
 *   (configs[1]) run exactly as before (bit-identical); the clips of the whole rounds keep their bits, the cut chains differ from the uncut
 *   launch in summation order (oracle tolerance).  pfnl_plan: "chain2_split", whole_chains / split_parts / part_frames.  The bf16 trunk's two chained
 *   launches and convmerge1's accumulating launch are cut the same way ("bf16_3_split").
 * key "split16_sf0" = "off" (default) | "on" (round 6): in the two-launch block ("chain2") the chain kernel writes the block's output - the
 *   next block's inp0 - a second time in the split format, and conv3x3_c1c10_kernel takes its halo from that copy by LDS-DMA in operand form
 *   (no fp32 -> binary16 split on the VALU, no register-staged commit).  Same operands in the same order: BIT-IDENTICAL results (tested).
 *   Measured slower on the same box (configs[1]: 4.86 against 4.45 ms per step - the copy costs the chain kernel 22 us per launch, the
 *   DMA halo saves conv1_i + conv10_i 0.6 us: DESIGN.md R6.1), so it is not the default; kept as the switch that reproduces the measurement.
 * key "conv1x1" = "split16" (default: streaming kernel on the f16 pipe, exactly split fp32 operands) | "stream" (streaming f32-MFMA
 *                 kernel) | "tiled" (conv_mfma.hip).
 * key "nonlocal" (fp32 precision only) = "auto" (default: "split16" from 1024 keys, "f32" below) | "f32" (f32 MFMA, nonlocal.hip) |
 *                 "split16" (f16 MFMA with exactly split operands: fp32-level accuracy at a third of the time, nonlocal_f16.hip).
 * key "graph"   = "off" (default) | "on" (every shape is captured into a hipGraph on its second call and replayed
 *                 between the staging buffers) | "auto" (only shapes with frames*H*W <= 65536 pixels).  Measured: no
 *                 gain - the small shapes are bound by per-kernel latency, not by launch gaps.
 * key "conv2"   = "grouped" (default with conv3x3=winograd: one launch per block, shared half in LDS; taken when there
 *                 are at least 224 (clip, 4x32-pixel tile) groups, like the accumulating convmerge1) | "split".
 * key "precision" = "fp32" (default: the reference's arithmetic, |dPSNR| <= 0.01 dB) | "bf16" (BASELINE.json configs[3]:
 *                 the 20 progressive-fusion blocks keep their activations and weights in bf16 and accumulate in fp32 on
 *                 bf16 MFMA (conv_bf16.hip); the non-local block, its logits, conv0's arithmetic, convmerge1, the tail and
 *                 the bicubic skip stay fp32; the interface tensors stay float32.  Tolerance: DESIGN.md section 4).
 * key "nl_type" = "auto" (default: 0 when the theta / phi variables are loaded, else 1 = PFNL's own call, model/pfnl.py:58) | "0" |
 *                 "1" | "2": utils.NonLocalBlock's nltype (embedded Gaussian / Gaussian / dot product; 0 and 2 need theta / phi).
 * key "nl_sub_sample" = "1" (default, PFNL's call) | n: average-pool g and phi n x n on the space_to_depth grid (utils.py:27-28,35-36).
 *                 nl_type != 1 or nl_sub_sample > 1 run on the f32-MFMA kernel (nonlocal.hip) in both precisions.
 *                 ONE DIVERGENCE FROM THE REFERENCE AS WRITTEN, by design: utils.py:57-58 takes exp(S) and then divides by its row sum
 *                 without subtracting the row maximum, so a row whose sum exceeds fp32's range (N exp(s) > 3.4e38: at N = 4096 every
 *                 logit > 80.4, i.e. a 128x128 input that is >= 0.98-white throughout) comes out as P = 0 / inf = 0 and the block returns
 *                 its bias there.  Every kernel here runs the softmax as a streaming, max-subtracted recurrence: on those inputs it
 *                 returns the mathematically defined softmax (tested against the stabilised fp64 oracle), everywhere else the same
 *                 values as the reference's formula.  There is no "as written" switch: reproducing an overflow is not a feature.
 * key "bf16_conv10" = "fused" (default: conv10_i runs inside the conv1_i launch of the bf16 trunk) | "separate".
 * key "bf16_mfma" = "16" (default since round 6: the two chained 3x3 launches of the bf16 trunk issue v_mfma_f32_16x16x32_bf16 - the K = 32 shape costs 14 %
 *   less energy per FLOP on a package that sits on its power cap; 1080p 3.77 -> 3.58 ms) | "32" (the 32x32x16 form: same rounding points, another summation order).
 * key "split16_mfma" = "16" (default since round 6: the chain launch of conv2_i - fp32 path, whole rounds of at least one (clip, tile) chain per CU - issues
 *   v_mfma_f32_16x16x32_f16, conv3x3_sf_chain16_kernel; split chains, split16_sf0 and launches with fewer chains stay on the 32x32x16 kernel) | "32".
 *   pfnl_plan's mfma field says which ran.  Same operands, another summation order: results agree within rounding (observed 2e-7), not bit for bit.
 * key "bf16_nonlocal" = "f16" (the only value since round 4: the non-local block of precision=bf16 on the f16 matrix pipe with
 *                 binary16 operands, fp32 accumulation and softmax state - nonlocal_f16.hip, hi parts only; within 1e-3 of the
 *                 fp64 block on [0,1]-scale outputs, measured 1e-4 ... 5e-4.  Round 1's split-operand bf16 kernel - 2.5x the
 *                 MFMAs at the same error - is kept out of the library under tools/experiments/nonlocal_bf16.hip).
 * The default can also be set with the environment variable PFNL_CONV3X3 read by pfnl_create.
 * Kernel generations that are NOT options (round 5): the library ships one kernel per job.  What remains switchable by environment, read
 * once per process, is the assignment of the two bf16 3x3 generations to the three modes - PFNL_BF16_V3 = 0 (second generation,
 * conv_bf16_v2.hip, for all three), 1 (third generation for all three), 2 (third for conv1_i + conv10_i only); default: third for conv1_i +
 * conv10_i and the per-frame half of conv2_i, second for the shared half.  Same arithmetic in every assignment (bit-equal results): it
 * exists as the same-box A/B switch and as the reference of the schedule soak test (tools/soak_r04.py), not as a fallback for a failure.
 * Gone since round 5: PFNL_BF16_V2 (first-generation bf16 kernel for modes 0 - 2), PFNL_NL_SW (first-generation non-local kernel; batches
 * beyond 2 GB of packed operands run through the one kernel in clip chunks). */
int pfnl_set_option(pfnl_handle* h, const char* key, const char* value);
/* The current value of option `key` in the spelling pfnl_set_option takes back - whatever set it (a call, an environment variable read by
 * pfnl_create, the default).  For callers that change an option temporarily (the harness's range-flag recovery, model.py) and must put back
 * what was there, not what they remember having set.  NUL-terminated into buf[buflen]. */
int pfnl_get_option(pfnl_handle* h, const char* key, char* buf, size_t buflen);

/* ---- the hot path ------------------------------------------------------------------------ */
/* Replaces sess.run(SR_test, feed_dict={L_test: ...}) (reference model/pfnl.py:252,309) /
 * PFNL.forward (model/pfnl.py:39-80).  `in`  = [B,T,H,W,3] float32, `out` = [B,1,sH,sW,3]
 * float32.  H and W must be even (tf.space_to_depth, model/pfnl.py:57).  With device pointers
 * the call is asynchronous on `stream`; with host pointers it returns when `out` is filled. */
int pfnl_forward(pfnl_handle* h, const void* in, int in_is_device, void* out, int out_is_device,
                 int B, int H, int W, void* stream);
/* Single-clip multi-GPU sharding (the reference is single-GPU, main.py:10): the rows
 * [scale*row0, scale*(row0+nrows)) of every SR frame, written at their place in the FULL-size `out`.  `in` is the full clip
 * (device pointers, asynchronous on `stream`).  The non-local block runs this strip's queries against ALL keys of the frame
 * (they need the whole input, which every rank holds: 7 LR frames), the trunk runs on the strip plus a halo of
 * 4 + 2*num_block LR rows per side that is recomputed instead of exchanged - ranks share nothing on the data path and the
 * union of the strips equals pfnl_forward's output up to summation order (tile alignment differs). */
int pfnl_forward_strip(pfnl_handle* h, const void* in, void* out, int B, int H, int W, int row0, int nrows, void* stream);
int pfnl_workspace_bytes(pfnl_handle* h, int B, int H, int W, size_t* bytes);
/* THE LAUNCH PLAN of the progressive-fusion trunk (reference model/pfnl.py:65-71) for a [B,T,H,W,3] forward under the handle's current
 * options, as text: "<structure> launches_per_block=<n> c1x1=<launches of class conv1x1 among them> precision=<..> conv3x3=<..> conv1x1=<..>
 * c10_fused=<0|1> chain=<0|1> sf0=<0|1> strict=<0|1> tiles=<8x32-pixel tiles per per-frame launch> chains=<(clip, tile) chains>
 * whole_chains=<n> split_parts=<s> part_frames=<q> nl=<split16 | f16 | f32 | general_f32: the non-local block's kernel family> nl_pack_fused=<0|1> mfma=<16|32: the MFMA shape of the chained 3x3 launches - 16 = v_mfma_f32_16x16x32_*: the bf16 trunk's, and the fp32 chain launch of conv2_i when every CU has a chain>".  Structures: "small2" / "small3" (conv_small.hip,
 * below ~0.78 tiles per CU: 200 on a 256-CU device), "mid4" (four per-tile launches, below ~0.53 chains per CU: 136), "chain2" (conv1_i +
 * conv10_i, then the whole of conv2_i), "chain2_split" (the same with the chains of a last, partial round cut by frames: option
 * split16_splitchains) and "chain2_sf0" (the same with a split-format copy of every block's output so that the next block's
 * conv1_i takes its halo by LDS-DMA: option split16_sf0=on), "split16_3" / "split16_4", "winograd_ws3" / "winograd_ws4", "winograd_tile4",
 * "direct4"; bf16: "bf16_3", "bf16_3_split", "bf16_4", "bf16_mid4".  The ONE statement of the dispatch rule: pfnl_forward runs it, pfnl_workspace_bytes sizes
 * from it, bench.py's byte model and the tests read it here.  The structure changes the summation order, hence the last bits: the same clip
 * gives bit-different (oracle-equal) results in a batch that takes "chain2" and alone ("mid4" / "small2"). */
int pfnl_plan(pfnl_handle* h, int B, int H, int W, char* buf, size_t buflen);
int pfnl_sync(pfnl_handle* h);
/* number of synchronous forwards that were redone on the f32-MFMA kernels because the f16-pipe range flag was set */
int pfnl_range_reruns(pfnl_handle* h, long long* count);
/* The range flag of ASYNCHRONOUS (device-pointer) forwards, read and cleared WITHOUT synchronising: *flagged = 1 when a
 * device-pointer forward that has completed since the last pfnl_sync / pfnl_range_flag wrote a non-finite value.  The caller
 * orders the read behind the forwards it asks about (an event / stream synchronise of its own) - what a pipelined harness uses
 * once per batch instead of pfnl_sync (pfnl_amd/model.py, the replacement of the loop around sess.run, reference
 * model/pfnl.py:249-258).  Synchronous (host-pointer) forwards keep a flag of their own and never consume this one.  The fence
 * is armed where a kernel with a binary16 domain runs and a non-finite value cannot be the reference's own result: not under
 * strict_fp32 (f32-MFMA kernels throughout) and not with nl_type 2 (0 / 0 for a query without a positive affinity).
 * The word is taken with one atomic exchange, so a flag raised by a forward that is STILL IN FLIGHT while this is called is not lost:
 * it is either returned now or by the next call - i.e. a returned 1 may also cover forwards that have not completed yet (a caller that
 * re-runs a batch on it should treat the batches already enqueued behind it as suspect too: pfnl_amd/model.py does). */
int pfnl_range_flag(pfnl_handle* h, int* flagged);

/* Page-locked host buffers (hipHostMalloc / hipHostFree) for the tensors a caller hands to a host-pointer pfnl_forward - the call that
 * replaces sess.run(SR_test, feed_dict={L_test: ...}) (reference model/pfnl.py:252,309): the copy engine then moves them directly, with
 * no staging copy through the handle's strips.  Process-wide, not tied to a handle; pfnl_amd/engine.py keeps a small, size-capped pool of
 * them for the arrays PFNLEngine.forward returns (no torch involved: ADVICE r4). */
int pfnl_host_alloc(size_t bytes, void** out);
int pfnl_host_free(void* p);

/* ---- multi-GPU (RCCL over xGMI) ------------------------------------ */
/* Clips are independent (reference model/pfnl.py:44,55: the batch is only the leading dimension), so ranks share NOTHING on
 * the data path; these entry points carry what the reference's single-GPU session (main.py:10) has no counterpart for:
 * the weight replica, the statistics, and an optional gather.  RCCL is loaded at run time (dlopen; PFNL_RCCL_LIB overrides
 * the search) - a single-GPU caller never needs it.  One communicator per process: rank 0 calls pfnl_comm_get_unique_id,
 * moves the PFNL_COMM_ID_BYTES bytes to the other ranks through any host channel, every rank calls pfnl_comm_init_rank
 * (collective).  One process driving several devices: pfnl_comm_init_all (comms[i] on devs[i], devs NULL = 0..ndev-1). */
typedef struct pfnl_comm pfnl_comm;
#define PFNL_COMM_ID_BYTES 128
enum { PFNL_COMM_SUM = 0, PFNL_COMM_MAX = 1 };
int pfnl_comm_get_unique_id(void* id /*[PFNL_COMM_ID_BYTES]*/);
int pfnl_comm_init_rank(int nranks, int rank, const void* id, int device_id, pfnl_comm** out);
int pfnl_comm_init_all(int ndev, const int* devs, pfnl_comm** comms /*[ndev]*/);
int pfnl_comm_destroy(pfnl_comm* c);
/* (rank, nranks) as RCCL itself reports them for this communicator (ncclCommUserRank / ncclCommCount when the library exports them; they
 * must equal what the communicator was created with, else PFNL_ERR_COMM): bench.py prints them as config.comm_nranks. */
int pfnl_comm_rank(pfnl_comm* c, int* rank, int* nranks);
/* ncclBroadcast of the packed DEVICE weight blobs of `h` (what pfnl_finalize_weights built on `root`) to every rank's
 * handle of the same geometry; a non-root handle needs no pfnl_set_weight calls at all.  Replaces nothing in the reference
 * (tf.train.Saver.restore per process, model/base_model.py:231-243): one read of the checkpoint instead of one per GPU. */
int pfnl_comm_bcast_weights(pfnl_comm* c, pfnl_handle* h, int root);
/* The same replica without a communicator: the packed device blobs of `src` copied into `dst` (same geometry; same device or a
 * peer-accessible one - hipMemcpyDefault).  `dst` needs no pfnl_set_weight calls.  For several handles per process (one per
 * stream or per device) fed from one checkpoint read; also what the receive side of the broadcast is tested with on one GPU. */
int pfnl_copy_weights(pfnl_handle* dst, pfnl_handle* src);
int pfnl_comm_bcast(pfnl_comm* c, void* dev_buf, size_t bytes, int root);                 /* in place, synchronous */
/* all-reduce of n <= 64 HOST doubles in place (squared error / frame counts: PFNL_COMM_SUM; elapsed time: PFNL_COMM_MAX) */
int pfnl_comm_allreduce_f64(pfnl_comm* c, double* vals, int n, int op);
int pfnl_comm_barrier(pfnl_comm* c);
/* gather of equal per-rank shards of SR frames (device pointers): recv_dev [nranks][count_per_rank]; stream NULL = synchronous */
int pfnl_comm_allgather(pfnl_comm* c, const float* send_dev, float* recv_dev, size_t count_per_rank, void* stream);

/* ---- measurement ------------------------------------------------------------------------- */
/* Per-kernel-class HIP-event timing on the launch stream.  enable = 1: every launch of the classes
 * below is bracketed by hipEvents until disabled; enable = 2: the same for the launches outside the
 * progressive-fusion blocks and for blocks 3, 13, ... (every 4th block of models with fewer than ten; blocks are identical, so average launch durations
 * are unbiased; ~30 instead of ~95 events per forward - each event costs the stream ~2 us, and more where it breaks up back-to-back
 * launches); enable = 3: the launches of ONE progressive-fusion block (block num_block / 2) and nothing else - 3 events per forward: the
 * dominant kernel classes timed live at next to no cost to the forward that is being timed; enable = 4 ("class runs", for the
 * launch-bound small shapes where an event per launch costs a fifth of the forward): ONE event wherever the class of consecutive
 * launches changes and at the end of the forward (~8 per forward), each interval credited with the launches it holds;
 * pfnl_profile_read synchronises and returns accumulated milliseconds and the number of TIMED launches
 * since the last reset. */
enum {
    PFNL_K_NL_PACK = 0, PFNL_K_NL_ATTN = 1, PFNL_K_CONV0 = 2, PFNL_K_CONV3X3 = 3,
    PFNL_K_CONV1X1 = 4, PFNL_K_MERGE1 = 5, PFNL_K_TAIL = 6, PFNL_K_COUNT = 7
};
int pfnl_profile_enable(pfnl_handle* h, int enable);
int pfnl_profile_reset(pfnl_handle* h);
int pfnl_profile_read(pfnl_handle* h, double* ms /*[PFNL_K_COUNT]*/, int64_t* launches /*[PFNL_K_COUNT]*/);
/* Tracing: with env PFNL_ROCTX=1 every kernel class of a forward is wrapped in a roctx range ("pfnl:conv3x3", ...; roctx is
 * resolved at run time, never linked): `rocprofv3 --marker-trace --kernel-trace` then shows the launches inside named ranges. */

/* ---- debugging / per-stage parity --------------------------------------------------------- */
/* Copy an internal buffer of the last forward to host (synchronises).  Names:
 *   "nl_out"  [B,H,W,3T]   frame stack after the non-local residual (model/pfnl.py:60)
 *   "trunk"   [B,T,H,W,64] inp0 after the last PF block          (model/pfnl.py:71)
 *   "merge1"  [B,H,W,48]   after convmerge1                       (model/pfnl.py:74)  */
int pfnl_debug_tap(pfnl_handle* h, const char* name, float* host_dst, size_t count);

/* ---- single ops (the TF kernels the reference calls), device pointers, async on stream ---- */
/* tf.layers.Conv2D(k=1|3, 'same') + bias + optional leaky_relu(0.2) on NHWC float32 with the MFMA
 * implicit-GEMM kernel.  in [items*frames_per_item, H, W, 64] viewed as [items,H,W,64*fpi];
 * kernel_host = HWIO [k,k,64*fpi,cout] (host), bias_host [cout] or NULL; cout <= 64;
 * addend (device, [items/add_div,H,W,64]) or NULL is added before the activation, resid (device,
 * [items,H,W,cout]) or NULL after it; out [items,H,W,cout].  (model/pfnl.py:49-52,66-74) */
int pfnl_op_conv2d(const float* in, const float* kernel_host, const float* bias_host,
                   const float* addend, int add_div, const float* resid, float* out,
                   int items, int frames_per_item, int H, int W, int ksize, int cout, int act,
                   void* stream);
/* The whole of conv2_i (reference model/pfnl.py:51, :69-71) in one launch of the persistent Winograd kernel:
 * out[c*T+t] = resid[c*T+t] + act(conv3x3(concat([base[c], in[c*T+t]]); kernel) + bias), the base half computed
 * once per clip and tile and kept in LDS.  in/resid/out [clips*T, H, W, 64], base [clips, H, W, 64],
 * kernel_host HWIO [3,3,128,64]; out may alias resid. */
int pfnl_op_conv2_grouped(const float* in, const float* base, const float* kernel_host, const float* bias_host,
                          const float* resid, float* out, int clips, int frames_per_clip, int H, int W, int act,
                          void* stream);
/* convmerge1 (reference model/pfnl.py:52, :73-74): 3x3 convolution over the concat of frames_per_clip frames,
 * (64*fpc) -> cout <= 64, as ONE launch of the persistent Winograd kernel in its accumulating mode.
 * in [clips*fpc, H, W, 64], kernel_host HWIO [3,3,64*fpc,cout], out [clips, H, W, 64] (channels >= cout: act(0)). */
int pfnl_op_conv3x3_accum(const float* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                          int frames_per_clip, int H, int W, int cout, int act, void* stream);
/* The same sum on the f16 matrix pipe with exactly split operands: the accumulating mode of conv3x3_split16_kernel (any H, W;
 * what the forward runs for convmerge1 when conv3x3 resolves to split16; option merge1 = auto | split16 | winograd). */
int pfnl_op_conv3x3_accum_split16(const float* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                                  int frames_per_clip, int H, int W, int cout, int act, void* stream);
/* conv10_i (reference model/pfnl.py:50, :67-68): the 1x1, (frames_per_item*64) -> 64 convolution through
 * the streaming kernel that reads its A operand straight from HBM (no LDS; conv1x1.hip).
 * in [items*frames_per_item, HW, 64], kernel_host HWIO [1,1,64*fpi,64], out [items, HW, 64]. */
int pfnl_op_conv1x1_stream(const float* in, const float* kernel_host, const float* bias_host, float* out,
                           int items, int frames_per_item, int HW, int act, void* stream);
/* ... and on the f16 matrix pipe with exactly split fp32 operands (option conv1x1=split16, the default): same contract. */
int pfnl_op_conv1x1_split16(const float* in, const float* kernel_host, const float* bias_host, float* out,
                            int items, int frames_per_item, int HW, int act, void* stream);
/* bf16 trunk (option precision=bf16; BASELINE.json configs[3]): the 3x3 64->64 convolution of a progressive-fusion
 * block (reference model/pfnl.py:49,51,66,69-71) on bf16 MFMA with fp32 accumulation.  Tensors are bf16 (uint16_t
 * bit patterns) [items, H, W, 64]; kernel_host fp32 HWIO [3,3,64,64] (rounded to bf16 inside), bias fp32.
 * out = act(conv + bias + addend[item / add_div]) + resid; addend and resid both NULL or both given; out may alias resid. */
int pfnl_op_conv3x3_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, const uint16_t* addend,
                         int add_div, const uint16_t* resid, uint16_t* out, int items, int H, int W, int act, void* stream);
/* The fp32 3x3 64->64 convolution on the f16 matrix pipe with split operands (option conv3x3=split16, conv_split16.hip): fp32
 * tensors in and out, fp32 accumulation, every operand taken as f16(x) + f16((x - f16(x)) 2^11) 2^-11 (>= 22 mantissa bits per
 * product).  Same contract as pfnl_op_conv3x3_winograd (any H, W); out may alias resid. */
int pfnl_op_conv3x3_split16(const float* in, const float* kernel_host, const float* bias_host, const float* addend, int add_div,
                            const float* resid, float* out, int items, int H, int W, int act, void* stream);
/* conv1_i + conv10_i of a progressive-fusion block as ONE launch of the fp32 path (reference model/pfnl.py:66-68;
 * conv3x3_c1c10_kernel, option split16_c10): in [clips*T,H,W,64] fp32 (device) -> out1 = lrelu(conv3x3(in; k1) + b1) per frame,
 * base = lrelu(conv1x1(concat_t out1; k10) + b10) [clips,H,W,64]; k1_host HWIO [3,3,64,64], k10_host HWIO [1,1,64T,64].  The kernel
 * writes both results in the split format; the hook returns them as fp32 (hi + lo' 2^-11). */
int pfnl_op_conv1_conv10_split16(const float* in, const float* k1_host, const float* b1_host, const float* k10_host,
                                 const float* b10_host, float* out1, float* base, int clips, int frames_per_clip, int H, int W,
                                 void* stream);
/* ... with its input taken in the split format by LDS-DMA (conv3x3_c1c10_kernel<true>: what blocks 1 .. nb-1 of the default "chain2_sf0"
 * plan run, reference model/pfnl.py:66-68).  The hook converts `in` with the producer's own split first; results are BIT-IDENTICAL to
 * pfnl_op_conv1_conv10_split16 (same binary16 operand pairs, same order). */
int pfnl_op_conv1_conv10_split16_sf0(const float* in, const float* k1_host, const float* b1_host, const float* k10_host,
                                     const float* b10_host, float* out1, float* base, int clips, int frames_per_clip, int H, int W,
                                     void* stream);
/* The whole of conv2_i in one launch (pfnl_op_conv3x3_split16_sf which = 2; reference model/pfnl.py:69-71) that ALSO writes its result in
 * the split format (conv3x3_sf_chain_kernel<true>): out [items,H,W,64] fp32 as before; out_sf [items,H,W,128] binary16 bit patterns
 * (device) = per pixel [channel half][hi 32 | lo' 32], hi = f16(out), lo' = f16((out - hi) 2^11). */
int pfnl_op_conv2_chain_sf0(const float* in, const float* kernel_host, const float* bias_host, const float* base, int add_div, const float* resid,
                            float* out, uint16_t* out_sf, int items, int H, int W, int act, void* stream);
/* The split-format variants of the split-f16 kernels (pfnl_amd/csrc/conv_split16.h "SF": an activation tensor that only feeds MFMA
 * operands - conv1_i's and conv10_i's outputs, model/pfnl.py:66-68 - is kept as (hi, lo') binary16 pairs, built once by its
 * producer).  fp32 at the hook's interface: conversions bracket the kernel under test.
 * which = 0: the 3x3 kernel of conv2_i (input SF by LDS-DMA, epilogue from registers; plain, or fused with addend + resid);
 * which = 1: the 3x3 kernel of conv1_i writing SF;
 * which = 2: the WHOLE of conv2_i in one launch (conv3x3_sf_chain_kernel): kernel_host = HWIO [3,3,128,64], `addend` = base
 *   [items/add_div][H][W][64] fp32, out = lrelu(conv(concat([base, in])) + bias) + resid. */
int pfnl_op_conv3x3_split16_sf(int which, const float* in, const float* kernel_host, const float* bias_host, const float* addend,
                               int add_div, const float* resid, float* out, int items, int H, int W, int act, void* stream);
int pfnl_op_conv1x1_split16_sf(const float* in, const float* kernel_host, const float* bias_host, float* out, int items,
                               int frames_per_item, int HW, int act, int in_sf, int out_sf, void* stream);
/* The small-shape trunk kernel (pfnl_amd/csrc/conv_small.hip; BASELINE.json configs[0] / configs[4]): conv1_i, conv10_i, the whole of
 * conv2_i (3x3 over concat([base, f]), model/pfnl.py:69-71) and convmerge1 as one template.  Source s of output item i is
 * s < nA ? a[i / a_div] : b[i * b_mul + (s - nA)] ([.][H][W][64] fp32 each); kernel HWIO [ks, ks, 64 nsrc, cout <= 64];
 * out [items][H][W][64] (channels >= cout are written as act(0)); resid may alias out. */
int pfnl_op_conv_small(const float* a, const float* b, int nA, int a_div, int b_mul, int nsrc, const float* kernel_host,
                       const float* bias_host, const float* resid, float* out, int items, int H, int W, int ks, int cout, int act,
                       void* stream);
/* One progressive-fusion block (reference model/pfnl.py:66-71) on the small-shape kernels as the forward launches it (option
 * small_c10 = on: two launches): inp1 = lrelu(conv3x3(x; k1) + b1) [clips*T, H, W, 64] - the same launch writes each frame's
 * 64 x 64 slice of conv10_i as a partial sum - and out = x + lrelu(conv3x3(concat([base, inp1_t]); k2) + b2), base = lrelu(sum of
 * the T partials + b10) formed in the second launch's prologue.  k1 HWIO [3,3,64,64], k10 [1,1,64T,64], k2 [3,3,128,64]. */
int pfnl_op_conv_small_pf_block(const float* x, const float* k1_host, const float* b1_host, const float* k10_host, const float* b10_host,
                                const float* k2_host, const float* b2_host, float* inp1, float* out, int clips, int T, int H, int W,
                                void* stream);
/* conv1_i and conv10_i of a progressive-fusion block (reference model/pfnl.py:66-68) in ONE launch of the bf16 3x3 kernel:
 * out1 = lrelu(conv3x3(in) + b1) [clips*fpc, H, W, 64], base = lrelu(conv1x1(concat_t out1_t) + b10) [clips, H, W, 64];
 * the 1x1 contraction reads every finished tile from LDS.  fpc in {3,5,7}. */
int pfnl_op_conv1_conv10_bf16(const uint16_t* in, const float* k1_host, const float* b1_host, const float* k10_host,
                              const float* b10_host, uint16_t* out1, uint16_t* base, int clips, int frames_per_clip, int H, int W,
                              void* stream);
/* convmerge1 (reference model/pfnl.py:52, :73-74) from the bf16 trunk: 3x3 over the concat of frames_per_clip frames,
 * (64*fpc) -> cout <= 64, as ONE launch of the bf16 3x3 kernel in its accumulating mode (the weight pack in LDS is replaced
 * between the frames of a chain).  in bf16 [clips*fpc, H, W, 64]; kernel_host fp32 HWIO [3,3,64*fpc,cout]; out fp32
 * [clips, H, W, 64] (channels >= cout: act(0)). */
int pfnl_op_conv3x3_accum_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                               int frames_per_clip, int H, int W, int cout, int act, void* stream);
/* ... and conv10_i (reference model/pfnl.py:50, :67-68): in [items*fpi, HW, 64] bf16, kernel_host fp32 HWIO
 * [1,1,64*fpi,64], out [items, HW, 64] bf16; fpi in {3,5,7}. */
int pfnl_op_conv1x1_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, uint16_t* out, int items,
                         int frames_per_item, int HW, int act, void* stream);
/* The same 3x3 64->64 convolution (frames_per_item = 1, cout = 64) through the fused Winograd
 * F(2x2,3x3) kernel; H and W must be even. */
int pfnl_op_conv3x3_winograd(const float* in, const float* kernel_host, const float* bias_host,
                             const float* addend, int add_div, const float* resid, float* out,
                             int items, int H, int W, int act, void* stream);
/* Same contract, the persistent wave-specialised kernel (conv_wino_ws.hip). */
int pfnl_op_conv3x3_winograd_ws(const float* in, const float* kernel_host, const float* bias_host,
                                const float* addend, int add_div, const float* resid, float* out, int items, int H,
                                int W, int act, void* stream);
/* utils.NonLocalBlock(nltype=1) + the residual of model/pfnl.py:55-60:
 * x [B,T,H,W,3] -> out [B,H,W,3T] = stack(x) + depth_to_space(NL(space_to_depth(stack(x)))). */
int pfnl_op_nonlocal(const float* x, const float* wg_host, const float* bg_host,
                     const float* ww_host, const float* bw_host, float* out,
                     int B, int T, int H, int W, void* stream);
/* The same block (fp32 in and out) on the f16 matrix pipe with exactly split operands (option nonlocal=split16, nonlocal_f16.hip):
 * Q, K, V and the probabilities are taken as f16(x) + f16(x - f16(x)), scaled by powers of two into binary16's normal range. */
int pfnl_op_nonlocal_split16(const float* x, const float* wg_host, const float* bg_host,
                     const float* ww_host, const float* bw_host, float* out,
                     int B, int T, int H, int W, void* stream);
/* The same kernel on the hi parts only - 16-bit (binary16) operands, fp32 accumulation and softmax state: the non-local block of
 * precision=bf16 (24 instead of 72 MFMAs per 64 keys). */
int pfnl_op_nonlocal_f16(const float* x, const float* wg_host, const float* bg_host,
                     const float* ww_host, const float* bw_host, float* out,
                     int B, int T, int H, int W, void* stream);
/* The embedded-Gaussian form of the block (reference utils.py:18-71 with nltype 0): same contract as pfnl_op_nonlocal plus
 * theta / phi 1x1 projections wt, wp [C,C] (row = input channel), bt, bp [C]. */
int pfnl_op_nonlocal_embedded(const float* x, const float* wg_host, const float* bg_host, const float* ww_host,
                              const float* bw_host, const float* wt_host, const float* bt_host, const float* wp_host,
                              const float* bp_host, float* out, int B, int T, int H, int W, void* stream);
/* utils.NonLocalBlock(input_x, out_channels, sub_sample, nltype) in its general form (reference utils.py:18-71) with the same
 * stack / space_to_depth / residual wrapper as pfnl_op_nonlocal: nltype 0 embedded Gaussian (theta, phi projections, exp / rowsum),
 * 1 Gaussian (theta = phi = x; wt .. bp may be NULL), 2 dot product (relu / rowsum, utils.py:59-62 - a query whose affinities are
 * all <= 0 divides by zero there and yields NaN here as well); sub_sample > 1: g and phi average-pooled (pool = stride = sub_sample,
 * 'valid') on the space_to_depth grid (utils.py:27-28,35-36).  nltype 3 ('concat') builds no graph in the reference and is rejected. */
int pfnl_op_nonlocal_block(const float* x, const float* wg_host, const float* bg_host, const float* ww_host,
                           const float* bw_host, const float* wt_host, const float* bt_host, const float* wp_host,
                           const float* bp_host, int nltype, int sub_sample, float* out, int B, int T, int H, int W, void* stream);
/* conv0 (reference model/pfnl.py:48,61-62): lrelu(conv5x5 'same' 3 -> 64 + b) of every frame.
 * x [B,T,H,W,3] (device), kernel_host HWIO [5,5,3,64], bias_host [64] or NULL, out [B*T,H,W,64] (device). */
int pfnl_op_conv0(const float* x, const float* kernel_host, const float* bias_host, float* out, int B, int T, int H, int W,
                  void* stream);
/* The tail (reference model/pfnl.py:53,63,76-80): depth_to_space(2) -> convmerge2 3x3 (no activation) -> depth_to_space(2)
 * (scale 4 only) -> + ResizeBicubic(x[:, T/2]) -> [B,1,sH,sW,3].  merge [B,H,W,48] (device, the output of convmerge1),
 * x [B,T,H,W,3] (device), kernel_host HWIO [3,3,12,12] (scale 4) or [3,3,12,3] (scale 2), bias_host or NULL. */
int pfnl_op_tail(const float* merge, const float* x, const float* kernel_host, const float* bias_host, float* out, int B,
                 int T, int H, int W, int scale, void* stream);
/* Harness helpers (reference model/pfnl.py:238-242, 254-257), device pointers, async on stream:
 * windows: frames [F,H,W,3] -> win [count,T,H,W,3], window w slot t = frame clip(first + w + t - T/2, 0, F-1);
 * quantise: uint8(np.round(np.clip(sr * 255, 0, 255))) (round half to even), n % 4 == 0. */
int pfnl_op_gather_windows(const float* frames, float* win, int F, int first, int count, int T, int H, int W, void* stream);
int pfnl_op_quantise_u8(const float* sr, uint8_t* out, size_t n, void* stream);
/* tf.image.resize_images(method=2) of TF1.12 (model/pfnl.py:63): x [B,H,W,3] -> [B,sH,sW,3]. */
int pfnl_op_bicubic(const float* x, float* out, int B, int H, int W, int scale, void* stream);
/* The step before the path in test_video_truth / eval (reference utils.py:95-105,169-192:
 * DownSample_4D with BLUR): reflect-pad 6, 13x13 Gaussian sigma 1.6, stride `scale`, VALID.
 * hr [F,H,W,3] -> lr [F,ceil(H/scale),ceil(W/scale),3], device pointers, async on stream. */
int pfnl_op_blur_decimate(const float* hr, float* lr, int F, int H, int W, int scale, void* stream);
/* MFMA operand/accumulator layout self-test (asymmetric operands); returns 0 if the f32
 * 32x32x2 fragment maps this library assumes hold on the device. */
int pfnl_selftest_mfma(int device_id);

#ifdef __cplusplus
}
#endif
#endif /* PFNL_HIP_H */
