/* libfrost_hip.so -- C ABI of the MI355X-native FrostNet QAT hot path.
 *
 * The reference (clovaai/frostnet) has NO FFI: its hot path is torch eager-mode QAT modules composed in
 * Python (frostnet.py:14-145, optimizer.py:121-667, Classification/train.py:166-173).  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md): every entry replaces one stage of the
 * torch module graph the reference executes, cited per function as `replaces:`.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.  All pointers are DEVICE pointers owned by
 *    the caller.  Every call is asynchronous on `stream` (a hipStream_t passed as void*) and never syncs.
 *  - return 0 on success, >0 = argument error, <0 = -(hipError_t).  Nothing aborts or throws.
 *  - Activations: NHWC, one signed byte per element holding (q - 128), q = the reference's quint8 index
 *    ("offset-binary index").  Channel counts are multiples of 4 (FrostNet: multiples of 8).
 *    Activation buffers must be allocated with >= 64 bytes of slack after the last element.
 *  - qrecord: 12 floats per fake-quantize site (FROST_Q_*): observer min/max, scale, zero_point(int bits),
 *    fake-quantised tensor min/max, 1/scale, observer_enabled / fake_quant_enabled.  torch's module buffers are views
 *    into these records.
 *  - gradients of activations: bf16, NHWC.  Parameter gradients: fp32.
 *  - alignment: tensors, weight packs, weight sums and coefficient rows are 16-byte aligned (vector loads and LDS DMA; the pointwise entries check
 *    the three tables and return an argument error otherwise).
 */
#ifndef FROST_HIP_H
#define FROST_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a caller built against an older header MUST NOT call a newer library, and vice versa: check frost_abi_version() at load time)
 *   2 -> 3 (round 3/4): every last-workgroup-done ticket buffer grew from 1 to FROST_TICKET_WORDS (40) zeroed uint32 words -- FrostFinDesc.counter, and
 *          the state of frost_add_minmax_observe / frost_pw_ew_emit_add is frost_add_state_floats() floats: {2 unused, ticket[FROST_TICKET_WORDS], per-workgroup
 *          range slots} (was {lo, hi, ticket}); FrostFinDesc gained
 *          cat_qrec_b / cat_qrec_y.  A 1-word ticket under the new kernels is an out-of-bounds device atomic: frost_ticket_words() and
 *          frost_fin_desc_bytes() let a binding verify both sizes before its first launch (frostnet_amd/_lib.py does).
 *   3 -> 4 (round 5): frost_g32_reduce / frost_g32_wgrad take a `scratch` pointer (>= frost_g32_scratch_bytes() bytes; NULL = the plain kernels) before `stream`.
 *   4 -> 5 (round 6): a layer's statistics scratch is FROST_STATS_TABLES replicated tables (FROST_STATS_BYTES_PER_CH 24 -> 96): a buffer sized for one table is
 *          overrun by the statistics kernels; a coefficient table is FROST_COEF_ROWS_ALLOC (14) rows: three more copies of the S1 / S2 rows behind the 8 named ones.
 *          A raw weight-gradient buffer (`dwq`) of a layer with at most FROST_DWQ_SPREAD_MAX (32768) weights is FROST_DWQ_NC (4) copies at a stride of round_up(weights, 64)
 *          floats, all zeroed by the caller: the fused pointwise / depthwise backward kernels spread their flush atomics over the copies, frost_weight_grad_finalize[_table]
 *          and frost_stem_wgrad_remap add them up; every other producer writes copy 0.  Larger layers keep one copy.
 *          New entries (additive): frost_step_prologue, frost_block_dw_bwd_c1 / _c1_ok, frost_hswish_converted.  The `relu` argument of the frost_float_* and
 *          frost_infer_pw / _dw / _stem entries is an activation code: 0 none, 1 ReLU, 2 hard-swish (0 / 1 mean what they meant). */
#ifndef FROST_DWQ_NC          /* (a -D override is a dev A/B build: the binding must be told the same value, FROST_DWQ_NC / FROST_COEF_ROWS_ALLOC / FROST_STATS_TABLES in the environment) */
#define FROST_DWQ_NC 4
#endif
#define FROST_DWQ_SPREAD_MAX 32768
#define FROST_ABI_VERSION 5

/* qrecord field indices (floats) */
#define FROST_Q_MIN 0
#define FROST_Q_MAX 1
#define FROST_Q_SCALE 2
#define FROST_Q_ZP 3      /* int32 bits */
#define FROST_Q_FQMIN 4
#define FROST_Q_FQMAX 5
#define FROST_Q_INV 6
#define FROST_Q_QMAX 7    /* float: upper index of an ACTIVATION site, 255 (qnnpack) or 127 (fbgemm qconfig: reduce_range); 0 = 255 */
#define FROST_Q_OBS_EN 8   /* 8 bytes (slots 8-9): FakeQuantize.observer_enabled, aliased by the torch buffer (uint8[1] or int64[1]);
                              kernels test the low 32-bit word != 0, so torch.quantization.disable_observer on ANY sub-module is
                              honoured per site without a host round trip (torch/ao/quantization/fake_quantize.py:170-184) */
#define FROST_Q_FQ_EN 10   /* 8 bytes (slots 10-11): FakeQuantize.fake_quant_enabled (must stay 1: checked on the host side) */
#define FROST_Q_STRIDE 12

/* per-conv-layer coefficient rows (floats, each row `cpad` long; cpad = round_up(cout,16)) */
#define FROST_COEF_A 0      /* y = fma(A, acc, B)                                   */
#define FROST_COEF_B 1
#define FROST_COEF_M 2      /* xhat = (acc - M) * R                                 */
#define FROST_COEF_R 3
#define FROST_COEF_K1 4     /* dc = K1 * (gy - S1/n - xhat*S2/n)                    */
#define FROST_COEF_S1 5     /* sum gy      (written by the backward reduce pass)    */
#define FROST_COEF_S2 6     /* sum gy*xhat                                          */
#define FROST_COEF_VFRAC 7  /* v/(v+eps): d(gamma) = S2*VFRAC + fold term           */
#define FROST_COEF_ROWS 8
/* rows a caller ALLOCATES per coefficient table (ABI 5): the 8 rows above + 3 more copies of the S1 / S2 rows (rows 8 + 2 (k - 1) + {0, 1}, k = 1 .. 3).  The backward
 * reduce passes spread their float atomics over the four copies by workgroup index; every reader (dc passes, parameter-gradient finalize) adds the copies up; the forward
 * finalize zeroes all of them.  Rows 5 / 6 alone are the totals only where a kernel writes totals (the fp32-gradient mode). */
#ifndef FROST_COEF_ROWS_ALLOC
#define FROST_COEF_ROWS_ALLOC 14
#endif

/* stats scratch per conv layer: FROST_STATS_TABLES identical tables of 24 bytes per (padded) channel -- int64 sum, uint64 sumsq, int32 min, int32 max,
 * laid out SoA: [cpad] i64 | [cpad] u64 | [cpad] i32 | [cpad] i32 -- one after the other.  The statistics kernels spread their flush atomics over the tables by
 * workgroup index, the finalize adds them up (exact integers: the result does not depend on the spread); frost_stats_init_table / frost_step_prologue reset all of
 * them.  A caller sizes the buffer as cpad * FROST_STATS_BYTES_PER_CH (ABI 5: was ONE table, 24 bytes per channel). */
#ifndef FROST_STATS_TABLES
#define FROST_STATS_TABLES 4
#endif
#define FROST_STATS_BYTES_PER_CH (24 * FROST_STATS_TABLES)

int frost_abi_version(void);
int frost_ticket_words(void);      /* == FROST_TICKET_WORDS of the library that was loaded */
int frost_fin_desc_bytes(void);    /* == sizeof(FrostFinDesc) of the library that was loaded */
const char* frost_last_error(void);

/* ---- element-wise / observer -------------------------------------------------------------------------- */
/* replaces: MovingAverageMinMaxObserver.forward's torch.aminmax (torch/ao/quantization/observer.py:668-683)
 * out2[0]=min, out2[1]=max (must be pre-set to +inf/-inf by frost_fill_minmax) */
int frost_minmax_f32(const float* x, int64_t n, float* out2, void* stream);
int frost_fill_minmax(float* out2, int count, void* stream);
/* min/max of the logical (N,C,H,W) fp32 input with arbitrary element strides (the QuantStub observer) */
int frost_minmax_input(const float* x, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                       float* out2, void* stream);
/* replaces: observer EMA + _calculate_qparams (observer.py:349-427) for ONE site.
 * cur2 = {min,max} of the current tensor (raw floats).  symmetric: 0 = quint8 affine 0..255, 1 = qint8 -128..127.
 * rule127: 0 = '/127.5' (qconfig version 0), 1 = max(-min/128,max/127) (version 1). fqmin/max are derived from cur2. */
int frost_observer_update(float* qrec, const float* cur2, int symmetric, int rule127, int observe, void* stream);
/* replaces: QuantStub's FakeQuantize on the input image (frostnet.py:319-320).  x: fp32 with arbitrary
 * strides (elements) for logical (N,C,H,W); out: NHWC int8 with C padded to cpad (pad lanes = zero-point). */
int frost_quantize_input(const float* x, int n, int c, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         const float* qrec, int8_t* out, int cpad, void* stream);
/* generic fake-quant fwd / STE bwd on fp32 (torch.fake_quantize_per_tensor_affine, fake_quantize.py:242-259) */
int frost_fake_quant_f32(const float* x, int64_t n, const float* qrec, int qmin, int qmax, float* y, uint8_t* mask,
                         void* stream);
int frost_fake_quant_bwd_f32(const float* dy, const uint8_t* mask, int64_t n, float* dx, void* stream);
/* dequantise an activation (offset-binary int8 NHWC) to fp32 (for taps / tests) */
int frost_dequant_act(const int8_t* q, int64_t n, const float* qrec, float* y, void* stream);

/* ---- weights: BN-fold + weight fake-quant + MFMA packing --------------------------------------------- */
/* replaces: ConvBn2d._forward_approximate lines 135-143 (scale_factor, weight_fake_quant) for a LIST of layers
 * in three launches.  Descriptor table lives in device memory (see FrostWDesc). */
typedef struct FrostWDesc {
  const float* w;        /* OIHW fp32 [cout][cin_g][k][k]                                    */
  const float* gamma;    /* bn.weight  (NULL for the classifier: no fold)                   */
  const float* rvar;     /* bn.running_var (value BEFORE this step's update)               */
  float* qrec;           /* weight fake-quant qrecord                                       */
  int8_t* wq_pack;       /* packed int8 weights, layout by `kind`                           */
  int32_t* wsum;         /* [cpad] sum_k wq                                                 */
  float* minmax2;        /* scratch {min,max}                                               */
  uint16_t* wt_pack;     /* bf16 transposed pack for dgrad (pw only, may be NULL): wq * (wscale[co] / qrec.scale)   */
  float* wscale;         /* [cpad] weight scale PER OUTPUT CHANNEL, always filled: per-tensor mode replicates qrec.scale  */
  float* wmin; float* wmax;   /* [cout] per-channel observer state (MovingAveragePerChannelMinMaxObserver); NULL per-tensor */
  int32_t cout, cin_g, kk, kind;   /* kind: 0 pointwise, 1 depthwise, 2 stem(3x3 dense cin<=4), 3 classifier */
  int32_t cpad, kpad;
  int32_t reserved0;   /* fold mode: 0 = gamma / sqrt(var+eps) (QAT forward), 1 = gamma * rsqrt(var+eps) (convert: fuse_conv_bn_weights) */
  int32_t reserved1;   /* 1 = per-channel symmetric weight quantisation (the reference's 'fbgemm' qconfig of its latency_check scripts):
                          scale per output channel, qrec.scale = the largest of them (the scalar the data-gradient kernels apply)  */
} FrostWDesc;
int frost_weight_prep(const FrostWDesc* descs, int nlayers, int max_elems, int rule127, int observe, void* stream);
/* The whole per-step prologue -- frost_save_sigma + frost_weight_prep + frost_stats_init_table, same results bit for bit -- in THREE launches over a flat workgroup
 * map the caller builds once: wgmap = nwg x {layer index, slot, slots of that layer, 0} (int32 x 4), every layer with >= 1 slot (round 6: the seven launches of
 * mostly idle workgroups cost 0.28 ms at the head of every captured step). */
int frost_step_prologue(const FrostWDesc* descs, int nlayers, const int32_t* wgmap, int nwg, float* const* sigma_outs, void* stats,
                        const int32_t* cpads, const int64_t* offs, int rule127, int observe, void* stream);
/* Export of the converted model (torch.quantization.convert + state_dict(), Classification/evaluate.py:130-143): the int8 weight values of layer
 * `layer` of the table as frost_weight_prep quantised them, in the module's own [cout][cin/g][kh][kw] order (nelem = cout * cin_g * kk bytes). */
int frost_export_wq(const FrostWDesc* descs, int layer, int64_t nelem, int8_t* out, void* stream);

/* ---- conv stats / finalize / emit ---------------------------------------------------------------------- */
/* reset the per-layer integer stats scratch (sum=0,sumsq=0,min=INT_MAX,max=INT_MIN); cpads/offs are device arrays */
int frost_stats_init_table(void* stats, const int32_t* cpads, const int64_t* offs, int nlayers, void* stream);
/* replaces: F.conv2d of a 1x1 conv on fake-quantised operands (conv_fused.py:152) -- int8 MFMA.
 * mode 0: accumulate per-channel stats only.  mode 1: emit y=fq(relu?(A*acc+B)) as offset-binary int8.
 * mode 2 / 3: emit in converted-inference form (QNNPACK integer bias / FBGEMM float bias + per-channel multiplier), see frost_conv_finalize_converted[_fb]. */
int frost_pw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix,
                      int cin, int cout, int mode, void* stats, const float* coef, const float* qrec_y, int relu,
                      int8_t* y, void* stream);
/* replaces: depthwise k x k conv (groups=C) on fake-quantised operands.  x NHWC [n,h,w,c]. */
int frost_dw_conv_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h,
                      int w, int c, int k, int stride, int mode, void* stats, const float* coef, const float* qrec_y,
                      int relu, int8_t* y, void* stream);
/* replaces: stem 3x3 s2 conv 3->cout (frostnet.py:277), as im2col + pointwise: 3x3/s2/p1 patches of the 4 B/pixel image -> [npix_out][40] bytes (k = tap*4 + c); the stem
 * then runs on frost_pw_conv_fwd / _bwd / frost_pw_wgrad with cin = 40; frost_stem_wgrad_remap maps dWq back to OIHW. */
int frost_stem_im2col(const int8_t* x, const float* qrec_x, int n, int h, int w, int8_t* out, void* stream);
int frost_stem_wgrad_remap(const float* dwq_col, int cout, int cin_g, float* dwq, void* stream);
/* replaces: conv/scale_factor, BatchNorm2d (train: batch stats + running-stat update; eval: running stats),
 * the activation observer + qparams (conv_fused.py:153-157, fake_quantize.py:229-240).  One launch per layer. */
int frost_conv_finalize(const void* stats, int64_t count, int cout, const float* qrec_x, const float* qrec_w,
                        const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt, int training,
                        int relu, int observe, float* coef, float* qrec_y, const float* wscale, void* stream);
/* (wscale: [cpad] per-output-channel weight scales, FrostWDesc.wscale; NULL = the scalar of qrec_w) */

/* ---- cat / add ------------------------------------------------------------------------------------------ */
/* replaces: FloatFunctional.cat + its FakeQuantize (frostnet.py:129) */
int frost_cat_observe(const float* qrec_a, const float* qrec_b, float* qrec_y, int observe, void* stream);
int frost_cat_requant(const int8_t* a, const float* qrec_a, int ca, const int8_t* b, const float* qrec_b, int cb,
                      int64_t npix, const float* qrec_y, int8_t* y, void* stream);
/* replaces: FloatFunctional.add + its FakeQuantize (frostnet.py:142): pass 0 = min/max of a+b, pass 1 = emit */
int frost_add_minmax(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                     float* minmax2, void* stream);
/* the same range pass with the MovingAverageMinMax update of the sum's FakeQuantize folded into its last workgroup (replaces frost_fill_minmax +
 * frost_add_minmax + frost_observer_update: one launch instead of three).  state3 = {lo, hi, arrival ticket[FROST_TICKET_WORDS]}: (+inf, -inf, 0...) on entry and again on exit. */
int frost_add_state_floats(void);      /* floats of the `state` buffer of frost_add_minmax_observe / frost_pw_ew_emit_add: {2 unused, ticket[FROST_TICKET_WORDS], 2 * 1024 per-workgroup range slots}, zero-filled once */
int frost_add_minmax_observe(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                             float* state3, float* qrec_y, int observe, void* stream);
int frost_add_requant(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n,
                      const float* qrec_y, int8_t* y, void* stream);

/* ---- head: avg-pool + dropout + classifier (frostnet.py:295-299) -------------------------------------- */
int frost_avgpool(const int8_t* x, const float* qrec_x, int n, int hw, int c, const float* drop_mask, float* y,
                  void* stream);
/* y[n][nclass] = x[n][cin] . wq^T * s_w + bias  (fp32, weights int8 fake-quantised) */
int frost_classifier_fwd(const float* x, const int8_t* wq, const float* qrec_w, const float* bias, int n, int cin,
                         int nclass, float* y, const float* wscale, void* stream);

/* ---- backward ------------------------------------------------------------------------------------------- */
/* pass 0: S1=sum gy, S2=sum gy*xhat (into coef rows S1,S2); pass 1: write dc (bf16 [npix][cout]) and
 * dx (bf16 [npix][cin], beta=accumulate flag) */
int frost_pw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                      const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout, int pass,
                      float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, uint16_t* dx,
                      int accumulate, void* stream);
/* dWq[cout][cin] += sum_p dc[p][cout] * (x[p][cin]-zp) * s_x  (fp32 accumulate into dwq, must be zeroed) */
int frost_pw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int64_t npix, int cin, int cout,
                   float* dwq, void* stream);

/* Fused pointwise backward for layers with Cout*Cin <= ~19 k (all 112^2 / 56^2 / 28^2 layers of FrostNet): the dc pass, the data gradient and
 * the weight gradient of frost_pw_conv_bwd(pass 1, 2) + frost_pw_wgrad in ONE kernel -- the dc tile stays in LDS (the 2-byte dc tensor is
 * neither written nor re-read: -6 B per output element of HBM traffic) and the weight gradient re-uses the staged x tile.
 * replaces: the same autograd stages as frost_pw_conv_bwd / frost_pw_wgrad (conv_fused.py:130-157 backward through BN-train + ReLU + FQ masks).
 * frost_pw_bwd_fused_ok() returns 1 when the shape is supported (npix a multiple of 128, LDS budget), else 0 (not an error code).
 * dx may be NULL (no data gradient wanted, e.g. the stem); dwq (fp32 [cout][cin]) is accumulated with atomics and must be pre-zeroed. */
int frost_pw_bwd_fused_ok(int64_t npix, int cin, int cout);
/* data gradient of a pointwise layer with long dc rows (Cout > 128) as a stand-alone bf16 GEMM: dx[p][ci] (+)= s_w * sum_co dc[p][co] * wq[co][ci]
 * (replaces: the grad_input of F.conv2d inside ConvBn2d / ConvBnReLU2d, frostnet.py:124-145 through torch.ao's _forward_approximate; same result as
 * pass 2 of frost_pw_conv_bwd).  wt_pack: the transposed bf16 pack of frost_weight_prep.  frost_pw_dgrad_wide_ok() = 1 if the shape qualifies. */
/* backward of a wide-K pointwise layer (Cin > 256) with ONE recomputation instead of two: frost_pw_conv_int stores the integer conv output
 * c[p][co] = sum_k xq*wq - (zp_x - 128)*wsum[co] ([npix][cout] int32), frost_pw_ew runs the reduce pass (mode 0: S1 / S2 into the coefficient rows) and
 * the dc pass (mode 1) element-wise over it -- the results of passes 0 and 1 of frost_pw_conv_bwd (replaces: autograd's BatchNorm + fake-quantise
 * backward of ConvBn(ReLU)2d, frostnet.py:124-145 through torch.ao's _forward_approximate). */
int frost_pw_conv_int(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                      int32_t* conv_out, void* stream);
int frost_pw_ew(const int32_t* conv_out, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, int mode, const uint16_t* gout,
                void* out, void* stream);          /* out: mode 1 dc (bf16), mode 2 y (int8, q - 128) */
int frost_pw_dgrad_wide_ok(int64_t npix, int cin, int cout);
int frost_pw_dgrad_wide(const uint16_t* dc, const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout,
                        uint16_t* dx, int accumulate, void* stream);
int frost_pw_conv_bwd_fused(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                            const uint16_t* wt_pack, const float* qrec_w, int64_t npix, int cin, int cout, float* coef,
                            const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc_scratch, uint16_t* dx,
                            int accumulate, float* dwq, void* stream);
int frost_dw_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                      const float* qrec_w, int n, int h, int w, int c, int k, int stride, int pass, float* coef,
                      const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream);
/* pass 1 of frost_dw_conv_bwd with the depthwise weight gradient (frost_dw_wgrad) accumulated in the same sweep */
int frost_dw_conv_bwd_dc_wgrad(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum,
                               const float* qrec_w, int n, int h, int w, int c, int k, int stride, float* coef,
                               const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, float* dwq, void* stream);
int frost_dw_dgrad(const uint16_t* dc, const int8_t* wq_pack, const float* qrec_w, int n, int h, int w, int c, int k,
                   int stride, uint16_t* dx, int accumulate, const float* wscale, void* stream);
int frost_dw_wgrad(const uint16_t* dc, const int8_t* x, const float* qrec_x, int n, int h, int w, int c, int k,
                   int stride, float* dwq, void* stream);
/* dc pass + weight gradient + data gradient of a depthwise k = 3, stride-1 layer in ONE sweep (csrc/frost_dwb.hip): replaces frost_dw_conv_bwd_dc_wgrad +
 * frost_dw_dgrad -- the backward of nniqat.ConvBnReLU2d with groups = channels (/root/reference/frostnet.py:96-101, conv2) after its reduce pass filled the S1 / S2
 * coefficient rows.  dc never reaches HBM (5 instead of 9 bytes per element); dx is overwritten (no accumulate form); dwq receives the raw weight-gradient sums
 * [c][k*k] like frost_dw_wgrad (atomics: zeroed by the caller).  Same expressions as the separate kernels; with FROST_SR=0 (round-to-nearest dc) dx is bit-identical
 * to theirs.  wscale: per-channel weight scales or NULL (qrec_w's scalar). */
int frost_dw_bwd_fused_ok(int h, int w, int c, int k, int stride);
int frost_dw_bwd_fused(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                       int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                       uint16_t* dx, float* dwq, void* stream);
/* the same sweep carrying the REDUCE PASS of the pointwise layer that produced x (conv1 of the bottleneck, frostnet.py:88-95: replaces frost_pw_conv_bwd pass 0 on
 * (c1_x -> x)): S1 = sum g, S2 = sum g * xhat of that layer accumulate into c1_coef (its coefficient rows; S1 / S2 zero on entry as for the separate pass) with g = the dx
 * value before its bf16 rounding, inside conv1's STE window (k_pw's reduce-pass expressions on conv1's integer accumulator, recomputed from c1_x and c1_wq_pack).
 * c1_x: conv1's input [n][h][w][c1_cin] (c1_cin = 16 or 24), c1_qrec_x its record; conv1's output record is qrec_x.  Stride-2 layers only (_c1_ok). */
int frost_dw_bwd_fused_c1_ok(int h, int w, int c, int k, int stride, int c1_cin);
int frost_dw_bwd_fused_c1(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                          int n, int h, int w, int c, int k, int stride, const float* coef, const float* qrec_y, int relu, const uint16_t* gout,
                          uint16_t* dx, float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef,
                          int c1_cin, int c1_relu, void* stream);
/* fold-path: dW = dWq*mask*sf ; dgamma = S2*vfrac + sum(dWq*mask*W)/sigma_r ; dbeta = S1  (SURVEY H-5) */
/* sigma_r[c] = sqrt(running_var+eps) as used by THIS step's forward (frost_save_sigma runs before the update) */
int frost_weight_grad_finalize(const float* dwq, const float* w, const float* gamma, const float* sigma_r,
                               const float* qrec_w, const float* coef, int cout, int cin_g, int kk, int cpad,
                               float* dw, float* dgamma, float* dbeta, int accumulate, const float* wscale, void* stream);
/* the same for a table of layers in one launch (device-resident descriptors; single-GPU path, where nothing waits on
 * per-layer gradients; the data-parallel path keeps the per-layer call so the bucketed all-reduce can start early) */
typedef struct FrostGDesc {
  const float* dwq; const float* w; const float* gamma; const float* sigma_r; const float* qw; const float* coef;
  float* dw; float* dgamma; float* dbeta;
  int32_t cout, per, cpad, reserved;
  const float* wscale;   /* [cpad] per-output-channel weight scale (NULL = qw's scalar) */
} FrostGDesc;
int frost_weight_grad_finalize_table(const FrostGDesc* descs, int nlayers, void* stream);
int frost_save_sigma(const FrostWDesc* descs, float* const* outs, int nlayers, void* stream);
/* STE mask of the logits' activation fake-quant: out = g * [0 <= rint(raw/s)+zp <= 255] */
int frost_mask_logits(const float* g, const float* raw, const float* qrec_y, int64_t n, float* out, void* stream);
int frost_cat_bwd(const uint16_t* gy, const int8_t* a, const float* qrec_a, int ca, const int8_t* b,
                  const float* qrec_b, int cb, int64_t npix, const float* qrec_y, uint16_t* ga, int acc_a,
                  uint16_t* gb, int acc_b, void* stream);
int frost_add_bwd(const uint16_t* gy, const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b,
                  int64_t n, const float* qrec_y, uint16_t* ga, int acc_a, uint16_t* gb, int acc_b, void* stream);
int frost_head_bwd(const float* dlogits_masked, const float* pooled, const int8_t* wq, const float* qrec_w, int n,
                   int cin, int nclass, int hw, const float* drop_mask, float* dwq, float* dbias, uint16_t* gx,
                   float* scratch_dpool, const float* wscale, void* stream);

/* ---- GradBoost optimizers (optimizer.py:121-206, 264-359, 411-512, 564-667) ------------------------- */
/* ---- bf16 inference of the float model (BASELINE.json config c2) ------------------------------------------- */
/* replaces (eval mode): frostnet.py:14-60 ConvBNReLU/ConvBN with BatchNorm folded, :108-121 block wiring, :295-299 head.
 * Activations NHWC bf16; fp32 accumulation; one bf16 rounding per layer output. */
typedef struct FrostIDesc {
  const float* w;        /* OIHW fp32                                                            */
  const float* gamma; const float* beta; const float* rmean; const float* rvar;   /* BN (gamma NULL: no BN, beta = conv bias or NULL) */
  void* pack;            /* kind 0/2: bf16 MFMA A-fragments [cpad/16][kpad/32][64][8]; kind 1: fp32 [kk][cpad] */
  float* biasf;          /* [cpad] folded bias                                                   */
  int32_t cout, cin_g, kk, kind;   /* kind: 0 pointwise, 1 depthwise, 2 stem (im2col K = tap*4 + c)       */
  int32_t cpad, kpad, reserved0, reserved1;
} FrostIDesc;
int frost_infer_weight_prep(const FrostIDesc* descs, int nlayers, void* stream);
int frost_infer_stem_im2col(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, uint16_t* out,
                            void* stream);
/* the stem in ONE launch, no im2col buffer (bit-identical to frost_infer_stem_im2col + frost_infer_pw): pack / biasf = the stem's FrostIDesc.pack / .biasf */
int frost_infer_stem_ok(int cout);
int frost_infer_stem(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const uint16_t* pack,
                     const float* biasf, int cout, int relu, uint16_t* y, void* stream);
int frost_infer_pw(const uint16_t* x, const uint16_t* pack, const float* biasf, int64_t npix, int cin, int cout, int relu,
                   uint16_t* y, void* stream);
int frost_infer_dw(const uint16_t* x, const float* wf, const float* biasf, int n, int h, int w, int c, int k, int stride, int relu,
                   uint16_t* y, void* stream);
int frost_infer_cat(const uint16_t* a, int ca, const uint16_t* b, int cb, int64_t npix, uint16_t* y, void* stream);
int frost_infer_add(const uint16_t* a, const uint16_t* b, int64_t n, uint16_t* y, void* stream);
int frost_infer_avgpool(const uint16_t* x, int n, int hw, int c, float* y, void* stream);
/* ---- fp32-GRADIENT parity mode of the fake-quant backward (csrc/frost_g32.hip) --------------------------------------------------------------
 * replaces: the reference's fp32 autograd (loss.backward(), Classification/utils/helper_functions.py:139-143) for one ConvBn(ReLU)2d + activation FakeQuantize, with
 * the production backward's formulas but fp32 gradient storage and fp64 sums.  Since round 5 the entries run tiled kernels (int8 MFMA conv output, fp32 MFMA data / weight
 * gradients, two-stage deterministic sums through `scratch`, >= frost_g32_scratch_bytes() bytes of device memory; scratch == NULL or a channel count that is not a
 * multiple of 4 selects the round-4 one-thread-per-output kernels).
 *   frost_g32_wq        fake-quantised weight INDICES [cout][per] (OIHW), q = clamp(rint(W * gamma/sigma_r / s_w), -128, 127)
 *   frost_g32_conv_acc  exact int32 conv output acc[p][co] = sum (q_x - zp_x) q_w;  kind 0 pointwise / 2 stem-on-im2col: (n, h, w) = the OUTPUT map, x = [npix][xc];
 *                       kind 1 depthwise: (n, h, w) = the input map
 *   frost_g32_reduce    S1 / S2 rows of `coef` from fp32 gout (STE window on fma(A, acc, B)/s_y, ReLU included);  frost_g32_dc: dc = K1 (gy - S1/n - xhat S2/n), fp32
 *   frost_g32_dgrad     gx (+)= sum dc * q_w * s_w   (kind 0 / 1);   frost_g32_wgrad: dwq[co][j] = s_x sum dc (q_x - zp_x)  -> frost_weight_grad_finalize as usual
 *   frost_g32_cat_bwd / _add_bwd / _pool_bwd: the block wiring and the head's pooled gradient in fp32 */
int frost_g32_wq(const float* w, const float* gamma, const float* sigma, const float* qrec_w, const float* wscale, int cout, int per, int8_t* out, void* stream);
int frost_g32_conv_acc(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride,
                       int32_t* acc, void* stream);
int64_t frost_g32_scratch_bytes(void);
int frost_g32_set_plain(int on);   /* 1: every frost_g32_* entry runs its plain (round-4) kernel -- the yardstick of the fast forms; default 0 (env FROST_G32_PLAIN) */
int frost_g32_reduce(const int32_t* acc, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, const float* gout, void* scratch, void* stream);
int frost_g32_dc(const int32_t* acc, int64_t npix, int cout, const float* coef, const float* qrec_y, int relu, const float* gout, float* dc, void* stream);
/* reduce / dc passes of a pointwise (kind 0) or stem (kind 2) layer that recompute the integer conv output on the int8 MFMA instead of reading the stored one: with them
 * frost_g32_conv_acc is not needed for such a layer (24 -> 12 bytes per output element).  frost_g32_x_ok: the shape has the form (else: conv_acc + reduce + dc). */
int frost_g32_x_ok(int kind, int64_t npix, int xc, int cin_g, int cout);
int frost_g32_reduce_x(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, float* coef,
                       const float* qrec_y, int relu, const float* gout, void* scratch, void* stream);
int frost_g32_dc_x(const int8_t* x, const float* qrec_x, const int8_t* qw, int kind, int n, int h, int w, int xc, int cin_g, int cout, const float* coef,
                   const float* qrec_y, int relu, const float* gout, float* dc, void* stream);
int frost_g32_dgrad(const float* dc, const int8_t* qw, const float* qrec_w, const float* wscale, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k,
                    int stride, float* gx, int accumulate, void* stream);
int frost_g32_wgrad(const float* dc, const int8_t* x, const float* qrec_x, int kind, int n, int h, int w, int xc, int cin_g, int cout, int k, int stride, float* dwq,
                    void* scratch, void* stream);
int frost_g32_cat_bwd(const float* gy, const int8_t* a, const float* qrec_a, int ca, const int8_t* b, const float* qrec_b, int cb, int64_t npix, const float* qrec_y,
                      float* ga, int acc_a, float* gb, int acc_b, void* stream);
int frost_g32_add_bwd(const float* gy, const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n, const float* qrec_y, float* ga, int acc_a,
                      float* gb, int acc_b, void* stream);
int frost_g32_pool_bwd(const float* dpool, const float* drop, int n, int hw, int c, float* gx, void* stream);

/* squeeze_conv emit + cat in ONE launch (frostnet.py:127-129: out = quant_cat.cat([squeeze_conv(x), x], 1)): replaces frost_pw_conv_fwd(mode 1) of the squeeze conv +
 * frost_cat_requant, bit-identical; the cat's record qrec_cat must be final (FrostFinDesc.cat_qrec_y of the squeeze's statistics launch).  y_sq: [npix][r], y_cat: [npix][r + cin]. */
int frost_sq_emit_cat_ok(int cin, int r);
int frost_sq_emit_cat(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, const float* coef,
                      const float* qrec_sq, const float* qrec_cat, int8_t* y_sq, int8_t* y_cat, int mode, void* stream); /* quant_cat's backward (frostnet.py:129) + the squeeze_conv's backward reduce pass in one launch (the backward sibling of frost_sq_emit_cat): g_cat [npix][r + cin]
 * bf16 = gradient of the concatenated tensor; ga [npix][r] (=, or += with acc_a) = its squeeze half inside the cat's STE window (the squeeze layer's gout, read by its
 * dc pass); gb [npix][cin] (+)= the input half; S1 / S2 of the squeeze layer accumulate into its coefficient rows as frost_pw_conv_bwd(mode 0) does.
 * frost_sq_bwd_cat_ok(cin, r) = 1 if the shape qualifies (cin <= 192, r <= 96, multiples of 8). */
int frost_sq_bwd_cat_ok(int cin, int r);
int frost_sq_bwd_cat(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, float* coef,
                     const float* qrec_sq, int relu, const float* qrec_cat, const uint16_t* g_cat, uint16_t* ga, int acc_a, uint16_t* gb, int acc_b, void* stream);
  /* mode 1 / 2 / 3 as frost_pw_conv_fwd's emit modes */

/* ---- whole-bottleneck fused bf16 inference (csrc/frost_iblock.hip) -----------------------------------------------------------
 * replaces (eval mode, BatchNorm folded): CascadePreExBottleneck.forward, frostnet.py:124-145 -- [squeeze_conv -> cat] -> conv1 -> conv2 (depthwise) ->
 * reduce_conv [-> + x] as ONE launch per bottleneck; the expanded tensors stay in LDS, only the block input and output touch HBM.
 * wsq / w1 / w3: bf16 A-fragment packs of frost_infer_weight_prep (kpad = round_up(K, 32)), bsq / b1 / bdw / b3 the folded biases, wdw fp32 taps
 * [k*k][round_up(cexp,16)].  wsq == NULL (r = 0): no squeeze / cat;  w1 == NULL (cexp == cin): the depthwise conv reads the block input (frostnet.py:105-108).
 * (th, tw): spatial output tile of one workgroup; frost_infer_block_ok says whether the geometry / tile fits (LDS <= 160 KB, register budgets). */
int frost_infer_block_ok(int h, int w, int cin, int r, int cexp, int cout, int k, int stride, int th, int tw);
int frost_infer_block(const uint16_t* x, const uint16_t* wsq, const float* bsq, const uint16_t* w1, const float* b1, const float* wdw,
                      const float* bdw, const uint16_t* w3, const float* b3, int n, int h, int w, int cin, int r, int cexp, int cout, int k,
                      int stride, int residual, int th, int tw, int waves, int chunk, uint16_t* y, void* stream);
/* The same bottleneck WITHOUT a squeeze conv (FrostNet's high-resolution blocks, /root/reference/frostnet.py:81-122: [conv1 ->] conv2 -> reduce_conv [-> + x]) with one
 * WAVE per th x tw (4 x 4 / 4 x 8) output tile and no workgroup barriers: conv1's operand straight from HBM into the MFMA fragment layout, 32-channel chunks of the expanded
 * width through 8 KB of wave-private LDS.  Bit-identical to frost_infer_block / the layer launches.  frost_infer_block_w_ok: 1 if the geometry is taken. */
int frost_infer_block_w_ok(int cin, int r, int cexp, int cout, int k, int stride, int has_conv1, int th, int tw);
int frost_infer_block_w(const uint16_t* x, const uint16_t* w1, const float* b1, const float* wdw, const float* bdw, const uint16_t* w3, const float* b3,
                        int n, int h, int w, int cin, int cexp, int cout, int k, int stride, int residual, int th, int tw, uint16_t* y, void* stream);
/* y[n][o] = sum_k x[n][k] * w[o][k] + bias[o], fp32 on the f32 MFMA (classifier of the float model) */
int frost_linear_f32(const float* x, const float* w, const float* bias, int n, int k, int o, float* y, void* stream);

/* ---- float (not fake-quantised) model on the device: StatAssist warm-up training + eval forward ----------------- */
/* replaces: frostnet.py:14-60 ConvBNReLU / ConvBN in TRAIN mode (Conv2d -> BatchNorm2d with batch statistics -> ReLU) and their
 * autograd backward, for the FP epochs Classification/train.py:149-165 runs before prepare_qat.  Activations and activation
 * gradients NHWC bf16 (default) or fp32 (the *_f32 entries below; FrostFDesc.fp32 = 1 makes frost_float_weight_prep write fp32 packs
 * [cpad/16][kpad/16][64][4], kpad / kpad_t multiples of 16), parameters / statistics fp32.  Per conv: forward = mode 0 (statistics) -> frost_float_bn_finalize -> mode 1
 * (emit); backward = mode 2 (S1, S2) -> frost_float_bwd_finalize -> mode 3 (dc) -> data gradient, weight gradient. */
typedef struct FrostFDesc {
  const float* w;        /* OIHW fp32 master weight                                                    */
  const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt;   /* BatchNorm2d parameters / buffers */
  void* pack;            /* kind 0/2: bf16 MFMA A-fragments [cpad/16][kpad/32][64][8]; kind 1: fp32 [kk][cpad] */
  void* pack_t;          /* kind 0: A-fragments of W^T [round16(cin)/16][kpad_t/32][64][8] (data gradient); else NULL */
  double* stat;          /* [4*cpad]: sum, sum of squares (forward), S1, S2 (backward); zeroed by frost_float_weight_prep */
  float* coef;           /* [8*cpad]: scale, bias, mean, invsigma, K1, E, F, batch variance                    */
  float* dgamma; float* dbeta;   /* gradient destinations (accumulated into)                                   */
  int32_t cout, cin_g, kk, kind;   /* kind: 0 pointwise, 1 depthwise, 2 stem (im2col K = tap*4 + c)             */
  int32_t cpad, kpad, kpad_t, fp32;
} FrostFDesc;
int frost_float_weight_prep(const FrostFDesc* descs, int nlayers, void* stream);
int frost_float_bn_finalize(const FrostFDesc* desc, int cout, int64_t count, void* stream);   /* train: batch stats -> coef, running stats */
int frost_float_bn_eval(const FrostFDesc* descs, int nlayers, void* stream);                  /* eval: running stats -> coef            */
int frost_float_bwd_finalize(const FrostFDesc* desc, int cout, int64_t count, void* stream);
/* mode 0..3 as above with the layer's forward pack (the data gradient dx = dc . W^T is frost_infer_pw on pack_t, no bias).
 * gy / out rows may be slices of wider tensors: ldg / ldy = row length in elements. */
int frost_float_pw(const FrostFDesc* desc, const uint16_t* x, const uint16_t* pack, int64_t npix, int cin, int cout, int relu, int mode,
                   const uint16_t* gy, int ldg, uint16_t* out, int ldy, void* stream);
int frost_float_dw(const FrostFDesc* desc, const uint16_t* x, int n, int h, int w, int c, int k, int stride, int relu, int mode,
                   const uint16_t* gy, uint16_t* out, void* stream);
/* element-wise passes over a KEPT conv output (mode 0 of frost_float_pw / frost_float_dw stores the convolution output into `out` when it is
 * non-NULL): mode 1 y = [relu](conv*scale + bias), mode 2 the backward statistics, mode 3 dc -- the same results as modes 1..3 of the conv entries
 * without recomputing the convolution (replaces: the second evaluation of F.conv2d inside _forward_approximate and autograd's saved tensors). */
int frost_float_ew(const FrostFDesc* desc, const uint16_t* conv, int64_t npix, int c, int relu, int mode, const uint16_t* gy, int ldg,
                   uint16_t* out, int ldy, void* stream);
int frost_float_ew_f32(const FrostFDesc* desc, const float* conv, int64_t npix, int c, int relu, int mode, const float* gy, int ldg,
                       float* out, int ldy, void* stream);
/* The depthwise layer of a bottleneck fed from the KEPT CONV OUTPUT of the layer in front (conv1) instead of its activation: y1 = [relu](c1 * scale + bias)
 * (BatchNorm2d + ReLU of frostnet.py:14-30, coefficient rows of desc_src) is applied as the input window is loaded, so the element-wise emit pass of conv1
 * (frost_float_ew mode 1: one read of c1, one write of y1) never runs in training.  frost_float_dw_src: forward modes 0 / 1 of frost_float_dw;
 * frost_float_dw_wgrad_src: the weight gradient of the same layer (it needs the same input).  fp32 mode: identical values to the two-pass form; bf16 mode: the
 * input is not rounded to bf16 on the way (closer to the reference's fp32). */
int frost_float_dw_src(const FrostFDesc* desc, const uint16_t* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                       int relu, int mode, uint16_t* out, void* stream);
int frost_float_dw_wgrad_src(const uint16_t* dc, const uint16_t* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                             float* dw, void* stream);
int frost_float_dw_dgrad(const FrostFDesc* desc, const uint16_t* dc, int n, int h, int w, int c, int k, int stride, uint16_t* dx, void* stream);
int frost_float_dw_wgrad(const uint16_t* dc, const uint16_t* x, int n, int h, int w, int c, int k, int stride, float* dw, void* stream);
int frost_float_pw_wgrad(const uint16_t* dc, const uint16_t* x, int64_t npix, int cin, int ldx, int cout, float* dw, int ldw, void* stream);
int frost_float_stem_wscatter(const float* tmp, int cout, float* dw, void* stream);
int frost_float_grad_merge(const uint16_t* res, const uint16_t* cat, int cs, int ccat, const uint16_t* sq, int64_t npix, int c, uint16_t* out,
                           void* stream);
int frost_float_avgpool(const uint16_t* x, int n, int hw, int c, const float* drop, float* y, void* stream);
int frost_float_head_bwd(const float* dlogits, const float* pooled, const float* wfc, int n, int cin, int nclass, int hw,
                         const float* drop_mask, float* dw, float* dbias, uint16_t* gx, float* scratch_dpool, void* stream);
/* fp32 activation storage (the reference's own precision; products on the fp32 MFMA): same arguments, float elements.
 * frost_float_pw_f32 mode 4 = plain GEMM out = x . pack^T (the data gradient on pack_t; desc may be NULL). */
int frost_float_pw_f32(const FrostFDesc* desc, const float* x, const float* pack, int64_t npix, int cin, int cout, int relu, int mode,
                       const float* gy, int ldg, float* out, int ldy, void* stream);
int frost_float_dw_f32(const FrostFDesc* desc, const float* x, int n, int h, int w, int c, int k, int stride, int relu, int mode,
                       const float* gy, float* out, void* stream);
int frost_float_dw_src_f32(const FrostFDesc* desc, const float* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                           int relu, int mode, float* out, void* stream);
int frost_float_dw_wgrad_src_f32(const float* dc, const float* conv_src, const FrostFDesc* desc_src, int relu_src, int n, int h, int w, int c, int k, int stride,
                                 float* dw, void* stream);
int frost_float_dw_dgrad_f32(const FrostFDesc* desc, const float* dc, int n, int h, int w, int c, int k, int stride, float* dx, void* stream);
int frost_float_dw_wgrad_f32(const float* dc, const float* x, int n, int h, int w, int c, int k, int stride, float* dw, void* stream);
int frost_float_pw_wgrad_f32(const float* dc, const float* x, int64_t npix, int cin, int ldx, int cout, float* dw, int ldw, void* stream);
int frost_float_grad_merge_f32(const float* res, const float* cat, int cs, int ccat, const float* sq, int64_t npix, int c, float* out, void* stream);
int frost_float_avgpool_f32(const float* x, int n, int hw, int c, const float* drop, float* y, void* stream);
int frost_float_head_bwd_f32(const float* dlogits, const float* pooled, const float* wfc, int n, int cin, int nclass, int hw,
                             const float* drop_mask, float* dw, float* dbias, float* gx, float* scratch_dpool, void* stream);
int frost_float_cat_f32(const float* a, int ca, const float* b, int cb, int64_t npix, float* y, void* stream);      /* y = cat([a, b], channels) */
int frost_float_add_f32(const float* a, const float* b, int64_t n, float* y, void* stream);
int frost_float_stem_im2col_f32(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* out, void* stream);

typedef struct FrostOptTensor {
  float* p; float* g; float* exp_min; float* exp_max; float* coin; float* buf0; float* buf1; float* buf2;
  int64_t n; float weight_decay; float lr; int32_t first_step; int32_t pad;
} FrostOptTensor;
typedef struct FrostOptHyper {
  int32_t kind;            /* 0 QSGD, 1 QRMS, 2 QAdam, 3 QAdamW */
  int32_t boost;           /* !is_warmup */
  int32_t toss_coin, nesterov, amsgrad, centered;
  float beta, momentum, dampening, alpha, eps, beta1, beta2, clip_by;
  float bc_beta;           /* 1 - beta^step            */
  float noise_scale;       /* (1-noise_decay)^restart_step */
  float bc1, bc2;          /* Adam: bc1 = 1-beta1^t ; bc2 = sqrt(1-beta2^t) (already square-rooted) */
  uint64_t seed, offset;   /* Philox stream when noise/coin are not injected */
} FrostOptHyper;
/* noise/coin: optional injected tensors (concatenated in table order) for parity tests; NULL = on-device Philox */
/* table, hyper, prefix (int64 element offset of each tensor in the noise/coin/Philox index space) are DEVICE
 * memory so a captured hipGraph picks up new lr / bias corrections / offsets without re-capture. */
int frost_gradboost_step(const FrostOptTensor* table, int ntensors, int64_t max_n, const FrostOptHyper* hyper,
                         const float* noise, const float* coin, const int64_t* prefix, void* stream);

/* ---- statistics pass with the finalize folded into its tail --------------------------------------------------------------------------
 * frost_pw_conv_fwd / frost_dw_conv_fwd (mode 0) + frost_conv_finalize in ONE launch: the last workgroup to finish (device-scope ticket)
 * turns the integer statistics into the BN coefficients / running statistics / activation qparams, so a conv forward is two launches
 * (statistics+finalize, emit) instead of three.  `fin` is a HOST struct (copied into the kernel arguments); `counter` is FROST_TICKET_WORDS
 * zeroed uint32 per layer in device memory (two-level arrival ticket: 32 sub-counters, re-armed by the kernel). */
#define FROST_TICKET_WORDS 40
typedef struct {
  const float* qrec_w; const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt;
  float* coef; float* qrec_y; uint32_t* counter;
  int32_t training, relu, observe, reserved;
  const float* wscale;     /* [cpad] per-output-channel weight scale (FrostWDesc.wscale); NULL = qrec_w's scalar */
  /* block-level fold: when this layer is a bottleneck's squeeze_conv, the FakeQuantize of quant_cat.cat([squeezed, x]) (frostnet.py:129)
   * is updated right here, in the same last-workgroup tail, from the fake-quantised ranges of the two halves -- frost_cat_observe without its launch.
   * cat_qrec_b = the qrecord of x (the second half), cat_qrec_y = the cat site's qrecord; both NULL = no fold. */
  const float* cat_qrec_b; float* cat_qrec_y;
} FrostFinDesc;
int frost_pw_conv_fwd_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                          void* stats, const FrostFinDesc* fin, void* stream);
/* The whole squeeze_conv forward (frostnet.py:127-129) as ONE persistent launch: integer statistics of the 1x1 conv -> device-wide barrier whose last arrival runs the
 * finalize (BatchNorm coefficients, running statistics, the conv's and the cat's FakeQuantize records: FrostFinDesc incl. cat_qrec_b / cat_qrec_y) -> emit + both halves of
 * the cat from the accumulators each workgroup kept.  Replaces frost_pw_conv_fwd_fin + frost_sq_emit_cat (same expressions; the integer statistics are exact here, so
 * coefficient rows agree to ~1e-7).  slots: >= frost_sq_fwd_slot_bytes(npix, r) bytes of scratch (per-workgroup statistics rows).  Every workgroup must be resident at once:
 * frost_sq_fwd_ok(npix, cin, r) checks ceil(npix / 128) against the device's capacity for the instance (and the shape); ticket words 36 / 37 of fin->counter are the
 * barrier's generation word (monotonic) and its give-up flag (non-zero = a workgroup stopped waiting: results invalid). */
int frost_sq_fwd_ok(int64_t npix, int cin, int r);
int64_t frost_sq_fwd_slot_bytes(int64_t npix, int r);
int frost_sq_fwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int r, void* stats, const FrostFinDesc* fin,
                 void* slots, int8_t* y_sq, int8_t* y_cat, void* stream);
int frost_dw_conv_fwd_fin(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k,
                          int stride, void* stats, const FrostFinDesc* fin, void* stream);
/* forward of a wide-K pointwise layer (see frost_pw_conv_int / frost_pw_ew) keeping the integer conv output: statistics pass + folded finalize (the job of frost_pw_conv_fwd_fin) on the stand-alone GEMM
 * kernel, conv_out stored; frost_pw_ew mode 2 emits y from it, and the backward needs no recomputation at all */
int frost_pw_conv_fwd_keep(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                           void* stats, const FrostFinDesc* fin, int32_t* conv_out, void* stream);

/* skip_add's backward (frostnet.py:142) folded into the element-wise reduce (mode 0) / dc (mode 1) passes of the reduce_conv whose output yb was the add's
 * second operand (kept integer conv output conv_out): g = gsum where q_sum's FakeQuantize passed fake(a) + fake(yb), else 0 (= frost_add_bwd); mode 0 also
 * writes the residual branch's gradient ga (+)= g; the reduce_conv's own gout is never materialised.  Bit-identical to frost_add_bwd + frost_pw_ew. */
int frost_pw_ew_add_bwd(const int32_t* conv_out, int64_t npix, int cout, float* coef, const float* qrec_y, int relu, int mode, const uint16_t* gsum,
                        const int8_t* a, const float* qrec_a, const int8_t* yb, const float* qrec_sum, uint16_t* ga, int acc_a, uint16_t* dc, void* stream);
/* ---- block-level fusion across the reduce_conv -> skip_add boundary of a Frost bottleneck -------------------------------------
 * replaces: the activation FakeQuantize of reduce_conv (emit: frost_pw_ew mode 2) TOGETHER with the observer pass of FloatFunctional.add(x, out)
 * (frostnet.py:138-142; frost_add_minmax_observe): y = emit(conv_out) is written and min / max of dequant(a) + dequant(y) -- the reference's fp32
 * expression -- go through the MovingAverageMinMax update of `qrec_sum` in the kernel's last workgroup.  frost_add_requant then emits the sum.
 * state3 = {lo, hi, ticket} as for frost_add_minmax_observe. */
int frost_pw_ew_emit_add(const int32_t* conv_out, int64_t npix, int cout, const float* coef, const float* qrec_y, int relu, const int8_t* a,
                         const float* qrec_a, int8_t* y, float* state3, float* qrec_sum, int observe, void* stream);

/* ---- reduce / dc passes of WIDE pointwise layers in the chunked layout (csrc/frost_pwc.hip) -----------------------------------------------------
 * replaces: what frost_pw_conv_bwd passes 0 / 1 replace (the backward of nniqat.ConvBn(ReLU)2d up to dc), for layers whose gout / dc rows are too long for k_pw's
 * LDS-staged tile I/O (Cout >= 256, input rows <= 320 bytes): a workgroup owns a 64-pixel tile and walks 64-channel chunks, so the gout window is read and the dc
 * window written with full 128-byte lines.  Same expressions as k_pw; pass 0 accumulates S1 / S2 into the coefficient rows, pass 1 writes dc (bf16). */
int frost_pwc_bwd_ok(int64_t npix, int cin, int cout);
/* the forward emit pass of the same layers (frost_pw_conv_fwd mode 1; the converted-inference modes stay on k_pw): y leaves through an LDS window as 64-byte row pieces */
int frost_pwc_conv_fwd_emit(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout,
                            const float* coef, const float* qrec_y, int8_t* y, void* stream);
int frost_pwc_conv_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int64_t npix, int cin, int cout, int pass,
                       float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dc, void* stream);

/* ---- whole-bottleneck forward fusion at the 14x14 / 7x7 stages (CascadePreExBottleneck.forward, frostnet.py:124-145) ------------
 * frost_block_supported: 1 when the fused kernels have an instance for a (conv1 -> conv2) pair: square 7x7 / 14x14 maps, depthwise k in {3,5} at
 * stride 1, conv1 input rows of <= 320 (7x7) / 192 (14x14) bytes.
 * frost_block_expand_dw_stats replaces: conv1's emit pass (frost_pw_conv_fwd mode 1: BN + ReLU + activation FakeQuantize of `self.conv1`, frostnet.py:134)
 * TOGETHER with conv2's statistics pass and folded finalize (frost_dw_conv_fwd_fin: batch statistics, running-stat update and observer update of
 * `self.conv2`, frostnet.py:137).  One workgroup per image: conv1's quantised output goes through an LDS plane straight into the depthwise stencil
 * and leaves for HBM once (y1: conv2's emit pass and the backward read it).  conv1 must be finalized (coef1 / qrec_y1 final); results are
 * bit-identical to the two separate launches. */
int frost_block_supported(int h, int w, int k, int stride, int cin, int c);
/* conv2's statistics pass + folded finalize alone, image-resident (the contract of frost_dw_conv_fwd_fin on 14x14 / 7x7 maps, stride 1): with frost_pwc_conv_fwd_emit in
 * front of it the two-launch alternative to frost_block_expand_dw_stats */
int frost_block_dw_stats(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k, void* stats,
                         const FrostFinDesc* fin, void* stream);
int frost_block_expand_dw_stats(const int8_t* x, const float* qrec_x, const int8_t* w1_pack, const int32_t* wsum1, const float* coef1,
                                const float* qrec_y1, int8_t* y1, int n, int h, int w, int cin, int c, const int8_t* wq2, const int32_t* wsum2,
                                int k, void* stats2, const FrostFinDesc* fin2, void* stream);

/* frost_block_dw_reduce replaces: conv2's emit pass (frost_dw_conv_fwd mode 1: BN + ReLU + activation FakeQuantize of `self.conv2`, frostnet.py:137)
 * TOGETHER with reduce_conv's statistics pass on the kept-output path (frost_pw_conv_fwd_keep: int8 GEMM, integer conv output stored, batch statistics,
 * folded finalize of `self.reduce_conv`, frostnet.py:138).  One workgroup per image, K-split over 64-channel chunks: y1 chunk -> LDS plane ->
 * depthwise -> quantised y2 chunk (LDS; out to HBM once for the backward) -> MFMA partial sums of reduce_conv in registers across the chunks.
 * conv2 must be finalized; conv_out / stats3 / fin3 as for frost_pw_conv_fwd_keep (frost_pw_ew mode 2 then emits reduce_conv's y).  Bit-identical
 * to the two separate launches.  frost_block_dw_reduce_supported: 7x7 / 14x14 maps, k in {3,5}, stride 1, cout <= 320 (7x7) / 128 (14x14). */
int frost_block_dw_reduce_supported(int h, int w, int k, int stride, int c, int cout);
int frost_block_dw_reduce(const int8_t* y1, const float* qrec_y1, const int8_t* wq2, const int32_t* wsum2, const float* coef2, const float* qrec_y2,
                          int relu2, int8_t* y2, int n, int h, int w, int c, int k, const int8_t* w3_pack, const int32_t* wsum3, int cout,
                          int32_t* conv_out, void* stats3, const FrostFinDesc* fin3, void* stream);

/* frost_block_dw_bwd replaces, for conv2 of such a block: frost_dw_conv_bwd pass 1 (dc) + frost_dw_wgrad + frost_dw_dgrad -- the backward of
 * nniqat.ConvBnReLU2d (depthwise) after its reduce pass (frost_dw_conv_bwd pass 0 filled the S1 / S2 coefficient rows): dc is assembled in an LDS plane
 * per image and 64-channel chunk, the raw weight-gradient sums (dwq[c][k*k], as frost_dw_wgrad) and the data gradient (dx, bf16, overwritten; may be
 * NULL) come straight from it; dc is not written to HBM.  wscale: per-channel weight scales or NULL (qrec_w's scalar).  Same expressions and, for the
 * data gradient, the same summation order as the separate kernels; dc's stochastic rounding draws differ (generator seeded per workgroup / thread). */
int frost_block_dw_bwd_supported(int h, int w, int k, int stride, int c);
/* the reduce pass of the same layers (replaces frost_dw_conv_bwd pass 0): S1 = sum gy, S2 = sum gy * xhat accumulated into the coefficient rows, image-resident
 * like frost_block_dw_bwd (lane = channel, the sums in registers across a workgroup's images, one pair of float atomics per channel and workgroup) */
int frost_block_dw_bwd_reduce(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, int n, int h, int w, int c, int k,
                              float* coef, const float* qrec_y, int relu, const uint16_t* gout, void* stream);
int frost_block_dw_bwd(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                       int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                       float* dwq, void* stream);
/* frost_block_dw_bwd carrying the backward REDUCE pass of the pointwise layer that produced x (conv1 of the bottleneck: frost_pw_conv_bwd pass 0 on c1_x -> x): conv1's
 * integer output of every (image, 64-channel chunk) is recomputed on the matrix cores from c1_x [n*h*w][c1_cin] (record c1_qrec_x) with conv1's weight pack / weight sums, and
 * its S1 / S2 rows of c1_coef accumulate from the dx values BEFORE their bf16 rounding; conv1's backward then starts at its dc pass.  dx is required.
 * frost_block_dw_bwd_c1_ok() = 1 if an instance exists (7 x 7: c1_cin <= 320; 14 x 14: <= 128 for k = 5, <= 192 for k = 3; FROST_BLK_C1 masks map sizes). */
int frost_block_dw_bwd_c1_ok(int h, int w, int k, int stride, int c, int c1_cin);
int frost_block_dw_bwd_c1(const int8_t* x, const float* qrec_x, const int8_t* wq_pack, const int32_t* wsum, const float* qrec_w, const float* wscale,
                          int n, int h, int w, int c, int k, const float* coef, const float* qrec_y, int relu, const uint16_t* gout, uint16_t* dx,
                          float* dwq, const int8_t* c1_x, const float* c1_qrec_x, const int8_t* c1_wq_pack, const int32_t* c1_wsum, float* c1_coef, int c1_cin,
                          int c1_relu, void* stream);

/* ---- frost_block_fwd / frost_block_bwd: the conv1 -> conv2 -> reduce_conv chain of one bottleneck as ONE call each (SURVEY 8(b) "frost_block_fwd/bwd") ------
 * replaces: `out = self.reduce_conv(self.conv2(self.conv1(out)))` of CascadePreExBottleneck.forward (frostnet.py:134-138) in training mode with observers on, and
 * its backward.  Host-side composites over the entries above (what Engine.conv_pair / Engine._conv_backward sequence): every launch goes to `stream`, nothing is
 * allocated -- all buffers are the caller's.  Supported where frost_block_supported and frost_block_dw_reduce_supported hold (14x14 / 7x7 maps, depthwise
 * stride 1, reduce_conv narrower than the expanded tensor); other blocks run layer by layer.
 *   forward : frost_pw_conv_fwd_fin on conv1 -> frost_block_expand_dw_stats -> frost_block_dw_reduce -> frost_pw_ew mode 2 (y3)
 *   backward: reduce_conv [frost_pw_ew 0, 1 -> frost_pw_dgrad_wide -> frost_pw_wgrad]; conv2 [frost_block_dw_bwd_reduce -> frost_block_dw_bwd];
 *             conv1 [reduce pass -> frost_pwc_conv_bwd / frost_pw_conv_bwd pass 1 -> frost_pw_dgrad_wide -> frost_pw_wgrad].  Raw weight-gradient sums land in
 *             `dwq` of each layer (zeroed by the caller; frost_weight_grad_finalize[_table] turns them into dW / dgamma / dbeta as for any layer).
 *             side_stream (may be NULL): the two pointwise weight gradients run there, forked / joined with events inside the call. */
typedef struct {
  const int8_t* wq_pack; const int32_t* wsum; void* stats; FrostFinDesc fin;     /* forward operands of one ConvBN(ReLU) layer; fin.coef / fin.qrec_y are its rows / output record */
  const uint16_t* wt_pack; float* dwq;                                           /* backward: transposed bf16 pack (pointwise layers; unused for conv2), raw weight-gradient sums */
  int32_t cout, k;
} FrostBlockLayer;
typedef struct {
  int32_t n, h, w, cin;                        /* conv1's input: n images of h x w x cin */
  const int8_t* x; const float* qrec_x;
  FrostBlockLayer conv1, conv2, reduce;
  int8_t* y1; int8_t* y2; int32_t* conv_out3; int8_t* y3;       /* conv1 / conv2 outputs (kept for the backward), reduce_conv's integer output and emitted output */
} FrostBlockDesc;
typedef struct {
  const uint16_t* gout3;                       /* gradient w.r.t. y3 (bf16, [n*h*w][cout3]) */
  uint16_t* dc3; uint16_t* g2; uint16_t* g1; uint16_t* dc1;     /* work buffers: dc of reduce_conv, gradients w.r.t. y2 and y1, dc of conv1 (bf16, sized like the tensors) */
  uint16_t* dx;                                /* gradient w.r.t. x (bf16, overwritten); NULL = not needed */
  void* side_stream;
} FrostBlockBwd;
int frost_block_fwd(const FrostBlockDesc* d, void* stream);
int frost_block_bwd(const FrostBlockDesc* d, const FrostBlockBwd* b, void* stream);

/* ---- loss and dropout mask of the training step -------------------------------------------------------------------
 * replaces: nn.CrossEntropyLoss(reduction='mean') forward + backward (Classification/train.py:147, helper_functions.py:140-142).
 * loss: one float (caller zeroes it; accumulated with atomics); dlogits = (softmax - onehot) * inv_n (may be NULL); target < 0 ignored. */
int frost_softmax_ce(const float* logits, const int64_t* target, int n, int c, float inv_n, float* loss, float* dlogits, void* stream);
/* replaces: nn.Dropout's Bernoulli mask on the pooled features (frostnet.py:297): out[i] in {0, 1/keep}, Philox4x32-10, counter =
 * (index, draw_counter[0]); draw_counter = TWO device-resident uint64 {draw, arrival ticket (zero)}: the last workgroup to finish advances the
 * draw counter itself (hipGraph replays draw fresh masks).  out: 16-byte aligned. */
int frost_dropout_mask(void* draw_counter, uint64_t seed, int64_t n, float keep, float* out, void* stream);

/* ---- converted int8 inference ------------------------------------------------------------------
 * replaces: torch.quantization.convert(model.eval()) + the QNNPACK kernels (Classification/evaluate.py:130-134).  The convolutions are
 * frost_pw_conv_fwd / frost_dw_conv_fwd with mode 2 (integer bias add + fp32 requantisation: y = clamp(rint(float(acc + b_q) * rs) + zp));
 * their coefficient rows come from frost_conv_finalize_converted (BN folded with the running statistics, bias at scale s_x*s_w);
 * weights are prepared once by frost_weight_prep with FrostWDesc.reserved0 = 1. */
int frost_conv_finalize_converted(const float* qrec_x, const float* qrec_w, const float* gamma, const float* beta,
                                  const float* rmean, const float* rvar, int cout, float* coef, const float* qrec_y, void* stream);
/* the same for a model prepared with the per-channel 'fbgemm' qconfig and converted on the FBGEMM engine (Classification/latency_check.py:221-226):
 * row A = (s_x * s_w[c]) / s_y, row B = b_fold[c] / (s_x * s_w[c]) (float) -- the emit kernels in mode 3 then compute
 * q = cvtps2dq((float(acc) + B) * A) + zp  (aten qconv.cpp -> fbgemm::ReQuantizeOutput with a float bias).  The residual add of that engine is the
 * float add frost_add_requant already computes; cat / average pool are shared with the QNNPACK form. */
int frost_conv_finalize_converted_fb(const float* qrec_x, const float* wscale, const float* gamma, const float* beta, const float* rmean,
                                     const float* rvar, int cout, float* coef, const float* qrec_y, void* stream);
int frost_classifier_q_fb(const int32_t* pooled, const float* qrec_x, const int8_t* wq, const float* coef, int n, int c, int cout,
                          const float* qrec_y, float* logits, uint8_t* idx, void* stream);
/* replaces: quantized::add (QNNPACK q8add, integer fixed point) -- FloatFunctional.add of a converted model (frostnet.py:142) */
int frost_add_qnnpack(const int8_t* a, const float* qrec_a, const int8_t* b, const float* qrec_b, int64_t n, const float* qrec_y,
                      int8_t* y, void* stream);
/* replaces: adaptive_avg_pool2d(1) on a quantised tensor: int32 index rint_half_even(mean), input qparams kept (frostnet.py:296) */
/* QuantStub + the converted stem conv in ONE launch from the fp32 image (replaces frost_quantize_input + frost_stem_im2col + frost_pw_conv_fwd mode 2 / 3,
 * bit-identical indices): frostnet.py:250, 319-320 after torch.quantization.convert */
int frost_stem_converted_ok(int cout);
int frost_stem_converted(const float* x, int n, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* qrec_x,
                         const int8_t* wq_pack, const int32_t* wsum, const float* coef, const float* qrec_y, int cout, int mode, int8_t* y,
                         void* stream);
int frost_avgpool_q(const int8_t* x, int n, int hw, int c, int32_t* pooled, void* stream);
/* replaces: quantized::conv2d of the classifier (frostnet.py:298) + DeQuantStub: exact int32 GEMV, coefficient rows as above */
int frost_classifier_q(const int32_t* pooled, const float* qrec_x, const int8_t* wq, const float* coef, int n, int c, int cout,
                       const float* qrec_y, float* logits, uint8_t* idx, void* stream);

/* ---- quantizable hard-swish (the reference keeps it in its MobileNetV3 baselines, FrostNet itself uses ReLU) -------------
 * replaces: `_Hswish` (Classification/models/imagenet/mobilenetv3.py:43-56): relu6(add_scalar(x, 3)) with the FakeQuantize prepare_qat hangs on
 * nn.ReLU6 (qrec_relu6), FloatFunctional.mul(x, .) with its FakeQuantize (qrec_site), then mul_scalar(1/6) (qrec_out: same indices, scale / 6).  x is an int8
 * activation (<= 256 distinct values): presence bitmap -> observer -> 256-entry tables (lut: 256 bytes forward + 256 floats backward).
 * present8: 8 zeroed uint32 (re-armed by the call).  Backward: dx (+)= gout * [STE mask * f'(x) / 6], bf16. */
int frost_hswish_fwd(const int8_t* x, const float* qrec_x, int64_t n, uint32_t* present8, float* qrec_relu6, float* qrec_site, float* qrec_out,
                     int observe, uint8_t* lut, int8_t* y, void* stream);
int frost_hswish_bwd(const uint16_t* gout, const int8_t* x, int64_t n, const uint8_t* lut, uint16_t* dx, int accumulate, void* stream);
/* replaces: the same `_Hswish` after torch.quantization.convert (Classification/evaluate.py:130-134 on a hard-swish network): QFunctional.add_scalar(x, 3) ->
 * nnq.ReLU6 -> QFunctional.mul(x, .) at quant_mul1's frozen record (qrec_site) -> mul_scalar(1/6), integer arithmetic on quint8 indices restated as a 256-entry
 * table of the input index (lut: >= 256 bytes).  qrec_out is written: qrec_site's indices at scale double(s) * (1/6).  n % 4 == 0.
 * qrec_site == NULL: `lut` and qrec_out were built by an earlier call on the same (frozen) records -- the table pass alone. */
int frost_hswish_converted(const int8_t* x, const float* qrec_x, int64_t n, const float* qrec_site, float* qrec_out, uint8_t* lut, int8_t* y, void* stream);

/* ---- SSD MultiBoxLoss (Object_Detection/layers/modules/multibox_loss.py:48-117 + layers/box_utils.py:71-139) ------------------------------------------------
 * frost_mbox_forward: loc [n][p][4], conf [n][p][c], priors [p][4] (cx, cy, w, h), boxes [n][k][5] (x1, y1, x2, y2, label) with valid [n][k] (padding rows 0).
 * Work buffers of the caller: bto [n][p] float (best overlaps during the matching, afterwards scratch: per-prior smooth-L1 terms, then the image's two sums in its
 * first two words -- the losses are summed in a fixed order, bit-reproducible run to run), bti [n][p] int32, loc_t [n][p][4], conf_t [n][p] int32, lc [n][p], sel [n][p] bytes, num_pos [n] int32.
 * out: frost_mbox_workspace_floats() floats, ZEROED before the first call: {loss_l, loss_c, 1 / N_pos, three running words that every call leaves zeroed, ...}.
 * frost_mbox_backward: dloc [n][p][4], dconf [n][p][c] from the saved loc_t / conf_t / sel / out and the device scalars g_l, g_c (gradients of the two losses). */
int frost_mbox_workspace_floats(void);
int frost_mbox_forward(const float* loc, const float* conf, const float* priors, const float* boxes, const uint8_t* valid, int n, int p, int c, int k,
                       float threshold, int negpos, float var0, float var1, float* bto, int32_t* bti, float* loc_t, int32_t* conf_t, float* lc, uint8_t* sel,
                       int32_t* num_pos, float* out, void* stream);
int frost_mbox_backward(const float* loc, const float* conf, const float* loc_t, const int32_t* conf_t, const uint8_t* sel, const float* out, const float* g_l,
                        const float* g_c, int n, int p, int c, float* dloc, float* dconf, void* stream);

#ifdef __cplusplus
}
#endif
#endif
