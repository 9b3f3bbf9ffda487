/*
 * qllm_mi355x.h -- C ABI of libqllm_mi355x.so: the MI355X (gfx950 / CDNA4) replacement for the native
 * extensions behind QLLM's quantized-linear forward (fused int4 dequant + matmul).
 *
 * Boundary it replaces (all citations relative to /root/reference):
 *   qllm.ort_ops.gemv        csrc/ort_cuda/ort_ops.cc:94-140   (op_gemv -> dq_gemv.cu gemv<half> / Gemv_g)
 *   qllm.ort_ops.dequant     csrc/ort_cuda/ort_ops.cc:58-92    (dequant_any_bit -> DequantizeAndUnpackWeight*)
 *   qllm.awq_inference_engine.gemm_forward_cuda
 *                            csrc/awq_cuda/quantization/gemm_cuda.h:3-4, gemm_cuda_gen.cu:1102-1161
 *   QuantLinearTorchFunction.forward (+bias) of the three q_layers
 *                            qllm/modeling/q_layers/quant_linear_gptq.py:71-85,136-143
 *                            qllm/modeling/q_layers/quant_linear_hqq.py:31-38,76-80
 *                            qllm/modeling/q_layers/quant_linear_awq.py:142-148
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer on the current HIP device unless noted;
 *     inputs are borrowed, contiguous, row-major; the library never allocates or frees device memory and never
 *     synchronises the host with the device -- every entry point is hipGraph-capturable.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The reference's default-stream
 *     launches (dq_gemv.cu:166,563; gemm_cuda_gen.cu:1083) are NOT reproduced.
 *   - every function returns a qllm_status_t (0 = ok).  Nothing aborts (contrast dq_gemv.cu:156-159,172-176);
 *     the message for the calling thread's last failure is qllm_last_error().
 *   - weight layouts are exactly the reference's state-dict buffers:
 *       GPTQ     qweight i32 [K*bits/32, N] (column n = little-endian bit stream along K)
 *                qzeros  i32 [ceil(K/g), N*bits/32] (row G = bit stream along N), or NULL => symmetric 2^(bits-1)
 *                scales  f16 [ceil(K/g), N];  g_idx i32 [K] or NULL (NULL = k / group_size)
 *       AWQ_GEMM qweight i32 [K, N/8], nibble i of word (k,j) = q[k, 8j+{0,2,4,6,1,3,5,7}[i]]; qzeros i32 [K/g, N/8]
 *                same interleave; scales f16 [K/g, N]; 4-bit only; no g_idx
 *       HQQ      qweight as GPTQ; qzeros f16 [ceil(K/g), N] (un-packed, non-integer); no g_idx
 *   - numerics, by kernel path (qllm_plan_describe() names the path a call takes):
 *       qllm_dequant / qllm_ort_dequant, the tile GEMMs ("gemm3": the wave-specialised 256x128 kernel, the default from M = 384 /
 *       768; "gemm2": 33 / 65 <= M below that and bf16 activations below 1024 rows; "gemm": ragged N, non-uniform act-order) and the
 *       split-K decode kernel ("skinny"):
 *         W[k,n] = fp16( fp16(s*q) - fp16(z*s) ) exactly as DequantizeLinearBlockWise (quant_linear_gptq.py:38-48) -- one IEEE
 *         rounding per op, bit-identical to the CPU path -- then y = x.W accumulated in fp32, bias added in fp32, one
 *         rounding to the activation dtype (bf16 activations on "gemm3": x is converted to fp16 first and the fp16 result is
 *         rounded to bf16, the arithmetic of the reference's own shim, quant_linear_awq.py:29-36).
 *       the full-K decode kernel ("strip": M <= 32 everywhere, M <= 64 for K <= 4096 and at most 4096 columns; every layout it
 *       serves -- the reference row streams in place and the native strip-major layout; the default decode path):
 *         y = sum_G s_G * ( sum_{k in G} x_k q_k  -  z_G * sum_{k in G} x_k ) evaluated in fp32, i.e. x.W for the UNROUNDED
 *         W = s*(q - z); it differs from the path above by the fp16 rounding noise of W (<= 3e-4 relative measured; the
 *         tests bound every decode case at 2e-3 against float64 of the reference's W and at 1e-2 against the CPU path).
 *     The bit-exactness guarantee therefore holds for the dequant entry points and the paths of the first group only.
 */
#ifndef QLLM_MI355X_H_
#define QLLM_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLLM_ABI_VERSION 6

typedef enum qllm_status {
  QLLM_OK = 0,
  QLLM_ERR_INVALID = 1,     /* bad shape / null pointer / misalignment (the reference's TORCH_CHECK / invalid_argument) */
  QLLM_ERR_UNSUPPORTED = 2, /* valid request this build has no fused kernel for (caller may use dequant + GEMM) */
  QLLM_ERR_WORKSPACE = 3,   /* workspace NULL or smaller than qllm_workspace_bytes() */
  QLLM_ERR_LAUNCH = 4,      /* HIP launch failure (message carries hipGetErrorString) */
  QLLM_ERR_DEVICE = 5       /* no gfx950 device / wrong architecture */
} qllm_status_t;

typedef enum qllm_layout {
  QLLM_LAYOUT_GPTQ = 0,     /* QuantLinearGPTQ   quant_linear_gptq.py:92-117 */
  QLLM_LAYOUT_AWQ_GEMM = 1, /* WQLinear_GEMM     quant_linear_awq.py:38-68   */
  QLLM_LAYOUT_HQQ = 2,      /* QuantLinearHQQ    quant_linear_hqq.py:47-68   */
  /* the library's own strip-major layout (no reference counterpart; see "native layout" below): built once at load time
   * from any of the three layouts above with qllm_repack_native(), converted back bit-exactly with qllm_unpack_native() */
  QLLM_LAYOUT_NATIVE = 3,      /* packed integer (or no) zero points: from GPTQ / AWQ_GEMM */
  QLLM_LAYOUT_NATIVE_F16Z = 4  /* fp16 zero points: from HQQ */
} qllm_layout_t;

typedef enum qllm_dtype {
  QLLM_F16 = 0,
  QLLM_BF16 = 1,
  /* qllm_linear_forward only (ABI 4): x is fp16 -- bf16 activations the CALLER converted once, e.g. for q/k/v which share their
   * input -- and y is written as bf16(fp16(result)): the reference's own bf16 shim around its fp16 kernels
   * (quant_linear_awq.py:29-36, 144-146) with the conversion of x hoisted out of the call.  Served where the 256x128 prefill
   * kernel serves a bf16 call (the call that would otherwise convert x into the workspace); QLLM_ERR_UNSUPPORTED elsewhere. */
  QLLM_F16_IN_BF16_OUT = 2
} qllm_dtype_t;

/* elementwise bf16 -> fp16 (round to nearest even), n a multiple of 8: the conversion the QLLM_F16_IN_BF16_OUT caller hoists (ABI 4) */
int qllm_convert_bf16_to_f16(const void *src, void *dst, size_t n, void *stream);

/* One quantized linear layer's buffers: the reference module's state dict, by pointer. */
typedef struct qllm_weight {
  const void *qweight;  /* i32, layout-dependent shape (see above) */
  const void *scales;   /* f16 [ceil(K/g), N] */
  const void *qzeros;   /* i32 packed (GPTQ/AWQ), f16 [ceil(K/g), N] (HQQ), or NULL (GPTQ symmetric) */
  const int32_t *g_idx; /* i32 [K] act-order group map, or NULL for k / group_size (GPTQ only) */
  const void *bias;     /* f16 [N] or NULL */
  int32_t K;            /* in_features  */
  int32_t N;            /* out_features */
  int32_t group_size;   /* > 0 (callers map the reference's -1 to K) */
  int32_t bits;         /* 2..8 (AWQ_GEMM: 4) */
  int32_t layout;       /* qllm_layout_t */
  int32_t add_zero_bias;/* COMPATIBLE_WITH_AUTOGPTQ: stored zero + this, masked (GPTQ packed zeros only) */
} qllm_weight_t;

typedef struct qllm_device_info {
  char arch[32];              /* e.g. "gfx950" (feature suffixes stripped) */
  int32_t compute_units;      /* 256 on MI355X */
  int32_t wavefront_size;     /* 64 */
  int32_t lds_bytes_per_cu;   /* 163840 */
  int32_t clock_khz;
  int64_t hbm_bytes;
} qllm_device_info_t;

/* ---- library ------------------------------------------------------------------------------------------- */
int qllm_abi_version(void);
/* 1 for a lab build of the library (-DQLLM_LAB: the dispatchers' tuning knobs QLLM_* are re-read from the environment at every call),
 * 0 for the release build (knobs are the measured defaults unless set through qllm_set_knob; only QLLM_NUM_CU is read from the environment).  Tools that A/B through the environment
 * assert on it instead of timing the same kernel twice (ABI 5). */
int qllm_is_lab_build(void);
/* Planner thresholds (ABI 6).  The kernel-selection tree was measured on Llama-2-7B / 70B shapes (profiles/r06_shape_table.md has other
 * families); a deployment may move these thresholds without rebuilding.  Process-global, not synchronised with forward calls running on
 * other threads: set them before serving.  Settable names (values outside the range every built kernel covers are refused):
 *   QLLM_STRIP1 0|1|2, QLLM_STRIP1_MAX_M 1..4, QLLM_STRIP1_3BIT 0|1, QLLM_PANEL 0|1, QLLM_PANEL_MIN_M 17..129, QLLM_PANEL_GROUP_MIN_M 17..129, QLLM_GEMM2 0|1, QLLM_GEMM3 0|1,
 *   QLLM_GEMM2_MIN_M >= 33, QLLM_GEMM3_MIN_M >= 0 (0: the measured 384 / 768 line), QLLM_GEMM2_SPLITK 0|1, QLLM_GEMM3_TAIL 0|1, QLLM_GEMM3_BF16 0|1, QLLM_GEMM3_GROUP 0|1,
 *   QLLM_SKINNY_MAX_M 0..64, QLLM_STRIP_MIN >= 0, QLLM_BITGEMV 0|1.
 * qllm_plan_describe() reflects them (it asks the same decision functions the forward calls execute).  QLLM_ERR_INVALID for any other name. */
int qllm_set_knob(const char *name, int32_t value);
int qllm_get_knob(const char *name, int32_t *value, int32_t *is_set);
void qllm_reset_knobs(void);
/* Thread-local, never NULL; "" when the calling thread's last call succeeded. */
const char *qllm_last_error(void);
/* Fills `out` for HIP device `device`; QLLM_ERR_DEVICE if there is none or it is not gfx950. */
int qllm_device_info(int device, qllm_device_info_t *out);

/* ---- workspace ----------------------------------------------------------------------------------------- */
/* Bytes of scratch qllm_linear_forward()/qllm_linear_forward_grouped() need for `w` at M rows (split-K slabs +
 * arrival counters).  The region must be zero-filled once (qllm_workspace_init) before first use; the kernels
 * leave it clean.  One workspace may be shared by calls that are ordered on one stream. */
size_t qllm_workspace_bytes(const qllm_weight_t *w, int32_t M);
/* The same for a known activation dtype: only bf16 calls of the 256x128 prefill kernel need the fp16 staging copy of x
 * (M * K * 2 bytes) that qllm_workspace_bytes() has to assume.  (ABI 4) */
size_t qllm_workspace_bytes_act(const qllm_weight_t *w, int32_t M, int32_t act_dtype);
int qllm_workspace_init(void *workspace, size_t bytes, void *stream);

/* ---- the hot path -------------------------------------------------------------------------------------- */
/* y[M,N] = x[M,K] . dequant(w) (+ bias).  x, y in `act_dtype`; scales/bias stay f16 (bf16 activations are
 * converted on load, replacing the reference's bf16->f16 shims, ort_ops.cc:119-138, quant_linear_awq.py:29-36).
 * Dispatch: M <= 32 (<= 64 on small shapes) -> weight-streaming MFMA matvec (HBM-bound); native 4-bit layers at 17 <= M <= 128 ->
 * the panel kernel (activation tiles shared through LDS, weight fragments from registers); larger M -> LDS-tiled MFMA GEMM
 * (qllm_plan_describe names the kernel; profiles/r03_mid_m.md, r04_mid_m.md hold the measurements behind the lines).
 * Fused widths: 4 bits everywhere; 3 bits (GPTQ / HQQ row stream; fp16, symmetric or packed zero points) for M <= 64 and, with
 * K % 64 == 0, N % 128 == 0 and fp16 activations, for every larger M; every other width / shape returns QLLM_ERR_UNSUPPORTED and the
 * caller takes the reference's own two-step branch (qllm_dequant + a dense GEMM, quant_linear_gptq.py:81-85).
 * Replaces QuantLinearTorchFunction.forward + bias for all three layouts. */
int qllm_linear_forward(const qllm_weight_t *w, const void *x, void *y, int32_t M, int32_t act_dtype,
                        void *workspace, size_t workspace_bytes, void *stream);

/* n_weights layers that share the SAME input x (q/k/v, gate/up) in ONE launch: y[i] = x . dequant(w[i]).
 * `w` and `y` are HOST arrays of length n_weights (<= 8); all w[i] must agree on K, bits, layout family and
 * group_size.  Decode and mid-batch sizes: M <= 32 (<= 64 for narrow groups) on the strip kernels; native 4-bit layers (N % 64 == 0,
 * K % 64 == 0, group size 32 to 64 rows / 64 / 128) up to M = 128 in one launch of the panel kernel, split-K partials in the
 * workspace (qllm_workspace_bytes of the widest layer x n_weights covers it).  Prefill sizes (ABI 6, round 6): from 384 rows, 2..4 four-bit
 * layers of one storage kind (row-stream or strip-major; not AWQ words in place), N % 128 == 0, K % 64 == 0, at least one 256x128 tile
 * per CU over the group, run as ONE grid of the prefill kernel (bf16 natively).  When the group's tiles do not fill whole rounds of CUs the
 * last round is split over K: that needs kCounter (16 KB) + tail_tiles x split x 128 KB of workspace -- at most 16 KB + 32 MB; with less
 * the launch simply does not split.  QLLM_ERR_UNSUPPORTED otherwise: call layer by layer. */
int qllm_linear_forward_grouped(const qllm_weight_t *w, void *const *y, int32_t n_weights, const void *x,
                                int32_t M, int32_t act_dtype, void *workspace, size_t workspace_bytes,
                                void *stream);

/* Diagnostics (process-global, not thread-safe; NULL switches it off): the next native-layout decode launches (4 bits, g128: the
 * batch-1 "lds-slab" form and the batch 2..32 "dma-A" form) each take one 192-byte slot of `buf` (device memory, n_slots x 24 x
 * u64, in launch order) and record 100 MHz device timestamps of wave 0 of their first, middle and last block: [entry, loads issued
 * (dma-A: ring requested), x staged (dma-A: first stage landed), rounds done, after the block barrier, exit, -, -] x 3.  A launch
 * captured into a hipGraph keeps its slot.  tools/lab/cbench.cpp --timeline [--m 16]. */
int qllm_debug_timeline(void *buf, int32_t n_slots);

/* W[K,N] (out_transposed = 0) or W[N,K] (out_transposed = 1) in `out_dtype`, bit-identical to
 * DequantizeLinearBlockWise / DequantAndUnpack / unpack().  All bits 2..8, all layouts, optional g_idx.
 * Replaces ort_ops.dequant (ort_ops.cc:58-92). */
int qllm_dequant(const qllm_weight_t *w, void *out, int32_t out_dtype, int32_t out_transposed, void *stream);

/* ---- reference-named entry points (flat signatures mirroring the pybind functions) ---------------------- */
/* ort_ops.gemv(x, qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias) -> y[M,N]
 * (ort_ops.cc:94-98).  The reference restricts this to M <= 8 and 4 bits at the call site
 * (quant_linear_gptq.py:76-80); here any M dispatches like qllm_linear_forward. */
int qllm_ort_gemv(const void *x, const void *qweight, const void *scales, const void *qzeros,
                  const int32_t *g_idx, int32_t groupsize, int32_t bits, int32_t in_features,
                  int32_t add_zero_bias, void *y, int32_t M, int32_t N, int32_t act_dtype, void *workspace,
                  size_t workspace_bytes, void *stream);

/* ort_ops.dequant(qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias) -> W[K,N] f16
 * (ort_ops.cc:58-63). */
int qllm_ort_dequant(const void *qweight, const void *scales, const void *qzeros, const int32_t *g_idx,
                     int32_t groupsize, int32_t bits, int32_t in_features, int32_t add_zero_bias, void *out_kn,
                     int32_t N, void *stream);

/* ort_ops.Dequantize4Bits(qweight u8 [N, K/block, block/2], scales [N*K/block], qzeros, g_idx, block_size, in_features,
 * out_features) -> W[N,K] f16 (ort_ops.cc:161-197; kernels dq.cu:79-245): the ORT / MatMulNBits blob layout that
 * QuantLinearORT stores (quant_linear_onnxruntime.py:85-153).  `qzeros` is either packed u8 (two 4-bit zero points per
 * byte, ceil(K/block / 2) bytes per row; zeros_f16 = 0) or fp16 [N, K/block] (zeros_f16 = 1); `g_idx` (NULL unless the
 * layer is act-order) maps each input channel to its block.  Needs block_size % 16 == 0 and in_features % block_size == 0.
 * Numerics follow the reference's Python path: fp16((q - z) * s), with the difference rounded to fp16 first when the
 * zero points are fp16. */
int qllm_ort_dequantize4bits(const void *qweight, const void *scales, const void *qzeros, int32_t zeros_f16,
                             const int32_t *g_idx, int32_t block_size, int32_t in_features, int32_t out_features,
                             void *out_nk, void *stream);

/* awq_inference_engine.gemm_forward_cuda(x[M,K], qweight[K,N/8], scales[K/g,N], qzeros[K/g,N/8], split_k_iters)
 * -> y[M,N] (gemm_cuda.h:3-4).  `split_k_iters` is accepted for signature parity and ignored: the reduction
 * over K is carried in fp32, never as the reference's fp16 partial sums (gemm_cuda_gen.cu:1115,1160). */
int qllm_awq_gemm_forward(const void *x, const void *qweight, const void *scales, const void *qzeros,
                          int32_t split_k_iters, void *y, int32_t M, int32_t K, int32_t N, int32_t group_size,
                          int32_t act_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Diagnostics: which kernel a forward call with these descriptors (1 = qllm_linear_forward, > 1 = the grouped call) and M
 * rows would run, written as text into buf ("strip ...", "skinny ...", "gemm2 ...", "gemm ...", "unsupported ...").  Pure host
 * code -- pointers are only tested for NULL / alignment, never dereferenced -- so the dispatch table can be checked without a
 * GPU.  No reference counterpart. */
int qllm_plan_describe(const qllm_weight_t *w, int32_t n_weights, int32_t M, int32_t have_workspace, char *buf,
                       size_t buflen);

/* ---- native layout ------------------------------------------------------------------------------------------ */
/* The reference's layouts are shaped for ITS kernels: GPTQ / HQQ store [K*bits/32][N] words (a 16-column strip of a layer is
 * K/8 separate 64-byte segments), AWQ stores [K][N/8] words (a row is N/2 bytes).  A batch-1 matvec on MI355X is fastest when
 * every workgroup streams ONE contiguous region (tools/lab/memlab2.hip: one Llama-2-7B decoder layer's four launches read
 * 25.96 us in the row-stream forms, 21.45 us strip-major), so the library has a layout of its own, built once at load time:
 *     qweight i32 [N/16][K*bits/32][16]   word (s, r, i) = GPTQ word (r, 16 s + i)
 *     scales  f16 [N/16][G][16]           G = K / group_size
 *     qzeros  NATIVE:      i32 [N/16][G][2]  4 bits: nibble e of word j = stored zero point of column 16 s + 8 j + e;
 *                                            3 bits: column 16 s + i at bit 3 i of the 64-bit little-endian pair;  or NULL
 *             NATIVE_F16Z: f16 [N/16][G][16]
 *     bias f16 [N] (natural order); g_idx must be NULL (act-order layers: sort the rows by group first, qllm_gather_columns)
 * Shapes: bits 3 or 4, K % 32 == 0, N % 16 == 0 (3-bit packed zero points: N % 32 == 0), group_size % 32 == 0, K % group_size == 0.
 * A descriptor with layout = QLLM_LAYOUT_NATIVE[_F16Z] is accepted by qllm_linear_forward / _grouped (M <= 64 with group size
 * 64 / 128, and -- 4 bits -- 32; larger M: see qllm_plan_describe); qllm_dequant reads the reference layouts only (QLLM_ERR_UNSUPPORTED: convert
 * back with qllm_unpack_native first).  Pure integer permutations: repack then unpack is the identity.  Counterpart in the reference: the load-time repacks of its own kernel formats (quant_linear_awq.py:95-140). */
int qllm_native_sizes(const qllm_weight_t *src, size_t *qweight_bytes, size_t *scales_bytes, size_t *qzeros_bytes);
int qllm_repack_native(const qllm_weight_t *src, void *qweight_out, void *scales_out, void *qzeros_out, void *stream);
/* dst_layout: QLLM_LAYOUT_GPTQ / _AWQ_GEMM (from NATIVE) or QLLM_LAYOUT_HQQ (from NATIVE_F16Z); outputs are the reference's buffers */
int qllm_unpack_native(const qllm_weight_t *native, int32_t dst_layout, void *qweight_out, void *scales_out, void *qzeros_out,
                       void *stream);

/* ---- layout conversion on device (repack) ----------------------------------- */
/* Integer grid q[K,N] (i32, natural order) <-> packed qweight of `layout`/`bits`.
 * Replaces general_pack_on_row / general_unpack_on_row (+ AWQ reorder) (compress_weight.py:46-92,
 * quant_linear_awq.py:95-140). */
int qllm_unpack_qweight(const void *qweight, int32_t layout, int32_t bits, int32_t K, int32_t N, int32_t *q_kn,
                        void *stream);
int qllm_pack_qweight(const int32_t *q_kn, int32_t layout, int32_t bits, int32_t K, int32_t N, void *qweight,
                      void *stream);

/* out[m, k] = x[m, perm[k]] for a row-major [M, K] matrix of 2-byte activations (fp16 or bf16; act_dtype only names the
 * element size).  The act-order helper: the reference's kernels index scales[g_idx[k]] per weight row
 * (quant_linear_gptq.py:38-43, ort_ops gemv/dequant with g_idx); this library serves act-order layers from a copy whose rows
 * are sorted by group (perm = argsort(g_idx), built at load with qllm_unpack_qweight / qllm_pack_qweight), which needs the same
 * permutation applied to the columns of x at every forward.  perm: K int32 on the device, a permutation of 0..K-1 (entries are
 * not range-checked).  x, perm, out 16-byte aligned, out must not alias x; K % 8 == 0 and K <= 28672, else QLLM_ERR_UNSUPPORTED. */
int qllm_gather_columns(const void *x, const int32_t *perm, void *out, int32_t M, int32_t K, int32_t act_dtype, void *stream);

/* ---- tensor-parallel decode: one-shot all-reduce over peer-mapped staging buffers (ABI 4; fused form ABI 5) -------------------------------------- */
/* For decode-sized tensors ([1, 8192] fp16 = 16 KB per row-parallel layer) a ring / tree all-reduce is pure latency.  On the xGMI
 * full mesh every rank instead writes its vector into every peer's staging buffer (one hop), waits for the world's flags and sums
 * locally in rank order (bit-identical on every rank).  One process per GPU: each rank allocates ONE staging buffer
 * (qllm_comm_buffer_bytes: 2 parities x world slots + a control block; fine-grained device memory -- qllm_comm_alloc is the only
 * allocation this library ever makes, and only on request), exports it as a 64-byte HIP IPC handle, imports the peers' handles, and
 * passes the device array of the `world` buffer addresses (its own at index `rank`) to every call.  x_inout: n elements (n % 8 == 0,
 * n * 2 <= slot_bytes), summed in place; the call is a single kernel on `stream`, keeps its epoch in device memory and is
 * hipGraph-capturable.  status_dev (nullable): set to 1 by the kernel if a peer never arrived.  Every rank must make the same
 * sequence of calls.  qllm_amd/comm.py wraps it behind torch.distributed; no counterpart in the reference (no distributed code). */
size_t qllm_comm_buffer_bytes(int32_t world, size_t slot_bytes);
int qllm_comm_alloc(size_t bytes, void **ptr);
int qllm_comm_free(void *ptr);
int qllm_comm_export(void *ptr, void *handle64);
int qllm_comm_import(const void *handle64, void **ptr);
int qllm_comm_close(void *ptr);
int qllm_allreduce_oneshot(void *const *peers_dev, int32_t rank, int32_t world, void *x_inout, int32_t n, int32_t act_dtype,
                           size_t slot_bytes, int32_t *status_dev, void *stream);
/* (ABI 5) A row-parallel layer at batch 1 fused with that all-reduce: y[1, N] = sum over ranks of (x_rank . dequant(W_rank)) in ONE
 * launch per rank.  Every block of the batch-1 kernel pushes its 16 partial outputs -- rounded to the activation type exactly as
 * the unfused path's y -- into every peer's staging slot; the rank's last block publishes the flags, waits for the world's and
 * writes the rank-ordered fp32 sum: bit-identical to qllm_linear_forward + qllm_allreduce_oneshot, one kernel boundary less per
 * row-parallel layer (o_proj, down_proj: 160 per Llama-2-70B token).  Same staging buffers, epoch and calling discipline as
 * qllm_allreduce_oneshot (the two may be mixed on one stream).  Serves M == 1 on native 4-bit layers with 128-wide groups whose
 * K per rank the batch-1 kernel takes (<= 16384), N * 2 <= slot_bytes, y 16-byte aligned; anything else: QLLM_ERR_UNSUPPORTED
 * and the caller runs the two calls separately.  No counterpart in the reference (no distributed code). */
int qllm_linear_forward_allreduce(const qllm_weight_t *w, const void *x, void *y, int32_t M, int32_t act_dtype, void *const *peers_dev,
                                  int32_t rank, int32_t world, size_t slot_bytes, int32_t *status_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QLLM_MI355X_H_ */
