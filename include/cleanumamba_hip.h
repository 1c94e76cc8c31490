/*
 * cleanumamba_hip.h -- C ABI of libcleanumamba_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the CleanUMamba forward+backward hot path.  Each entry
 * point replaces one native op the reference reaches through third-party wheels
 * (mamba-ssm 1.2.2, causal-conv1d 1.1.0; pinned at /root/reference
 * environment.yml:29-30) or through cuDNN/cuBLAS via torch.nn:
 *
 *   cum_selective_scan_fwd/bwd    <- selective_scan_cuda.fwd/bwd, reached from
 *                                    Mamba.forward via create_block
 *                                    (src/network/CleanUMamba.py:172-189, 289-290)
 *   cum_causal_conv1d_fwd/bwd     <- causal_conv1d_cuda.causal_conv1d_fwd/bwd
 *                                    (call pattern src/network/S4/MambaS4.py:454-463)
 *   cum_causal_conv1d_update      <- causal_conv1d_cuda.causal_conv1d_update   (Mamba.step,
 *   cum_selective_state_update    <- selective_state_update                     CleanUMamba.py:451-454)
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *     caller (PyTorch caching allocator).  Kernels never allocate or free.
 *   - strides are in ELEMENTS.  "len" is the time axis, "dim" the channel axis.
 *     The kernels are tuned for channel-contiguous (stride_dim == 1) tensors and
 *     remain correct for any stride.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     the call returns immediately (no synchronisation, graph-capturable).
 *   - return value: CUM_OK (0) or a negative CUM_E* code; nothing throws.
 *     cum_last_error() returns a static description of the last failure on the
 *     calling thread.
 *   - re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef CLEANUMAMBA_HIP_H
#define CLEANUMAMBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUM_OK 0
#define CUM_EINVAL (-1)      /* bad argument / unsupported shape */
#define CUM_ELAUNCH (-2)     /* hipLaunch failed */
#define CUM_EWORKSPACE (-3)  /* workspace too small */

#define CUM_ABI_VERSION 15

int cum_abi_version(void);
const char *cum_last_error(void);

/* Time steps between saved scan states (fixed by the kernels). */
int cum_scan_chunk(void);

/* ------------------------------------------------------------ selective scan
 * Logical shapes: u, delta, z, out: (batch, dim, len); A: (dim, dstate) row-major
 * contiguous fp32; Bm, Cm: (batch, dstate, len); D, delta_bias: (dim) or NULL;
 * z may be NULL (no gate).  x_{t} = exp(delta_t A) x_{t-1} + delta_t B_t u_t,
 * y_t = <C_t, x_t> + D u_t, out = y * silu(z).   
 *
 * ckpt: NULL (inference) or fp32 buffer of cum_scan_ckpt_elems() elements that
 * receives the state entering every half of every chunk of cum_scan_chunk() steps
 * (opaque layout, 16-byte aligned); the backward consumes it.  last_state: NULL or (batch, dim, dstate) contiguous.
 */
#define CUM_SCAN_SOFTPLUS 1
#define CUM_SCAN_A_IS_LOG 2
typedef struct {
  int32_t batch, dim, dstate, len;
  int64_t u_sb, u_sd, u_sl;          /* u strides: batch, dim, len */
  int64_t dt_sb, dt_sd, dt_sl;       /* delta */
  int64_t z_sb, z_sd, z_sl;          /* z (ignored if z == NULL) */
  int64_t o_sb, o_sd, o_sl;          /* out */
  int64_t B_sb, B_sn, B_sl;          /* Bm */
  int64_t C_sb, C_sn, C_sl;          /* Cm */
  int32_t delta_softplus;            /* flag word.  bit 0: apply softplus (threshold 20) to delta + bias.
                                        bit 1 (CUM_SCAN_A_IS_LOG = 2): `A` holds A_log -- the op uses A = -exp(A_log), as
                                        upstream Mamba.forward forms it from its parameter, and the backward returns
                                        dA_log = dA * A in `dA` (no elementwise launches around the op) */
  int32_t io_dtype;                  /* CUM_F32 / CUM_BF16 / CUM_F16: element type of u, delta, z, out and of dout, du,
                                        ddelta, dz (what autocast hands over); all arithmetic and every other
                                        tensor stay fp32 */
} cum_scan_shape;

int64_t cum_scan_ckpt_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len);

int cum_selective_scan_fwd(const cum_scan_shape *s, const void *u, const void *delta,
                           const float *A, const float *Bm, const float *Cm, const float *D,
                           const void *z, const float *delta_bias, void *out,
                           float *last_state, float *ckpt, void *stream);

/* The same op with a workspace, which lets the library take its TIME-PARALLEL form where the sequential grid
 * (batch * ceil(dim/64) * ceil(dstate/8) waves) does not fill the chip -- file denoising at batch 1
 * (src/examples/denoise.py), the 442K model, the pruned checkpoints: the sequence is cut into segments, every segment is
 * walked from a zero state, the segments' (decay, end state) pairs are composed with the scan's associative operator
 * (a, b) o (a', b') = (a' a, a' b + b'), and every segment is re-walked from its true entering state (scan_seg.hip).
 * Outputs, last_state and the checkpoints have the same meaning and layout; results equal the sequential kernels' up to
 * f32 rounding of the segment decay.  workspace: f32, cum_scan_fwd_workspace_elems() elements -- 0 means the sequential
 * kernels run (pass NULL); a NULL workspace always selects them. */
int64_t cum_scan_fwd_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len);
int cum_selective_scan_fwd_ws(const cum_scan_shape *s, const void *u, const void *delta,
                              const float *A, const float *Bm, const float *Cm, const float *D,
                              const void *z, const float *delta_bias, void *out, void *y_pre,
                              float *last_state, float *ckpt, float *workspace, void *stream);
/* y_pre: NULL, or a buffer of out's element type and strides that receives y before the gate (y_t = <C_t, x_t> + D u_t),
 * which the backward then reads instead of rebuilding it (one packed fma per state pair and step saved there).  Only where
 * cum_scan_fwd_keeps_y() returns 1 for the shape (the sequential kernel for d_state > 16: the E6 / E8 bottleneck). */
int32_t cum_scan_fwd_keeps_y(int32_t batch, int32_t dim, int32_t dstate, int32_t len, int32_t with_workspace);

/* Strides (batch, dim, len) of the three per-element gradient outputs. */
typedef struct {
  int64_t du_sb, du_sd, du_sl;
  int64_t dd_sb, dd_sd, dd_sl;       /* ddelta */
  int64_t dz_sb, dz_sd, dz_sl;       /* dz (ignored if z == NULL) */
} cum_scan_grad_strides;

/* Backward.  dout strides = the o_* strides of the shape; du/ddelta/dz use `gs`.
 * dB, dC: (batch, len, dstate) CONTIGUOUS fp32 outputs.
 * dA: (dim, dstate), dD, ddelta_bias: (dim) -- fully overwritten (not accumulated).
 * workspace: fp32, cum_scan_bwd_workspace_elems() elements.
 * y_pre: NULL, or the forward's y_pre (dout's strides): with z != NULL the kernel needs y before the gate and rebuilds it
 * when it is not given. */
int64_t cum_scan_bwd_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len);

int cum_selective_scan_bwd(const cum_scan_shape *s, const cum_scan_grad_strides *gs,
                           const void *u, const void *delta,
                           const float *A, const float *Bm, const float *Cm, const float *D,
                           const void *z, const float *delta_bias, const void *dout, const void *y_pre,
                           const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                           float *dC, float *dD, void *dz, float *ddelta_bias,
                           float *workspace, void *stream);

/* Time-parallel backward for grids that leave the chip mostly idle (d_state <= 16: the 442K model, the pruned
 * checkpoints, reached when the reference trains / fine-tunes them, src/network/CleanUMamba.py:289-290): the reverse
 * recurrence dx_t = C_t dy_t + a_{t+1} dx_{t+1} is the forward's linear operator run backwards, so time splits into
 * segments of whole 16-step chunks -- pass 1 walks every segment but the first from a zero carry (its leaving carry and
 * its sum of delta'), pass 2 walks every segment with the carry composed from the later ones and computes all
 * gradients from the forward's checkpoints (csrc/scan_bwd_small.hip).  Same arguments and results as
 * cum_selective_scan_bwd (bit-reproducible; equal to it up to the f32 rounding of the segment decay);
 * workspace: cum_scan_bwd_tp_workspace_elems() elements -- 0 means "the plan keeps this shape sequential: call
 * cum_selective_scan_bwd". */
int64_t cum_scan_bwd_tp_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len);
int cum_selective_scan_bwd_tp(const cum_scan_shape *s, const cum_scan_grad_strides *gs,
                              const void *u, const void *delta,
                              const float *A, const float *Bm, const float *Cm, const float *D,
                              const void *z, const float *delta_bias, const void *dout, const void *y_pre,
                              const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                              float *dC, float *dD, void *dz, float *ddelta_bias,
                              float *workspace, void *stream);

/* One time step for `batch` concurrent streams (Mamba.step).  state (batch, dim,
 * dstate) contiguous, updated in place.  x, dt, z, out: (batch, dim) contiguous;
 * Bv, Cv: (batch, dstate) with element stride 1 and row strides B_sb / C_sb. */
int cum_selective_state_update(int32_t batch, int32_t dim, int32_t dstate, float *state,
                               const float *x, const float *dt, const float *A,
                               const float *Bv, int64_t B_sb, const float *Cv, int64_t C_sb,
                               const float *D, const float *z, const float *dt_bias,
                               int32_t dt_softplus, float *out, void *stream);

/* ------------------------------------------------------- causal depthwise conv
 * y[b,d,t] = act(bias_d + sum_k w[d,k] x[b,d,t-(W-1)+k]), zero left pad, W <= 4.
 * (SURVEY.md Appendix A.4).  weight (dim, width) contiguous, bias (dim) or NULL.
 * silu: 1 -> SiLU activation, 0 -> identity. */
typedef struct {
  int32_t batch, dim, len, width;
  int64_t x_sb, x_sd, x_sl;
  int64_t y_sb, y_sd, y_sl;
  int32_t silu;
  int32_t io_dtype;                  /* CUM_F32 / CUM_BF16 / CUM_F16: element type of x, y, dy, dx; weights stay fp32 */
} cum_conv_shape;

int cum_causal_conv1d_fwd(const cum_conv_shape *s, const void *x, const float *weight,
                          const float *bias, void *y, void *stream);

/* dy has y's strides; dx has strides (dx_sb, dx_sd, dx_sl).  dweight (dim, width),
 * dbias (dim): overwritten.  workspace: cum_conv_bwd_workspace_elems() fp32 elements. */
int64_t cum_conv_bwd_workspace_elems(int32_t batch, int32_t dim, int32_t len, int32_t width);

int cum_causal_conv1d_bwd(const cum_conv_shape *s, const void *x, const float *weight,
                          const float *bias, const void *dy, void *dx, int64_t dx_sb,
                          int64_t dx_sd, int64_t dx_sl, float *dweight, float *dbias,
                          float *workspace, void *stream);

/* Streaming step: conv_state (batch, dim, width) contiguous is shifted left by one
 * and x (batch, dim) appended; y (batch, dim) = act(bias + <state, w>). */
int cum_causal_conv1d_update(int32_t batch, int32_t dim, int32_t width, float *conv_state,
                             const float *x, const float *weight, const float *bias,
                             int32_t silu, float *y, void *stream);

/* ------------------------------------------- encoder / decoder layers as fused GEMMs
 * Replaces cuDNN/cuBLAS behind nn.Conv1d / nn.ConvTranspose1d (+ ReLU / GLU / skip add)
 * of the encoder and decoder layers (src/network/CleanUMamba.py:108-113, 121-130,
 * 313-316; GLU src/network/layers.py:26-33).
 *
 *   out[m][n] = epilogue( sum_k A[m*lda + k] * W[n*ldw + k] + bias[n] )   (+ res[m][n])
 *
 * A rows may overlap (lda < K): with channels-last activations [B, T+2, C] a Conv1d
 * (k=4, s=2) output row reads 4*C contiguous elements at row stride 2*C, and a
 * ConvTranspose1d (k=4, s=2) output row pair reads 2*C contiguous elements at row
 * stride C, so both are this GEMM with re-packed weights (DESIGN.md "conv stack").
 * W is [N][ldw] with the K axis contiguous and zero-padded to K; bias is f32[N].
 * epilogue: 0 bias, 1 bias+ReLU, 2 bias+GLU (weights packed 16 a-rows then 16 b-rows per
 * 32 rows; the output has N/2 columns).  Rows with (m % pitch) >= valid are written as
 * zeros.  aux (optional): GLU -> the pre-activation [M][ldz] (N columns);
 * otherwise -> the activation before the residual add.  res is added after the activation.
 * Two backward-data epilogues fold the elementwise backward of the NEXT op into the GEMM
 * that produces its input gradient (autograd of ReLU / GLU in the same reference lines):
 *   3 (MASK): res = the ReLU output Y of the layer below; out = (Y > 0) ? acc + bias : 0 and,
 *     if aux != NULL, aux = acc + bias ungated (the gradient that also feeds a skip path);
 *   4 (GLU_BWD): d = acc (+ res, e.g. the gradient arriving over a skip path) is the gradient
 *     of a GLU output; aux = its saved pre-activation Z (INPUT, [M][ldz], 16 a | 16 b per 32
 *     columns); out = dZ in Z's layout ([M][ldc]): for the 16-column tile t of row m,
 *     out[32t + j] = d_j * sig(b_j), out[32t + 16 + j] = d_j * a_j * sig(b_j) * (1 - sig(b_j)).
 *     n_store counts columns of d; zero_head / zero_tail must be 0.
 * dtype: element type of A, W, res, out, aux; accumulation is always f32. */
#define CUM_F32 0
#define CUM_BF16 1
#define CUM_F16 2   /* IEEE half: what torch.autocast("cuda") hands over by default (the reference's training mode,
                       configs/config.json:14, src/training/train.py:158-160, 278-280) */

typedef struct {
  int32_t dtype, epilogue;
  int32_t M, N, K;           /* N multiple of 16 (32 for GLU); K multiple of 64 (bf16, f16) / 32 (f32) */
  int64_t lda, ldw, ldc, ldr, ldz;
  int32_t pitch, valid;
  int32_t n_store;           /* output columns written (multiple of 4) */
  int64_t zero_head;         /* elements in front of out[0] to clear (the buffer's leading zero row) */
  int64_t zero_tail;         /* elements behind out[M*ldc] to clear (the buffer's slack rows); both also */
                             /* applied to aux when it has out's geometry (epilogue != GLU)             */
  int32_t gate_only;         /* GLU (2): aux receives only the gate pre-activation b, [M][ldz] in the output's   */
                             /* column order (half the bytes of the (a | b) form).  GLU_BWD (4): aux is that b   */
                             /* and aux2 the GLU output y = a*sig(b) saved by the forward ([M][ldy]):            */
                             /* da = d*sig(b), db = d*y*(1 - sig(b)) -- a itself is never needed                 */
  int64_t ldy;
  int32_t mask_bits;         /* RELU (1): aux receives only the SIGN (activation > 0) of each element, four       */
                             /* consecutive channels per byte (low nibble; byte index (m*ldz + n) / 4).  MASK (3): */
                             /* res is such an array (byte index (m*ldr + n) / 4).  1/8 of the bytes of a bf16 mask */
  int32_t allow_split_k;     /* 1: launches with few tiles (M = streams x a handful of rows) may use the small-M      */
                             /* kernel, which splits the K axis over the four waves of a 64x64 tile: same result up  */
                             /* to summation order.  The host sets it for inference (streaming hops, no-grad forward);  */
                             /* training keeps one summation order for every shape.                                 */
} cum_gemm_desc;

int cum_gemm_nt(const cum_gemm_desc *d, const void *A, const void *W, const float *bias,
                const void *res, void *out, void *aux, const void *aux2, void *stream);
/* Which kernel cum_gemm_nt runs for this problem (only dtype, M, N, K, allow_split_k are read): 64 = 64 x 64 tiles with
 * K split over the waves, 128 = 128 x 128, 256 = 256 x 128 (f32), 384 = 128 x 256 through a three-stage LDS ring (16-bit,
 * launches with 40 ... 256 such tiles and K >= 512), 512 = 256 x 256 with two wave groups in ping-pong (16-bit).  Lets a
 * parity test state which kernel a shape was verified on. */
int cum_gemm_nt_tile(const cum_gemm_desc *d);

/* GLU backward on the packed pre-activation Z [M][ldz] (n_groups x (16 a | 16 b));
 * dOut [M][ldo] has 16 channels per group (n_out valid columns); dZ has Z's layout. */
int cum_glu_bwd(int32_t dtype, int64_t M, int32_t n_groups, int32_t n_out, const void *Z,
                int64_t ldz, const void *dOut, int64_t ldo, void *dZ, void *stream);
/* Same from the gate-only form: Bg [M][ldb] (gate pre-activations, 16 per group) and Y [M][ldy] (the GLU
 * output); dZ [M][ldz] in the packed (16 a | 16 b) layout. */
int cum_glu_bwd_gate(int32_t dtype, int64_t M, int32_t n_groups, int32_t n_out, const void *Bg,
                     int64_t ldb, const void *Y, int64_t ldy, const void *dOut, int64_t ldo, void *dZ,
                     int64_t ldz, void *stream);

/* dZ = dOut * (Y > 0) over M rows x n_cols columns (n_cols multiple of 4).  zero_head / zero_tail:
 * elements in front of dZ[0] / behind dZ[M*ldz] to clear (leading zero row / slack rows). */
int cum_relu_bwd(int32_t dtype, int64_t M, int32_t n_cols, const void *Y, int64_t ldy,
                 const void *dOut, int64_t ldo, void *dZ, int64_t ldz, int64_t zero_head,
                 int64_t zero_tail, void *stream);

/* out[c] = sum_m X[m*ld + c] in f32 (bias gradients); deterministic two-stage reduction.
 * workspace: cum_colsum_workspace_elems() f32 elements. */
int64_t cum_colsum_workspace_elems(int64_t M, int32_t n_cols);
int cum_colsum(int32_t dtype, int64_t M, int32_t n_cols, const void *X, int64_t ld, float *out,
               float *workspace, void *stream);

/* Weight gradient of any of the layers above, with the bias gradient fused:
 *   dW[n*ldw + k] = sum_m dZ[m*ldz + n] * X[m*ldx + k]   (n < N, k < K),   db[n] = sum_m dZ[m*ldz + n]
 * X rows may overlap (ldx < K) as in cum_gemm_nt.  dW, db are f32 and fully overwritten; db may
 * be NULL.  The m axis is split over workgroups and combined deterministically.
 * workspace: cum_gemm_tn_workspace_elems() f32 elements. */
int64_t cum_gemm_tn_workspace_elems(int32_t dtype, int64_t M, int32_t N, int32_t K);
int cum_gemm_tn(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz,
                const void *X, int64_t ldx, float *dW, int64_t ldw, float *db, float *workspace,
                void *stream);
/* cum_gemm_tn_scatter: the same product, its result written where the PARAMETERS live instead of as dW / db in GEMM
 * layout (the reference's autograd leaves every weight gradient in its parameter's own layout,
 * src/training/train.py:282-285; the GEMM layout is this library's, so un-doing it is the library's job too: one launch
 * less and no f32 round trip per weight gradient).  A destination is a contiguous row-major f32 matrix rows x cols (cols a
 * multiple of 4, 16-byte aligned) -- a parameter's gradient seen as a matrix -- whose element (r, c) is the element
 * rowoff[r] + coloff[c] of the virtual result: positions n K + k of dW [N][K], positions N K + n the bias gradient
 * (reads_bias != 0 if any position of the job is one; bias_fold: below).  rowoff / coloff: int32 device arrays.  One or
 * two destinations (a layer's weight and bias).  workspace as cum_gemm_tn.  Deterministic. */
typedef struct {
  float *dst;
  const int32_t *rowoff, *coloff;
  int32_t rows, cols;
  int32_t reads_bias;
  int32_t bias_fold;         /* > 0: a bias position N K + n delivers db[n] + db[n + bias_fold] (the transposed conv's bias
                                gradient: the GEMM sees output rows in pairs, N = 2 Cp, and each channel twice) */
} cum_tn_scatter;
int cum_gemm_tn_scatter(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz, const void *X,
                        int64_t ldx, const cum_tn_scatter *jobs, int32_t njobs, float *workspace, void *stream);

/* Kernel cum_gemm_tn runs for this problem: 256 (ping-pong kernel, 256 x 256 tiles), 384 (streaming kernel: the whole
 * 128 x 256 / 256 x 128 result per workgroup) or 128 (128 x 128 tiles). */
int cum_gemm_tn_tile(int32_t dtype, int64_t M, int32_t N, int32_t K);

/* ---- multi-resolution STFT loss around rocFFT (src/util/stft_loss.py:16-184) -------------------------------------
 * One resolution = (n_fft, hop, win_length, window[win_length]); torch.stft(center=True, reflect) framing:
 * n_frames = 1 + len / hop, bins = n_fft / 2 + 1.  The caller runs the batched r2c / c2r (rocFFT) in between.
 *
 * cum_stft_frames: frames[b][f][n] = window[n - (n_fft - win_length)/2] * x_reflect[b][f*hop + n]  (:29-33)
 * cum_stft_loss_fwd: spec_x / spec_y are (batch, n_frames, bins) interleaved complex f32.  With
 *   X = sqrt(max(re^2 + im^2, 1e-7)) (:38) over frames >= frame0 (band == "high" keeps frames >= n_frames / 2, :117-119):
 *   stats = { |Y - X|_F / |Y|_F (:59),  mean |log Y - log X| (:80),  |Y - X|_F,  |Y|_F }.
 * cum_stft_loss_bwd: zspec = d(g_sc * stats[0] + g_mag * stats[1]) / d spec_x with interior bins halved, so that an
 *   unnormalised c2r (irfft(norm="forward")) of zspec is the gradient wrt the real frames.  g_sc, g_mag: device scalars.
 * cum_stft_fold: dx[b][m] (+)= sum of window * dframes over every frame position that reads sample m (incl. reflections). */
int cum_stft_frames(const float *x, int64_t batch, int64_t len, int64_t x_stride_b, int32_t n_fft, int32_t hop,
                    int32_t win_length, const float *window, float *frames, int64_t n_frames, void *stream);
int64_t cum_stft_loss_workspace_elems(int64_t batch, int64_t n_frames);
int cum_stft_loss_fwd(const float *spec_x, const float *spec_y, int64_t batch, int64_t n_frames, int32_t bins,
                      int64_t frame0, float *workspace, float *stats, void *stream);
int cum_stft_loss_bwd(const float *spec_x, const float *spec_y, int64_t batch, int64_t n_frames, int32_t bins,
                      int64_t frame0, const float *stats, const float *g_sc, const float *g_mag, float *zspec,
                      void *stream);
int cum_stft_fold(const float *dframes, int64_t batch, int64_t len, int32_t n_fft, int32_t hop, int32_t win_length,
                  const float *window, int64_t n_frames, float *dx, int64_t dx_stride_b, int32_t accumulate,
                  void *stream);

/* ---- operand packing: dst[i] = idx[i] < 0 ? 0 : (dst_dtype) src[idx[i]], i < n.  The GEMM operand layouts above are
 * index permutations (with zero padding) of the reference's parameter tensors (encoder/decoder state-dict keys,
 * src/network/CleanUMamba.py:108-130); the same call unpacks weight gradients into parameter layout. */
int cum_gather(int32_t src_dtype, const void *src, const int32_t *idx, int64_t n, int32_t dst_dtype, void *dst,
               void *stream);

/* cum_pack2d: the same re-pack without a per-element index, for layouts that separate into dst[r][c] =
 * src[rowoff[r] + coloff[c]] (all weight layouts of the conv stack and the projections: network/convstack.py PackPlan).
 * jobs: device array of { int64 dst_off; int32 rows, cols (multiple of 8), row_tab, col_tab, transpose, runs8 } -- dst_off
 * in elements (multiple of 8) from `dst`, row_tab / col_tab positions in `tables` (int32, INT32_MIN = zero padding),
 * transpose = 1 when the source is fast along destination rows (the 64 x 64 tile then goes through LDS); runs8 (only
 * read when transpose = 0): 0 = no promise; 1 = every aligned group of 8 destination columns is all padding or 8
 * consecutive source elements (one table entry and one run per group instead of eight gathers); 2 = and every run starts
 * at a multiple of 4 elements from `src` (16-byte loads).
 * tiles: device array of n_tiles x { job, tile row, tile column }.  src: f32, 16-byte aligned. */
int cum_pack2d(const float *src, const void *jobs, const int32_t *tiles, int32_t n_tiles, const int32_t *tables,
               int32_t dst_dtype, void *dst, void *stream);

/* Batched 1-D FFTs for the STFT loss, on hipFFT / rocFFT (torch.stft's transform, src/util/stft_loss.py:29-33, and its
 * autograd).  Unnormalised in both directions.  The library keeps NO plan cache and allocates NO device memory for a
 * transform: a plan is an object the caller creates once per (kind, n, batch), keeps, and destroys; its work area is
 * caller memory passed to every cum_fft_exec (size returned by cum_fft_plan_create; may be 0).
 *   kind 0: real -> complex, in [batch][n] real, out [batch][n/2+1] interleaved complex;
 *   kind 1: complex -> real, in [batch][n/2+1] complex, out [batch][n] real;
 *   kind 2: complex -> complex, [batch][n] interleaved complex, `inverse` selects the direction, in == out allowed.
 * The INPUT of the real transforms may be overwritten (rocFFT does that for some lengths): pass scratch.
 * A plan is not re-entrant (one stream / one thread at a time); cum_fft_exec is graph-capturable. */
int cum_fft_plan_create(int32_t kind, int32_t n, int64_t batch, void **plan, int64_t *work_bytes);
int cum_fft_plan_destroy(void *plan);
int cum_fft_exec(void *plan, float *in, float *out, int32_t inverse, void *work, void *stream);

/* The same loss on PACKED transforms: a frame's n_fft real samples are read as n_fft/2 complex numbers
 * (even samples real, odd imaginary) and transformed by a kind-2 plan (cum_fft_exec); zx / zy: [batch * n_frames][n_fft / 2] complex.  The
 * real-input spectrum X[k], k = 0..n_fft/2, is recovered inside the kernels (twiddle: [n_fft/2 + 1] complex,
 * e^{-2 pi i k / n_fft}).  cum_stft_loss_bwd_packed writes gz = dL/dRe(Z) + i dL/dIm(Z); an unnormalised inverse
 * complex transform of gz is the gradient wrt the frames (what cum_stft_fold takes).  Replaces rocFFT's r2c post- / c2r
 * pre-processing passes. */
int cum_stft_loss_fwd_packed(const float *zx, const float *zy, int64_t batch, int64_t n_frames, int32_t n_fft,
                             int64_t frame0, const float *twiddle, float *workspace, float *stats, void *stream);
int cum_stft_loss_bwd_packed(const float *zx, const float *zy, int64_t batch, int64_t n_frames, int32_t n_fft,
                             int64_t frame0, const float *stats, const float *g_sc, const float *g_mag,
                             const float *twiddle, float *gz, void *stream);

/* The same loss with framing, both transforms and the loss terms in ONE kernel per direction (n_fft 512 / 1024 / 2048:
 * cum_stft_fused_supported): a wave reads a frame's samples of x and y from the waveforms, transforms both in LDS
 * (n_fft / 2 packed complex points, in-place radix-4, spectrum consumed in digit-reversed order) and accumulates the
 * loss partial sums -- no frame and no spectrum reaches HBM.  The backward rebuilds the two transforms, forms the gradient
 * spectrum in place, inverts it in LDS and writes the frame gradient over the window's support only, which is what
 * cum_stft_fold reads (dframes [batch][n_frames][n_fft], elements outside the support are left untouched).
 * x, y: [batch][len] f32 with row strides x_stride_b / y_stride_b (any; 8-byte aligned frames load in pairs); window [win_length];
 * twiddle [n_fft/2 + 1] complex e^{-2 pi i k / n_fft}; frame0 / stats / g_sc / g_mag as above.
 * workspace: cum_stft_fused_workspace_elems() f32.  Same values as the rocFFT path up to f32 rounding of the transform. */
int cum_stft_fused_supported(int32_t n_fft);
int64_t cum_stft_fused_workspace_elems(int64_t batch, int64_t n_frames);
int cum_stft_fused_fwd(const float *x, const float *y, int64_t batch, int64_t len, int64_t x_stride_b,
                       int64_t y_stride_b, int32_t n_fft, int32_t hop, int32_t win_length, const float *window,
                       const float *twiddle, int64_t n_frames, int64_t frame0, float *workspace, float *stats,
                       void *stream);
int cum_stft_fused_bwd(const float *x, const float *y, int64_t batch, int64_t len, int64_t x_stride_b,
                       int64_t y_stride_b, int32_t n_fft, int32_t hop, int32_t win_length, const float *window,
                       const float *twiddle, int64_t n_frames, int64_t frame0, const float *stats,
                       const float *g_sc, const float *g_mag, float *dframes, void *stream);

/* ---- residual add + LayerNorm of the Mamba blocks (mamba-ssm Block.forward with fused_add_norm=False as the reference
 * runs it, src/network/CleanUMamba.py:156-189, 288-294, and the final add + norm_f, :292-294):
 *   residual_out = x + residual (fp32);   y = (residual_out - mean) * rstd * weight + bias
 * x: (batch, len, dim) of x_dtype with element strides (x_sb, x_sl, 1); residual: contiguous fp32 or NULL;
 * y: contiguous, y_dtype; mean, rstd: fp32 [batch*len] (saved for the backward).  1 <= dim <= 2048: 16-byte vector
 * kernels when dim and the hidden strides are multiples of 8, element-access kernels otherwise (pruned checkpoints'
 * d_model 55, 114, 477 ...).
 * Backward: dy (y_dtype) and dres_out (fp32 or NULL, the gradient arriving at residual_out) ->
 *   dx = d(x) = d(residual) written as fp32 (dx32, may be NULL) and/or in h_dtype (dxh, may be NULL);
 *   dweight, dbias (dbias may be NULL): fully overwritten.  workspace: cum_add_layernorm_bwd_workspace_elems(dim) fp32. */
int cum_add_layernorm_fwd(int32_t x_dtype, int32_t y_dtype, int64_t batch, int32_t len, int32_t dim, const void *x,
                          int64_t x_sb, int64_t x_sl, const float *residual, const float *weight, const float *bias,
                          float eps, float *residual_out, void *y, float *mean, float *rstd, void *stream);
int64_t cum_add_layernorm_bwd_workspace_elems(int32_t dim);
int cum_add_layernorm_bwd(int32_t y_dtype, int32_t h_dtype, int64_t rows, int32_t dim, const void *dy,
                          const float *dres_out, const float *residual_out, const float *mean, const float *rstd,
                          const float *weight, float *dx32, void *dxh, float *dweight, float *dbias, float *workspace,
                          void *stream);

/* ---- one Mamba block for one token of every stream in ONE launch (Block.forward + Mamba.step of the streaming path,
 * src/network/CleanUMamba.py:451-454): residual_out = hidden_in (+ residual_in); h = LayerNorm(residual_out);
 * xz = in_proj h; x = silu(conv_update(x)); (dt, B, C) = x_proj x; dt = softplus(dt_proj dt + dt_bias);
 * state = exp(dt A) state + dt B x; y = (C . state + D x) silu(z); hidden_out = out_proj y.  All tensors f32 and
 * contiguous: hidden / residual [streams][d_model], conv_state [streams][d_inner][d_conv] and ssm_state
 * [streams][d_inner][d_state] updated in place, A = -exp(A_log) [d_inner][d_state].  Optional (NULL): residual_in,
 * norm_b, in_proj_b, conv_b, dt_proj_b, D, out_proj_b.  For small (pruned) models: every stream's workgroup re-reads
 * the projection matrices from L2 -- cum_mamba_step_supported() tells whether the sizes qualify. */
int cum_mamba_step_supported(int32_t d_model, int32_t d_inner, int32_t d_state, int32_t dt_rank, int32_t d_conv);
int cum_mamba_step(int32_t streams, int32_t d_model, int32_t d_inner, int32_t d_state, int32_t dt_rank, int32_t d_conv,
                   float eps, const float *hidden_in, const float *residual_in, const float *norm_w,
                   const float *norm_b, const float *in_proj_w, const float *in_proj_b, float *conv_state,
                   const float *conv_w, const float *conv_b, const float *x_proj_w, const float *dt_proj_w,
                   const float *dt_proj_b, const float *A, const float *D, float *ssm_state, const float *out_proj_w,
                   const float *out_proj_b, float *hidden_out, float *residual_out, void *stream);

/* out[s][j] = bias[j] + sum_k W[j*cols + k] * x[s*x_stride + k], f32, one workgroup per row s: the 1x1 bottleneck
 * convolutions of a streaming hop (tsfm_conv1 / tsfm_conv2 on one-column inputs, src/network/CleanUMamba.py:277,310).
 * cols <= 1024; bias may be NULL. */
int cum_small_linear(int32_t streams, int32_t rows, int32_t cols, const float *x, int64_t x_stride, const float *W,
                     const float *bias, float *out, int64_t out_stride, void *stream);

/* ---- streaming decoder glue (CleanUMamba._denoise_frame, src/network/CleanUMamba.py:476-488): overlap-add of a frame's
 * transposed-conv output with the previous frame's tail, activation, skip add and tail update for S streams in
 * lock-step, channels-last rows of Cp (C real channels):
 *   out[s][t]  = act(y[s][t] + (t < 2 ? tail[s][t] : 0)) + skip[s][t]      t < L2
 *   tail[s][t] = y[s][L2 + t] - bias                                          t < 2
 * y, skip, out: stream s starts y_pitch / skip_pitch / out_pitch rows after stream s-1; tail: [S][2][Cp]. */
/* Streaming encoder window of one layer (per-layer caches of _denoise_frame, :425-447): for every stream drop the n_new
 * oldest of `rows` rows and append n_new rows of `fresh`: window row t >= rows - n_new takes fresh row t - fresh_row0
 * (fresh_row0 = 0: `fresh` is a whole recomputed window; fresh_row0 = rows - n_new: `fresh` holds only the new rows).
 * window / fresh: stream s starts `pitch` / `fresh_pitch` rows after stream s-1 (fresh must not alias window).  One
 * in-place launch when rows - n_new <= 2048 and rows are 16-byte aligned; other windows go through tmp (streams * rows * Cp elements, else
 * it may be NULL).  tail_dst (optional, in-place path only): also receives the n_new + 2 newest rows of the updated
 * window, stream s at row s * tail_pitch -- what cum_stream_tail_rows would copy for the next layer. */
int cum_stream_window_update(int32_t dtype, int32_t streams, int32_t rows, int32_t n_new, int32_t Cp, void *window,
                             const void *fresh, int64_t pitch, int64_t fresh_pitch, int32_t fresh_row0, void *tmp,
                             void *tail_dst, int64_t tail_pitch, void *stream);
/* Input of the next layer's incremental step: dst[s][t] = src[s][src_row0 + t] for t < rows (the newest rows of a
 * window: 2 carried rows + the hop's new rows); rows of dst beyond `rows` are not touched (they stay zero). */
int cum_stream_tail_rows(int32_t dtype, int32_t streams, int32_t rows, int32_t Cp, const void *src, int64_t src_pitch,
                         int32_t src_row0, void *dst, int64_t dst_pitch, void *stream);
int cum_stream_overlap_add(int32_t dtype, int32_t streams, int32_t L2, int32_t Cp, int32_t C, const void *y,
                           int64_t y_pitch, void *tail, const float *bias, const void *skip, int64_t skip_pitch,
                           void *out, int64_t out_pitch, int32_t relu, void *stream);

/* ---- the whole streaming hop in ONE launch (csrc/hop.hip): CleanUMamba.feed / _denoise_frame of
 * src/network/CleanUMamba.py:370-490 (running input std :399-401, encoder caches :425-447, Mamba.step :451-454, decoder
 * overlap-add :476-488, with the skip order fixed as SURVEY fact 9 describes) for `n_hops` consecutive hops of `streams`
 * concurrent streams; a workgroup owns one stream, activations stay in LDS, f32 throughout (exact-f32 matrix cores).
 *   plan     device int32[cum_stream_hop_plan_ints()]: sizes, LDS regions, offsets into `weights` and into a stream's
 *            state block, in the field order of csrc/hop.hip::HopPlan (built by cleanumamba_amd/network/hopplan.py):
 *            a 16-int header, the op list (24 ints per op), per op and wave the position and length of the wave's stage
 *            list, and the stage lists of the matrix products (4 ints per stage: blob offset, LDS operand offset, chunk
 *            count | first | last flags, destination) -- the kernel walks what the host compiled, it derives nothing
 *   weights  device f32 blob: every matrix zero-padded and in MFMA fragment order ([tile][16-deep k chunk][lane][4]),
 *            biases / LayerNorm parameters / -exp(A_log) / D padded to the pitches the plan names
 *   state    device f32 [streams][state_stride]: encoder rings, decoder tails, conv / SSM states, running std, ring phase
 *   in       stream s, hop h reads the frame in[s * in_stride + h * hop_len ...+ frame_len) (raw samples)
 *   out      stream s, hop h writes out[s * out_stride + h * hop_len ...+ hop_len)
 *   lds_bytes dynamic LDS the plan needs (<= cum_stream_hop_max_lds_bytes(), multiple of 16)
 * The first frame of a stream (whole windows, no history) is the per-layer path's; its state is converted once. */
int cum_stream_hop_plan_ints(void);
int cum_stream_hop_max_lds_bytes(void);
int cum_stream_hop(const void *plan, const float *weights, float *state, int64_t state_stride, int32_t streams,
                   const float *in, int64_t in_stride, float *out, int64_t out_stride, int32_t n_hops,
                   int32_t lds_bytes, void *stream);

/* ---- first encoder layer, fused (csrc/enc0.hip): Conv1d(1 -> 64, k 4, s 2) + ReLU + Conv1d(64 -> 128, 1x1) + GLU of
 * src/network/CleanUMamba.py:108-113 at channels_input = 1, channels_H = 64 (E6 / E8), 16-bit element types.  The ReLU
 * output of the one-input-channel conv is rebuilt from the four input samples wherever it is needed instead of being
 * stored: forward and backward each move the layer's output-sized tensors once.
 *   xin      the layer's input row buffer [1 + 2 M + >= 2][8] (column 0 = sample, as cum_frame_rows writes it)
 *   w1, b1   the conv's parameters as stored: (64, 1, 4) and (64) f32
 *   w2p, b2p the 1x1 conv in the forward GEMM's packed operand order: [128][64] element type / [128] f32, per 32 rows
 *            16 a-rows then the 16 b-rows of the same channels
 *   M rows = clips x pitch; rows with (m mod pitch) >= valid are the zero rows between clips
 * cum_enc0_fwd: out = row buffer [1 + M + slack][64] (row 0 and zero_tail elements behind row M are cleared), gate
 *   [M][64] = the GLU's gate pre-activations for the backward (NULL: not kept).
 * cum_enc0_bwd: dZ [M][128] (gradient wrt the 1x1 conv's output, packed column order) -> slot_w2 = dW2 [128][64] then
 *   db2 [128], slot_w1 = dW1 [64][32] (element (h, 8 k) = tap k, other columns 0) then db1 [64]: the layouts
 *   cum_gemm_tn writes for these two layers.  workspace: cum_enc0_bwd_workspace_elems(M) f32.  Deterministic (slabs
 *   per workgroup, fixed-order sum).  The layer has no input gradient. */
int cum_enc0_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *xin, const float *w1,
                 const float *b1, const void *w2p, const float *b2p, void *out, int64_t zero_tail, void *gate,
                 void *stream);
int32_t cum_enc0_bwd_workgroups(int64_t M);
int64_t cum_enc0_bwd_workspace_elems(int64_t M);
int cum_enc0_bwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *dZ, const void *xin,
                 const float *w1, const float *b1, const void *w2p, float *slot_w2, float *slot_w1, float *workspace,
                 void *stream);

/* An encoder layer of width 128, forward, fused:  Conv1d(64 -> 128, k 4, s 2) + ReLU + Conv1d(128 -> 256, 1x1) + GLU of
 * src/network/CleanUMamba.py:108-113 at channels_H = 64 (the second encoder layer of E6 / E8), 16-bit element types, one
 * launch instead of cum_gemm_nt (EPI_RELU) + cum_gemm_nt (EPI_GLU); it writes what those two write, so the backward is
 * theirs (csrc/ench.hip).
 *   xin      row 1 of the input row buffer [..][64]: output row m reads input rows 2 m .. 2 m + 3; x_rows = rows readable
 *            from xin (>= 2 M + 2)
 *   w1p/b1p  conv weight packed [128][256] (column tap * 64 + channel: lay_conv_fwd) and bias [128] f32
 *   w2p/b2p  1x1 weight packed [256][128] (rows 32 g + i: i < 16 value channel 16 g + i, else gate channel: lay_glu_fwd),
 *            bias [256] f32 in the same row order
 *   y1       row 1 of the hidden row buffer [..][128] (ReLU output; the 128 elements before it and y1_tail elements behind
 *            row M are cleared) or NULL (inference: not stored); bits: its sign nibbles, one byte per four channels,
 *            32 bytes per row (NULL: not kept)
 *   out      row 1 of the output row buffer [..][128] (framing cleared as for y1); gate: [M][128] gate pre-activations for
 *            the backward or NULL
 *   M rows = clips x pitch; rows with (m mod pitch) >= valid are zero rows of y1 / out. */
int cum_ench_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *xin, int64_t x_rows,
                 const void *w1p, const float *b1p, const void *w2p, const float *b2p, void *y1, int64_t y1_tail,
                 void *bits, void *out, int64_t out_tail, void *gate, void *stream);

/* A decoder layer of width 128, forward, fused:  Conv1d(128 -> 256, 1x1) + GLU + ConvTranspose1d(128 -> 64, k 4, s 2) +
 * ReLU (+ the next layer's skip connection) of src/network/CleanUMamba.py:121-130, 313-316 at channels_H = 64 (the
 * second-to-last decoder layer of E6 / E8), 16-bit element types, one launch instead of cum_gemm_nt (EPI_GLU) +
 * cum_gemm_nt (EPI_RELU with residual and sign nibbles); it writes what those two write (csrc/dech.hip).
 *   u        row 0 (the leading zero row) of the input row buffer [..][128]; u_rows = rows readable from it
 *   w1p/b1p  1x1 weight packed [256][128] (lay_glu_fwd) and bias [256] f32 in the same row order
 *   wtp/btp  transposed-conv weight packed [128 = (output-row parity, 64 channels)][256 = (row m - 1 | row m, 128
 *            channels)] (lay_convt_fwd) and bias [128] f32 (the 64 biases twice)
 *   skip     row 1 of the skip row buffer [..][64] (added after the ReLU) or NULL
 *   g        row 0 of the GLU-output row buffer [..][128] (row 0 and g_tail elements behind row M are cleared) with its gate
 *            pre-activations gate [M][128], or both NULL (inference: not stored)
 *   out      row 1 of the output row buffer [..][64] (GEMM row m = output rows 2 m, 2 m + 1; the 64 elements before it and
 *            out_tail elements behind output row 2 M are cleared); bits: sign nibbles of the ReLU, 32 bytes per GEMM row
 *            (NULL: not kept)
 *   M input rows = clips x pitch; data row d is real iff (d mod pitch) < valid. */
int cum_dech_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *u, int64_t u_rows,
                 const void *w1p, const float *b1p, const void *wtp, const float *btp, const void *skip, void *g,
                 int64_t g_tail, void *gate, void *out, int64_t out_tail, void *bits, void *stream);

/* The last decoder layer, fused:  Conv1d(64 -> 128, 1x1) + GLU + ConvTranspose1d(64 -> 1, k 4, s 2) of
 * src/network/CleanUMamba.py:121-130 at channels_output = 1, channels_H = 64 (E6 / E8), 16-bit element types.  The GLU
 * output is rebuilt from the layer input wherever it is needed instead of being stored.
 *   u        the layer's input row buffer [1 + M + slack][64]; M rows = clips x pitch, rows with (m mod pitch) >= valid
 *            are the zero rows between clips; pair row p <= valid of a clip produces output rows 2 p, 2 p + 1
 *   w1p, b1p the 1x1 conv in the forward GEMM's packed operand order ([128][64] element type / [128] f32, per 32 rows
 *            16 a-rows then the 16 b-rows of the same channels); wt (64, 1, 4), bt (1): the transposed conv as stored, f32
 * cum_dec7_fwd: out = output row buffer [1 + 2 M + slack][8] (channel 0 = the signal, channels 1-7 zero; row 0 and
 *   zero_tail elements behind row 2 M are cleared).
 * cum_dec7_bwd: dY = gradient of that buffer (channel 0 is read), mask = sign bits of the ReLU of the layer below as
 *   cum_gemm_nt writes them (16 bytes per row, data row 0 first) -> dU = gradient of u, dpre = dU where the bit is set
 *   (row buffers [1 + M + slack][64]; row 0 and zero_tail elements behind row M cleared), slot_w1 = dW1 [128][64] then
 *   db1 [128], slot_wt = dW [16][128] then db [16]: the layouts cum_gemm_tn writes for the 1x1 and for the transposed
 *   conv taken as a GEMM (N = output-row parity x 8 channels, K = (row m - 1 | row m) x 64 channels).
 *   workspace: cum_dec7_bwd_workspace_elems(M) f32.  Deterministic (slabs per workgroup, added in index order). */
int cum_dec7_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *u, const void *w1p, const float *b1p,
                 const float *wt, const float *bt, void *out, int64_t zero_tail, void *stream);
int32_t cum_dec7_bwd_workgroups(int64_t M);
int64_t cum_dec7_bwd_workspace_elems(int64_t M);
int cum_dec7_bwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *dY, const void *u, const void *mask,
                 const void *w1p, const float *b1p, const float *wt, void *dU, void *dpre, int64_t zero_tail,
                 float *slot_w1, float *slot_wt, float *workspace, void *stream);

/* ---- waveform ends of the train step (csrc/loss.hip).  All sums are per-workgroup partials in fixed order + one
 * finishing workgroup: deterministic, graph-capturable, nothing returns to the host.
 *
 * cum_lp_loss_fwd / _bwd: the time-domain term of loss_fn, F.l1_loss (p = 1) / F.mse_loss (p = 2) of (denoised, clean)
 * with reduction "mean" (src/util/util.py:262-268).  y, c, dy: n contiguous f32; partials: cum_lp_loss_parts(n) f32;
 * out: 1 f32; gout: 1 f32 in device memory (the incoming gradient of the scalar).  dy = gout * sign(y - c) / n (p = 1;
 * 0 where y == c, as torch) or gout * 2 (y - c) / n (p = 2). */
int32_t cum_lp_loss_parts(int64_t n);
int cum_lp_loss_fwd(int32_t p, const float *y, const float *c, int64_t n, float *partials, float *out, void *stream);
int cum_lp_loss_bwd(int32_t p, const float *y, const float *c, int64_t n, const float *gout, float *dy, void *stream);

/* cum_clip_std: out[r] = unbiased std of row r of x (rows x len f32, row stride `stride` elements) + eps -- the per-clip
 * input normalisation `noisy_audio.std(dim=2, keepdim=True) + 1e-3` (src/network/CleanUMamba.py:260-262).  One pass
 * (Welford per thread, Chan merges in fixed order).  partials: 3 * rows * cum_clip_std_parts(len) f32. */
int32_t cum_clip_std_parts(int64_t len);
int cum_clip_std(const float *x, int32_t rows, int64_t len, int64_t stride, float eps, float *partials, float *out,
                 void *stream);

/* cum_frame_rows: (batch, len) f32 signal -> the channels-last row buffer of a ONE-channel activation (8 columns per
 * row: column 0 the sample, 1-7 zero; row 0 zero; per clip T rows + 2 zero rows; total_rows rows written in all):
 * out = x / scale[b] (invert = 1: `noisy / std` + pad_signal, src/network/CleanUMamba.py:262-264) or x * scale[b]
 * (invert = 0: the backward of cum_unframe_rows); steps len .. T-1 are zero; scale may be NULL (= 1).
 * cum_unframe_rows: y[b][t] = rows[1 + b (T + 2) + t][0] * scale[b], t < len: `x[:, :, :L] * std`
 * (src/network/CleanUMamba.py:319) read straight from the last transposed conv's row buffer. */
int cum_frame_rows(int32_t dtype, const float *x, int32_t batch, int64_t len, int64_t stride, int64_t T,
                   int64_t total_rows, const float *scale, int32_t invert, void *out, void *stream);
int cum_unframe_rows(int32_t dtype, const void *rows, int32_t batch, int64_t len, int64_t T, const float *scale,
                     float *y, void *stream);

/* ---- optimizer section of the train step on flat buffers (src/training/train.py:303-310: scaler.unscale_,
 * clip_grad_norm_(clip_grad_norm_max), scaler.step(Adam), scaler.update; Adam built at :145-148 with betas / eps /
 * weight_decay of configs/config.json, GradScaler at :158-160).  p, g, m, v: f32 buffers of n elements, 16-byte
 * aligned (all parameters of the model back to back; training/flat_optim.py).  `state`: cum_optim_state_elems() f32 in
 * device memory, zero-initialised by the caller except [3] = initial loss scale and [8] = learning rate (written by
 * the caller before every step):
 *   [0] total gradient norm (unscaled)  [1] multiplier applied to g (clip coefficient / loss scale)  [2] found inf/nan
 *   [3] loss scale  [4] growth tracker  [5] Adam step count  [6] 1 - beta1^t  [7] sqrt(1 - beta2^t)  [8] learning rate
 *   [9] skipped steps
 * Sequence per optimizer step: cum_optim_sumsq (partials: cum_optim_sumsq_parts(n) f32) -> cum_optim_prepare ->
 * cum_optim_adam.  Nothing returns to the host; with found inf/nan the Adam launch leaves p, m, v untouched and the
 * loss scale backs off (torch.amp.GradScaler semantics: growth x `growth` after `growth_interval` clean steps).
 * max_norm <= 0 disables clipping; use_scaler = 0: gradients are unscaled, state[3] is ignored. */
int32_t cum_optim_state_elems(void);
int32_t cum_optim_sumsq_parts(int64_t n);
int cum_optim_sumsq(const float *g, int64_t n, float *partials, void *stream);
int cum_optim_prepare(float *state, const float *partials, int32_t nparts, float max_norm, double beta1, double beta2,
                      int32_t use_scaler, float growth, float backoff, int32_t growth_interval, void *stream);
int cum_optim_adam(float *p, const float *g, float *m, float *v, int64_t n, const float *state, double beta1,
                   double beta2, float eps, float weight_decay, void *stream);   /* betas as double: 1 - beta must not
                                                                                    be formed in f32 (0.999f: 5e-5 off) */

#ifdef __cplusplus
}
#endif
#endif /* CLEANUMAMBA_HIP_H */
