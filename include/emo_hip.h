/*
 * emo_hip.h — C-ABI of libemo_hip.so: the MI355X (gfx950) kernels behind the
 * stage-2 causal-LM hot path of EMO-Disentanger (Performer / GPT-2 backbones).
 *
 * The reference has NO FFI: its seam is the Python object contract of
 * MusicPerformer / MusicGPT2.  Each entry point below cites
 * the reference lines (relative to /root/reference/stage2_accompaniment) whose
 * arithmetic it replaces; the Python classes in emo-disentanger_amd/model/
 * bind them through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (torch tensors);
 *     the library never allocates or frees tensor memory; scratch is caller-provided
 *     after the matching *_workspace_bytes() query;
 *   - every launch takes an explicit hipStream_t (as void*) and is asynchronous;
 *   - return value: 0 = ok, <0 = error (emo_last_error() gives a thread-local message);
 *   - dims are int64_t, row-major; "ld" = row stride in ELEMENTS;
 *   - dtype: EMO_F32 (parity mode: exact-f32 MFMA/VALU math) or EMO_BF16 (speed mode:
 *     bf16 storage + bf16 MFMA, fp32 accumulation / statistics / scan state).
 *   - dropout masks are never stored: forward and backward regenerate them from
 *     (seed, offset, linear element index).
 */
#ifndef EMO_HIP_H
#define EMO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* emo_stream_t; /* hipStream_t */

enum { EMO_OK = 0, EMO_ERR_INVALID = -1, EMO_ERR_LAUNCH = -2, EMO_ERR_UNSUPPORTED = -3 };
enum { EMO_F32 = 0, EMO_BF16 = 1, EMO_I64 = 2 /* emo_comm_* payloads only */ };
enum { EMO_ACT_NONE = 0, EMO_ACT_RELU = 1, EMO_ACT_GELU_NEW = 2, EMO_ACT_GELU = 3 };   /* GELU_NEW: HF tanh form (GPT-2); GELU: exact erf form
                                                                                       * (F.gelu: the Performer stack's activation='gelu', fast_transformer_decoder.py:50) */
enum { EMO_MUL_NONE = 0, EMO_MUL_NONZERO = 1, EMO_MUL_DGELU_NEW = 2, EMO_MUL_BITMASK = 3, EMO_MUL_DGELU = 4 };   /* DGELU: *= d/dx of the erf form at mul_aux */

int emo_version(void);
/* build options of the library: bit 0 = the opt-in experimental GEMM kernels (emo_gemm_p256.hip: persistent 256 x 256 tile walk, 128 x 512 tile — measured
 * negatives inside the training step, built only with `make EXTRA=-DEMO_EXPERIMENTAL`; without them EMO_GEMM_P256 / EMO_GEMM_Q512 have no effect) */
int emo_build_flags(void);
const char* emo_last_error(void);
/* number of CUs of the current device (for grid sizing on the host side) */
int emo_device_cus(void);

/* ------------------------------------------------------------------ K2/K7/K8: dense GEMM on MFMA
 * C[M,N] = epilogue( A·B )   with
 *   a_trans=0: A stored [M,K] (k contiguous)      a_trans=1: A stored [K,M]
 *   b_trans=0: B stored [N,K] (nn.Linear weight)  b_trans=1: B stored [K,N] (HF Conv1D weight)
 * epilogue order: +bias[n] -> (aux_out = value) -> act -> *mul(mul_aux) -> dropout -> +residual
 * Replaces: F.linear in fast-transformers AttentionLayer / TransformerEncoderLayer
 * (called from model/fast_transformer_decoder.py:28-51), HF Conv1D addmm in GPT2Attention /
 * GPT2MLP (model/music_gpt2.py:42-51,86), dec_out_proj (music_performer.py:27,65), and their
 * autograd backward (dgrad / wgrad).
 */
typedef struct {
    const float* bias;    /* [N] fp32 or NULL */
    int act;              /* EMO_ACT_* */
    void* aux_out;        /* NULL or [M,N] (dtype_out, ld=ldc): value before the activation */
    const void* mul_aux;  /* NULL or [M,N] (dtype_out, ld=ldc), see mul_mode */
    int mul_mode;         /* EMO_MUL_NONZERO: *= (mul_aux!=0)*mul_scale ; EMO_MUL_DGELU_NEW: *= gelu_new'(mul_aux) ;
                           * EMO_MUL_BITMASK: mul_aux is the uint8 bit mask (M*N/8 bytes, tiled layout) written by mask_out, *= bit ? mul_scale : 0 */
    float mul_scale;
    float p_drop;         /* dropout after the activation; element index = m*N+n */
    uint64_t seed, offset;
    const void* residual; /* NULL or [M,N] (dtype_out, ld=ldc) added last */
    /* LayerNorm folded around a decode-step GEMM (bf16, M <= 32, NT; any other shape is refused).  With A' = LN(A)*gamma + beta:
     *   A'.W^T [m][n] = rstd[m] * ((A.(gamma*W)^T)[m][n] - mean[m]*c1[n]) + (beta.W^T)[n],   c1[n] = sum_k gamma[k] W[n][k]
     * the caller passes B = gamma*W, ln_c1 = c1 and folds beta.W^T into `bias`; mean/rstd of the A rows are computed in-kernel
     * (biased variance, ln_eps inside the sqrt: nn.LayerNorm) and optionally exported.  rln_*: the residual is LayerNorm(rln_x)
     * rebuilt from exported statistics (added before bias; needs act = none, no dropout).  Replaces the standalone norm1 / norm2
     * launches of fast-transformers' TransformerEncoderLayer on the one-token decode path. */
    const float* ln_c1;       /* NULL or [N] */
    float ln_eps;
    float* ln_stats_out;      /* NULL or [M,2] (mean, rstd) of the A rows */
    const void* rln_x;        /* NULL or [M,N] (dtype_out, ld=ldc): raw tensor whose LayerNorm is the residual */
    const float* rln_stats;   /* [M,2] (mean, rstd) of the rln_x rows */
    const float* rln_gamma;   /* [N] */
    const float* rln_beta;    /* [N] */
    float* a_rowsum;      /* NULL or [M] fp32: += sum_k op(A)[m][k]; needs a_trans.  For a weight gradient dW = dY^T X this is the bias
                           * gradient (column sums of dY), taken inside the GEMM from the operand fragments instead of a second pass. */
    float* b_rowsum;      /* NULL or [N] fp32: += sum_k op(B)[n][k]; needs a_trans and b_trans (HF Conv1D layout, where dY is the B operand) */
    uint8_t* mask_out;    /* NULL or M*N/8 bytes: one bit per output = (value after act and dropout != 0) — the 1-bit relu.dropout mask the FFN2
                           * dgrad needs (instead of re-reading the [M,N] activation).  TILED layout (r03, byte order r05; only emo_gemm reads it back): the 256
                           * bytes of a 32-row x 64-column tile are contiguous at ((m/32)*(N/64) + n/64)*256; inside, byte
                           * ((n%32)/8*16 + m%16)*4 + (m%32)/16*2 + (n%64)/32 holds columns 8*(n/8) .. +7 of row m, bit j = column 8*(n/8)+j
                           * (a lane's four bytes of a tile are one dword: ONE 256-byte store / load per wave of the A-stationary kernel).  Only with EMO_MUL_BITMASK's
                           * shape class: bf16 in/out, NT, K = 512, M % 128 == 0, M >= 4096, N % 64 == 0, N <= 2048 (the A-stationary kernel); refused elsewhere. */
    void* workspace;      /* NULL or caller scratch for split-K partial sums (plain fp32-output GEMMs = weight gradients): */
    int64_t workspace_bytes; /* with it the splits are summed in a fixed order by a reduce kernel (deterministic, no atomics);
                              * without it they are fp32 atomics into C.  Size: emo_gemm_workspace_bytes(). */
    /* LayerNorm of the A operand inside the product (r05; replaces the standalone norm launch in front of a Linear — norm1 -> linear1 of
     * fast-transformers' TransformerEncoderLayer via fast_transformer_decoder.py:45-51, ln_1 -> c_attn / ln_2 -> c_fc of HF GPT2Block via
     * music_gpt2.py:42-51): A is the RAW tensor; the A-stationary kernel, which holds complete 512-wide rows in registers, computes
     * mean / rstd per row (biased variance, ln_eps inside the sqrt), normalises in place with lna_gamma / lna_beta [K] and, besides the product,
     * writes the normalised rows to lna_out [M, K] (dtype_in, contiguous: residual of the next Linear, operand of the weight gradient) and the
     * statistics to lna_mean / lna_rstd [M].  Only in the A-stationary shape class (bf16, NT, K = 512, M % 128 == 0, M >= 4096); refused elsewhere. */
    const float* lna_gamma;
    const float* lna_beta;
    void* lna_out;
    float* lna_mean;
    float* lna_rstd;
    /* Per-row, per-64-column-block divisor (r06): C[m][n] /= hdiv[((m / hdiv_T) * (N / 64) + n / 64) * hdiv_T + m % hdiv_T] — with hdiv = the FAVOR+
     * normaliser den [B, H, T] (emo_favor_attn_fwd) and C = d(attention output) = the out-projection's dgrad, the product leaves as dN = dout / den
     * (the first thing both backward sweeps of causal_product form), in ONE rounding from the fp32 accumulators, and
     * emo_favor_attn_bwd_dn takes it without the normaliser.  Replaces the d(out)/den scalings inside the causal_product backward reached from
     * fast_transformer_decoder.py:33-40.  Only with a plain epilogue in the A-stationary class (bf16 in / out, NT, K = 512, M % 128 == 0, M >= 4096,
     * N % 64 == 0) and hdiv_T % 32 == 0, M % hdiv_T == 0; refused elsewhere. */
    const float* hdiv;
    int64_t hdiv_T;
} emo_epilogue_t;

/* scratch that lets emo_gemm() run its split-K without atomics (0 = this problem is not split) */
/* diagnostics: kernel family of the calling thread's last emo_gemm (1 skinny, 2 A-stationary K = 512, 3 / 4 256 x 256 tile wgrad / NT,
 * 5 / 6 128 x 128 LDS-DMA / register-staged, 7 exact-fp32, 8 persistent 256 x 256 tile walk (EMO_GEMM_P256=1), 9 128 x 512 tile (EMO_GEMM_Q512=1); + 16 split-K via the workspace, + 32 with the reduce-and-epilogue pass) */
int emo_gemm_last_kernel(void);
/* sizeof(emo_epilogue_t) as the library was built — a binding checks its own mirror of the struct against it before the first call */
int emo_epilogue_size(void);
int64_t emo_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype_in, int dtype_out);

int emo_gemm(const void* A, int a_trans, int64_t lda, const void* B, int b_trans, int64_t ldb,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
             int dtype_in, int dtype_out, int accumulate /* C += result; fp32 out only */,
             const emo_epilogue_t* epi /* may be NULL */, emo_stream_t stream);

/* ------------------------------------------------------------------ K7f: the feed-forward block in one launch (r06)
 *   h1 = LayerNorm(x1) * gamma + beta;  f = dropout(relu(h1 W1^T + b1));  x2 = h1 + dropout(f W2^T + b2)
 * Replaces `norm1 -> linear1 -> activation -> dropout -> linear2 -> dropout -> + residual` of fast-transformers' TransformerEncoderLayer.forward
 * as called from model/fast_transformer_decoder.py:45-51 (two emo_gemm launches until r05).  W1 [d_ff, d_model], W2 [d_model, d_ff] are the
 * nn.Linear weights (bf16 mirrors, k-contiguous rows), b1 / b2 fp32.  Outputs: h1_out [M, d_model] (the residual; operand of the FFN1 weight gradient),
 * mean_out / rstd_out [M] (LayerNorm backward), f_out [M, d_ff] (operand of the FFN2 weight gradient), mask_out (M * d_ff / 8 bytes: the 1-bit
 * relu.dropout mask in emo_epilogue_t.mask_out's tiled layout, read back by EMO_MUL_BITMASK in the FFN2 dgrad), x2_out [M, d_model].  The hidden
 * activation is written once and never re-read in the forward; the residual never leaves the registers.  Dropout element indices as in emo_gemm:
 * (seed, offset_f, m * d_ff + n) for f and (seed, offset_y, m * d_model + n) for the FFN2 output — results are bit-identical to the two-launch
 * form.  Served for bf16, d_model 512, d_ff 2048, M % 128 == 0, M >= 32768 (emo_ffn_fwd_supported() = 1); EMO_ERR_INVALID elsewhere. */
int emo_ffn_fwd_supported(int dtype, int64_t M, int64_t d_model, int64_t d_ff);
int emo_ffn_fwd(const void* x1, const float* gamma, const float* beta, float ln_eps, const void* W1,
                const float* b1, const void* W2, const float* b2, void* h1_out, float* mean_out,
                float* rstd_out, void* f_out, uint8_t* mask_out, void* x2_out, int64_t M, int64_t d_model,
                int64_t d_ff, int dtype, float p_drop, uint64_t seed, uint64_t offset_f, uint64_t offset_y,
                emo_stream_t stream);

/* out[n] (+)= sum_m X[m,n] — bias gradients */
int emo_colsum(const void* X, int dtype, int64_t M, int64_t N, int64_t ld, float* out,
               int accumulate, emo_stream_t stream);

/* ------------------------------------------------------------------ K1: embedding prologue
 * out[b,t,:] = dropout( (E[tok[b,t]] + S[seg[b,t]]) * scale + pe[pos0 + pos_ids[b] + t] )   (pos_ids NULL = zeros)
 * Replaces TokenEmbedding.forward x2 + PositionalEncoding + emb_dropout
 * (model/transformer_helpers.py:81-87,57-63; model/music_performer.py:51-62).
 * pe: fp32 rows of length D (the `pe.pe` buffer [max_pos,1,D] is exactly that).
 * pos_ids (nullable, device int64 [B]): per-sequence position of its first token, replacing pos0 —
 * decode streams of different lengths, and hipGraph replay (no host-side position baked in). */
int emo_embed_fwd(const int64_t* tok, const int64_t* seg, const float* E, const float* S,
                  const float* pe, void* out, int dtype, int64_t B, int64_t T, int64_t D,
                  int64_t V, int64_t n_seg, int64_t pos0, const int64_t* pos_ids, float scale,
                  float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream);
/* dE[tok] += dout*mask*scale ; dS[seg] += ...  (fp32 accumulate, caller zeroes) */
int emo_embed_bwd(const int64_t* tok, const int64_t* seg, const void* dout, int dtype, float* dE,
                  float* dS, int64_t B, int64_t T, int64_t D, int64_t V, int64_t n_seg,
                  float scale, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream);

/* ------------------------------------------------------------------ K6: LayerNorm (eps inside sqrt, biased var)
 * Replaces nn.LayerNorm norm1/norm2 (fast-transformers TransformerEncoderLayer) and ln_1/ln_2 (HF GPT2Block). */
int emo_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int dtype, int64_t M, int64_t D, float eps, emo_stream_t stream);
/* dx = LN'(dy) (+ dres);  dx_drop (optional) = dx * dropmask(p,seed,offset);  dgamma/dbeta += (atomic);
 * dcol (optional, fp32 [D]) += column sums of dx_drop (of dx when dx_drop is NULL) — the bias gradient of the Linear
 * whose output fed this LayerNorm's residual branch, fused here to save one pass over the tensor. */
int emo_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, void* dx_drop, float* dgamma,
                      float* dbeta, float* dcol, int dtype, int64_t M, int64_t D, float p_drop,
                      uint64_t seed, uint64_t offset, emo_stream_t stream);
/* The same with a caller-owned scratch for the three column sums (dgamma, dbeta, dcol): every block writes its partial sums to the workspace
 * and a second small kernel adds them in block order — no atomics, so these gradients are bit-reproducible and the 3 x D atomics per block
 * (a third of the kernel's time at 8192 rows) are gone.  emo_layernorm_bwd_workspace_bytes() = 0 means the shape takes the generic
 * kernel (atomics); workspace = NULL or too small falls back to the atomic accumulation as well.  Contents need not survive the call. */
int64_t emo_layernorm_bwd_workspace_bytes(int dtype, int64_t M, int64_t D);
int emo_layernorm_bwd_ws(const void* dy, const void* x, const float* gamma, const float* mean,
                         const float* rstd, const void* dres, void* dx, void* dx_drop, float* dgamma,
                         float* dbeta, float* dcol, int dtype, int64_t M, int64_t D, float p_drop,
                         uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                         emo_stream_t stream);
/* out = x * dropmask  (backward of a dropout whose forward was fused in a GEMM epilogue) */
int emo_dropout_apply(const void* x, void* out, int dtype, int64_t n, float p_drop, uint64_t seed,
                      uint64_t offset, emo_stream_t stream);

/* ------------------------------------------------------------------ K3+K4: FAVOR+ causal linear attention
 * q,k,v: [B*T, H*dh] views with row stride ld (a fused [B*T,3*H*dh] projection works);
 * omega [dh, n_feat/2] fp32; out [B*T, H*dh] (ld_out); den [B,H,T] fp32 (saved for backward).
 * state_S [B,H,n_feat,dh] / state_z [B,H,n_feat] fp32: optional final scan state (decode prefill).
 * Replaces fast-transformers Favor.forward + CausalLinearAttention.forward + native
 * causal_product (called via model/fast_transformer_decoder.py:28-40) and their backward.
 *
 * workspace: when B*H workgroups cannot fill the GPU (the reference's default batch_size 4 x 8 heads
 * = 32), the scan is cut into P time segments that run in parallel: a state-only pass writes each
 * segment's state increment to the workspace and the main pass starts every segment from the sum
 * of the increments before it (behind it, for the reverse sweep of dk/dv).  The caller owns the
 * scratch: emo_favor_attn_workspace_bytes() gives its size (0 = single segment; same value for
 * fwd and bwd, contents need not survive between calls).  workspace = NULL forces P = 1. */
int64_t emo_favor_attn_workspace_bytes(int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat);
int emo_favor_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                       void* out, int64_t ld_out, float* den, float* state_S, float* state_z,
                       int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat,
                       float eps, void* workspace, int64_t workspace_bytes, emo_stream_t stream);
int emo_favor_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                       const void* out, const void* dout, int64_t ld_out, const float* den,
                       void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T,
                       int64_t H, int64_t dh, int64_t n_feat, float eps, void* workspace,
                       int64_t workspace_bytes, emo_stream_t stream);
/* The same with the forward's workspace handed over: when the segment-parallel scan is in use (emo_favor_attn_workspace_bytes() > 0) the
 * backward first recomputes the per-segment K-state increments that the forward call left in ITS workspace.  A caller that kept that buffer
 * untouched for the matching backward passes it here with kstate_valid = 1 and saves the state-only pass (one launch per layer); the buffer is
 * then reused for the R-state increments as usual.  kstate_valid = 0: exactly emo_favor_attn_bwd. */
int emo_favor_attn_bwd_kstate(const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                              const void* out, const void* dout, int64_t ld_out, const float* den, void* dq,
                              void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                              int64_t dh, int64_t n_feat, float eps, void* workspace, int64_t workspace_bytes,
                              int kstate_valid, emo_stream_t stream);
/* The backward with the incoming gradient already divided by the normaliser (r06): `dn` = dN = dout / den, as the out-projection's dgrad leaves it
 * when its epilogue carries emo_epilogue_t.hdiv = den (one rounding from the fp32 accumulators instead of bf16(dout) then bf16(dout / den)); the kernels
 * then need neither den nor the rescaled operand copies (dD_t = -(dN_t . out_t)).  Same reference lines as emo_favor_attn_bwd (the native
 * causal_product backward reached from fast_transformer_decoder.py:33-40).  Served by the single-segment slice kernels only — bf16, d_head 64,
 * 128 features, T % 32 == 0, B * H >= 256 (emo_favor_attn_bwd_dn_supported() = 1); EMO_ERR_UNSUPPORTED elsewhere, the caller then keeps the plain form. */
int emo_favor_attn_bwd_dn_supported(int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat);
int emo_favor_attn_bwd_dn(const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                          const void* out, const void* dn, int64_t ld_out, void* dq, void* dk, void* dv,
                          int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat,
                          float eps, emo_stream_t stream);
/* one recurrent step per stream: state += phi(k) (x) v ; out = phi(q)^T S / (phi(q).z + eps) */
int emo_favor_decode_step(const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                          float* state_S, float* state_z, void* out, int64_t ld_out, int dtype,
                          int64_t n_streams, int64_t H, int64_t dh, int64_t n_feat, float eps,
                          emo_stream_t stream);
/* ONE-LAUNCH Performer decode step (the token loop of inference.py:250-277 -> MusicPerformer.forward(keep_last_only), music_performer.py:50-70):
 * embedding (transformer_helpers.py:81-87 + PE) -> n_layers post-LN FAVOR+ encoder layers (fast_transformer_decoder.py:54-74: fused q/k/v projection,
 * FAVOR+ recurrent step on the fp32 state, out-projection + residual + LayerNorm, ReLU FFN + residual + LayerNorm) -> dec_out_proj logits, for
 * n_streams <= 32 streams (a multiple of 4), bf16 weights, in a single persistent kernel: 8 groups x 32 workgroups of 12 waves (4 poller + 2 x 4 compute), a group owns 4 streams and
 * exchanges the activations of the 5 dependent products of a layer through tagged 8-byte granules (csrc/emo_decode_persist.hip).
 * Built for d_model 512 / 8 heads / 128 features / d_ff 2048, n_layers <= 15, n_token <= 512.
 *   layer_table : device array [n_layers][16] of pointers: wqkv_packed, bqkv (f32 [3 d], q|k|v), wo_packed, bo, norm1 gamma, beta, w1_packed, b1,
 *                 w2_packed, b2, norm2 gamma, beta, omega (f32 [64][64]), state_S (f32 [n][8][128][64]), state_z (f32 [n][8][128]), unused.
 *                 *_packed = the bf16 nn.Linear weight [N][K] re-ordered per (member, wave 0..3 of the compute half, column tile, k step) into 1-KB MFMA B fragments
 *                 (element (lane, j) = W[16 tile + lane % 16][32 kstep + 8 (lane / 16) + j]); the host mirror builds them (inference.py).
 *   tok, seg    : int64 [n_streams] (seg may be NULL); position of stream s = pos0 + (pos_ids ? pos_ids[s] : 0)
 *   logits      : f32 [n_streams][n_token]
 *   sync_ws     : emo_performer_decode_step_workspace_bytes() bytes, ZEROED ONCE by the caller before the first step and then left alone
 *                 (it carries the launch counter the granule tags are derived from); its last 8 words: [0] != 0 after a step that gave up
 *                 (a workgroup could not be scheduled next to the others within 50 ms) - the caller must check it before trusting the logits.
 *   diag        : NULL, or int64 [32][16][8][4] device words that receive group 0's per-phase time stamps (tools/pd_diag.py). */
int64_t emo_performer_decode_step_workspace_bytes(void);
/* 1 when this device can hold the launch (>= 256 CUs, 96 KB of LDS per workgroup granted, one workgroup per CU by the occupancy query), else 0:
 * callers keep the chain of launches (emo_gemm + emo_favor_decode_step ...) of stage2_accompaniment/inference.py:250-277 then */
int emo_performer_decode_step_supported(void);
int emo_performer_decode_step(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E,
                              const float* Sg, const float* pe, float emb_scale, int64_t pos0, const int64_t* pos_ids,
                              const void* wout_packed, const float* bout, int64_t n_token, float* logits, int64_t n_streams,
                              int64_t d_model, int64_t n_head, int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes,
                              float eps, float ln_eps, int64_t* diag, emo_stream_t stream);
/* The same launch with the NEXT token drawn inside it (the sampling half of the loop of inference.py:252-277, arithmetic of emo_sample_nucleus_step:
 * same device code): member s < 4 of a group draws stream 4 g + s from `logits` AS THE PREVIOUS STEP LEFT IT (temperature, nucleus top_p, uniform
 * u_steps[step[r] * n_real + r]), writes the token to tok_out[r] and seq[r * ld_seq + col0 + step[r]], and the step then embeds it at position
 * pos0 + step[r] + 1; step[r] is incremented when the launch is done.  Streams r >= n_real are idle padding (n_streams = n_real rounded up to 4). */
int emo_performer_decode_step_sampled(const void* layer_table, int64_t n_layers, const int64_t* seg, const float* E, const float* Sg,
                                      const float* pe, float emb_scale, int64_t pos0, const void* wout_packed, const float* bout,
                                      int64_t n_token, float* logits, int64_t n_streams, int64_t n_real, int64_t d_model, int64_t n_head,
                                      int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float eps, float ln_eps,
                                      float temperature, float top_p, const float* u_steps, int64_t* step, int64_t* seq, int64_t ld_seq,
                                      int64_t col0, int64_t* tok_out, emo_stream_t stream);

/* ONE-LAUNCH GPT-2 decode step (r06): the same persistent launch for the GPT-2 backbone (stage2_accompaniment/model/music_gpt2.py -> HF GPT2Block:
 * pre-LN, gelu_new, no ln_f; token loop of inference.py:250-277 with the KV cache BASELINE configs[3] names).  d_model 512 / 8 heads / d_ff 2048,
 * n_layers <= 15, n_token <= 512, n_streams <= 32 (a multiple of 4), bf16.  Arguments as emo_performer_decode_step, except:
 *   layer_table : [n_layers][16] pointers: c_attn packed, bias (f32 [3 d]), attn.c_proj packed, bias, ln_2 gamma, beta, c_fc packed, bias, mlp.c_proj
 *                 packed, bias, ln_1 gamma, beta OF THE NEXT BLOCK (any valid pointer for the last block), unused, K cache, V cache, unused.
 *                 Packed = the TRANSPOSED Conv1D weight ([out][in]) in the fragment order of emo_performer_decode_step.
 *                 K / V cache: bf16 [n_streams][8][kv_tmax][64] (head-major, kv_tmax <= 2048); the step appends the token's key / value row at index
 *                 pos = pos0 + pos_ids[s] (= the number of rows already cached) and attends over rows 0 .. pos.
 *   ln0         : f32 [2][512]: gamma | beta of block 0's ln_1. */
int emo_gpt2_decode_step_supported(void);
int emo_gpt2_decode_step(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E, const float* Sg,
                         const float* pe, float emb_scale, int64_t pos0, const int64_t* pos_ids, const float* ln0, int64_t kv_tmax,
                         const void* wout_packed, const float* bout, int64_t n_token, float* logits, int64_t n_streams, int64_t d_model,
                         int64_t n_head, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float ln_eps, int64_t* diag, emo_stream_t stream);
/* with the next token drawn inside the launch, as emo_performer_decode_step_sampled (row index of the appended key = pos0 + step[r] + 1) */
int emo_gpt2_decode_step_sampled(const void* layer_table, int64_t n_layers, const int64_t* seg, const float* E, const float* Sg, const float* pe,
                                 float emb_scale, int64_t pos0, const float* ln0, int64_t kv_tmax, const void* wout_packed, const float* bout,
                                 int64_t n_token, float* logits, int64_t n_streams, int64_t n_real, int64_t d_model, int64_t n_head, int64_t d_ff,
                                 void* sync_ws, int64_t sync_ws_bytes, float ln_eps, float temperature, float top_p, const float* u_steps,
                                 int64_t* step, int64_t* seq, int64_t ld_seq, int64_t col0, int64_t* tok_out, emo_stream_t stream);

/* FAVOR+ omega draw (fast-transformers orthogonal_random_matrix_, called from new_feature_map() on every
 * forward — SURVEY F8): gauss [n_layers, ceil((n_feat/2)/dh), dh, dh] ~ N(0,1) from the caller's RNG ->
 * omega [n_layers, dh, n_feat/2] with orthogonal columns per block scaled by the row norms of the block. */
int emo_favor_draw_omega(const float* gauss, float* omega, int64_t n_layers, int64_t dh, int64_t n_feat,
                         emo_stream_t stream);

/* ------------------------------------------------------------------ K5: causal softmax attention (GPT-2)
 * Replaces HF GPT2Attention._attn (model/music_gpt2.py:86): softmax(q k^T/sqrt(dh) + causal) [dropout] v.
 * lse [B,H,T] fp32 saved for backward.  Dropout index = ((b*H+h)*T + i)*T + j. */
int emo_softmax_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, void* out,
                         int64_t ld_out, float* lse, int dtype, int64_t B, int64_t T, int64_t H,
                         int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream);
/* delta_ws: caller scratch [B,H,T] fp32 (dO.O per query row: written by the dQ pass, read by the dK/dV pass). */
int emo_softmax_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* out,
                         const void* dout, int64_t ld_out, const float* lse, float* delta_ws,
                         void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T,
                         int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset,
                         emo_stream_t stream);
/* The same two calls with the attention-dropout keep decisions handed from the forward to the backward: the forward writes one bit per
 * score at or below the diagonal (32-bit words [B*H][T/64 key tiles][2][T rows], emo_softmax_attn_keep_bytes() bytes — 0 when the call is
 * not served by the 32 x 32 MFMA kernels: then pass keep = NULL), the dK/dV pass reads the bits instead of re-evaluating the keyed hash per
 * score.  Results are bit-identical to the calls without the buffer (same hash, evaluated once).  keep = NULL: exactly the calls above. */
int64_t emo_softmax_attn_keep_bytes(int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop);
int emo_softmax_attn_fwd_keep(const void* q, const void* k, const void* v, int64_t ld, void* out,
                              int64_t ld_out, float* lse, int dtype, int64_t B, int64_t T, int64_t H,
                              int64_t dh, float p_drop, uint64_t seed, uint64_t offset, void* keep,
                              int64_t keep_bytes, emo_stream_t stream);
int emo_softmax_attn_bwd_keep(const void* q, const void* k, const void* v, int64_t ld, const void* out,
                              const void* dout, int64_t ld_out, const float* lse, float* delta_ws,
                              void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B,
                              int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed,
                              uint64_t offset, const void* keep, int64_t keep_bytes, emo_stream_t stream);
/* decode: one query row per stream against a KV cache [n_streams, T_max, H*dh]; lens[s] + lens_off = valid keys INCLUDING the new
 * token.  k_new / v_new [n_streams, H*dh] (ld_new; both or neither NULL): the new token's key / value rows, appended to the caches
 * at position len-1 by the kernel itself (the HF `past_key_values` concat of GPT2Attention). */
int emo_softmax_attn_decode(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max,
                            const int64_t* lens, int64_t lens_off, const void* k_new, const void* v_new,
                            int64_t ld_new, void* out, int64_t ld_out, int dtype, int64_t n_streams,
                            int64_t H, int64_t dh, emo_stream_t stream);
/* The same with the cache layout chosen by the caller (r06): head_major = 0: [n_streams, T_max, H*dh] as above; head_major = 1:
 * [n_streams, H, T_max, dh] — the keys / values of one (stream, head) are one contiguous run, which is what the one workgroup per (stream, head)
 * streams (the layout HF's `past_key_values` itself uses: [batch, head, seq, head_dim]). */
int emo_softmax_attn_decode_layout(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max,
                                   const int64_t* lens, int64_t lens_off, const void* k_new, const void* v_new,
                                   int64_t ld_new, void* out, int64_t ld_out, int dtype, int64_t n_streams,
                                   int64_t H, int64_t dh, int head_major, emo_stream_t stream);

/* ------------------------------------------------------------------ K5r: relative-position causal attention (stage-1 Transformer-XL)
 * SURVEY §8 f-1.  Replaces RelPartialLearnableMultiHeadAttn's score / softmax / value product
 * (stage1_compose/model/optimus_txl_decoder.py:331-366) including `_rel_shift` (:280-293):
 *   score[i][j] = ((q_i + r_w_bias).k_j + (q_i + r_r_bias).R[i-j]) / sqrt(dh),  j <= i
 *   prob = softmax -> dropout -> p / (sum_j p + 1e-8);  out = prob v
 * q,k,v: [B*T, H*dh] views (row stride ld; batch-major).  r_dist [n_dist >= T, H*dh] (ld_r) = r_net(pos_emb) indexed BY DISTANCE
 * (row d = the reference's r_head_k[klen-1-d]).  r_w_bias / r_r_bias [H, dh] fp32.  lse [B,H,T], zden [B,H,T] (may be NULL:
 * the renormalisation denominator E/l + 1e-8) are saved for the backward pass. */
int emo_relpos_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const void* r_dist,
                        int64_t ld_r, int64_t n_dist, const float* r_w_bias, const float* r_r_bias,
                        void* out, int64_t ld_out, float* lse, float* zden, int dtype, int64_t B,
                        int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed,
                        uint64_t offset, emo_stream_t stream);
/* Backward, query-tile pass: recomputes the probabilities per query tile and returns dq = dq_content + dq_relative
 * (dq_content = ds.K/sqrt(dh), dq_relative[i] = sum_j ds_ij R[i-j]/sqrt(dh), both accumulated in-kernel) plus dq_rel = the relative part
 * alone ([B*T, H*dh], pitch ld_rel, dtype of q): d r_r_bias = colsum(dq_rel), d r_w_bias = colsum(dq) - colsum(dq_rel).
 * delta (may be NULL) [B,H,T] fp32: dO.O per query row, for emo_relpos_attn_bwd_kv / emo_relpos_attn_bwd_r. */
int emo_relpos_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* r_dist,
                        int64_t ld_r, int64_t n_dist, const float* r_w_bias, const float* r_r_bias,
                        const void* out, const void* dout, int64_t ld_out, const float* lse,
                        const float* zden, void* dq, int64_t ld_d, void* dq_rel, int64_t ld_rel,
                        float* delta, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh,
                        float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream);
/* Distance-window pass of the backward: dR [T, ld_dr] fp32 (overwritten) = gradient of r_dist rows 0..T-1,
 * dR[dist][h*dh + d] = sum_{b,i} ds[b,h,i,i-dist] (q_i + r_r_bias)[d] / sqrt(dh).  One workgroup per (b, h, tile diagonal); the partial
 * windows go through `workspace` (emo_relpos_attn_bwd_r_workspace_bytes) and are summed in a fixed order (deterministic).  qu, qv, delta as
 * for emo_relpos_attn_bwd_kv.  Replaces the reference's autograd through _rel_shift (optimus_txl_decoder.py:280-293, 331-366). */
int64_t emo_relpos_attn_bwd_r_workspace_bytes(int64_t B, int64_t T, int64_t H, int64_t dh);
int emo_relpos_attn_bwd_r(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v,
                          int64_t ld, const void* r_dist, int64_t ld_r, int64_t n_dist, const void* dout,
                          int64_t ld_out, const float* lse, const float* zden, const float* delta,
                          float* dR, int64_t ld_dr, void* workspace, int64_t workspace_bytes, int dtype,
                          int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed,
                          uint64_t offset, emo_stream_t stream);
/* Key-tile pass of the backward: dk, dv in one kernel.  qu = q + r_w_bias, qv = q + r_r_bias
 * [B*T, H*dh] (pitch ld_q) materialised by the caller; delta [B,H,T] = dO.O per query row as exported by emo_relpos_attn_bwd. */
int emo_relpos_attn_bwd_kv(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v,
                           int64_t ld, const void* r_dist, int64_t ld_r, int64_t n_dist, const void* dout,
                           int64_t ld_out, const float* lse, const float* zden, const float* delta,
                           void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                           int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream);
/* one query row per stream against a KV cache (the reference re-projects its cached hidden states `mems` every step,
 * plain_transformer.py:52-59; k / v of a position do not change, so they are cached instead).  Keys j in
 * [max(0, len-1-mem_len), len) with len = lens[s] + lens_off; the distance of key j is len-1-j.  k_new / v_new as in
 * emo_softmax_attn_decode.  Eval semantics (no dropout). */
int emo_relpos_attn_decode(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max,
                           const int64_t* lens, int64_t lens_off, int64_t mem_len, const void* k_new,
                           const void* v_new, int64_t ld_new, const void* r_dist, int64_t ld_r,
                           int64_t n_dist, const float* r_w_bias, const float* r_r_bias, void* out,
                           int64_t ld_out, int dtype, int64_t n_streams, int64_t H, int64_t dh,
                           emo_stream_t stream);

/* ------------------------------------------------------------------ K9: cross-entropy with ignore_index
 * Replaces F.cross_entropy in compute_loss (model/music_performer.py:72-81).
 * fwd: row_lse[m]; acc[0] += sum of -logp over kept rows, acc[1] += #kept rows (caller zeroes acc).
 * bwd: dlogits[m,v] = (exp(l - lse) - [v==tgt]) * (tgt!=ignore) * gscale[0]   (gscale: device scalar) */
int emo_xent_fwd(const float* logits, const int64_t* tgt, int64_t M, int64_t V, int64_t ignore_index,
                 float* row_lse, float* acc, emo_stream_t stream);
int emo_xent_bwd(const float* logits, const int64_t* tgt, const float* row_lse, const float* gscale,
                 void* dlogits, int64_t ld_out /* >= V; columns V..ld_out-1 are zero-filled */,
                 int dtype_out, int64_t M, int64_t V, int64_t ignore_index, emo_stream_t stream);

/* ------------------------------------------------------------------ K10/K12: sampling + accuracy
 * argmax over V per row (greedy / parity mode; first max wins like np.argmax / torch.argmax).
 * nucleus: softmax(l/temp) -> sort desc -> keep through the token that crosses top_p
 * (inference.py:71-100 semantics incl. F12) -> renormalise -> draw from u[row] in [0,1). */
int emo_argmax(const float* logits, int64_t rows, int64_t V, int64_t* out, emo_stream_t stream);
int emo_sample_nucleus(const float* logits, int64_t rows, int64_t V, float temperature, float top_p,
                       const float* u, int64_t* out, emo_stream_t stream);
/* One step of the lock-step generation loop (inference.py:252-327 with grammar checks off) with all loop state on the device, so
 * the step can be captured once in a hipGraph and replayed: stream r draws with u_steps[step[r], r] (u_steps: [n_steps, rows]),
 * the sampled id goes to out[r] and to seq[r, col0 + step[r]] (seq may be NULL), then step[r] += 1. */
int emo_sample_nucleus_step(const float* logits, int64_t rows, int64_t V, float temperature,
                            float top_p, const float* u_steps, int64_t* step, int64_t* seq,
                            int64_t ld_seq, int64_t col0, int64_t* out, emo_stream_t stream);
/* counts[0..5] += {nonpad, nonpad&correct, chord, chord&correct, melody, melody&correct} (train.py:184-193) */
int emo_accuracy_counts(const float* logits, const int64_t* tgt, const int64_t* chord,
                        const int64_t* melody, int64_t M, int64_t V, int64_t pad, int64_t* counts,
                        emo_stream_t stream);

/* ------------------------------------------------------------------ optimizer plumbing (K11, SURVEY f-3)
 * sumsq: acc[0] = sum x^2, bitwise reproducible (fixed summation order); acc = EMO_SUMSQ_FLOATS floats of caller scratch, zeroed once
 * before the first call (acc[1..1024] block partials, acc[1025] a ticket word the kernel returns to zero).  clip_coef: coef[0] = min(1, max_norm/(sqrt(sumsq*pre*pre)+1e-6)) * pre
 * (pre = 1/world for DP-averaged grads).  adam: torch.optim.Adam semantics (no amsgrad, wd=0),
 * grads scaled by gscale[0]; optionally refreshes the bf16 compute copy of the weights. */
#define EMO_SUMSQ_FLOATS 1026
int emo_sumsq(const float* x, int64_t n, float* acc, emo_stream_t stream);
/* coef = pre' * min(1, max_norm / (sqrt(sumsq) * pre' + 1e-6)) with pre' = pre / denom[0] (denom NULL => 1): the clip of
 * torch.nn.utils.clip_grad_norm_ (train.py:79) applied to the pre-scaled gradient.  Data parallel: pre = 1/world for equal
 * token counts, or pre = 1 and denom = the all-reduced number of non-pad target tokens (token-weighted global mean). */
int emo_clip_coef(const float* sumsq, float max_norm, float pre, const float* denom, float* coef, emo_stream_t stream);
int emo_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                  float beta1, float beta2, float eps, int64_t step, const float* gscale,
                  emo_stream_t stream);
int emo_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, emo_stream_t stream);
/* o1 = x + b1, o2 = x + b2 with per-column fp32 biases [D]; x [M, D] (row pitch ld), o1 / o2 contiguous [M, D], same dtype.  The biased
 * query copies qu = q + r_w_bias, qv = q + r_r_bias of the relative-position attention (stage1_compose/model/optimus_txl_decoder.py:331-341:
 * rw_head_q / rr_head_q) for emo_relpos_attn_bwd_kv / _r, in one launch. */
int emo_add_bias2(const void* x, int64_t ld, const float* b1, const float* b2, void* o1, void* o2, int dtype,
                  int64_t M, int64_t D, emo_stream_t stream);
/* n bf16 transposes in one launch (no reference counterpart: the transposed weight mirrors that let every dgrad run as a k-contiguous NT
 * product, refreshed after an optimizer step).  desc: DEVICE array of n records of six int64 {src pointer, dst pointer, rows, cols, index of
 * the record's first 64 x 64 tile, tiles per row = ceil(cols / 64)}; src is [rows, cols] row-major, dst [cols, rows]; total_tiles = sum over
 * records of ceil(rows / 64) * ceil(cols / 64); rows and the pointers must allow 16-B accesses (rows % 8 == 0, cols % 8 == 0 fast path). */
int emo_transpose_batch(const int64_t* desc, int n, int64_t total_tiles, emo_stream_t stream);
/* Stream fork / join without a host round trip through torch's Stream objects (no reference counterpart: the weight-gradient products of
 * a layer run on a second stream beside the dgrad chain at small token counts): `waiter` continues only after everything queued on `signaler`
 * so far.  Asynchronous; events are owned by the library (per calling thread). */
int emo_stream_wait(emo_stream_t waiter, emo_stream_t signaler);

/* ------------------------------------------------------------------ data-parallel exchange (RCCL over xGMI)
 * The reference trains on one GPU; the build shards the batch over one process per GPU and adds ONE exchange per optimizer
 * step between loss.backward() and clip_grad_norm_ (insertion point train.py:76-81): a sum all-reduce of the flat fp32
 * gradient buffer (+ the non-pad token count in its last slot), and one broadcast of parameters / omega at start-up.
 * RCCL is bound with dlopen at the first call (no link-time dependency).  One communicator per process, bound to the
 * CURRENT HIP device at emo_comm_init; all transfers are in place, asynchronous on `stream`.
 *   emo_comm_bind      : dlopen RCCL and resolve its symbols, nothing else (every rank can test locally that the plane is usable)
 *   emo_comm_unique_id : rank 0 creates the 128-byte rendezvous id; the caller ships it to the other ranks (any side channel)
 *   emo_comm_init      : collective over all ranks (ncclCommInitRank)
 *   emo_comm_allreduce : buf[i] = sum over ranks (dtype EMO_F32 / EMO_BF16 / EMO_I64)
 *   emo_comm_broadcast : buf <- root's buf
 *   emo_comm_world/rank: 0 / -1 before init
 */
int emo_comm_bind(void);
int emo_comm_unique_id(void* id128);
int emo_comm_init(const void* id128, int rank, int world);
int emo_comm_world(void);
int emo_comm_rank(void);
int emo_comm_allreduce(void* buf, int64_t count, int dtype, emo_stream_t stream);
int emo_comm_broadcast(void* buf, int64_t count, int dtype, int root, emo_stream_t stream);
int emo_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* EMO_HIP_H */
