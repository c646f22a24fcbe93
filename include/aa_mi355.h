/*
 * aa_mi355.h -- C ABI of libaa_mi355.so: the MI355X (gfx950) kernels behind the animate-anything
 * denoising hot path (UNet3DConditionModel forward + AutoencoderKL encode/decode).
 *
 * The reference has no native/FFI boundary: its arithmetic is reached through Python modules of
 * diffusers==0.24.0.  Each entry point below therefore cites the reference
 * Python operation(s) it replaces.  Binding: ctypes (animate_anything_amd/_lib.py); see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator); the library never
 *    allocates, frees or retains device memory;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, nothing
 *    synchronises, so calls are hipGraph-capturable;
 *  - return 0 on success, a negative AA_E_* code otherwise; aa_last_error() gives the message
 *    (thread-local); no C++ exception crosses the ABI;
 *  - activations are channels-last: an image batch [N,C,H,W] of the reference is stored as
 *    [N*H*W tokens][C] (C contiguous); a clip [B,C,T,H,W] as [(B*T)*H*W][C];
 *  - dtype AA_F16 / AA_BF16 selects the storage type of activations and weights; all accumulation
 *    and all statistics are fp32.
 */
#ifndef AA_MI355_H
#define AA_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AA_VERSION 109

enum { AA_F16 = 0, AA_BF16 = 1, AA_F32 = 2 };
enum { AA_OK = 0, AA_E_SHAPE = -1, AA_E_DTYPE = -2, AA_E_ALIGN = -3, AA_E_WORKSPACE = -4, AA_E_HIP = -5 };
enum { AA_ACT_NONE = 0, AA_ACT_SILU = 1,
       AA_ACT_GELU = 2, AA_ACT_QUICK_GELU = 3 /* aa_blend only: erf GELU / x*sigmoid(1.702 x) (CLIP MLP, transformers hidden_act) */ };

int aa_version(void);
const char* aa_last_error(void);

/* ----------------------------------------------------------------------------------------------
 * aa_conv_gemm: out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )  on the matrix cores.
 *
 * A is gathered on the fly from a channels-last activation (implicit GEMM):
 *   m -> (img, y, x) over an [n_img, h_out, w_out] output grid,
 *   k -> (tap = dy*kw+dx, channel c), source pixel (y*stride - pad_h + dy, x*stride - pad_w + dx)
 *   of a virtual [h_virt, w_virt] grid that is the nearest-neighbour resize of the stored
 *   [h_in, w_in] grid (h_virt == h_in when there is no upsample); out-of-range taps read zero.
 *   Channels c < c0 come from a0, the rest from a1 (channel concat without a copy).
 * One entry point covers every contraction of the path:
 *   nn.Linear / Conv2d 1x1      kh=kw=1               (diffusers Attention.to_q/k/v/out, FeedForward,
 *                                                     proj_in/out, time_emb_proj, conv_shortcut)
 *   nn.Conv2d 3x3               kh=kw=3, pad 1        (ResnetBlock2D.conv1/2, conv_in2, conv_out;
 *                                                     reference unet_3d_condition_mask.py:140-142,264)
 *   Downsample2D                stride 2              (unet_3d_blocks.py:476,591)
 *   Upsample2D                  h_virt = 2*h_in or output_size (unet_3d_blocks.py:709,819,761-763)
 *   nn.Conv3d (3,1,1)           kh=3, kw=1 with h=T, w=H*W  (TemporalConvLayer, unet_3d_blocks.py:276..808)
 * Epilogue, in order: + bias[n] (or bias[m]); + rowvec[m / rowvec_div, n] (time embedding,
 * diffusers ResnetBlock2D); activation; GEGLU pairing (diffusers GEGLU: value * gelu_erf(gate));
 * * acc_scale; + residual[m, n]; * out_scale; store as `out_dtype`.
 * W is pre-packed by the host: [n_pad, k_pad] row-major, k ordered (tap, channel) - or, see k_order,
 * (64-channel chunk, tap, channel) for multi-tap filters - zero padded,
 * and for GEGLU interleaved in blocks of 32 value rows / 32 gate rows (geglu = 32: the value block and its gate
 * block are neighbouring 32-column accumulator blocks of one wavefront; needs n_out == n_pad).
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaConvGemm {
    const void* a0;
    const void* a1;        /* NULL when c1 == 0 */
    const void* w;         /* packed weights [n_pad][k_pad] */
    const void* bias;      /* [n_out] (or [M] when bias_per_row), storage dtype, may be NULL */
    const void* rowvec;    /* [M / rowvec_div][n_out], storage dtype, may be NULL */
    const void* residual;  /* [M][ldr], storage dtype, may be NULL */
    void* out;             /* [M][ldo] */
    int32_t c0, c1;
    int32_t n_img, h_in, w_in, h_virt, w_virt, h_out, w_out;
    int32_t kh, kw, stride, pad_h, pad_w;
    int32_t n_out;         /* logical output columns (before GEGLU halving) */
    int32_t n_pad, k_pad;  /* packed weight extents: n_pad % 128 == 0 (or 64), k_pad % 64 == 0 */
    int32_t rowvec_div;
    int32_t ldo, ldr;
    int32_t act;           /* AA_ACT_* */
    int32_t geglu;         /* 0, or 32: packed columns are (32 value | 32 gate) blocks, output has n_out/2 columns */
    int32_t bias_per_row;
    int32_t dtype;         /* AA_F16 | AA_BF16: activations + weights */
    int32_t out_dtype;     /* AA_F16 | AA_BF16 (== dtype) or AA_F32 */
    float out_scale;
    void* workspace;       /* optional fp32 scratch for split-K (aa_conv_gemm_workspace bytes); NULL = never split */
    int64_t workspace_bytes;
    int32_t k_order;       /* 0: packed K is (tap, channel); 1: (64-channel chunk, tap, channel) - LDS-DMA path only */
    int32_t debug;         /* 0 in production.  Ablation bits for profiling: 1 = skip operand DMA after the first tile,
                              2 = skip the MFMA phase (results are garbage); 4 = size the
                              round-splitting heuristic for a 2-CU chip (exercises it on small test shapes);
                              16 = row-major tile order inside an XCD's range instead of the grouped one (A/B of round 6) */
    int32_t tile;          /* -1: library picks the tile shape; >= 0: index into the tile table (autotuning) */
    int32_t k_splits;      /* 0: library decides whether to split K; >= 1: this many K ranges (needs the workspace
                              aa_conv_gemm_workspace reports for the same descriptor; ignored when not applicable) */
    int32_t rowvec_ld;     /* row pitch of `rowvec` in elements (a slice of a wider matrix); 0 = n_out */
    float acc_scale;       /* factor on the activated value in front of the residual add; 0 means 1 (version 102: the learned
                              spatial/temporal blend of diffusers AlphaBlender, x_spatial + (1 - alpha) * temporal branch) */
    int32_t out_sy, out_sx; /* version 105.  0 / 1: GEMM row m = output row m.  > 1: GEMM row (img, y, x) of the [n_img, h_out, w_out]
                              grid is written to pixel (y * out_sy + out_oy, x * out_sx + out_ox) of an [n_img, h_out * out_sy,
                              w_out * out_sx] output grid (`out` points at that grid; rowvec / residual stay indexed by GEMM row).
                              Upsample2D (nearest x2 + 3x3 convolution, unet_3d_blocks.py:709,819) is carried out as four 2x2
                              convolutions on the STORED grid - one per output parity class, the 3x3 taps that read the same
                              stored pixel pre-summed at pack time: 4/9 of the multiply-adds, identical result up to the
                              rounding of the summed weights */
    int32_t out_oy, out_ox;
    /* version 105: LayerNorm folded into the contraction that consumes it (diffusers BasicTransformerBlock: norm1 -> to_q|to_k|to_v,
     * norm2 -> to_q, norm3 -> GEGLU.proj; the LayerNorm kernel and the normalised tensor disappear).  With W' = W diag(gamma):
     *   LN(x) W^T + b  =  rstd[m] * (x W'^T)[m, n]  -  rstd[m] * mean[m] * colsum(W')[n]  +  (b + beta W^T)[n]
     * The row statistics come from the epilogue of the contraction that PRODUCED x (row_stats). */
    const float* ln_stats;  /* consumer, per row of the A operand; NULL = no fold.  ln_parts == 0: [M][4] fp32 (-mean, sqrt(var + eps), rstd, 0) -
                               aa_ln_finalize's output.  ln_parts > 0 (version 106): [M][ln_parts][2] fp32, the producing call's `row_stats`
                               as it left them - the kernel finalises them itself (biased variance over c0 channels, `ln_eps`), no launch in between */
    const float* ln_cols;   /* consumer: [2][n_pad] fp32: colsum(W')[n], then (b + beta W^T)[n]; `bias` must be NULL */
    int32_t ln_parts;
    float ln_eps;           /* with ln_parts > 0: the LayerNorm's epsilon (> 0) */
    float* row_stats;       /* producer: [M][row_stats_parts][2] fp32: (sum, sum of squares) of the stored output row over the columns of
                               one wave of the tile - written iff row_stats_parts == aa_conv_gemm_row_stats_parts(d) > 0 */
    int32_t row_stats_parts;
    /* version 106: K-split calls that finish INSIDE the kernel.  `tickets` = a caller-owned, zero-initialised int32 array of at
     * least aa_conv_gemm_tickets(d) entries that holds zeros between calls (the kernels leave it zeroed; one array per stream -
     * calls in flight on different streams must not share one).  With it a K-split launch of a hand-scheduled tile needs no reduce
     * launch: every workgroup writes its fp32 partial tile, takes a ticket of its output tile, and the LAST one to arrive sums the
     * partials in split order (a fixed order whichever workgroup it is: results are bit-reproducible, no floating-point atomics)
     * and runs the usual epilogue.  NULL: partials + splitk reduce launch as before. */
    int32_t tickets_len;
    int32_t* tickets;
    /* version 107: a producer whose tile spans the whole output row (one column tile: aa_conv_gemm_row_coef_ok(d) == 1) finishes the row
     * statistics itself - its waves exchange their partial sums through LDS behind the epilogue - and writes the per-row coefficients
     * [M][4] fp32 (-mean, sqrt(var + eps), rstd, 0) that aa_ln_finalize would have produced from `row_stats` (same arithmetic, same
     * summation order): the consuming call takes them as `ln_stats` with ln_parts == 0 and no launch sits in between.  With `row_coef`
     * set, `row_stats` must be NULL (row_stats_parts 0); the row width is n_out. */
    float* row_coef;
    float row_coef_eps;     /* the consuming LayerNorm's epsilon (> 0) */
    int32_t _reserved107;
} AaConvGemm;

/* Bytes of fp32 scratch with which aa_conv_gemm would split the K loop of this call over several workgroups
 * (few output tiles, long K: the small-M levels); 0 when it would not. */
size_t aa_conv_gemm_workspace(const AaConvGemm* d);
int aa_conv_gemm(const AaConvGemm* d, void* stream);
/* Number of kernel launches aa_conv_gemm makes for this descriptor (version 103): 1, plus a launch for a split-off last
 * round of tiles, plus split-K reduce launches - what a profiler counts per call (bench.py `roofline.kernel_launches_per_step`). */
int aa_conv_gemm_launch_count(const AaConvGemm* d);
/* Ticket counters this call would use when `tickets` is given (version 106): the workgroups of its K-split launch, 0 when the call
 * is not split along K or its tile keeps the reduce launch (compiled tiles, a folded LayerNorm). */
int aa_conv_gemm_tickets(const AaConvGemm* d);
/* How many of aa_conv_gemm_launch_count's launches are split-K reduce launches (version 106; bench.py reports the contraction
 * launches and the reduce launches of a step separately). */
int aa_conv_gemm_reduce_launches(const AaConvGemm* d);
/* Partial row statistics per output row that aa_conv_gemm would write for this descriptor when `row_stats` is set (version 105):
 * 0 when the way it carries the call out cannot emit them (compiled tiles, K splits, a split-off last round) - the caller then runs
 * the LayerNorm it wanted to fold as a kernel. */
int aa_conv_gemm_row_stats_parts(const AaConvGemm* d);
/* 1 iff this call (with `row_coef` in place of `row_stats`) would write finished row coefficients (version 107): a call that can emit row
 * statistics at all, carried out by a tile as wide as the packed output (one column tile).  Answered for the descriptor as it stands
 * (tile, k_splits, workspace), like aa_conv_gemm_row_stats_parts. */
int aa_conv_gemm_row_coef_ok(const AaConvGemm* d);
/* Between the two (version 105): stats [rows][parts][2] fp32 partial (sum, sum of squares) over `channels` values per row ->
 * coef [rows][4] fp32 = (-mean, sqrt(var + eps), rstd, 0), the `ln_stats` operand of the consuming aa_conv_gemm
 * (torch.nn.LayerNorm statistics: biased variance; reference use: diffusers BasicTransformerBlock.norm1 / norm2 / norm3). */
int aa_ln_finalize(const float* stats, int32_t parts, float* coef, int64_t rows, int32_t channels, float eps, void* stream);

/* Tile table of the LDS-DMA contraction kernel (what `AaConvGemm.tile` indexes): fills info[0..6] = rows, columns, wave
 * rows, wave columns, K step, ring stages, workgroups per CU of entry `idx`; returns 0, or -1 past the end of the table. */
int aa_conv_gemm_tile_info(int idx, int32_t info[7]);

/* Kernel family / schedule of entry `idx` (version 106): bit 0 halo-slab 3x3 kernel (no K split), bit 1 hand-scheduled stream
 * (conv_gemm_x.h), bit 2 staggered / bit 3 spread DMA issue of the compiled kernel, bits 8.. the hand-scheduled kernel's DMA
 * schedule (DP3 | DP0 << 4 | DP1 << 8); -1 past the end of the table.  Part of what identifies a saved tile choice. */
int aa_conv_gemm_tile_flags(int idx);

/* 1 when entry `idx` of the tile table can carry out this call (packed width divisible by the tile, GEGLU pairing, the
 * geometry preconditions of the halo-slab 3x3 kernel), else 0: what an autotuner should restrict itself to. */
int aa_conv_gemm_tile_ok(const AaConvGemm* d, int idx);

/* Tuning / test aid: force tile shape `cfg` (index into the table in csrc/aa_api_impl.h) for every
 * following aa_conv_gemm call OF THE CALLING THREAD whose packed width it divides; cfg < 0 restores the automatic
 * choice (thread-local, like aa_last_error: the library keeps no process-global mutable state).
 * cfg <= -100 (version 103, bisecting aid): withdraw hand-scheduled tiles from every later choice of the calling thread -
 * bit i of (-100 - cfg) stands for table entry 36 + i; -100 offers them all again. */
void aa_set_tile_override(int cfg);

/* ----------------------------------------------------------------------------------------------
 * aa_groupnorm: y = [silu]( (x - mean) * rstd * gamma + beta ), statistics per (image group, channel
 * group) over `tokens_per_group` consecutive tokens x (C / num_groups) channels.
 *   tokens_per_group = H*W     -> nn.GroupNorm on [N,C,H,W]   (ResnetBlock2D.norm1/2, Transformer2DModel.norm,
 *                                 conv_norm_out; reference unet_3d_condition_mask.py:255-257)
 *   tokens_per_group = T*H*W   -> nn.GroupNorm on [B,C,T,H,W] (TemporalConvLayer, TransformerTemporalModel.norm)
 * x may be the channel concat of two tensors (up-block resnets, reference unet_3d_blocks.py:731,828).
 * `workspace` holds fp32 partial sums: aa_groupnorm_workspace() bytes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaGroupNorm {
    const void* x0;
    const void* x1;      /* NULL when c1 == 0 */
    const void* gamma;   /* [c0+c1] storage dtype */
    const void* beta;
    void* y;             /* [tokens][c0+c1] */
    int32_t c0, c1;
    int32_t n_groups_img;      /* number of statistic groups along the token axis */
    int32_t tokens_per_group;
    int32_t num_groups;        /* channel groups (32) */
    int32_t silu;
    int32_t dtype;
    float eps;
} AaGroupNorm;

size_t aa_groupnorm_workspace(const AaGroupNorm* d);
/* 1: always the two-kernel form (statistics, then normalise); 0 (default): one kernel that keeps x in registers between the
 * two wherever the shape allows.  Per calling thread; for tests and A/B timing. */
void aa_set_groupnorm_two_pass(int on);
/* 1 and info = {channel groups per workgroup, 16-byte pieces per thread, workgroups per (image group, channel block), grid size}
 * if this call runs as one kernel, 0 if it runs as the statistics + normalise pair. */
int aa_groupnorm_plan(const AaGroupNorm* d, int32_t info[4]);
int aa_groupnorm(const AaGroupNorm* d, void* workspace, size_t workspace_bytes, void* stream);
/* aa_groupnorm_coef (version 109): the statistics of aa_groupnorm without the normalisation pass: coef = fp32 [n_groups_img][2][c0+c1], the
 * per-channel (scale, shift) = (gamma / sigma_g, beta - mu_g gamma / sigma_g) of every image group, for a consumer that applies them to rows it
 * holds anyway (AaLinearRows.row_affine: GroupNorm -> proj_in of diffusers Transformer2DModel / TransformerTemporalModel, an affine-only norm;
 * reference models/unet_3d_blocks.py:287,446,681 / :379,526,759).  d->y is not used, d->silu must be 0; workspace as for aa_groupnorm. */
int aa_groupnorm_coef(const AaGroupNorm* d, void* workspace, size_t workspace_bytes, float* coef, void* stream);

/* aa_layernorm: nn.LayerNorm(C, eps) over the last dim of [rows][C]
 * (diffusers BasicTransformerBlock.norm1/2/3). */
int aa_layernorm(const void* x, const void* gamma, const void* beta, void* y,
                 int64_t rows, int32_t channels, float eps, int32_t dtype, void* stream);

/* ----------------------------------------------------------------------------------------------
 * aa_attention: O = softmax(Q K^T * scale) V per (sequence, head), head_dim 64 (matrix cores) or 8 / 80 (vector
 * ALUs: the 32-head x 8-channel attention of the layerdiffuse alpha decoder, reference
 * models/layerdiffuse_VAE.py:58; the 16-head x 80-channel CLIP ViT-H/14 vision tower of the SVD path, once per clip),
 * flash-style (scores never leave the chip).  Replaces F.scaled_dot_product_attention as selected by the
 * reference's AttnProcessor2_0 (train.py:124-138).
 * A sequence is addressed as (outer o, inner i); the row of position p in operand X is
 *     (o / X.outer_div) * X.outer_stride + i * X.inner_stride + p * X.pos_stride      [tokens]
 * and element (head h, d) sits at column X.col0 + h*head_dim + d of a row of X.ld elements:
 *   spatial self-attention   outer = image, inner = 1,      pos_stride = 1
 *   temporal self-attention  outer = clip,  inner = pixel,  pos_stride = H*W  (no permute copies)
 *   text cross-attention     K/V outer_div = frames per clip (text K/V computed once per clip)
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaAttnOperand {
    const void* ptr;
    int64_t outer_stride, inner_stride, pos_stride;   /* in rows (tokens) */
    int32_t ld, col0, outer_div;
    int32_t seq_mod;   /* 0: the addressing above.  > 0 (version 102): the operand is a table of seq_mod sequences, outer_stride rows
                          apart, and sequence number o * n_inner + i reads entry (number % seq_mod) - the context addressing of
                          diffusers==0.24.0 TransformerSpatioTemporalModel (time_context broadcast as [h*w, batch]) */
} AaAttnOperand;

typedef struct AaAttention {
    AaAttnOperand q, k, v, o;   /* o.ptr is written */
    int32_t n_outer, n_inner, heads, head_dim;
    int32_t q_len, kv_len;
    int32_t dtype;
    float scale;
    int32_t causal;    /* 1: key position > query position is masked (the CLIP text encoder's causal attention, transformers
                          CLIPTextModel - reference train.py:88); head_dim 64 only.  0: none (every attention of the UNets) */
    int32_t _pad;
} AaAttention;

int aa_attention(const AaAttention* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * aa_seq_self_attention (version 108): the front half of a self-attention sub-block over SHORT sequences as ONE kernel -
 *     O = softmax(q k^T * scale) v      with      [q | k | v] = LayerNorm(x) [Wq | Wk | Wv]^T,      head_dim 64
 * i.e. diffusers BasicTransformerBlock.norm1 / norm2 -> Attention.to_q / to_k / to_v -> F.scaled_dot_product_attention as
 * TransformerTemporalModel runs them over the frames of one pixel (reference models/unet_3d_blocks.py:379,526,759,
 * models/unet_3d_condition_mask.py:433-437; double_self_attention: both attention layers of the block).  Q, K and V are never
 * written: the caller follows up with the output projection (aa_conv_gemm on `o`, + bias + residual).
 * A sequence is (outer o, inner i) as in aa_attention: position p of sequence number n = o * n_inner + i is row
 *     o * outer_stride + i * inner_stride + p * pos_stride
 * of x ([rows][ldx]) and of o ([rows][ldo], columns [0, channels)); seq_len <= 32.
 * `w` is packed by the host: [heads][3][64][channels] storage dtype - per head the 64 to_k rows, the 64 to_v rows in the order
 * v(i) = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) for packed row i of each 32-row half, then the 64 to_q rows; with `normalize` the
 * LayerNorm's gamma is folded in (W' = W diag(gamma), rounded to the storage type) and its beta arrives as `w_bias` = W beta (fp32, the
 * same packed row order; none of the three projections has a bias of its own in the reference): the kernel itself only normalises
 * the rows, x~ = (x - mean) / sqrt(var + ln_eps) - the re-association aa_conv_gemm's folded LayerNorm uses (ln_cols).
 * Supported (aa_seq_self_attention_ok): channels 320, 512 or 640 (what one wave can hold of x in registers), heads * 64 == channels.
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaSeqSelfAttn {
    const void* x;          /* [rows][ldx] storage dtype */
    const void* w;          /* [channels / 64][3][64][channels] storage dtype */
    const float* w_bias;    /* [channels / 64][3][64] fp32: what the projections start from (W beta of a folded LayerNorm), or NULL */
    /* A linear layer IN FRONT, inside the same kernel (NULL pre_w: none): x' = x W_pre^T + pre_bias (+ pre_residual) is computed per tile,
     * rounded to the storage type, written to pre_out, and the attention runs over (the normalised) x' - diffusers
     * TransformerTemporalModel.proj_in in front of the first attention layer, or the first layer's to_out[0] + residual in front of the
     * second (reference unet_3d_blocks.py:379 via BasicTransformerBlock).  pre_w: [channels / 64][64][channels] storage dtype, the rows of
     * each 32-row half in the order 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); pre_bias: fp32 [channels] in the same order. */
    const void* pre_w;
    const float* pre_bias;      /* may be NULL */
    const void* pre_residual;   /* [rows][ld_res] storage dtype, rows addressed like x; may be NULL */
    void* pre_out;              /* [rows][ld_pre] storage dtype; required with pre_w */
    void* o;                /* [rows][ldo] storage dtype */
    int64_t outer_stride, inner_stride, pos_stride;   /* in rows */
    int64_t x_bytes, o_bytes;   /* extents of x and o the call may touch (each < 2 GiB: 32-bit offsets through buffer descriptors) */
    int32_t n_outer, n_inner, seq_len;
    int32_t channels;
    int32_t ldx, ldo;       /* row pitches in elements, multiples of 8 */
    int32_t ld_res, ld_pre; /* row pitches of pre_residual / pre_out (multiples of 8; extents like x's: < 2 GiB) */
    int32_t normalize;      /* 1: rows of x are normalised to zero mean / unit variance (ln_eps) in front of the projections; 0: x as it is */
    float ln_eps;
    float scale;            /* head_dim ** -0.5 */
    int32_t dtype;          /* AA_F16 | AA_BF16 */
    int32_t flags;          /* 0 in production; timing ablations (results are garbage): 1 no attention phase, 2 no weight DMA behind the first
                               two stages, 4 no projection MFMAs, 8 no x fetch / normalisation, 16 no output stores, 32 no normalisation */
} AaSeqSelfAttn;

/* 1 if aa_seq_self_attention carries out this call, 0 if the caller has to take the three-launch form (aa_conv_gemm on the
 * concatenated Q|K|V rows, aa_attention) - looks at channels, seq_len, strides and extents only. */
int aa_seq_self_attention_ok(const AaSeqSelfAttn* d);
int aa_seq_self_attention(const AaSeqSelfAttn* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * aa_ff_fused (version 108): the FeedForward of a transformer block and the transformer wrapper's proj_out behind it as ONE kernel -
 *     out = [ GEGLU(LayerNorm(x) W1^T + b1) | x ] Wm^T + bm + outer,        Wm = [Wp W2 | Wp],  bm = Wp b2 + bp
 * diffusers BasicTransformerBlock.norm3 -> FeedForward(GEGLU, Linear) -> + x, then Transformer2DModel / TransformerTemporalModel.proj_out
 * -> + the transformer's input `outer` (reference models/unet_3d_blocks.py:287,446,681 and :379,526,759).  The GEGLU activation
 * [rows][4 * channels] is never written.  channels == 320 (aa_ff_fused_ok); elsewhere the caller runs the two aa_conv_gemm calls.
 * `w` is the weight STREAM the kernel copies into its LDS ring stage by stage (ops.pack_ff_fused builds it; csrc/kernels/ff_fused.h consumes
 * it): 65 stage images of 41 KB (AA_FF_STAGE_BYTES) in the order  X0..X4, A(0), B(0), { A(g), H(g - 1), B(g) } g = 1..19, H(19):
 *   X(i)   rows 64 i .. + 63 of Wm_x = Wp (every 32-row block in the order 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)), all 320 K;
 *   A / B(g) the 64 rows of W1 chunk 2 g / 2 g + 1: the value rows of hidden units 32 ch .. + 31, then their gate rows (GEGLU.proj with
 *          LayerNorm's gamma folded in), all 320 K;
 *   H(g)   Wm_h = Wp W2: all 320 rows (in the row order of X), the 64 hidden units 64 g .. + 63 as K in the order
 *          position 32 cc + 16 h + 8 t + e  <-  unit 32 cc + (e & 3) + 4 h + 8 (e >> 2) + 16 t;
 * an image = five [64][64] chunks (K chunks of the 64 rows for X / A / B, 64-row chunks of the 64 K for H), every row 128 bytes whose 16-byte
 * slots s hold source slot s ^ ((row >> 1) & 7), followed by 1 KB of bias rows [64][8] = (hi, lo, 0, 0, 0, 0, 0, 0) per weight row with
 * hi + lo = the row's fp32 bias (bm = Wp b2 + bp for X; GEGLU bias + W1 beta for A / B; zeros for H) - the kernel adds them on the matrix pipe.
 * ---------------------------------------------------------------------------------------------- */
#define AA_FF_STAGE_BYTES 41984
#define AA_FF_STAGES 65
typedef struct AaFFFused {
    const void* x;          /* [rows][ldx] storage dtype: the block's residual stream (un-normalised) */
    const void* outer;      /* [rows][ld_outer] storage dtype: the transformer's input (the wrapper's residual); may be NULL */
    void* out;              /* [rows][ldo] */
    const void* w;          /* AA_FF_STAGES x AA_FF_STAGE_BYTES: the weight stream (see above), storage dtype */
    int64_t rows;
    int32_t channels;
    int32_t ldx, ld_outer, ldo;    /* row pitches in elements, multiples of 8; every operand < 2 GiB */
    int32_t normalize;      /* 1: rows of x are normalised (ln_eps) in front of W1 (gamma / beta folded into the stream) */
    float ln_eps;
    int32_t dtype;          /* AA_F16 | AA_BF16 */
    int32_t flags;          /* 0 (timing ablations of scripts/bench_ff_fused.py: see csrc/kernels/ff_fused.h) */
} AaFFFused;

int aa_ff_fused_ok(const AaFFFused* d);
int aa_ff_fused(const AaFFFused* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * aa_linear_rows (version 109): out = [LayerNorm](x) W^T + b (+ residual) over 320-channel token rows held in registers - the K = C projections
 * of the 320-channel transformers (diffusers Transformer2DModel / TransformerTemporalModel.proj_in, Attention.to_q / to_out[0], the fused
 * to_q | to_k | to_v of a spatial self-attention; reference models/unet_3d_blocks.py:287,446,681 / :379,526,759).  channels == 320, n_out a
 * multiple of 32 (aa_linear_rows_ok); elsewhere the caller uses aa_conv_gemm.  `w` is a weight STREAM like aa_ff_fused's: n_out / 32 stage
 * images of AA_LR_STAGE_BYTES; image s = the 32 weight rows of output channels 32 s + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) (i = 0..31;
 * LayerNorm's gamma folded in), all 320 K as five [32][64] chunks whose 128-byte rows hold 16-byte source slot s' ^ ((row >> 1) & 7) at slot s',
 * then [32][8] = (hi, lo, 0 x 6) of each row's fp32 bias (+ W beta) and 512 zero bytes (ops.pack_linear_rows builds it).
 * ---------------------------------------------------------------------------------------------- */
#define AA_LR_STAGE_BYTES 21504
typedef struct AaLinearRows {
    const void* x;          /* [rows][ldx] storage dtype */
    const void* residual;   /* [rows][ld_res] storage dtype, added to the result; may be NULL */
    void* out;              /* [rows][ldo] */
    const void* w;          /* (n_out / 32) x AA_LR_STAGE_BYTES */
    int64_t rows;
    int32_t channels, n_out;
    int32_t ldx, ld_res, ldo;      /* row pitches in elements, multiples of 8; every operand < 2 GiB */
    int32_t normalize;      /* 1: rows of x are normalised (ln_eps) in front of W (gamma / beta folded into the stream) */
    float ln_eps;
    int32_t dtype;          /* AA_F16 | AA_BF16 */
    int32_t flags;          /* 0; debug bits: 1 = plan the launch for a 2-CU chip (tests), 2 = never split the tiles of the last round over their stages */
    const float* row_affine;   /* NULL, or aa_groupnorm_coef's [groups][2][channels]: row r is replaced by x * scale[g] + shift[g], g = r / rows_per_group,
                                  rounded to the storage type (what aa_groupnorm would have written), in front of `normalize` and W */
    int32_t rows_per_group;    /* with row_affine: a multiple of 32 */
    int32_t _pad;
} AaLinearRows;

int aa_linear_rows_ok(const AaLinearRows* d);
int aa_linear_rows(const AaLinearRows* d, void* stream);

/* aa_softmax_rows: y[r, :] = softmax(x[r, :]) for fp32 scores (VAE mid-block single-head
 * attention, head_dim 512, where scores are materialised: diffusers Attention with
 * upcast_softmax). x fp32 [rows][x_ld], y storage dtype [rows][y_ld]; only the first `cols` columns
 * of each row are read / written. */
int aa_softmax_rows(const float* x, void* y, int64_t rows, int32_t cols, int32_t x_ld, int32_t y_ld,
                    int32_t dtype, void* stream);

/* ----------------------------------------------------------------------------------------------
 * aa_cfg_dpm_step: fused classifier-free guidance + DPM-Solver++(2M) update of one denoising
 * step (reference models/pipeline.py:179-192 + diffusers DPMSolverMultistepScheduler.step):
 *   eps = eps_uncond + g * (eps_text - eps_uncond);  x0 = (x - sigma_s*eps) / alpha_s
 *   x' = c_x * x - c_d0 * x0 - c_d1 * (x0 - x0_prev);  x0_prev <- x0
 * all tensors fp32 or storage dtype element streams of `n` elements.
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaDpmStep {
    const void* eps_uncond;   /* storage dtype [n] */
    const void* eps_text;     /* storage dtype [n] (== eps_uncond when guidance is off) */
    void* latents;            /* fp32 [n], updated in place */
    void* x0_prev;            /* fp32 [n], updated in place */
    void* latents_lp;         /* storage dtype copy of the new latents [n] (next UNet input) */
    int64_t n;
    float guidance, sigma_s, alpha_s, c_x, c_d0, c_d1;
    int32_t dtype;
} AaDpmStep;

int aa_cfg_dpm_step(const AaDpmStep* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Step glue (version 101): with these three the whole denoising step is library launches only -
 * [timestep embedding + input packing + UNet] captured in one hipGraph, then the guidance / solver update.
 *
 * aa_timestep_embedding: diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)
 * (reference unet_3d_condition_mask.py:408-416): out[r, :] = [cos(t[r]*f_k) | sin(t[r]*f_k)],
 * f_k = exp(-ln(10000) * k / (dim/2)); `t` is a DEVICE array of n fp32 values (so a captured graph can
 * be replayed with a new timestep), out is [n][dim] storage dtype.
 * ---------------------------------------------------------------------------------------------- */
int aa_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, int32_t dtype, void* stream);

/* aa_pack_latents: the UNet's input assembly (reference unet_3d_condition_mask.py:376,424-428):
 * sample [sample_batch][channels][frames][hw] (fp32 or storage dtype) gets the condition latent
 * [cond_batch][channels][1][hw] prepended as frame 0 and, when `mask` is given, the mask
 * [mask_batch][1][1][hw] prepended as channel 0 of every frame; the result is written as the channels-last
 * token matrix [batch][frames+1][hw][8] (zero padded to 8 channels) that conv_in / conv_in2 read.
 * Batch element b reads sample b % sample_batch, cond b % cond_batch, mask b % mask_batch (classifier-free
 * guidance runs the same latents twice: models/pipeline.py:165). */
typedef struct AaPackLatents {
    const void* sample;
    const void* cond;
    const void* mask;      /* NULL: no mask channel */
    void* out;             /* [batch*(frames+1)*hw][8] storage dtype */
    int32_t batch, sample_batch, cond_batch, mask_batch;
    int32_t channels;      /* latent channels (4); channels + (mask ? 1 : 0) <= 8 */
    int32_t frames;        /* frames of `sample` (the output has frames + 1) */
    int32_t hw;
    int32_t dtype;         /* storage dtype of cond / mask / out */
    int32_t sample_dtype;  /* AA_F32 or == dtype */
} AaPackLatents;

int aa_pack_latents(const AaPackLatents* d, void* stream);

/* aa_cfg_dpm_step_tokens: aa_cfg_dpm_step reading the UNet output where the UNet left it - the token matrix
 * [2*clips or clips][frames+1][hw][eps_ld] with frame 0 (the condition frame) skipped (reference
 * unet_3d_condition_mask.py:522) and, under guidance, the unconditional clips first (models/pipeline.py:165,181-183).
 * Updates the fp32 latents / x0_prev [clips][channels][frames][hw] in place, optionally writes their storage-dtype
 * copy (the next UNet `sample`) and the next timestep value(s) for aa_timestep_embedding. */
typedef struct AaDpmStepTok {
    const void* eps_tokens;
    void* latents;            /* fp32, updated in place */
    void* x0_prev;            /* fp32, updated in place */
    void* latents_lp;         /* storage dtype copy of the new latents, may be NULL */
    float* next_t;            /* device array receiving `next_t_value` (next_t_count <= 256 entries), may be NULL */
    int32_t next_t_count;
    float next_t_value;
    int32_t clips, channels, frames, hw;
    int32_t eps_ld;           /* row pitch of the token matrix in elements (>= channels) */
    int32_t guidance_on;
    float guidance, sigma_s, alpha_s, c_x, c_d0, c_d1;
    int32_t dtype;
} AaDpmStepTok;

int aa_cfg_dpm_step_tokens(const AaDpmStepTok* d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Stable-Video-Diffusion path (version 102; reference models/pipeline.py:223-731, train_svd.py:726-826 driving diffusers'
 * UNetSpatioTemporalConditionModel / AutoencoderKLTemporalDecoder / EulerDiscreteScheduler).
 *
 * aa_blend: out[r, :] = act( a * x[r, :] + b * y[r, :] + rowvec[(r / rowvec_div) % rowvec_mod, :] ) on [rows][channels]
 * token matrices (channels % 8 == 0, contiguous rows); y and rowvec may be NULL, rowvec_mod 0 = no wrap.  Covers
 *   diffusers AlphaBlender                   alpha * x_spatial + (1 - alpha) * x_temporal (TransformerSpatioTemporalModel),
 *   `hidden_states_mix + emb`                the per-frame position embedding in front of the temporal transformer block
 *                                            (rowvec = [frames][C], rowvec_div = h*w, rowvec_mod = frames),
 *   silu(emb + aug_emb)                      the summed time / added-time-ids embedding in front of the resnets' projections.
 * ---------------------------------------------------------------------------------------------- */
typedef struct AaBlend {
    const void* x;
    const void* y;         /* may be NULL */
    const void* rowvec;    /* may be NULL */
    void* out;             /* may alias x */
    int64_t rows;
    int32_t channels;
    int32_t rowvec_div, rowvec_mod, rowvec_ld;   /* rowvec_ld: row pitch in elements (0 = channels) */
    float a, b;
    int32_t act;           /* AA_ACT_* */
    int32_t dtype;
} AaBlend;

int aa_blend(const AaBlend* d, void* stream);

/* aa_pack_frames: the SVD UNet's input assembly (reference models/pipeline.py:417-422, :683-691):
 *   latent_model_input = scheduler.scale_model_input(cat([latents] * 2), t)          (* 1 / sqrt(sigma^2 + 1))
 *   cat([mask, latent_model_input, image_latents], dim=2)                            [B, F, 9, h, w]
 * written as the channels-last token matrix [batch*frames*hw][out_channels] (zero padded) that conv_in reads.  Up to three
 * sources [src_batch][frames][src_channels][hw] are concatenated along the channel axis; batch element b reads
 * b % src_batch of each (the guidance halves share latents and mask).  Source `scaled_src` (or none: -1) is multiplied
 * by *scale, a DEVICE float (so a captured graph can be replayed with the next sigma). */
typedef struct AaPackFrames {
    const void* src[3];        /* unused entries NULL */
    int32_t src_channels[3];
    int32_t src_batch[3];
    int32_t src_f32[3];        /* 1: fp32 source, 0: storage dtype */
    const float* scale;        /* may be NULL (= 1) */
    int32_t scaled_src;
    void* out;
    int32_t batch, frames, hw;
    int32_t out_channels;      /* 8 or 16, >= sum of src_channels */
    int32_t dtype;
} AaPackFrames;

int aa_pack_frames(const AaPackFrames* d, void* stream);

/* aa_cfg_euler_step_tokens: per-frame classifier-free guidance + EulerDiscreteScheduler.step (v-prediction) reading the
 * UNet's token-layout output [(2*)clips*frames*hw][ld] (reference models/pipeline.py:435-440, :704-708):
 *   v = v_uncond + guidance[frame] * (v_cond - v_uncond)            (guidance NULL: clips rows only, no guidance)
 *   x0 = v * (-sigma / sqrt(sigma^2+1)) + x / (sigma^2+1);  x' = x + (x - x0) / sigma * (sigma_next - sigma)
 * folded by the caller into x' = c_x * x + c_v * v.  Updates the fp32 latents [clips][frames][channels][hw] in place and
 * writes the next step's timestep(s) / input scale for aa_timestep_embedding / aa_pack_frames. */
typedef struct AaEulerStepTok {
    const void* v_tokens;
    void* latents;            /* fp32, updated in place */
    const float* guidance;    /* DEVICE [frames] or NULL */
    float* next_t;            /* device array receiving next_t_value (next_t_count <= 256 entries), may be NULL */
    int32_t next_t_count;
    float next_t_value;
    float* next_scale;        /* device float receiving next_scale_value, may be NULL */
    float next_scale_value;
    int32_t clips, channels, frames, hw;
    int32_t ld;               /* row pitch of the token matrix in elements (>= channels) */
    float c_x, c_v;
    int32_t dtype;
} AaEulerStepTok;

int aa_cfg_euler_step_tokens(const AaEulerStepTok* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AA_MI355_H */
