/* mv2d_hip.h — C ABI of libmv2d_hip.so: the MI355X (gfx950) kernels behind MV2D's sparse cross-attention
 * decoder hot path.
 *
 * The reference (tusen-ai/MV2D) is pure Python and has NO native boundary of its own; its hot path runs through
 * third-party native ops (torch ATen, mmcv RoIAlign).  This header is therefore the boundary a maintainer binds
 * (ctypes) in place of those calls; every entry point cites the reference code it replaces
 * (paths relative to the reference root; RH = mmdet3d_plugin/models/roi_heads, MU = mmdet3d_plugin/models/utils,
 * CB = mmdet3d_plugin/core/bbox).
 *
 * Conventions
 *  - plain C types only: device pointers, sizes, strides; no torch types.
 *  - every function returns 0 on success, <0 on error (-1 bad argument, -2 launch failure);
 *    mv2d_last_error() returns a thread-local description.  No exceptions cross the ABI.
 *  - all buffers (inputs, outputs, workspaces) are allocated and freed by the caller; kernels never allocate.
 *  - asynchronous: work is only enqueued on `stream` (a hipStream_t; pass torch.cuda.current_stream().cuda_stream);
 *    no implicit device synchronisation; re-entrant.
 *  - "bf16" buffers are raw uint16 (upper half of an IEEE fp32, round-to-nearest-even).
 *  - all matrices are row-major; "position-major map" = [V*h*w, 256] with the channel index fastest.
 */
#ifndef MV2D_HIP_H
#define MV2D_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* mv2d_last_error(void);
int mv2d_abi_version(void);
/* The 16-bit storage / MFMA operand format of the KEY SIDE ("key16": gathered key / value rows, RoI cells, PE-MLP operands + hidden layer and
 * the weights those kernels read; csrc/common.h): 1 = IEEE fp16 (round 4: 11 significand bits, conversions saturate at +-65504), 0 = bf16 (a
 * -DMV2D_KEY16_BF16 build).  Every `void*` below that is documented as key16 holds that format; the split-precision kernels ("bf16x3" / "x3" in the comments
 * below, written in rounds 1-4) carry their operands in the q16 format -- IEEE fp16 pairs since round 5, see mv2d_q16_format(); the generic tile GEMM
 * mv2d_gemm_bf16 and the training route's mv2d_gemm_f32x3 are bf16 in either build. */
int mv2d_key16_format(void);
/* fp32 -> key16: hi [n] = key16(x) and, when lo != NULL, lo [n] = key16(x - hi) (x ~ hi + lo: static weights of the key-side kernels) */
int mv2d_f32_to_key16(const float* x, void* hi, void* lo, long long n, void* stream);
int mv2d_device_arch(char* buf, int buflen);
/* calibration kernel of the stream planner: one wave spinning for usec microseconds on `stream` (no reference counterpart) */
int mv2d_spin(int usec, void* stream);

/* ---- dense contractions -------------------------------------------------------------------------------- */

/* bf16 MFMA GEMM  C = epi(A[M,K] . W[N,K]^T + bias)  (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
 * Replaces: PE 1x1-conv MLPs + SE gate (MU/pe.py:64-77,36-48,158-166), key/value in_proj of all decoder layers
 * (torch.nn.MultiheadAttention inside MU/petr_transformer.py:503-508), QueryGenerator shared 3x3 conv
 * (RH/utils/query_generator.py:298-304,352-358) as an implicit GEMM (a_mode=1: A = RoI features [R,49,256]).
 *  n_split>0: columns >= n_split read A2 (e.g. K columns from feat+pe rows, V columns from feat rows).
 *  m_dev: optional device int; rows >= *m_dev are skipped (row count known only on the device).
 *  epilogue: v = acc + bias; v *= mul[m,n]; v += add[m,n]; act (0 none, 1 relu, 2 sigmoid);
 *            C[(n / c_blk_cols) * c_blk_stride + m * ldc + n % c_blk_cols] = v   (c_blk_cols = 0: C[m*ldc+n]);
 *            C2[m*ldc2+n] = bf16(v + add2[m,n]). */
int mv2d_gemm_bf16(const void* A, const void* A2, int n_split, int a_mode, const void* W, const float* bias, int M, int N,
                   int K, int lda, const int* m_dev, int act, const float* mul, int ldmul, const float* add, int ldadd,
                   void* C, int c_bf16, int ldc, long long c_blk_stride, int c_blk_cols, void* C2, const float* add2,
                   int ldc2, int ldadd2, void* stream);
/* mv2d_gemm_bf16 with two additions for the engine's index-exact route, which runs fp32-class products through the plain bf16 GEMM by
 * K-concatenation: [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]^T = a_hi w_hi + a_lo w_hi + a_hi w_lo (K' = 3 K, one fp32 accumulation).
 * c_split3 = 1: the (bf16) output is written as that operand for the next GEMM: [hi | lo | hi] in column blocks of N (ldc >= 3 N).
 * Conv mode (a_mode 1): lda = channels per RoI cell (256, or 768 for [hi | lo | hi] cells), K = 9 * lda with k = (tap, channel). */
int mv2d_gemm_bf16_ex(const void* A, const void* A2, int n_split, int a_mode, const void* W, const float* bias, int M, int N, int K, int lda,
                      const int* m_dev, int act, const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16, int ldc,
                      long long c_blk_stride, int c_blk_cols, void* C2, const float* add2, int ldc2, int ldadd2, int c_split3, const int* add_idx /* optional:
                      the add operand's row for output row m is add_idx[m] % add_period (a gathered table) */, int add_period,
                      int k_splits /* >= 1; > 1: split-K, split s writes the plain fp32 partial product of its K range to C + s * c_split_stride
                      (bias in slab 0; no activation / fused operands): sum the slabs with mv2d_colsum */, long long c_split_stride, void* stream);
/* out [M, 3 cols] bf16 = [hi | lo | hi] of a (+ b, optional) fp32 [M, cols]: hi = bf16(x), lo = bf16(x - hi); rows >= *m_dev (optional)
 * are not written. */
int mv2d_split3_rows(const float* a, const float* b, void* out, int M, int cols, const int* m_dev, void* stream);

/* exact-fp32 MFMA GEMM for the per-query (M = #queries) ops  C = epi((A . W^T + bias) * scale).
 * Replaces nn.Linear calls of: query/out in_proj/out_proj and FFN (MU/petr_transformer.py:358-363,503-508; mmcv FFN),
 * query_embedding / cls / reg branches (RH/bbox_heads/cross_attention_head.py:118-146,199-227),
 * QueryGenerator fcs (RH/utils/query_generator.py:360-373,404).
 *  split_k>1 writes split_k partial slabs (slab z at C + z*c_slice_stride elements); bias is added in slab 0.
 *  clamp>0: v = min(max(v,-clamp),clamp) (RH/utils/query_generator.py:369).
 *  groups>1 (needs split_k==1): grouped GEMM, group g reads A + g*a_gs, W + g*w_gs, bias + g*b_gs, writes C + g*c_gs
 *  (the 6 per-layer cls/reg branches of cross_attention_head.py:216-227 in one launch). */
int mv2d_gemm_f32(const float* A, const float* A2, int n_split, const float* W, const float* bias, int M, int N, int K,
                  int lda, int ldw, int split_k, int act, float scale, float clamp, void* C, int c_bf16, int ldc,
                  long long c_slice_stride, int groups, long long a_gs, long long w_gs, long long b_gs, long long c_gs,
                  void* stream);

/* Row-block fused attention tail: x_out = LayerNorm(ctx . Wo^T + bo + resid); optional second stage
 * q_out = ((x_out + qpos) . Wq^T + bq) * qscale.  Replaces out_proj + identity add of FlattenMHSelfAttention /
 * PETRMultiheadAttention (MU/petr_transformer.py:358-370, 503-513), the following 'norm' of the mmcv layer and the query
 * in_proj of the next cross attention.  All [M,256] fp32, weights [256,256] fp32 (nn.Linear layout). */
int mv2d_attn_out_fused(const float* ctx, const float* resid, const float* Wo, const float* bo, const float* ln_w, const float* ln_b,
                        float* x_out, const float* qpos, const float* Wq, const float* bq, float qscale, float* q_out, int M, float eps,
                        void* stream);

/* The position-encoding block of the PE module as one launch (MU/pe.py:36-48,64-77,150-166; csrc/pe_tab96.hip):
 *   pe = sine_tab[position] + position_encoder(A1) * sigmoid(conv_expand(relu(conv_reduce(Xf)))),  Xk = key16(pe + Xf32)
 * The adapt_pos3d(sine) branch depends only on the weights and on the padding geometry, so it is read from sine_tab [tab_period][256] fp32 =
 * adapt_pos3d(sine)(position) + b2b, indexed by the key's map position (row_index[m], or m) modulo tab_period (= positions of one sample when
 * the samples of a batch share their geometry).  A1 [M,192], Xfb [M,256] key16 rows (mv2d_pe_inputs); Xf32 = fp32 feature rows [M,256] or,
 * with row_index != NULL, the position-major feature map itself (row_index [M] = the map row of every key); the four weights key16 in the
 * FRAGMENT-MAJOR order of mv2d_pack_wfrag_bf16 (W1a [1024,192], W1b [256,1024], Wr / We [256,256]); m_dev: optional device-side row count.
 * Xk may be NULL (then Xf32 is not read either): the S path's keys are RoI-aligned rows, it only needs pe [M,256] fp32; pe may be NULL when
 * Xk [M,256] key16 is given: on the T path nothing reads pe.  mv2d_pe_fused_tab2: the same with the block shape exposed (1 = 96 rows x 8 waves,
 * the default; 0 = 64 rows x 4 waves, two blocks per CU; bit-identical). */
int mv2d_pe_fused_tab(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                      const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                      const void* Wr, const float* br, const void* We, const float* be,
                      const float* sine_tab, int tab_period, float* pe, void* Xk, void* stream);
int mv2d_pe_fused_tab2(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                       const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                       const void* Wr, const float* br, const void* We, const float* be,
                       const float* sine_tab, int tab_period, float* pe, void* Xk, int shape, void* stream);

/* The same block in SPLIT PRECISION on unrounded fp32 inputs (index-exact route; csrc/pe_x3.hip): every product a_hi w_hi + a_lo w_hi + a_hi w_lo on
 * bf16 MFMAs (2^-17 per operand), hidden layer split hi / lo in LDS.  A1 [M,192] fp32 (mv2d_pe_inputs' A_frustum_f32), Xmap = fp32 feature rows
 * [.,256] indexed by row_index[m] (or m when NULL); weights as bf16 hi / lo pairs (mv2d_split_bf16x2), each in the fragment-major order of
 * mv2d_pack_wfrag_bf16; pe [M,256] fp32 (optional) = sine_tab[position] + position_encoder(A1) * gate; Xk_hi / Xk_lo / Xv_hi / Xv_lo [M,256] key16
 * (all four or none): key rows pe + feat and value rows feat as hi + lo pairs (what mv2d_xattn_tile_fwd gathers on that route).
 * lo_fmt (round 6, ABI 6): 0 = the lo outputs are key16 rows [M,256] (512 B); 1 = "lo8" rows [M,256] of BYTES: OCP e4m3 (bias 7, max 448) of
 * key16_lo * 2^12, round-to-nearest-even, saturating (csrc/common.h) -- 256 B per row, what the cross attention gathers by default.
 * pe_at_index (ABI 6): 1 = pe row m is written at row row_index[m] of `pe` -- a position-indexed map that mv2d_roi_align_ex reads without the
 * position -> row table (one dependent load less per bilinear tap); rows no m lists keep what they held (the caller zero-fills the map once).
 * lo8_flag (device int, may be NULL): |= 1 when a remainder leaves the e4m3 range (|x| beyond ~224: that element keeps key16 precision only). */
int mv2d_pe_fused_x3(const float* A1, const float* Xmap, const int* row_index, const int* m_dev, int M,
                     const void* W1a_hi, const void* W1a_lo, const float* b1a, const void* W1b_hi, const void* W1b_lo, const float* b1b,
                     const void* Wr_hi, const void* Wr_lo, const float* br, const void* We_hi, const void* We_lo, const float* be,
                     const float* sine_tab, int tab_period, float* pe, void* Xk_hi, void* Xk_lo, void* Xv_hi, void* Xv_lo, int lo_fmt, int pe_at_index,
                     int* lo8_flag, void* stream);
/* The same block on the second shape of the kernel (round 6, csrc/pe_x3b.hip): a wave owns 16 rows through both layers of each MLP, the hidden layer
 * stays in registers (no LDS image, no barrier between the layers), the weights go through a 4-deep LDS ring (LDS-DMA) shared by the 8 waves of a 128-row
 * block, two waves per SIMD.  Same operands EXCEPT that W1a and Wr (the first layers) are packed from the weight with its rows in the order
 * packed row 32 b + 16 t + 4 g + e = original row 32 b + 8 g + 4 t + e (ops.rowperm32); outputs bitwise those of mv2d_pe_fused_x3. */
int mv2d_pe_fused_x3b(const float* A1, const float* Xmap, const int* row_index, const int* m_dev, int M,
                     const void* W1a_hi, const void* W1a_lo, const float* b1a, const void* W1b_hi, const void* W1b_lo, const float* b1b,
                     const void* Wr_hi, const void* Wr_lo, const float* br, const void* We_hi, const void* We_lo, const float* be,
                     const float* sine_tab, int tab_period, float* pe, void* Xk_hi, void* Xk_lo, void* Xv_hi, void* Xv_lo, int lo_fmt, int pe_at_index,
                     int* lo8_flag, void* stream);

/* QueryGenerator shared conv + pooling fused, one block per RoI (RH/utils/query_generator.py:298-304,322-331,352-358):
 * out[r, n] = mean over the 49 cells of relu(conv3x3(roi_feat[r])[cell, n] + bias[n]).  roi_feat [R,49,256] key16 (cell-major),
 * Wp = the conv weight [256][tap][cin] (key16, K = 2304) in FRAGMENT-MAJOR order as produced by mv2d_pack_wfrag_bf16 (weights are
 * static: one fragment = one contiguous 1 KB load); out [R, ld_out] fp32.  Same k order as mv2d_gemm_bf16(a_mode = 1). */
int mv2d_pack_wfrag_bf16(const void* W, void* Wp, int N, int K, void* stream);   /* Wp[K/32][N/16][64][8] <- W[N][K] */
int mv2d_qg_conv_pool(const void* roi_feat, const void* W, const float* bias, float* out, int ld_out, int R, void* stream);
/* The same in split precision (index-exact route): RoI cells as key16 hi + lo pairs [R,49,256] (mv2d_roi_align_ex), weights as fragment-major
 * key16 hi / lo copies of the [256, 2304] matrix (mv2d_f32_to_key16 + mv2d_pack_wfrag_bf16); products a_lo w_hi + a_hi w_lo + a_hi w_hi. */
int mv2d_qg_conv_pool_x3(const void* roi_feat_hi, const void* roi_feat_lo, const void* W_hi, const void* W_lo, const float* bias, float* out,
                         int ld_out, int R, void* stream);

/* The same chain in split precision (bf16x3, ~1e-5 relative): Wo / Wq as bf16 hi/lo pairs (mv2d_split_bf16x2), each in the
 * fragment-major order of mv2d_pack_wfrag_bf16. */
int mv2d_attn_out_fused_x3(const float* ctx, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo,
                           const float* ln_w, const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi,
                           const void* Wq_lo, const float* bq, float qscale, float* q_out, int M, float eps, void* stream);

/* Self-attention core (all 8 heads of 16 queries against all R keys, exact fp32) + out_proj + residual + LayerNorm + optional
 * cross-attention q projection in one row-fused kernel: mv2d_self_attn_fwd followed by mv2d_attn_out_fused_x3 without the context
 * round trip (FlattenMHSelfAttention + norm + PETRMultiheadAttention q in_proj, MU/petr_transformer.py:317-370, 487-513).
 * qkv [M,768] fp32 = in_proj(q | k | v), q unscaled; the other arguments as in mv2d_attn_out_fused_x3. */
int mv2d_sa_block_fused_x3(const float* qkv, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo,
                           const float* ln_w, const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi,
                           const void* Wq_lo, const float* bq, float qscale, float* q_out, int M, float eps, void* stream);

/* Tail of the query generator + query positional embedding, row-fused: center = fc_center(enc2) (exact fp32), xyz = center2lidar,
 * ref = normalised reference point (no clamp), posemb = pos2posemb3d(ref), qpos = query_embedding(posemb) (Linear-ReLU-Linear in
 * bf16x3; W0 [256,384] and W2 [256,256] as bf16 hi/lo pairs, each fragment-major).  RH/utils/query_generator.py:333-341,404,
 * RH/mv2d_t_head.py:51-57, MU/pe.py:21-33, RH/bbox_heads/cross_attention_head.py:118-125.  pc_range: host array of 6. */
int mv2d_query_embed_fused_x3(const float* enc2, const float* Wc, const float* bc, const float* minv, const float* dim_t,
                              const float* pc_range, const void* W0_hi, const void* W0_lo, const float* b0, const void* W2_hi,
                              const void* W2_lo, const float* b2, float* center, float* xyz, float* ref, float* posemb,
                              float* qpos, int R, void* stream);

/* FFN tail + the next layer's self-attention in_proj, row-fused: y = LN(sum_z parts[z] + b2 + resid); x_out = y; xq_out = y + qpos (may be NULL);
 * outs = post_norm(y) (optional); qkv [M,768] = [xq.Wq^T + bq | xq.Wk^T + bk | y.Wv^T + bv] in bf16x3 split precision (optional:
 * Win_hi = null for the last layer).  Win_hi / Win_lo: nn.MultiheadAttention in_proj_weight [768,256] as a bf16 hi/lo pair
 * (mv2d_split_bf16x2), each fragment-major (mv2d_pack_wfrag_bf16).  Replaces mv2d_row_ln + mv2d_gemm_f32 between two layers
 * (mmcv FFN identity + norm, decoder post_norm, FlattenMHSelfAttention in_proj: MU/petr_transformer.py:269-311, 346-363, 563-565). */
int mv2d_ffn_out_fused_x3(const float* parts, int n_parts, long long part_stride, const float* b2, const float* resid,
                          const float* ln_w, const float* ln_b, const float* post_w, const float* post_b, float* x_out,
                          const float* qpos, float* xq_out, float* outs, const void* Win_hi, const void* Win_lo,
                          const float* b_in, float* qkv, int M, float eps, void* stream);

/* All per-layer prediction branches in one launch (RH/bbox_heads/cross_attention_head.py:127-146, 216-238; velocity / dt of
 * RH/mv2d_t_head.py:136-140).  outs [L,M,256]; cls_w = {w0,b0,ln1w,ln1b,w3,b3,ln4w,ln4b,w6,b6}, reg_w = {w0,b0,w2,b2,w4,b4}: HOST
 * arrays of device pointers, every tensor stacked over the L layers; ref [M,3]; out cls, reg [L,M,10] (reg final: sigmoid / ref /
 * pc_range / dt applied).  The four 256x256 matrices per layer (cls w0, w3; reg w0, w2) are passed FRAGMENT-MAJOR: the stacked [L*256, 256]
 * tensor run through mv2d_pack_wfrag_f32 (static weights: one contiguous 1 KB per fragment load); the 10x256 output layers stay row-major. */
int mv2d_pack_wfrag_f32(const float* W, float* Wp, int N, int K, int ldw, void* stream);   /* Wp[ceil(N/16)][K/16][64][4] <- W[N][ldw] */
int mv2d_heads_fused(const float* outs, const float* const* cls_w, const float* const* reg_w, const float* ref, float* cls, float* reg,
                     int M, int L, float eps, const float* pc_range, float dt, const float* dt_rows /* optional [M]: per-row dt of a
                     batch of samples, overrides dt */, void* stream);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias) in split precision (bf16x3, ~1e-5 relative) for the per-query MLPs (QueryGenerator fcs
 * RH/utils/query_generator.py:359-381, first self-attention in_proj): LDS-tiled (A chunk shared by 8 column tiles, fragment-major
 * weights Whi / Wlo = mv2d_split_bf16x2 + mv2d_pack_wfrag_bf16 of W), act 1 = ReLU, clamp > 0 clamps to [-clamp, clamp];
 * columns >= n_split (multiple of 128) read A2 instead of A.  groups > 1: a batch of independent linears of the same shape in one
 * launch, group g at A + g*a_gs, W + g*w_gs, bias + g*b_gs, C + g*c_gs (element strides). */
int mv2d_linear_x3(const float* A, const float* A2, int n_split, int lda, const void* Whi, const void* Wlo, const float* bias,
                   float* C, int ldc, int M, int N, int K, int act, float clamp, int groups, long long a_gs, long long w_gs,
                   long long b_gs, long long c_gs, void* stream);
/* mv2d_linear_x3 with the operands the engine's index-exact route needs (MU/pe.py:36-48,64-77,150-166 and the QueryGenerator conv,
 * RH/utils/query_generator.py:298-304, all in fp32-class arithmetic without a host synchronisation): m_dev = device-side row count
 * (rows >= *m_dev are skipped; NULL: M), conv3x3 = 1: A is [R,49,256] RoI cells and C = conv3x3(A) as an implicit GEMM (K = 2304 =
 * [tap][cin], zero padding, M = 49 R), act 2 = sigmoid, then C = C * mul + add with optional fp32 operands [M, ld_ma]. */
int mv2d_linear_x3_ex(const float* A, const float* A2, int n_split, int lda, const void* Whi, const void* Wlo, const float* bias,
                      float* C, int ldc, int M, int N, int K, int act, float clamp, int groups, long long a_gs, long long w_gs,
                      long long b_gs, long long c_gs, const int* m_dev, int conv3x3, const float* mul, const float* add, int ld_ma,
                      void* stream);
/* (hi, lo) key16 pair of every element of a (+ b, optional): hi = key16(x), lo = key16(x - hi); a, b fp32 [M, cols]; rows >= *m_dev
 * (optional) are not written.  The key / value rows of the index-exact route (key = feat + pe, value = feat). */
int mv2d_split_rows_key16(const float* a, const float* b, void* hi, void* lo, int M, int cols, const int* m_dev, void* stream);

/* The same branches with their four 256x256 linears per (layer, branch) in split precision (bf16x3, ~1e-5 relative; the 256 -> 10
 * output layers stay exact fp32).  cls_w = {w0_hi,w0_lo,b0,ln1w,ln1b,w3_hi,w3_lo,b3,ln4w,ln4b,w6,b6}, reg_w = {w0_hi,w0_lo,b0,w2_hi,
 * w2_lo,b2,w4,b4}: every tensor stacked over the L layers, the *_hi/_lo matrices are per-layer mv2d_split_bf16x2 +
 * mv2d_pack_wfrag_bf16 copies. */
int mv2d_heads_fused_x3(const float* outs, const void* const* cls_w, const void* const* reg_w, const float* ref, float* cls, float* reg,
                        int M, int L, float eps, const float* pc_range, float dt, const float* dt_rows, void* stream);

/* Fused FFN partial sums (mmcv FFN 256 -> hidden -> 256 of the decoder layer, configs/mv2d/exp/*:78-79):
 * slabs[s] = relu(X . W1[64s:64s+64]^T + b1[64s:64s+64]) . W2[:, 64s:64s+64]^T  for the hidden/64 slices s, exact fp32.
 * X [M,256], W1 [hidden,256], W2 [256,hidden], slabs [hidden/64, M, 256]; the caller sums the slabs + b2 + residual
 * (mv2d_row_ln with n_parts = hidden/64) — fixed summation order, deterministic.
 * W1p / W2p: the weights in the fragment-major order produced by mv2d_ffn_pack_weights (static data: every weight load of a wave is one
 * contiguous 1 KB). */
int mv2d_ffn_pack_weights(const float* W1, const float* W2, float* W1p, float* W2p, int hidden, void* stream);
int mv2d_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2, float* slabs, int M, int hidden, void* stream);

/* The same fused FFN in split precision on the bf16 matrix cores ("bf16x3": x = x_hi + x_lo as a bf16 pair, three bf16 MFMAs per
 * product, fp32 accumulation): ~1e-5 relative error instead of bit-exact fp32, 3/16 of the matrix-pipe time.  W1/W2 are given as
 * the pairs produced by mv2d_split_bf16x2 ([hidden,256] and [256,hidden] bf16 each), each stored fragment-major
 * (mv2d_pack_wfrag_bf16); same slab output as mv2d_ffn_fused, except that slices_per_block (1, 2 or 4) consecutive hidden slices are
 * accumulated per block: slabs [hidden/64/slices_per_block, M, 256] (less slab traffic when M is large enough to fill the chip). */
int mv2d_ffn_fused_x3(const float* X, const void* W1hi, const void* W1lo, const float* b1, const void* W2hi, const void* W2lo,
                      float* slabs, int M, int hidden, int slices_per_block, void* stream);

/* Whi = bf16(W), Wlo = bf16(W - Whi): bf16 pairs (the training route's K-concatenated operands, mv2d_gemm_bf16 partners) */
int mv2d_split_bf16x2(const float* x, void* hi, void* lo, long long n, void* stream);
/* The SPLIT format of the query side ("q16", csrc/common.h): every `*_hi / *_lo` weight pair of the split-precision kernels below
 * (documented as "bf16x3" / mv2d_split_bf16x2 in rounds 1-4) is produced by mv2d_split_q16x2 of the SAME library.  mv2d_q16_format():
 * 1 = IEEE fp16 pairs (round 5: 11 + 11 significand bits, 2.7e-7 relative on a 256-term product, saturating at +-65504),
 * 0 = bf16 pairs (8 + 8 bits, 4.5e-6; a -DMV2D_Q16_BF16 build).  Same bytes, same three MFMAs per product. */
int mv2d_q16_format(void);
int mv2d_split_q16x2(const float* x, void* hi, void* lo, long long n, void* stream);

/* ---- row-wise ops on the [M,256] query state ------------------------------------------------------------- */

/* y = [ReLU] [LayerNorm]( sum_z parts[z] + bias + residual );  out = y;  out_plus = y + addvec;  out2 = LN2(y).
 * rows_per_group>0: bias/ln_w/ln_b of row r are taken at offset (r / rows_per_group) * 256 (per-layer parameters).
 * Replaces: mmcv BaseTransformerLayer residual adds + norms and the shared post_norm
 * (MU/petr_transformer.py:563-565,586-592), Linear-LN-ReLU of the cls branch (cross_attention_head.py:127-133). */
int mv2d_row_ln(const float* parts, int n_parts, long long part_stride, const float* bias, const float* residual,
                const float* ln_w, const float* ln_b, int relu, float* out, const float* addvec, float* out_plus,
                const float* ln2_w, const float* ln2_b, float* out2, int M, float eps, int rows_per_group, void* stream);

/* Tail of CrossAttentionBoxHead.forward (RH/bbox_heads/cross_attention_head.py:219-238) + velocity / dt of
 * MV2DTHead._bbox_forward (RH/mv2d_t_head.py:136-140, dt = 0: skipped).  reg [L,R,10] in place, ref [R,3]. */
int mv2d_finalize_reg(float* reg, const float* ref, int L, int R, const float* pc_range, float dt, void* stream);

/* AvgPool2d(7) over [R,49,256] fp32 -> out[r*ld_out + c]  (RH/utils/query_generator.py:322-331). */
int mv2d_avgpool49(const float* x, float* out, int ld_out, int R, void* stream);

int mv2d_f32_to_bf16(const float* x, void* y, long long n, void* stream);

/* NCHW fp32 [V,C,HW] -> position-major [V*HW, C] fp32 (the layout every gather below reads). */
int mv2d_nchw_to_nhwc(const float* x, float* y, int V, int C, int HW, void* stream);
/* the same for the listed positions only: mask [V * HW] bytes (1 = some RoI's rectangle holds the position: mv2d_roi_positions* / mv2d_mask_compact's
 * roi_mask); unlisted rows of y are not written (HW, C multiples of 4; x, y 16-byte aligned) */
int mv2d_nchw_to_nhwc_masked(const float* x, float* y, const unsigned char* mask, int V, int C, int HW, void* stream);

/* "next" row f2 — the extra FPN level between the 2-D detector and the RoI head (mmdet FPN, start_level = end_level = 2, num_outs = 1:
 * configs/mv2d/exp/*:32-39, mmdet3d_plugin/models/detectors/mv2d.py:122-127): out = conv3x3(conv1x1(x) + b_lat) + b_fpn.
 *  mv2d_nchw_to_nhwc_bf16: detector output [V,C,HW] fp32 -> position-major bf16 (operand of the 1x1 lateral conv = mv2d_gemm_bf16);
 *  mv2d_map_conv3x3: in [V,h,w,256] bf16 (position-major), Wp = conv weight [256][tap][cin] in the fragment-major order of
 *  mv2d_pack_wfrag_bf16, out [V,h,w,256] fp32 position-major (what the RoI-head engine consumes without a transpose). */
int mv2d_nchw_to_nhwc_bf16(const float* x, void* y, int V, int C, int HW, void* stream);
int mv2d_map_conv3x3(const void* in, const void* Wp, const float* bias, float* out, int V, int h, int w, void* stream);

/* ---- attention ---------------------------------------------------------------------------------------- */

/* ---- batches of samples ---------------------------------------------------------------------------------
 * The reference runs ONE sample (V views) per GPU and call (RH/mv2d_head.py asserts batch size 1).  Here several
 * samples can share every launch: their views are numbered consecutively (view a belongs to sample a / V), their RoIs / queries are
 * concatenated, and grp_start[n_samples + 1] holds the first query row of every sample.  Everything that couples queries or
 * views (box correlation, self attention, top-k decode, result packing, velocity / dt) stays inside a sample, so a sample's
 * result does not depend on what else is in the batch.  grp_start == NULL means one sample. */

/* FlattenMHSelfAttention core (MU/petr_transformer.py:317-370): qkv [R,768] fp32 = in_proj(q|k|v) -> ctx [R,256]. */
int mv2d_self_attn_fwd(const float* qkv, float* ctx, int R, const int* grp_start, int n_samples, void* stream);

/* The same attention on bf16 MFMAs in split precision (default; csrc/self_attn_x3.hip): block = (64 queries, head, sample), the sample's
 * K / V rows staged once per block through LDS with whole-row loads (hi / lo bf16 images, V transposed), S^T = K.Q^T and O^T = V^T.P^T as
 * three-term bf16x3 products, online softmax over 32-key steps: 1e-5 against fp64 like mv2d_self_attn_fwd.  max_grp_rows: upper
 * bound of the rows of one sample (sizes the grid; 0 = R).  dn_pad > 0 (one sample, grp_start NULL): the denoising mask of
 * mv2d_self_attn_dn_fwd. */
int mv2d_self_attn_x3_fwd(const float* qkv, float* ctx, int R, const int* grp_start, int n_samples, int max_grp_rows, int dn_pad,
                          int dn_single, void* stream);

/* Training variant: the first dn_pad rows are denoising queries in groups of dn_single rows; the attention mask of
 * prepare_for_dn (mmdet3d_plugin/models/roi_heads/mv2d_s_head.py:95-107) is evaluated in the kernel: a key is visible iff
 * key >= dn_pad, or query < dn_pad and key / dn_single == query / dn_single.  One sample per launch. */
int mv2d_self_attn_dn_fwd(const float* qkv, float* ctx, int R, int dn_pad, int dn_single, void* stream);

/* PETRMultiheadAttention core (MU/petr_transformer.py:426-513) over the allowed (query,key) pairs only.
 * q [R,256] fp32 pre-scaled by 1/sqrt(32); K,V [S,256] bf16; CSR row_ptr[R+1], col_idx[nnz] (key indices);
 * ctx [R,256] fp32.  A query with no allowed key yields NaN like nn.MultiheadAttention (empty_nan = 1) or 0 (empty_nan = 0).
 * dbg_logits (optional): pre-softmax logits, head h at dbg_logits[h*dbg_stride + e], e in CSR order. */
int mv2d_sparse_xattn_fwd(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, float* ctx,
                          float* dbg_logits, long long dbg_stride, int R, int empty_nan, void* stream);

/* ---- cross attention on MFMA tiles, K/V projections folded into the query side (the default route; csrc/xattn_tile.hip) ----
 * Replaces PETRMultiheadAttention's in_proj of key / value + attention core (MU/petr_transformer.py:426-513,
 * torch.nn.MultiheadAttention with attn_mask / key_padding_mask): no per-layer K/V is written.  Per layer three enqueues:
 *
 * mv2d_xattn_qmap: q [R,256] fp32 (query in_proj output, pre-scaled by 1/sqrt(32)) -> Qt [R][8][64][8] key16: the per-head maps
 *   Wk_h^T q_h of the query into the 256-dim key INPUT space as a 16 x 256 MFMA operand per query (rows 0-7: key16 hi parts of
 *   the 8 heads, rows 8-15: lo remainders), fragment-major.  (The map itself is a bf16x3 product; its result is stored in the key-side format.)  WA_hi / WA_lo = the packed key in_proj weight
 *   (mv2d_amd.ops.pack_xattn_maps: [8 heads][16 tiles][64 lanes][8] bf16, layout in csrc/xattn_tile.hip).
 * mv2d_xattn_tile_fwd: one block per query; Xk / Xv [S,256] key16 = the UNPROJECTED key / value input rows (key + key_pos, key)
 *   shared by all layers and heads; CSR row_ptr [R+1] / col_idx [nnz]; z [R,8,256] fp32 = sum_j p_hj v_j per head.  Key tiles of
 *   16 rows are gathered with whole-row coalesced loads into swizzled LDS tiles, logits and P.V run on key16 (fp16) MFMAs (hi / lo split
 *   of the query map and of P: fp32-class on the query side), online softmax.  waves = 1 | 2 | 4 | 8 waves per query (0: default = 2).
 *   The row arrays are addressed with 32-bit byte offsets: fewer than 2^23 rows (4 GB) each.
 *   Xk_lo / Xv_lo (both or neither, may be NULL): key16 remainders of the rows (rows = Xk + Xk_lo): the fp32-class key side of the
 *   engine's index-exact route; mv2d_xattn_tile_fwd_ordered / mv2d_xattn_fused_fwd take lo_fmt: 0 = key16 lo rows, 1 = e4m3 "lo8" rows (256 B per row,
 *   see mv2d_pe_fused_x3; decoded to key16 in registers: results bitwise those of key16 lo rows holding the decoded values).  Rows without an allowed key: z = NaN (empty_nan = 1) or 0.  dbg_logits (optional): head h at dbg_logits[h*dbg_stride + e],
 *   e in CSR order, WITHOUT the per-(query, head) constant q_h . bk_h that cancels in the softmax.
 * mv2d_xattn_ctxmap: ctx [R,256] = Wv_h z_h + bv (bf16x3; WB_hi / WB_lo = the packed value in_proj weight); rows without an
 *   allowed key (row_ptr) give NaN / 0 like nn.MultiheadAttention / the 'zero' policy of the engine. */
int mv2d_xattn_qmap(const float* q, const void* WA_hi, const void* WA_lo, void* Qt, int R, void* stream);
int mv2d_xattn_tile_fwd(const void* Qt, const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr,
                        const int* col_idx, float* z, float* dbg_logits, long long dbg_stride, int R, int empty_nan, int waves, void* stream);
/* The same with a block -> query order (order [R]: a permutation of the query rows, e.g. mv2d_xattn_query_order's perm = the queries of every
 * sample sorted by their smallest key): neighbouring blocks then read overlapping key sets and share an L2.  Results are identical. */
int mv2d_xattn_tile_fwd_ordered(const void* Qt, const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr,
                                const int* col_idx, float* z, float* dbg_logits, long long dbg_stride, int R, int empty_nan, int waves,
                                const int* order, int lo_fmt, void* stream);
/* The three launches above as ONE (round 5, csrc/xattn_fused.hip): ctx [R,256] fp32 = xattn_ctxmap(xattn_tile(xattn_qmap(q))) for blocks of 8 queries,
 * Qt and z never leave the chip (16 KB per query and layer less HBM traffic, two launches less).  Same operands as the three calls (q = the scaled,
 * projected query rows [R,256] fp32; WA / WB = the packed map weights; Xk / Xv (+ _lo: index-exact route) the key16 row arrays; CSR; order = optional
 * launch order of the queries); results bitwise equal to the three kernels with waves = 1.  Meant for rows of similar length (S path): a block
 * waits for the longest of its 8 rows. */
int mv2d_xattn_fused_fwd(const float* q, const void* WA_hi, const void* WA_lo, const void* WB_hi, const void* WB_lo, const float* bv, const void* Xk,
                         const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr, const int* col_idx, float* ctx, int R, int empty_nan,
                         const int* order, int lo_fmt, void* stream);
/* Key tiles SHARED BETWEEN QUERIES (round 6, csrc/xattn_group.hip; the masks of RH/mv2d_t_head.py:79-109 / the duplication of
 * RH/mv2d_s_head.py:184-192 let 2.9-6.2 queries list the same key row).  mv2d_xattn_group_tables (once per frame): the queries of every sample
 * (grp_start [n_samples + 1]; rows behind grp_start[n_samples] are bucket padding and form groups of their own), taken in `order` (a permutation that
 * keeps every sample's rows in its own slot range, e.g. mv2d_xattn_query_order's; NULL = natural), are cut into groups of up to 8 consecutive slots;
 * per group g < ng_max (>= mv2d_xattn_group_max(R, n_samples)): g_slot / g_cnt = first slot / members (0 = no such group), g_ptr / g_len = its UNION
 * key list in ucol / umask (capacity ucap entries >= nnz + 16 ng_max; umask bit j = member j lists the key), padded to a multiple of 16 and sorted by
 * (umask, key) inside windows of 16384 consecutive key indices.  u_total [1] and flags [1] int32 must be zero on entry (flags[0] != 0 afterwards:
 * ucap exceeded).  A row of the CSR must not list a key twice.  mv2d_xattn_group_fwd: ctx [R,256] = the cross attention of mv2d_xattn_fused_fwd
 * (same operands, same arithmetic per (query, key) pair) with one block per group walking the group's union once; the keys of a softmax row are
 * visited in union order, so results agree with the per-query kernels to fp32 rounding, not bitwise. */
int mv2d_xattn_group_max(int R, int n_samples);
int mv2d_xattn_group_tables(const int* row_ptr, const int* col_idx, const int* order, const int* grp_start, int n_samples, int R, int ng_max, int* g_slot,
                            int* g_cnt, int* g_ptr, int* g_len, int* ucol, unsigned char* umask, int ucap, int* u_total, int* flags, void* stream);
int mv2d_xattn_group_fwd(const float* q, const void* WA_hi, const void* WA_lo, const void* WB_hi, const void* WB_lo, const float* bv, const void* Xk,
                         const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr, const int* order, const int* g_slot, const int* g_cnt,
                         const int* g_ptr, const int* g_len, const int* ucol, const unsigned char* umask, float* ctx, int ng_max, int empty_nan,
                         void* stream);
int mv2d_xattn_ctxmap(const float* z, const void* WB_hi, const void* WB_lo, const float* bv, const int* row_ptr, float* ctx, int R,
                      int empty_nan, void* stream);

/* T path (masked-map cross attention, RH/mv2d_t_head.py:79-109), launch order of the per-query blocks (csrc/xattn_order.hip): perm [R] = the rows of every sample sorted by their smallest key index (CSR rows ascending, as mv2d_mask_compact writes
 * them; stride = 0) -- or, stride > 0 (S path: 49), by the smallest of every stride-th entry (the first cell of every RoI a row lists);
 * flags[0] != 0: more than 4096 queries in a sample (natural order kept).  For mv2d_xattn_tile_fwd_ordered. */
int mv2d_xattn_query_order(const int* row_ptr, const int* col_idx, const int* grp_start, int n_samples, int R, int* perm, int* flags, int stride,
                           void* stream);
/* The two row kernels around the tile cross attention with its per-head maps fused in (one launch each instead of two; bitwise the
 * same results):
 * mv2d_attn_out_qmap_x3 = mv2d_attn_out_fused_x3 (out_proj + residual + LayerNorm of the self attention, cross-attention q projection)
 *   followed by mv2d_xattn_qmap on the q tile while it is still in LDS; writes x_out and Qt (q itself is not written).
 * mv2d_attn_out_zmap_x3 = mv2d_xattn_ctxmap on z [M,8,256] (+ the empty-row rule) followed by mv2d_attn_out_fused_x3 without a q
 *   stage (out_proj + residual + LayerNorm of the cross attention). */
int mv2d_attn_out_qmap_x3(const float* ctx, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo, const float* ln_w,
                          const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi, const void* Wq_lo, const float* bq,
                          float qscale, const void* WA_hi, const void* WA_lo, void* Qt, int M, float eps, void* stream);
int mv2d_attn_out_zmap_x3(const float* z, const void* WB_hi, const void* WB_lo, const float* bv, const int* row_ptr, int empty_nan,
                          const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo, const float* ln_w, const float* ln_b,
                          float* x_out, int M, float eps, void* stream);

/* Backward of mv2d_sparse_xattn_fwd ("next" row f3, the training path of the head): given dctx [R,256] returns dq [R,256] (gradient
 * with respect to the pre-scaled q) and dK, dV [S,256] fp32 (every key row is written; keys nobody reads get 0).  Two launches, no
 * atomics, deterministic: a pass over the queries (softmax statistics recomputed, no forward state kept; writes dq and, per allowed pair
 * e and head, p and ds into pair_ws [nnz,16]) and a pass over the keys that needs the pairs sorted by key: key_ptr [S+1], pair_idx [nnz]
 * (pair ids in CSR order, grouped by key), pair_row [nnz] (query of pair e).  ctx = the forward output.  Rows without an allowed key
 * get dq = 0. */
int mv2d_sparse_xattn_bwd(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                          const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                          float* dq, float* dK, float* dV, int R, int S, void* stream);
/* The same pair with ATTENTION-PROBABILITY DROPOUT (nn.MultiheadAttention(dropout=p) in training, MU/petr_transformer.py:404-418): the softmax
 * probabilities are dropped with probability p_drop and the kept ones scaled by 1 / (1 - p_drop) before the value sum.  The keep decision of
 * (allowed pair e in CSR order, head h) is a counter-based hash of (seed, 8 e + h) (murmur3 finaliser; u >= p_drop * 2^32 keeps), so the
 * backward regenerates the forward's mask from the same (p_drop, seed) and nothing is stored.  p_drop = 0: identical to the plain entries. */
int mv2d_sparse_xattn_fwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, float* ctx, float* dbg_logits,
                               long long dbg_stride, int R, int empty_nan, float p_drop, unsigned int seed, void* stream);
int mv2d_sparse_xattn_bwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                               const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws, float* dq, float* dK,
                               float* dV, int R, int S, float p_drop, unsigned int seed, void* stream);
/* long_rows != 0: for patterns with hundreds of keys per query AND of queries per key (the decoder's self attention in training): 16 waves per
 * query in the query pass, one 4-wave block per key in the key pass; the results differ from long_rows = 0 by summation order only.
 * dq_scale: factor on dq (the 1 / sqrt(d) of the scaled query projection). */
int mv2d_sparse_xattn_bwd_ex(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx, const float* dctx,
                             const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws, float* dq, float* dK, float* dV, int R,
                             int S, float p_drop, unsigned int seed, int long_rows, float dq_scale, void* stream);

/* ---- geometry / gather ---------------------------------------------------------------------------------- */

/* MV2DHead.get_box_params + process_intrins_feat (RH/mv2d_head.py:51-72,95-101) and inverse(K_roi @ E^T).float()
 * of QueryGenerator.center2lidar (RH/utils/query_generator.py:337-339).
 * rois [R,5] fp32 (view,x1,y1,x2,y2); viewK/viewE [V,16] fp64; K_roi [R,16] fp64 (optional);
 * intr[r*ld_intr + 0..15] fp32 (x0.1, zeroed for boxes < min_size, clamped to +-5e3); minv [R,16] fp32. */
int mv2d_box_params(const float* rois, const double* viewK, const double* viewE, double* K_roi, float* intr, int ld_intr,
                    float* minv, int R, float roi_size, float intr_scale, float min_size, void* stream);

/* center2lidar mat-vec + pc_range normalisation (NOT clamped, RH/mv2d_t_head.py:51-57) + pos2posemb3d (MU/pe.py:21-33).
 * center_pred rows (u,v,depth) at stride ld_cp; dim_t[128]; out xyz [R,3], ref [R,3], posemb [R,384] (y|x|z). */
int mv2d_refpoint_posemb(const float* center_pred, int ld_cp, const float* minv, const float* dim_t, float* xyz, float* ref,
                         float* posemb, int R, const float* pc_range, void* stream);

/* inverse(bmm(K_roi, E^T)).float() of QueryGenerator.center2lidar (RH/utils/query_generator.py:337-339) for per-RoI
 * fp64 matrices K_roi, E [R,16] -> minv [R,16] fp32. */
int mv2d_lidar2img_inverse(const double* K_roi, const double* E, float* minv, int R, void* stream);

/* pos2posemb3d (MU/pe.py:21-33) alone: ref [R,3] (x,y,z normalised) -> posemb [R,384] in (y|x|z) order. */
int mv2d_posemb3d(const float* ref, const float* dim_t, float* posemb, int R, void* stream);

/* mmcv.ops.RoIAlign(7, 1/16, sampling_ratio, 'avg', aligned=True) (call site RH/mv2d_head.py:114-115) on one or two
 * position-major maps -> [R,49,256] key16 and/or fp32 per map.  map1_index (optional): map1 is row-compacted, row of
 * position p is map1_index[p].  out1_is_sum: out1 = key16(map0 value + map1 value) (the S-path key input feat + pe). */
int mv2d_roi_align(const float* map0, const float* map1, const float* rois, void* out0, void* out1, float* out0_f32,
                   float* out1_f32, int R, int H, int W, int channels, float spatial_scale, int sampling_ratio,
                   const int* map1_index, int out1_is_sum, void* stream);
/* mv2d_roi_align with key16 REMAINDER outputs: out0_lo / out1_lo = key16(x - key16(x)) next to out0 / out1 (x ~ hi + lo, ~2^-22 relative):
 * the fp32-class key / value / conv-input rows of the index-exact route.  out0_lo8 / out1_lo8 (optional, ABI 6): the same remainders as e4m3 "lo8"
 * rows [R,49,256] bytes; lo8_flag: saturation report (see mv2d_pe_fused_x3). */
int mv2d_roi_align_ex(const float* map0, const float* map1, const float* rois, void* out0, void* out1, float* out0_f32, float* out1_f32,
                      int R, int H, int W, int channels, float spatial_scale, int sampling_ratio, const int* map1_index, int out1_is_sum,
                      void* out0_lo, void* out1_lo, void* out0_lo8, void* out1_lo8, int* lo8_flag, void* stream);

/* BoxCorrelation.epipolar_in_box, 'topk_matched:k:thr:ratio' (RH/utils/box_correlation.py:196-398).
 * V = views per sample; view_start[n_views+1]: first RoI of each view; trans [n_views,V,16] fp64 = lidar2img[b] @ inv(lidar2img[a])
 * for source view a (global index) and destination view b of the same sample (local index);
 * lin[sample_size] = linspace(0,1); depths[num_depth] (LID); match [R,V,topk] int32: (global) RoI id or -1, rank order. */
int mv2d_box_correlation(const float* rois, const int* view_start, const double* trans, const float* lin, const float* depths,
                         int* match, int R, int V, int sample_size, int num_depth, int topk, int pad_h, int pad_w,
                         float depth_start, float iou_thr, float ratio, int max_per_view, void* stream);
/* The feature-independent geometry of a frame in ONE launch (a one-sample frame is bound by its number of kernels): mv2d_box_params +
 * mv2d_box_correlation (same arguments, same results) + the clearing of zero_bytes bytes at zero_ptr (16-byte aligned, a multiple of 16; the
 * engine's per-frame mask / flag bytes; may be NULL / 0). */
int mv2d_frame_geometry(const float* rois, const double* viewK, const double* viewE, double* K_roi, float* intr, int ld_intr, float* minv,
                        float roi_size, float intr_scale, float min_size, const int* view_start, const double* trans, const float* lin,
                        const float* depths, int* match, int R, int V, int sample_size, int num_depth, int topk, int pad_h, int pad_w,
                        float depth_start, float iou_thr, float ratio, int max_per_view, void* zero_ptr, long long zero_bytes, void* stream);

long long mv2d_csr_workspace_bytes(int R, int V, int h, int w);

/* T-path masks -> compacted key list + CSR (BoxCorrelation.gen_box_correlation RH/utils/box_correlation.py:95-162 and
 * the mask/gather block RH/mv2d_t_head.py:67-88).  roi_mask [V*h*w] bytes must be zeroed by the caller.
 * out: rect [R,5]; pos2s [P]; s2pos [<=P]; *S_out; row_ptr [R+1]; col_idx [min(nnz,col_cap)];
 * nnz_out[0] = nnz, nnz_out[1] = 1 if nnz exceeded col_cap (caller pre-zeroes nnz_out[1]). */
int mv2d_mask_compact(const float* rois, const int* match, const unsigned char* pad_mask, unsigned char* roi_mask, int* rect,
                      int* pos2s, int* s2pos, int* S_out, unsigned int* bits_ws, int* row_count, int* row_ptr, int* col_idx,
                      int* nnz_out, int col_cap, int R, int V, int h, int w, int topk, float stride, float expand_stride,
                      int n_samples /* maps of n_samples * V views; V = views per sample */, void* stream);

/* (expand_stride < 0, round 5: instead of an expanded rectangle, EXACTLY the cells the bilinear taps of mv2d_roi_align touch for that RoI --
 * aligned, adaptive sampling grid, spatial_scale = 1 / stride; the S path evaluates the PE block only there)
 * mark + scan only: compact list of the map positions inside any RoI rect expanded by expand_stride cells
 * (S-path: the positions RoIAlign can touch, so that PE is evaluated only there). */
int mv2d_roi_positions(const float* rois, const unsigned char* pad_mask, unsigned char* roi_mask, int* rect, int* pos2s,
                       int* s2pos, int* S_out, int R, int V, int h, int w, float stride, float expand_stride, void* stream);

/* S-path CSR over the RoI-feature memory rows r*49+cell (RH/mv2d_s_head.py:184-192). */
int mv2d_csr_from_corr(const int* match, int* row_ptr, int* col_idx, int* nnz_out, int R, int V, int topk, void* stream);
/* mv2d_roi_positions + mv2d_csr_from_corr in two launches instead of three (the position scan and the CSR run side by side in one): V = all
 * views of the maps, Vg = views per sample (match is [R, Vg, topk]); Vg * topk <= 4096.  order (optional, with grp_start [n_samples + 1]): the
 * launch order of the attention blocks for mv2d_xattn_tile_fwd_ordered from the same launch -- the queries of every sample ranked by the
 * smallest RoI they list (own or matched), so that matched RoIs of different views share an L2.  A sample with more than 4096 queries keeps its
 * natural order and sets order_flags[0] = 1 (optional int[1], cleared by the caller; results are the same either way). */
int mv2d_roi_positions_csr(const float* rois, const unsigned char* pad_mask, unsigned char* roi_mask, int* rect, int* pos2s, int* s2pos,
                           int* S_out, int R, int V, int h, int w, float stride, float expand_stride, const int* match, int* row_ptr,
                           int* col_idx, int* nnz_out, int Vg, int topk, const int* grp_start, int n_samples, int* order, int* order_flags, void* stream);

/* Frustum rows of the index-exact route's PE block alone: out [S, 3 depth_num] fp32 = float(inverse_sigmoid(normalised 3-D point of every depth bin)),
 * computed in fp64 like the reference (MU/pe.py:96-131) at the positions s2pos[0 .. *S_dev); position_range = 6 doubles on the HOST.  Replaces the
 * A_frustum_f32 output of mv2d_pe_inputs on the inference path (same values up to one fp32 ulp in ~1 element per 1e8; 3-4 x faster).  Its 2 KB
 * logarithm table travels as a kernel argument (round 6): no per-device state, safe inside a stream capture and with several GPUs per process. */
int mv2d_pe_frustum_f32(const int* s2pos, const int* S_dev, int S_max, const double* img2lidar, const double* coords_w, const double* coords_h,
                        const double* coords_d, float* out, int V, int h, int w, int depth_num, const double* position_range, void* stream);

/* PE inputs at the listed key positions only (MU/pe.py:84-135 frustum, MU/positional_encoding.py:78-95 sine) + feature gather.
 * out: A_frustum [S,3*D] key16, A_sine [S,384] key16, Xf_k16 [S,256] key16, Xf_f32 [S,256] (optional: NULL when mv2d_pe_fused_tab reads the map).
 * A_sine may be NULL (the sine branch of the PE block comes from the engine's folded table: the row is not produced).
 * A_frustum_f32 / A_sine_f32 (optional, the engine's index-exact validation mode): the same rows unrounded, with the logarithm in
 * fp64 and library sin / cos. */
int mv2d_pe_inputs(const int* s2pos, const int* S_dev, int S_max, const float* featcl, const double* img2lidar,
                   const double* coords_w, const double* coords_h, const double* coords_d, const float* embeds,
                   const float* dim_t, void* A_frustum, void* A_sine, void* Xf_k16, float* Xf_f32, float* A_frustum_f32,
                   float* A_sine_f32, int V, int h, int w, int depth_num, const double* position_range, void* stream);

/* NMSFreeCoder.decode_single + get_bboxes (CB/coders/nms_free_coder.py:49-102, CB/util.py:60-87,
 * RH/bbox_heads/cross_attention_head.py:357-377): top-k over R*num_classes logits, denormalise, centre-range filter.
 * out: boxes [<=max_num,9], scores, labels (int64), bbox_index (int64), *count_out.
 * A batch (grp_start != NULL): one top-k per sample, outputs [n_samples][max_num], count_out [n_samples], bbox_index relative to
 * the sample's first row; max_grp_rows = rows of the largest sample.
 * payload (optional): [n_samples][max_num * 11 + 1] fp32, the wire format of the per-step all-gather (what mv2d_pack_detections writes) from the
 * same launch. */
int mv2d_decode_topk(const float* cls, const float* reg, int R, int num_classes, int max_num, const float* post_center_range,
                     float* boxes, float* scores, long long* labels, long long* bbox_index, int* count_out,
                     long long* topk_index_dbg, const int* grp_start, int n_samples, int max_grp_rows, float* payload, void* stream);

/* Rotated bird's-eye-view NMS for nms_thr < 1 (not a shipped value: with 1.0 nothing is suppressed and mv2d_result_pack alone is the
 * step after the head, mmdet3d_plugin/models/detectors/mv2d.py:265-287): per class, greedy in score order, IoU of the rotated
 * rectangles (x, y, dx, dy, yaw) of boxes [n_samples][in_stride][9]; scores_out = scores with the suppressed entries at -inf (feed it
 * to mv2d_result_pack).  mmdet3d box3d_multiclass_nms / mmcv nms_rotated are third party: parity unpinned. */
int mv2d_nms_bev(const float* boxes, const float* scores, const long long* labels, const int* count, float nms_thr, float* scores_out,
                 int n_samples, int in_stride, void* stream);

/* "next" row f1 — the caller's post-decoder step (mmdet3d_plugin/models/detectors/mv2d.py:265-287): mmdet3d
 * box3d_multiclass_nms(score_thr, nms_thr = 1.0 => no suppression, max_num) + result ordering: class-major, score-descending
 * (global score order only when more than max_num boxes survive).  in: boxes [n,9], scores [n], labels [n] int64, *count = n (<= 1024). */
int mv2d_result_pack(const float* boxes, const float* scores, const long long* labels, const int* count, float score_thr, int max_num,
                     float* out_boxes, float* out_scores, long long* out_labels, int* out_count,
                     int n_samples /* inputs [n_samples][in_stride], count [n_samples]; outputs [n_samples][max_num] */, int in_stride,
                     void* stream);

/* Wire format of the evaluation step's all-gather of decoded boxes (replaces multi_gpu_test's pickle / tmpdir collection,
 * tools/test.py:249-250): out [n_samples][max_num*11 + 1] fp32 = max_num rows of (box[9], score, label), rows >= count zeroed, then
 * the count.  in: boxes [n_samples][in_stride][9], scores / labels [n_samples][in_stride], count [n_samples]. */
int mv2d_pack_detections(const float* boxes, const float* scores, const long long* labels, const int* count, float* out, int n_samples,
                         int max_num, int in_stride, void* stream);

/* ---- training targets of the box head ------------------------------------------------------------------------
 * Cost matrix of HungarianAssigner3D.assign (mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:120-131) for n_layers decoder
 * layers in one launch: cost[l][r][g] = FocalLossCost(cls[l][r], gt_labels[g]) * cls_weight + L1(box[l][r][:8], normalize_bbox(gt[g])[:8])
 * * reg_weight, then nan_to_num(nan = 100, +inf = 100, -inf = -100).  cls [n_layers][R][C] logits, box [n_layers][R][10] head codes,
 * gt [G][9] = (cx, cy, cz, w, l, h, yaw, vx, vy) (the caller's cat(gravity_center, tensor[:, 3:]), cross_attention_head.py:449-451).
 * The assignment (scipy linear_sum_assignment in the reference) stays on the host. */
int mv2d_match_cost(const float* cls, const float* box, const float* gt, const int* gt_labels, float* cost, int n_layers, int R, int G,
                    int C, float cls_weight, float reg_weight, float alpha, float gamma, void* stream);

/* Set-prediction loss of CrossAttentionBoxHead.loss_single (cross_attention_head.py:380-434) and dn_loss_single (:477-538) with its
 * gradient, all layers in one launch.  match [n_layers][R]: row of gt / gt_labels assigned to the query, -1 = background; a gt label
 * equal to C is a background target (denoising negatives; their boxes are skipped when skip_background_boxes != 0).  loss [n_layers][2]
 * = (loss_cls, loss_bbox) = (sum focal / cls_avg_factor * loss_cls_weight, sum |box - normalize_bbox(gt)| * code_weights over rows with
 * finite targets / box_avg_factor * loss_bbox_weight), nan_to_num applied.  dcls / dbox (may be null): gradient of
 * sum_l layer_weights[l] * (loss_cls[l] + loss_bbox[l]) (layer_weights null = ones). */
int mv2d_set_loss(const float* cls, const float* box, const int* match, const float* gt, const int* gt_labels, const float* code_weights,
                  const float* layer_weights, float* loss, float* dcls, float* dbox, int n_layers, int R, int G, int C,
                  float cls_avg_factor, float box_avg_factor, float alpha, float gamma, float loss_cls_weight, float loss_bbox_weight,
                  int skip_background_boxes, void* stream);

/* Denoising queries of MV2DSHead.prepare_for_dn (mmdet3d_plugin/models/roi_heads/mv2d_s_head.py:39-78), one sample: row i = (repeat
 * i / G, box i % G) for i < G * scalar.  rnd [G*scalar][3] uniform in [0, 1) (the reference's torch.rand_like).  With noise_scale > 0:
 * centre += (2 rnd - 1) * (size / 2 + noise_trans) * noise_scale, normalised by pc_range_host (6 floats, host memory), clamped to
 * [eps, 1 - eps]; rows with |2 rnd - 1|_2 > split get label num_classes.  With noise_scale <= 0 the centre is copied unnormalised (as the
 * reference does).  ref [G*scalar][3], labels [G*scalar] int64, boxes [G*scalar][9] (the repeated targets). */
int mv2d_dn_queries(const float* gt, const int* gt_labels, const float* rnd, int G, int scalar, float noise_scale, float noise_trans, float split,
                    int num_classes, const float* pc_range_host, float eps, float* ref, long long* labels, float* boxes, void* stream);

/* ---- dense building blocks of the training route (csrc/train_ops.hip) ----------------------------------------------------
 * Every product of a linear layer's forward and backward runs on mv2d_gemm_f32x3 (below): forward y = x W^T, dx = dy W (W transposed),
 * dW = dy^T x (both operands transposed).  (mv2d_split3_operand / mv2d_matmul_nt_x3, round 3's first build on operand images, are retired.) */
/* out [cols] = column sums of x [rows, cols] (row stride ld), fixed summation order (bias gradients, split-K slabs).  Long matrices are summed
 * in two passes through scratch [mv2d_colsum_scratch_rows(rows), cols] (0 rows: not needed; NULL: one pass). */
int mv2d_colsum_scratch_rows(int rows);
int mv2d_colsum(const float* x, long long ld, int rows, int cols, float* out, float* scratch, void* stream);
/* nn.LayerNorm(256) backward (MU/petr_transformer.py norms, cross_attention_head.py:127-133): dx [M,256], dw [256], db [256] from x, dy, w
 * (mean / rstd recomputed); dw_part / db_part: scratch [mv2d_layer_norm_bwd_blocks(M), 256] each. */
int mv2d_layer_norm_bwd_blocks(int M);
int mv2d_layer_norm_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw_part, float* db_part, float* dw, float* db, int M,
                        float eps, void* stream);
/* Composite entry: the whole launch sequence of one linear layer's backward per call, on a caller-provided workspace (256-byte aligned, >=
 * mv2d_linear_bwd_x3_ws_bytes of the shape), so that a training step is not bound by per-launch host time.
 * mv2d_linear_bwd_x3: backward of y = act(x W^T + b): g = dy masked by y > 0 (y NULL: no activation; dense [M,N] rows, any alignment); dx [M,K] =
 * g W, dW [N,K] = g^T x (both on mv2d_gemm_f32x3), db [N] = column sums of g; each output optional (NULL). */
/* Split-precision product on fp32 operands read IN PLACE in either orientation (csrc/gemm_f32x3.hip): C [M, ldc] = act(op(A) op(B)^T + bias),
 * op(X) = X^T when trans_x; operands are split into bf16 hi / lo while a tile is staged into LDS (a transposed operand through a register
 * transpose), three MFMAs per k-step; split-K slabs + their fixed-order sum in `ws` (mv2d_gemm_f32x3_ws_bytes; 256-byte aligned; optional). */
long long mv2d_gemm_f32x3_ws_bytes(int M, int N, int K);
int mv2d_gemm_f32x3(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act, float* C,
                    long long ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream);
long long mv2d_linear_bwd_x3_ws_bytes(int M, int N, int K);
int mv2d_linear_bwd_x3(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW, float* db, int M, int N, int K, void* ws,
                       long long ws_bytes, void* stream);

/* mv2d_gemm_f32x3 with C = act((op(A) op(B)^T + bias) * alpha); accumulate: C += (fp32 C; with split-K the slabs are summed onto C);
 * out_bf16: C is a bf16 [M, ldc] image (one pass; the keys / values the sparse attention kernels read).
 * mv2d_colsum_add: out[c] = add[c] + sum_r x[r, c] (add NULL: the plain sum; add == out accumulates). */
int mv2d_gemm_f32x3_ex(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act, float alpha,
                       int accumulate, int out_bf16, void* C, long long ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream);
int mv2d_colsum_add(const float* x, long long ld, int rows, int cols, float* out, float* scratch, const float* add, void* stream);
/* dW [N,K] = g^T x and db [N] = column sums of g (g [M,N], x [M,K] dense rows) -- db inside the product's kernel when it runs in one pass, a
 * separate column sum after a split-K product (cs_scratch: [mv2d_colsum_scratch_rows(M), N] floats or NULL). */
/* `batch` products of one shape in one launch: C_b [M, ldc] = op(A_b) op(B_b)^T, X_b = X + b * batch_x elements (the per-head products of a dense
 * attention block: head b = a 32-column slice of [rows, 256] operands).  C = alpha * product, no bias / activation.  ws (16-byte aligned,
 * mv2d_gemm_f32x3_batched_ws_bytes; NULL = one pass): split-K slabs for few output tiles with a long contraction, summed in fixed order. */
/* Softmax backward of a dense attention block, in place: dP [rows, ld] (first cols columns) <- P * (dP m - rowsum(P dP m)), m = keep_scale where the
 * dropped probabilities Pd are non-zero, 0 elsewhere (Pd == P: no dropout). */
int mv2d_softmax_bwd_rows(const float* P, const float* Pd, float* dP, long long ld, int rows, int cols, float keep_scale, void* stream);
long long mv2d_gemm_f32x3_batched_ws_bytes(int M, int N, int K, int batch);
int mv2d_gemm_f32x3_batched(const float* A, long long lda, long long batch_a, int trans_a, const float* B, long long ldb, long long batch_b, int trans_b,
                            float* C, long long ldc, long long batch_c, int M, int N, int K, int batch, float alpha, void* ws, long long ws_bytes,
                            void* stream);
/* dx [M,K] = (g [M,N] W [N,K]) * alpha, zeroed where relu_y [M,K] <= 0 (NULL: no mask) -- the ReLU (+ dropout scale) of the forward applied to the
 * input gradient in the product's epilogue. */
int mv2d_dgrad_relu_f32x3(const float* g, const float* W, const float* relu_y, float alpha, float* dx, int M, int N, int K, void* stream);
int mv2d_wgrad_f32x3(const float* g, const float* x, float* dW, float* db, int M, int N, int K, void* ws, long long ws_bytes, float* cs_scratch,
                     void* stream);

/* The decoder of the training route as ONE call per direction (replaces the per-operator autograd graph over PETRTransformerDecoder,
 * MU/petr_transformer.py:195-311,404-418,501-508,563-590: six post-norm layers self_attn - norm - cross_attn - norm - ffn - norm, the shared
 * post_norm on every intermediate output; mmcv residual / dropout rules).  The launch sequence is issued from C: the dx chain on `stream`, the
 * parameter-gradient products and the key side of the cross attention on internal side streams that are joined before the call returns
 * (device-side; the call itself never blocks).
 * params / grads: 18 L + 2 device pointers -- per layer: self-attention in_proj weight [768,256] / bias, out_proj weight / bias, norm-0 weight /
 * bias; the same six for the cross attention and norm 1; FFN linear-1 weight [F,256] / bias, linear-2 weight [256,F] / bias, norm 2; then
 * post_norm weight / bias.  Gradients are written (not accumulated).
 * qpos [T,256]; key_in / val_in [S,256] fp32; (row_ptr, col) CSR patterns of the self / cross attention, (key_ptr, pair_idx, pair_row) their
 * transposes; outs / d_outs [L,T,256].  Dropout (probabilities in dims, 0 = eval): counter-hash masks of (seed, layer, site, element),
 * regenerated by the backward.  Denoising queries (dims.pad > 0): the cross-attention pattern covers the rows pad..T-1 only, the first pad rows attend
 * to the key rows dn_keys [nk] (sorted, unique; NULL = all S rows) as a dense block (RH/mv2d_t_head.py:90-98) on unrounded fp32 keys / values.
 * act (mv2d_train_decoder_act_bytes) carries the activations from the forward to the backward;
 * ws: mv2d_train_decoder_ws_bytes(dims, backward) bytes of scratch; both 256-byte aligned. */
typedef struct mv2d_td_dims {
    int T, S, L, F;
    int sa_nnz, ca_nnz;
    float p_sa_attn, p_sa_out, p_ca_attn, p_ca_out, p_ffn_act, p_ffn_out;
    unsigned int seed;
    float eps;
    int pad, nk;     /* the first `pad` rows are denoising queries: they see the nk key rows dn_keys (a dense block); 0, 0 = none */
} mv2d_td_dims;
long long mv2d_train_decoder_act_bytes(const mv2d_td_dims* d);
long long mv2d_train_decoder_ws_bytes(const mv2d_td_dims* d, int backward);
int mv2d_train_decoder_fwd(const mv2d_td_dims* d, const float* const* params, const float* qpos, const float* key_in, const float* val_in,
                           const int* sa_row_ptr, const int* sa_col, const int* ca_row_ptr, const int* ca_col, const int* dn_keys, float* outs, void* act,
                           void* ws, void* stream);
int mv2d_train_decoder_bwd(const mv2d_td_dims* d, const float* const* params, float* const* grads, const float* qpos, const float* key_in,
                           const float* val_in, const int* sa_row_ptr, const int* sa_col, const int* sa_key_ptr, const int* sa_pair_idx,
                           const int* sa_pair_row, const int* ca_row_ptr, const int* ca_col, const int* ca_key_ptr, const int* ca_pair_idx,
                           const int* ca_pair_row, const int* dn_keys, const float* d_outs, const void* act, void* ws, float* d_qpos, float* d_key_in,
                           float* d_val_in, void* stream);

/* The classification / regression branches of all L intermediate outputs in one call per direction (RH/bbox_heads/cross_attention_head.py:118-142,
 * 200-218): cls_l = Linear(ReLU(LN(Linear(ReLU(LN(Linear(out_l))))))), reg_l = Linear(ReLU(Linear(ReLU(Linear(out_l))))) (the raw box code).
 * params / grads: 16 L device pointers -- per layer cls_branches.{0,1,3,4,6}.{weight,bias}, reg_branches.{0,2,4}.{weight,bias}.
 * outs / d_outs [L,T,256]; cls / d_cls [L,T,NC]; reg / d_reg [L,T,10].  Layers run side by side on the caller's stream and the internal side
 * streams (joined before the call returns, device-side).  act / ws as for mv2d_train_decoder_*. */
typedef struct mv2d_th_dims { int T, L, NC; float eps; } mv2d_th_dims;
long long mv2d_train_heads_act_bytes(const mv2d_th_dims* d);
long long mv2d_train_heads_ws_bytes(const mv2d_th_dims* d, int backward);
int mv2d_train_heads_fwd(const mv2d_th_dims* d, const float* const* params, const float* outs, float* cls, float* reg, void* act, void* ws,
                         void* stream);
int mv2d_train_heads_bwd(const mv2d_th_dims* d, const float* const* params, float* const* grads, const float* outs, const float* d_cls,
                         const float* d_reg, const void* act, void* ws, float* d_outs, void* stream);

/* Box code of all L intermediate outputs, forward and backward (RH/bbox_heads/cross_attention_head.py:216-238, RH/mv2d_t_head.py:136-140):
 * out[0,1,4] = sigmoid(t[0,1,4] + inverse_sigmoid(ref)) * range + low; out[8,9] = t[8,9] / dt for the rows >= pad when dt != 0; the rest passes.
 * t / out / g / d_t [L,T,10], ref / d_ref [T,3] (d_ref: summed over the layers; NULL = not wanted); pc_range: 6 HOST floats. */
int mv2d_box_code_fwd(const float* t, const float* ref, float* out, int L, int T, int pad, float dt, const float* pc_range, void* stream);
int mv2d_box_code_bwd(const float* g, const float* out, const float* ref, float* d_t, float* d_ref, int L, int T, int pad, float dt,
                      const float* pc_range, void* stream);

/* HOST function (no device work): the linear sum assignment of HungarianAssigner3D for all decoder layers of a step
 * (mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:137 calls scipy.optimize.linear_sum_assignment per layer): the same
 * shortest-augmenting-path algorithm (Crouse 2016) with the same scan order and tie rule, one host thread per layer (threads <= 0) or
 * `threads` threads.  cost [L,R,G] fp32 and match [L,R] int32 are host pointers; match = assigned box or -1.  An error for NaN / -inf costs
 * or an infeasible problem (SciPy raises there). */
int mv2d_lsap_layers(const float* cost, int L, int R, int G, int* match, int threads);

/* The unfolded input of the query generator's 3 x 3 convolution over the 7 x 7 RoI features (RH/utils/query_generator.py:352-366; padding 1), so
 * that the convolution is one product: x [R,49,256] -> cols [R*49, 2304], column order (tap = 3 ky + kx, channel); and its gradient. */
int mv2d_im2col3x3(const float* x, float* cols, int R, void* stream);
int mv2d_col2im3x3(const float* dcols, float* dx, int R, void* stream);

/* center2lidar + normalisation of the reference points for training (RH/utils/query_generator.py:333-341, RH/mv2d_s_head.py:146-152):
 * c [R,3] = (u, v, depth), minv [R,16] -> ref [R,3] = ((minv (u d, v d, d, 1))[:3] - low) / range; the backward returns d c.  pc_range: 6 HOST floats. */
int mv2d_center2lidar_fwd(const float* c, const float* minv, float* ref, int R, const float* pc_range, void* stream);
int mv2d_center2lidar_bwd(const float* g, const float* c, const float* minv, float* dc, int R, const float* pc_range, void* stream);

/* Dense 8-head attention block of the denoising queries (RH/mv2d_t_head.py:90-98; MU/petr_transformer.py:404-418,501-508), forward and backward
 * without materialising the logits: q [n,256] fp32 (already scaled by 1/sqrt(32)), k / v [nk,256] fp32 (head h = columns 32 h .. 32 h + 31) ->
 * ctx [n,256] = dropout(softmax(q_h k_h^T)) v_h, lse [8][n] (log-sum-exp per row, for the backward).  The backward takes ctx / lse / dctx and the
 * same (p_drop, seed) -- the mask is a counter hash of (seed, head, query, key) -- and returns dq (times dq_scale), dk, dv.
 * ws: mv2d_dense_attn_ws_bytes(n, nk, backward) bytes, 256-byte aligned (bf16 hi / lo images of the operands, row-major and per-head transposed). */
long long mv2d_dense_attn_ws_bytes(int n, int nk, int backward);
int mv2d_dense_attn_fwd(const float* q, const float* k, const float* v, int n, int nk, float p_drop, unsigned int seed, float* ctx, float* lse, void* ws,
                        void* stream);
int mv2d_dense_attn_bwd(const float* q, const float* k, const float* v, const float* ctx, const float* dctx, const float* lse, int n, int nk, float p_drop,
                        unsigned int seed, float dq_scale, float* dq, float* dk, float* dv, void* ws, void* stream);
/* parts: 1 = dq only (also fills the per-row dctx . ctx in ws), 2 = dk / dv only (after a parts = 1 call on the same ws, in stream order or behind an
 * event: the training decoder issues it on a side stream), 3 = both. */
int mv2d_dense_attn_bwd_parts(const float* q, const float* k, const float* v, const float* ctx, const float* dctx, const float* lse, int n, int nk,
                              float p_drop, unsigned int seed, float dq_scale, float* dq, float* dk, float* dv, void* ws, int parts, void* stream);

/* Backward of mv2d_roi_align w.r.t. one map (training; mmcv's roi_align backward): grad_out [R][49][256] fp32 ->
 * grad_map [rows][256] fp32, ACCUMULATED with hardware fp32 atomics (the caller zeroes it; the summation order varies between runs).
 * index (may be null): position -> row of a compacted map, negative = no row (as map1_index of the forward). */
int mv2d_roi_align_bwd(const float* grad_out, const float* rois, float* grad_map, const int* index, int R, int H, int W, int channels,
                       float spatial_scale, int sampling_ratio, void* stream);

#ifdef __cplusplus
}
#endif
#endif
