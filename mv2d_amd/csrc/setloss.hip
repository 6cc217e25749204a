// Training targets of the box head: the cost matrix of the Hungarian assignment and the set-prediction loss
// (sigmoid focal + weighted L1) with its gradient, for all decoder layers in one launch each.
//
// Reference: CB/assigners/hungarian_assigner_3d.py:120-131 (cost = FocalLossCost + BBox3DL1Cost on the first 8 normalised box codes,
// nan_to_num(100, 100, -100)), CB/util.py:37-58 (normalize_bbox), RH/bbox_heads/cross_attention_head.py:380-434 (loss_single) and
// :477-538 (dn_loss_single).  FocalLossCost / FocalLoss / L1Loss are mmdet 2.25.1 (third party, restated).
// The assignment itself (scipy.optimize.linear_sum_assignment in the reference) stays on the host: mv2d_amd/train.py.
#include "common.h"

namespace {

constexpr int BOX_CODE = 10;   // (cx, cy, log w, log l, cz, log h, sin, cos, vx, vy)
constexpr int GT_CODE = 9;     // (cx, cy, cz, w, l, h, yaw, vx, vy)

__device__ __forceinline__ void normalize_gt(const float* __restrict__ g, float* t) {
    t[0] = g[0];
    t[1] = g[1];
    t[2] = logf(g[3]);
    t[3] = logf(g[4]);
    t[4] = g[2];
    t[5] = logf(g[5]);
    t[6] = sinf(g[6]);
    t[7] = cosf(g[6]);
    t[8] = g[7];
    t[9] = g[8];
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// cost[l][r][g]: one thread per element, g fastest (coalesced stores; the row's codes come from L1/L2).
__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ cls, const float* __restrict__ box,
                                                          const float* __restrict__ gt, const int* __restrict__ gt_labels,
                                                          float* __restrict__ cost, long long total, int G, int C, float cls_w,
                                                          float reg_w, float alpha, float gamma, float eps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long row = i / G;
    const int g = (int)(i - row * G);
    const int lab = gt_labels[g];
    float c = 0.f;
    if (lab >= 0 && lab < C) {
        const float p = sigmoid_f(cls[row * C + lab]);
        const float neg = -logf(1.f - p + eps) * (1.f - alpha) * powf(p, gamma);
        const float pos = -logf(p + eps) * alpha * powf(1.f - p, gamma);
        c = (pos - neg) * cls_w;
    }
    float t[BOX_CODE];
    normalize_gt(gt + (long long)g * GT_CODE, t);
    const float* b = box + row * BOX_CODE;
    float l1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) l1 += fabsf(b[k] - t[k]);
    c += l1 * reg_w;
    if (c != c) c = 100.f;                       // torch.nan_to_num(cost, nan=100, posinf=100, neginf=-100)
    else if (c == INFINITY) c = 100.f;
    else if (c == -INFINITY) c = -100.f;
    cost[i] = c;
}

struct LossArgs {
    const float* cls;        // [L][R][C] logits
    const float* box;        // [L][R][10]
    const int* match;        // [L][R] index into gt / gt_labels, -1 = background
    const float* gt;         // [G][9]
    const int* gt_labels;    // [G]; a label == C is a background ("negative") target row
    const float* code_w;     // [10]
    const float* layer_w;    // [L] weight of the layer in the total (gradient only)
    float* loss;             // [L][2] (loss_cls, loss_bbox), unweighted by layer_w
    float* dcls;             // [L][R][C] or null
    float* dbox;             // [L][R][10] or null
    int R, C, G;
    float cls_avg, box_avg, alpha, gamma, cls_w, box_w;
    int skip_bg_box;
};

// One block per layer; a thread walks rows r = tid, tid + 256, ...; fp64 partial sums reduced in a fixed order (deterministic).
__global__ __launch_bounds__(256) void set_loss_kernel(LossArgs a) {
    const int l = blockIdx.x;
    const int tid = threadIdx.x;
    const float lw = a.layer_w ? a.layer_w[l] : 1.f;
    const float gs_cls = lw * a.cls_w / a.cls_avg;
    const float gs_box = lw * a.box_w / a.box_avg;
    double s_cls = 0.0, s_box = 0.0;
    for (int r = tid; r < a.R; r += 256) {
        const long long row = (long long)l * a.R + r;
        const int m = a.match[row];
        const int lab = m >= 0 ? a.gt_labels[m] : a.C;
        const float* x = a.cls + row * a.C;
        for (int c = 0; c < a.C; ++c) {
            const float v = x[c];
            const float p = sigmoid_f(v);
            const float sp = log1pf(expf(-fabsf(v)));            // softplus(-|v|)
            float li, gi;
            if (c == lab) {
                const float bce = fmaxf(-v, 0.f) + sp;             // -log p
                const float pg = powf(1.f - p, a.gamma);
                li = a.alpha * pg * bce;
                gi = -a.alpha * pg * ((1.f - p) + a.gamma * p * bce);
            } else {
                const float bce = fmaxf(v, 0.f) + sp;              // -log(1 - p)
                const float pg = powf(p, a.gamma);
                li = (1.f - a.alpha) * pg * bce;
                gi = (1.f - a.alpha) * pg * (a.gamma * (1.f - p) * bce + p);
            }
            s_cls += (double)li;
            if (a.dcls) a.dcls[row * a.C + c] = gi * gs_cls;
        }
        float gb[BOX_CODE];
#pragma unroll
        for (int k = 0; k < BOX_CODE; ++k) gb[k] = 0.f;
        if (m >= 0 && !(a.skip_bg_box && lab == a.C)) {
            float t[BOX_CODE];
            normalize_gt(a.gt + (long long)m * GT_CODE, t);
            bool fin = true;
#pragma unroll
            for (int k = 0; k < BOX_CODE; ++k) fin = fin && (fabsf(t[k]) <= 3.402823466e38f);
            if (fin) {
                const float* b = a.box + row * BOX_CODE;
#pragma unroll
                for (int k = 0; k < BOX_CODE; ++k) {
                    const float d = b[k] - t[k];
                    const float w = a.code_w[k];
                    s_box += (double)(fabsf(d) * w);
                    gb[k] = (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * gs_box;
                }
            }
        }
        if (a.dbox) {
#pragma unroll
            for (int k = 0; k < BOX_CODE; ++k) a.dbox[row * BOX_CODE + k] = gb[k];
        }
    }
    __shared__ double red[2][256];
    red[0][tid] = s_cls;
    red[1][tid] = s_box;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        float lc = (float)red[0][0] / a.cls_avg * a.cls_w;
        float lb = (float)red[1][0] / a.box_avg * a.box_w;
        // torch.nan_to_num(loss): nan -> 0, +-inf -> +-FLT_MAX
        if (lc != lc) lc = 0.f;
        if (lb != lb) lb = 0.f;
        lc = fminf(fmaxf(lc, -3.402823466e38f), 3.402823466e38f);
        lb = fminf(fmaxf(lb, -3.402823466e38f), 3.402823466e38f);
        a.loss[2 * l] = lc;
        a.loss[2 * l + 1] = lb;
    }
}

// prepare_for_dn (RH/mv2d_s_head.py:39-78): row i = (repeat i / G, ground-truth box i % G).
__global__ __launch_bounds__(256) void dn_queries_kernel(const float* __restrict__ gt, const int* __restrict__ gt_labels,
                                                          const float* __restrict__ rnd, int G, int n, float noise_scale, float noise_trans,
                                                          float split, int num_classes, float r0, float r1, float r2, float r3, float r4,
                                                          float r5, float eps, float* __restrict__ ref, long long* __restrict__ labels,
                                                          float* __restrict__ boxes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* g = gt + (long long)(i % G) * GT_CODE;
#pragma unroll
    for (int k = 0; k < GT_CODE; ++k) boxes[(long long)i * GT_CODE + k] = g[k];
    float c[3] = {g[0], g[1], g[2]};
    long long lab = gt_labels[i % G];
    if (noise_scale > 0.f) {
        const float lo[3] = {r0, r1, r2}, hi[3] = {r3, r4, r5};
        float nrm = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float rp = rnd[(long long)i * 3 + k] * 2.f - 1.0f;
            const float diff = g[3 + k] / 2.f + noise_trans;
            c[k] += rp * diff * noise_scale;
            c[k] = (c[k] - lo[k]) / (hi[k] - lo[k]);
            c[k] = fminf(fmaxf(c[k], 0.0f + eps), 1.0f - eps);
            nrm += rp * rp;
        }
        if (sqrtf(nrm) > split) lab = num_classes;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) ref[(long long)i * 3 + k] = c[k];
    labels[i] = lab;
}

}  // namespace

extern "C" int mv2d_match_cost(const float* cls, const float* box, const float* gt, const int* gt_labels, float* cost, int n_layers,
                               int R, int G, int C, float cls_weight, float reg_weight, float alpha, float gamma, void* stream) {
    MV2D_CHECK_ARG(n_layers >= 0 && R >= 0 && G >= 0 && C > 0, "mv2d_match_cost: bad sizes");
    const long long total = (long long)n_layers * R * G;
    if (total == 0) return MV2D_OK;
    MV2D_CHECK_ARG(cls && box && gt && gt_labels && cost, "mv2d_match_cost: null pointer");
    const long long blocks = (total + 255) / 256;
    MV2D_CHECK_ARG(blocks < (1ll << 31), "mv2d_match_cost: too many elements");
    hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, cls, box, gt, gt_labels, cost, total,
                       G, C, cls_weight, reg_weight, alpha, gamma, 1e-12f);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_set_loss(const float* cls, const float* box, const int* match, const float* gt, const int* gt_labels,
                             const float* code_weights, const float* layer_weights, float* loss, float* dcls, float* dbox, int n_layers,
                             int R, int G, int C, float cls_avg_factor, float box_avg_factor, float alpha, float gamma,
                             float loss_cls_weight, float loss_bbox_weight, int skip_background_boxes, void* stream) {
    MV2D_CHECK_ARG(n_layers >= 0 && R >= 0 && G >= 0 && C > 0, "mv2d_set_loss: bad sizes");
    MV2D_CHECK_ARG(cls_avg_factor > 0.f && box_avg_factor > 0.f, "mv2d_set_loss: averaging factors must be positive");
    if (n_layers == 0) return MV2D_OK;
    MV2D_CHECK_ARG(loss && code_weights, "mv2d_set_loss: null pointer");
    MV2D_CHECK_ARG(R == 0 || (cls && box && match), "mv2d_set_loss: null pointer");
    MV2D_CHECK_ARG(G == 0 || (gt && gt_labels), "mv2d_set_loss: null ground truth");
    LossArgs a{cls, box, match, gt, gt_labels, code_weights, layer_weights, loss, dcls, dbox, R, C, G, cls_avg_factor, box_avg_factor,
               alpha, gamma, loss_cls_weight, loss_bbox_weight, skip_background_boxes};
    hipLaunchKernelGGL(set_loss_kernel, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, a);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_dn_queries(const float* gt, const int* gt_labels, const float* rnd, int G, int scalar, float noise_scale, float noise_trans,
                               float split, int num_classes, const float* pc_range_host, float eps, float* ref, long long* labels,
                               float* boxes, void* stream) {
    MV2D_CHECK_ARG(G >= 0 && scalar >= 0, "mv2d_dn_queries: bad sizes");
    const long long n = (long long)G * scalar;
    if (n == 0) return MV2D_OK;
    MV2D_CHECK_ARG(n < (1ll << 30), "mv2d_dn_queries: too many denoising queries");
    MV2D_CHECK_ARG(gt && gt_labels && ref && labels && boxes && pc_range_host && (rnd || noise_scale <= 0.f), "mv2d_dn_queries: null pointer");
    const float* r = pc_range_host;
    hipLaunchKernelGGL(dn_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gt, gt_labels, rnd, G, (int)n,
                       noise_scale, noise_trans, split, num_classes, r[0], r[1], r[2], r[3], r[4], r[5], eps, ref, labels, boxes);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
