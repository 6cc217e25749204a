// Rotated bird's-eye-view NMS of the post-decoder step ("next" row f1): MV2D.simple_test hands the decoded boxes to
// mmdet3d's box3d_multiclass_nms (mmdet3d_plugin/models/detectors/mv2d.py:265-287); with the shipped nms_thr = 1.0 nothing is
// suppressed (mv2d_result_pack alone is the whole step).  This kernel serves nms_thr < 1: per class, greedy suppression in score
// order by the IoU of the rotated BEV rectangles (x, y, dx, dy, yaw) = LiDARInstance3DBoxes.bev.
// mmdet3d / mmcv are third party and absent from the reference tree (parity unpinned): the intersection is computed by clipping one
// rectangle with the four half planes of the other (Sutherland-Hodgman) instead of mmcv's vertex/intersection hull; both are exact for
// convex quadrilaterals up to rounding.
// One block: ranks by counting, then one barrier-separated round per box; n <= 1024 (the head returns at most 300).
#include "common.h"

namespace {

struct Pt { float x, y; };

__device__ __forceinline__ void box_corners(const float* b, Pt (&c)[4]) {
    const float cx = b[0], cy = b[1], hw = 0.5f * b[3], hl = 0.5f * b[4], cs = cosf(b[6]), sn = sinf(b[6]);
    const float dx[4] = {hw, -hw, -hw, hw}, dy[4] = {hl, hl, -hl, -hl};
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = Pt{cx + dx[i] * cs - dy[i] * sn, cy + dx[i] * sn + dy[i] * cs};      // counter-clockwise
}

// area of (polygon P clipped by the convex counter-clockwise quadrilateral Q)
__device__ float clip_area(const Pt (&P)[4], const Pt (&Q)[4]) {
    Pt a[10], b[10];
    int n = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = P[i];
    for (int e = 0; e < 4 && n > 0; ++e) {
        const Pt q0 = Q[e], q1 = Q[(e + 1) & 3];
        const float ex = q1.x - q0.x, ey = q1.y - q0.y;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const Pt s = a[i], t = a[i + 1 == n ? 0 : i + 1];
            const float ds = ex * (s.y - q0.y) - ey * (s.x - q0.x), dt = ex * (t.y - q0.y) - ey * (t.x - q0.x);     // >= 0: inside (left of the edge)
            if (ds >= 0.f) b[m++] = s;
            if ((ds >= 0.f) != (dt >= 0.f)) {
                const float u = ds / (ds - dt);
                b[m++] = Pt{s.x + u * (t.x - s.x), s.y + u * (t.y - s.y)};
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) a[i] = b[i];
    }
    float area = 0.f;
    for (int i = 0; i < n; ++i) {
        const Pt s = a[i], t = a[i + 1 == n ? 0 : i + 1];
        area += s.x * t.y - t.x * s.y;
    }
    return 0.5f * fabsf(area);
}

__global__ __launch_bounds__(1024) void nms_bev_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                       const long long* __restrict__ labels, const int* __restrict__ count, float thr,
                                                       float* __restrict__ scores_out, int in_stride) {
    __shared__ float sc[1024], area[1024];
    __shared__ int lab[1024], order[1024];
    __shared__ unsigned char dead[1024];
    __shared__ Pt cor[1024][4];
    const int b = blockIdx.x, t = threadIdx.x;
    boxes += (long long)b * in_stride * 9; scores += (long long)b * in_stride; labels += (long long)b * in_stride; scores_out += (long long)b * in_stride;
    const int n = min(count[b], 1024);
    if (t < n) {
        sc[t] = scores[t]; lab[t] = (int)labels[t]; dead[t] = 0;
        Pt c[4];
        box_corners(boxes + (long long)t * 9, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) cor[t][i] = c[i];
        area[t] = fabsf(boxes[(long long)t * 9 + 3] * boxes[(long long)t * 9 + 4]);
    }
    __syncthreads();
    if (t < n) {                                                      // rank: class ascending, score descending, index ascending
        int r = 0;
        for (int j = 0; j < n; ++j)
            r += (lab[j] < lab[t]) || (lab[j] == lab[t] && (sc[j] > sc[t] || (sc[j] == sc[t] && j < t)));
        order[r] = t;
    }
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const int bi = order[i];
        if (!dead[bi]) {                                              // (block-uniform: dead[] is only written between barriers)
            for (int k = i + 1 + t; k < n; k += 1024) {
                const int bj = order[k];
                if (lab[bj] != lab[bi] || dead[bj]) continue;
                Pt P[4], Q[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { P[c] = cor[bj][c]; Q[c] = cor[bi][c]; }
                const float inter = clip_area(P, Q);
                const float iou = inter / fmaxf(area[bi] + area[bj] - inter, 1e-8f);
                if (iou > thr) dead[bj] = 1;
            }
        }
        __syncthreads();
    }
    if (t < n) scores_out[t] = dead[t] ? -INFINITY : sc[t];
}

}  // namespace

extern "C" int mv2d_nms_bev(const float* boxes, const float* scores, const long long* labels, const int* count, float nms_thr, float* scores_out,
                            int n_samples, int in_stride, void* stream) {
    MV2D_CHECK_ARG(boxes && scores && labels && count && scores_out && n_samples >= 1 && in_stride >= 1 && in_stride <= 1024,
                   "mv2d_nms_bev: bad args (at most 1024 boxes per sample)");
    hipLaunchKernelGGL(nms_bev_kernel, dim3(n_samples), dim3(1024), 0, (hipStream_t)stream, boxes, scores, labels, count, nms_thr, scores_out, in_stride);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
